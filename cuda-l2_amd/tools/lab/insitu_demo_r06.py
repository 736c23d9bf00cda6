"""Round 6: what the first-use selection (HGEMM_MI355X_INSITU / hgemm_mi355x_set_insitu) does on THIS box for the part of the table that moves with the box
(DESIGN.md section 6.8): 64 rows -- every 4th skinny row, every 4th mid-class row with K >= 4096, the ten rows of round 5's demo -- each at the table's plan
(selection off) and after the selection (on), three interleaved repetitions of six launches, median of the medians.
    python cuda-l2_amd/tools/lab/insitu_demo_r06.py [SHAPES.txt] > gpurun_out/insitu_demo_r06.jsonl"""
import ctypes
import json
import re
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[3]
sys.path[:0] = [str(REPO), str(REPO / "cuda-l2_amd")]
import bench  # noqa: E402

L = bench.load_library()
L.hgemm_mi355x_config_name.restype = ctypes.c_char_p
OLD = ["256_16384_16384", "16384_256_16384", "12288_12288_256", "8192_8192_256", "16384_128_16384", "512_4096_4096", "4096_4096_4096", "64_64_8192", "2048_8192_8192", "12288_4096_128"]


def shapes():
    rows = [tuple(map(int, m.groups())) for m in re.finditer(r"\{(\d+), (\d+), (\d+), \"", (REPO / "cuda-l2_amd" / "csrc" / "hgemm_tuned_table.inc").read_text())]
    skinny = [r for r in rows if min(r[0], r[1]) <= 256 and max(r[0], r[1]) >= 4096 and r[2] >= 2048]
    mid = [r for r in rows if 5e9 <= 2.0 * r[0] * r[1] * r[2] < 3e11 and r[2] >= 4096 and r not in skinny]
    out = list(OLD)
    for r in skinny[::4] + mid[::4]:
        s = "_".join(map(str, r))
        if s not in out:
            out.append(s)
    return out[:64]


def name(c):
    return L.hgemm_mi355x_config_name(c).decode() if c >= 0 else "ragged"


def call_us(a, b, bt, c, m, n, k, reps=6):
    st = torch.cuda.current_stream().cuda_stream
    L.hgemm_mi355x_fp32(a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); L.hgemm_mi355x_fp32(a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, st); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def shape_list():
    if len(sys.argv) > 1:      # a shape file (e.g. tools/offgrid_shapes.txt: off the grid the candidates are the runners-up among the corner plans)
        return [ln.strip() for ln in open(sys.argv[1]) if ln.strip() and not ln.startswith("#")]
    return shapes()


for mnk in shape_list():
    m, n, k = map(int, mnk.split("_"))
    a = torch.randn(m, k, dtype=torch.half, device="cuda"); b = torch.randn(k, n, dtype=torch.half, device="cuda"); bt = b.t().contiguous()
    c = torch.empty(m, n, dtype=torch.half, device="cuda")
    L.hgemm_mi355x_set_insitu(0)
    L.hgemm_mi355x_set_insitu(1)
    call_us(a, b, bt, c, m, n, k, reps=1)                      # the first call selects (kept for the process: set_insitu(1) again does not forget it)
    cfg, sp, gm = (ctypes.c_int * 3)(), (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
    nc = L.hgemm_mi355x_insitu_candidates(m, n, k, cfg, sp, gm)
    c0, s0, g0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0))
    st = torch.cuda.current_stream().cuda_stream
    L.hgemm_mi355x_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]

    def plan_us(p, reps=6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.hgemm_mi355x_launch(p[0], p[1], p[2], a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n, st)
        ts = []
        for _ in range(reps):
            e0.record(); L.hgemm_mi355x_launch(p[0], p[1], p[2], a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n, st); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    table, chosen = (cfg[0], sp[0], gm[0]), (c0.value, s0.value, g0.value)
    t_tab, t_cho = [], []
    for _ in range(3):                                          # interleaved: table plan, chosen plan, table plan, ...
        t_tab.append(plan_us(table)); t_cho.append(plan_us(chosen))
    print(json.dumps({"mnk": mnk, "candidates": [[name(cfg[i]), sp[i], gm[i]] for i in range(nc)], "choice": [name(chosen[0]), chosen[1], chosen[2]],
                      "kept_the_table_plan": chosen == table, "table_plan_us": round(sorted(t_tab)[1], 2), "chosen_plan_us": round(sorted(t_cho)[1], 2)}), flush=True)
    L.hgemm_mi355x_set_insitu(0)
