"""Round-5 call M (the last GPU seconds): what the opt-in first-use selection (hgemm_mi355x_set_insitu) picks on this box for a few grid rows --
candidates (the table's plan first), the choice, and the device time of a following call.  python cuda-l2_amd/tools/lab/insitu_demo.py > out.jsonl"""
import ctypes
import json
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[3]
sys.path[:0] = [str(REPO), str(REPO / "cuda-l2_amd")]
import bench  # noqa: E402

L = bench.load_library()
L.hgemm_mi355x_config_name.restype = ctypes.c_char_p
SHAPES = ["256_16384_16384", "16384_256_16384", "12288_12288_256", "8192_8192_256", "16384_128_16384", "512_4096_4096", "4096_4096_4096", "64_64_8192", "2048_8192_8192", "12288_4096_128"]


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


for mnk in SHAPES:
    m, n, k = map(int, mnk.split("_"))
    a = torch.randn(m, k, dtype=torch.half, device="cuda"); b = torch.randn(k, n, dtype=torch.half, device="cuda"); bt = b.t().contiguous()
    c = torch.empty(m, n, dtype=torch.half, device="cuda")
    L.hgemm_mi355x_set_insitu(0)
    table_us = call_us(a, b, bt, c, m, n, k)
    L.hgemm_mi355x_set_insitu(1)
    chosen_us = call_us(a, b, bt, c, m, n, k)
    cfg, sp, gm = (ctypes.c_int * 3)(), (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
    nc = L.hgemm_mi355x_insitu_candidates(m, n, k, cfg, sp, gm)
    c0, s0, g0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0))
    print(json.dumps({"mnk": mnk, "candidates": [[name(cfg[i]), sp[i], gm[i]] for i in range(nc)], "choice": [name(c0.value), s0.value, g0.value],
                      "kept_the_table_plan": (c0.value, s0.value, g0.value) == (cfg[0], sp[0], gm[0]), "table_plan_us": round(table_us, 2), "after_selection_us": round(chosen_us, 2)}), flush=True)
    L.hgemm_mi355x_set_insitu(0)
