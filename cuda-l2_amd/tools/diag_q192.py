"""Diagnosis of the withdrawn 192 x 192 member of family q: where are the wrong elements, and which
K positions do they miss or count twice?

    HGEMM_LIB_SUFFIX=q192 HGEMM_EXTRA_HIPFLAGS=-DHGEMM_EXPERIMENT_Q192X192 python build.py      (-> lib_q192/)
    HGEMM_LIB_DIR=cuda-l2_amd/lib_q192 python cuda-l2_amd/tools/diag_q192.py > diag.jsonl        (on an MI355X)

Probe 1: 0/1 operands against the exact integer result: wrong elements by output tile, wave quadrant and bounding box.
Probe 2: A = ones, B = ones on ONE 8-element K chunk: every output must be 8; a wrong element tells which chunk its dot
product lost (0), doubled (16) or took from elsewhere.  Shapes: the failing one (three K-steps), the same M, N with other K-step
counts, and exact multiples of the tile.  Other configurations run the same probes as controls."""
import ctypes
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

PKG = Path(__file__).resolve().parent.parent


def main() -> int:
    so = Path(os.environ.get("HGEMM_LIB_DIR", PKG / "lib")) / "libhgemm_mi355x.so"
    L = ctypes.CDLL(str(so))
    L.hgemm_mi355x_config_by_name.argtypes = [ctypes.c_char_p]
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.hgemm_mi355x_launch.argtypes = [ci, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]

    def run(cfg, a, b, splits=1, group=1):
        m, k = a.shape
        n = b.shape[1]
        bt = b.t().contiguous()
        c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")
        st = L.hgemm_mi355x_launch(cfg, splits, group, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n, None)
        torch.cuda.synchronize()
        return st, c

    def describe(bad, bm, bn, tm, tn):
        idx = bad.nonzero()
        if idx.numel() == 0:
            return {"wrong": 0}
        r, c = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
        tiles = {}
        for tr, tc in zip(r // bm, c // bn):
            tiles[(int(tr), int(tc))] = tiles.get((int(tr), int(tc)), 0) + 1
        quads = {}
        for qr, qc in zip((r % bm) // tm, (c % bn) // tn):
            quads[f"{int(qr)}{int(qc)}"] = quads.get(f"{int(qr)}{int(qc)}", 0) + 1
        return {"wrong": int(idx.shape[0]), "rows": [int(r.min()), int(r.max())], "cols": [int(c.min()), int(c.max())],
                "rows_in_tile": sorted(set((r % bm).tolist()))[:6] + ["..."] + sorted(set((r % bm).tolist()))[-3:],
                "cols_in_tile": sorted(set((c % bn).tolist()))[:6] + ["..."] + sorted(set((c % bn).tolist()))[-3:],
                "n_rows_in_tile": len(set((r % bm).tolist())), "n_cols_in_tile": len(set((c % bn).tolist())),
                "by_tile": {f"{k[0]},{k[1]}": v for k, v in sorted(tiles.items())}, "by_wave_quadrant": quads}

    torch.manual_seed(7)
    geos = {"q192x192_w2x2": (192, 192), "q192x256_w2x2": (192, 256), "q256x192_w2x2": (256, 192), "q256x256_w2x2": (256, 256)}
    for name, (bm, bn) in geos.items():
        cfg = L.hgemm_mi355x_config_by_name(name.encode())
        if cfg < 0:
            print(json.dumps({"config": name, "skipped": "not in this build"})); continue
        for (m, n, k) in [(1000, 520, 192), (1000, 520, 320), (1000, 520, 256), (960, 384, 192), (192, 192, 192), (384, 384, 192), (1000, 520, 64), (1000, 520, 448)]:
            a = (torch.rand((m, k), device="cuda") < 0.5).half()
            b = (torch.rand((k, n), device="cuda") < 0.5).half()
            st, c = run(cfg, a, b)
            ref = (a.float() @ b.float()).half()
            rec = {"config": name, "mnk": f"{m}_{n}_{k}", "probe": "zero_one", "status": st}
            rec.update(describe(c != ref, bm, bn, bm // 2, bn // 2))
            print(json.dumps(rec)); sys.stdout.flush()
            if name != "q192x192_w2x2" and rec.get("wrong", 0) == 0:
                continue
            if rec.get("wrong", 0) == 0 and (m, n, k) != (1000, 520, 192):
                continue
            # probe 2: one K chunk at a time
            a1 = torch.ones((m, k), dtype=torch.half, device="cuda")
            chunks = []
            for ch in range(k // 8):
                b1 = torch.zeros((k, n), dtype=torch.half, device="cuda")
                b1[ch * 8:(ch + 1) * 8, :] = 1
                st, c1 = run(cfg, a1, b1)
                bad = c1 != 8
                if bad.any():
                    vals, cnt = torch.unique(c1[bad].float(), return_counts=True)
                    d = describe(bad, bm, bn, bm // 2, bn // 2)
                    chunks.append({"chunk": ch, "k": [ch * 8, ch * 8 + 7], "values": {str(v.item()): int(x.item()) for v, x in zip(vals, cnt)},
                                   "wrong": d["wrong"], "rows": d["rows"], "cols": d["cols"], "by_tile": d["by_tile"], "by_wave_quadrant": d["by_wave_quadrant"],
                                   "n_rows_in_tile": d["n_rows_in_tile"], "n_cols_in_tile": d["n_cols_in_tile"], "rows_in_tile": d["rows_in_tile"], "cols_in_tile": d["cols_in_tile"]})
            print(json.dumps({"config": name, "mnk": f"{m}_{n}_{k}", "probe": "one_k_chunk", "bad_chunks": chunks})); sys.stdout.flush()
            # probe 3: one A row block / one B row block at a time is implied by the maps above
    return 0


if __name__ == "__main__":
    sys.exit(main())
