"""Round-5 call A, part 3: fabric traffic / L2 hit rate of the compute-bound plans against K -> one JSON.
Input: gpurun_out/r5a/pmc_k (three --pmc passes of `hgemm_tune bench --shapes LIST --lib --reps 3`) and pmc_hipblaslt/."""
import csv, glob, json, sys, collections

SHAPES = ["4096_4096_4096", "4096_4096_8192", "4096_4096_16384", "8192_8192_8192", "12288_16384_8192", "16384_16384_16384"]


def per_dispatch(root, want):
    out = collections.defaultdict(dict)
    for f in glob.glob(f"{root}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if want(r["Kernel_Name"]):
                out[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                out[int(r["Dispatch_Id"])]["kernel"] = r["Kernel_Name"].split("(")[0][-70:]
    return out


def floor_bytes(m, n, k, bm, bn, gm=4, per_xcd=32):
    # every XCD (private L2) runs patches of per_xcd tiles, gm tile rows x per_xcd/gm tile columns: each operand panel of a patch is
    # fetched once per XCD and patch; C is written once
    tiles = -(-m // bm) * -(-n // bn)
    patches = -(-tiles // per_xcd)
    panel = (gm * bm + (per_xcd // gm) * bn) * k * 2
    return patches * panel + 2 * m * n


def main(root="gpurun_out/r5a"):
    res = {"note": "bytes = 2 x FETCH_SIZE KiB (gfx950 tallies 128-B reads at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE KiB; floor = every operand panel "
                   "of an 8 x 4 tile patch fetched exactly once per XCD and patch (8 private L2s) + C once; hit rate = 1 - TCC_MISS / TCC_REQ", "ours": {}, "hipblaslt": {}}
    passes = [per_dispatch(f"{root}/pmc_k/pass{p}", lambda k: "hgemm_tn" in k) for p in range(3)]
    ids = sorted(passes[0])
    groups, cur = [], []
    for i in ids:   # consecutive dispatch ids = one shape's launches (operand fills sit between shapes)
        if cur and i != cur[-1] + 1:
            groups.append(cur); cur = []
        cur.append(i)
    groups.append(cur)
    assert len(groups) == len(SHAPES), (len(groups), groups)
    for mnk, g in zip(SHAPES, groups):
        m, n, k = map(int, mnk.split("_"))
        mean = lambda p, c: sum(passes[p][i][c] for i in g) / len(g)
        rd = 2 * mean(0, "FETCH_SIZE") * 1024
        wr = mean(1, "WRITE_SIZE") * 1024
        kern = passes[0][g[0]]["kernel"]
        bm = 192 if "192, 256" in kern else 256
        alg = 2 * (m * k + n * k + m * n)
        fl = floor_bytes(m, n, k, bm, 256)
        res["ours"][mnk] = {"kernel": kern, "launches": len(g), "read_bytes": round(rd), "write_bytes": round(wr), "traffic_over_algorithmic": round((rd + wr) / alg, 2),
                            "floor_over_algorithmic": round(fl / alg, 2), "traffic_over_floor": round((rd + wr) / fl, 3),
                            "l2_hit_rate": round(1 - mean(1, "TCC_MISS_sum") / mean(1, "TCC_REQ_sum"), 3),
                            "ea_rdreq_per_launch": round(mean(2, "TCC_EA0_RDREQ_sum")), "ea_wrreq_64B_per_launch": round(mean(2, "TCC_EA0_WRREQ_64B_sum"))}
    for mnk in ("8192_8192_8192", "16384_16384_16384"):
        m, n, k = map(int, mnk.split("_"))
        p0 = per_dispatch(f"{root}/pmc_hipblaslt/{mnk}_pass0", lambda k: "Cijk" in k)
        p1 = per_dispatch(f"{root}/pmc_hipblaslt/{mnk}_pass1", lambda k: "Cijk" in k)
        if not p0 or not p1:
            continue
        rd = 2 * sum(v["FETCH_SIZE"] for v in p0.values()) / len(p0) * 1024
        wr = sum(v["WRITE_SIZE"] for v in p1.values()) / len(p1) * 1024
        alg = 2 * (m * k + n * k + m * n)
        res["hipblaslt"][mnk] = {"kernel": next(iter(p0.values()))["kernel"], "launches": len(p0), "read_bytes": round(rd), "write_bytes": round(wr),
                                 "traffic_over_algorithmic": round((rd + wr) / alg, 2),
                                 "l2_hit_rate": round(1 - sum(v["TCC_MISS_sum"] for v in p1.values()) / sum(v["TCC_REQ_sum"] for v in p1.values()), 3)}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
