"""Build profiles/pmc_summary.json from the rocprofv3 --pmc passes written by tools/pmc_sweep.sh.

    python tools/pmc_summary.py PASS_ROOT --kernel sp_kernel --mnk 4096_4096_4096 --first 15 \
           --source "..." > ../profiles/pmc_summary.json

Every pass re-runs the same command, so the launches of the wanted kernel are taken in dispatch order
and the first --first of them (the launches of --mnk; later shapes of the same command are ignored)
are averaged per counter.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB and
wide coalesced reads are under-counted 2x on gfx950, so hbm = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import argparse
import collections
import csv
import glob
import json


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--mnk", required=True)
    ap.add_argument("--first", type=int, default=0, help="use only the first N launches of the kernel per pass (0 = all)")
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    m, n, k = (int(x) for x in a.mnk.split("_"))

    counters = {}
    kernel_name = None
    for f in sorted(glob.glob(f"{a.root}/pass*/**/*_counter_collection.csv", recursive=True)):
        per = collections.defaultdict(dict)  # dispatch -> counter -> value
        for r in csv.DictReader(open(f)):
            if a.kernel not in r["Kernel_Name"]:
                continue
            kernel_name = r["Kernel_Name"]
            d = per[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(per)
        if a.first:
            ids = ids[: a.first]
        names = {c for i in ids for c in per[i]}
        for c in names:
            v = [per[i][c] for i in ids if c in per[i]]
            counters[c] = sum(v) / len(v)
    durs = []
    for f in sorted(glob.glob(f"{a.root}/pass0/**/*_kernel_trace.csv", recursive=True)):
        rows = sorted((r for r in csv.DictReader(open(f)) if a.kernel in r["Kernel_Name"]), key=lambda r: int(r["Dispatch_Id"]))
        if a.first:
            rows = rows[: a.first]
        durs += [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]

    c = counters
    us = sum(durs) / max(1, len(durs))
    derived = {"avg_kernel_us_profiled": round(us, 1)}
    # GRBM_GUI_ACTIVE also counts the few microseconds of dispatch around the kernel: for short kernels the
    # clock derived from it is an upper bound and the MFMA-busy fraction a lower bound (suffix says so)
    sfx = "" if us >= 100.0 else "_short_kernel_bound"
    if "GRBM_GUI_ACTIVE" in c and us:
        derived["effective_clock_ghz" + sfx] = round(c["GRBM_GUI_ACTIVE"] / 8 / us * 1e-3, 3)  # summed over 8 XCDs
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        # MFMA-busy is summed over the 4 SIMDs of every CU; normalise by 1024 SIMDs x active cycles per XCD
        derived["mfma_pipe_busy_frac" + sfx] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and us:
        # same fraction against wall time at the 2.4 GHz peak clock (what the 2.5 PFLOP/s spec assumes)
        derived["mfma_busy_vs_peak_clock"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0), 4)
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        derived["wave_time_split"] = {k2: round(c[k1] / w, 3) for k1, k2 in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"),
                                                                             ("SQ_ACTIVE_INST_ANY", "active_inst_any")) if k1 in c}
    if "SQ_LDS_IDX_ACTIVE" in c and c.get("SQ_INSTS_LDS"):
        derived["lds_cycles_per_read"] = round(c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"], 2)
    if "SQ_LDS_BANK_CONFLICT" in c:
        derived["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
    if c.get("TCC_REQ_sum"):
        derived["l2_hit_rate"] = round(c["TCC_HIT_sum"] / c["TCC_REQ_sum"], 4)
    if c.get("TCP_TCC_READ_REQ_sum"):
        derived["tcp_to_tcc_avg_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
    out = {"source": a.source, "kernel": kernel_name, "launches_averaged": len(durs), "counters": c, "derived": derived}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["dominant_kernel"] = {
            "mnk": a.mnk,
            "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
            "algorithmic_bytes_per_launch": 2.0 * (m * k + n * k + m * n),
            "note": "hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE/WRITE_SIZE in KiB; gfx950 counts wide coalesced reads at half); "
                    "reads above the unique operand bytes are per-XCD L2 refetches of shared A/B panels, served mostly by the 256 MiB Infinity Cache",
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
