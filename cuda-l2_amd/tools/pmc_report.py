"""Summarise rocprofv3 --pmc passes written by tools/pmc_sweep.sh: mean counter value per kernel."""
import collections
import csv
import glob
import sys


def main(root: str, match: str = "") -> None:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(f"{root}/pass*/**/*_counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(glob.glob(f"{root}/pass0/**/*_kernel_trace.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, counters in agg.items():
        d = dur.get(k, [])
        print(f"== {k[:110]}  calls={len(d)} avg_us={sum(d) / max(1, len(d)):.1f}")
        for c, v in sorted(counters.items()):
            print(f"   {c:40s} {sum(v) / len(v):14.5g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
