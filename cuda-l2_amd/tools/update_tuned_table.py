"""Apply a targeted re-tune to the shipped tuned table IN PLACE: rows of shapes the re-tune does not improve stay byte-identical.

    python tools/update_tuned_table.py --verified parity.jsonl [--min-gain 1.015] retune1.jsonl [retune2.jsonl ...]

Each re-tune record (bin/hgemm_tune tune --cand-file ...) holds the SHIPPED plan measured beside the new candidates in the same run
on the same box.  A shape's row is replaced when the fastest candidate that has a passing oracle record (tests/tools/verify_plans.py
--plans; the reference checks every shape before it benchmarks it, eval_one_file.sh:71-80) beats the shipped plan's own figure of
that run by at least --min-gain (ranking figure of the run: sqrt(isolated x back-to-back) with --rank both), or when the shipped
plan fell out of the run (slower than 1.25x the best).  Several files are applied in order.  The per-shape kernel files of
changed rows are rewritten (tools/gen_shape_kernels.py).
"""
from __future__ import annotations

import argparse
import json
import re
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(PKG_DIR))
from tools.gen_shape_kernels import ACC_DIRS, write_shape_file  # noqa: E402

TABLE = PKG_DIR / "csrc" / "hgemm_tuned_table.inc"
ROW = re.compile(r'\s*\{(\d+), (\d+), (\d+), "([^"]+)", (\d+), (\d+)\},(.*)')


FLAG_TEXT = ((0x80000, "K stagger per XCD"), (0x100000, "NT loads of the streamed operand"), (0x200000, "phase offset"),
             (0x800000, "phase offset x4"), (0x400000, "wave priority"))


def form_text(splits: int) -> str:
    """Row annotation: split-K form + the plan flags that change the schedule (include/hgemm_mi355x.h; NT stores are in the
    number itself, 0x20000, as before)."""
    flags = "".join(f", {t}" for bit, t in FLAG_TEXT if splits & bit)
    if (splits & 0xA00000) == 0xA00000:
        flags = flags.replace(", phase offset, phase offset x4", ", phase offset x8")
    if splits & 0x40000:
        return f" stream-K, {splits & 0xFFFF or 'one wave of'} workgroups" + flags
    if (splits & 0xFFFF) == 1:
        return flags.replace(", ", " ", 1)
    return (" fused split-K" if splits & 0x10000 else " two-pass split-K") + flags


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("runs", nargs="+")
    ap.add_argument("--verified", required=True)
    ap.add_argument("--min-gain", type=float, default=1.015)
    ap.add_argument("--no-shape-files", action="store_true")
    ap.add_argument("--require-shipped", action="store_true", help="only touch shapes whose shipped plan was measured in the same run (a run "
                    "that times one family's plans says nothing about a row that ships another family's)")
    ap.add_argument("--report", default="", help="write the list of changed rows (JSON lines)")
    ap.add_argument("--round", default="", help='provenance label of the regenerated kernel files, e.g. "round 6" (default: taken from the '
                    "first run file's name, rNN_... -> round NN; ADVICE r5: the label used to be hard-coded)")
    ap.add_argument("--stability", action="append", default=[], help="a SECOND measurement of the same candidates (another pass / box): a pick is "
                    "adopted only if it is in this run too, its two times agree within --stability-tol, and it still beats the shipped plan "
                    "there by --min-gain (VERDICT r5 item 5: 2048x1024x4096 read 26 us in the tune passes and 36 us in the closing run)")
    ap.add_argument("--stability-tol", type=float, default=0.10)
    ap.add_argument("--exclude", action="append", default=[], help="M_N_K:config:splits -- a candidate not to adopt although it measured fastest "
                    "(e.g. a one-tile-per-CU lock-step plan whose time is known to depend on the box, DESIGN.md section 4.12)")
    a = ap.parse_args(argv)
    if not a.round:
        m0 = re.match(r"r(\d+)_", Path(a.runs[0]).name)
        a.round = f"round {int(m0[1])}" if m0 else "round ?"
    second = {}   # (mnk, config, splits, group_m) -> us in the stability run(s) (the slowest reading counts)
    for path in a.stability:
        for line in open(path):
            rec = json.loads(line)
            for c in rec["candidates"]:
                key = (rec["mnk"], c["config"], int(c["splits"]), int(c["group_m"]))
                second[key] = max(second.get(key, 0.0), c["us"])
    unstable = []
    ok = set()
    for line in open(a.verified):
        r = json.loads(line)
        if r.get("pass") and r.get("bitwise_equal_unmasked", True):
            p = r["plan"]
            ok.add((r["mnk"], p["config"], int(p["splits"]), int(p["group_m"])))
    lines = TABLE.read_text().splitlines()
    index = {}
    for i, ln in enumerate(lines):
        m = ROW.match(ln)
        if m:
            index[f"{m[1]}_{m[2]}_{m[3]}"] = i
    changed = []
    for path in a.runs:
        for line in open(path):
            rec = json.loads(line)
            mnk = rec["mnk"]
            if mnk not in index:
                continue
            m = ROW.match(lines[index[mnk]])
            shipped = (m[4], int(m[5]), int(m[6]))
            cands = sorted(rec["candidates"], key=lambda c: c["us"])
            ship_us = next((c["us"] for c in cands if (c["config"], int(c["splits"]), int(c["group_m"])) == shipped), None)
            if ship_us is None:   # the same geometry and form with another raster group stands in for it (one tile row: the group is moot)
                ship_us = next((c["us"] for c in cands if (c["config"], int(c["splits"])) == shipped[:2]), None)
            barred = {tuple(x.split(":")) for x in a.exclude}
            pick = next((c for c in cands if (mnk, c["config"], int(c["splits"]), int(c["group_m"])) in ok
                         and (mnk, c["config"], str(int(c["splits"]))) not in barred), None)
            if pick is None or (pick["config"], int(pick["splits"]), int(pick["group_m"])) == shipped:
                continue
            if ship_us is None and a.require_shipped:
                continue
            if ship_us is not None and ship_us < pick["us"] * a.min_gain:
                continue
            if a.stability:
                key = (mnk, pick["config"], int(pick["splits"]), int(pick["group_m"]))
                us2 = second.get(key)
                ship2 = second.get((mnk,) + shipped)
                if us2 is None or abs(us2 - pick["us"]) > a.stability_tol * min(us2, pick["us"]) or (ship2 is not None and ship2 < us2 * a.min_gain):
                    unstable.append({"mnk": mnk, "config": pick["config"], "splits": int(pick["splits"]), "us": pick["us"], "us_second": us2,
                                     "shipped_us_second": ship2})
                    continue
            mm, nn, kk = map(int, mnk.split("_"))
            iso = pick.get("isolated_us", pick["us"])
            note = f"{iso:.1f} us, {2.0 * mm * nn * kk / iso * 1e-6:.1f} TFLOP/s{form_text(int(pick['splits']))}"
            if "stream_us" in pick:
                note += f" (back to back {pick['stream_us']:.1f} us)"
            lines[index[mnk]] = f'    {{{mm}, {nn}, {kk}, "{pick["config"]}", {int(pick["splits"])}, {int(pick["group_m"])}}},  // {note}'
            changed.append({"mnk": mnk, "from": {"config": shipped[0], "splits": shipped[1], "group_m": shipped[2], "us": ship_us},
                            "to": {"config": pick["config"], "splits": int(pick["splits"]), "group_m": int(pick["group_m"]), "us": pick["us"],
                                   "isolated_us": pick.get("isolated_us"), "stream_us": pick.get("stream_us")}, "run": Path(path).name})
            if not a.no_shape_files:
                for acc in ACC_DIRS:
                    write_shape_file(mnk, acc, plan=(pick["config"], int(pick["splits"]), int(pick["group_m"])),
                                     source=f"tuned on MI355X ({a.round}): {note}, verified against the CPU oracle")
    head = [ln for ln in lines if not ROW.match(ln)]
    text = "\n".join(lines) + "\n"
    text = text.replace("split-K (| 0x10000 = single-launch form, | 0x20000 = non-temporal C stores), raster group}",
                        "split-K (| 0x10000 = single-launch form, | 0x20000 = non-temporal C stores, | 0x40000 = stream-K: the count is the number of workgroups), raster group}")
    TABLE.write_text(text)
    if a.report:
        with open(a.report, "w") as f:
            for c in changed:
                f.write(json.dumps(c) + "\n")
    if a.stability:
        print(f"stability gate: {len(unstable)} picks rejected (absent from / unstable in / not winning in the second measurement)")
        for u in unstable[:40]:
            print("  rejected", json.dumps(u))
    print(f"{len(changed)} rows changed of {len(index)}; header lines {len(head)}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
