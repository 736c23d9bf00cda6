"""Merge the benchmark_result_*.json files of one shape into the 10-row speedup table.

Row rule kept from the reference (summarize_result.py:22-79): for each of rocBLAS, hipBLASLt-heuristic,
hipBLASLt-auto-tuning the "-max" row is the tn/nn variant against which cuda_l2's speedup is LOWER
(i.e. the stronger baseline).  Display names change with the libraries: cuBLAS -> rocBLAS,
cuBLASLt -> hipBLASLt.  Also writes {base_dir}/summary.json for the sweep driver.
"""
import argparse
import json
from pathlib import Path

import pandas

from harness_common import cuda_l2_name
from tools.utils import DEVICE_TYPES

NAME_ORDER = [
    "torch.matmul",
    "rocBLAS-tn", "rocBLAS-nn", "rocBLAS-max",
    "hipBLASLt-heuristic-tn", "hipBLASLt-heuristic-nn", "hipBLASLt-heuristic-max",
    "hipBLASLt-auto-tuning-tn", "hipBLASLt-auto-tuning-nn", "hipBLASLt-auto-tuning-max",
]


def show_name(method: str) -> str:
    if method == "matmul":
        return "torch.matmul"
    return method.replace("hgemm_", "").replace("cublaslt", "hipBLASLt").replace("cublas", "rocBLAS").replace("_", "-")


def summarize(base_dir: Path, func_name: str) -> list[dict]:
    rows = {}
    for file in sorted(base_dir.glob("benchmark_result_*.json")):
        method = file.stem.replace("benchmark_result_", "")
        rec = json.loads(file.read_text())["records"]
        if func_name not in rec:   # e.g. a --device cpu plumbing result
            continue
        rows[show_name(method)] = {
            "Baseline Method Name": show_name(method),
            "Baseline TFLOPS": rec[method],
            "CUDA-L2 TFLOPS": rec[func_name],
            "Speedup": rec[func_name] / rec[method],
        }
    for family in ("rocBLAS", "hipBLASLt-heuristic", "hipBLASLt-auto-tuning"):
        tn, nn = rows.get(f"{family}-tn"), rows.get(f"{family}-nn")
        if tn is None or nn is None:
            continue
        worst = tn if tn["Speedup"] < nn["Speedup"] else nn
        rows[f"{family}-max"] = dict(worst, **{"Baseline Method Name": f"{family}-max"})
    return [rows[name] for name in NAME_ORDER if name in rows]


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__)
    parser.add_argument("--base_dir", type=str, required=True)
    parser.add_argument("--acc_precise", type=str, required=True, choices=["fp16", "fp32"])
    parser.add_argument("--device_type", type=str, required=True, choices=DEVICE_TYPES)
    args = parser.parse_args(argv)
    data = summarize(Path(args.base_dir), cuda_l2_name(args.device_type, args.acc_precise))
    print("Summary of Benchmark Results:")
    print(pandas.DataFrame.from_records(data).to_markdown(floatfmt=".3f", missingval="-"))
    (Path(args.base_dir) / "summary.json").write_text(json.dumps(data, indent=1))
    return data


if __name__ == "__main__":
    main()
