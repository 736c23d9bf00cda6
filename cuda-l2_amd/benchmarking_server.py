"""Server-mode benchmark: the offline loop plus Poisson arrivals.

As in the reference's benchmarking_server.py (:127-128, :144-145) every iteration of the warm-up and
of the benchmark loop is followed by time.sleep(Exp(1/target_qps)), so the GPU idles between requests
(clocks and caches cool down).  Output JSON is the offline one (mean TFLOPS per function) plus the
p50/p99 latency block BASELINE.json config 4 asks for.
"""
import time

import numpy as np

import benchmarking_offline as offline

offline.MODE_NAME = "Server"


def build_arg_parser():
    parser = offline.build_arg_parser()
    parser.add_argument("--target_qps", type=float, required=True)
    return parser


def _sleep(args) -> None:
    time.sleep(np.random.exponential(1.0 / args.target_qps))


offline.inter_arrival_sleep = _sleep


def main(argv=None):
    args = build_arg_parser().parse_args(argv)
    if args.target_qps <= 0:
        raise SystemExit("--target_qps must be positive")
    return offline.run(args, extra={"target_qps": args.target_qps})


if __name__ == "__main__":
    main()
