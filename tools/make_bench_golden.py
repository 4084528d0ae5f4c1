#!/usr/bin/env python
"""First-step loss of bench.py's exact N = 1 setup (seed-333 weights, seed-333 synthetic fields, area-weighted MSE): the bf16 /
engine value the benchmark computes and the fp32 CPU oracle's value for the same weights and fields.  Writes
gpurun_out/bench_first_loss.json; the reviewed copy lives in tests/golden/ (bench.py compares its first step with it,
tests/test_parity_gpu.py::test_bench_first_step_loss checks both numbers again)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    vals = bench.first_step_losses(with_oracle=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_first_loss.json"), "w") as f:
        json.dump(vals, f, indent=1)
    print(json.dumps(vals))


if __name__ == "__main__":
    main()
