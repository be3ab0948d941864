#!/usr/bin/env python3
"""Time / profile the encoder GEMM kernel alone on the C3 layer shapes.

    python tools/gemm_probe.py [--reps 20] [--layers 0 1 2]      # prints TFLOP/s per layer (HIP events)
    rocprofv3 --pmc ... -- python3 tools/gemm_probe.py --reps 3   # counters for the same launches
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

DIMS = [768, 2048, 1024, 512, 256, 128, 64, 32]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--layers", type=int, nargs="+", default=[0, 1, 2, 3, 4])
    ap.add_argument("--in_dim", type=int, default=768)
    a = ap.parse_args()
    dims = [a.in_dim] + DIMS[1:]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    for l in a.layers:
        k, n = dims[l], dims[l + 1]
        x = torch.randn((a.rows, k), generator=g, device=dev)
        w = torch.randn((n, k), generator=g, device=dev) * (2.0 / (k + n)) ** 0.5
        b = torch.zeros(n, device=dev)
        for _ in range(3):
            ops.linear_forward(x, w, b, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            ops.linear_forward(x, w, b, relu=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print(f"layer {l}: {k:5d} -> {n:5d}  rows {a.rows}  {ms * 1e3:9.1f} us  {2.0 * a.rows * k * n / ms / 1e9:7.1f} TFLOP/s",
              flush=True)


if __name__ == "__main__":
    main()
