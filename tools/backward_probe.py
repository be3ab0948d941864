#!/usr/bin/env python3
"""Backward products of one Linear layer at training-batch size: lcrec_linear_backward (operands read
as stored) against the forward kernel on explicitly transposed copies (what the first version did).

    python tools/backward_probe.py [--batch 2048] [--in_dim 768]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--in_dim", type=int, default=768)
    a = ap.parse_args()
    dims = [a.in_dim, 2048, 1024, 512, 256, 128, 64, 32]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    tot = [0.0] * 4
    for l in range(len(dims) - 1):
        k, out = dims[l], dims[l + 1]
        x = torch.randn((a.batch, k), generator=g, device=dev)
        w = torch.randn((out, k), generator=g, device=dev) * 0.02
        gy = torch.randn((a.batch, out), generator=g, device=dev)
        t_gx = timed(lambda: ops.linear_backward(gy, x, w, True, False))
        t_gw = timed(lambda: ops.linear_backward(gy, x, w, False, True))
        o_gx = timed(lambda: ops.linear_forward(gy, w.t().contiguous()))
        o_gw = timed(lambda: ops.linear_forward(gy.t().contiguous(), x.t().contiguous()))
        for i, v in enumerate((t_gx, t_gw, o_gx, o_gw)):
            tot[i] += v
        print(f"layer {k:5d} -> {out:5d}: gx {t_gx:7.1f} us (copies+forward kernel {o_gx:7.1f})   gw {t_gw:7.1f} us ({o_gw:7.1f})", flush=True)
    print(f"sum: gx {tot[0]:.0f} us ({tot[2]:.0f})   gw {tot[1]:.0f} us ({tot[3]:.0f})")


if __name__ == "__main__":
    main()
