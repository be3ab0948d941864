#!/usr/bin/env python3
"""Per-layer time of the three GEMMs of a training step (forward, dX, dW) at batch-sized M, free of host launch overhead:
20 calls captured in a hipGraph, replayed.   python tools/train_gemm_probe.py [--rows 1024 2048] [--in_dim 768]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

HID = [2048, 1024, 512, 256, 128, 64, 32]


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3          # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[1024, 2048])
    ap.add_argument("--in_dim", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = [a.in_dim] + HID
    for m in a.rows:
        tot = {"fwd": 0.0, "dX": 0.0, "dW": 0.0}
        print(f"== batch {m}")
        for l in range(len(dims) - 1):
            k, n = dims[l], dims[l + 1]
            x = torch.randn((m, k), device=dev)
            w = torch.randn((n, k), device=dev) * 0.03
            b = torch.zeros(n, device=dev)
            gy = torch.randn((m, n), device=dev)
            t_f = timed(lambda: ops.linear_forward(x, w, b, relu=True))
            if n % 32 == 0:
                t_x = timed(lambda: ops.linear_backward(gy, x, w, True, False))
                t_w = timed(lambda: ops.linear_backward(gy, x, w, False, True))
            else:
                t_x = t_w = float("nan")
            gf = 2.0 * m * k * n / 1e6                      # MFLOP -> us * TFLOP/s
            tot["fwd"] += t_f; tot["dX"] += t_x; tot["dW"] += t_w
            print(f"  {k:5d} -> {n:5d}: fwd {t_f:6.1f} us {gf / t_f:6.1f} TF | dX {t_x:6.1f} us {gf / t_x:6.1f} TF | dW {t_w:6.1f} us {gf / t_w:6.1f} TF"
                  f"  (dW splits {ops.linear_backward_splits(m, k, n)})", flush=True)
        print(f"  encoder-shaped MLP total: fwd {tot['fwd']:.0f} us, dX {tot['dX']:.0f} us, dW {tot['dW']:.0f} us")


if __name__ == "__main__":
    main()
