#!/usr/bin/env python3
"""What folding BatchNorm into the training step's forward GEMMs buys, layer by layer (batch-sized launches, replayed from a
hipGraph so that host time is out of the picture): lcrec_linear_forward + lcrec_bn_relu_forward (two launches, the
activation written and read back) against lcrec_linear_bn_forward with statistics only / input fold only / both.
    python tools/fused_fwd_probe.py [--rows 1024] [--in_dim 768]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_gemm_probe import timed  # noqa: E402

HID = [2048, 1024, 512, 256, 128, 64, 32]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--in_dim", type=int, default=768)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = [a.in_dim] + HID
    m = a.rows
    tot = [0.0] * 6
    for l in range(len(dims) - 1):
        k, n = dims[l], dims[l + 1]
        x = torch.randn((m, k), device=dev)
        w = torch.randn((n, k), device=dev) * 0.03
        b = torch.zeros(n, device=dev)
        sc, sh = torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev) * 0.1
        g, be = torch.ones(n, device=dev), torch.zeros(n, device=dev)
        rm, rv = torch.zeros(n, device=dev), torch.ones(n, device=dev)
        t_lin = timed(lambda: ops.linear_forward(x, w, b, relu=False))
        y = ops.linear_forward(x, w, b, relu=False)
        t_bn = timed(lambda: ops.bn_relu_forward(y, g, be, 1e-5, 0.1, rm, rv, relu=True))
        t_s = timed(lambda: ops.linear_bn_forward(x, w, b, bn=(g, be, 1e-5, 0.1, rm, rv)))
        t_p = timed(lambda: ops.linear_bn_forward(x, w, b, in_fold=(sc, sh)))
        t_ps = timed(lambda: ops.linear_bn_forward(x, w, b, in_fold=(sc, sh), bn=(g, be, 1e-5, 0.1, rm, rv)))
        for i, v in enumerate((t_lin, t_bn, t_lin + t_bn, t_s, t_p, t_ps)):
            tot[i] += v
        print(f"  {k:5d} -> {n:5d}: linear {t_lin:6.1f} + bn {t_bn:5.1f} = {t_lin + t_bn:6.1f} us | fused: stats {t_s:6.1f}  fold {t_p:6.1f}  both {t_ps:6.1f}",
              flush=True)
    print(f"  total: linear {tot[0]:.0f} + bn {tot[1]:.0f} = {tot[2]:.0f} us | fused: stats {tot[3]:.0f}  fold {tot[4]:.0f}  both {tot[5]:.0f}")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
