#!/usr/bin/env python3
"""BatchNorm(+ReLU) forward / backward per layer width of the run.sh MLP at one batch size, replayed from a hipGraph (40 calls
per replay) so the figure is the kernel's, not the Python launch's.   LCREC_BN_V4=0|8|16|32 python tools/bn_probe.py [--n 1024]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
tot_f = tot_b = 0.0
for F in (2048, 1024, 512, 256, 128, 64):
    t = torch.randn(a.n, F, device=dev, generator=g)
    gy = torch.randn(a.n, F, device=dev, generator=g)
    gamma, beta = torch.ones(F, device=dev), torch.zeros(F, device=dev)
    rm, rv = torch.zeros(F, device=dev), torch.ones(F, device=dev)
    y, mean, rstd = ops.bn_relu_forward(t, gamma, beta, 1e-5, 0.1, rm, rv, relu=True)
    res = {}
    for name, fn in (("fwd", lambda: ops.bn_relu_forward(t, gamma, beta, 1e-5, 0.1, rm, rv, relu=True)),
                     ("bwd", lambda: ops.bn_relu_backward(gy, t, y, gamma, mean, rstd, relu=True))):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(gr, stream=s):
                for _ in range(40):
                    fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / 400
    tot_f += res["fwd"]; tot_b += res["bwd"]
    print(f"n {a.n} F {F:5d}: forward {res['fwd']:6.2f} us  backward {res['bwd']:6.2f} us")
print(f"LCREC_BN_V4={os.environ.get('LCREC_BN_V4', '(default)')} n {a.n}: sum over the six widths forward {tot_f:.1f} us backward {tot_b:.1f} us")
