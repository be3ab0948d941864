#!/usr/bin/env python3
"""One forward GEMM shape, repeated -- a target for rocprofv3 --pmc.   python tools/one_gemm.py M K N [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

m, k, n = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
x = torch.randn((m, k), device=dev)
w = torch.randn((n, k), device=dev) * 0.03
b = torch.zeros(n, device=dev)
for _ in range(reps):
    ops.linear_forward(x, w, b, relu=True)
torch.cuda.synchronize()
