#!/usr/bin/env python3
"""The three MFMA kernels of the path at fixed shapes -- the program rocprofv3 --pmc is pointed at (tools/pmc_mfma.sh).

  linear_fwd_pp3   the dominant encoder GEMM: 131 072 rows (one chunk of lcrec_encode_assign) x 768 -> 2048 and x 2048 -> 1024
  rq_assign        1 000 000 latents x 32 through 4 x 256 codes (C3's quantiser pass)
  linear_fwd 64x64 the batch-sized kernel of a training step: 1024 x 2048 -> 1024
  linear_s16       the 32 x 64-tile kernel (v_mfma_f32_16x16x4_f32) on a launch it is dispatched for: 1024 x 1024 -> 512

    python tools/pmc_target.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(2024)
rnd = lambda *s: torch.randn(s, device=dev, generator=g)

x0 = rnd(131072, 768)
w0, b0 = rnd(2048, 768) * 0.03, torch.zeros(2048, device=dev)
w1, b1 = rnd(1024, 2048) * 0.03, torch.zeros(1024, device=dev)
for _ in range(reps):
    h = ops.linear_forward(x0, w0, b0, relu=True)
    ops.linear_forward(h, w1, b1, relu=True)

z = rnd(1000000, 32)
cbs = [rnd(256, 32) * 0.5 ** l for l in range(4)]
flat, ks = ops.flatten_codebooks(cbs)
for _ in range(reps):
    ops.rq_assign(z, flat, ks)

xb = rnd(1024, 2048)
for _ in range(4 * reps):
    ops.linear_forward(xb, w1, b1, relu=True)
w2, b2 = rnd(512, 1024) * 0.03, torch.zeros(512, device=dev)
xc = rnd(1024, 1024)
for _ in range(4 * reps):
    ops.linear_forward(xc, w2, b2, relu=True)
torch.cuda.synchronize()
print("pmc_target done")
