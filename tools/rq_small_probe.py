#!/usr/bin/env python3
"""lcrec_rq_assign on inputs between a training batch and a chunk (1 k .. 64 k items, 4 x 256 codes), replayed from a hipGraph:
where the split form (one 64-item tile per workgroup, code blocks dealt over its waves) stops paying.
    LCREC_RQ_SPLIT_TILES=<n> python tools/rq_small_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
cbs = [torch.randn((256, 32), generator=g, device=dev) * 0.5 ** l for l in range(4)]
flat, ks = ops.flatten_codebooks(cbs)
for n in (1024, 4096, 8192, 16859, 32768, 65536):
    z = torch.randn((n, 32), generator=g, device=dev)
    ops.rq_assign(z, flat, ks)
    torch.cuda.synchronize()
    gr, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.rq_assign(z, flat, ks)
        with torch.cuda.graph(gr, stream=s):
            for _ in range(20):
                ops.rq_assign(z, flat, ks)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"LCREC_RQ_SPLIT_TILES={os.environ.get('LCREC_RQ_SPLIT_TILES', '(default)')} n {n:6d}: {e0.elapsed_time(e1) * 1e3 / 100:7.1f} us per call")
