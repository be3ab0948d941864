#!/usr/bin/env python3
"""One or two chunk pipelines in lcrec_encode_assign (lcrec_context_set_pipelines) at C3's size: wall time of every pass over
many passes -- the figure of interest is the WORST pass as much as the median (round 2 saw two full-sized persistent launches
starve one another in a few percent of passes).   python tools/pipeline_probe.py [--passes 150]"""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd  # noqa: E402
from lcrec_amd import generate_indices as gen, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=1_000_000)
ap.add_argument("--in_dim", type=int, default=768)
ap.add_argument("--passes", type=int, default=150)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(2024)
model = lcrec_amd.RQVAE(in_dim=a.in_dim, num_emb_list=[256] * 4, e_dim=32, layers=[2048, 1024, 512, 256, 128, 64], kmeans_init=False,
                        sk_epsilons=[0.0] * 4, sk_iters=50).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(2024)
x = torch.randn((a.items, a.in_dim), generator=g, device=dev)
ref = None
for rnd in range(2):
    for P in (1, 2):
        ops.set_pipelines(P)
        idx = gen.assign_all(model, x)[0]
        if ref is None:
            ref = idx.clone()
        assert torch.equal(idx, ref)
        ts = []
        for _ in range(a.passes):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gen.assign_all(model, x)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        print(f"{P} pipeline(s), {a.passes} passes of {a.items} x {a.in_dim}: median {statistics.median(ts):.2f} ms, min {ts[0]:.2f}, "
              f"p95 {ts[int(0.95 * len(ts))]:.2f}, max {ts[-1]:.2f}; passes over 1.2 x median: {sum(t > 1.2 * statistics.median(ts) for t in ts)}",
              flush=True)
ops.set_pipelines(1)
