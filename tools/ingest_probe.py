#!/usr/bin/env python3
"""The step in front of the path: a `<DS>.emb-<plm>-td.npy` on the host -> fp32 rows resident in HBM
(EmbDataset.to_device: memory-mapped file, cast into two pinned staging buffers, chunked H2D copies on a side stream), and
what encode+assign makes per second when the items have to come over PCIe first.
    python tools/ingest_probe.py [--items 1000000] [--dim 768]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd.datasets import EmbDataset  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--workers", type=int, nargs="*", default=[1, 4, 8, 16])
a = ap.parse_args()
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    path = os.path.join(tmp, "Synth.emb-test-td.npy")
    rs = np.random.default_rng(0)
    x = rs.standard_normal((a.items, a.dim), dtype=np.float32)
    np.save(path, x)
    del x
    torch.zeros(1, device="cuda:0")
    t0 = time.perf_counter()
    pin = torch.empty((256 << 20) // 4, dtype=torch.float32, pin_memory=True)
    print(f"pinning 256 MB: {time.perf_counter() - t0:.3f} s")
    del pin
    for mmap in (False, True):
        for w in a.workers:
            t0 = time.perf_counter()
            ds = EmbDataset(path, mmap=mmap)
            t1 = time.perf_counter()
            dev = ds.to_device("cuda:0", workers=w)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            gb = dev.numel() * 4 / 1e9
            print(f"{a.items} x {a.dim} fp32 ({gb:.2f} GB), mmap={mmap}, {w} host threads: np.load {t1 - t0:.2f} s, to_device {t2 - t1:.3f} s = "
                  f"{gb / (t2 - t1):.1f} GB/s = {a.items / (t2 - t1) / 1e6:.2f} M items/s over the link (page cache warm: the file was just written)")
            assert torch.equal(dev[-3:].cpu(), torch.from_numpy(np.ascontiguousarray(ds.embeddings[-3:])))
            del ds, dev
            torch.cuda.empty_cache()
