#!/usr/bin/env python3
"""Where does a K-tile of the generic (batch-sized) GEMM kernel spend its cycles?  Diagnostic library needed:

    make -C lc-rec_amd/csrc STAMP=1 ... (tools/build_diag.sh) ; LCREC_LIB_PATH=tools/diag/liblcrec_hip_stamp.so python tools/generic_stamp_probe.py M K N

Stamps (s_memtime) of the workgroup with tile number 9, lane 0 of each of its 4 waves, per K-tile:
  0 entry | 1 global loads of K-tile +2 issued | 2 ds_reads done and the 16 MFMAs issued | 3 after the barrier |
  4 K-tile +1 written to LDS (includes the wait for its global loads) | 5 after the second barrier"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import _lib, ops  # noqa: E402

m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 2048, 1024)
dev = torch.device("cuda:0")
x = torch.randn((m, k), device=dev)
w = torch.randn((n, k), device=dev) * 0.03
b = torch.zeros(n, device=dev)
for _ in range(5):
    ops.linear_forward(x, w, b, relu=True)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (4 * 64 * 6))()
lib.lcrec_debug_generic_stamps.argtypes = [ctypes.c_void_p]
assert lib.lcrec_debug_generic_stamps(ctypes.cast(buf, ctypes.c_void_p)) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(4, 64, 6).astype(np.int64)
nk = min(64, k // 32)
names = ["issue loads", "ds_read+MFMA issue", "barrier 1", "vmcnt wait + LDS write", "barrier 2"]
print(f"{m} x {k} -> {n}: per K-tile cycles, wave 0 (median over K-tiles 4..{nk - 3}); K-tile period "
      f"{int(np.median(np.diff(s[0, 4:nk - 2, 0])))} cycles")
for w_ in range(4):
    d = np.diff(s[w_, 4:nk - 2, :], axis=1)
    print(f"  wave {w_}: " + "  ".join(f"{nm} {int(np.median(d[:, i])):5d}" for i, nm in enumerate(names)) +
          f"   | loop-back {int(np.median(s[w_, 5:nk - 2, 0] - s[w_, 4:nk - 3, 5])):4d}")
