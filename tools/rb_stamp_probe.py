#!/usr/bin/env python3
"""Where a K-tile of the 64 x 64 GEMM kernel (register-buffered form) spends its cycles.  Diagnostic library needed:

    tools/build_variant.sh stamp -DLCREC_GEMM_STAMP ; LCREC_LIB_PATH=tools/diag/liblcrec_hip_stamp.so python tools/rb_stamp_probe.py M K N   (K <= 1920)

Stamps (s_memtime) of the workgroup with tile number 9, lane 0 of each of its 4 waves, per K-tile:
  0 entry | 1 global loads issued | 2 eight MFMAs + the eight fragment reads issued | 3 all sixteen MFMAs and the four stores issued |
  4 lgkmcnt(0) | 5 after the barrier;  and the whole-kernel marks: entry, K loop start, K loop end, epilogue end."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import _lib, ops  # noqa: E402

m, k, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1024, 1024, 1024)
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
nk = min(60, k // 32)
names = ["issue loads", "8 MFMA + 8 reads", "8 MFMA + 4 stores", "lgkmcnt(0)", "barrier"]
print(f"{m} x {k} -> {n}: per K-tile cycles (median over K-tiles 4..{nk - 3}); K-tile period {int(np.median(np.diff(s[0, 4:nk - 2, 0])))} cycles")
for w_ in range(4):
    d = np.diff(s[w_, 4:nk - 2, :], axis=1)
    print(f"  wave {w_}: " + "  ".join(f"{nm} {int(np.median(d[:, i])):5d}" for i, nm in enumerate(names)) +
          f"   | loop-back {int(np.median(s[w_, 5:nk - 2, 0] - s[w_, 4:nk - 3, 5])):4d}")
mk = s[:, 60, :4]
for w_ in range(4):
    print(f"  wave {w_}: prologue {mk[w_, 1] - mk[w_, 0]:6d}  K loop {mk[w_, 2] - mk[w_, 1]:7d} ({(mk[w_, 2] - mk[w_, 1]) / max(1, k // 32):.0f} per K-tile)  "
          f"epilogue {mk[w_, 3] - mk[w_, 2]:6d}  total {mk[w_, 3] - mk[w_, 0]:7d}")
# per-K-tile trace of wave 0 (first 12 K-tiles): where the prologue's latency is paid
print("  wave 0, K-tiles 0..11, cycles per phase:")
for t in range(min(12, nk)):
    print("    kt %2d: " % t + " ".join(f"{int(v):5d}" for v in np.diff(s[0, t, :])))
