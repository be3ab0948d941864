#!/usr/bin/env python3
"""Where does a ping-pong GEMM phase spend its cycles?  Needs the diagnostic library
(make -C lc-rec_amd/csrc STAMP=1 into another directory) loaded through LCREC_LIB_PATH:

    LCREC_LIB_PATH=tools/diag/liblcrec_hip_stamp.so python tools/stamp_probe.py

Stamps (s_memtime, shader cycles) of workgroup 8, lane 0 of every wave, first 64 phases:
  compute role: 0 = phase entry, 1 = before the mid-phase barrier, 2 = after it, 3 = all 64 MFMAs issued
  staging role: 0 = phase entry, 1 = LDS writes done (before the barrier), 2 = after the barrier
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    rows, k, n = 131072, int(os.environ.get("PROBE_K", "2048")), int(os.environ.get("PROBE_N", "1024"))
    x = torch.randn((rows, k), generator=g, device=dev)
    w = torch.randn((n, k), generator=g, device=dev) * 0.02
    b = torch.zeros(n, device=dev)
    for _ in range(3):
        ops.linear_forward(x, w, b, relu=True)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * (8 * 64 * 4))()
    lib.lcrec_debug_gemm_stamps.argtypes = [ctypes.c_void_p]
    rc = lib.lcrec_debug_gemm_stamps(ctypes.cast(buf, ctypes.c_void_p))
    assert rc == 0, rc
    s = np.frombuffer(buf, dtype=np.uint64).reshape(8, 64, 4).astype(np.int64)
    t0 = s[:, 8, 0].min()
    print("phase | wave0 (group 0)                       | wave4 (group 1)        [cycles since phase 8 entry]")
    for p in range(int(os.environ.get("PROBE_P0", "8")), 24):
        def fmt(w):
            v = s[w, p] - t0
            role = "C" if (p % 2) == (w // 4) else "S"
            return role + " " + " ".join(f"{int(t):7d}" for t in v[: (4 if role == "C" else 3)])
        print(f"{p:5d} | {fmt(0):38s} | {fmt(4)}")
    # arrival of each computing wave at the mid-phase barrier, relative to the earliest of the four: who is late?
    print("phase | computing group | arrival at the barrier, cycles after the first of the four waves | barrier exit - last arrival")
    for p in range(int(os.environ.get("PROBE_P0", "8")), 24):
        g = p % 2
        arr = s[4 * g:4 * g + 4, p, 1]
        stg = s[4 * (1 - g):4 * (1 - g) + 4, p, 3]          # staging waves: just before their barrier (persistent form only)
        late = f" | staging waves arrive {' '.join(f'{int(a - arr.max()):6d}' for a in stg)} after the last computing wave" if stg.any() else ""
        print(f"{p:5d} | group {g}         | " + " ".join(f"{int(a - arr.min()):6d}" for a in arr) +
              f" | {int(s[4 * g, p, 2] - arr.max()):6d}" + late)
    # per-phase durations for a computing wave: entry -> barrier wait start -> barrier exit -> last MFMA issued
    comp = []
    for w in (0, 4):
        for p in range(8, 60):
            if (p % 2) == (w // 4):
                v = s[w, p]
                comp.append((v[1] - v[0], v[2] - v[1], v[3] - v[2]))
    comp = np.array(comp)
    print("compute phase, mean cycles: entry->before barrier %.0f | in barrier %.0f | barrier->64th MFMA issued %.0f"
          % tuple(comp.mean(0)))
    mbuf = (ctypes.c_ulonglong * (8 * 4))()
    if hasattr(lib, "lcrec_debug_gemm_marks"):
        lib.lcrec_debug_gemm_marks.argtypes = [ctypes.c_void_p]
        assert lib.lcrec_debug_gemm_marks(ctypes.cast(mbuf, ctypes.c_void_p)) == 0
        m = np.frombuffer(mbuf, dtype=np.uint64).reshape(8, 4).astype(np.int64)
        for w in (0, 4):
            d = np.diff(m[w])
            print("wave %d of workgroup 8: prologue %d | K loop %d (%d K-tiles) | epilogue issued %d cycles"
                  % (w, d[0], d[1], k // 32, d[2]))
    per = np.diff(s[0, 8:60:2, 0])
    print("period of two phases (wave 0 entry to entry): mean %.0f cycles (ideal: 8192 at K slice 32, 16384 at 64)" % per.mean())


if __name__ == "__main__":
    main()
