#!/usr/bin/env python3
"""Where an iteration of the one-launch Sinkhorn solver spends its cycles (diagnostic build only).

    make -C lc-rec_amd/csrc STAMP=1 ...  ->  tools/diag/liblcrec_hip_stamp.so
    LCREC_LIB_PATH=$PWD/tools/diag/liblcrec_hip_stamp.so python tools/sk_stamp_probe.py [--rows 2048] [--codes 256]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lcrec_amd import _lib, ops  # noqa: E402

SCALING = ["row sums E b, reduced together", "row scales (1 division), column accumulation, LDS", "8-wave sums -> global (publish)",
           "gather: poll + add all workgroups' partials (thread 0's share)", "barrier after the gather (slowest thread)",
           "column scales (1 division) + barrier", "(unused)", "(loop back, re-arm)"]
PHASES = ["row/col normalise + LDS partials", "partials -> global (A)", "owner gathers its columns (B: poll + load)", "owner adds + stores sums",
          "(unused)", "(unused)", "all sums (C: poll + load)", "(loop back)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    fn = lib.lcrec_debug_sk_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    g = torch.Generator(device="cuda:0").manual_seed(3)
    r = torch.randn((a.rows, 32), generator=g, device="cuda:0")
    cb = torch.randn((a.codes, 32), generator=g, device="cuda:0")
    buf = (ctypes.c_ulonglong * 8)()
    ops.sinkhorn_assign(r, cb, 0.003, a.iters)
    torch.cuda.synchronize()
    fn(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ops.sinkhorn_assign(r, cb, 0.003, a.iters)
    e1.record()
    torch.cuda.synchronize()
    fn(buf, 0)
    n = a.reps * a.iters
    print(f"rows {a.rows} codes {a.codes}: {e0.elapsed_time(e1) / a.reps * 1e3:.1f} us per solve (distances + solve, stamped build)")
    tot = sum(buf)
    scaling = os.environ.get("LCREC_SINKHORN_SCALING", "1") != "0" and a.codes % 64 == 0
    for name, c in zip(SCALING if scaling else PHASES, buf):
        print(f"  {name:45s} {c / n:9.0f} cycles/iteration  {100.0 * c / tot:5.1f} %")
    print(f"  {'sum':45s} {tot / n:9.0f} cycles/iteration (s_memtime ticks)")


if __name__ == "__main__":
    main()
