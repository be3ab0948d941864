#!/usr/bin/env python3
"""Can a hipGraph (torch.cuda.CUDAGraph) capture launches made through the C-ABI from ctypes?  Captures a forward GEMM,
a backward pair, a Sinkhorn solve and a fused torch optimizer step, replays them and compares with the eager results."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd  # noqa: E402
from lcrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(1024, 768, device=dev)
W = torch.randn(2048, 768, device=dev) * 0.03
b = torch.zeros(2048, device=dev)
cb = torch.randn(256, 32, device=dev)
r = torch.randn(1024, 32, device=dev)


pieces = {
    "linear_forward": lambda: ops.linear_forward(x, W, b, relu=True),
    "linear_backward": lambda: ops.linear_backward(yy, x, W)[1],
    "rq_assign": lambda: ops.rq_assign(r, cb.reshape(-1), [256], want_xq=True, want_sse=True, want_resid=True)[0],
    "code_stats": lambda: ops.code_stats(col, r, 256)[1],
    "sinkhorn(batch)": lambda: ops.sinkhorn_assign(r, cb, 0.003, 50, out=sk_out),
    "apply_level": lambda: ops.rq_apply_level(r, cb, col, want_sse=True)[1],
    "torch fused AdamW": lambda: (opt.step(), W_p)[1],
}
yy = ops.linear_forward(x, W, b, relu=True)
col = torch.randint(0, 256, (1024,), device=dev)
sk_out = torch.zeros(1024, dtype=torch.int64, device=dev)
W_p = torch.nn.Parameter(W.clone())
W_p.grad = torch.randn_like(W_p)
opt = torch.optim.AdamW([W_p], lr=1e-3, fused=True, capturable=True)
opt.step()


def try_capture(name, fn):
    with ops.deferred_checks() as chk:          # no host reads inside the capture
        for _ in range(2):
            want = fn()
        chk._keep = None
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g, stream=s):
                got = fn()
        except Exception as exc:   # noqa: BLE001
            print(f"{name:22s} CAPTURE FAILED: {str(exc).splitlines()[0]}")
            ops._deferred = []
            return None
        ops._deferred = []
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 100 * 1e6
    same = bool(torch.equal(got, want)) if name != "torch fused AdamW" else None
    print(f"{name:22s} captured; replay == eager: {same}; {tg:.1f} us per replay")
    return g


for name, fn in pieces.items():
    try:
        try_capture(name, fn)
    except Exception as exc:   # noqa: BLE001
        print(f"{name:22s} ERROR {exc!r}")
