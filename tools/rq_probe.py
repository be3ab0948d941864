import torch, sys
sys.path.insert(0, ".")
from lcrec_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
for L, K, n in ((4, 256, 1000000), (8, 1024, 1000000), (4, 256, 1024)):
    z = torch.randn((n, 32), generator=g, device=dev)
    cbs = [torch.randn((K, 32), generator=g, device=dev) * 0.5 ** l for l in range(L)]
    flat, ks = ops.flatten_codebooks(cbs)
    for _ in range(3): ops.rq_assign(z, flat, ks)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.rq_assign(z, flat, ks)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"L={L} K={K} n={n}: {ms:.3f} ms, {n / ms / 1e6:.3f} G items/s, {2 * 32 * K * L * n / ms / 1e9:.1f} TFLOP/s")
