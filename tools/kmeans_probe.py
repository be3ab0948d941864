"""Wall time of the device k-means initialiser (layers.kmeans_device) at the sizes the trainer calls it with."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lcrec_amd as hip
from lcrec_amd import layers

dev = "cuda:0"
for n, K, iters in ((1024, 256, 100), (2048, 256, 100), (8192, 1024, 100)):
    x = torch.randn(n, 32, device=dev)
    for rep in range(2):
        g = torch.Generator(device=dev).manual_seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = layers.kmeans_pp_seed(x, K, g)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        c = layers.kmeans_device(x, K, num_iters=iters, init=s)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n} K={K}: seed {1e3*(t1-t0):.1f} ms, lloyd({iters}) {1e3*(t2-t1):.1f} ms", flush=True)
