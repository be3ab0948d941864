#!/usr/bin/env python3
"""Do two streams of persistent GEMM launches that share the CUs hurt each other?

Two lcrec_encode_assign calls (one chunk pipeline each, ops.set_pipelines(1)), 524 288 items each: first one after the
other on one stream, then at the same time on two streams.  The tiles of a persistent launch are dealt statically over
256 workgroups, so a launch that gets only part of the CUs runs in extra partial rounds.
"""
import os
import sys
import time

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd  # noqa: E402
from lcrec_amd import ops  # noqa: E402


def main():
    dev = "cuda:0"
    ops.set_pipelines(1)
    torch.manual_seed(1)
    model = lcrec_amd.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=[2048, 1024, 512, 256, 128, 64],
                            kmeans_init=False, sk_epsilons=[0.0] * 4, sk_iters=50).to(dev).eval()
    Ws, bs, scs, shs = model.encoder.folded()
    flat, ks = ops.flatten_codebooks([q.embedding.weight.detach() for q in model.rq.vq_layers])
    xs = [torch.randn((524288, 768), device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(two_streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for i, x in enumerate(xs):
            with torch.cuda.stream(streams[i if two_streams else 0]):
                outs.append(ops.encode_assign(x, Ws, bs, flat, ks, scs, shs)[0])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3, outs

    for two in (False, True):              # warm-up: workspaces of both streams
        run(two)
    ref = None
    for rep in range(3):
        for two in (False, True):
            ms, outs = run(two)
            if ref is None:
                ref = [o.clone() for o in outs]
            same = all(torch.equal(a, b) for a, b in zip(ref, outs))
            print(f"{'two streams at once' if two else 'one after the other':20s} {ms:8.1f} ms   indices identical: {same}", flush=True)


if __name__ == "__main__":
    main()
