#!/usr/bin/env python3
"""Wall-clock of the index-emission flow (generate_indices.py:51-145) at BASELINE sizes on one GPU:
pass 1 (encode + assign), the conflict rounds, the .index.json text.

    python tools/generate_probe.py [--items 1000000] [--in_dim 768] [--levels 4] [--codes 256]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd  # noqa: E402
from lcrec_amd import generate_indices as gen, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--in_dim", type=int, default=768)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--out", type=str, default="/tmp/probe.index.json")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(2024)
    model = lcrec_amd.RQVAE(in_dim=a.in_dim, num_emb_list=[a.codes] * a.levels, e_dim=32,
                            layers=[2048, 1024, 512, 256, 128, 64], kmeans_init=False,
                            sk_epsilons=[0.0] * a.levels, sk_iters=50).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(2024)
    x = torch.randn((a.items, a.in_dim), generator=g, device=dev)
    with torch.no_grad():
        z = model.encoder(x[:65536])
        resid = z
        unused = torch.ones(resid.shape[0], dtype=torch.bool, device=dev)
        for l in range(a.levels):                          # data-scale codebooks: rows of the level's residuals
            perm = torch.randperm(resid.shape[0], generator=g, device=dev)
            pick = perm[unused[perm]][:a.codes]            # never a row that was a code before (its residual is 0)
            unused[pick] = False
            cb = resid[pick].clone()
            model.rq.vq_layers[l].embedding.weight.data.copy_(cb)
            resid = resid - cb[ops.rq_assign(resid.contiguous(), cb.reshape(-1), [a.codes])[0][:, 0]]

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    def alloc_stats():
        st = torch.cuda.memory_stats()
        return {k: st.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries",
                                          "reserved_bytes.all.current")}

    def clocks():
        """current sclk / mclk of card 0 as the driver reports them (a pass that runs 3x slow at a third of the clock
        is a power-state matter, not a scheduling one)"""
        out = []
        for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
            for card in ("card0", "card1"):
                path = f"/sys/class/drm/{card}/device/{name}"
                try:
                    with open(path) as fh:
                        cur = [ln.strip() for ln in fh if "*" in ln]
                    out.append(f"{name}={cur[0] if cur else '?'}")
                    break
                except OSError:
                    continue
        return " ".join(out) or "clocks unreadable"

    w_idx, w_resid, w_ks = gen.assign_all(model, x[:300_000])   # warm-up: every kernel form and both helper streams used once
    gen.resolve_collisions(model, w_idx, w_resid, w_ks)
    # Pass 1, first time at full size.  What the round-1 version of this probe timed here -- and sometimes saw take 185 ms
    # instead of 62 -- includes every first-time allocation at the 1 M-item sizes (idx 32 MB, latents 128 MB, the
    # [L][n][e] residual stack 512 MB, its clone): the allocator statistics around the call say whether it went to the device
    # for memory, and the traced repeats below give kernel time against wall time once the sizes are cached.
    a0 = alloc_stats()
    (idx, resid_last, ks), t_pass1 = timed(lambda: gen.assign_all(model, x))
    a1 = alloc_stats()
    print(f"pass 1, first call at full size: {t_pass1 * 1e3:.1f} ms; allocator: "
          + ", ".join(f"{k} +{a1[k] - a0[k]}" for k in a0) + f"; {clocks()}")
    reps = []
    for streams in (2, 1, 2):
        ops.set_pipelines(streams)
        b0 = alloc_stats()
        ops.trace_enable(True)
        _, t_rep = timed(lambda: gen.assign_all(model, x))
        tr = ops.trace_collect()
        ops.trace_enable(False)
        b1 = alloc_stats()
        reps.append(t_rep)
        print(f"pass 1 repeated, {streams} pipeline(s): wall {t_rep * 1e3:.1f} ms, kernel brackets {sum(v[1] for v in tr.values()):.1f} ms, "
              f"device allocs +{b1['num_device_alloc'] - b0['num_device_alloc']}; "
              + ", ".join(f"{k} {v[1]:.1f}/{v[0]}" for k, v in sorted(tr.items(), key=lambda kv: -kv[1][1])[:4]))
    ops.set_pipelines(2)
    if t_pass1 > 2.0 * min(reps):
        print(f"SLOW FIRST CALL: {t_pass1 * 1e3:.1f} ms against {min(reps) * 1e3:.1f} ms repeated -- see the allocator line above")
    t_pass1 = min(t_pass1, *reps)
    first = ops.collision_groups(idx, ks, want_groups=False)
    ops.trace_enable(True)
    (idx, history), t_rounds = timed(lambda: gen.resolve_collisions(model, idx, resid_last, ks))
    trace = ops.trace_collect()
    ops.trace_enable(False)
    final = ops.collision_groups(idx, ks, want_groups=False)
    _, t_json = timed(lambda: gen.dump_index_json(idx, a.out))
    size = os.path.getsize(a.out)
    os.remove(a.out)
    print(f"items {a.items}  in_dim {a.in_dim}  {a.levels} x {a.codes} codes")
    print(f"pass 1 (encode+assign)   {t_pass1 * 1e3:9.1f} ms   {a.items / t_pass1 / 1e6:7.2f} M items/s")
    print(f"conflict rounds ({len(history):2d})     {t_rounds * 1e3:9.1f} ms   groups/round {history[:6]}{' ...' if len(history) > 6 else ''}")
    print("  kernels: " + ", ".join(f"{k} {v[1]:.1f} ms/{v[0]}" for k, v in sorted(trace.items(), key=lambda kv: -kv[1][1])))
    print(f"  collision rate {first['collision_rate']:.6f} -> {final['collision_rate']:.6f}")
    print(f".index.json ({size / 1e6:.0f} MB)     {t_json * 1e3:9.1f} ms   {a.items / t_json / 1e6:7.2f} M items/s (D2H + text + write)")


if __name__ == "__main__":
    main()
