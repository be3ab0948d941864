#!/usr/bin/env python3
"""Headline benchmark: item-embeddings/sec through 4-level RQ encode+assign.

One "step" = one pass of RQVAE.get_indices(use_sk=False) semantics (reference
index/models/rqvae.py:68-72) over the rank's shard of synthetic item embeddings
already resident in HBM: fp32 [n, d_in] -> int64 [n, L].

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json configs):
  c3 (default)  1M items x 768-d per GPU, 4 levels x 256 codes, e_dim 32, MLP 2048-1024-512-256-128-64
  c4            1.25M items x 4096-d per GPU (10M over 8 GPUs), same model
  c2            16 859 items x 4096-d (Games-sized)
Items shard across ranks with no data-path collective (weak scaling: fixed items per GPU).

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP events on the launch
stream) and, at N=1, `cpu_baseline` (the torch-CPU restatement of the reference path timed on
this host's cores).  lcrec_encode_assign runs one chunk pipeline by default: the hipEvent brackets
of the timed region are then un-overlapped launches and `roofline` is measured in the timed region
itself.  With `--pipelines 2` (opt-in, see include/lcrec.h) launches of the two pipelines overlap;
the kernel's own rate is then taken in a short untimed pass with one pipeline and the in-region
figures are reported next to it (`roofline.in_region`).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c3": dict(items=1_000_000, in_dim=768, name="C3 synthetic 1M x 768-d, 4x256 codes"),
    "c4": dict(items=1_250_000, in_dim=4096, name="C4 synthetic 10M x 4096-d over 8 GPUs (1.25M per GPU), 4x256 codes"),
    "c2": dict(items=16_859, in_dim=4096, name="C2 Games-sized 16859 x 4096-d, 4x256 codes"),
    "c5": dict(items=1_250_000, in_dim=4096, codes=[1024] * 8,
               name="C5 synthetic 10M x 4096-d over 8 GPUs (1.25M per GPU), 8x1024 codes (encode+assign pass)"),
}
HIDDEN = [2048, 1024, 512, 256, 128, 64]   # index/run.sh:15
E_DIM = 32                                  # index/run.sh:10
CODES = [256, 256, 256, 256]                # index/run.sh:13
PEAK_F32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0


def synth_model(in_dim, device, seed=2024, codes=None):
    """Random-init encoder in the reference's init (xavier_normal_ weights, zero bias;
    layers.py:33-40) and data-scale codebooks (rows sampled from each level's residuals,
    the stand-in for k-means named in SURVEY.md section 8d)."""
    import lcrec_amd
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dims = [in_dim] + HIDDEN + [E_DIM]
    Ws, bs = [], []
    for l in range(len(dims) - 1):
        std = (2.0 / (dims[l] + dims[l + 1])) ** 0.5
        Ws.append(torch.randn((dims[l + 1], dims[l]), generator=g, device=device, dtype=torch.float32) * std)
        bs.append(torch.zeros(dims[l + 1], device=device, dtype=torch.float32))
    # small enough that the set-up below never launches the kernels the bench prices on workload-sized inputs (the PMC
    # summaries average per kernel name), large enough to hold every level's codes
    probe = torch.randn((max(4096, 2 * sum(codes or CODES)), in_dim), generator=g, device=device, dtype=torch.float32)
    z = probe
    for l in range(len(Ws)):
        z = lcrec_amd.ops.linear_forward(z, Ws[l], bs[l], relu=l != len(Ws) - 1)
    cbs, resid = [], z
    unused = torch.ones(resid.shape[0], dtype=torch.bool, device=device)
    for K in (codes or CODES):
        # a row that already served as a code has residual exactly 0: drawing it again would put duplicate all-zero
        # codes into the deeper codebooks (exact ties for every item near them), which no k-means init produces
        perm = torch.randperm(resid.shape[0], generator=g, device=device)
        pick = perm[unused[perm]][:K]
        unused[pick] = False
        cbs.append(resid[pick].clone())
        flat, ks = lcrec_amd.ops.flatten_codebooks(cbs)
        _, xq, _, _ = lcrec_amd.ops.rq_assign(z, flat, ks, want_xq=True)
        resid = z - xq
    return dims, Ws, bs, cbs


def host_cores():
    """CPU threads this process may actually use: affinity mask, capped by the cgroup CPU quota.
    A one-GPU box exposes all host CPUs but grants a share of them; with no readable quota on a
    large host we take 16, the share a one-GPU box gets."""
    env = os.environ.get("LCREC_CPU_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()
            if q != "max":
                quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                period = int(fh.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    elif n > 64:
        n = 16
    return n


def cpu_model():
    """Model string of the host CPU (the CPU baseline is quoted with core count AND model)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(x_cpu, dims, Ws, bs, cbs, budget_s=14.0):
    """The reference's CPU path (torch CPU ops of rqvae.py:68-72, restated in oracle/torch_ref.py) timed on this host's
    cores on a bounded sample, as BASELINE.md section 4 asks: batch 4096 AND batch 64 (the batch index/generate_indices.py:77-79
    uses), each the median of >= 3 passes.  `value` is the batch-4096 median (the reference's faster setting)."""
    from oracle import torch_ref
    codes = [int(c.shape[0]) for c in cbs]
    spec = torch_ref.Spec(dims[0], codes, dims[-1], dims[1:-1], sk_epsilons=[0.0] * len(codes))
    sd = {}
    for l, (W, b) in enumerate(zip(Ws, bs)):
        slot = torch_ref.linear_slot(l, False)
        sd[f"encoder.mlp_layers.{slot}.weight"] = W.cpu()
        sd[f"encoder.mlp_layers.{slot}.bias"] = b.cpu()
    for l, c in enumerate(cbs):
        sd[f"rq.vq_layers.{l}.embedding.weight"] = c.cpu()
    cores = host_cores()
    torch.set_num_threads(cores)

    def one_pass(batch, rows):
        t0 = time.perf_counter()
        for lo in range(0, rows, batch):
            torch_ref.get_indices(spec, sd, x_cpu[lo:min(rows, lo + batch)])
        return rows / (time.perf_counter() - t0)

    res = {}
    for batch, share in ((4096, 0.55), (64, 0.45)):
        torch_ref.get_indices(spec, sd, x_cpu[:batch])          # warm-up
        probe_rows = min(x_cpu.shape[0], 8 * batch if batch >= 4096 else 64 * batch)
        rate = one_pass(batch, probe_rows)
        # three passes inside this batch size's share of the budget, each over the same leading rows of the sample
        rows = int(max(batch, min(x_cpu.shape[0], rate * budget_s * share / 3.3)))
        rows -= rows % batch
        passes = sorted(one_pass(batch, rows) for _ in range(3))
        res[batch] = {"median": passes[1], "passes": [round(v, 1) for v in passes], "rows_per_pass": rows}
    return {"value": res[4096]["median"], "unit": "items/s", "cores": torch.get_num_threads(), "cpu": cpu_model(), "kind": "port",
            "batch_4096": res[4096], "batch_64": res[64],
            "sample": f"leading rows of the same synthetic tensor ({res[4096]['rows_per_pass']} per pass at batch 4096, "
                      f"{res[64]['rows_per_pass']} at batch 64 = generate_indices.py's), median of 3 passes each, fp32, "
                      f"oracle/torch_ref.get_indices (torch CPU ops of rqvae.py:68-72)"}


def secondary_metrics(dev, cbs, ks):
    """SURVEY.md section 8d's secondary figures, measured after the timed region on rank 0 at N=1 (never part of `value`):
    quantizer-only throughput ([n, 32] latents resident in HBM -> indices, lcrec_rq_assign alone) and the training step of
    the shipped recipe (index/run.sh: batch 1024, 4 x 256 codes, Sinkhorn on the last level, BatchNorm on) as
    lcrec_amd.engine runs it -- one captured hipGraph per step, synthetic 768-d batch, random-init weights."""
    import lcrec_amd
    from lcrec_amd import ops
    from lcrec_amd.engine import TrainEngine
    out = {}
    n = 1_000_000
    z = torch.randn((n, cbs[0].shape[1]), device=dev)
    flat, kk = ops.flatten_codebooks(cbs)
    for _ in range(2):
        ops.rq_assign(z, flat, kk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.rq_assign(z, flat, kk)
    e1.record()
    torch.cuda.synchronize()
    out["quantizer_only_items_per_s"] = n * 10 / (e0.elapsed_time(e1) * 1e-3)
    del z
    torch.manual_seed(2024)
    model = lcrec_amd.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=HIDDEN, bn=True, kmeans_init=False,
                            sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    eng = TrainEngine(model, opt, "linear", 10, 10_000)
    batch = torch.randn((1024, 768), device=dev)
    for _ in range(5):
        eng.step(batch)
    torch.cuda.synchronize()
    steps = 50
    e0.record()
    for _ in range(steps):
        eng.step(batch)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    eng.end_epoch(None)                                        # raises if a step's loss was NaN or a solver gave up
    out["train_step_ms"] = ms
    out["train_items_per_s"] = 1024 / (ms * 1e-3)
    out["train_config"] = "batch 1024 x 768-d, 4 x 256 codes, Sinkhorn on the last level, bn=True, AdamW + clip + linear warm-up; one hipGraph per step"
    return out


def dp_training_secondary(device, world, rank, dp_graph):
    """N > 1 (or a one-rank group): the shipped recipe's training step as item-sharded data parallel -- every rank 1024 rows of
    a global batch of 1024 x N (weak), the exchanges of DESIGN.md section 6 in the step (BatchNorm statistics, the Sinkhorn
    level's rows, losses, the flat gradient buffer in two asynchronous spans) -- launched eagerly, and (--dp-graph) captured
    in the step's hipGraph.  After K steps every rank's parameter checksum is all-gathered: `train_ranks_seen` ranks reported,
    and they must agree (same initial weights, same global gradients)."""
    import torch.distributed as dist
    import lcrec_amd
    from lcrec_amd import dist as ldist
    from lcrec_amd.engine import TrainEngine
    out = {}
    ctx = ldist.adopt(device)
    try:
        for mode in (["off", "on"] if dp_graph else ["off"]):
            torch.manual_seed(2024)                                   # the same initial weights on every rank
            model = lcrec_amd.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=HIDDEN, bn=True, kmeans_init=False,
                                    sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50).to(device)
            model.train()
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
            eng = TrainEngine(model, opt, "linear", 10, 10_000, dist=ctx, dp_graph=mode)
            g = torch.Generator(device=device).manual_seed(77 + rank)
            batch = torch.randn((1024, 768), generator=g, device=device)
            ctx.set_batch(1024, 1024 * world)
            for _ in range(5):
                eng.step(batch)
            dist.barrier()
            torch.cuda.synchronize()
            steps = 30
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.step(batch)
            dist.barrier()
            torch.cuda.synchronize()
            ms = torch.tensor([(time.perf_counter() - t0) / steps * 1e3], dtype=torch.float64, device=device)
            dist.all_reduce(ms, op=dist.ReduceOp.MAX)
            eng.end_epoch(None)
            mine = torch.stack([torch.tensor(float(rank), dtype=torch.float64, device=device), eng.flat_p.double().sum()])
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            got = torch.stack(got).cpu()
            key = "graph" if mode == "on" else "eager"
            out[f"dp_train_step_ms_{key}"] = float(ms.item())
            out[f"dp_train_items_per_s_{key}"] = 1024 * world / (float(ms.item()) * 1e-3)
            out[f"dp_train_graph_replays_{key}"] = eng.graph_replays
            if mode == "off":                                         # (a captured step counts its collectives once, at capture)
                out["dp_collectives_per_step"] = eng.collectives // max(1, eng.host_steps)
            out["train_ranks_seen"] = int(got[:, 0].unique().numel())
            out[f"train_param_checksums_agree_{key}"] = bool((got[:, 1] == got[0, 1]).all())
            del eng, model, opt
    finally:
        ldist.release()
    out["dp_train_config"] = ("batch 1024 per rank (global 1024 x N) x 768-d, 4 x 256 codes, Sinkhorn on the last level over the gathered "
                              "global batch, bn=True with global-batch statistics, AdamW + clip on the all-reduced gradient")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--items", type=int, default=0, help="override items per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the quantizer-only and training-step figures (N=1 only)")
    ap.add_argument("--trace-pass", choices=["auto", "timed", "separate"], default="auto",
                    help="where the per-launch hipEvent brackets behind `roofline` are taken: inside the timed region (default), or "
                         "in an untimed pass of the same launches right after it (default for --workload c2, whose launches are short)")
    ap.add_argument("--pipelines", type=int, default=0,
                    help="chunk pipelines of lcrec_encode_assign (lcrec_context_set_pipelines); 0 = the library's default (1)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="with --gpus 1: build a ONE-rank nccl (= RCCL) process group and run the N>1 code path through it "
                         "(barriers, the MAX all-reduce of the time, the all-gather behind ranks_seen)")
    ap.add_argument("--dp-graph", action="store_true",
                    help="N > 1: also time the data-parallel training step with its collectives CAPTURED in the hipGraph "
                         "(unverified on more than one GPU; default: eager collectives only)")
    ap.add_argument("--no-dp-train", action="store_true", help="N > 1: skip the data-parallel training-step figure")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real multi-GPU run); gloo = rehearsal of the N>1 code path, "
                         "ranks may then share one GPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torch.distributed.run --nproc-per-node {args.gpus}")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a MI355X: no HIP device visible (lcrec_amd has no CPU path)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        sys.exit(f"LOCAL_RANK {local_rank} but only {ndev} HIP device(s) visible")
    device = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.rehearse_rccl
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    import lcrec_amd
    from lcrec_amd import ops
    lcrec_amd._lib.load()
    pipelines = max(1, min(2, args.pipelines or 1))
    ops.set_pipelines(pipelines)

    wl = WORKLOADS[args.workload]
    n = args.items or wl["items"]
    dims, Ws, bs, cbs = synth_model(wl["in_dim"], device, codes=wl.get("codes"))
    flat, ks = ops.flatten_codebooks(cbs)
    g = torch.Generator(device=device)
    g.manual_seed(2024 + rank)
    x = torch.randn((n, wl["in_dim"]), generator=g, device=device, dtype=torch.float32)

    def step():
        return ops.encode_assign(x, Ws, bs, flat, ks)[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        idx = step()
    barrier()
    # The hipEvent pair around every launch is itself stream work (two marker packets per launch).  At C3 -- 15 launches of
    # 0.1-20 ms per pass -- that is nothing; at C2 a pass is 11 launches in 2.7 ms and the markers cost it ~4 % (measured:
    # --trace-pass timed vs separate).  "separate" keeps the timed region free of them and brackets the same launches in an
    # untimed pass right after it (the default for c2 only; the headline workloads are bracketed inside the timed region).
    trace_pass = args.trace_pass if args.trace_pass != "auto" else ("separate" if args.workload == "c2" else "timed")
    ops.trace_enable(trace_pass == "timed")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx = step()
    barrier()
    elapsed = time.perf_counter() - t0
    trace = ops.trace_collect()
    ops.trace_enable(False)
    # With more than one chunk pipeline two layers' kernels share the chip, so the hipEvent brackets of the timed region
    # measure launches that overlap each other.  The roofline of the kernel itself is therefore taken in a second,
    # untimed pass with ONE pipeline (same inputs, same launches, nothing overlapped); the in-region figures are kept
    # next to it.  `python bench.py --pipelines 1` makes the two coincide (that is the command profiled for profiles/).
    solo_trace, solo_steps = None, 0
    if pipelines > 1 or trace_pass == "separate":
        solo_steps = min(args.steps, 3) if pipelines > 1 else args.steps
        ops.set_pipelines(1)
        step()
        barrier()
        ops.trace_enable(True)
        for _ in range(solo_steps):
            idx_solo = step()
        barrier()
        solo_trace = ops.trace_collect()
        ops.trace_enable(False)
        ops.set_pipelines(pipelines)
        assert torch.equal(idx_solo, idx)                      # pipelines change scheduling, never results
    # near-tie audit (untimed): rows whose top-2 code gap at some level is within ops.NEARTIE_TAU of the rounding
    # magnitude -- the only rows a CPU run of the reference could index differently (tests/golden/f9_neartie_*.npz)
    audit = {}
    idx_audit = ops.encode_assign(x, Ws, bs, flat, ks, audit=audit, tie_tau=ops.NEARTIE_TAU)[0]
    assert torch.equal(idx_audit, idx)
    neartie_rows = int((audit["neartie"] != 0).sum())
    del audit, idx_audit
    checksum = int(idx.sum())                        # per-rank: proves every rank computed, and what
    ranks_seen, checksums = 1, [checksum]
    if use_dist:
        cdev = device if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # one all-gather of (rank, index checksum, near-tie rows) through the collective backend: `ranks_seen` is the
        # number of distinct ranks RCCL actually delivered data from
        mine = torch.tensor([rank, checksum, neartie_rows], dtype=torch.int64, device=cdev)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        got = torch.stack(got).cpu()
        ranks_seen = int(got[:, 0].unique().numel())
        checksums = [int(v) for v in got[:, 1]]
        neartie_rows = int(got[:, 2].sum())

    # Algorithmic flops per GEMM kernel: replay the library's dispatch rule (gemm_f32.hip, linear_forward)
    # over the chunks lcrec_encode_assign walks (lcrec_encode_assign_chunk_rows() each).
    def gemm_pieces(rows, out, k):
        """[(kernel name, rows)] of one Linear launch: the library's dispatch rule (gemm_f32.hip, linear_forward)."""
        if out <= 32:
            return [("linear_fwd_128x32", rows)]
        if out <= 64:
            return [("linear_fwd_64x64" if -(-rows // 128) < 256 else "linear_fwd_128x64", rows)]
        ntile = -(-out // 128)
        pp_tiles = -(-rows // 256) * ntile
        rounds = -(-pp_tiles // 256)
        fits = pp_tiles >= 256 and (rounds >= 8 or pp_tiles * 5 >= rounds * 256 * 4)
        forced = os.environ.get("LCREC_GEMM_PP", "-1")
        last0 = pp_tiles % 256
        if k % 32 == 0 and forced == "-1" and rounds < 8 and last0 != 0 and last0 * 10 < 256 * 9:
            # mid-sized launch with a mostly empty last round: whole rounds to the ping-pong kernel, the tail apart
            p = rows // 256
            while p >= 1 and p * ntile >= 256 and p >= rows // 256 - 256 // ntile:
                last = (p * ntile) % 256
                if last == 0 or last * 10 >= 256 * 9:
                    if p * 256 >= rows:
                        break
                    return [("linear_fwd_pp_256x128", p * 256)] + gemm_pieces(rows - p * 256, out, k)
                p -= 1
        if k % 32 == 0 and (forced == "1" or (forced != "0" and fits)):
            return [("linear_fwd_pp_256x128", rows)]
        return [("linear_fwd_64x64" if -(-rows // 128) * ntile < 512 else "linear_fwd_128x128", rows)]

    macs_all = sum(dims[l] * dims[l + 1] for l in range(len(dims) - 1)) + sum(E_DIM * k for k in ks)
    flops, alg_bytes = {}, {}
    chunk = int(lcrec_amd._lib.load().lcrec_encode_assign_chunk_rows())
    for lo in range(0, n, chunk):
        rows = min(chunk, n - lo)
        for l in range(len(dims) - 1):
            out = dims[l + 1]
            for kname, part in gemm_pieces(rows, out, dims[l]):
                flops[kname] = flops.get(kname, 0.0) + 2.0 * part * dims[l] * out * args.steps
                # read the activations and the weights once, write the outputs once
                alg_bytes[kname] = alg_bytes.get(kname, 0.0) + 4.0 * (part * dims[l] + out * dims[l] + part * out) * args.steps
    dom = max(flops, key=lambda k: flops[k])

    def kernel_rate(tr, steps):
        launches, total_ms = tr.get(dom, (0, 0.0))
        fl = flops[dom] * steps / args.steps
        ach = fl / (total_ms * 1e-3) / 1e12 if total_ms > 0 else None
        return launches, total_ms, fl, ach

    separate = trace_pass == "separate" and pipelines == 1
    if separate:
        trace = solo_trace                      # the timed region carried no brackets: every per-kernel figure is the untimed pass's
    in_launches, in_ms, in_flops, in_ach = kernel_rate(trace, solo_steps if separate else args.steps)
    launches, total_ms, flops_dom_total, achieved = kernel_rate(solo_trace, solo_steps) if solo_trace else \
        (in_launches, in_ms, in_flops, in_ach)
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_dominant_kernel.json")
    if os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            traffic = json.load(fh).get(args.workload, {}).get(dom, {}).get("hbm_bytes_per_launch")
    roofline = {
        "kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS,
        "unit": "TFLOP/s", "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None, "traffic": traffic,
        "traffic_source": "static: profiles/pmc_dominant_kernel.json (separate rocprofv3 --pmc passes, FETCH_SIZE/WRITE_SIZE "
                          "corrected per the guide), not measured in this run",
        "launches": launches, "avg_launch_ms": (total_ms / launches) if launches else None,
        "flops_per_launch": (flops_dom_total / launches) if launches else None,
        "algorithmic_bytes_per_launch": (alg_bytes[dom] / in_launches) if in_launches else None,
        "measured": ("hipEvent pairs around every launch of the kernel, in the timed region" if not solo_trace else
                     f"hipEvent pairs around every launch of the kernel in an untimed pass of {solo_steps} steps right after the timed "
                     f"region (same inputs, same launches); the timed region carries no brackets (--trace-pass separate: at this "
                     f"workload the marker packets cost the pass ~4 %)" if separate else
                     f"hipEvent pairs around every launch of the kernel in an untimed pass of {solo_steps} steps with one chunk "
                     f"pipeline; the timed region runs {pipelines} pipelines whose launches overlap (see in_region)"),
        "in_region": {"pipelines": pipelines, "launches": in_launches,
                      "avg_launch_ms": (in_ms / in_launches) if in_launches else None, "achieved_per_launch": in_ach},
        "kernel_ms": {k: round(v[1], 3) for k, v in trace.items()},
    }

    # the quantiser kernel beside it (BASELINE north_star: "HBM GB/s on the argmin sweep and MFMA utilisation on the distance GEMM"):
    # one launch per pass over all n items; it never materialises the [n, K] distance matrix the reference sweeps (vq.py:71-75)
    rq_launches, rq_ms = trace.get("rq_assign", (0, 0.0))
    rq_flops = 2.0 * E_DIM * sum(ks) * n                       # per launch (all levels that fit LDS together run in one launch)
    rq_bytes = (4.0 * E_DIM + 8.0 * len(ks)) * n
    rq_launches_per_pass = max(1, rq_launches // max(1, args.steps))
    rq_avg_ms = rq_ms / rq_launches if rq_launches else None
    rq_ach = (rq_flops / rq_launches_per_pass) / (rq_avg_ms * 1e-3) / 1e12 if rq_avg_ms else None
    rq_traffic = None
    if os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            rq_traffic = json.load(fh).get(args.workload, {}).get("rq_assign", {}).get("hbm_bytes_per_launch")
    roofline_rq = {
        "kernel": "rq_assign", "bound": "mfma", "achieved": rq_ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": (rq_ach / PEAK_F32_MFMA_TFLOPS) if rq_ach else None, "launches": rq_launches, "avg_launch_ms": rq_avg_ms,
        "flops_per_launch": rq_flops / rq_launches_per_pass, "algorithmic_bytes_per_launch": rq_bytes / rq_launches_per_pass,
        "hbm_gbps_algorithmic": (rq_bytes / rq_launches_per_pass) / (rq_avg_ms * 1e-3) / 1e9 if rq_avg_ms else None,
        "hbm_peak_gbps": PEAK_HBM_GBPS, "traffic": rq_traffic,
        "mfma_counters": "profiles/r03_pmc_mfma.txt (SQ_VALU_MFMA_BUSY_CYCLES per launch: 0.62 of the pipe cycles at 1 M items)",
        "note": "the reference materialises the [n, K] fp32 distance matrix per level (8 KB/item at 4 x 256) and sweeps it for the "
                "argmin; this kernel keeps it in MFMA accumulators, so its HBM traffic is the 160 B/item of latents in and indices out",
    }

    out = None
    if rank == 0:
        total_items = n * world * args.steps
        value = total_items / elapsed
        out = {
            "metric": f"item-embeddings/sec through {len(ks)}-level RQ encode+assign", "value": value, "unit": "items/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "items_per_gpu": n, "in_dim": wl["in_dim"], "mlp": HIDDEN,
                       "e_dim": E_DIM, "levels": len(ks), "codes_per_level": ks[0], "sharding": f"items/{world}",
                       "chunk_pipelines": pipelines,
                       "flop_per_item": 2 * macs_all, "bytes_per_item": 4 * wl["in_dim"] + 8 * len(ks)},
            "roofline": roofline, "roofline_rq_assign": roofline_rq,
            "e2e_mfma_frac": value / world * 2 * macs_all / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "ranks_seen": ranks_seen, "idx_checksums": checksums,
            "neartie_rows": neartie_rows, "neartie_tau": ops.NEARTIE_TAU,
            "neartie_note": "rows (all ranks) whose top-2 code gap at some level is <= tau x distance magnitude: the only rows "
                            "on which the reference's CPU arithmetic can differ from the canonical order this library and "
                            "its oracle implement (measured: 33 of 1 M at C3, 0 of 16 859 at C2; DESIGN.md section 2)",
            "io_gbps": value / world * (4 * wl["in_dim"] + 8 * len(ks)) / 1e9,
        }
        if world == 1:
            # parity spot check (not timed): 1024 rows against the bit-exact C oracle
            from oracle import cpu_oracle
            m = min(1024, n)
            want = cpu_oracle.encode_assign(x[:m].cpu().numpy(), [w.cpu().numpy() for w in Ws],
                                            [b.cpu().numpy() for b in bs], [c.cpu().numpy() for c in cbs],
                                            threads=host_cores())["idx"]
            out["parity_rows_checked"] = m
            out["parity_mismatch_rows"] = int((idx[:m].cpu().numpy() != want).any(axis=1).sum())
            out["parity_checker"] = "oracle/lcrec_oracle.c (canonical order); reference-differing rows: see neartie_rows"
            if not args.no_cpu_baseline:
                sample = x[: min(n, 200_000)].cpu()
                out["cpu_baseline"] = cpu_baseline(sample, dims, Ws, bs, cbs)
            if not args.no_secondary:
                try:
                    out["secondary"] = secondary_metrics(device, cbs, ks)
                except Exception as err:                       # never at the expense of the line itself
                    out["secondary"] = {"error": f"{type(err).__name__}: {err}"}
    # N > 1 (every rank takes part): the data-parallel training step as a secondary figure, under a watchdog -- a hang in it
    # (this path has never run on more than one GPU) must not cost the line: after 240 s rank 0 prints the line without it
    dp = None
    printed = [False]

    def emit():
        if rank == 0 and not printed[0]:
            printed[0] = True
            print(json.dumps(out), flush=True)

    timer = None
    if use_dist and args.backend == "nccl" and not args.no_dp_train and not args.no_secondary:
        import threading

        def give_up():
            if rank == 0 and "secondary_dp" not in out:
                out["secondary_dp"] = {"error": "timeout after 240 s (data-parallel training step)"}
            emit()
            os._exit(0)

        timer = threading.Timer(240.0, give_up)
        timer.daemon = True
        timer.start()
        try:
            dp = dp_training_secondary(device, world, rank, args.dp_graph)
        except Exception as err:
            dp = {"error": f"{type(err).__name__}: {err}"}
    if rank == 0 and dp is not None:
        out["secondary_dp"] = dp
    emit()
    if use_dist:
        # (a rank that failed in the secondary above may leave its peers inside a collective: the line is out, the watchdog is
        # still armed, and an error in the farewell barrier is not an error of the measurement)
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    if timer is not None:
        timer.cancel()


if __name__ == "__main__":
    main()
