#!/usr/bin/env python3
"""Secondary metric: RQ-VAE training-step throughput on one MI355X (reference index/trainer.py:111-120;
the reference's CPU path does ~7-12 k items/s at batch 2048).

    python tools/train_probe.py [--in_dim 768] [--batch 2048] [--steps 30] [--bn] [--ema]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd  # noqa: E402
from lcrec_amd import ops  # noqa: E402
from lcrec_amd.trainer import linear_schedule_with_warmup  # noqa: E402


def trainer_probe(a):
    import tempfile
    from lcrec_amd import main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    dev = "cuda:0"
    n = a.batch * 16
    with tempfile.TemporaryDirectory() as tmp:
        argv = ["--data_path", "unused", "--ckpt_dir", tmp, "--device", dev, "--batch_size", str(a.batch), "--epochs", "4",
                "--no_kmeans_init", "--num_emb_list", "256", "256", "256", "256",          # run.sh: 4 levels, Sinkhorn on the last
                "--sk_epsilons", "0.0", "0.0", "0.0", "0.0" if a.no_sk else "0.003"] \
            + (["--bn", "True"] if a.bn else ["--no_bn"]) + (["--strict_nan_check"] if a.strict else []) + ["--train_engine", a.engine]
        args = cli.parse_args(argv)
        ctx = None
        if a.rccl1:
            # a ONE-rank RCCL group: the data-parallel step with every exchange going through the backend (what an 8-GPU
            # rank launches, minus the wire time)
            from lcrec_amd import dist as ldist
            os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
            ctx = ldist.init_from_env(args, backend="nccl", force=True)
        cli.seed_everything(2024)
        model = cli.build_model(args, a.in_dim)
        data = torch.randn((n, a.in_dim), device=dev)
        loader = DeviceLoader(data, a.batch, True, dev)
        trainer = Trainer(args, model, len(loader))
        if ctx is not None:
            ldist.attach(trainer, ctx)
        trainer._train_epoch(loader, 0)
        torch.cuda.synchronize()
        prof = None
        if a.cprofile:                                   # host side of the timed epochs only (main thread)
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        epochs = max(1, a.steps // len(loader))
        for e in range(epochs):
            trainer._train_epoch(loader, e + 1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats("tottime").print_stats(30)
        steps = epochs * len(loader)
        eng = trainer.engine
        if ctx is not None:
            ldist.shutdown(ctx)
        print(f"Trainer._train_epoch{' [one-rank RCCL group, %d collectives/step]' % (eng.collectives // 2 if eng is not None else -1) if a.rccl1 else ''}: "
              f"in_dim {a.in_dim} batch {a.batch} levels 4 sinkhorn {not a.no_sk} bn {a.bn} strict_nan_check {a.strict} "
              f"engine {'hipGraph (%d replays)' % eng.graph_replays if eng is not None else 'off (autograd path)'}: "
              f"{dt / steps * 1e3:.3f} ms/step, {a.batch * steps / dt:,.0f} items/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in_dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--bn", action="store_true")
    ap.add_argument("--ema", action="store_true")
    ap.add_argument("--no_sk", action="store_true")
    ap.add_argument("--strict", action="store_true", help="with --trainer: the per-step NaN host sync of the reference")
    ap.add_argument("--cprofile", action="store_true", help="with --trainer: cProfile of the timed epochs")
    ap.add_argument("--engine", default="auto", choices=["auto", "off"], help="with --trainer: --train_engine of lcrec_amd.main")
    ap.add_argument("--rccl1", action="store_true", help="with --trainer: data-parallel step on a one-rank RCCL group")
    ap.add_argument("--trainer", action="store_true",
                    help="time lcrec_amd.trainer.Trainer._train_epoch itself (loader, NaN check, fused AdamW, schedule)")
    a = ap.parse_args()
    if a.trainer:
        return trainer_probe(a)
    dev = torch.device("cuda:0")
    torch.manual_seed(2024)
    model = lcrec_amd.RQVAE(in_dim=a.in_dim, num_emb_list=[256] * 4, e_dim=32, layers=[2048, 1024, 512, 256, 128, 64],
                            bn=a.bn, kmeans_init=False, sk_epsilons=[0.0, 0.0, 0.0, 0.0 if a.no_sk else 0.003],
                            sk_iters=50, ema_decay=0.99 if a.ema else None).to(dev)
    x = torch.randn((a.batch, a.in_dim), device=dev)
    with torch.no_grad():   # data-scale codebooks
        z = model.encoder(x)
        for q in model.rq.vq_layers:
            q.embedding.weight.copy_(z[torch.randperm(a.batch, device=dev)[:256]] * 0.5)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)      # what Trainer builds on a HIP device
    sched = linear_schedule_with_warmup(opt, 10, 10000)

    def step():
        opt.zero_grad()
        out, rq_loss, _ = model(x)
        loss, _ = model.compute_loss(out, rq_loss, xs=x)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ops.trace_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr = ops.trace_collect()
    ops.trace_enable(False)
    print(f"in_dim {a.in_dim} batch {a.batch} bn {a.bn} ema {a.ema}: {dt / a.steps * 1e3:.3f} ms/step, "
          f"{a.batch * a.steps / dt:,.0f} items/s, loss {loss.item():.4f}")
    lib_ms = sum(v[1] for v in tr.values()) / a.steps
    print(f"  lcrec kernels: {lib_ms:.3f} ms/step ->", {k: (v[0] // a.steps, round(v[1] / a.steps, 3)) for k, v in tr.items()})


if __name__ == "__main__":
    main()
