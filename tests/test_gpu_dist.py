"""Item-sharded data-parallel training on real kernels: two ranks (gloo rendezvous, both on cuda:0 -- the
box has one GPU; on the 8-GPU node the same code runs over RCCL) must reproduce the single-process epoch on
the same global batches: gradients all-reduced with n_local weights, Sinkhorn on the gathered global batch,
code statistics all-reduced before the EMA update, rank-0 evaluation over gathered indices."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, tmp, ema, bn=False):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if world > 1:
        os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    else:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
    from lcrec_amd import dist as ldist, main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    argv = ["--data_path", "unused", "--ckpt_dir", os.path.join(tmp, f"ck{world}"), "--device", "cuda:0", "--batch_size", "96",
            "--epochs", "2", "--layers", "64", "32", "--e_dim", "16", "--num_emb_list", "32", "32", "32",
            "--sk_epsilons", "0.0", "0.0", "0.003", "--no_kmeans_init"] + (["--bn", "True"] if bn else ["--no_bn"]) \
        + (["--ema_decay", "0.95"] if ema else [])
    args = cli.parse_args(argv)
    ctx = ldist.init_from_env(args, backend="gloo")
    cli.seed_everything(2024)
    model = cli.build_model(args, 48)
    g = torch.Generator().manual_seed(7)
    data = torch.randn((300, 48), generator=g).to("cuda:0")          # 300 = 3 batches of 96 + a ragged one of 12
    loader = DeviceLoader(data, 96, True, "cuda:0", rank=ctx.rank, world_size=ctx.world_size)
    trainer = Trainer(args, model, len(loader))
    ldist.attach(trainer, ctx)
    torch.manual_seed(11)                                              # same shuffles in every configuration
    losses = [trainer._train_epoch(loader, e) for e in range(2)]
    rate = trainer._valid_epoch(DeviceLoader(data, 96, False, "cuda:0", rank=ctx.rank, world_size=ctx.world_size))
    if ctx.rank == 0:
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        np.savez(os.path.join(tmp, f"world{world}.npz"), losses=np.array(losses), rate=rate,
                 ckpt_dirs=np.array(sorted(os.listdir(os.path.join(tmp, f"ck{world}")))), **sd)
    ldist.shutdown(ctx)


@pytest.mark.parametrize("ema,bn", [(False, False), (True, False), (False, True)])
def test_two_ranks_reproduce_the_single_process_epoch(hip, tmp_path, ema, bn):
    """bn=True is the de-facto recipe (index/run.sh:9 passes `--bn False`, which type=bool parses as True): its batch
    statistics must be those of the GLOBAL batch (SyncBatchNorm semantics, one all-reduce per layer and direction in
    layers._BatchNormAct), including on the ragged last batch where the ranks hold 6 rows each of 12."""
    tmp = str(tmp_path)
    mp.spawn(_run, args=(1, 0, tmp, ema, bn), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), tmp, ema, bn), nprocs=2, join=True)
    one, two = np.load(os.path.join(tmp, "world1.npz")), np.load(os.path.join(tmp, "world2.npz"))
    assert len(two["ckpt_dirs"]) == 1            # one time-stamped checkpoint directory for the job, not one per rank
    np.testing.assert_allclose(two["losses"], one["losses"], rtol=2e-4)
    assert float(two["rate"]) == pytest.approx(float(one["rate"]), abs=2e-2)
    worst = 0.0
    for k in one.files:
        if k in ("losses", "rate", "ckpt_dirs"):
            continue
        a, b = one[k], two[k]
        assert a.shape == b.shape, k
        if bn and k.endswith(".bias"):
            # the bias of a Linear that feeds a BatchNorm has an exactly-zero gradient -- rounding noise in any
            # implementation -- which Adam normalises into +-lr steps: not comparable between two runs of anything
            part, _, idx, _ = k.split(".")
            if f"{part}.mlp_layers.{int(idx) + 1}.running_mean" in one.files:
                continue
        if np.issubdtype(a.dtype, np.floating):
            scale = max(1e-6, float(np.abs(a).max()))
            worst = max(worst, float(np.abs(a - b).max()) / scale)
    assert worst < 5e-3, worst        # different reduction orders, a few Sinkhorn near-ties; not bitwise
