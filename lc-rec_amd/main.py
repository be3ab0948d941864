"""Command line of the RQ-VAE trainer -- the flags, defaults and quirks of the reference's
index/main.py:14-49 (and index_improve/main.py:48-51 for the EMA options), so index/run.sh keeps
working unchanged:

    python -m lcrec_amd.main --data_path .../Games.emb-llama-td.npy --num_emb_list 256 256 256 256 ...
    (or `cd index && python main.py ...` through the shim in index/main.py)

Quirk kept on purpose: `--bn` and `--kmeans_init` are `type=bool`, so ANY
non-empty value -- including the `--bn False` in index/run.sh:9 -- parses as True.  Use `--no_bn` /
`--no_kmeans_init` (additions) to switch them off explicitly.
"""
import argparse
import logging
import random

import numpy as np
import torch

from .datasets import DeviceLoader, EmbDataset
from .rqvae import RQVAE
from .trainer import Trainer


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Index")

    parser.add_argument('--lr', type=float, default=1e-3, help='learning rate')
    parser.add_argument('--epochs', type=int, default=5000, help='number of epochs')
    parser.add_argument('--batch_size', type=int, default=2048, help='batch size')
    parser.add_argument('--num_workers', type=int, default=4, help='kept for CLI compatibility; batches are HBM-resident')
    parser.add_argument('--eval_step', type=int, default=50, help='eval step')
    parser.add_argument('--learner', type=str, default="AdamW", help='optimizer')
    parser.add_argument('--lr_scheduler_type', type=str, default="constant", help='scheduler')
    parser.add_argument('--warmup_epochs', type=int, default=50, help='warmup epochs')
    parser.add_argument("--data_path", type=str, default="../data/Games/Games.emb-llama-td.npy", help="Input data path.")

    parser.add_argument("--weight_decay", type=float, default=0.0, help='l2 regularization weight')
    parser.add_argument("--dropout_prob", type=float, default=0.0, help="dropout ratio")
    parser.add_argument("--bn", type=bool, default=False, help="use bn or not (type=bool: any value is True)")
    parser.add_argument("--loss_type", type=str, default="mse", help="loss_type")
    parser.add_argument("--kmeans_init", type=bool, default=True, help="use kmeans_init or not (type=bool)")
    parser.add_argument("--kmeans_iters", type=int, default=100, help="max kmeans iters")
    parser.add_argument('--sk_epsilons', type=float, nargs='+', default=[0.0, 0.0, 0.0], help="sinkhorn epsilons")
    parser.add_argument("--sk_iters", type=int, default=50, help="max sinkhorn iters")

    parser.add_argument("--device", type=str, default="cuda:0", help="HIP device (cuda:N)")

    parser.add_argument('--num_emb_list', type=int, nargs='+', default=[256, 256, 256], help='emb num of every vq')
    parser.add_argument('--e_dim', type=int, default=32, help='vq codebook embedding size')
    parser.add_argument('--quant_loss_weight', type=float, default=1.0, help='vq quantion loss weight')
    parser.add_argument("--beta", type=float, default=0.25, help="Beta for commitment loss")
    parser.add_argument('--layers', type=int, nargs='+', default=[2048, 1024, 512, 256, 128, 64],
                        help='hidden sizes of every layer')

    parser.add_argument('--save_limit', type=int, default=5)
    parser.add_argument("--ckpt_dir", type=str, default="", help="output directory for model")

    # index_improve/main.py:48-51 -- EMA codebook update; off (None) reproduces index/
    parser.add_argument("--ema_decay", type=float, default=None, help="EMA decay rate for codebook update")
    parser.add_argument("--epsilon", type=float, default=1e-5, help="Small epsilon for numerical stability")
    parser.add_argument("--reset_threshold", type=float, default=1e-5, help="Threshold for codebook reset")
    parser.add_argument("--reset_interval", type=int, default=1000, help="Interval (steps) for codebook reset")

    # additions
    parser.add_argument("--no_bn", action="store_true", help="force bn=False (the bool flag cannot)")
    parser.add_argument("--no_kmeans_init", action="store_true", help="force kmeans_init=False")
    parser.add_argument("--strict_nan_check", action="store_true",
                        help="raise 'Training loss is nan' before the offending step's backward (a host sync per step, the "
                             "reference's timing) instead of one step later")
    parser.add_argument("--reset_seed", type=int, default=None,
                        help="seed a dedicated device generator for the dead-code reset draws of --ema_decay runs "
                             "(default: torch's global generators, as the reference)")
    parser.add_argument("--train_engine", type=str, default="auto", choices=["auto", "off"],
                        help="auto = run the training step as one captured hipGraph when the configuration allows it "
                             "(lcrec_amd.engine); off = always the autograd path")
    parser.add_argument("--dp_graph", type=str, default="auto", choices=["auto", "on", "off"],
                        help="data-parallel runs: capture the step WITH its RCCL collectives into the hipGraph (on), or launch "
                             "the same straight line eagerly (off).  auto = off for more than one rank: the captured "
                             "multi-rank step has not been run on more than one GPU yet (DESIGN.md section 6)")
    parser.add_argument("--kmeans_impl", type=str, default="sklearn", choices=["sklearn", "device"],
                        help="sklearn = the reference's host KMeans call; device = k-means++/Lloyd in HBM")
    args = parser.parse_args(argv)
    from . import layers
    layers.KMEANS_IMPL = args.kmeans_impl
    if args.no_bn:
        args.bn = False
    if args.no_kmeans_init:
        args.kmeans_init = False
    return args


def seed_everything(seed=2024):
    """index/main.py:54-60."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def build_model(args, in_dim):
    return RQVAE(in_dim=in_dim, num_emb_list=args.num_emb_list, e_dim=args.e_dim, layers=args.layers,
                 dropout_prob=args.dropout_prob, bn=args.bn, loss_type=args.loss_type,
                 quant_loss_weight=args.quant_loss_weight, beta=args.beta, kmeans_init=args.kmeans_init,
                 kmeans_iters=args.kmeans_iters, sk_epsilons=args.sk_epsilons, sk_iters=args.sk_iters,
                 ema_decay=getattr(args, "ema_decay", None), epsilon=getattr(args, "epsilon", 1e-5),
                 reset_threshold=getattr(args, "reset_threshold", 1e-5),
                 reset_interval=getattr(args, "reset_interval", 1000))


def main(argv=None):
    seed_everything(2024)
    args = parse_args(argv)
    print("=================================================")
    print(args)
    print("=================================================")
    logging.basicConfig(level=logging.DEBUG)

    from . import dist as ldist
    ctx = ldist.init_from_env(args)          # single process unless launched under torchrun
    # to a GPU the file goes from the page cache through the pinned ring of EmbDataset.to_device: no host copy of the matrix
    data = EmbDataset(args.data_path, mmap=str(args.device).startswith("cuda"))
    model = build_model(args, data.dim)
    if getattr(args, "reset_seed", None) is not None and str(args.device).startswith("cuda"):
        for l, q in enumerate(model.rq.vq_layers):      # one stream of draws per level, the same on every rank
            q.reset_generator = torch.Generator(device=args.device).manual_seed(int(args.reset_seed) + l)
    if ctx.rank == 0:
        print(model)
    loader = DeviceLoader(data, batch_size=args.batch_size, shuffle=True, device=args.device, rank=ctx.rank,
                          world_size=ctx.world_size)
    trainer = Trainer(args, model, len(loader))
    ldist.attach(trainer, ctx)
    best_loss, best_collision_rate = trainer.fit(loader)
    if ctx.rank == 0:
        print("Best Loss", best_loss)
        print("Best Collision Rate", best_collision_rate)
    ldist.shutdown(ctx)
    return best_loss, best_collision_rate


if __name__ == '__main__':
    main()
