"""RQ-VAE -- host-side mirror of the reference's index/models/rqvae.py (RQVAE :10-85; improve fork:
ema_decay/epsilon/reset_* arguments, use_ema, get_codebook_usage).  Same constructor, methods,
sub-module names and state-dict keys; the arithmetic runs on the MI355X through include/lcrec.h.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .layers import MLPLayers
from .rq import ResidualVectorQuantizer


class RQVAE(nn.Module):
    def __init__(self, in_dim=768, num_emb_list=None, e_dim=64, layers=None, dropout_prob=0.0, bn=False,
                 loss_type="mse", quant_loss_weight=1.0, beta=0.25, kmeans_init=False, kmeans_iters=100,
                 sk_epsilons=None, sk_iters=100, ema_decay=None, epsilon=1e-5, reset_threshold=1e-5,
                 reset_interval=1000):
        super().__init__()
        self.in_dim = in_dim
        self.num_emb_list = num_emb_list
        self.e_dim = e_dim
        self.layers = layers
        self.dropout_prob = dropout_prob
        self.bn = bn
        self.loss_type = loss_type
        self.quant_loss_weight = quant_loss_weight
        self.beta = beta
        self.kmeans_init = kmeans_init
        self.kmeans_iters = kmeans_iters
        self.sk_epsilons = sk_epsilons
        self.sk_iters = sk_iters

        self.encode_layer_dims = [self.in_dim] + self.layers + [self.e_dim]
        self.encoder = MLPLayers(layers=self.encode_layer_dims, dropout=self.dropout_prob, bn=self.bn)
        self.rq = ResidualVectorQuantizer(num_emb_list, e_dim, beta=self.beta, kmeans_init=self.kmeans_init,
                                          kmeans_iters=self.kmeans_iters, sk_epsilons=self.sk_epsilons,
                                          sk_iters=self.sk_iters, ema_decay=ema_decay, epsilon=epsilon,
                                          reset_threshold=reset_threshold, reset_interval=reset_interval)
        self.decode_layer_dims = self.encode_layer_dims[::-1]
        self.decoder = MLPLayers(layers=self.decode_layer_dims, dropout=self.dropout_prob, bn=self.bn)

    def forward(self, x, use_sk=True, use_ema=True):
        x = self.encoder(x)
        x_q, rq_loss, indices = self.rq(x, use_sk=use_sk, use_ema=use_ema)
        out = self.decoder(x_q)
        return out, rq_loss, indices

    @torch.no_grad()
    def get_indices(self, xs, use_sk=False):
        """rqvae.py:68-72.  In eval mode with hard assignment this is ONE library call
        (lcrec_encode_assign): encoder GEMM chain + all quantiser levels, nothing else computed."""
        levels = list(self.rq.vq_layers)
        hard = (not use_sk) or all(q.sk_epsilon <= 0 for q in levels)
        if hard and not self.training and self.encoder.fusable() and xs.dim() == 2:
            Ws, bs, scs, shs = self.encoder.folded()
            flat, ks = ops.flatten_codebooks([q.embedding.weight.detach() for q in levels])
            return ops.encode_assign(xs, Ws, bs, flat, ks, scs, shs)[0]
        x_e = self.encoder(xs)
        _, _, indices = self.rq(x_e, use_sk=use_sk, use_ema=False)
        return indices

    @torch.no_grad()
    def get_indices_audited(self, xs, tie_tau=None):
        """get_indices(xs, use_sk=False) plus the near-tie audit of the quantiser (include/lcrec.h, lcrec_rq_assign):
        returns (indices int64 [n, L], neartie int32 [n], margin float32 [n, L]).  Bit l of neartie[i] is set when the
        two best codes of item i at level l are closer than tie_tau * (xx + cc) -- the rows on which the reference's own
        CPU arithmetic (vq.py:71-75, summation order unspecified) may choose differently; all other rows carry the
        reference's tuple (measured rates: ops.NEARTIE_TAU).  Eval mode, 2-d input, ReLU encoder."""
        if self.training or xs.dim() != 2 or not self.encoder.fusable():
            raise ops._lib.LcrecError("get_indices_audited: eval mode, a 2-d batch and a ReLU encoder are required")
        levels = list(self.rq.vq_layers)
        Ws, bs, scs, shs = self.encoder.folded()
        flat, ks = ops.flatten_codebooks([q.embedding.weight.detach() for q in levels])
        audit = {}
        idx = ops.encode_assign(xs, Ws, bs, flat, ks, scs, shs, audit=audit,
                                tie_tau=ops.NEARTIE_TAU if tie_tau is None else tie_tau)[0]
        return idx, audit["neartie"], audit["margin"]

    def compute_loss(self, out, quant_loss, xs=None):
        if self.loss_type == "mse":
            loss_recon = F.mse_loss(out, xs, reduction="mean")
        elif self.loss_type == "l1":
            loss_recon = F.l1_loss(out, xs, reduction="mean")
        else:
            raise ValueError("incompatible loss type")
        loss_total = loss_recon + self.quant_loss_weight * quant_loss
        return loss_total, loss_recon

    def get_codebook_usage(self):
        return self.rq.get_codebook_usage()
