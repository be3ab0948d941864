"""Residual quantiser -- host-side mirror of the reference's index/models/rq.py
(ResidualVectorQuantizer :13-55) and of the improve fork's additions
(index_improve/models/rq.py: use_ema, get_codebook_usage).  The arithmetic is in quantize.py.
"""
import torch
import torch.nn as nn

from . import ops
from .quantize import quantize
from .vq import VectorQuantizer


class ResidualVectorQuantizer(nn.Module):
    """rq.py:13-55.  n_e_list and sk_epsilons are zipped, so the shorter one sets the depth (:30)."""

    def __init__(self, n_e_list, e_dim, sk_epsilons, beta=0.25, kmeans_init=False, kmeans_iters=100,
                 sk_iters=100, ema_decay=None, epsilon=1e-5, reset_threshold=1e-5, reset_interval=1000):
        super().__init__()
        self.n_e_list = n_e_list
        self.e_dim = e_dim
        self.num_quantizers = len(n_e_list)
        self.beta = beta
        self.kmeans_init = kmeans_init
        self.kmeans_iters = kmeans_iters
        self.sk_epsilons = sk_epsilons
        self.sk_iters = sk_iters
        self.vq_layers = nn.ModuleList([
            VectorQuantizer(n_e, e_dim, beta=beta, kmeans_init=kmeans_init, kmeans_iters=kmeans_iters,
                            sk_epsilon=sk_epsilon, sk_iters=sk_iters, ema_decay=ema_decay, epsilon=epsilon,
                            reset_threshold=reset_threshold, reset_interval=reset_interval)
            for n_e, sk_epsilon in zip(n_e_list, sk_epsilons)])

    def get_codebook(self):
        return torch.stack([q.get_codebook() for q in self.vq_layers])

    def forward(self, x, use_sk=True, use_ema=True):
        latent = x.reshape(-1, self.e_dim)
        layers = list(self.vq_layers)
        if self.training and any(not q.initted for q in layers):
            self._lazy_kmeans(latent, use_sk)
        x_q, loss, idx, side = quantize(latent, layers, self.beta, use_sk, self.training)
        if self.training and use_ema:
            for t, q in enumerate(layers):
                if q.ema_decay is not None:
                    q.ema_step(side["stats"][t], side["resid_in"][t])
        return x_q.view(x.shape), loss, idx.view(*x.shape[:-1], len(layers))

    @torch.no_grad()
    def _lazy_kmeans(self, latent, use_sk):
        """vq.py:67-68 inside the residual loop: level l is seeded from the residual that reaches it,
        computed with the (already seeded) levels before it."""
        r = latent.detach().contiguous()
        for q in self.vq_layers:
            if not q.initted:
                q.init_emb(r)
            w = q.embedding.weight.detach().contiguous()
            if use_sk and q.sk_epsilon > 0:
                col = ops.sinkhorn_assign(r, w, q.sk_epsilon, q.sk_iters)
            else:
                flat, ks = ops.flatten_codebooks([w])
                col = ops.rq_assign(r, flat, ks)[0][:, 0].contiguous()
            _, r, _ = ops.rq_apply_level(r, w, col)

    def get_codebook_usage(self):
        """index_improve/models/rq.py:67-74."""
        out = []
        for i, q in enumerate(self.vq_layers):
            s = q.get_codebook_usage()
            s["quantizer_id"] = i
            out.append(s)
        return out
