"""torch-CPU restatement of the reference's RQ-VAE arithmetic, as pure functions
over a state-dict.

TEST INFRASTRUCTURE ONLY (see oracle/lcrec_oracle.c header): used by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

Why a second oracle next to lcrec_oracle.c: the reference IS a sequence of
torch CPU ops, and what `north_star` calls "the reference index/trainer.py CPU
path" is that sequence.  This file issues the same aten ops in the same order
(so on one machine it is bit-identical to the imported reference -- checked by
oracle/make_golden.py, which records the comparison in tests/golden/manifest.json)
and is what bench.py times on the GPU box's host cores.  It also carries the
pieces that are defined by torch semantics rather than by a summation order:
autograd through the straight-through estimator, AdamW, the fp64 Sinkhorn.

Restates (paths relative to the reference root):
  index/models/layers.py:7-43,85-108   MLP stack, sinkhorn_algorithm
  index/models/vq.py:51-99             centring, distance, argmin / Sinkhorn argmax, losses, STE
  index/models/rq.py:39-55             residual loop
  index/models/rqvae.py:61-85          forward, get_indices, compute_loss
  index_improve/models/vq.py:147-184,205-217   EMA statistics / blend, utilisation
  index/trainer.py:111-120             one optimisation step
"""
import torch
import torch.nn.functional as F


def linear_slot(layer, bn):
    """Index of layer `layer`'s nn.Linear inside MLPLayers.mlp_layers (layers.py:18-30):
    groups are [Dropout, Linear, (BatchNorm1d), ReLU]; the last group has neither BN nor ReLU."""
    return layer * (4 if bn else 3) + 1


def mlp(sd, prefix, x, n_layers, bn, training=False):
    """MLPLayers.forward (layers.py:42).  `sd` maps state-dict names to tensors; in training mode
    the BatchNorm running statistics in `sd` are updated in place, as nn.BatchNorm1d does."""
    for l in range(n_layers):
        slot = linear_slot(l, bn)
        x = F.linear(x, sd[f"{prefix}.mlp_layers.{slot}.weight"], sd[f"{prefix}.mlp_layers.{slot}.bias"])
        if l != n_layers - 1:
            if bn:
                b = f"{prefix}.mlp_layers.{slot + 1}"
                if training:
                    sd[b + ".num_batches_tracked"] += 1
                x = F.batch_norm(x, sd[b + ".running_mean"], sd[b + ".running_var"], sd[b + ".weight"],
                                 sd[b + ".bias"], training, 0.1, 1e-5)
            x = F.relu(x)
    return x


def sinkhorn(distances, epsilon, iterations):
    """layers.py:85-108 on an fp64 [B, K] matrix."""
    with torch.no_grad():
        Q = torch.exp(-distances / epsilon)
        B, K = Q.shape
        Q /= Q.sum(-1, keepdim=True).sum(-2, keepdim=True)
        for _ in range(iterations):
            Q /= torch.sum(Q, dim=1, keepdim=True)
            Q /= B
            Q /= torch.sum(Q, dim=0, keepdim=True)
            Q /= K
        Q *= B
    return Q


def centre_distances(d):
    """vq.py:51-61: map d onto [-1, 1] using the GLOBAL max and min of the [B, K] matrix."""
    hi, lo = d.max(), d.min()
    mid = (hi + lo) / 2
    amp = hi - mid + 1e-5
    assert amp > 0
    return (d - mid) / amp


def distances(latent, codebook):
    """vq.py:71-73."""
    return torch.sum(latent ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1, keepdim=True).t() \
        - 2 * torch.matmul(latent, codebook.t())


def vq(x, codebook, beta, use_sk, sk_epsilon, sk_iters, force_idx=None):
    """VectorQuantizer.forward (vq.py:63-99) without the k-means lazy init.
    Returns (straight-through x_q, loss, indices).  `force_idx`: take these codes instead of searching (the tests' way to
    differentiate exactly the problem another evaluation solved when a near-tied row went to another code there)."""
    latent = x.view(-1, codebook.shape[1])
    d = distances(latent, codebook)
    if force_idx is not None:
        idx = force_idx.reshape(-1).to(torch.int64)
    elif not use_sk or sk_epsilon <= 0:
        idx = torch.argmin(d, dim=-1)
    else:
        Q = sinkhorn(centre_distances(d).double(), sk_epsilon, sk_iters)
        idx = torch.argmax(Q, dim=-1)
    q = F.embedding(idx, codebook).view(x.shape)
    loss = F.mse_loss(q, x.detach()) + beta * F.mse_loss(q.detach(), x)
    q = x + (q - x).detach()
    return q, loss, idx.view(x.shape[:-1])


def rq(x, codebooks, beta, use_sk, sk_epsilons, sk_iters, hook=None, force_idx=None):
    """ResidualVectorQuantizer.forward (rq.py:39-55).  `hook(level, residual, idx)` sees each level's input."""
    losses, indices = [], []
    total, residual = 0, x
    for l, cb in enumerate(codebooks):
        q, loss, idx = vq(residual, cb, beta, use_sk, sk_epsilons[l], sk_iters, None if force_idx is None else force_idx[..., l])
        if hook is not None:
            hook(l, residual, idx)
        residual = residual - q
        total = total + q
        losses.append(loss)
        indices.append(idx)
    return total, torch.stack(losses).mean(), torch.stack(indices, dim=-1)


class Spec:
    """Hyper-parameters of an RQVAE (rqvae.py:11-44) without the module machinery."""

    def __init__(self, in_dim, num_emb_list, e_dim, layers, bn=False, loss_type="mse", quant_loss_weight=1.0,
                 beta=0.25, sk_epsilons=None, sk_iters=100):
        self.dims = [in_dim] + list(layers) + [e_dim]
        self.num_emb_list = list(num_emb_list)
        self.bn = bn
        self.loss_type = loss_type
        self.quant_loss_weight = quant_loss_weight
        self.beta = beta
        self.sk_epsilons = list(sk_epsilons) if sk_epsilons is not None else [0.0] * len(num_emb_list)
        self.sk_iters = sk_iters
        self.n_layers = len(self.dims) - 1
        # rq.py:30 zips the two lists: fewer epsilons silently means fewer levels
        self.levels = min(len(self.num_emb_list), len(self.sk_epsilons))

    def codebooks(self, sd):
        return [sd[f"rq.vq_layers.{l}.embedding.weight"] for l in range(self.levels)]


def forward(spec, sd, x, use_sk=True, training=False, hook=None, force_idx=None):
    """RQVAE.forward (rqvae.py:61-66): returns (out, rq_loss, indices)."""
    z = mlp(sd, "encoder", x, spec.n_layers, spec.bn, training)
    q, rq_loss, idx = rq(z, spec.codebooks(sd), spec.beta, use_sk, spec.sk_epsilons, spec.sk_iters, hook, force_idx)
    out = mlp(sd, "decoder", q, spec.n_layers, spec.bn, training)
    return out, rq_loss, idx


def get_indices(spec, sd, x, use_sk=False):
    """RQVAE.get_indices (rqvae.py:68-72): the path the headline metric measures."""
    with torch.no_grad():
        z = mlp(sd, "encoder", x, spec.n_layers, spec.bn, False)
        _, _, idx = rq(z, spec.codebooks(sd), spec.beta, use_sk, spec.sk_epsilons, spec.sk_iters)
    return idx


def compute_loss(spec, out, quant_loss, xs):
    """rqvae.py:74-85."""
    if spec.loss_type == "mse":
        recon = F.mse_loss(out, xs, reduction="mean")
    elif spec.loss_type == "l1":
        recon = F.l1_loss(out, xs, reduction="mean")
    else:
        raise ValueError("incompatible loss type")
    return recon + spec.quant_loss_weight * quant_loss, recon


def ema_step(codebook, ema_count, ema_sum, latent, idx, decay, eps):
    """index_improve/models/vq.py:151-184, in place on the three tensors."""
    with torch.no_grad():
        K, e = codebook.shape
        flat = idx.view(-1)
        count = torch.zeros(K, dtype=torch.float32)
        count.scatter_add_(0, flat, torch.ones_like(flat, dtype=torch.float32))
        ema_count.mul_(decay).add_(count, alpha=1 - decay)
        dw = torch.zeros(K, e, dtype=latent.dtype)
        lat = latent.view(-1, e)
        for dim in range(e):
            dw[:, dim].index_add_(0, flat, lat[:, dim])
        ema_sum.mul_(decay).add_(dw, alpha=1 - decay)
        centre = ema_sum / (ema_count.unsqueeze(1) + eps)
        used = ema_count > eps
        if used.any():
            rate = 1 - decay
            codebook[used] = codebook[used] * (1 - rate) + centre[used] * rate
    return count, dw


def utilisation(ema_count, eps, threshold):
    """index_improve/models/vq.py:205-217."""
    usage = ema_count / (ema_count.sum() + eps)
    used = int((usage > threshold).sum().item())
    return {"utilization": used / ema_count.numel(), "used_codes": used, "total_codes": ema_count.numel()}
