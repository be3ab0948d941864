"""Single-level vector quantiser -- host-side mirror of the reference's index/models/vq.py
(VectorQuantizer :9-99) plus the improve fork's EMA statistics, dead-code reset and utilisation
(index_improve/models/vq.py:41-42, 79-114, 147-193, 205-217; enabled by passing ema_decay).
Values come from the HIP kernels through quantize.py; this class owns the parameters/buffers
with the reference's names so checkpoints interchange.
"""
import torch
import torch.nn as nn

from . import dist as ldist
from . import ops
from . import layers
from .quantize import quantize


class VectorQuantizer(nn.Module):

    def __init__(self, n_e, e_dim, beta=0.25, kmeans_init=False, kmeans_iters=10, sk_epsilon=0.003, sk_iters=100,
                 ema_decay=None, epsilon=1e-5, reset_threshold=1e-5, reset_interval=1000):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.kmeans_init = kmeans_init
        self.kmeans_iters = kmeans_iters
        self.sk_epsilon = sk_epsilon          # mutated from outside by the index generator
        self.sk_iters = sk_iters
        self.ema_decay = ema_decay            # None = plain index/ quantiser (no EMA buffers in the state dict)
        self.epsilon = epsilon
        self.reset_threshold = reset_threshold
        self.reset_interval = reset_interval
        self.step_count = 0
        # draws of the dead-code reset (index_improve/models/vq.py:79-114): None = torch's global generators, as the
        # reference; a torch.Generator on the codebook's device (main.py --reset_seed) makes a run reproducible
        self.reset_generator = None

        self.embedding = nn.Embedding(self.n_e, self.e_dim)
        if ema_decay is not None:
            self.register_buffer("_ema_cluster_size", torch.zeros(n_e))
            self.register_buffer("_ema_w", torch.zeros(n_e, e_dim))
        if not kmeans_init:
            self.initted = True
            self.embedding.weight.data.uniform_(-1.0 / self.n_e, 1.0 / self.n_e)
        else:
            self.initted = False
            self.embedding.weight.data.zero_()

    def get_codebook(self):
        return self.embedding.weight

    def get_codebook_entry(self, indices, shape=None):
        z_q = self.embedding(indices)
        if shape is not None:
            z_q = z_q.view(shape)
        return z_q

    def init_emb(self, data):
        """vq.py:40-49: seed the codebook with k-means centres of `data` (host sklearn, layers.kmeans)."""
        world = ldist.current()
        if world.enabled:                      # rank 0 clusters the gathered batch, everyone gets its centres
            data = world.gather_rows(data.contiguous())
            centers = torch.zeros_like(self.embedding.weight.data)
            # the device k-means seeds its draws from torch's global CPU generator, which the loaders' shuffles share:
            # EVERY rank makes the draw, so the ranks' generators -- and with them the epoch orders -- stay in step
            generator = None
            if layers.KMEANS_IMPL == "device":
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                generator = torch.Generator(device=data.device).manual_seed(seed)
            if world.rank == 0:
                centers.copy_(layers.kmeans(data, self.n_e, self.kmeans_iters, generator=generator))
            world.broadcast_(centers, src=0)
        else:
            centers = layers.kmeans(data, self.n_e, self.kmeans_iters)
        self.embedding.weight.data.copy_(centers)
        self.initted = True

    @staticmethod
    def center_distance_for_constraint(distances):
        """vq.py:51-61 as device tensor ops (API parity; the fused Sinkhorn kernel does this itself)."""
        max_distance = distances.max()
        min_distance = distances.min()
        middle = (max_distance + min_distance) / 2
        amplitude = max_distance - middle + 1e-5
        assert amplitude > 0
        return (distances - middle) / amplitude

    def forward(self, x, use_sk=True, use_ema=True):
        latent = x.reshape(-1, self.e_dim)
        if not self.initted and self.training:
            self.init_emb(latent.detach())
        x_q, loss, idx, side = quantize(latent, [self], self.beta, use_sk, self.training)
        if self.training and use_ema and self.ema_decay is not None:
            self.ema_step(side["stats"][0], side["resid_in"][0])
        return x_q.view(x.shape), loss, idx.view(x.shape[:-1])

    # ------------------------------------------------------------------ improve fork
    @torch.no_grad()
    def ema_step(self, stats, latent):
        """index_improve/models/vq.py:147-193 given this step's (count, per-code sum)."""
        count, total = stats
        world = ldist.current()
        if world.enabled:                      # statistics of the global batch (dist.py)
            count, total = count.clone(), total.clone()
            world.all_reduce_sum_(count, total)
            latent = world.gather_rows(latent)
        ops.ema_update(self._ema_cluster_size, self._ema_w, self.embedding.weight.data, count, total,
                       self.ema_decay, self.epsilon)
        self.step_count += 1
        if self.step_count % self.reset_interval == 0:
            self._reset_unused_codes(latent)

    @torch.no_grad()
    def _reset_unused_codes(self, latent, draws=None):
        """index_improve/models/vq.py:79-114: re-seed codes whose EMA usage share is below the
        threshold with random batch latents + N(0, 0.01^2) noise and zero their EMA statistics.
        `draws` = (sample_indices, permutation, noise) injects the three random draws (tests)."""
        usage = self._ema_cluster_size / (self._ema_cluster_size.sum() + self.epsilon)
        unused = torch.where(usage < self.reset_threshold)[0]
        num_unused = int(unused.numel())
        if num_unused == 0 or len(latent) == 0:
            return 0
        num = min(num_unused, len(latent))
        dev = latent.device
        gen = self.reset_generator
        sample = draws[0] if draws else torch.randint(0, len(latent), (num,), device=dev, generator=gen)
        vectors = latent[sample]
        if num_unused > num:
            perm = draws[1] if draws else torch.randperm(num_unused, device=dev, generator=gen)
            target = unused[perm[:num]]
        else:
            target = unused
        noise = (draws[2] if draws else torch.randn(vectors.shape, device=dev, dtype=vectors.dtype, generator=gen)) * 0.01
        self.embedding.weight.data[target] = (vectors + noise).detach()
        self._ema_cluster_size[target] = 0
        self._ema_w[target] = 0
        return num

    def get_codebook_usage(self):
        """index_improve/models/vq.py:205-217."""
        if self.ema_decay is None:
            raise AttributeError("codebook usage needs the EMA statistics (construct with ema_decay)")
        with torch.no_grad():
            total = self._ema_cluster_size.sum() + self.epsilon
            usage = self._ema_cluster_size / total
            used = int((usage > self.reset_threshold).sum().item())
        return {"utilization": used / self.n_e, "used_codes": used, "total_codes": self.n_e}
