"""MLP stack, k-means seeding and the Sinkhorn entry point -- host-side mirror of the reference's
index/models/layers.py (MLPLayers :7-43, activation_layer :45-67, kmeans :69-82,
sinkhorn_algorithm :85-108) with the arithmetic on the MI355X.

The module tree (nn.Sequential of Dropout / Linear / BatchNorm1d / ReLU at the reference's
indices) is kept because checkpoints are keyed by it (`encoder.mlp_layers.1.weight`, ...); what
runs is different:
  * eval / no-grad : one fused fp32-MFMA kernel per layer (bias + folded BatchNorm + ReLU in the
    GEMM epilogue), see csrc/gemm_f32.hip;
  * training       : the same kernel family for the forward GEMM and for both backward GEMMs
    (dX = dY W, dW = dY^T X; lcrec_linear_backward reads every operand as stored, no transposed
    copies); batch-statistics BatchNorm (+ReLU) and its backward are one launch each
    (lcrec_bn_relu_forward / _backward), global-batch statistics under data parallelism.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.init import xavier_normal_

from . import dist as ldist
from . import ops


class _BatchNormAct(torch.autograd.Function):
    """y = [relu](BatchNorm1d(t)) in training mode (layers.py:25-30 of the reference) on the library's own kernels:
    batch statistics, affine, ReLU and the running-statistics update in one launch, the backward (d t, d gamma, d beta)
    in one more.  Under item-sharded data parallelism the batch is the union of all ranks' rows -- what
    torch.nn.SyncBatchNorm computes -- with ONE all-reduce per direction: per-rank (n, mean, M2) rows merged in rank
    order going forward, (sum g, sum g*xhat) going back; it needs nothing but all_reduce, so it also runs over gloo."""

    @staticmethod
    def forward(ctx, t, gamma, beta, bn, relu):
        world = ldist.current()
        n_local = t.shape[0]
        if world.enabled:
            mean, rstd, n_total = sharded_bn_statistics(world, t, bn)
            if n_total < 2:                    # the GLOBAL row count, known to every rank: all of them raise here together
                raise ValueError(f"Expected more than 1 value per channel when training, got a global batch of {n_total} row(s)")
            y = ops.bn_relu_apply(t, gamma.detach(), beta.detach(), mean, rstd, relu)
        else:
            n_total = n_local
            y, mean, rstd = ops.bn_relu_forward(t, gamma.detach(), beta.detach(), bn.eps, bn.momentum, bn.running_mean,
                                                bn.running_var, relu=relu)
        bn.num_batches_tracked += 1
        ctx.relu, ctx.n_total, ctx.synced = relu, n_total, world.enabled
        ctx.save_for_backward(t, y, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        t, y, gamma, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.synced:
            world = ldist.current()
            local = ops.bn_backward_reduce(gy, t, y, mean, rstd, ctx.relu)
            total = local.clone()
            world.all_reduce_(total)
            dt, _ = ops.bn_backward_apply(gy, t, y, gamma.detach(), mean, rstd, total, ctx.n_total, ctx.relu)
            return dt, local[1], local[0], None, None       # parameter gradients stay local: the gradient all-reduce sums them
        dt, dgamma, dbeta, _ = ops.bn_relu_backward(gy, t, y, gamma.detach(), mean, rstd, ctx.relu)
        return dt, dgamma, dbeta, None, None


def sharded_bn_statistics(world, t, bn):
    """(mean, rstd, rows) of the GLOBAL batch for this rank's rows t, and the running-statistics update: local (mean, M2)
    written into an exchange row, ONE collective, one merge launch (lcrec_bn_stats / lcrec_bn_merge_stats)."""
    n_local, F = t.shape
    row = torch.empty(2 * F + 1, dtype=torch.float32, device=t.device)
    row[:1].fill_(float(n_local))
    ops.bn_stats(t, row_out=row)
    rows = world.exchange_rows(row)
    known = getattr(world, "batch_rows", None)
    n_total = known[1] if known is not None and known[0] == n_local else int(rows[:, 0].sum().item())
    with torch.no_grad():
        mean, rstd = ops.bn_merge_stats(rows, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
    return mean, rstd, n_total


def _own_batchnorm(bn, x):
    """Whether the library's BatchNorm kernels apply to this module call (else torch's own module runs).
    Under data parallelism the batch is the GLOBAL one: a rank may hold a single row of it (the last global batch of an
    epoch with W .. 2W-1 rows on W ranks), and torch's own module would normalise with per-rank statistics -- or raise on
    that rank alone while its peers wait in the statistics exchange.  There the library kernels are the only path; a
    module they do not cover raises on every rank alike."""
    covered = (type(bn) is nn.BatchNorm1d and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
               and x.is_cuda and x.dim() == 2)
    if ldist.current().enabled and bn.training:
        if not covered:
            raise ops._lib.LcrecError("data-parallel training: this BatchNorm configuration is not covered by the library's "
                                      "global-batch kernels (needs BatchNorm1d, affine, running statistics, momentum, 2-d device input)")
        return True
    return covered and x.shape[0] >= 2 and torch.is_grad_enabled()


class _LinearAct(torch.autograd.Function):
    """y = [relu](x W^T + b) with the HIP GEMM in all three products."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        y = ops.linear_forward(x, weight, bias, relu=relu)
        ctx.relu = relu
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.relu:
            gy = torch.ops.aten.threshold_backward(gy, y, 0.0)
        gx = gw = gb = None
        out_dim, in_dim = weight.shape
        if ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        # both products read gy, x and W as they are stored (k-major operand staging in the kernel), for any
        # batch size.  A narrow layer whose width is not a multiple of the K slice (e_dim 16, say) is padded
        # with zero columns / rows first: fma(0, w, acc) adds nothing, and the copies are a few KB.
        pad = (-out_dim) % 32
        if pad:
            gy = F.pad(gy, (0, pad))
            weight = F.pad(weight, (0, 0, 0, pad))
        gx, gw = ops.linear_backward(gy, x, weight, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if pad and gw is not None:
            gw = gw[:out_dim]
        return gx, gw, gb, None


class _MlpChain(torch.autograd.Function):
    """A run of [Linear, ReLU?] groups as ONE autograd node: the same library calls in the same order as a chain of
    _LinearAct nodes (bit-identical values and gradients), without a Python `apply` and an autograd-engine hop per
    layer -- at batch sizes up to ~1 k a training step is bound by exactly that host work."""

    @staticmethod
    def forward(ctx, x, relus, *params):
        saved = []
        h = x
        for l, relu in enumerate(relus):
            w, b = params[2 * l], params[2 * l + 1]
            y = ops.linear_forward(h, w, b, relu=relu)
            saved += [h, w, y if relu else None]
            h = y
        ctx.relus = relus
        ctx.save_for_backward(*saved)
        return h

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        L = len(ctx.relus)
        grads = [None] * (2 * L)
        g = gy.contiguous()
        for l in range(L - 1, -1, -1):
            x, weight, y = saved[3 * l], saved[3 * l + 1], saved[3 * l + 2]
            if ctx.relus[l]:
                g = torch.ops.aten.threshold_backward(g, y, 0.0)
            if ctx.needs_input_grad[2 + 2 * l + 1]:
                grads[2 * l + 1] = g.sum(0)
            out_dim = weight.shape[0]
            pad = (-out_dim) % 32                      # see _LinearAct.backward
            if pad:
                g = F.pad(g, (0, pad))
                weight = F.pad(weight, (0, 0, 0, pad))
            need_gx = l > 0 or ctx.needs_input_grad[0]
            gx, gw = ops.linear_backward(g, x, weight, need_gx, ctx.needs_input_grad[2 + 2 * l])
            if gw is not None:
                grads[2 * l] = gw[:out_dim] if pad else gw
            g = gx
        return (g if ctx.needs_input_grad[0] else None, None, *grads)


def fold_batchnorm(bn):
    """Eval-mode BatchNorm1d as an affine y = t*scale + shift (fp32, on the module's device)."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    return scale.contiguous(), shift.contiguous()


def activation_layer(activation_name="relu", emb_dim=None):
    """layers.py:45-67: name -> activation module (None for "none")."""
    if activation_name is None:
        return None
    if isinstance(activation_name, str):
        table = {"sigmoid": nn.Sigmoid, "tanh": nn.Tanh, "relu": nn.ReLU, "leakyrelu": nn.LeakyReLU}
        key = activation_name.lower()
        if key == "none":
            return None
        if key in table:
            return table[key]()
        return None
    if issubclass(activation_name, nn.Module):
        return activation_name()
    raise NotImplementedError("activation function {} is not implemented".format(activation_name))


class MLPLayers(nn.Module):
    """[Dropout, Linear, (BatchNorm1d), activation] per layer; the last layer has neither
    BatchNorm nor activation; Xavier-normal weights, zero biases (layers.py:7-40)."""

    def __init__(self, layers, dropout=0.0, activation="relu", bn=False):
        super().__init__()
        self.layers = layers
        self.dropout = dropout
        self.activation = activation
        self.use_bn = bn
        mods, self._groups = [], []
        last = len(layers) - 2
        for i, (fan_in, fan_out) in enumerate(zip(layers[:-1], layers[1:])):
            group = {"drop": len(mods)}
            mods.append(nn.Dropout(p=dropout))
            group["linear"] = len(mods)
            mods.append(nn.Linear(fan_in, fan_out))
            if bn and i != last:
                group["bn"] = len(mods)
                mods.append(nn.BatchNorm1d(num_features=fan_out))
            act = activation_layer(activation, fan_out)
            if act is not None and i != last:
                group["act"] = len(mods)
                mods.append(act)
            self._groups.append(group)
        self.mlp_layers = nn.Sequential(*mods)
        self.apply(self.init_weights)

    def init_weights(self, module):
        if isinstance(module, nn.Linear):
            xavier_normal_(module.weight.data)
            if module.bias is not None:
                module.bias.data.fill_(0.0)

    # ---- what the fused inference path needs: per-layer (W, b, bn_scale, bn_shift)
    def folded(self):
        Ws, bs, scs, shs = [], [], [], []
        for g in self._groups:
            lin = self.mlp_layers[g["linear"]]
            Ws.append(lin.weight.detach())
            bs.append(lin.bias.detach())
            if "bn" in g:
                sc, sh = fold_batchnorm(self.mlp_layers[g["bn"]])
            else:
                sc = sh = None
            scs.append(sc)
            shs.append(sh)
        return Ws, bs, scs, shs

    def fusable(self):
        """True when every activation is ReLU (the only one the GEMM epilogue implements)."""
        return all(("act" not in g) or isinstance(self.mlp_layers[g["act"]], nn.ReLU) for g in self._groups)

    def _chainable(self):
        """Differentiating through plain [Linear, ReLU?] groups (no BatchNorm, no active dropout, bias present):
        the whole stack is one autograd node."""
        if not torch.is_grad_enabled() or (self.training and self.dropout > 0):
            return False
        mods = self.mlp_layers
        return all("bn" not in g and mods[g["linear"]].bias is not None
                   and ("act" not in g or isinstance(mods[g["act"]], nn.ReLU)) for g in self._groups)

    def forward(self, input_feature):
        x = input_feature
        if x.dim() != 2:
            x = x.reshape(-1, x.shape[-1])
        if self._chainable():
            mods = self.mlp_layers
            relus = tuple("act" in g for g in self._groups)
            params = []
            for g in self._groups:
                params += [mods[g["linear"]].weight, mods[g["linear"]].bias]
            x = _MlpChain.apply(x, relus, *params)
            return x.reshape(*input_feature.shape[:-1], x.shape[-1])
        for g in self._groups:
            mods = self.mlp_layers
            if self.training and self.dropout > 0:
                x = mods[g["drop"]](x)
            lin = mods[g["linear"]]
            relu_mod = mods[g["act"]] if "act" in g else None
            is_relu = isinstance(relu_mod, nn.ReLU)
            if "bn" in g:
                bn = mods[g["bn"]]
                if self.training:
                    x = _LinearAct.apply(x, lin.weight, lin.bias, False)
                    if _own_batchnorm(bn, x):
                        x = _BatchNormAct.apply(x, bn.weight, bn.bias, bn, is_relu)     # batch statistics (+ ReLU), own kernels
                        if relu_mod is not None and not is_relu:
                            x = relu_mod(x)
                        continue
                    x = bn(x)                      # batch statistics + running-stat update (torch)
                    if relu_mod is not None:
                        x = relu_mod(x)
                    continue
                if not torch.is_grad_enabled() and (relu_mod is None or is_relu):
                    sc, sh = fold_batchnorm(bn)
                    x = ops.linear_forward(x, lin.weight.detach(), lin.bias.detach(), sc, sh, relu=is_relu)
                    continue
                x = _LinearAct.apply(x, lin.weight, lin.bias, False)
                x = bn(x)
                if relu_mod is not None:
                    x = relu_mod(x)
                continue
            fuse = relu_mod is None or is_relu
            x = _LinearAct.apply(x, lin.weight, lin.bias, bool(is_relu and fuse))
            if relu_mod is not None and not is_relu:
                x = relu_mod(x)
        return x.reshape(*input_feature.shape[:-1], x.shape[-1])


KMEANS_IMPL = "sklearn"          # "sklearn" (the reference's host call) or "device"; main.py --kmeans_impl


def kmeans(samples, num_clusters, num_iters=10, generator=None):
    """layers.py:69-82: sklearn KMeans on the host (k-means++ from numpy's global RNG), centres
    returned on the samples' device.  Runs once per level per training run; the sklearn call is the
    reference's own behaviour and its result is not bit-pinned.
    With KMEANS_IMPL == "device" the clustering stays in HBM (kmeans_device; `generator` seeds its k-means++ draws)."""
    if KMEANS_IMPL == "device":
        return kmeans_device(samples, num_clusters, num_iters, generator=generator)
    from sklearn.cluster import KMeans
    x = samples.detach().cpu().numpy()
    cluster = KMeans(n_clusters=num_clusters, max_iter=num_iters).fit(x)
    return torch.from_numpy(cluster.cluster_centers_).to(samples.device)


@torch.no_grad()
def kmeans_pp_seed(x, num_clusters, generator=None, trials=None):
    """Greedy k-means++ seeding on the device, as sklearn's _kmeans_plusplus does it: the first centre uniformly, every
    next one the best of `trials` = 2 + floor(ln K) candidates drawn with probability proportional to the squared
    distance to the nearest centre chosen so far -- best = smallest resulting potential.  All draws come from
    `generator` (a device generator; default: one seeded from torch's global CPU generator, so torch.manual_seed governs
    it).  No host synchronisation: K rounds of queued launches."""
    import math
    n = x.shape[0]
    if generator is None:
        generator = torch.Generator(device=x.device)
        generator.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    trials = int(trials) if trials else 2 + int(math.log(max(num_clusters, 1)))
    first = torch.randint(n, (1,), device=x.device, generator=generator)
    picks = [first]
    nearest = ((x - x[first]) ** 2).sum(1)
    for _ in range(1, num_clusters):
        w = nearest.clamp_min(0)
        # fewer distinct points than clusters: fall back to uniform -- decided on the device, no host round trip per centre
        w = torch.where(w.sum() > 0, w, torch.ones_like(w))
        cand = torch.multinomial(w, trials, replacement=True, generator=generator)            # [T]
        d = torch.cdist(x, x[cand]) ** 2                                                       # [n, T]
        pot = torch.minimum(nearest[:, None], d).sum(0)                                        # potential per candidate
        best = pot.argmin()
        picks.append(cand[best].reshape(1))
        nearest = torch.minimum(nearest, d[:, best])
    return x[torch.cat(picks)].clone()


@torch.no_grad()
def kmeans_device(samples, num_clusters, num_iters=10, tol=1e-4, generator=None, init=None, relocate_empty=True):
    """Device-resident replacement for the host sklearn call of layers.py:69-82 (SURVEY.md section 8f,
    rank 3): k-means++ seeding, then Lloyd iterations made of the library's own kernels --
    lcrec_rq_assign (one level) for the nearest centre and lcrec_code_stats for the per-cluster sums, both
    deterministic, so a run is reproducible bit for bit and equals oracle/cpu_oracle.kmeans_lloyd from the
    same seeds.  Stops after num_iters iterations or when the summed squared centre shift falls below
    tol * mean feature variance (sklearn's rule).  An empty cluster's centre moves to a point far from its own centre, as
    sklearn does (relocate_empty=False keeps it, which is what oracle/cpu_oracle.kmeans_lloyd restates).  Not
    bit-comparable with sklearn -- nor is sklearn with itself across versions; the reference's init is "parity unpinned" (SURVEY.md section 8c)."""
    x = samples.detach().to(torch.float32).contiguous()
    if not x.is_cuda:
        raise ops._lib.LcrecError("kmeans_device expects a device tensor (lcrec_amd has no CPU path)")
    centres = (kmeans_pp_seed(x, num_clusters, generator) if init is None
               else init.detach().to(device=x.device, dtype=torch.float32).clone())
    limit = tol * x.var(dim=0, unbiased=False).mean()
    for _ in range(int(num_iters)):
        res = ops.rq_assign(x, centres.reshape(-1), [num_clusters], want_xq=True)
        nearest, x_q = res[0][:, 0], res[1]                 # x_q = r + (c - r): the assigned centre up to rounding
        count, total = ops.code_stats(nearest, x, num_clusters)
        moved = torch.where(count[:, None] > 0, total / count[:, None].clamp_min(1.0), centres)
        empty = count == 0
        n_empty = int(empty.sum())                       # (the loop reads one scalar per iteration anyway: `shift` below)
        if n_empty and relocate_empty:
            # sklearn's rule (_relocate_empty_clusters_dense): an empty cluster's centre moves to one of the points farthest
            # from the centre they are assigned to
            far = ((x - x_q) ** 2).sum(1).topk(min(n_empty, x.shape[0])).indices
            moved[torch.nonzero(empty).flatten()[: far.numel()]] = x[far]
        shift = ((moved - centres) ** 2).sum()
        centres = moved
        if bool(shift <= limit):
            break
    return centres


@torch.no_grad()
def sinkhorn_algorithm(distances, epsilon, sinkhorn_iterations):
    """layers.py:85-108 on an already centred [B, K] matrix: returns Q (fp64, rows sum to 1).

    Kept for API parity (the quantiser itself calls the fused lcrec_sinkhorn_assign, which never
    materialises Q on the host side).  Runs as device tensor ops in the reference's order."""
    if not distances.is_cuda:
        raise ops._lib.LcrecError("sinkhorn_algorithm expects a device tensor (lcrec_amd has no CPU path)")
    Q = torch.exp(-distances / epsilon)
    B, K = Q.shape
    Q /= Q.sum(-1, keepdim=True).sum(-2, keepdim=True)
    for _ in range(sinkhorn_iterations):
        Q /= torch.sum(Q, dim=1, keepdim=True)
        Q /= B
        Q /= torch.sum(Q, dim=0, keepdim=True)
        Q /= K
    Q *= B
    return Q
