"""The quantiser's single autograd node, shared by VectorQuantizer (one level) and
ResidualVectorQuantizer (L levels): values from the HIP kernels, gradients in closed form.

  runs of hard (argmin) levels   -> lcrec_rq_assign   (all levels of the run fused, residual in registers)
  Sinkhorn levels (use_sk, eps>0)-> lcrec_sinkhorn_assign + lcrec_rq_apply_level
  per-code statistics            -> lcrec_code_stats   (codebook gradient and EMA input)

Backward is what autograd derives from the reference's vq.py:87-95 / rq.py:45-48:
d residual_{l+1} / d residual_l = 0 and d x_q / d z = I, so
  dL/dz   = g_xq + g_loss * beta * 2/(L*n*e) * (z - C_0[idx_0])
  dL/dC_l = g_loss * 2/(L*n*e) * (count_l * C_l - sum_l).
"""
import torch

from . import dist as ldist
from . import ops


def quantize_values(zc, cbs, beta, plan, want_stats, training, want_code_grads=True, want_loss=True):
    """Values of the L-level quantiser on detached, contiguous inputs -- shared by the autograd node below and by the
    graph-captured training step (engine.py), which calls it without autograd.
    Returns dict(xq, rq_loss, idx, stats, resid_in, code_grads, commit); the last two (closed-form gradient factors, see
    the module docstring) only with want_stats."""
    n, e = zc.shape
    L = len(cbs)
    dev = zc.device
    idx = torch.empty((n, L), dtype=torch.int64, device=dev)
    sse = torch.empty(L, dtype=torch.float64, device=dev)      # every level's kernel writes its slot
    resid_in = [None] * L
    r, xq, l = zc, None, 0
    while l < L:
        if plan[l] is not None:
            eps, iters = plan[l]
            world = ldist.current()
            if world.enabled and training:
                # data parallel: the Sinkhorn problem is the GLOBAL batch (dist.py)
                r_all, (lo, hi) = world.gather_rows_with_slice(r)
                idx[:, l] = ops.sinkhorn_assign(r_all, cbs[l], eps, iters)[lo:hi]
            else:
                ops.sinkhorn_assign(r, cbs[l], eps, iters, out=idx[:, l])
            resid_in[l] = r
            xq, r, _ = ops.rq_apply_level(r, cbs[l], idx[:, l], xq=xq, want_sse=True, sse_out=sse[l:l + 1])
            l += 1
            continue
        m = l
        while m < L and plan[m] is None:
            m += 1
        flat, ks = ops.flatten_codebooks(cbs[l:m])
        # (the run's index columns are written in place into the [n, L] matrix: no copy launch afterwards)
        _, xq, _, resid = ops.rq_assign(r, flat, ks, want_xq=True, want_sse=True, want_resid=True, xq_init=xq,
                                        sse_out=sse[l:m], idx_into=(idx, l))
        for t in range(l, m):
            resid_in[t] = resid[t - l]
        r = resid[m - l]
        l = m
    # vq.py:90-92: loss_l = mse + beta*mse (fp32), rq.py:53: mean over levels  (the engine does this in lcrec_step_losses)
    rq_loss = None
    if want_loss:
        mse = (sse / float(n * e)).to(torch.float32)
        level_loss = mse + beta * mse
        rq_loss = level_loss.mean()
    out = {"xq": xq, "rq_loss": rq_loss, "idx": idx, "stats": None, "resid_in": resid_in,
           "code_grads": None, "commit": None, "sse": sse}
    if want_stats:
        stats = [ops.code_stats(idx[:, t], resid_in[t], cbs[t].shape[0]) for t in range(L)]
        scale = 2.0 / (L * n * e)
        out["stats"] = stats
        out["scale"] = scale
        if want_code_grads:
            out["code_grads"] = [scale * (cnt.unsqueeze(1) * cbs[t] - tot) for t, (cnt, tot) in enumerate(stats)]
        out["commit"] = (beta * scale) * (zc - cbs[0].index_select(0, idx[:, 0]))
    return out


class _Quantize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, beta, plan, want_stats, side, *codebooks):
        """plan[l] = (epsilon, iters) for a Sinkhorn level, None for an argmin level.
        side: dict that receives the per-level statistics and level inputs (EMA needs them)."""
        zc = z.detach().contiguous()
        cbs = [c.detach().contiguous() for c in codebooks]
        v = quantize_values(zc, cbs, beta, plan, want_stats, bool(side.get("training")))
        ctx.code_grads = v["code_grads"]
        ctx.commit = v["commit"]
        ctx.has_grads = want_stats
        ctx.mark_non_differentiable(v["idx"])
        side["stats"] = v["stats"]
        side["resid_in"] = v["resid_in"]
        return v["xq"], v["rq_loss"], v["idx"]

    @staticmethod
    def backward(ctx, g_xq, g_loss, _g_idx):
        if not ctx.has_grads:
            raise RuntimeError("quantiser ran without statistics (no-grad forward); nothing to back-propagate")
        gz = None
        if ctx.needs_input_grad[0]:
            gz = ctx.commit * g_loss
            if g_xq is not None:
                gz = gz + g_xq
        gcs = [g * g_loss if need else None for g, need in zip(ctx.code_grads, ctx.needs_input_grad[5:])]
        return (gz, None, None, None, None, *gcs)


def level_plan(layers, use_sk):
    """plan[l] = (epsilon, iters) for a level assigned by Sinkhorn (vq.py:76-83), None for an argmin level (:75)."""
    return [((q.sk_epsilon, q.sk_iters) if (use_sk and q.sk_epsilon > 0) else None) for q in layers]


def quantize(z, layers, beta, use_sk, training):
    """Shared body of VectorQuantizer.forward / ResidualVectorQuantizer.forward on [n, e] latents.

    Returns (x_q, mean level loss, idx [n, L], side) where side["stats"] is the per-level (count, sum)
    list (None when neither training nor differentiating) and side["resid_in"] the level inputs."""
    plan = level_plan(layers, use_sk)
    cbs = [q.embedding.weight for q in layers]
    want = torch.is_grad_enabled() and (z.requires_grad or any(c.requires_grad for c in cbs))
    want_stats = want or training
    side = {"training": training}
    x_q, loss, idx = _Quantize.apply(z, float(beta), plan, want_stats, side, *cbs)
    return x_q, loss, idx, side
