"""Judging an fp32 evaluation of the run.sh-width training step against fixture F11 (oracle/make_golden.py
fixture_run_sh_step): the imported reference's gradients in fp64 are the yardstick, the reference's own fp32 run says how
far from it fp32 arithmetic lands, tensor by tensor.  Shared by the CPU test (the oracle's torch restatement) and the GPU
tests (engine, autograd path)."""
import numpy as np

import golden_inputs as gi

SLACK = 4.0          # an fp32 path may sit this many times further from fp64 than the reference's own fp32 run does ...
FLOOR = 2e-6         # ... where that distance is at least a few fp32 ulps of accumulated rounding
NOISE = 16.0         # tensors whose true gradient is zero (a Linear bias in front of BatchNorm): rounding noise, bounded by
                     # this multiple of the reference's own noise


def tensors(g):
    return [f[len("f64__sample__"):] for f in g.files if f.startswith("f64__sample__")]


def report(g, grads, f64=None):
    """grads: parameter name -> full gradient array of the path under test.  Returns rows
    (name, kind, path_err, ref_err, bound) and the list of violations.  `f64`: full fp64 gradients to judge against
    instead of the fixture's (f64_gradients_for below, when the path assigned a near-tied row to another code); the
    reference's fp32-vs-fp64 distance stays the fixture's -- it measures the arithmetic, not the assignment."""
    rows, bad = [], []
    for k in tensors(g):
        s64 = (g["f64__sample__" + k] if f64 is None else gi.strided_sample(np.asarray(f64[k]))).astype(np.float64)
        s32 = g["f32__sample__" + k].astype(np.float64)
        sp = gi.strided_sample(np.asarray(grads[k])).astype(np.float64)
        assert sp.shape == s64.shape, (k, sp.shape, s64.shape)
        n64, n32 = np.linalg.norm(s64), np.linalg.norm(s32)
        if n64 < 1e-6 * n32:                      # the true gradient is zero; the fp32 values are pure rounding noise
            path, ref, bound, kind = np.linalg.norm(sp), n32, NOISE * n32, "noise"
        else:
            path, ref = np.linalg.norm(sp - s64) / n64, np.linalg.norm(s32 - s64) / n64
            bound, kind = SLACK * max(ref, FLOOR), "rel"
        rows.append((k, kind, path, ref, bound))
        if not path <= bound:
            bad.append((k, kind, path, ref, bound))
    return rows, bad


def f64_gradients_for(g, idx):
    """The run.sh step in fp64 with the codes FORCED to `idx` [1024, 4] (oracle/torch_ref.py, checked against the fixture's
    fp64 values with the fixture's own codes by tests/test_oracle_golden.py): ([loss, recon, rq_loss, grad norm], name ->
    gradient).  A path whose latents differ in the last bit may send a near-tied row of the Sinkhorn level to another code --
    a different problem for that row, not different arithmetic; this evaluates that problem exactly."""
    import torch
    from oracle import torch_ref
    sd_np, x = gi.run_sh_train_case()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    for l in range(4):
        sd[f"rq.vq_layers.{l}.embedding.weight"] = torch.from_numpy(g["codebooks"][l].copy())
    leaf = {}
    for k, v in sd.items():
        v = v.double() if v.dtype.is_floating_point else v.clone()
        leaf[k] = v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v
    spec = torch_ref.Spec(768, [256] * 4, 32, gi.RUN_SH_LAYERS, bn=True, sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    xt = torch.from_numpy(x).double()
    out, rq_loss, got = torch_ref.forward(spec, leaf, xt, use_sk=True, training=True, force_idx=torch.from_numpy(np.asarray(idx)))
    loss, recon = torch_ref.compute_loss(spec, out, rq_loss, xt)
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in leaf.items() if v.requires_grad}
    norm = np.sqrt(sum(float((v ** 2).sum()) for v in grads.values()))
    return [loss.item(), recon.item(), rq_loss.item(), norm], grads


def table(rows):
    return "\n".join(f"{k:40s} {kind:5s} path {p:.3e}  reference-fp32 {r:.3e}  bound {b:.3e}" for k, kind, p, r, b in rows)
