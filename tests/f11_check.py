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


def report(g, grads):
    """grads: parameter name -> full gradient array of the path under test.  Returns rows
    (name, kind, path_err, ref_err, bound) and the list of violations."""
    rows, bad = [], []
    for k in tensors(g):
        s64 = g["f64__sample__" + k].astype(np.float64)
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


def table(rows):
    return "\n".join(f"{k:40s} {kind:5s} path {p:.3e}  reference-fp32 {r:.3e}  bound {b:.3e}" for k, kind, p, r, b in rows)
