"""Tensor-level entry points over the C-ABI (include/lcrec.h).

torch is plumbing here: it owns device memory and the stream; every function
hands raw device pointers to liblcrec_hip.so and enqueues on torch's current
stream.  Inputs must already live on a HIP device; nothing here computes on
the CPU.
"""
import ctypes

import torch

from . import _lib

_c_int_p = ctypes.POINTER(ctypes.c_int)
_workspaces = {}

# Default threshold of the near-tie audit (lcrec_rq_assign's tie_tau): a level is flagged when its top-2 distance gap is
# <= NEARTIE_TAU * (xx + cc[idx]).  Measured, not derived (oracle/neartie_audit.py, tests/golden/f9_neartie_*.npz): on
# 1 M x 768-d items 33 tuples differ between the reference's CPU ops (batch 64 or 4096; the two batch sizes differ from
# EACH OTHER on 9) and this library's canonical order, the worst of them at a relative gap of 2.5e-6; 2^-17 covers that
# with a power of two of slack and flags 995 of the 1 M items (0.1 %).  On the Games shape (16 859 x 4096-d) 0 differ.
NEARTIE_TAU = 2.0 ** -17


# Host cost matters: a training step makes ~50 calls into the library, and torch.cuda.current_stream() /
# torch.cuda.device() cost 5-10 us each (Stream objects, index normalisation) -- as much as the launch itself.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream_int(index=None):
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device() if index is None else index)
    return torch.cuda.current_stream(index).cuda_stream


def _stream_ptr():
    return ctypes.c_void_p(_stream_int())


class _on:
    """`with _on(device)`: torch.cuda.device(device), skipped when that device is already current."""
    __slots__ = ("guard",)

    def __init__(self, device):
        same = _cur_device is not None and device.index is not None and device.index == _cur_device()
        self.guard = None if same else torch.cuda.device(device)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.LcrecError(f"{name} must be a tensor on a HIP device (lcrec_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise _lib.LcrecError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _ints(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def _workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream); reuse is safe because calls are stream-ordered."""
    key = (device.index, _stream_int(device.index))
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces():
    _workspaces.clear()
    _tickets.clear()


# `ticket` arguments of include/lcrec.h: one zeroed 4-byte word per (device, stream) -- every call leaves it zero, and calls on
# one stream are ordered, so they can all share it.  LCREC_TICKETS=0 (diagnostic): the two-launch forms.
_tickets = {}
USE_TICKETS = __import__("os").environ.get("LCREC_TICKETS", "1") != "0"


def _ticket(device, force=False):
    """int32 [96] zeros: word 0 is the ticket of the reduction tails; words 16 .. 79 are the 64 column-strip tickets of
    lcrec_linear_bn_forward (force=True: that call has no ticket-less form)."""
    if not USE_TICKETS and not force:
        return None
    key = (device.index, _stream_int(device.index))
    t = _tickets.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.LcrecError("the stream's ticket word must exist before a graph capture (run the step once eagerly first)")
        t = _tickets[key] = torch.zeros(96, dtype=torch.int32, device=device)
    return t


# One lcrec_context per device (include/lcrec.h): the library's helper streams, their events and the pinned upload ring.
# Created on first use by the two calls that can overlap launches, destroyed at interpreter exit / release_contexts().
_contexts = {}
_pipelines = 1


def _context(device):
    ctx = _contexts.get(device.index)
    if ctx is None:
        lib = _lib.load()
        h = ctypes.c_void_p()
        with _on(device):
            _lib.check(lib.lcrec_context_create(ctypes.byref(h)), "lcrec_context_create")
        _lib.check(lib.lcrec_context_set_pipelines(h, _pipelines), "lcrec_context_set_pipelines")
        ctx = _contexts[device.index] = h
    return ctx


def set_pipelines(n):
    """Chunk pipelines of encode_assign (lcrec_context_set_pipelines): 1 (default) = everything on the current stream,
    2 = odd chunks on a helper stream (+1.5-1.9 % on C3, but see include/lcrec.h: intermittently much slower)."""
    global _pipelines
    n = int(n)
    if n not in (1, 2):
        raise _lib.LcrecError("pipelines must be 1 or 2")
    _pipelines = n
    for h in _contexts.values():
        _lib.check(_lib.load().lcrec_context_set_pipelines(h, n), "lcrec_context_set_pipelines")


def release_contexts():
    """Drain and destroy the library's helper streams (lcrec_context_destroy)."""
    while _contexts:
        _, h = _contexts.popitem()
        try:
            _lib.load().lcrec_context_destroy(h)
        except Exception:          # interpreter shutdown: the runtime may already be gone
            pass


def _forget_contexts_at_exit():
    """Interpreter exit: drop the handles WITHOUT calling into the library.  lcrec_context_destroy synchronises and destroys
    streams; at exit the HIP runtime (and a profiler's preloaded tool library) is tearing down underneath it, and a profiled run
    was seen not to return from its exit handlers (tools/prof_train.sh).  Process teardown reclaims streams, events and pinned
    memory anyway; release_contexts() remains the explicit, orderly way while the process lives."""
    _contexts.clear()


import atexit  # noqa: E402
atexit.register(_forget_contexts_at_exit)


def _audit_buffers(audit, tie_tau, n, L, dev):
    """(margin tensor | None, neartie tensor | None, tau) for the near-tie audit outputs of lcrec_rq_assign."""
    if audit is None:
        return None, None, 0.0
    margin = torch.empty((n, L), dtype=torch.float32, device=dev)
    neartie = torch.zeros(n, dtype=torch.int32, device=dev) if tie_tau is not None else None
    audit["margin"] = margin
    if neartie is not None:
        audit["neartie"] = neartie           # bit l: level l's top-2 gap <= tie_tau * (xx + cc[idx])
        audit["tie_tau"] = float(tie_tau)
    return margin, neartie, float(tie_tau or 0.0)


def linear_forward(x, weight, bias=None, bn_scale=None, bn_shift=None, relu=False):
    """y = [relu]([bn](x @ weight.T + bias)) -- one group of MLPLayers.forward (layers.py:18-30,42)."""
    lib = _lib.load()
    x = _dev(x, "x")
    weight = _dev(weight, "weight")
    bias = None if bias is None else _dev(bias, "bias")
    bn_scale = None if bn_scale is None else _dev(bn_scale, "bn_scale")
    bn_shift = None if bn_shift is None else _dev(bn_shift, "bn_shift")
    n, k = x.shape
    out_dim = weight.shape[0]
    if weight.shape[1] != k:
        raise _lib.LcrecError(f"weight is {tuple(weight.shape)}, x is {tuple(x.shape)}")
    y = torch.empty((n, out_dim), dtype=torch.float32, device=x.device)
    with _on(x.device):
        rc = lib.lcrec_linear_forward(_ptr(x), n, k, _ptr(weight), _ptr(bias), _ptr(bn_scale), _ptr(bn_shift),
                                      int(bool(relu)), out_dim, _ptr(y), _stream_ptr())
    _lib.check(rc, "lcrec_linear_forward")
    return y


def linear_bn_supported(n, in_dim, out_dim):
    """Whether lcrec_linear_bn_forward takes this shape (a batch-sized launch; include/lcrec.h)."""
    return (n >= 2 and in_dim % 32 == 0 and out_dim % 4 == 0 and out_dim <= 4096
            and ((n + 127) // 128) * ((out_dim + 127) // 128) < 512 and in_dim * 4 * (n + 128) < (1 << 31))


def linear_bn_forward(x, weight, bias, in_fold=None, in_relu=True, bn=None):
    """One Linear of a training step with the BatchNorms around it folded in (lcrec_linear_bn_forward):
    t = max(x * in_fold[0] + in_fold[1], 0 if in_relu) @ weight.T + bias, and -- with bn = (gamma, beta, eps, momentum,
    running_mean, running_var) -- the batch statistics of t.  Returns (t, None) or (t, (mean, rstd, scale, shift)):
    scale / shift are this BatchNorm's folded affine, the next layer's in_fold."""
    lib = _lib.load()
    x, weight, bias = _dev(x, "x"), _dev(weight, "weight"), _dev(bias, "bias")
    n, k = x.shape
    out_dim = weight.shape[0]
    dev = x.device
    t = torch.empty((n, out_dim), dtype=torch.float32, device=dev)
    sc, sh = (None, None) if in_fold is None else (_vec(in_fold[0], "in_scale", k), _vec(in_fold[1], "in_shift", k))
    stats = None
    gamma = beta = rm = rv = None
    eps = momentum = 0.0
    if bn is not None:
        gamma, beta, eps, momentum, rm, rv = bn
        stats = torch.empty((4, out_dim), dtype=torch.float32, device=dev)     # mean, rstd, scale, shift
        gamma, beta = _vec(gamma, "gamma", out_dim), _vec(beta, "beta", out_dim)
    with _on(dev):
        ws = _workspace(lib.lcrec_linear_bn_forward_workspace(n, out_dim), dev) if bn is not None else None
        tk = _ticket(dev, force=True) if bn is not None else None
        rc = lib.lcrec_linear_bn_forward(_ptr(x), n, k, _ptr(sc), _ptr(sh), int(bool(in_relu)), _ptr(weight), _ptr(bias), out_dim,
                                         _ptr(t), int(bn is not None), _ptr(gamma), _ptr(beta), float(eps), float(momentum or 0.0),
                                         _ptr(rm), _ptr(rv), _ptr(None if stats is None else stats[0]),
                                         _ptr(None if stats is None else stats[1]), _ptr(None if stats is None else stats[2]),
                                         _ptr(None if stats is None else stats[3]), _ptr(ws), 0 if ws is None else ws.numel(),
                                         ctypes.c_void_p(tk.data_ptr() + 64) if tk is not None else None, _stream_ptr())
    _lib.check(rc, "lcrec_linear_bn_forward")
    return t, (None if stats is None else (stats[0], stats[1], stats[2], stats[3]))


def linear_backward(gy, x, weight, need_gx=True, need_gw=True, gw_out=None):
    """(gx, gw) of y = x W^T for a given gy = dL/dy (autograd's LinearBackward; include/lcrec.h,
    lcrec_linear_backward): gx = gy W, gw = gy^T x, every operand read as stored.  Raises
    LcrecError(LCREC_EUNSUPPORTED) for shapes the k-major kernels do not cover.
    gw_out: optional contiguous [out_dim, in_dim] tensor the weight gradient is written into (a view of a flat
    gradient buffer)."""
    lib = _lib.load()
    gy, x, weight = _dev(gy, "gy"), _dev(x, "x"), _dev(weight, "weight")
    n, out_dim = gy.shape
    in_dim = x.shape[1]
    if x.shape[0] != n or tuple(weight.shape) != (out_dim, in_dim):
        raise _lib.LcrecError(f"linear_backward: shapes gy {tuple(gy.shape)}, x {tuple(x.shape)}, W {tuple(weight.shape)}")
    gx = torch.empty((n, in_dim), dtype=torch.float32, device=gy.device) if need_gx else None
    gw = None
    if need_gw:
        gw = gw_out if gw_out is not None else torch.empty((out_dim, in_dim), dtype=torch.float32, device=gy.device)
        if tuple(gw.shape) != (out_dim, in_dim) or not gw.is_contiguous() or gw.dtype != torch.float32:
            raise _lib.LcrecError("gw_out must be a contiguous float32 [out_dim, in_dim] tensor")
    with _on(gy.device):
        nbytes = lib.lcrec_linear_backward_workspace(n, in_dim, out_dim) if need_gw else 0
        ws = _workspace(nbytes, gy.device)
        rc = lib.lcrec_linear_backward(_ptr(gy), _ptr(x), _ptr(weight), n, in_dim, out_dim, _ptr(gx), _ptr(gw), _ptr(ws),
                                       ws.numel(), _stream_ptr())
    _lib.check(rc, "lcrec_linear_backward")
    return gx, gw


def linear_backward_weights(problems, splits=0):
    """Weight gradients of several Linear layers in ONE launch (lcrec_linear_backward_weights): `problems` is a list of
    (gy [n, out], x [n, in], gw_out [out, in]) device tensors; every gw_out is written in place, bit-identical to what
    linear_backward(gy, x, W, need_gx=False) computes for that layer.  A fourth entry (scale, shift, relu) makes the layer
    input max(x * scale + shift, 0 if relu) -- x being a pre-BatchNorm tensor of lcrec_linear_bn_forward -- formed on the fly.
    splits > 0: that many K-runs over the batch for every problem instead of lcrec_linear_backward_splits' (1 = one chain)."""
    lib = _lib.load()
    count = len(problems)
    if count == 0:
        return
    arr = (_lib.DwProblem * count)()
    keep = []
    for i, prob in enumerate(problems):
        gy, x, gw = prob[:3]
        fold = prob[3] if len(prob) > 3 else None
        gy, x = _dev(gy, "gy"), _dev(x, "x")
        n, out_dim = gy.shape
        in_dim = x.shape[1]
        if x.shape[0] != n or tuple(gw.shape) != (out_dim, in_dim) or not gw.is_contiguous() or gw.dtype != torch.float32:
            raise _lib.LcrecError(f"linear_backward_weights: problem {i}: gy {tuple(gy.shape)}, x {tuple(x.shape)}, gw {tuple(gw.shape)}")
        if fold is None:
            arr[i] = _lib.DwProblem(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), n, in_dim, out_dim, None, None, 0, int(splits))
        else:
            fs, fh = _vec(fold[0], "x_scale", in_dim), _vec(fold[1], "x_shift", in_dim)
            arr[i] = _lib.DwProblem(gy.data_ptr(), x.data_ptr(), gw.data_ptr(), n, in_dim, out_dim, fs.data_ptr(), fh.data_ptr(),
                                    int(bool(fold[2])), int(splits))
            keep += [fs, fh]
        keep += [gy, x]
    dev = problems[0][0].device
    with _on(dev):
        nbytes = lib.lcrec_linear_backward_weights_workspace(ctypes.cast(arr, ctypes.c_void_p), count)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_linear_backward_weights(ctypes.cast(arr, ctypes.c_void_p), count, _ptr(ws), ws.numel(), _stream_ptr())
    _lib.check(rc, "lcrec_linear_backward_weights")


def linear_backward_splits(n, in_dim, out_dim):
    """Number of batch runs whose partial products make up gw (part of the arithmetic contract, include/lcrec.h)."""
    return int(_lib.load().lcrec_linear_backward_splits(n, in_dim, out_dim))


def flatten_codebooks(codebooks):
    """List of [K_l, e] tensors -> (flat fp32 tensor, [K_l]) in the layout lcrec_rq_assign expects."""
    ks = [int(c.shape[0]) for c in codebooks]
    cbs = [_dev(c, "codebook") for c in codebooks]
    # codebooks that already lie back to back in ONE storage (the training engine's flat parameter buffer) are passed as a
    # view of it: no concatenation launch per step.  Every tensor must belong to that storage and the view must fit in it --
    # separately allocated parameters can be neighbours in memory too, and a view past the end of a storage resizes it.
    base = cbs[0].untyped_storage()
    total = sum(c.numel() for c in cbs)
    if (len(cbs) > 1 and all(c.untyped_storage().data_ptr() == base.data_ptr() for c in cbs)
            and all(a.data_ptr() + a.numel() * 4 == b.data_ptr() for a, b in zip(cbs, cbs[1:]))
            and (cbs[0].storage_offset() + total) * 4 <= base.nbytes()):
        flat = cbs[0].new_empty(0).set_(base, cbs[0].storage_offset(), (total,), (1,))
        return flat, ks
    flat = cbs[0].reshape(-1) if len(cbs) == 1 else torch.cat([c.reshape(-1) for c in cbs])
    return flat, ks


def rq_assign(z, codebooks_flat, ks, want_xq=False, want_sse=False, want_resid=False, xq_init=None, audit=None,
              tie_tau=None, sse_out=None, idx_into=None):
    """ResidualVectorQuantizer.forward values with use_sk=False (rq.py:39-55).

    xq_init: optional [n, e] tensor that the x_q sum starts from (it is updated in place and returned).
    audit: optional dict that receives the near-tie audit outputs (include/lcrec.h): "margin" float32 [n, L] (top-2
    distance gap per level) and, with tie_tau, "neartie" int32 [n] (bit l set = level l is a near tie under tie_tau).
    idx_into: (matrix, col0) -- write the L index columns straight into columns col0 .. col0+L-1 of a wider contiguous int64
    [n, L_total] matrix (no copy afterwards); the returned idx is then that column block as a view.
    Returns (idx int64 [n, L], xq [n, e] | None, sse float64 [L] | None, resid [L+1, n, e] | None);
    resid[l] is the residual entering level l, resid[L] the residual left after the last level."""
    lib = _lib.load()
    z = _dev(z, "z")
    cb = _dev(codebooks_flat, "codebooks")
    n, e = z.shape
    L = len(ks)
    dev = z.device
    if idx_into is not None:
        matrix, col0 = idx_into
        if not (matrix.is_cuda and matrix.dtype == torch.int64 and matrix.dim() == 2 and matrix.is_contiguous()
                and matrix.shape[0] == n and 0 <= col0 and col0 + L <= matrix.shape[1]):
            raise _lib.LcrecError("idx_into must be (contiguous int64 [n, L_total] device matrix, first column)")
        idx, idx_ptr, idx_stride = matrix[:, col0:col0 + L], ctypes.c_void_p(matrix.data_ptr() + 8 * col0), int(matrix.shape[1])
    else:
        idx = torch.empty((n, L), dtype=torch.int64, device=dev)
        idx_ptr, idx_stride = _ptr(idx), L
    if xq_init is not None:
        xq = _dev(xq_init, "xq_init")
        if xq.data_ptr() != xq_init.data_ptr() or tuple(xq.shape) != (n, e):
            raise _lib.LcrecError("xq_init must be a contiguous [n, e] float32 device tensor")
    else:
        xq = torch.empty((n, e), dtype=torch.float32, device=dev) if want_xq else None
    sse = _sse_buffer(sse_out, L, dev, n) if want_sse else None
    resid = torch.empty((L + 1, n, e), dtype=torch.float32, device=dev) if want_resid else None
    margin, neartie, tau = _audit_buffers(audit, tie_tau, n, L, dev)
    karr = _ints(ks)
    with _on(dev):
        nbytes = lib.lcrec_rq_assign_workspace(n, e, karr, L)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_rq_assign(_ptr(z), n, e, _ptr(cb), karr, L, idx_ptr, idx_stride, _ptr(xq), int(xq_init is not None),
                                 _ptr(sse), _ptr(resid), _ptr(margin), _ptr(neartie), tau, _ptr(ws), ws.numel(),
                                 _ptr(_ticket(dev)) if want_sse else None, _stream_ptr())
    _lib.check(rc, "lcrec_rq_assign")
    return idx, xq, sse, resid


def encode_assign(x, weights, biases, codebooks_flat, ks, bn_scales=None, bn_shifts=None, want_latent=False,
                  want_xq=False, want_sse=False, audit=None, tie_tau=None):
    """RQVAE.get_indices(xs, use_sk=False) (rqvae.py:68-72): encoder MLP + L-level argmin assignment.
    audit / tie_tau: as rq_assign (near-tie audit of the quantiser).

    weights[l] is nn.Linear.weight of encoder layer l ([out_l, in_l]); bn_scales/bn_shifts are
    per-layer folded eval-mode BatchNorm affines or None.
    Returns (idx, latent | None, xq | None, sse | None)."""
    lib = _lib.load()
    x = _dev(x, "x")
    nl = len(weights)
    ws_ = [_dev(w, "weight") for w in weights]
    bs_ = [_dev(b, "bias") for b in biases]
    scs = [None] * nl if bn_scales is None else [None if s is None else _dev(s, "bn_scale") for s in bn_scales]
    shs = [None] * nl if bn_shifts is None else [None if s is None else _dev(s, "bn_shift") for s in bn_shifts]
    cb = _dev(codebooks_flat, "codebooks")
    dims = [int(ws_[0].shape[1])] + [int(w.shape[0]) for w in ws_]
    if x.shape[1] != dims[0]:
        raise _lib.LcrecError(f"x has {x.shape[1]} features, encoder expects {dims[0]}")
    for l in range(nl):
        if ws_[l].shape[1] != dims[l]:
            raise _lib.LcrecError(f"encoder layer {l}: weight {tuple(ws_[l].shape)} does not chain")
    n, e, L, dev = x.shape[0], dims[-1], len(ks), x.device
    idx = torch.empty((n, L), dtype=torch.int64, device=dev)
    latent = torch.empty((n, e), dtype=torch.float32, device=dev) if want_latent else None
    xq = torch.empty((n, e), dtype=torch.float32, device=dev) if want_xq else None
    sse = torch.zeros(L, dtype=torch.float64, device=dev) if want_sse else None
    PA = ctypes.c_void_p * nl
    wp = PA(*[t.data_ptr() for t in ws_])
    bp = PA(*[t.data_ptr() for t in bs_])
    scp = PA(*[0 if t is None else t.data_ptr() for t in scs])
    shp = PA(*[0 if t is None else t.data_ptr() for t in shs])
    darr, karr = _ints(dims), _ints(ks)
    margin, neartie, tau = _audit_buffers(audit, tie_tau, n, L, dev)
    with _on(dev):
        nbytes = lib.lcrec_encode_assign_workspace(n, darr, nl, karr, L)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_encode_assign(_ptr(x), n, darr, nl, wp, bp, scp, shp, _ptr(cb), karr, L, _ptr(idx),
                                     _ptr(latent), _ptr(xq), _ptr(sse), _ptr(margin), _ptr(neartie), tau, _ptr(ws),
                                     ws.numel(), _context(dev), _stream_ptr())
    _lib.check(rc, "lcrec_encode_assign")
    return idx, latent, xq, sse


def _idx_col(idx, n):
    """(tensor, element stride) for an int64 index column: a [n] vector or a column view of [n, L]."""
    if not idx.is_cuda or idx.dtype != torch.int64 or idx.dim() != 1 or idx.shape[0] != n:
        raise _lib.LcrecError("idx must be an int64 device vector of length n (a column view is fine)")
    return idx, int(idx.stride(0)) if n > 1 else 1


def sinkhorn_assign(resid, codebook, epsilon, iters, group_offsets=None, out=None):
    """use_sk branch of VectorQuantizer.forward (vq.py:76-83): int64 [n] Sinkhorn assignments.

    group_offsets: host sequence of ascending row offsets delimiting independent problems
    (default: one group = all rows, the training case).  `out` may be a column view of [n, L]."""
    lib = _lib.load()
    resid = _dev(resid, "resid")
    codebook = _dev(codebook, "codebook")
    n, e = resid.shape
    K = codebook.shape[0]
    import numpy as np
    offs = np.ascontiguousarray([0, n] if group_offsets is None else group_offsets, dtype=np.int64)
    G = len(offs) - 1
    if out is None:
        out = torch.zeros(n, dtype=torch.int64, device=resid.device)
    out, stride = _idx_col(out, n)
    oarr = offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    with _on(resid.device):
        nbytes = lib.lcrec_sinkhorn_assign_workspace(n, K, oarr, G)
        ws = _workspace(nbytes, resid.device)
        rc = lib.lcrec_sinkhorn_assign(_ptr(resid), n, e, _ptr(codebook), K, oarr, G, float(epsilon), int(iters),
                                       _ptr(out), stride, _ptr(ws), ws.numel(), _context(resid.device),
                                       _ptr(_ticket(resid.device)), _stream_ptr())
    _lib.check(rc, "lcrec_sinkhorn_assign")
    if G > 0 and int(np.diff(offs).max()) * K > 16384:
        # the one-launch solver for batch-sized problems poisons its output with -1 if its (bounded)
        # grid barrier ever times out; turn that into an error rather than training on garbage --
        # here and now, or (inside deferred_checks(), the trainer's epoch loop) when the block ends,
        # so that a training step has no host synchronisation of its own
        # (a single problem: the kernel poisons EVERY row -- whoever timed out set the flag all workgroups read before they
        # write -- so the first row tells; several groups: only the rows of the group that took that path)
        if _deferred is not None and _deferred_raw and G == 1:
            # the caller folds the test into a kernel of its own (lcrec_step_losses' poison_probe): hand it the element to look at
            _deferred.append((_POISON_MSG, out[0:1]))
            return out
        bad = (out[0] < 0) if G == 1 else (out < 0).any()
        if _deferred is not None:
            _deferred.append(("lcrec_sinkhorn_assign: grid barrier timed out (device oversubscribed?); "
                              "set LCREC_SINKHORN_PERSISTENT=0 to use the multi-launch solver", bad))
        elif bool(bad):
            raise _lib.LcrecError("lcrec_sinkhorn_assign: grid barrier timed out (device oversubscribed?); "
                                  "set LCREC_SINKHORN_PERSISTENT=0 to use the multi-launch solver")
    return out


_deferred = None
_deferred_raw = False
_POISON_MSG = ("lcrec_sinkhorn_assign: grid barrier timed out (device oversubscribed?); "
               "set LCREC_SINKHORN_PERSISTENT=0 to use the multi-launch solver")


class deferred_checks:
    """Context in which result checks that need a device->host read (the Sinkhorn poison flag) are
    collected as device booleans and evaluated together at exit, or every `every` collected checks."""

    def __init__(self, every=256, raw=False):
        """raw: single-problem Sinkhorn calls append (message, int64 [1] view of their first assignment) instead of a device
        boolean -- no compare launch; the caller (engine.py) drains them and tests the element itself (< 0 = poisoned)."""
        self.every = every
        self.raw = raw

    def __enter__(self):
        global _deferred, _deferred_raw
        self._outer = (_deferred, _deferred_raw)
        _deferred, _deferred_raw = [], self.raw
        return self

    def flush(self):
        global _deferred
        pending, _deferred = _deferred, []
        if pending and bool(torch.stack([b for _, b in pending]).any()):
            for msg, b in pending:
                if bool(b):
                    raise _lib.LcrecError(msg)

    def poll(self):
        if _deferred is not None and len(_deferred) >= self.every:
            self.flush()

    @staticmethod
    def drain():
        """Hand the collected (message, device bool) pairs to the caller without evaluating them (a graph capture folds
        them into a device-side flag instead of reading them back)."""
        global _deferred
        pending, _deferred = (_deferred or []), []
        return pending

    def __exit__(self, exc_type, exc, tb):
        global _deferred, _deferred_raw
        try:
            if exc_type is None and not self.raw:
                self.flush()
        finally:
            _deferred, _deferred_raw = self._outer
        return False


def _sse_buffer(sse_out, L, dev, n=1):
    """float64 [L] for a kernel's per-level sums of squares: the caller's view (a slice of its own buffer) or a new one.
    The kernels write every slot, so there is no zero fill -- except for an empty batch, which launches nothing."""
    if sse_out is None:
        return (torch.zeros if n == 0 else torch.empty)(L, dtype=torch.float64, device=dev)
    if not (sse_out.is_cuda and sse_out.dtype == torch.float64 and sse_out.is_contiguous() and sse_out.numel() == L):
        raise _lib.LcrecError(f"sse_out must be a contiguous float64 [{L}] device tensor")
    if n == 0:
        sse_out.zero_()
    return sse_out


def rq_apply_level(resid, codebook, idx, xq=None, want_sse=False, sse_out=None):
    """Gather + STE + residual update of one level for given indices (vq.py:87-95, rq.py:47-48).

    xq: None (start a new x_q sum) or the running [n, e] sum, updated in place.
    Returns (xq, resid_next, sse float64 [1] | None)."""
    lib = _lib.load()
    resid = _dev(resid, "resid")
    codebook = _dev(codebook, "codebook")
    n, e = resid.shape
    K = codebook.shape[0]
    idx, stride = _idx_col(idx, n)
    accumulate = xq is not None
    if xq is None:
        xq = torch.empty((n, e), dtype=torch.float32, device=resid.device)
    elif not (xq.is_cuda and xq.is_contiguous() and xq.dtype == torch.float32 and tuple(xq.shape) == (n, e)):
        raise _lib.LcrecError("xq must be a contiguous [n, e] float32 device tensor")
    nxt = torch.empty_like(resid)
    sse = _sse_buffer(sse_out, 1, resid.device, n) if want_sse else None
    with _on(resid.device):
        ws = _workspace(8192, resid.device)
        rc = lib.lcrec_rq_apply_level(_ptr(resid), n, e, _ptr(codebook), K, _ptr(idx), stride, _ptr(xq),
                                      int(accumulate), _ptr(nxt), _ptr(sse), _ptr(ws), ws.numel(),
                                      _ptr(_ticket(resid.device)) if want_sse else None, _stream_ptr())
    _lib.check(rc, "lcrec_rq_apply_level")
    return xq, nxt, sse


def code_stats(idx, resid, K):
    """count [K] and per-code residual sums [K, e] in item order (index_improve vq.py:151-167)."""
    lib = _lib.load()
    resid = _dev(resid, "resid")
    n, e = resid.shape
    idx, stride = _idx_col(idx, n)
    count = torch.empty(K, dtype=torch.float32, device=resid.device)
    total = torch.empty((K, e), dtype=torch.float32, device=resid.device)
    with _on(resid.device):
        rc = lib.lcrec_code_stats(_ptr(idx), stride, _ptr(resid), n, e, K, _ptr(count), _ptr(total), _stream_ptr())
    _lib.check(rc, "lcrec_code_stats")
    return count, total


def code_stats_levels(idx, resid_in, ks, codebooks=None, grads_out=None, scale=0.0, weight=0.0):
    """[(count, sum)] for every level of an [n, L] index matrix in one launch (lcrec_code_stats_levels); with codebooks and
    grads_out (lists of [K_l, e] tensors) the codebook gradients (scale * (count*C - sum)) * weight are written too."""
    lib = _lib.load()
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 2 and idx.is_contiguous()):
        raise _lib.LcrecError("idx must be a contiguous int64 [n, L] device tensor")
    n, L = idx.shape
    resid = [_dev(r, "resid") for r in resid_in]
    e = resid[0].shape[1]
    dev = idx.device
    counts = [torch.empty(int(k), dtype=torch.float32, device=dev) for k in ks]
    sums = [torch.empty((int(k), e), dtype=torch.float32, device=dev) for k in ks]
    PA = ctypes.c_void_p * L
    fused = codebooks is not None
    cbs = [_dev(c, "codebook") for c in codebooks] if fused else None
    with _on(dev):
        rc = lib.lcrec_code_stats_levels(_ptr(idx), PA(*[r.data_ptr() for r in resid]), n, e, _ints(ks), L,
                                         PA(*[c.data_ptr() for c in counts]), PA(*[t.data_ptr() for t in sums]),
                                         PA(*[c.data_ptr() for c in cbs]) if fused else None,
                                         PA(*[g.data_ptr() for g in grads_out]) if fused else None, float(scale), float(weight),
                                         _stream_ptr())
    _lib.check(rc, "lcrec_code_stats_levels")
    return list(zip(counts, sums))


def ema_update(ema_count, ema_sum, codebook, count, total, decay, eps, skip_flag=None):
    """In-place EMA step of index_improve vq.py:155-184 on three contiguous fp32 device tensors.  skip_flag: a device
    bool/uint8 scalar; when set the call changes nothing (lcrec_ema_update)."""
    lib = _lib.load()
    for name, t in (("ema_count", ema_count), ("ema_sum", ema_sum), ("codebook", codebook)):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise _lib.LcrecError(f"{name} must be a contiguous float32 device tensor (it is updated in place)")
    count, total = _dev(count, "count"), _dev(total, "sum")
    K, e = codebook.shape
    alpha = 1 - decay                  # the reference's python doubles, rounded to fp32 at the ABI
    keep = 1 - (1 - decay)
    with _on(codebook.device):
        rc = lib.lcrec_ema_update(_ptr(ema_count), _ptr(ema_sum), _ptr(codebook), _ptr(count), _ptr(total), K, e,
                                  decay, alpha, keep, eps, _ptr(skip_flag), _stream_ptr())
    _lib.check(rc, "lcrec_ema_update")


def collision_groups(idx, ks, want_groups=True):
    """Tuple collisions of an int64 [n, L] index matrix (trainer.py:139-150, generate_indices.py:18-42).

    Returns dict(unique, max_count, collision_rate, n_groups) and, with want_groups, the groups in
    get_collision_item's order (first occurrence of the tuple; ids ascending inside a group):
      want_groups=True      `groups`: a list of item-id lists (small inputs, tests, the reference's helper API);
      want_groups="device"  `members` int64 [items in groups] and `offsets` int64 [n_groups + 1] device
                            tensors -- group g = members[offsets[g]:offsets[g+1]] -- no Python lists."""
    lib = _lib.load()
    if not (idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 2):
        raise _lib.LcrecError("idx must be an int64 [n, L] device tensor")
    idx = idx.contiguous()
    n, L = idx.shape
    dev = idx.device
    counters = torch.zeros(4, dtype=torch.int64, device=dev)
    members = torch.empty(max(n, 1), dtype=torch.int64, device=dev) if want_groups else None
    offsets = torch.empty(n // 2 + 2, dtype=torch.int64, device=dev) if want_groups else None
    karr = _ints(ks)
    with _on(dev):
        nbytes = lib.lcrec_collision_groups_workspace(n, L)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_collision_groups(_ptr(idx), n, L, karr, _ptr(members), _ptr(offsets), _ptr(counters), _ptr(ws),
                                        ws.numel(), _stream_ptr())
    _lib.check(rc, "lcrec_collision_groups")
    c = counters.tolist()
    out = {"unique": c[0], "max_count": c[3], "collision_rate": (n - c[0]) / n if n else 0.0}
    if want_groups:
        out["n_groups"] = c[1]
        if want_groups == "device":
            out["members"] = members[: c[2]]
            out["offsets"] = offsets[: c[1] + 1]
        else:
            offs = offsets[: c[1] + 1].tolist()
            mem = members[: c[2]].tolist()
            out["groups"] = [mem[offs[g]:offs[g + 1]] for g in range(c[1])]
    return out


def index_json_text(idx_rows, first_item=0):
    """bytes of the `.index.json` entries of items first_item.. for a HOST int64 [n, L] array
    (generate_indices.py:83-92,138-145; see lcrec_index_json_format in include/lcrec.h)."""
    import numpy as np
    lib = _lib.load()
    a = np.ascontiguousarray(idx_rows, dtype=np.int64)
    if a.ndim != 2:
        raise _lib.LcrecError("idx_rows must be [n, L]")
    n, L = a.shape
    if n == 0:
        return b""
    cap = lib.lcrec_index_json_bound(n, L)
    buf = np.empty(cap, dtype=np.uint8)
    got = lib.lcrec_index_json_format(a.ctypes.data, n, L, int(first_item), buf.ctypes.data, cap)
    if got < 0:
        _lib.check(int(got), "lcrec_index_json_format")
    return buf[:got].tobytes()


# ---- training-step kernels (csrc/train_ops.hip)
def _vec(t, name, F):
    if t is None:
        return None
    t = _dev(t, name)
    if t.numel() != F:
        raise _lib.LcrecError(f"{name} must have {F} elements")
    return t


def bn_relu_forward(t, gamma, beta, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, relu=True):
    """Training-mode BatchNorm1d (+ReLU) of a Linear's output (layers.py:25-30): returns (y, mean, rstd); the running
    statistics (contiguous fp32 buffers of the module) are updated in place."""
    lib = _lib.load()
    t = _dev(t, "t")
    n, F = t.shape
    gamma, beta = _vec(gamma, "gamma", F), _vec(beta, "beta", F)
    for name, buf in (("running_mean", running_mean), ("running_var", running_var)):
        if buf is not None and not (buf.is_cuda and buf.is_contiguous() and buf.dtype == torch.float32 and buf.numel() == F):
            raise _lib.LcrecError(f"{name} must be a contiguous float32 device buffer of {F} elements")
    y = torch.empty_like(t)
    mean = torch.empty(F, dtype=torch.float32, device=t.device)
    rstd = torch.empty(F, dtype=torch.float32, device=t.device)
    with _on(t.device):
        rc = lib.lcrec_bn_relu_forward(_ptr(t), n, F, _ptr(gamma), _ptr(beta), float(eps), float(momentum), _ptr(running_mean),
                                       _ptr(running_var), _ptr(y), _ptr(mean), _ptr(rstd), int(bool(relu)), _stream_ptr())
    _lib.check(rc, "lcrec_bn_relu_forward")
    return y, mean, rstd


def bn_relu_backward(gy, t, y, gamma, mean, rstd, relu=True, dgamma_out=None, dbeta_out=None, dbias_out=None, fold=None, beta=None):
    """(dt, dgamma, dbeta, dbias) of y = [relu](bn(t)) for gy = dL/dy (see lcrec_bn_relu_backward); the three vector
    outputs may be given (views of a flat gradient buffer).  y None + fold = (scale, shift): the ReLU mask is recomputed as
    [t * scale + shift > 0].  y None + beta: the mask is recomputed with bn_relu_forward's own expression -- the bits of
    the y it wrote, without reading it."""
    lib = _lib.load()
    gy, t = _dev(gy, "gy"), _dev(t, "t")
    y = None if y is None else _dev(y, "y")
    n, F = t.shape
    gamma = _vec(gamma, "gamma", F)
    mk = lambda o: o if o is not None else torch.empty(F, dtype=torch.float32, device=t.device)
    dgamma, dbeta, dbias = mk(dgamma_out), mk(dbeta_out), mk(dbias_out)
    dt = torch.empty_like(t)
    with _on(t.device):
        rc = lib.lcrec_bn_relu_backward(_ptr(gy), _ptr(t), _ptr(y), n, F, _ptr(gamma), _ptr(_vec(mean, "mean", F)),
                                        _ptr(_vec(rstd, "rstd", F)), int(bool(relu)), _ptr(dt), _ptr(dgamma), _ptr(dbeta),
                                        _ptr(dbias), _ptr(None if fold is None else _vec(fold[0], "fold_scale", F)),
                                        _ptr(_vec(fold[1], "fold_shift", F) if fold is not None else
                                             (_vec(beta, "beta", F) if (y is None and beta is not None) else None)), _stream_ptr())
    _lib.check(rc, "lcrec_bn_relu_backward")
    return dt, dgamma, dbeta, dbias


def bn_stats(t, row_out=None):
    """(mean, m2) of this rank's rows: m2 = sum (t - mean)^2 (lcrec_bn_stats).
    row_out: a float32 [2F+1] exchange row (n_r, mean[F], m2[F]) to write them into (slot 0 is the caller's)."""
    lib = _lib.load()
    t = _dev(t, "t")
    n, F = t.shape
    if row_out is not None:
        if not (row_out.is_cuda and row_out.dtype == torch.float32 and row_out.is_contiguous() and row_out.numel() == 2 * F + 1):
            raise _lib.LcrecError("row_out must be a contiguous float32 [2F+1] device tensor")
        mean, m2 = row_out[1:F + 1], row_out[F + 1:]
    else:
        mean = torch.empty(F, dtype=torch.float32, device=t.device)
        m2 = torch.empty(F, dtype=torch.float32, device=t.device)
    with _on(t.device):
        rc = lib.lcrec_bn_stats(_ptr(t), n, F, _ptr(mean), _ptr(m2), _stream_ptr())
    _lib.check(rc, "lcrec_bn_stats")
    return mean, m2


def bn_merge_stats(rows, eps, momentum=0.0, running_mean=None, running_var=None):
    """(mean, rstd) of the union of the ranks' rows from rows [world, 2F+1] = (n_r, mean_r, m2_r); updates the running
    statistics in place when given (lcrec_bn_merge_stats)."""
    lib = _lib.load()
    rows = _dev(rows, "rows")
    world, width = rows.shape
    F = (width - 1) // 2
    mean = torch.empty(F, dtype=torch.float32, device=rows.device)
    rstd = torch.empty(F, dtype=torch.float32, device=rows.device)
    with _on(rows.device):
        rc = lib.lcrec_bn_merge_stats(_ptr(rows), world, F, float(eps), float(momentum), _ptr(mean), _ptr(rstd),
                                      _ptr(None if running_mean is None else _vec(running_mean, "running_mean", F)),
                                      _ptr(None if running_var is None else _vec(running_var, "running_var", F)), _stream_ptr())
    _lib.check(rc, "lcrec_bn_merge_stats")
    return mean, rstd


def bn_relu_apply(t, gamma, beta, mean, rstd, relu=True):
    lib = _lib.load()
    t = _dev(t, "t")
    n, F = t.shape
    y = torch.empty_like(t)
    with _on(t.device):
        rc = lib.lcrec_bn_relu_apply(_ptr(t), n, F, _ptr(_vec(gamma, "gamma", F)), _ptr(_vec(beta, "beta", F)),
                                     _ptr(_vec(mean, "mean", F)), _ptr(_vec(rstd, "rstd", F)), int(bool(relu)), _ptr(y),
                                     _stream_ptr())
    _lib.check(rc, "lcrec_bn_relu_apply")
    return y


def bn_backward_reduce(gy, t, y, mean, rstd, relu=True, dbeta_out=None, dgamma_out=None):
    """float32 [2, F]: (sum g, sum g * xhat) over this rank's rows (lcrec_bn_backward_reduce); dbeta_out / dgamma_out
    receive a copy of the two rows (this rank's share of the BatchNorm parameter gradients)."""
    lib = _lib.load()
    gy, t = _dev(gy, "gy"), _dev(t, "t")
    n, F = t.shape
    sums = torch.empty((2, F), dtype=torch.float32, device=t.device)
    with _on(t.device):
        rc = lib.lcrec_bn_backward_reduce(_ptr(gy), _ptr(t), _ptr(None if y is None else _dev(y, "y")), n, F,
                                          _ptr(_vec(mean, "mean", F)), _ptr(_vec(rstd, "rstd", F)), int(bool(relu)),
                                          _ptr(sums[0]), _ptr(sums[1]),
                                          _ptr(None if dbeta_out is None else _vec(dbeta_out, "dbeta_out", F)),
                                          _ptr(None if dgamma_out is None else _vec(dgamma_out, "dgamma_out", F)), _stream_ptr())
    _lib.check(rc, "lcrec_bn_backward_reduce")
    return sums


def bn_backward_apply(gy, t, y, gamma, mean, rstd, sums, n_total, relu=True, dbias_out=None):
    """(dt, dbias) from the all-reduced [2, F] sums (lcrec_bn_backward_apply)."""
    lib = _lib.load()
    gy, t = _dev(gy, "gy"), _dev(t, "t")
    n, F = t.shape
    sums = _dev(sums, "sums")
    dt = torch.empty_like(t)
    dbias = _vec(dbias_out, "dbias_out", F) if dbias_out is not None else torch.empty(F, dtype=torch.float32, device=t.device)
    with _on(t.device):
        rc = lib.lcrec_bn_backward_apply(_ptr(gy), _ptr(t), _ptr(None if y is None else _dev(y, "y")), n, F,
                                         _ptr(_vec(gamma, "gamma", F)), _ptr(_vec(mean, "mean", F)), _ptr(_vec(rstd, "rstd", F)),
                                         int(bool(relu)), _ptr(sums[0]), _ptr(sums[1]), float(n_total), _ptr(dt), _ptr(dbias),
                                         _stream_ptr())
    _lib.check(rc, "lcrec_bn_backward_apply")
    return dt, dbias


def relu_bias_backward(gy, y, relu=True, dbias_out=None, inplace=False):
    """(g, dbias): g = gy * [y > 0] (or gy itself when relu is False), dbias = column sums of g."""
    lib = _lib.load()
    gy = _dev(gy, "gy")
    n, F = gy.shape
    y = _dev(y, "y") if relu else None
    g = gy if (inplace or not relu) else torch.empty_like(gy)
    dbias = dbias_out if dbias_out is not None else torch.empty(F, dtype=torch.float32, device=gy.device)
    with _on(gy.device):
        rc = lib.lcrec_relu_bias_backward(_ptr(gy), _ptr(y), n, F, int(bool(relu)), _ptr(g) if relu else None, _ptr(dbias),
                                          _stream_ptr())
    _lib.check(rc, "lcrec_relu_bias_backward")
    return g, dbias


def recon_loss_grad(out, x, loss_type="mse", want_grad=True, global_rows=None):
    """(loss float32 scalar tensor, grad | None) of rqvae.py:74-85's reconstruction term.  global_rows: rows of the global
    batch when `out` is one rank's shard of it -- loss and gradient are then this rank's share of the global mean's."""
    lib = _lib.load()
    out, x = _dev(out, "out"), _dev(x, "x")
    if out.shape != x.shape:
        raise _lib.LcrecError(f"recon_loss_grad: shapes {tuple(out.shape)} vs {tuple(x.shape)}")
    if loss_type not in ("mse", "l1"):
        raise ValueError("incompatible loss type")
    g = torch.empty_like(out) if want_grad else None
    loss = torch.empty((), dtype=torch.float32, device=out.device)
    with _on(out.device):
        ws = _workspace(lib.lcrec_train_reduce_workspace(), out.device)
        total = 0 if global_rows is None else int(global_rows) * (out.numel() // max(1, out.shape[0]))
        rc = lib.lcrec_recon_loss_grad(_ptr(out), _ptr(x), out.numel(), total, int(loss_type == "l1"), _ptr(g), _ptr(loss), _ptr(ws),
                                       ws.numel(), _ptr(_ticket(out.device)), _stream_ptr())
    _lib.check(rc, "lcrec_recon_loss_grad")
    return loss, g


def grad_norm_clip(flat_grads, max_norm=1.0, out=None):
    """float32 [2] device tensor: (global L2 norm, clip coefficient) of a flat gradient buffer (trainer.py:118)."""
    lib = _lib.load()
    g = _dev(flat_grads, "grads")
    res = out if out is not None else torch.empty(2, dtype=torch.float32, device=g.device)
    with _on(g.device):
        ws = _workspace(lib.lcrec_train_reduce_workspace(), g.device)
        rc = lib.lcrec_grad_norm_clip(_ptr(g), g.numel(), float(max_norm), _ptr(res), _ptr(ws), ws.numel(), _ptr(_ticket(g.device)),
                                      _stream_ptr())
    _lib.check(rc, "lcrec_grad_norm_clip")
    return res


def codebook_grad(count, total, codebook, scale, weight, out):
    """out[k] = (scale * (count[k] * C[k] - sum[k])) * weight (see lcrec_codebook_grad); `out` is written in place."""
    lib = _lib.load()
    count, total, codebook = _dev(count, "count"), _dev(total, "sum"), _dev(codebook, "codebook")
    K, e = codebook.shape
    if not (out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (K, e)):
        raise _lib.LcrecError("out must be a contiguous float32 [K, e] device tensor")
    with _on(codebook.device):
        rc = lib.lcrec_codebook_grad(_ptr(count), _ptr(total), _ptr(codebook), K, e, float(scale), float(weight), _ptr(out),
                                     _stream_ptr())
    _lib.check(rc, "lcrec_codebook_grad")
    return out


def step_losses(sse, n, e, beta, quant_loss_weight, recon, losses_out, sums=None, nan_flag=None, poison_probe=None,
                poison_flag=None):
    """losses_out[3] = (loss, recon, rq_loss) from the per-level sse (float64 [L]) and the reconstruction loss; optional
    running sums (float64 [2], +=) and NaN flag (bool/uint8 scalar, set) -- lcrec_step_losses.  poison_probe (int64 [1]
    device view) / poison_flag (bool/uint8 scalar): the flag is set when the probed assignment is negative."""
    lib = _lib.load()
    if not (sse.is_cuda and sse.dtype == torch.float64 and sse.is_contiguous()):
        raise _lib.LcrecError("sse must be a contiguous float64 device tensor")
    with _on(sse.device):
        rc = lib.lcrec_step_losses(_ptr(sse), sse.numel(), int(n), int(e), float(beta), float(quant_loss_weight), _ptr(recon),
                                   _ptr(losses_out), _ptr(sums), _ptr(nan_flag), _ptr(poison_probe),
                                   _ptr(poison_flag) if poison_probe is not None else None, _stream_ptr())
    _lib.check(rc, "lcrec_step_losses")
    return losses_out


def quantizer_input_grad(z, codebook0, idx_col, coef, weight, g_xq, dbias_out=None):
    """(coef * (z - C0[idx0])) * weight + g_xq (lcrec_quantizer_input_grad); with dbias_out [e] also its column sums, the bias
    gradient of the encoder's last Linear, in the same launch (lcrec_quantizer_input_grad_bias)."""
    lib = _lib.load()
    z, codebook0, g_xq = _dev(z, "z"), _dev(codebook0, "codebook"), _dev(g_xq, "g_xq")
    n, e = z.shape
    idx_col, stride = _idx_col(idx_col, n)
    out = torch.empty_like(z)
    with _on(z.device):
        if dbias_out is None:
            rc = lib.lcrec_quantizer_input_grad(_ptr(z), _ptr(codebook0), _ptr(idx_col), stride, n, e, float(coef), float(weight),
                                                _ptr(g_xq), _ptr(out), _stream_ptr())
        else:
            rc = lib.lcrec_quantizer_input_grad_bias(_ptr(z), _ptr(codebook0), _ptr(idx_col), stride, n, e, float(coef), float(weight),
                                                     _ptr(g_xq), _ptr(out), _ptr(_vec(dbias_out, "dbias_out", e)), _stream_ptr())
    _lib.check(rc, "lcrec_quantizer_input_grad")
    return out


def adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=True,
               clip=None, schedule=-1, warmup_steps=0, total_steps=0, lr_out=None, skip_flag=None):
    """One AdamW/Adam step on flat fp32 buffers, all updated in place (see lcrec_adamw_step); `step` is a device int64
    scalar the call increments.  skip_flag: a device bool/uint8 scalar; when set nothing is updated."""
    lib = _lib.load()
    for name, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.numel() == params.numel()):
            raise _lib.LcrecError(f"{name} must be a contiguous float32 device buffer of {params.numel()} elements")
    if not (step.is_cuda and step.dtype == torch.int64 and step.numel() == 1):
        raise _lib.LcrecError("step must be a device int64 scalar")
    with _on(params.device):
        rc = lib.lcrec_adamw_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), params.numel(), _ptr(clip),
                                  _ptr(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                  int(bool(decoupled)), int(schedule), int(warmup_steps), int(total_steps), _ptr(lr_out),
                                  _ptr(_ticket(params.device)), _ptr(skip_flag), _stream_ptr())
    _lib.check(rc, "lcrec_adamw_step")


def trace_enable(on=True):
    """Bracket every kernel launch with hipEvents (include/lcrec.h, lcrec_trace_enable)."""
    _lib.check(_lib.load().lcrec_trace_enable(int(bool(on))), "lcrec_trace_enable")


def trace_collect():
    """{kernel name: (launches, total_ms)} since trace_enable(); waits for the recorded events."""
    buf = (_lib.TraceEntry * 32)()
    n = _lib.load().lcrec_trace_collect(ctypes.cast(buf, ctypes.c_void_p), 32)
    if n < 0:
        _lib.check(n, "lcrec_trace_collect")
    return {buf[i].kernel.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(n)}
