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


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def release_workspaces():
    _workspaces.clear()


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
    with torch.cuda.device(x.device):
        rc = lib.lcrec_linear_forward(_ptr(x), n, k, _ptr(weight), _ptr(bias), _ptr(bn_scale), _ptr(bn_shift),
                                      int(bool(relu)), out_dim, _ptr(y), _stream_ptr())
    _lib.check(rc, "lcrec_linear_forward")
    return y


def flatten_codebooks(codebooks):
    """List of [K_l, e] tensors -> (flat fp32 tensor, [K_l]) in the layout lcrec_rq_assign expects."""
    ks = [int(c.shape[0]) for c in codebooks]
    flat = torch.cat([_dev(c, "codebook").reshape(-1) for c in codebooks])
    return flat, ks


def rq_assign(z, codebooks_flat, ks, want_xq=False, want_sse=False, want_resid=False):
    """ResidualVectorQuantizer.forward values with use_sk=False (rq.py:39-55).

    Returns (idx int64 [n, L], xq [n, e] | None, sse float64 [L] | None, resid [L, n, e] | None)."""
    lib = _lib.load()
    z = _dev(z, "z")
    cb = _dev(codebooks_flat, "codebooks")
    n, e = z.shape
    L = len(ks)
    dev = z.device
    idx = torch.empty((n, L), dtype=torch.int64, device=dev)
    xq = torch.empty((n, e), dtype=torch.float32, device=dev) if want_xq else None
    sse = torch.zeros(L, dtype=torch.float64, device=dev) if want_sse else None
    resid = torch.empty((L, n, e), dtype=torch.float32, device=dev) if want_resid else None
    karr = _ints(ks)
    with torch.cuda.device(dev):
        nbytes = lib.lcrec_rq_assign_workspace(n, e, karr, L)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_rq_assign(_ptr(z), n, e, _ptr(cb), karr, L, _ptr(idx), _ptr(xq), _ptr(sse), _ptr(resid),
                                 _ptr(ws), ws.numel(), _stream_ptr())
    _lib.check(rc, "lcrec_rq_assign")
    return idx, xq, sse, resid


def encode_assign(x, weights, biases, codebooks_flat, ks, bn_scales=None, bn_shifts=None, want_latent=False,
                  want_xq=False, want_sse=False):
    """RQVAE.get_indices(xs, use_sk=False) (rqvae.py:68-72): encoder MLP + L-level argmin assignment.

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
    with torch.cuda.device(dev):
        nbytes = lib.lcrec_encode_assign_workspace(n, darr, nl, karr, L)
        ws = _workspace(nbytes, dev)
        rc = lib.lcrec_encode_assign(_ptr(x), n, darr, nl, wp, bp, scp, shp, _ptr(cb), karr, L, _ptr(idx),
                                     _ptr(latent), _ptr(xq), _ptr(sse), _ptr(ws), ws.numel(), _stream_ptr())
    _lib.check(rc, "lcrec_encode_assign")
    return idx, latent, xq, sse


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
