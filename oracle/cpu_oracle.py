"""numpy front-end of oracle/liblcrec_oracle.so (the CPU restatement in lcrec_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package (lc-rec_amd/).
"""
import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblcrec_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int)


def build():
    """Compile lcrec_oracle.c (gcc is present here and on the GPU box)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblcrec_oracle.so"])


def _load():
    src = os.path.join(_HERE, "lcrec_oracle.c")
    if not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        build()
    lib = ctypes.CDLL(_SO)
    lib.lcrec_oracle_linear.restype = ctypes.c_int
    lib.lcrec_oracle_linear.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, _f32p, _f32p,
                                        ctypes.c_int, ctypes.c_int, _f32p]
    lib.lcrec_oracle_rq_assign.restype = ctypes.c_int
    lib.lcrec_oracle_rq_assign.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _i32p, ctypes.c_int,
                                           _i64p, _f32p, _f64p, _f32p]
    lib.lcrec_oracle_encode_assign.restype = ctypes.c_int
    lib.lcrec_oracle_encode_assign.argtypes = [
        _f32p, ctypes.c_int64, _i32p, ctypes.c_int, ctypes.POINTER(_f32p), ctypes.POINTER(_f32p),
        ctypes.POINTER(_f32p), ctypes.POINTER(_f32p), _f32p, _i32p, ctypes.c_int, _i64p, _f32p, _f32p, _f64p]
    lib.lcrec_oracle_rq_assign_m.restype = ctypes.c_int
    lib.lcrec_oracle_rq_assign_m.argtypes = lib.lcrec_oracle_rq_assign.argtypes + [_f32p, _f32p]
    lib.lcrec_oracle_encode_assign_m.restype = ctypes.c_int
    lib.lcrec_oracle_encode_assign_m.argtypes = lib.lcrec_oracle_encode_assign.argtypes + [_f32p, _f32p]
    lib.lcrec_oracle_code_stats.restype = ctypes.c_int
    lib.lcrec_oracle_code_stats.argtypes = [_i64p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, _f32p, _f32p]
    lib.lcrec_oracle_ema_update.restype = ctypes.c_int
    lib.lcrec_oracle_ema_update.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float]
    lib.lcrec_oracle_affine_relu.restype = ctypes.c_int
    lib.lcrec_oracle_affine_relu.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, _f32p, ctypes.c_int, _f32p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a, t=_f32p):
    return None if a is None else a.ctypes.data_as(t)


def _split(n, threads):
    threads = max(1, min(threads, (n + 63) // 64))
    step = -(-n // threads)
    step = (step + 3) // 4 * 4
    return [(s, min(n, s + step)) for s in range(0, n, step)] if n else []


def linear(x, W, b=None, bn_scale=None, bn_shift=None, relu=False, threads=1):
    """layers.py:42 for one Linear(+BN eval)(+ReLU) group, canonical fma-chain arithmetic."""
    x, W = _f32(x), _f32(W)
    b = None if b is None else _f32(b)
    sc = None if bn_scale is None else _f32(bn_scale)
    sh = None if bn_shift is None else _f32(bn_shift)
    n, k = x.shape
    out = W.shape[0]
    assert W.shape[1] == k
    y = np.empty((n, out), dtype=np.float32)

    def run(lo, hi):
        rc = lib().lcrec_oracle_linear(_p(x[lo:hi]), hi - lo, k, _p(W), _p(b), _p(sc), _p(sh), int(relu), out,
                                       _p(y[lo:hi]))
        assert rc == 0, rc

    parts = _split(n, threads)
    if len(parts) <= 1:
        for lo, hi in parts:
            run(lo, hi)
    else:
        with ThreadPoolExecutor(len(parts)) as ex:
            list(ex.map(lambda t: run(*t), parts))
    return y


def affine_relu(x, scale, shift, relu=True):
    """max(fma(x, scale, shift), 0): a folded training-mode BatchNorm + ReLU as the fused kernels apply it to an operand."""
    x, scale, shift = _f32(x), _f32(scale), _f32(shift)
    n, f = x.shape
    u = np.empty_like(x)
    rc = lib().lcrec_oracle_affine_relu(_p(x), n, f, _p(scale), _p(shift), int(bool(relu)), _p(u))
    assert rc == 0
    return u


def rq_assign(z, codebooks, want_resid=False, want_margin=False):
    """rq.py:39-55 over vq.py:63-99 (argmin branch).

    codebooks: list of [K_l, e] arrays.  Returns dict(idx, xq, sse[, resid])."""
    z = _f32(z)
    n, e = z.shape
    Ks = np.asarray([c.shape[0] for c in codebooks], dtype=np.int32)
    cb = _f32(np.concatenate([_f32(c).reshape(-1) for c in codebooks]))
    L = len(codebooks)
    idx = np.empty((n, L), dtype=np.int64)
    xq = np.empty((n, e), dtype=np.float32)
    sse = np.zeros(L, dtype=np.float64)
    resid = np.empty((L + 1, n, e), dtype=np.float32) if want_resid else None
    margin = np.empty((n, L), dtype=np.float32) if want_margin else None
    scale = np.empty((n, L), dtype=np.float32) if want_margin else None
    rc = lib().lcrec_oracle_rq_assign_m(_p(z), n, e, _p(cb), _p(Ks, _i32p), L, _p(idx, _i64p), _p(xq), _p(sse, _f64p),
                                        _p(resid), _p(margin), _p(scale))
    assert rc == 0, rc
    out = {"idx": idx, "xq": xq, "sse": sse}
    if want_margin:
        out["margin"], out["scale"] = margin, scale
    if want_resid:
        out["resid"] = resid
    return out


def encode_assign(x, weights, biases, codebooks, bn_scale=None, bn_shift=None, threads=1, want_margin=False):
    """rqvae.py:68-72 (get_indices, use_sk=False).  weights[l] is [out_l, in_l]."""
    x = _f32(x)
    n = x.shape[0]
    nl = len(weights)
    weights = [_f32(w) for w in weights]
    biases = [_f32(b) for b in biases]
    dims = np.asarray([weights[0].shape[1]] + [w.shape[0] for w in weights], dtype=np.int32)
    Ks = np.asarray([c.shape[0] for c in codebooks], dtype=np.int32)
    cb = _f32(np.concatenate([_f32(c).reshape(-1) for c in codebooks]))
    L = len(codebooks)
    e = int(dims[-1])
    PA = _f32p * nl
    Wp = PA(*[_p(w) for w in weights])
    bp = PA(*[_p(b) for b in biases])
    scs = [None if (bn_scale is None or s is None) else _f32(s) for s in (bn_scale or [None] * nl)]
    shs = [None if (bn_shift is None or s is None) else _f32(s) for s in (bn_shift or [None] * nl)]
    scp = PA(*[_p(s) if s is not None else ctypes.cast(None, _f32p) for s in scs])
    shp = PA(*[_p(s) if s is not None else ctypes.cast(None, _f32p) for s in shs])
    idx = np.empty((n, L), dtype=np.int64)
    lat = np.empty((n, e), dtype=np.float32)
    xq = np.empty((n, e), dtype=np.float32)
    margin = np.empty((n, L), dtype=np.float32) if want_margin else None
    scale = np.empty((n, L), dtype=np.float32) if want_margin else None
    sses = []

    def run(lo, hi):
        sse = np.zeros(L, dtype=np.float64)
        rc = lib().lcrec_oracle_encode_assign_m(_p(x[lo:hi]), hi - lo, _p(dims, _i32p), nl, Wp, bp, scp, shp, _p(cb),
                                                _p(Ks, _i32p), L, _p(idx[lo:hi], _i64p), _p(lat[lo:hi]),
                                                _p(xq[lo:hi]), _p(sse, _f64p),
                                                _p(margin[lo:hi]) if want_margin else None,
                                                _p(scale[lo:hi]) if want_margin else None)
        assert rc == 0, rc
        sses.append((lo, sse))

    parts = _split(n, threads)
    if len(parts) <= 1:
        for lo, hi in parts:
            run(lo, hi)
    else:
        with ThreadPoolExecutor(len(parts)) as ex:
            list(ex.map(lambda t: run(*t), parts))
    sse = np.zeros(L, dtype=np.float64)
    for _, s in sorted(sses, key=lambda t: t[0]):
        sse += s
    out = {"idx": idx, "latent": lat, "xq": xq, "sse": sse}
    if want_margin:
        out["margin"], out["scale"] = margin, scale
    return out


def code_stats(idx_col, resid, K):
    """SURVEY a9/a11 segmented reduce: counts [K] and sums [K, e] in item order."""
    idx_col = np.ascontiguousarray(idx_col, dtype=np.int64)
    resid = _f32(resid)
    n, e = resid.shape
    count = np.empty(K, dtype=np.float32)
    s = np.empty((K, e), dtype=np.float32)
    rc = lib().lcrec_oracle_code_stats(_p(idx_col, _i64p), 1, _p(resid), n, e, K, _p(count), _p(s))
    assert rc == 0, rc
    return count, s


def ema_update(ema_count, ema_sum, codebook, count, s, decay, eps):
    """index_improve/models/vq.py:155-184 in place on copies; returns the three updated arrays."""
    ema_count, ema_sum, codebook = _f32(ema_count).copy(), _f32(ema_sum).copy(), _f32(codebook).copy()
    K, e = codebook.shape
    alpha = np.float32(1 - decay)
    keep = np.float32(1 - (1 - decay))
    rc = lib().lcrec_oracle_ema_update(_p(ema_count), _p(ema_sum), _p(codebook), _p(_f32(count)), _p(_f32(s)), K, e,
                                       np.float32(decay), alpha, keep, np.float32(eps))
    assert rc == 0, rc
    return ema_count, ema_sum, codebook


def distances(z, codebook):
    """vq.py:71-73 as a [n, K] fp32 matrix in the canonical fma-chain order."""
    z, cb = _f32(z), _f32(codebook)
    n, e = z.shape
    K = cb.shape[0]
    d = np.empty((n, K), dtype=np.float32)
    fn = lib().lcrec_oracle_distances
    fn.restype = ctypes.c_int
    fn.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
    rc = fn(_p(z), n, e, _p(cb), K, _p(d))
    assert rc == 0, rc
    return d


def kmeans_lloyd(x, init, iters, tol=1e-4):
    """Lloyd iterations from given centres -- the checker for lcrec_amd.layers.kmeans_device (which
    replaces the sklearn call of the reference's index/models/layers.py:69-82; that call itself is not
    bit-pinned).  Nearest centre by the canonical distance (rq_assign, one level), per-cluster sums in
    item order (code_stats), centre = sum / count in fp32, empty clusters keep their centre; stop when the
    summed squared shift <= tol * mean feature variance.  Returns (centres, iterations run)."""
    x = _f32(x)
    c = _f32(init).copy()
    K = c.shape[0]
    limit = np.float32(tol) * x.var(axis=0, dtype=np.float32).mean(dtype=np.float32)
    done = 0
    for _ in range(int(iters)):
        idx = rq_assign(x, [c])["idx"][:, 0]
        count, total = code_stats(idx, x, K)
        moved = np.where(count[:, None] > 0, total / np.maximum(count[:, None], np.float32(1.0)), c).astype(np.float32)
        shift = ((moved - c) ** 2).sum(dtype=np.float32)
        c = moved
        done += 1
        if shift <= limit:
            break
    return c, done


def linear_backward(gy, x, W, splits=1, threads=1):
    """(gx, gw) of y = x W^T for gy = dL/dy -- what autograd derives for nn.Linear (reference
    index/models/layers.py:23 under loss.backward(), index/trainer.py:117), in the arithmetic of
    lcrec_linear_backward (include/lcrec.h): gx = gy W as one fma chain per element over out_dim;
    gw = gy^T x as `splits` runs of 32*ceil(ceil(n/32)/splits) consecutive items, each run one fma chain
    from 0, the runs added in order."""
    gy, x, W = _f32(gy), _f32(x), _f32(W)
    n = gy.shape[0]
    gx = linear(gy, np.ascontiguousarray(W.T), threads=threads)
    per = -(-(-(-n // 32)) // splits) * 32
    gw = None
    for lo in range(0, n, per):
        part = linear(np.ascontiguousarray(gy[lo:lo + per].T), np.ascontiguousarray(x[lo:lo + per].T), threads=threads)
        gw = part if gw is None else (gw + part).astype(np.float32)
    return gx, gw
