"""Seeded input recipes shared by oracle/make_golden.py (which feeds them to the imported
reference) and the tests (which feed them to the oracle / the HIP path).

numpy's legacy RandomState streams are frozen across numpy versions, so the large inputs
(weights, codebooks, embeddings) are regenerated from seeds instead of being committed; the
golden files under tests/golden/ hold the reference's OUTPUTS (and the few inputs that came
from torch's RNG)."""
import numpy as np

RUN_SH_LAYERS = [2048, 1024, 512, 256, 128, 64]   # reference index/run.sh:15


def rs(seed):
    return np.random.RandomState(seed)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- F1: RQ known-answer test
def rq_kat(levels, codes, e=32, n=512, seed=100):
    r = rs(seed + levels * 1000 + codes)
    z = f32(r.standard_normal((n, e)))
    cbs = [f32(r.standard_normal((codes, e)) * (0.7 ** l)) for l in range(levels)]
    return z, cbs


# ---------------------------------------------------------------- F2: exact ties
def tie_case(seed=7, n=1024, K=256, e=32):
    """Small integers: every product and partial sum is exact in fp32 in ANY summation order,
    so equal distances are exactly equal and argmin's first-index rule is what is tested."""
    r = rs(seed)
    cb = f32(r.randint(-3, 4, size=(K, e)))
    cb[100] = cb[5]
    cb[200] = cb[5]
    cb[37] = cb[36]
    cb[255] = cb[0]
    z = f32(cb[r.randint(0, K, size=n)] + r.randint(-1, 2, size=(n, e)))
    z[:8] = cb[[5, 100, 200, 36, 37, 0, 255, 5]]       # exact hits on duplicated codes
    z[8] = (cb[10] + cb[11]) / 2                         # exact midpoint (halves are exact in fp32)
    return z, cb


# ---------------------------------------------------------------- F3: Sinkhorn
def sinkhorn_case(B, K=256, seed=300):
    r = rs(seed + B)
    z = f32(r.standard_normal((B, 32)))
    cb = f32(r.standard_normal((K, 32)) * 0.8)
    return z, cb


# ---------------------------------------------------------------- F8: full-size encoder
def encoder_weights(dims, seed=800, bn=False):
    """Xavier-normal-scaled weights (layers.py:33-40 uses xavier_normal_; the draw itself comes
    from numpy here), small non-zero biases, and -- for bn -- random eval-mode statistics."""
    r = rs(seed + dims[0])
    Ws, bs, bns = [], [], []
    nl = len(dims) - 1
    for l in range(nl):
        std = np.sqrt(2.0 / (dims[l] + dims[l + 1]))
        Ws.append(f32(r.standard_normal((dims[l + 1], dims[l])) * std))
        bs.append(f32(0.02 * r.standard_normal(dims[l + 1])))
        if bn and l != nl - 1:
            f = dims[l + 1]
            bns.append(dict(weight=f32(1 + 0.1 * r.standard_normal(f)), bias=f32(0.1 * r.standard_normal(f)),
                            running_mean=f32(0.1 * r.standard_normal(f)),
                            running_var=f32(0.5 + r.uniform(size=f))))
        else:
            bns.append(None)
    return Ws, bs, bns


def encoder_case(in_dim, n, levels=4, codes=256, e=32, bn=False, seed=800):
    dims = [in_dim] + RUN_SH_LAYERS + [e]
    Ws, bs, bns = encoder_weights(dims, seed, bn)
    r = rs(seed + 1 + in_dim + n)
    x = f32(r.standard_normal((n, in_dim)))
    return dims, Ws, bs, bns, x


def fold_bn(bn, eps=1e-5):
    """Eval-mode BatchNorm1d as y = t*scale + shift, computed in fp32 the way the product does."""
    if bn is None:
        return None, None
    scale = f32(bn["weight"] / np.sqrt(bn["running_var"] + np.float32(eps)))
    shift = f32(bn["bias"] - bn["running_mean"] * scale)
    return scale, shift


def state_dict_names(n_layers, bn, levels):
    """State-dict key layout of the reference RQVAE (checkpoint row)."""
    step = 4 if bn else 3
    names = {}
    for part in ("encoder", "decoder"):
        names[part] = [f"{part}.mlp_layers.{l * step + 1}" for l in range(n_layers)]
    names["bn"] = {part: [f"{part}.mlp_layers.{l * step + 2}" for l in range(n_layers - 1)]
                   for part in ("encoder", "decoder")} if bn else {}
    names["codebooks"] = [f"rq.vq_layers.{l}.embedding.weight" for l in range(levels)]
    return names


# ---------------------------------------------------------------- F6: index generation end to end
def toy_items(seed, n=3000, d=128):
    """Clustered toy embeddings with a few exact duplicates (items that can never be separated)."""
    r = rs(seed)
    centres = f32(r.standard_normal((400, d)))
    x = f32(centres[r.randint(0, 400, size=n)] + 0.08 * r.standard_normal((n, d)))
    x[17] = x[5]
    x[1200] = x[5]
    x[2999] = x[2998]
    return x


# ---------------------------------------------------------------- F9: near-tie audit at bench scale
NEARTIE_CASES = {"c3": (1_000_000, 768), "c2": (16_859, 4096)}   # BASELINE.json configs[2], configs[1]


def neartie_items(n, in_dim, seed=900, chunk=65536):
    """[n, in_dim] fp32 N(0,1) items, generated a chunk at a time (PCG64 float32 ziggurat: 768 M values in a
    few seconds).  numpy does not freeze Generator streams across versions, so the fixture stores a sha256 of
    the first chunk and the test refuses to run on a drifted stream instead of reporting false mismatches."""
    g = np.random.Generator(np.random.PCG64(seed + in_dim))
    x = np.empty((n, in_dim), dtype=np.float32)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        x[lo:lo + m] = g.standard_normal((m, in_dim), dtype=np.float32)
    return x


def neartie_encoder(in_dim, e=32, seed=900):
    dims = [in_dim] + RUN_SH_LAYERS + [e]
    Ws, bs, _ = encoder_weights(dims, seed)
    return dims, Ws, bs


# ---------------------------------------------------------------- F10: k-means init (sklearn, host)
def kmeans_case(seed=1000, n=2048, e=32, centres=300):
    """A first-training-batch-sized set of latents with cluster structure (so k-means has something to find)."""
    r = rs(seed)
    c = f32(r.standard_normal((centres, e)))
    return f32(c[r.randint(0, centres, size=n)] + 0.15 * r.standard_normal((n, e)))


# ---------------------------------------------------------------- F11: one training step at the run.sh width
def run_sh_train_case(in_dim=768, batch=1024, e=32, seed=1100):
    """The model index/run.sh trains (MLP 2048-...-64, e 32, BatchNorm -- run.sh's `--bn False` parses as True) with
    Xavier-normal-scaled weights from numpy, BatchNorm affine parameters off 1/0, fresh running statistics, and one
    batch of run.sh's size.  Returns (state dict without codebooks: name -> array, x [batch, in_dim])."""
    dims = [in_dim] + RUN_SH_LAYERS + [e]
    names = state_dict_names(len(dims) - 1, True, 4)
    sd = {}
    for part, d, s in (("encoder", dims, seed), ("decoder", dims[::-1], seed + 7)):
        Ws, bs, bns = encoder_weights(d, s, bn=True)
        for l, nme in enumerate(names[part]):
            sd[nme + ".weight"], sd[nme + ".bias"] = Ws[l], bs[l]
        for l, nme in enumerate(names["bn"][part]):
            f = d[l + 1]
            sd[nme + ".weight"], sd[nme + ".bias"] = bns[l]["weight"], bns[l]["bias"]
            sd[nme + ".running_mean"], sd[nme + ".running_var"] = np.zeros(f, np.float32), np.ones(f, np.float32)
            sd[nme + ".num_batches_tracked"] = np.zeros((), np.int64)
    x = f32(rs(seed + 1).standard_normal((batch, in_dim)))
    return sd, x


def strided_sample(a, limit=2048):
    """The entries of `a` (flattened) a fixture keeps: all of a small tensor, every stride-th of a large one."""
    flat = np.asarray(a).reshape(-1)
    step = max(1, -(-flat.size // limit))
    return flat[::step]
