"""CPU restatement of the index-generation flow (reference index/generate_indices.py:77-145) on top
of the two oracles: canonical-order encoder/argmin/distances from lcrec_oracle.c, fp64 Sinkhorn from
torch_ref.  One forward PER GROUP, Python dict keys -- deliberately the reference's structure, not
the product's batched one.

TEST INFRASTRUCTURE ONLY (see oracle/lcrec_oracle.c header)."""
import json

import numpy as np
import torch

from . import cpu_oracle, torch_ref

PREFIX = ["<a_{}>", "<b_{}>", "<c_{}>", "<d_{}>", "<e_{}>"]      # generate_indices.py:83


def collision_groups(keys):
    """generate_indices.py:29-42."""
    seen = {}
    for i, k in enumerate(keys):
        seen.setdefault(k, []).append(i)
    return [v for v in seen.values() if len(v) > 1]


def run(x, weights, biases, codebooks, bn_scale=None, bn_shift=None, last_eps=0.003, sk_iters=50, max_rounds=20):
    """Returns (final idx [N, L] int64, groups per round, json text of the .index.json)."""
    first = cpu_oracle.encode_assign(x, weights, biases, codebooks, bn_scale, bn_shift, threads=8)
    idx = first["idx"].copy()
    L = idx.shape[1]
    # residual entering the last level (levels 0..L-2 are hard in every round, :101-103)
    if L > 1:
        resid_last = cpu_oracle.rq_assign(first["latent"], codebooks[:-1], want_resid=True)["resid"][L - 1]
    else:
        resid_last = first["latent"]
    cb_last = np.ascontiguousarray(codebooks[-1], dtype=np.float32)
    history = []
    for _ in range(max_rounds):
        groups = collision_groups([tuple(r) for r in idx.tolist()])
        if not groups:
            break
        history.append(len(groups))
        for g in groups:                              # one "forward" per group, as :113-119
            d = torch.from_numpy(cpu_oracle.distances(resid_last[g], cb_last))
            Q = torch_ref.sinkhorn(torch_ref.centre_distances(d).double(), last_eps, sk_iters)
            idx[g, L - 1] = torch.argmax(Q, dim=-1).numpy()
    tokens = {i: [PREFIX[l].format(int(v)) for l, v in enumerate(row)] for i, row in enumerate(idx.tolist())}
    return idx, history, json.dumps(tokens)


def reader_views(index):
    """What the downstream stage derives from a loaded `.index.json` (reference data.py:38-89, BaseDataset):
    get_new_tokens() = sorted set of all token strings (:43-54), get_all_items() = set of "".join(tokens)
    per item (:56-66), and the per-position token sets behind get_prefix_allowed_tokens_fn (:68-79, before
    tokenisation).  `index` is the dict json.load returns."""
    new_tokens = sorted({tok for toks in index.values() for tok in toks})
    all_items = {"".join(toks) for toks in index.values()}
    allowed = {}
    for toks in index.values():
        for i, tok in enumerate(toks):
            allowed.setdefault(i, set()).add(tok)
    return new_tokens, all_items, allowed
