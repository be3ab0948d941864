"""Index emission + conflict resolution -- the flow of the reference's index/generate_indices.py
(:51-145), as functions and a small CLI instead of a script with hard-coded paths (:44-49).

    python -m lcrec_amd.generate_indices --ckpt_path CKPT.pth --output_dir DIR [--dataset Games]

Same observable behaviour: pass 1 = get_indices(use_sk=False) over all items (:77-95); levels
0..L-2 forced to hard assignment and the last level's sk_epsilon set to 0.003 if it was 0 (:101-105);
up to 20 rounds in which every group of items sharing a tuple is re-assigned with Sinkhorn on the
last level (:107-128); statistics (:131-136); `{item: ["<a_i>", "<b_j>", ...]}` written with
json.dump's default separators (:138-145) so data.py:38-89 reads it unchanged.

What runs where: pass 1 is one lcrec_encode_assign call; the residual entering the last level is
kept in HBM, so a round is {lcrec_collision_groups -> gather -> ONE batched lcrec_sinkhorn_assign
over all groups -> scatter}: no re-encoding, no per-group forward, no Python string keys.
Groups within a round are independent in the reference too (they are listed before any is
re-assigned), so batching them does not change the result.

Documented divergences from the reference script:
  * its 5-entry prefix list (:83) raises IndexError for L > 5; here prefixes continue <f_..>, <g_..>;
  * it stores tokens in fixed-width numpy unicode arrays (:98-99) which silently truncate a
    replacement longer than anything seen in pass 1; integer tuples are kept here, and a warning is
    logged if a run ever hits that case (the reference's output would be corrupt there).
"""
import argparse
import collections
import logging
import os

import numpy as np
import torch

from . import ops
from .datasets import EmbDataset
from .rqvae import RQVAE

PREFIX = ["<{}_{{}}>".format(chr(ord("a") + i)) for i in range(26)]   # "<a_{}>", "<b_{}>", ... (:83 has a..e)
MAX_ROUNDS = 20                                                        # :110
log = logging.getLogger(__name__)


# ---- the reference's helper API (:18-42), on any sequence of hashable index keys
def check_collision(all_indices_str):
    keys = all_indices_str.tolist() if hasattr(all_indices_str, "tolist") else list(all_indices_str)
    keys = [tuple(k) if isinstance(k, list) else k for k in keys]
    return len(keys) == len(set(keys))


def get_indices_count(all_indices_str):
    counts = collections.defaultdict(int)
    for key in all_indices_str:
        counts[tuple(key) if isinstance(key, list) else key] += 1
    return counts


def get_collision_item(all_indices_str):
    """Groups of item ids sharing a key: first-occurrence order, ids ascending (:29-42)."""
    seen = {}
    for i, key in enumerate(all_indices_str):
        seen.setdefault(tuple(key) if isinstance(key, list) else key, []).append(i)
    return [ids for ids in seen.values() if len(ids) > 1]


# Globals a trainer checkpoint ({args, epoch, best_*, state_dict, optimizer}, trainer.py:158-166) needs beyond
# plain containers and numbers: tensor rebuild helpers, typed storages, OrderedDict, and argparse.Namespace for `args`.
_CKPT_GLOBALS = {("collections", "OrderedDict"), ("argparse", "Namespace"),
                 ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
                 ("torch", "Size"), ("torch.storage", "UntypedStorage")} | \
                {("torch", t + "Storage") for t in ("Float", "Double", "Half", "BFloat16", "Long", "Int", "Short",
                                                    "Char", "Byte", "Bool")}


class _CheckpointPickle:
    """`pickle_module` for torch.load: the real unpickler (so protocol 4's FRAME opcode, which
    trainer.py:167 makes every checkpoint carry and torch's own weights_only unpickler rejects, is read)
    with find_class restricted to _CKPT_GLOBALS -- any other global raises instead of being imported."""
    import pickle as _pickle
    __name__ = "lcrec_amd.checkpoint_pickle"
    UnpicklingError = _pickle.UnpicklingError

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) in _CKPT_GLOBALS:
                return super().find_class(module, name)
            import pickle
            raise pickle.UnpicklingError(f"checkpoint names the global {module}.{name}, which is not on the "
                                         "allow-list of a trainer checkpoint (use --trust_checkpoint for your own files)")

    @classmethod
    def load(cls, fh, **kw):
        return cls.Unpickler(fh, **kw).load()

    @classmethod
    def loads(cls, data, **kw):
        import io
        return cls.Unpickler(io.BytesIO(data), **kw).load()


def load_checkpoint(ckpt_path, trust=False):
    """torch.load of a trainer checkpoint with a restricted unpickler: tensors, plain containers and
    argparse.Namespace (the `args` entry), nothing else -- a file that names any other global is refused
    with pickle.UnpicklingError, never executed.  `trust=True` (CLI: --trust_checkpoint) loads the way the
    reference does (generate_indices.py:51, full pickle): only for a file you wrote yourself."""
    if trust:
        log.warning("loading %s with the unrestricted unpickler (--trust_checkpoint)", ckpt_path)
        return torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=False)
    return torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=False, pickle_module=_CheckpointPickle)


def build_model_from_args(args, in_dim):
    """:58-70 -- note beta is NOT forwarded (RQVAE's default 0.25 applies; irrelevant in eval)."""
    return RQVAE(in_dim=in_dim, num_emb_list=args.num_emb_list, e_dim=args.e_dim, layers=args.layers,
                 dropout_prob=args.dropout_prob, bn=args.bn, loss_type=args.loss_type,
                 quant_loss_weight=args.quant_loss_weight, kmeans_init=args.kmeans_init,
                 kmeans_iters=args.kmeans_iters, sk_epsilons=args.sk_epsilons, sk_iters=args.sk_iters,
                 ema_decay=getattr(args, "ema_decay", None), epsilon=getattr(args, "epsilon", 1e-5),
                 reset_threshold=getattr(args, "reset_threshold", 1e-5),
                 reset_interval=getattr(args, "reset_interval", 1000))


@torch.no_grad()
def assign_all(model, data, chunk_rows=1 << 20, audit=None):
    """Pass 1 (:77-95): int64 [N, L] hard indices plus the residual entering the last level.
    audit: optional dict; receives "neartie" (int32 [N], ops.NEARTIE_TAU) -- the near-tie flags of pass 1."""
    levels = list(model.rq.vq_layers)
    Ws, bs, scs, shs = model.encoder.folded()
    cbs = [q.embedding.weight.detach() for q in levels]
    flat, ks = ops.flatten_codebooks(cbs)
    idx_parts, last_parts, tie_parts = [], [], []
    for lo in range(0, data.shape[0], chunk_rows):
        x = data[lo:lo + chunk_rows]
        a = {} if audit is not None else None
        idx, latent, _, _ = ops.encode_assign(x, Ws, bs, flat, ks, scs, shs, want_latent=True, audit=a,
                                              tie_tau=ops.NEARTIE_TAU)
        idx_parts.append(idx)
        if a is not None:
            tie_parts.append(a["neartie"])
        if len(levels) > 1:
            pflat, pks = ops.flatten_codebooks(cbs[:-1])
            _, _, _, resid = ops.rq_assign(latent, pflat, pks, want_resid=True)
            last_parts.append(resid[len(levels) - 1].clone())
        else:
            last_parts.append(latent)
    if audit is not None:
        audit["neartie"] = torch.cat(tie_parts) if tie_parts else torch.zeros(0, dtype=torch.int32, device=data.device)
    return torch.cat(idx_parts), torch.cat(last_parts), ks


@torch.no_grad()
def resolve_collisions(model, idx, resid_last, ks, max_rounds=MAX_ROUNDS, on_round=None, ctx=None):
    """:101-128.  Mutates and returns idx; also returns the number of groups seen in each round.

    With a distributed context (torchrun) a round's groups are SHARDED over the ranks: every
    rank lists the groups (a device sort of the full index matrix, identical everywhere), solves a contiguous run of
    them holding about 1/world of the colliding items, and one all-gather of the new last-level codes -- in rank order,
    which is group order -- updates every rank's copy.  Groups are independent within a round, so the result is the
    single-process one."""
    levels = list(model.rq.vq_layers)
    for q in levels[:-1]:
        q.sk_epsilon = 0.0
    if levels[-1].sk_epsilon == 0.0:
        levels[-1].sk_epsilon = 0.003
    last = levels[-1]
    cb_last = last.embedding.weight.detach().contiguous()
    L = len(levels)
    history = []
    sharded = ctx is not None and ctx.enabled
    for _ in range(max_rounds):
        found = ops.collision_groups(idx, ks, want_groups="device")
        if found["n_groups"] == 0:
            break
        history.append(found["n_groups"])
        if on_round is not None:
            on_round(len(history) - 1, found["n_groups"])
        members = found["members"]
        offs = found["offsets"].cpu().numpy()
        if sharded and found["n_groups"] >= ctx.world_size:
            # group boundaries nearest to equal shares of the colliding items
            cuts = np.searchsorted(offs, [offs[-1] * r / ctx.world_size for r in range(1, ctx.world_size)], side="left")
            bounds = np.concatenate([[0], np.minimum(cuts, found["n_groups"]), [found["n_groups"]]])
            bounds = np.maximum.accumulate(bounds)
            g_lo, g_hi = int(bounds[ctx.rank]), int(bounds[ctx.rank + 1])
            mine = members[int(offs[g_lo]):int(offs[g_hi])]
            if g_hi > g_lo:
                rows = resid_last.index_select(0, mine)
                new_mine = ops.sinkhorn_assign(rows, cb_last, last.sk_epsilon, last.sk_iters,
                                               group_offsets=offs[g_lo:g_hi + 1] - offs[g_lo])
            else:
                new_mine = torch.zeros(0, dtype=torch.int64, device=idx.device)
            counts = [int(offs[bounds[r + 1]] - offs[bounds[r]]) for r in range(ctx.world_size)]
            new_last = ctx.gather_rows(new_mine, counts=counts)
        else:
            rows = resid_last.index_select(0, members)
            new_last = ops.sinkhorn_assign(rows, cb_last, last.sk_epsilon, last.sk_iters, group_offsets=offs)
        idx[members, L - 1] = new_last
    return idx, history


# ---- opt-in: re-evaluate near-tie items in the reference's own operation order (--recheck_neartie) -----------------------------
def reference_order_indices(state_dict, n_layers, bn, levels, x, eps=1e-5):
    """RQVAE.get_indices(x, use_sk=False) (rqvae.py:68-72) as the reference's torch CPU op sequence on ONE batch `x` (a CPU
    float tensor): nn.Linear / BatchNorm1d(eval) / ReLU per encoder layer (layers.py:18-30,42), then per level
    d = sum(x^2) + sum(w^2)^T - 2 x w^T, argmin, gather, straight-through, residual (vq.py:71-95, rq.py:45-48).
    This is the ONE place the product computes on the host: by request, for the ~0.1 % of items whose assignment hangs on
    the last bits of d -- which bits come out depends on the BLAS under torch (MKL's blocking changes with the batch size and
    the CPU), so the call reproduces "the reference's CPU run" only on a host like the reference's; DESIGN.md section 2.1.
    Returns (indices int64 [n, L], residual entering the last level [n, e])."""
    import torch.nn.functional as F
    step = 4 if bn else 3
    h = x
    for l in range(n_layers):
        slot = l * step + 1
        h = F.linear(h, state_dict[f"encoder.mlp_layers.{slot}.weight"], state_dict[f"encoder.mlp_layers.{slot}.bias"])
        if l != n_layers - 1:
            if bn:
                b = f"encoder.mlp_layers.{slot + 1}"
                h = F.batch_norm(h, state_dict[b + ".running_mean"], state_dict[b + ".running_var"], state_dict[b + ".weight"],
                                 state_dict[b + ".bias"], False, 0.1, eps)
            h = F.relu(h)
    residual, cols, before_last = h, [], h
    for l in range(levels):
        w = state_dict[f"rq.vq_layers.{l}.embedding.weight"]
        if l == levels - 1:
            before_last = residual
        d = torch.sum(residual ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1, keepdim=True).t() \
            - 2 * torch.matmul(residual, w.t())
        idx = torch.argmin(d, dim=-1)
        q = F.embedding(idx, w)
        q = residual + (q - residual).detach()
        residual = residual - q
        cols.append(idx)
    return torch.stack(cols, dim=-1), before_last


@torch.no_grad()
def recheck_neartie(state_dict, args, data, idx, resid_last, flags, batch_size=64):
    """For every item flagged by the near-tie audit, recompute its tuple the way index/generate_indices.py:77-79 computes it
    -- the reference's op sequence on the CPU, on the very 64-row batch the reference's DataLoader would have put the item in
    (shuffle=False: batch b = items 64 b .. 64 b + 63) -- and overwrite it; items whose code changed before the last level also
    get that evaluation's last-level residual (the conflict rounds re-assign from it).  `data`: EmbDataset or a host array.
    Returns (items re-evaluated, items whose tuple changed)."""
    flagged = torch.nonzero(flags != 0).flatten().cpu().numpy()
    if flagged.size == 0:
        return 0, 0
    sd = {k: v.detach().to("cpu", torch.float32) if v.dtype.is_floating_point else v.detach().cpu() for k, v in state_dict.items()}
    n_layers = len(args.layers) + 1
    levels = idx.shape[1]
    rows = data.embeddings if isinstance(data, EmbDataset) else data
    n = idx.shape[0]
    changed = 0
    new_idx, new_res = [], []
    for b in np.unique(flagged // batch_size):
        lo, hi = int(b) * batch_size, min(n, (int(b) + 1) * batch_size)
        x = torch.from_numpy(np.array(rows[lo:hi], dtype=np.float32))                 # a copy: the file may be a read-only memory map
        bi, br = reference_order_indices(sd, n_layers, bool(args.bn), levels, x)
        mine = flagged[(flagged >= lo) & (flagged < hi)] - lo
        new_idx.append(bi[mine])
        new_res.append(br[mine])
    new_idx, new_res = torch.cat(new_idx).to(idx.device), torch.cat(new_res).to(resid_last.device)
    where = torch.from_numpy(flagged).to(idx.device)
    old = idx[where]
    differ = (old != new_idx).any(1)
    changed = int(differ.sum())
    idx[where] = new_idx
    early = (old[:, :-1] != new_idx[:, :-1]).any(1) if levels > 1 else torch.zeros_like(differ)
    if bool(early.any()):
        resid_last[where[early]] = new_res[early]
    return int(flagged.size), changed


def tokens_for(idx_rows):
    """[[i, j, ...], ...] -> [["<a_i>", "<b_j>", ...], ...] (:83-92)."""
    L = len(idx_rows[0]) if idx_rows else 0
    if L > len(PREFIX):
        raise ValueError(f"{L} levels: no token prefix beyond <z_..>")
    return [[PREFIX[l].format(int(v)) for l, v in enumerate(row)] for row in idx_rows]


def dump_index_json(idx_rows, path, chunk_items=1 << 20):
    """Exactly the bytes of `json.dump({str(item): tokens}, fp)` (:138-145) for an int64 [N, L] index
    matrix (tensor, ndarray or nested list): keys are item ids in order, separators are json's defaults.
    The text is produced by the library (lcrec_index_json_format), a chunk of items at a time."""
    if isinstance(idx_rows, torch.Tensor):
        idx_rows = idx_rows.detach().cpu().numpy()
    rows = np.asarray(idx_rows, dtype=np.int64)
    if rows.ndim != 2:
        rows = rows.reshape(len(rows), -1)
    if rows.shape[1] > len(PREFIX):
        raise ValueError(f"{rows.shape[1]} levels: no token prefix beyond <z_..>")
    with open(path, "wb") as fp:
        fp.write(b"{")
        for lo in range(0, rows.shape[0], chunk_items):
            if lo:
                fp.write(b", ")
            fp.write(ops.index_json_text(rows[lo:lo + chunk_items], first_item=lo))
        fp.write(b"}")


def _digits(t):
    """decimal digit count of non-negative int64 values, elementwise"""
    d = torch.ones_like(t)
    p = 10
    for _ in range(18):
        d += (t >= p).to(t.dtype)
        p *= 10
    return d


def _warn_if_reference_would_truncate(first_pass, final):
    def widths(t):
        if t.numel() == 0:
            return 0, 0
        d = _digits(t.clamp_min(0))
        return int(d.max()), int(d.sum(1).max())
    tok0, sum0 = widths(first_pass)
    tok1, sum1 = widths(final)
    if tok1 > tok0 or sum1 > sum0:
        log.warning("a re-assigned index is wider than anything in pass 1: the reference's fixed-width numpy "
                    "string arrays (generate_indices.py:98-99) would truncate it; this output keeps it intact")


def sharded_assign(ctx, data, assign_fn, device):
    """Pass 1 over this rank's contiguous item range, then ONE gather of (index rows, last-level
    residuals) in rank order: afterwards every rank holds what a single process would have computed
    (items are independent in pass 1, so the shard boundaries do not show in the result).
    `assign_fn(x) -> (idx [n, L], resid_last [n, e], ks)`; `data` is an EmbDataset or a tensor."""
    from . import dist as ldist
    n = len(data)
    lo, hi = ldist.shard_range(n, ctx.rank, ctx.world_size)
    x = data.to_device(device, rows=(lo, hi)) if isinstance(data, EmbDataset) else data[lo:hi].to(device)
    idx, resid_last, ks = assign_fn(x)
    return ctx.gather_rows(idx), ctx.gather_rows(resid_last), ks


def generate(ckpt_path, output_file, device="cuda:0", data_path=None, verbose=True, ctx=None, trust_checkpoint=False,
             recheck=False):
    """Whole flow of generate_indices.py:51-145.  Returns a dict of the statistics it prints.
    recheck: re-evaluate the near-tie items of pass 1 in the reference's CPU operation order (recheck_neartie).

    Under torchrun (ctx = dist.init_from_env()) pass 1 is item-sharded over the ranks and each conflict round's
    groups are sharded too (resolve_collisions); rank 0 writes the file."""
    from . import dist as ldist
    ctx = ctx or ldist.current()
    lead = ctx.rank == 0
    verbose = verbose and lead
    ckpt = load_checkpoint(ckpt_path, trust=trust_checkpoint)
    args = ckpt["args"]
    data = EmbDataset(data_path or args.data_path, mmap=ctx.enabled or str(device).startswith("cuda"))
    model = build_model_from_args(args, data.dim)
    model.load_state_dict(ckpt["state_dict"])
    model = model.to(torch.device(device)).eval()
    if verbose:
        print(model)
    audit = {}
    idx, resid_last, ks = sharded_assign(ctx, data, lambda x: assign_all(model, x, audit=audit), device)
    first_pass = idx.clone()
    # near-tie audit of pass 1: items whose two best codes at some level are closer than the rounding noise of
    # vq.py:71-73 -- the only ones a CPU run of the reference could index differently (ops.NEARTIE_TAU)
    neartie_items = int(ctx.sum_int(int((audit["neartie"] != 0).sum())))
    rechecked = (0, 0)
    if recheck:
        flags = ctx.gather_rows(audit["neartie"]) if ctx.enabled else audit["neartie"]
        if lead:
            rechecked = recheck_neartie(ckpt["state_dict"], args, data, idx, resid_last, flags)
        if ctx.enabled:                                    # rank 0's host decides (ranks on other CPUs could round differently)
            ctx.broadcast_(idx)
            ctx.broadcast_(resid_last)
        if lead:
            log.info("--recheck_neartie: %d near-tie items re-evaluated with the reference's torch CPU ops on their batch-64 "
                     "neighbours, %d tuples changed", *rechecked)
        first_pass = idx.clone()

    def show(round_no, n_groups):
        if verbose:
            print(n_groups)

    idx, history = resolve_collisions(model, idx, resid_last, ks, on_round=show, ctx=ctx)
    _warn_if_reference_would_truncate(first_pass, idx)
    final = ops.collision_groups(idx, ks, want_groups=False)
    n = idx.shape[0]
    stats = {"items": n, "max_conflicts": final["max_count"], "collision_rate": (n - final["unique"]) / n if n else 0.0,
             "rounds": len(history), "groups_per_round": history, "neartie_items": neartie_items,
             "neartie_tau": ops.NEARTIE_TAU, "rechecked_items": rechecked[0], "recheck_changed": rechecked[1]}
    if lead:
        log.info("near-tie items in pass 1: %d of %d (top-2 code gap <= %.3g x distance magnitude at some level); only "
                 "these could receive a different tuple from a CPU run of the reference", neartie_items, n, ops.NEARTIE_TAU)
    if verbose:
        print("All indices number: ", n)
        print("Max number of conflicts: ", stats["max_conflicts"])
        print("Collision Rate", stats["collision_rate"])
    if lead:
        os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
        dump_index_json(idx, output_file)
    ctx.barrier()
    return stats


def main(argv=None):
    ap = argparse.ArgumentParser(description="Generate <dataset>.index.json from an RQ-VAE checkpoint")
    ap.add_argument("--dataset", type=str, default="Games")
    ap.add_argument("--ckpt_path", type=str, required=True)
    ap.add_argument("--output_dir", type=str, required=True)
    ap.add_argument("--data_path", type=str, default=None, help="override the data path stored in the checkpoint")
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--recheck_neartie", action="store_true",
                    help="re-evaluate the ~0.1 %% of items flagged by the near-tie audit with the reference's own torch CPU op "
                         "sequence on the 64-row batches generate_indices.py:77-79 forms, and use those tuples (the only items a "
                         "CPU run of the reference can index differently; the outcome depends on the host's BLAS)")
    ap.add_argument("--trust_checkpoint", action="store_true",
                    help="load the checkpoint with the unrestricted unpickler (executes code from the file)")
    a = ap.parse_args(argv)
    from . import dist as ldist
    ctx = ldist.init_from_env(a)                       # torchrun: one rank per GPU; plain python: inert
    out = os.path.join(a.output_dir, f"{a.dataset}.index.json")
    try:
        return generate(a.ckpt_path, out, device=a.device, data_path=a.data_path, ctx=ctx,
                        trust_checkpoint=a.trust_checkpoint, recheck=a.recheck_neartie)
    finally:
        ldist.shutdown(ctx)


if __name__ == "__main__":
    main()
