"""Item-sharded data parallelism for the indexing path: one process per GPU, torch.distributed
over RCCL/xGMI ("nccl" backend on ROCm), gloo for CPU rehearsal.

The reference's index/ stage is single-process (no collective on this path);
this module is new capability.  What is exchanged, and why it reproduces the single-process step on
the concatenated batch:

  encode+assign / index generation   no collective: contiguous item ranges per rank, weights and
                                     codebooks replicated; index rows are gathered once at the end.
  training step
    gradients      ONE flat fp32 all-reduce per step of [grads * n_local ..., n_local]; dividing by the
                   summed n_local gives the gradient of the global-batch mean loss.  This covers the
                   codebook gradients too: they are linear in the per-code (count, sum) statistics.
    EMA statistics (count, sum) per level are all-reduced before the EMA update (improve fork).
    Sinkhorn level residual rows are all-gathered and every rank solves the global B x K problem
                   redundantly, keeping its slice: 1 collective instead of ~100 latency-bound ones.
    BatchNorm      SyncBatchNorm (batch statistics over the global batch).
    k-means init   rank 0 runs sklearn on the gathered first batch and broadcasts the centres.

Message sizes at the run.sh architecture: gradients 35 MB (768-d) / 90 MB (4096-d), bandwidth-bound
-> left to RCCL's multi-ring over the 7 xGMI links; statistics <= 1.1 MB and the Sinkhorn gather
(128 B/item) are latency-bound -> one packed buffer each.
"""
import os

import torch
import torch.distributed as tdist


class DistContext:
    def __init__(self, rank=0, world_size=1, device=None, enabled=False):
        self.rank = rank
        self.world_size = world_size
        self.device = device
        self.enabled = enabled

    # ---- collectives (all no-ops when disabled)
    def reduce_gradients(self, model, n_local=None):
        """Gradient of the global-batch mean loss from per-rank mean-loss gradients."""
        if not self.enabled:
            return
        params = [p for p in model.parameters() if p.grad is not None]
        if not params:
            return
        if n_local is None:
            n_local = getattr(self, "last_batch_rows", 1)
        w = float(n_local)
        flat = torch.cat([p.grad.reshape(-1) * w for p in params] +
                         [torch.tensor([w], dtype=params[0].grad.dtype, device=params[0].grad.device)])
        tdist.all_reduce(flat, op=tdist.ReduceOp.SUM)
        total = flat[-1]
        off = 0
        for p in params:
            k = p.grad.numel()
            p.grad.copy_((flat[off:off + k] / total).view_as(p.grad))
            off += k

    def _row_counts(self, n, device):
        counts = torch.zeros(self.world_size, dtype=torch.int64, device=device)
        counts[self.rank] = n
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
        return [int(c) for c in counts.tolist()]

    def gather_rows(self, rows):
        """Concatenate every rank's [n_r, ...] rows in rank order (n_r may differ)."""
        if not self.enabled:
            return rows
        counts = self._row_counts(rows.shape[0], rows.device)
        width = max(counts)
        # gloo (the CPU rehearsal backend) has no all_gather for device tensors: stage through the host there
        via_host = rows.is_cuda and tdist.get_backend() == "gloo"
        work = torch.device("cpu") if via_host else rows.device
        pad = torch.zeros((width,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=work)
        pad[:rows.shape[0]] = rows
        out = [torch.empty_like(pad) for _ in range(self.world_size)]
        tdist.all_gather(out, pad)
        return torch.cat([o[:c] for o, c in zip(out, counts)]).to(rows.device)

    def gather_rows_with_slice(self, rows):
        """(all rows in rank order, (lo, hi) of this rank's rows inside them)."""
        if not self.enabled:
            return rows, (0, rows.shape[0])
        counts = self._row_counts(rows.shape[0], rows.device)
        allrows = self.gather_rows(rows)
        lo = sum(counts[:self.rank])
        return allrows, (lo, lo + counts[self.rank])

    def global_means(self, values, n_local):
        """Means over the GLOBAL batch of per-rank means `values` (a 1-d tensor) taken over n_local items each --
        the loss scalars the trainer logs."""
        if not self.enabled:
            return values
        w = float(n_local)
        flat = torch.cat([values.detach().double() * w, torch.tensor([w], dtype=torch.float64, device=values.device)])
        tdist.all_reduce(flat, op=tdist.ReduceOp.SUM)
        return flat[:-1] / flat[-1]

    def sum_int(self, value):
        """Sum of a Python int over the ranks."""
        if not self.enabled:
            return int(value)
        dev = self.device if (self.device is not None and tdist.get_backend() != "gloo") else "cpu"
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
        return int(t.item())

    def all_reduce_sum_(self, *tensors):
        if not self.enabled:
            return
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
        tdist.all_reduce(flat, op=tdist.ReduceOp.SUM)
        off = 0
        for t in tensors:
            k = t.numel()
            t.copy_(flat[off:off + k].view_as(t))
            off += k

    def broadcast_(self, tensor, src=0):
        if self.enabled:
            tdist.broadcast(tensor, src=src)
        return tensor

    def barrier(self):
        if self.enabled:
            tdist.barrier()


_CTX = DistContext()


def current():
    return _CTX


def shard_range(n, rank, world_size):
    """Contiguous item range of `rank` when n items are split over world_size ranks."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def init_from_env(args=None, backend=None):
    """Join the job torchrun started (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); single process otherwise."""
    global _CTX
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        _CTX = DistContext(device=getattr(args, "device", None))
        return _CTX
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available() and (args is None or str(getattr(args, "device", "cuda")).startswith("cuda"))
    backend = backend or ("nccl" if use_gpu else "gloo")
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(device)
    if not tdist.is_initialized():
        if backend == "nccl":
            tdist.init_process_group(backend, device_id=device)
        else:
            tdist.init_process_group(backend)
    if args is not None:
        args.device = str(device)
    _CTX = DistContext(rank=rank, world_size=world, device=device, enabled=True)
    return _CTX


def attach(trainer, ctx):
    """Make a Trainer data-parallel: SyncBatchNorm, identical initial weights, gradient all-reduce."""
    if not ctx.enabled:
        return trainer
    trainer.dist = ctx
    model = trainer.model
    if getattr(model, "bn", False):
        for part in ("encoder", "decoder"):
            mlp = getattr(model, part)
            mlp.mlp_layers = torch.nn.SyncBatchNorm.convert_sync_batchnorm(mlp.mlp_layers)
    for t in list(model.parameters()) + list(model.buffers()):
        tdist.broadcast(t.data, src=0)
    return trainer


def shutdown(ctx):
    if ctx.enabled and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
