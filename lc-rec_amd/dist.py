"""Item-sharded data parallelism for the indexing path: one process per GPU, torch.distributed
over RCCL/xGMI ("nccl" backend on ROCm), gloo for CPU rehearsal.

The reference's index/ stage is single-process (no collective on this path);
this module is new capability.  What is exchanged, and why it reproduces the single-process step on
the concatenated batch:

  encode+assign / index generation   no collective: contiguous item ranges per rank, weights and
                                     codebooks replicated; index rows are gathered once at the end.
  training step
    gradients      every rank back-propagates loss_r * n_r / N (n_r its rows, N the global batch), so the SUM of the
                   ranks' gradients is the gradient of the global-batch mean loss -- also through the batch
                   statistics of BatchNorm, whose backward sums over ranks (GradReducer below).  The parameters'
                   .grad are views of ONE persistent flat buffer; it is all-reduced in buckets, each launched
                   asynchronously from a backward hook the moment its last gradient has been accumulated, so the
                   decoder's buckets travel while the encoder is still being differentiated.  No per-step
                   allocation, no copy-back.  Codebook gradients are covered too: they are linear in the
                   per-code (count, sum) statistics.
    EMA statistics (count, sum) per level are all-reduced before the EMA update (improve fork).
    Sinkhorn level residual rows are all-gathered and every rank solves the global B x K problem
                   redundantly, keeping its slice: 1 collective instead of ~100 latency-bound ones.
    BatchNorm      statistics of the global batch (SyncBatchNorm semantics) by the library's own kernels with one
                   all-reduce per layer and direction (layers._BatchNormAct).
    k-means init   rank 0 runs sklearn on the gathered first batch and broadcasts the centres.
    NaN check      made on the all-reduced global loss, so every rank raises in the same step.

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
        """Gradient of the global-batch mean loss from per-rank mean-loss gradients: weighted all-reduce of whatever
        .grad tensors exist, in place.  One-shot form (tests, callers without a GradReducer): packs into a temporary."""
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

    def set_batch(self, n_local, n_global):
        """Rows of this rank / of all ranks in the current training batch (the loader knows both without a collective)."""
        self.batch_rows = (int(n_local), int(n_global))

    def row_counts(self, n_local):
        """Rows of every rank in the current batch: from set_batch (the loader's split, no collective) when it describes
        this tensor, else by an all-reduce."""
        rows = getattr(self, "batch_rows", None)
        if rows is not None and rows[0] == n_local:
            counts = batch_counts(rows[1], self.world_size)
            if counts[self.rank] == n_local:
                return counts
        return None

    def exchange_rows(self, row):
        """[world, len(row)]: every rank's copy of a small 1-d float tensor, in rank order -- ONE collective (all-gather
        over RCCL; gloo has none for device tensors, so there a zero-padded all-reduce does the same)."""
        if tdist.get_backend() == "gloo":
            buf = torch.zeros((self.world_size, row.numel()), dtype=row.dtype, device=row.device)
            buf[self.rank].copy_(row)
            tdist.all_reduce(buf, op=tdist.ReduceOp.SUM)
            return buf
        out = torch.empty((self.world_size, row.numel()), dtype=row.dtype, device=row.device)
        tdist.all_gather_into_tensor(out.view(-1), row.contiguous())
        return out

    def merge_batch_stats(self, n_local, mean, m2):
        """Per-rank BatchNorm statistics -> (mean, M2, n) of the global batch: every rank puts (n_r, mean_r, M2_r) into its
        row of a [world, 2F+1] buffer, ONE all-reduce makes all rows visible everywhere, and the rows are merged by
        mean = sum n_r mean_r / N,  M2 = sum (M2_r + n_r (mean_r - mean)^2) -- identical on every rank."""
        F = mean.numel()
        buf = torch.zeros((self.world_size, 2 * F + 1), dtype=torch.float32, device=mean.device)
        row = buf[self.rank]
        row[0] = float(n_local)
        row[1:F + 1] = mean
        row[F + 1:] = m2
        tdist.all_reduce(buf, op=tdist.ReduceOp.SUM)
        n = buf[:, :1]
        rows = getattr(self, "batch_rows", None)
        n_total = rows[1] if rows is not None and rows[0] == n_local else int(n.sum().item())
        means, m2s = buf[:, 1:F + 1], buf[:, F + 1:]
        g_mean = (n * means).sum(0) / float(n_total)
        g_m2 = (m2s + n * (means - g_mean) ** 2).sum(0)
        return g_mean, g_m2, n_total

    def _row_counts(self, n, device):
        counts = torch.zeros(self.world_size, dtype=torch.int64, device=device)
        counts[self.rank] = n
        tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
        return [int(c) for c in counts.tolist()]

    def gather_rows(self, rows, counts=None):
        """Concatenate every rank's [n_r, ...] rows in rank order (n_r may differ).  `counts`: the n_r of all ranks when
        the caller already knows them (saves the all-reduce that would find them out)."""
        if not self.enabled:
            return rows
        counts = [int(c) for c in counts] if counts is not None else self._row_counts(rows.shape[0], rows.device)
        width = max(counts)
        # gloo (the CPU rehearsal backend) has no all_gather for device tensors: stage through the host there
        via_host = rows.is_cuda and tdist.get_backend() == "gloo"
        work = torch.device("cpu") if via_host else rows.device
        if rows.shape[0] == width:
            pad = rows.contiguous().to(work)
        else:
            pad = torch.zeros((width,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=work)
            pad[:rows.shape[0]] = rows
        if not via_host and tdist.get_backend() != "gloo":
            flat = torch.empty((self.world_size * width,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=work)
            tdist.all_gather_into_tensor(flat, pad)                  # one collective, no per-rank output tensors
            if min(counts) == width:
                return flat
            return torch.cat([flat[r * width:r * width + c] for r, c in enumerate(counts)])
        out = [torch.empty_like(pad) for _ in range(self.world_size)]
        tdist.all_gather(out, pad)
        return torch.cat([o[:c] for o, c in zip(out, counts)]).to(rows.device)

    def gather_rows_with_slice(self, rows):
        """(all rows in rank order, (lo, hi) of this rank's rows inside them)."""
        if not self.enabled:
            return rows, (0, rows.shape[0])
        counts = self.row_counts(rows.shape[0]) or self._row_counts(rows.shape[0], rows.device)
        allrows = self.gather_rows(rows, counts)
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

    def all_reduce_(self, tensor, async_op=False):
        """In-place SUM over the ranks of one contiguous tensor; the work handle when async_op."""
        if self.enabled:
            return tdist.all_reduce(tensor, op=tdist.ReduceOp.SUM, async_op=async_op)
        return None

    def broadcast_(self, tensor, src=0):
        if self.enabled:
            tdist.broadcast(tensor, src=src)
        return tensor

    def barrier(self):
        if self.enabled:
            tdist.barrier()


class GradReducer:
    """Bucketed, overlapped gradient all-reduce over a persistent flat buffer.

    Parameters' .grad become views of `flat`; `begin()` zeroes it (instead of optimizer.zero_grad, which would detach the
    views); a post-accumulate hook on every parameter counts its bucket down and launches that bucket's all-reduce
    (async) when the last gradient of the bucket has landed -- autograd produces them decoder-first, i.e. from the end
    of the buffer towards its start; `finish()` waits for the outstanding buckets.  Ranks back-propagate a loss already
    weighted by n_r / N, so the reduction is a plain SUM and nothing is rescaled afterwards."""

    def __init__(self, ctx, params, bucket_bytes=8 << 20):
        self.ctx = ctx
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []                        # [lo, hi) element ranges, in parameter order
        self.bucket_of = {}
        lo, members = 0, []
        limit = max(1, bucket_bytes // 4)
        for i, (p, off) in enumerate(zip(self.params, offs)):
            members.append(p)
            end = off + (p.numel() + 63) // 64 * 64
            if end - lo >= limit or i == len(self.params) - 1:
                for q in members:
                    self.bucket_of[q] = len(self.buckets)
                self.buckets.append((lo, end, len(members)))
                lo, members = end, []
        for p, off in zip(self.params, offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            p.register_post_accumulate_grad_hook(self._on_grad)
        self._pending = [0] * len(self.buckets)
        self._works = []
        self.launched = 0

    def begin(self):
        self.flat.zero_()
        self._pending = [b[2] for b in self.buckets]
        self._works = []

    def _on_grad(self, p):
        b = self.bucket_of.get(p)
        if b is None or not self._pending:
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            lo, hi, _ = self.buckets[b]
            self._works.append(tdist.all_reduce(self.flat[lo:hi], op=tdist.ReduceOp.SUM, async_op=True))
            self.launched += 1

    def finish(self):
        for b, left in enumerate(self._pending):          # parameters that received no gradient this step: still reduce
            if left > 0:
                lo, hi, _ = self.buckets[b]
                self._works.append(tdist.all_reduce(self.flat[lo:hi], op=tdist.ReduceOp.SUM, async_op=True))
        for w in self._works:
            w.wait()
        self._works, self._pending = [], []


_CTX = DistContext()


def current():
    return _CTX


def batch_counts(m, world_size):
    """Rows per rank when a batch of m rows is dealt to the ranks as evenly as contiguous slices allow: the first m % W
    ranks hold one row more (DeviceLoader; 475 rows on 8 ranks = 60,60,60,59,59,59,59,59)."""
    base, extra = divmod(int(m), int(world_size))
    return [base + (1 if r < extra else 0) for r in range(world_size)]


def batch_slice(m, rank, world_size):
    counts = batch_counts(m, world_size)
    lo = sum(counts[:rank])
    return lo, lo + counts[rank]


def shard_range(n, rank, world_size):
    """Contiguous item range of `rank` when n items are split over world_size ranks."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def init_from_env(args=None, backend=None, force=False):
    """Join the job torchrun started (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); single process otherwise.
    force: build the process group even for WORLD_SIZE=1 (a one-rank RCCL group: every collective of the data-parallel
    path then really goes through the backend -- how the one-GPU test box exercises RCCL)."""
    global _CTX
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force:
        _CTX = DistContext(device=getattr(args, "device", None))
        return _CTX
    rank = int(os.environ.get("RANK", "0"))
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


def adopt(device):
    """A DistContext over a process group somebody else has already initialised (bench.py): no environment reading."""
    global _CTX
    if not tdist.is_initialized():
        _CTX = DistContext(device=device)
        return _CTX
    _CTX = DistContext(rank=tdist.get_rank(), world_size=tdist.get_world_size(), device=torch.device(device), enabled=True)
    return _CTX


def release():
    """Forget the current context (the process group itself is the owner's to destroy)."""
    global _CTX
    _CTX = DistContext()


def attach(trainer, ctx):
    """Make a Trainer data-parallel: identical initial weights, one checkpoint directory (rank 0's), and the bucketed
    gradient all-reduce.  BatchNorm needs no module surgery: layers._BatchNormAct takes its statistics over the global
    batch whenever a distributed context is active."""
    if not ctx.enabled:
        return trainer
    trainer.dist = ctx
    model = trainer.model
    for t in list(model.parameters()) + list(model.buffers()):
        tdist.broadcast(t.data, src=0)
    # Trainer.__init__ made a time-stamped directory on every rank (trainer.py:37-38 names it by wall clock): keep rank 0's
    names = [trainer.ckpt_dir]
    tdist.broadcast_object_list(names, src=0)
    if ctx.rank != 0 and names[0] != trainer.ckpt_dir:
        try:
            os.rmdir(trainer.ckpt_dir)
        except OSError:
            pass
    trainer.ckpt_dir = names[0]
    # the gradient exchange belongs to whichever step implementation the trainer picks in its first epoch: the captured
    # engine all-reduces its own flat buffer (engine.py); the autograd path builds a GradReducer then (Trainer._reducer)
    return trainer


def shutdown(ctx):
    if ctx.enabled and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
