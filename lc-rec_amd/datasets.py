"""Item-embedding input -- mirror of the reference's index/datasets.py (EmbDataset :6-21) plus the
device-resident loader the MI355X trainer iterates instead of a worker-process DataLoader.

Contract kept from the reference: `np.load` of an [N, d] float array (the `.emb-*-td.npy` written
by data_process/amazon_text_emb.py:101-105), `.dim`, `len()`, and `__getitem__` returning a
float32 tensor -- also for a LIST of indices (generate_indices.py:117 fancy-indexes collision
groups).
"""
import numpy as np
import torch
import torch.utils.data as data


class EmbDataset(data.Dataset):

    def __init__(self, data_path, mmap=False):
        self.data_path = data_path
        self.embeddings = np.load(data_path, mmap_mode="r" if mmap else None)
        self.dim = self.embeddings.shape[-1]
        self._device_copy = None

    def __getitem__(self, index):
        return torch.FloatTensor(np.asarray(self.embeddings[index]))

    def __len__(self):
        return len(self.embeddings)

    def to_device(self, device, chunk_rows=None, rows=None, workers=None, stages=None):
        """The matrix (or the item range rows=(lo, hi), one rank's shard) as one fp32 tensor in HBM
        (Games: 16 859 x 4096 = 276 MB; a 288 GB MI355X holds 17 M such rows).  Cast and copied in row
        chunks so a large or memory-mapped file never needs a second full host copy.

        On a GPU the chunks (~32 MB) go through a ring of pinned staging buffers: `workers` host threads
        cast/copy chunks into them side by side (numpy releases the GIL for the copy; one thread moves
        ~10 GB/s out of the page cache, the link takes five times that), the calling thread queues the H2D
        copies in order on a side stream.  Measured: tools/ingest_probe.py, DESIGN.md section 5."""
        device = torch.device(device)
        whole = rows is None
        if whole and self._device_copy is not None and self._device_copy.device == device:
            return self._device_copy
        first, last = (0, len(self)) if whole else (int(rows[0]), int(rows[1]))
        n = last - first
        out = torch.empty((n, self.dim), dtype=torch.float32, device=device)
        if chunk_rows is None:
            chunk_rows = max(1, (32 << 20) // (4 * max(self.dim, 1)))
        if device.type != "cuda":
            for lo in range(0, n, chunk_rows):
                hi = min(n, lo + chunk_rows)
                out[lo:hi].copy_(torch.from_numpy(np.ascontiguousarray(self.embeddings[first + lo:first + hi],
                                                                       dtype=np.float32)))
        elif n > 0:
            import concurrent.futures as cf
            import os
            step = min(chunk_rows, n)
            chunks = [(lo, min(n, lo + step)) for lo in range(0, n, step)]
            workers = max(1, min(workers or min(8, os.cpu_count() or 1), len(chunks)))
            ring = max(2, min(stages or 2 * workers, len(chunks)))
            stage = [torch.empty((step, self.dim), dtype=torch.float32, pin_memory=True) for _ in range(ring)]
            done = [torch.cuda.Event() for _ in range(ring)]
            copier = torch.cuda.Stream(device)
            src = self.embeddings

            def fill(i, wait):
                lo, hi = chunks[i]
                if wait:
                    done[i % ring].synchronize()         # the H2D copy that last read this buffer has finished
                np.copyto(stage[i % ring][:hi - lo].numpy(), src[first + lo:first + hi], casting="unsafe")

            try:
                with cf.ThreadPoolExecutor(max_workers=workers) as pool:
                    fills = {i: pool.submit(fill, i, False) for i in range(min(ring, len(chunks)))}
                    for i, (lo, hi) in enumerate(chunks):
                        fills.pop(i).result()
                        with torch.cuda.stream(copier):
                            out[lo:hi].copy_(stage[i % ring][:hi - lo], non_blocking=True)
                            done[i % ring].record(copier)
                        if i + ring < len(chunks):       # its buffer's event is recorded now: safe to hand to a worker
                            fills[i + ring] = pool.submit(fill, i + ring, True)
            finally:
                copier.synchronize()                     # (also on an error: no copy may still read a staging buffer when it is freed)
            torch.cuda.current_stream(device).wait_stream(copier)
        if whole:
            self._device_copy = out
        return out


class DeviceLoader:
    """Batches of a device-resident embedding matrix, with the reference DataLoader's semantics
    (index/main.py:86: shuffle=True, drop_last=False; generate_indices.py:77: shuffle=False).

    Shuffling reproduces torch's RandomSampler draw sequence from the global CPU generator -- one
    int64 for the loader iterator's base seed, one for the sampler seed, then
    torch.randperm(n, generator=Generator().manual_seed(seed)) -- so under `torch.manual_seed(s)`
    the batch composition is the one the reference's DataLoader yields.  No worker processes, no
    per-item Python: a batch is one index_select on the GPU.
    """

    def __init__(self, dataset, batch_size, shuffle, device, rank=0, world_size=1):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self.data = dataset.to_device(self.device) if isinstance(dataset, EmbDataset) else dataset.to(self.device)

    def __len__(self):
        return (self.data.shape[0] + self.batch_size - 1) // self.batch_size

    def _check_same_seed(self, seed):
        """Data parallel: every rank must walk the same permutation (the global batches are cut from it).  The seeds come
        from each process's global CPU generator; anything that draws from it on one rank only would silently give that
        rank another order -- duplicated and missing items per global batch.  One tiny collective per epoch says so."""
        import torch.distributed as tdist
        if not tdist.is_initialized():
            return
        dev = self.device if tdist.get_backend() != "gloo" else "cpu"
        # int64 seeds as two exact doubles-free halves: MAX and MIN of (hi, lo) pairs must agree
        t = torch.tensor([seed, -seed], dtype=torch.int64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        if int(t[0]) != seed or int(t[1]) != -seed:
            raise RuntimeError("lcrec_amd.DeviceLoader: the ranks' shuffle seeds differ (something drew from torch's global "
                               "generator on one rank only); the epoch order would not be the single-process one")

    def __iter__(self):
        for lo, hi, order in self._walk():
            yield self.data[lo:hi] if order is None else self.data.index_select(0, order[lo:hi])

    def iter_selections(self):
        """The same walk as __iter__ without materialising the batches: (matrix, row-index vector) pairs -- the captured
        training step gathers straight into its input buffer (engine.TrainEngine.step_selected)."""
        for lo, hi, order in self._walk():
            if order is None:
                if getattr(self, "_arange", None) is None:
                    self._arange = torch.arange(self.data.shape[0], device=self.device)
                yield self.data, self._arange[lo:hi]
            else:
                yield self.data, order[lo:hi]

    def _walk(self):
        n = self.data.shape[0]
        if self.shuffle:
            torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter base seed
            seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler.__iter__
            if self.world_size > 1:
                self._check_same_seed(seed)
            g = torch.Generator()
            g.manual_seed(seed)
            order = torch.randperm(n, generator=g).to(self.device)
        else:
            order = None
        for lo in range(0, n, self.batch_size):
            hi = min(n, lo + self.batch_size)
            self.last_global_rows = hi - lo
            if self.world_size > 1:
                # item-sharded data parallel: every rank walks the same global batches and gathers only its own slice of each
                from .dist import batch_slice
                a, b = batch_slice(hi - lo, self.rank, self.world_size)
                lo, hi = lo + a, lo + b
            yield lo, hi, order
