"""CPU, world_size 2, gloo: the multi-GPU plumbing of lc-rec_amd/dist.py (what runs over RCCL on
the 8-GPU node) -- sharding, gradient all-reduce weighting, row gathers, loader slices."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import argparse
    from lcrec_amd import dist as ldist
    from lcrec_amd.datasets import DeviceLoader
    args = argparse.Namespace(device="cpu")
    ctx = ldist.init_from_env(args, backend="gloo")
    assert ctx.enabled and ctx.rank == rank and ctx.world_size == world and ldist.current() is ctx

    # ---- gradient all-reduce = gradient of the global-batch mean loss, also with uneven shards
    torch.manual_seed(0)
    model = torch.nn.Linear(6, 3)
    x = torch.randn(11, 6)
    y = torch.randn(11, 3)
    ref = torch.nn.Linear(6, 3)
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.mse_loss(ref(x), y).backward()
    lo, hi = (0, 7) if rank == 0 else (7, 11)
    torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi]).backward()
    ctx.reduce_gradients(model, n_local=hi - lo)
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)

    # ---- the training path's form of the same thing: persistent flat gradient buffer (.grad are views of it), buckets
    # all-reduced from backward hooks as soon as their last gradient has landed, loss pre-weighted by n_r / N
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 40), torch.nn.ReLU(), torch.nn.Linear(40, 3))
    ref2 = torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 40), torch.nn.ReLU(), torch.nn.Linear(40, 3))
    ref2.load_state_dict(net.state_dict())
    torch.nn.functional.mse_loss(ref2(x), y).backward()
    red = ldist.GradReducer(ctx, list(net.parameters()), bucket_bytes=4096)     # several buckets
    assert len(red.buckets) >= 2 and all(p.grad.data_ptr() >= red.flat.data_ptr() for p in net.parameters())
    for _ in range(2):                                                          # reusable step after step, no reallocation
        ptr = red.flat.data_ptr()
        red.begin()
        (torch.nn.functional.mse_loss(net(x[lo:hi]), y[lo:hi]) * ((hi - lo) / 11)).backward()
        launched_in_backward = red.launched
        red.finish()
        assert red.flat.data_ptr() == ptr
        for p, q in zip(net.parameters(), ref2.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-7)
    assert launched_in_backward == 2 * len(red.buckets)                         # every bucket left from a hook, none from finish()

    # ---- BatchNorm statistics of the global batch from per-rank (n, mean, M2): one all-reduce, same on every rank
    t_all = torch.randn(11, 5, generator=torch.Generator().manual_seed(3)) * 3 + 7
    mine_t = t_all[lo:hi]
    ctx.set_batch(hi - lo, 11)
    g_mean, g_m2, n_tot = ctx.merge_batch_stats(hi - lo, mine_t.mean(0), ((mine_t - mine_t.mean(0)) ** 2).sum(0))
    assert n_tot == 11
    assert torch.allclose(g_mean, t_all.mean(0), rtol=1e-6) and torch.allclose(g_m2, ((t_all - t_all.mean(0)) ** 2).sum(0), rtol=1e-5)

    # ---- logged losses are global-batch means
    means = ctx.global_means(torch.tensor([2.0 if rank == 0 else 4.0, 1.0]), n_local=hi - lo)
    assert torch.allclose(means, torch.tensor([(2.0 * 7 + 4.0 * 4) / 11, 1.0], dtype=torch.float64))

    # ---- row gathers keep rank order with unequal counts
    rows = torch.arange((3 if rank == 0 else 5) * 2, dtype=torch.int64).view(-1, 2) + 100 * rank
    allrows = ctx.gather_rows(rows)
    assert allrows.shape == (8, 2) and allrows[0, 0] == 0 and allrows[3, 0] == 100
    both, (a, b) = ctx.gather_rows_with_slice(rows.float())
    assert (a, b) == ((0, 3) if rank == 0 else (3, 8)) and torch.equal(both[a:b], rows.float())

    # ---- statistics all-reduce and broadcast
    cnt = torch.full((4,), float(rank + 1))
    tot = torch.full((4, 2), float(rank + 1))
    ctx.all_reduce_sum_(cnt, tot)
    assert torch.all(cnt == 3) and torch.all(tot == 3)
    t = torch.full((3,), float(rank))
    ctx.broadcast_(t, src=0)
    assert torch.all(t == 0)

    # ---- loader: every rank walks the same global batch and keeps its contiguous slice
    data = torch.arange(50, dtype=torch.float32).view(25, 2)
    torch.manual_seed(5)
    mine = list(DeviceLoader(data, batch_size=8, shuffle=True, device="cpu", rank=rank, world_size=world))
    torch.manual_seed(5)
    full = list(DeviceLoader(data, batch_size=8, shuffle=True, device="cpu"))
    for part, whole in zip(mine, full):
        a, b = ldist.batch_slice(whole.shape[0], rank, world)
        assert torch.equal(part, whole[a:b])
    # as even as contiguous slices allow (Games' last batch: 475 rows on 8 ranks), never an empty rank unless m < world
    assert ldist.batch_counts(475, 8) == [60, 60, 60, 59, 59, 59, 59, 59] and ldist.batch_counts(3, 4) == [1, 1, 1, 0]
    assert ldist.batch_slice(475, 3, 8) == (180, 239) and ldist.batch_slice(9, 1, 2) == (5, 9)
    # the engine's exchanges: row counts from the loader's split (no collective), one-collective exchange of a small row
    ctx.set_batch(5 if rank == 0 else 4, 9)
    assert ctx.row_counts(5 if rank == 0 else 4) == [5, 4] and ctx.row_counts(3) is None
    got = ctx.exchange_rows(torch.tensor([float(rank), 2.0 * rank + 1.0, 7.0]))
    assert torch.equal(got, torch.tensor([[0.0, 1.0, 7.0], [1.0, 3.0, 7.0]]))
    red = torch.tensor([1.0 + rank, 10.0])
    ctx.all_reduce_(red)
    assert torch.equal(red, torch.tensor([3.0, 20.0]))
    assert ldist.shard_range(10, rank, world) == ((0, 5) if rank == 0 else (5, 10))
    assert ldist.shard_range(7, 1, 4) == (2, 4) and ldist.shard_range(3, 3, 4) == (3, 3)

    # ---- index generation: sharded pass 1 + one gather == the single-process pass (the per-rank
    # assignment is the CPU oracle here; on the GPU node it is lcrec_encode_assign)
    from lcrec_amd import generate_indices as gen
    from lcrec_amd.datasets import EmbDataset
    from oracle import cpu_oracle
    rs = np.random.RandomState(3)
    items = rs.standard_normal((37, 16)).astype(np.float32)
    cbs = rs.standard_normal((2, 8, 16)).astype(np.float32)

    def assign(x):
        out = cpu_oracle.rq_assign(x.numpy(), list(cbs), want_resid=True)
        return torch.from_numpy(out["idx"]), torch.from_numpy(out["resid"][1].copy()), [8, 8]

    whole_idx, whole_res, _ = assign(torch.from_numpy(items))
    got_idx, got_res, ks = gen.sharded_assign(ctx, torch.from_numpy(items), assign, "cpu")
    assert torch.equal(got_idx, whole_idx) and torch.equal(got_res, whole_res) and ks == [8, 8]
    path = os.path.join(tmp, "items.npy")
    if rank == 0:
        np.save(path, items.astype(np.float64))          # the loader casts to fp32 (datasets.py:19)
    ctx.barrier()
    got_idx2, _, _ = gen.sharded_assign(ctx, EmbDataset(path, mmap=True), assign, "cpu")
    assert torch.equal(got_idx2, whole_idx)
    ctx.barrier()
    if rank == 0:
        os.remove(path)

    with open(os.path.join(tmp, f"ok{rank}"), "w") as fh:
        fh.write("ok")
    ldist.shutdown(ctx)


def test_dist_collectives_world_size_2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_single_process_context_is_inert():
    from lcrec_amd import dist as ldist
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    ctx = ldist.init_from_env(None)
    assert not ctx.enabled and ctx.world_size == 1
    t = torch.ones(3)
    assert ctx.gather_rows(t) is t
    ctx.reduce_gradients(torch.nn.Linear(2, 2))
    ctx.barrier()
