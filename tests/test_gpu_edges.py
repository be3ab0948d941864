"""Edge cases on the MI355X: empty and single-row inputs, maximum and unsupported sizes, overflow,
subnormals, degenerate codebooks -- always against the CPU oracle or an explicit expectation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_empty_batches_are_a_no_op(hip):
    ops = hip.ops
    w = torch.randn(64, 32, device=DEV)
    y = ops.linear_forward(torch.empty((0, 32), device=DEV), w, torch.zeros(64, device=DEV), relu=True)
    assert tuple(y.shape) == (0, 64)
    flat, ks = ops.flatten_codebooks([torch.randn(256, 32, device=DEV)] * 2)
    idx, xq, sse, resid = ops.rq_assign(torch.empty((0, 32), device=DEV), flat, ks, want_xq=True, want_sse=True,
                                        want_resid=True)
    assert tuple(idx.shape) == (0, 2) and tuple(xq.shape) == (0, 32) and tuple(resid.shape) == (3, 0, 32)
    assert sse.tolist() == [0.0, 0.0]
    idx, _, _, _ = ops.encode_assign(torch.empty((0, 32), device=DEV), [w, torch.randn(32, 64, device=DEV)],
                                     [torch.zeros(64, device=DEV), torch.zeros(32, device=DEV)], flat, ks)
    assert tuple(idx.shape) == (0, 2)
    got = ops.collision_groups(torch.empty((0, 4), dtype=torch.int64, device=DEV), [256] * 4)
    assert got["unique"] == 0 and got["groups"] == [] and got["collision_rate"] == 0.0
    cnt, tot = ops.code_stats(torch.empty(0, dtype=torch.int64, device=DEV), torch.empty((0, 32), device=DEV), 256)
    assert float(cnt.sum()) == 0.0 and float(tot.abs().sum()) == 0.0


def test_single_rows_and_maximum_depth(hip, oracle):
    rs = np.random.RandomState(1)
    # (the last case: a codebook that fits in LDS only without the split form's hand-over buffers -- small batches then take
    # the one-tile-per-wave form)
    for n, e, Ks in ((1, 64, [512]), (3, 16, [32] * 16), (65, 32, [1024, 32, 1, 7]), (100, 32, [1088, 256])):
        z = rs.standard_normal((n, e)).astype(np.float32)
        cbs = [rs.standard_normal((K, e)).astype(np.float32) for K in Ks]
        want = oracle.rq_assign(z, cbs)
        flat, ks = hip.ops.flatten_codebooks([t(c) for c in cbs])
        idx, xq, _, _ = hip.ops.rq_assign(t(z), flat, ks, want_xq=True)
        assert np.array_equal(idx.cpu().numpy(), want["idx"]) and np.array_equal(xq.cpu().numpy(), want["xq"])


def test_unsupported_shapes_fail_loudly(hip):
    ops = hip.ops
    z = torch.randn(8, 32, device=DEV)
    with pytest.raises(hip.LcrecError, match="does not fit"):
        flat, ks = ops.flatten_codebooks([torch.randn(4096, 32, device=DEV)])
        ops.rq_assign(z, flat, ks)
    with pytest.raises(hip.LcrecError, match="e_dim=24"):
        flat, ks = ops.flatten_codebooks([torch.randn(64, 24, device=DEV)])
        ops.rq_assign(torch.randn(8, 24, device=DEV), flat, ks)
    with pytest.raises(hip.LcrecError, match="multiple of 8"):
        ops.linear_forward(torch.randn(4, 12, device=DEV), torch.randn(16, 12, device=DEV))
    with pytest.raises(hip.LcrecError):
        ops.linear_forward(torch.randn(4, 16, device=DEV, dtype=torch.float64), torch.randn(16, 16, device=DEV))
    with pytest.raises(hip.LcrecError, match="128"):
        ops.collision_groups(torch.zeros((4, 16), dtype=torch.int64, device=DEV), [1024] * 16)   # 160-bit tuples


def test_overflow_and_subnormal_inputs_match_oracle(hip, oracle):
    rs = np.random.RandomState(2)
    e, K = 32, 256
    cb = rs.standard_normal((K, e)).astype(np.float32)
    z = rs.standard_normal((256, e)).astype(np.float32)
    z[0] *= 1e30            # ||z||^2 overflows: every distance is +inf or nan-free inf-inf? -> oracle decides
    z[1] = 1e-41            # subnormal latents survive both the CPU and the MFMA chain
    z[2] = 0.0
    cb[7] = 1e-42
    want = oracle.rq_assign(z, [cb])
    flat, ks = hip.ops.flatten_codebooks([t(cb)])
    idx, xq, _, _ = hip.ops.rq_assign(t(z), flat, ks, want_xq=True)
    rows = np.array([i for i in range(256) if np.isfinite(want["xq"][i]).all()])
    assert np.array_equal(idx.cpu().numpy()[rows], want["idx"][rows])
    assert np.array_equal(xq.cpu().numpy()[rows], want["xq"][rows])
    assert 1 in rows and 2 in rows
    # Linear with subnormal weights / huge activations
    x = rs.standard_normal((130, 64)).astype(np.float32)
    x[0] *= 1e-38
    W = (rs.standard_normal((128, 64)) * 1e-3).astype(np.float32)
    W[3] = 1e-43
    b = np.zeros(128, np.float32)
    got = hip.ops.linear_forward(t(x), t(W), t(b), relu=False).cpu().numpy()
    assert np.array_equal(got, oracle.linear(x, W, b))


def test_degenerate_codebooks(hip, oracle):
    rs = np.random.RandomState(3)
    z = rs.standard_normal((500, 32)).astype(np.float32)
    same = np.tile(rs.standard_normal((1, 32)).astype(np.float32), (256, 1))      # all codes identical -> index 0
    zero = np.zeros((256, 32), np.float32)
    want = oracle.rq_assign(z, [same, zero])
    flat, ks = hip.ops.flatten_codebooks([t(same), t(zero)])
    idx, _, _, _ = hip.ops.rq_assign(t(z), flat, ks)
    assert np.array_equal(idx.cpu().numpy(), want["idx"]) and int(idx.max()) == 0


def test_sinkhorn_ragged_groups_with_empty_and_singleton(hip):
    rs = np.random.RandomState(4)
    sizes = [0, 1, 5, 0, 2, 1]
    offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
    n = offs[-1]
    z = rs.standard_normal((n, 32)).astype(np.float32)
    cb = rs.standard_normal((64, 32)).astype(np.float32)
    out = hip.ops.sinkhorn_assign(t(z), t(cb), 0.003, 50, group_offsets=offs).cpu().numpy()
    assert out.shape == (n,) and (out >= 0).all() and (out < 64).all()
    # every non-empty group equals the CPU restatement run on that group alone; a 1 x K problem is
    # degenerate (column normalisation makes all entries equal) and yields index 0, as torch's argmax does
    from oracle import cpu_oracle, torch_ref
    for g in range(len(sizes)):
        lo, hi = offs[g], offs[g + 1]
        if hi == lo:
            continue
        d = torch.from_numpy(cpu_oracle.distances(z[lo:hi], cb))
        Q = torch_ref.sinkhorn(torch_ref.centre_distances(d).double(), 0.003, 50)
        assert np.array_equal(out[lo:hi], torch.argmax(Q, -1).numpy()), g
    assert out[0] == 0


def test_collision_groups_property_random_shapes(hip):
    """lcrec_collision_groups against the reference's Python helpers (generate_indices.py:18-42) on many small
    random index matrices: level counts 1..8, code ranges 1..1024 (mixed per level), heavy and zero duplication."""
    from lcrec_amd import generate_indices as gen
    rs = np.random.RandomState(123)
    for trial in range(60):
        L = int(rs.randint(1, 9))
        Ks = [int(k) for k in rs.choice([1, 2, 3, 7, 16, 100, 256, 1024], size=L)]
        n = int(rs.choice([1, 2, 3, 17, 64, 257, 1000]))
        span = [max(1, int(k * rs.choice([1.0, 0.5, 0.05]))) for k in Ks]          # smaller span = more collisions
        rows = np.stack([rs.randint(0, s, size=n) for s in span], axis=1).astype(np.int64)
        keys = [tuple(r) for r in rows.tolist()]
        got = hip.ops.collision_groups(torch.from_numpy(rows).to("cuda:0"), Ks, want_groups=True)
        want = gen.get_collision_item(keys)
        counts = gen.get_indices_count(keys)
        assert got["groups"] == want, (trial, L, Ks, n)
        assert got["unique"] == len(counts) and got["max_count"] == max(counts.values())
        dev = hip.ops.collision_groups(torch.from_numpy(rows).to("cuda:0"), Ks, want_groups="device")
        offs, mem = dev["offsets"].tolist(), dev["members"].tolist()
        assert [mem[offs[g]:offs[g + 1]] for g in range(dev["n_groups"])] == want


def test_context_pipelines_and_lifetime(hip):
    """lcrec_context (include/lcrec.h): the second chunk pipeline and the Sinkhorn size classes on helper streams change
    scheduling, never results; contexts can be destroyed and re-created; NULL context = every launch on the caller's stream."""
    import ctypes
    from lcrec_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    dims = [256, 512, 128, 32]
    Ws = [torch.randn((dims[l + 1], dims[l]), generator=g, device=dev) * 0.05 for l in range(3)]
    bs = [0.01 * torch.randn(dims[l + 1], generator=g, device=dev) for l in range(3)]
    cbs = [torch.randn((64, 32), generator=g, device=dev) * 0.5 ** l for l in range(3)]
    flat, ks = ops.flatten_codebooks(cbs)
    x = torch.randn((3 * hip._lib.load().lcrec_encode_assign_chunk_rows() + 1000, 256), generator=g, device=dev)   # 4 chunks (the last ragged): both pipelines get work
    try:
        ops.set_pipelines(1)
        one = ops.encode_assign(x, Ws, bs, flat, ks, want_latent=True)
        ops.set_pipelines(2)
        two = ops.encode_assign(x, Ws, bs, flat, ks, want_latent=True)
        assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1])
        ops.release_contexts()                                              # drains and destroys the helper streams ...
        again = ops.encode_assign(x, Ws, bs, flat, ks, want_latent=True)    # ... and the next call makes a new context
        assert torch.equal(again[0], one[0])
    finally:
        ops.set_pipelines(1)
    # Sinkhorn over many groups: with the context (helper streams + pinned ring) == with NULL (one stream, host wait)
    lib = hip._lib.load()
    rs = np.random.RandomState(5)
    sizes = list(rs.randint(2, 40, size=300)) + [70, 300, 900] + [120] * 20
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    r = torch.randn((n, 32), generator=g, device=dev)
    cb = torch.randn((256, 32), generator=g, device=dev)
    with_ctx = ops.sinkhorn_assign(r, cb, 0.003, 50, group_offsets=offs)
    out = torch.full((n,), -1, dtype=torch.int64, device=dev)
    oarr = offs.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    nbytes = lib.lcrec_sinkhorn_assign_workspace(n, 256, oarr, len(sizes))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    rc = lib.lcrec_sinkhorn_assign(r.data_ptr(), n, 32, cb.data_ptr(), 256, oarr, len(sizes), 0.003, 50, out.data_ptr(), 1,
                                   ws.data_ptr(), nbytes, None, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and torch.equal(out, with_ctx)
    h = ctypes.c_void_p()
    assert lib.lcrec_context_create(ctypes.byref(h)) == 0 and h.value
    assert lib.lcrec_context_set_pipelines(h, 3) == -1 and b"supported" in lib.lcrec_last_error()
    assert lib.lcrec_context_destroy(h) == 0
