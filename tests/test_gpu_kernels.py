"""HIP kernels vs the CPU oracle, bit for bit, through the C-ABI (ctypes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rs(seed):
    return np.random.RandomState(seed)


@pytest.mark.parametrize("n,k,out,relu,bn", [
    (128, 32, 128, True, False),
    (1000, 768, 2048, True, False),
    (333, 128, 64, True, True),
    (517, 64, 32, False, False),
    (200, 64, 16, False, False),
    (4096, 2048, 1024, True, True),
    (77, 4096, 2048, True, False),
    (1, 8, 32, False, False),
    # launches that fill the CUs in whole rounds of 256 x 128 tiles go to the ping-pong kernel (K % 32 == 0):
    (8192, 64, 2048, True, True),        # every tile interior
    (16000, 32, 2000, True, False),      # ragged: last row panel half empty, last column tile partial
    (8192, 72, 2048, True, False),       # K % 32 != 0 at that size: generic kernel, guarded loads
    (8192, 40, 2001, False, True),       # ... and an odd width (scalar stores)
    (8192, 256, 2048, True, True),       # BatchNorm epilogue
    (16000, 192, 2048, True, False),     # six K-tiles, last row panel half empty
    (8200, 384, 4096, False, False),     # 33 row panels over 8 XCDs (holes in the XCD-aware tile order), no ReLU
    # launches of at most 128 tiles of 64 x 64 (K % 32 == 0) take the 32 x 64 tiles on v_mfma_f32_16x16x4_f32:
    (1024, 1024, 512, True, True),       # a training step's layer: 128 tiles -> 256 workgroups of the small kernel
    (475, 2048, 1024, True, False),      # the row tail of a Games-sized launch: ragged rows (475 = 14 x 32 + 27)
    (1024, 512, 252, False, True),       # ragged columns: the last column tile reads weight rows past N as zeros
    (31, 32, 4, False, False),           # one K-tile, one partial tile
])
def test_linear_bit_exact(hip, oracle, n, k, out, relu, bn):
    rs = _rs(n + k + out)
    x = rs.standard_normal((n, k)).astype(np.float32)
    W = (rs.standard_normal((out, k)) / np.sqrt(k)).astype(np.float32)
    b = (0.1 * rs.standard_normal(out)).astype(np.float32)
    sc = (1 + 0.1 * rs.standard_normal(out)).astype(np.float32) if bn else None
    sh = (0.1 * rs.standard_normal(out)).astype(np.float32) if bn else None
    want = oracle.linear(x, W, b, sc, sh, relu=relu, threads=8)
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    hip.ops.trace_enable(True)
    got = hip.ops.linear_forward(t(x), t(W), t(b), t(sc), t(sh), relu=relu).cpu().numpy()
    trace = hip.ops.trace_collect()
    hip.ops.trace_enable(False)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
    assert ("linear_fwd_pp_256x128" in trace) == (n >= 8192 and k % 32 == 0), trace      # the intended kernel ran
    if n < 8192:         # (a launch cut into whole rounds for the ping-pong kernel hands its row tail on: that tail may be small too)
        small = k % 32 == 0 and out % 4 == 0 and -(-n // 64) * -(-out // 64) <= 128
        assert ("linear_fwd_32x64" in trace) == small, trace


@pytest.mark.parametrize("n,k,out", [(2048, 768, 2048), (475, 128, 64), (1000, 64, 32), (300, 2048, 1024), (77, 32, 128),
                                     (2048, 36, 96), (1, 64, 32),
                                     (1024, 1024, 2048),      # dW [2048, 1024]: 64 x 128 tiles, both operands k-major
                                     (1000, 2048, 512),       # dX [1000, 2048]: 64 x 128 tiles, ragged rows
                                     (1024, 512, 1024),       # dX [1024, 512] over K = 1024: 128 tiles -> the 32 x 64 tiles, k-major W
                                     (475, 252, 96)])         # ... ragged rows and columns (252 = 3 x 64 + 60), K = 96
def test_linear_backward_bit_exact(hip, oracle, n, k, out):
    """lcrec_linear_backward (k-major operand staging, no transposed copies) against the oracle's restatement
    on explicitly transposed operands: gx = gy W one fma chain per output; gw = gy^T x as the ordered sum of
    lcrec_linear_backward_splits() runs over the batch -- including batches that are not a multiple of the
    K slice (475, 77, 1) and narrow layers that split 16 ways."""
    rs = _rs(n * 3 + k + out)
    x = rs.standard_normal((n, k)).astype(np.float32)
    W = (rs.standard_normal((out, k)) / np.sqrt(k)).astype(np.float32)
    gy = rs.standard_normal((n, out)).astype(np.float32)
    gy[rs.random_sample(gy.shape) < 0.4] = 0.0                    # as after a ReLU mask
    splits = hip.ops.linear_backward_splits(n, k, out)
    assert splits == 1 or (n >= 256 and 2 <= splits <= 16)
    want_gx, want_gw = oracle.linear_backward(gy, x, W, splits=splits, threads=8)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    gx, gw = hip.ops.linear_backward(t(gy), t(x), t(W))
    assert np.array_equal(gx.cpu().numpy(), want_gx), f"gx max abs diff {np.abs(gx.cpu().numpy() - want_gx).max()}"
    assert np.array_equal(gw.cpu().numpy(), want_gw), f"gw max abs diff {np.abs(gw.cpu().numpy() - want_gw).max()}"
    only_gw = hip.ops.linear_backward(t(gy), t(x), t(W), need_gx=False)
    assert only_gw[0] is None and torch.equal(only_gw[1], gw)


@pytest.mark.parametrize("n,e,Ks", [(1024, 32, [256] * 4), (475, 16, [32, 256, 7]), (2048, 32, [1024] * 8), (9000, 32, [64, 64])])
def test_code_stats_levels_equal_per_level_calls(hip, oracle, n, e, Ks):
    """lcrec_code_stats_levels: (count, sum) of every level and the fused codebook gradient in one launch -- the same bits as
    lcrec_code_stats (pinned to the CPU's index_add_ order by the oracle) + lcrec_codebook_grad per level; n = 9000 is
    beyond the one-workgroup sort and takes the level-by-level path."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(n + e)
    L = len(Ks)
    idx = torch.stack([torch.randint(0, K, (n,), generator=g, device=dev) for K in Ks], 1).contiguous()
    idx[: n // 2, 0] = 3                                           # one code owns half the batch (early training)
    resid = [torch.randn((n, e), generator=g, device=dev) for _ in Ks]
    cbs = [torch.randn((K, e), generator=g, device=dev) for K in Ks]
    grads = [torch.full((K, e), float("nan"), device=dev) for K in Ks]
    got = hip.ops.code_stats_levels(idx, resid, Ks, cbs, grads, scale=1.7e-4, weight=0.5)
    for l, K in enumerate(Ks):
        cnt, tot = hip.ops.code_stats(idx[:, l], resid[l], K)
        assert torch.equal(got[l][0], cnt) and torch.equal(got[l][1], tot)
        want = torch.empty((K, e), device=dev)
        hip.ops.codebook_grad(cnt, tot, cbs[l], 1.7e-4, 0.5, want)
        assert torch.equal(grads[l], want)
        assert torch.equal(want, (1.7e-4 * (cnt.unsqueeze(1) * cbs[l] - tot)) * 0.5)       # quantize.py's expression
        oc, os_ = oracle.code_stats(idx[:, l].cpu().numpy(), resid[l].cpu().numpy(), K)
        assert np.array_equal(cnt.cpu().numpy(), oc) and np.array_equal(tot.cpu().numpy(), os_)
    plain = hip.ops.code_stats_levels(idx, resid, Ks)              # statistics only
    assert all(torch.equal(a[1], b[1]) for a, b in zip(plain, got))


def test_grouped_weight_gradients_equal_per_layer_calls(hip, oracle):
    """lcrec_linear_backward_weights: the 14 weight gradients of a training step (run.sh widths, batch 1024 and a ragged
    475) in one launch -- bit-identical to lcrec_linear_backward layer by layer (same K-runs, same order), which the test
    above pins to the oracle; plus a 16-wide layer (out_dim % 32 != 0) and a single-problem call."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9)
    widths = [768, 2048, 1024, 512, 256, 128, 64, 32]
    for n in (1024, 475):
        problems, want = [], []
        for a, b in list(zip(widths[:-1], widths[1:])) + list(zip(widths[::-1][:-1], widths[::-1][1:])) + [(64, 16)]:
            gy = torch.randn((n, b), generator=g, device=dev)
            gy[torch.rand((n, b), generator=g, device=dev) < 0.4] = 0.0
            x = torch.randn((n, a), generator=g, device=dev)
            gw = torch.full((b, a), float("nan"), device=dev)
            problems.append((gy, x, gw))
            if b % 32 == 0:
                want.append(hip.ops.linear_backward(gy, x, torch.empty((b, a), device=dev), need_gx=False)[1])
            else:
                _, gx_w = oracle.linear_backward(gy.cpu().numpy(), x.cpu().numpy(), np.zeros((b, a), np.float32),
                                                 splits=hip.ops.linear_backward_splits(n, a, b), threads=8)
                want.append(torch.from_numpy(gx_w).to(dev))
        for lo in range(0, len(problems), 16):                     # at most 16 problems per launch
            hip.ops.linear_backward_weights(problems[lo:lo + 16])
        for (gy, x, gw), w in zip(problems, want):
            assert torch.equal(gw, w), (n, tuple(gw.shape), (gw - w).abs().max().item())
    one = torch.empty((2048, 768), device=dev)
    hip.ops.linear_backward_weights([(problems[0][0], problems[0][1], one)])
    assert torch.equal(one, problems[0][2])
    # `splits` sets every problem's S itself: 1 = one fma chain over the batch per element (what the training engine asks for),
    # 3 = three runs added in order -- against the oracle's sums with that S, narrow and wide layers alike
    for S in (1, 3):
        picked = [problems[i] for i in (2, 4, 6, 8, 14)]
        outs = [torch.full_like(p[2], float("nan")) for p in picked]
        hip.ops.linear_backward_weights([(p[0], p[1], o) for p, o in zip(picked, outs)], splits=S)
        for (gy, x, _), o in zip(picked, outs):
            _, w = oracle.linear_backward(gy.cpu().numpy(), x.cpu().numpy(), np.zeros((gy.shape[1], x.shape[1]), np.float32),
                                          splits=S, threads=8)
            assert torch.equal(o, torch.from_numpy(w).to(dev)), (S, tuple(o.shape))
    with pytest.raises(hip.LcrecError):
        hip.ops.linear_backward_weights([problems[0]] * 17)


@pytest.mark.parametrize("n,e,Ks", [
    (64, 32, [256] * 4),
    (1000, 32, [256] * 4),
    (70001, 32, [256] * 4),
    (515, 16, [256] * 4),
    (300, 32, [256] * 3),
    (2049, 32, [1024] * 8),
    (513, 64, [256, 128]),
    (1, 32, [32]),
    (3000, 16, [16, 16, 16]),
    (777, 32, [100, 7, 250]),
])
def test_rq_assign_bit_exact(hip, oracle, n, e, Ks):
    rs = _rs(n + e + sum(Ks))
    z = rs.standard_normal((n, e)).astype(np.float32)
    cbs = [(rs.standard_normal((K, e)) * (0.8 ** l)).astype(np.float32) for l, K in enumerate(Ks)]
    want = oracle.rq_assign(z, cbs, want_resid=True)
    dev = torch.device("cuda:0")
    flat, ks = hip.ops.flatten_codebooks([torch.from_numpy(c).to(dev) for c in cbs])
    idx, xq, sse, resid = hip.ops.rq_assign(torch.from_numpy(z).to(dev), flat, ks, want_xq=True, want_sse=True,
                                            want_resid=True)
    assert np.array_equal(idx.cpu().numpy(), want["idx"])
    assert np.array_equal(xq.cpu().numpy(), want["xq"])
    assert np.array_equal(resid.cpu().numpy(), want["resid"])
    np.testing.assert_allclose(sse.cpu().numpy(), want["sse"], rtol=1e-6)
    # the index-only variant (no x_q registers) must agree
    idx2, _, _, _ = hip.ops.rq_assign(torch.from_numpy(z).to(dev), flat, ks)
    assert torch.equal(idx2, idx)
    # near-tie audit outputs: the top-2 gap per level is the oracle's bit for bit, and the flag word is
    # margin <= tau * (xx + cc[idx]) in fp32 -- also when the levels are split over several launches (8 x 1024)
    wm = oracle.rq_assign(z, cbs, want_margin=True)
    for tau in (2.0 ** -17, 0.05):
        audit = {}
        idx3, xq3, _, _ = hip.ops.rq_assign(torch.from_numpy(z).to(dev), flat, ks, want_xq=True, audit=audit, tie_tau=tau)
        assert torch.equal(idx3, idx) and torch.equal(xq3, xq)
        assert np.array_equal(audit["margin"].cpu().numpy(), wm["margin"])
        flags = (wm["margin"] <= np.float32(tau) * wm["scale"])
        want_bits = (flags.astype(np.int64) << np.arange(len(Ks))).sum(1).astype(np.int32)
        assert np.array_equal(audit["neartie"].cpu().numpy(), want_bits)
    assert (want_bits != 0).any() or n < 64                       # tau = 0.05 does flag rows on these inputs


def test_rq_assign_exact_ties_take_first_index(hip, oracle):
    # small integers: every product and sum is exact in fp32 in any order, so ties are real ties
    rs = _rs(7)
    e, K = 32, 256
    cb = rs.randint(-3, 4, size=(K, e)).astype(np.float32)
    cb[100] = cb[5]      # duplicate codes: the lower index must win (vq.py:75 argmin)
    cb[200] = cb[5]
    cb[37] = cb[36]
    z = cb[rs.randint(0, K, size=4096)] + rs.randint(-1, 2, size=(4096, e)).astype(np.float32)
    want = oracle.rq_assign(z, [cb, cb])
    d = ((z[:, None, :] - cb[None, :, :]) ** 2).sum(-1)          # exact in fp32
    assert np.array_equal(want["idx"][:, 0], d.argmin(1))
    dev = torch.device("cuda:0")
    flat, ks = hip.ops.flatten_codebooks([torch.from_numpy(cb).to(dev)] * 2)
    idx, _, _, _ = hip.ops.rq_assign(torch.from_numpy(z).to(dev), flat, ks)
    assert np.array_equal(idx.cpu().numpy(), want["idx"])
    assert not np.isin(idx.cpu().numpy()[:, 0], [100, 200, 37]).any()
    # an exact tie has margin 0 and is flagged under any tau, 0 included
    audit = {}
    hip.ops.rq_assign(torch.from_numpy(z).to(dev), flat, ks, audit=audit, tie_tau=0.0)
    wm = oracle.rq_assign(z, [cb, cb], want_margin=True)
    assert np.array_equal(audit["margin"].cpu().numpy(), wm["margin"])
    tied = np.isin(want["idx"][:, 0], [5, 36])
    assert tied.any() and (audit["margin"].cpu().numpy()[tied, 0] == 0).all()
    assert ((audit["neartie"].cpu().numpy()[tied] & 1) == 1).all()


@pytest.mark.parametrize("n,dims,Ks,bn", [
    (1000, [128, 64, 32, 16], [256] * 4, False),
    (2500, [768, 2048, 1024, 512, 256, 128, 64, 32], [256] * 4, False),
    (700, [768, 2048, 1024, 512, 256, 128, 64, 32], [256] * 4, True),
    (300, [4096, 2048, 1024, 512, 256, 128, 64, 32], [256] * 3, False),
])
def test_encode_assign_bit_exact(hip, oracle, n, dims, Ks, bn):
    rs = _rs(n + sum(dims))
    x = rs.standard_normal((n, dims[0])).astype(np.float32)
    Ws = [(rs.standard_normal((dims[l + 1], dims[l])) * np.sqrt(2.0 / (dims[l] + dims[l + 1]))).astype(np.float32)
          for l in range(len(dims) - 1)]
    bs = [(0.05 * rs.standard_normal(dims[l + 1])).astype(np.float32) for l in range(len(dims) - 1)]
    nl = len(Ws)
    scs = [(1 + 0.1 * rs.standard_normal(dims[l + 1])).astype(np.float32) if (bn and l < nl - 1) else None
           for l in range(nl)]
    shs = [(0.1 * rs.standard_normal(dims[l + 1])).astype(np.float32) if (bn and l < nl - 1) else None
           for l in range(nl)]
    # codebooks at the scale of the latents so margins are data-like
    lat = oracle.encode_assign(x[:256], Ws, bs, [np.zeros((32, dims[-1]), np.float32)], scs, shs)["latent"]
    cbs = [(lat[rs.randint(0, len(lat), size=K)] * (0.7 ** l)).astype(np.float32) for l, K in enumerate(Ks)]
    want = oracle.encode_assign(x, Ws, bs, cbs, scs, shs, threads=8)
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    flat, ks = hip.ops.flatten_codebooks([t(c) for c in cbs])
    idx, latent, xq, sse = hip.ops.encode_assign(t(x), [t(w) for w in Ws], [t(b) for b in bs], flat, ks,
                                                 [t(s) for s in scs], [t(s) for s in shs], want_latent=True,
                                                 want_xq=True, want_sse=True)
    assert np.array_equal(latent.cpu().numpy(), want["latent"]), \
        f"latent max diff {np.abs(latent.cpu().numpy() - want['latent']).max()}"
    assert np.array_equal(idx.cpu().numpy(), want["idx"])
    assert np.array_equal(xq.cpu().numpy(), want["xq"])
    np.testing.assert_allclose(sse.cpu().numpy(), want["sse"], rtol=1e-6)


# ---------------------------------------------------------------------------- Sinkhorn / training kernels
def _ref_sinkhorn_idx(z, cb, eps, iters):
    """Sinkhorn branch of vq.py:76-83 on the CPU: fp32 distances in the canonical fma-chain order
    (C oracle), then the reference's fp32 centring and fp64 sinkhorn_algorithm as torch CPU ops."""
    from oracle import cpu_oracle, torch_ref
    d = torch.from_numpy(cpu_oracle.distances(z, cb))
    Q = torch_ref.sinkhorn(torch_ref.centre_distances(d).double(), eps, iters)
    top2 = torch.topk(Q, 2, dim=-1).values
    margin = ((top2[:, 0] - top2[:, 1]) / top2[:, 0]).numpy()
    return torch.argmax(Q, -1).numpy(), margin


@pytest.mark.parametrize("B,K,e", [(2048, 256, 32), (1000, 256, 32), (256, 256, 32), (300, 100, 16), (4096, 1024, 32), (1000, 1024, 32),
                                   (130, 256, 64),
                                   # every (columns per lane, rows per wave) form of the scaling-form solver: K = 64 .. 1024
                                   (500, 64, 32), (700, 128, 16), (1500, 512, 32), (3000, 256, 32), (600, 512, 32), (333, 192, 32)])
def test_sinkhorn_training_batch(hip, B, K, e):
    rs = _rs(B + K)
    z = rs.standard_normal((B, e)).astype(np.float32)
    cb = (0.8 * rs.standard_normal((K, e))).astype(np.float32)
    want, margin = _ref_sinkhorn_idx(z, cb, 0.003, 50)
    dev = torch.device("cuda:0")
    hip.ops.trace_enable(True)
    got = hip.ops.sinkhorn_assign(torch.from_numpy(z).to(dev), torch.from_numpy(cb).to(dev), 0.003, 50).cpu().numpy()
    trace = hip.ops.trace_collect()
    hip.ops.trace_enable(False)
    bad = got != want
    # fp64 exp/sum order differs from torch CPU by ulps: only rows whose top-2 margin is at that level may move
    assert not (bad & (margin > 1e-9)).any(), f"{bad.sum()} rows differ, min margin of those {margin[bad].min()}"
    assert bad.mean() < 1e-3
    if B * K > 16384:
        # a lone problem beyond the LDS classes (any --batch_size from 65 rows up) must take the multi-workgroup
        # solver, not the one-workgroup-per-group kernels of the collision rounds (3 ms at 256 rows instead of 0.4)
        assert "sinkhorn" in trace and not ({"sinkhorn_small", "sinkhorn_slab", "sinkhorn_tiny"} & set(trace)), trace


def test_sinkhorn_golden_fixture(hip):
    import os
    import golden_inputs as gi
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    dev = torch.device("cuda:0")
    for B in (8, 2048):
        g = np.load(os.path.join(gold, f"f3_sinkhorn_{B}.npz"))
        z, cb = gi.sinkhorn_case(B)
        got = hip.ops.sinkhorn_assign(torch.from_numpy(z).to(dev), torch.from_numpy(cb).to(dev), 0.003, 50).cpu().numpy()
        bad = np.flatnonzero(got != g["idx"].astype(np.int64))
        # The reference's own fp32 distances (MKL summation order) differ from the canonical chains by ~1e-7 relative
        # and exp(-d/0.003) amplifies that, so rows with a near-tied argmax could flip.  oracle/make_golden.py recorded
        # the exact set of rows on which the reference's fp64 solve over CANONICAL-order distances differs from the
        # reference (`canonical_differ_rows`: empty for both fixtures).  The GPU may differ from the reference only there,
        # or where the canonical solve's own top-2 margin is inside the 1e-9 the multi-workgroup association allows (no row of
        # the 2048-row fixture: its smallest margin is 5.9e-3) -- so both fixtures are pinned row for row.
        allowed = set(g["canonical_differ_rows"].tolist())
        if B > 8:     # (8 rows x 256 columns saturates: every row's top-2 entries of Q are exactly tied in the reference too)
            allowed |= set(np.flatnonzero(g["canonical_margin"] <= 1e-9).tolist())
        assert set(bad.tolist()) <= allowed, (B, sorted(set(bad.tolist()) - allowed))


@pytest.mark.parametrize("K,e", [(256, 32), (16, 16), (1024, 32)])
def test_sinkhorn_collision_groups(hip, K, e):
    # many small independent problems, as generate_indices.py:113-119 issues one by one
    rs = _rs(K + e)
    cap = max(2, min(40, 16384 // K))
    sizes = list(rs.randint(2, cap + 1, size=60)) + [1, 2, cap]
    # groups too large for LDS keep Q in the workspace slab, still one workgroup each and side by side
    # (enough of them that side by side beats one batch-sized solve after another: see sk_plan)
    sizes += {256: [65, 300, 700] + [100] * 20, 16: [1500, 1025] + [1100] * 4, 1024: [17, 130] + [20] * 12}[K]
    if K == 16:
        sizes.append(5000)       # beyond the slab limit: a batch-sized problem, multi-launch path
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = int(offs[-1])
    z = rs.standard_normal((n, e)).astype(np.float32)
    # colliding items are near-duplicates in practice: make half of the groups tight clusters
    for g in range(0, len(sizes), 2):
        z[offs[g]:offs[g + 1]] = z[offs[g]] + 1e-3 * rs.standard_normal((sizes[g], e)).astype(np.float32)
    cb = (0.8 * rs.standard_normal((K, e))).astype(np.float32)
    dev = torch.device("cuda:0")
    idx = torch.full((n, 3), -1, dtype=torch.int64, device=dev)
    hip.ops.trace_enable(True)
    hip.ops.sinkhorn_assign(torch.from_numpy(z).to(dev), torch.from_numpy(cb).to(dev), 0.003, 50,
                            group_offsets=offs.tolist(), out=idx[:, 2])
    trace = hip.ops.trace_collect()
    hip.ops.trace_enable(False)
    assert "sinkhorn_slab" in trace and ("sinkhorn_small" in trace or "sinkhorn_tiny" in trace), trace
    got = idx[:, 2].cpu().numpy()
    assert (idx[:, :2] == -1).all()
    nbad = 0
    for g in range(len(sizes)):
        lo, hi = offs[g], offs[g + 1]
        want, margin = _ref_sinkhorn_idx(z[lo:hi], cb, 0.003, 50)
        bad = got[lo:hi] != want
        assert not (bad & (margin > 1e-9)).any(), (g, sizes[g])
        nbad += bad.sum()
    assert nbad <= max(1, n // 500)


@pytest.mark.parametrize("n,e,K", [(1000, 32, 256), (77, 16, 48), (5000, 64, 128), (8192, 32, 1024), (20000, 32, 256)])
def test_apply_level_and_code_stats_bit_exact(hip, oracle, n, e, K):
    rs = _rs(n + e + K)
    z = rs.standard_normal((n, e)).astype(np.float32)
    cb = rs.standard_normal((K, e)).astype(np.float32)
    cb2 = (0.5 * rs.standard_normal((K, e))).astype(np.float32)
    want = oracle.rq_assign(z, [cb, cb2], want_resid=True)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    idx = t(want["idx"])
    xq, r1, sse0 = hip.ops.rq_apply_level(t(z), t(cb), idx[:, 0], want_sse=True)
    assert np.array_equal(r1.cpu().numpy(), want["resid"][1])
    xq, r2, sse1 = hip.ops.rq_apply_level(r1, t(cb2), idx[:, 1], xq=xq, want_sse=True)
    assert np.array_equal(r2.cpu().numpy(), want["resid"][2])
    assert np.array_equal(xq.cpu().numpy(), want["xq"])
    np.testing.assert_allclose([sse0.item(), sse1.item()], want["sse"], rtol=1e-6)
    for l, c in enumerate((cb, cb2)):
        wc, ws = oracle.code_stats(want["idx"][:, l], want["resid"][l], K)
        gc, gs = hip.ops.code_stats(idx[:, l], t(want["resid"][l]), K)
        assert np.array_equal(gc.cpu().numpy(), wc)
        assert np.array_equal(gs.cpu().numpy(), ws)


def test_ema_update_matches_reference_fixture(hip, oracle):
    import os
    import golden_inputs as gi
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f5_ema.npz"))
    r = gi.rs(500)
    z = gi.f32(r.standard_normal((512, 32)))
    cb = gi.f32(r.standard_normal((256, 32)) * 0.9)
    cb[200:] *= 40.0
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    flat, ks = hip.ops.flatten_codebooks([t(cb)])
    idx, _, _, resid = hip.ops.rq_assign(t(z), flat, ks, want_resid=True)
    assert np.array_equal(idx[:, 0].cpu().numpy(), g["idx"].astype(np.int64))
    cnt, tot = hip.ops.code_stats(idx[:, 0], resid[0], 256)
    assert np.array_equal(cnt.cpu().numpy(), g["count"]) and np.array_equal(tot.cpu().numpy(), g["sum"])
    en, ew, w = t(g["ema_count0"]), t(g["ema_sum0"]), t(cb)
    hip.ops.ema_update(en, ew, w, cnt, tot, 0.99, 1e-5)
    # bit-identical to the reference's CPU result on this fixture (and to the C oracle)
    assert np.array_equal(en.cpu().numpy(), g["ema_count1"])
    assert np.array_equal(ew.cpu().numpy(), g["ema_sum1"])
    assert np.array_equal(w.cpu().numpy(), g["codebook1"])


@pytest.mark.parametrize("n,Ks", [(5000, [16, 16, 16]), (20000, [256] * 4), (3000, [1024] * 8), (1, [256] * 4), (64, [4])])
def test_collision_groups_match_reference_helper(hip, n, Ks):
    """lcrec_collision_groups vs the dict-based helper semantics of generate_indices.py:18-42."""
    from lcrec_amd import generate_indices as gen
    rs = _rs(n + len(Ks))
    L = len(Ks)
    if L == 8:   # wide tuples (80 bits): force collisions by duplicating rows
        base = np.stack([rs.randint(0, K, size=n // 3 + 1) for K in Ks], 1)
        rows = base[rs.randint(0, len(base), size=n)]
    else:
        rows = np.stack([rs.randint(0, min(K, 6), size=n) for K in Ks], 1) if n > 64 else \
            np.stack([rs.randint(0, K, size=n) for K in Ks], 1)
    rows = rows.astype(np.int64)
    keys = [tuple(r) for r in rows.tolist()]
    want_groups = gen.get_collision_item(keys)
    counts = gen.get_indices_count(keys)
    got = hip.ops.collision_groups(torch.from_numpy(rows).to("cuda:0"), Ks)
    assert got["unique"] == len(counts)
    assert got["max_count"] == max(counts.values())
    assert got["groups"] == want_groups                       # same order: first occurrence, ids ascending
    assert abs(got["collision_rate"] - (n - len(counts)) / n) < 1e-15
    assert gen.check_collision(keys) == (len(want_groups) == 0)
    rate_only = hip.ops.collision_groups(torch.from_numpy(rows).to("cuda:0"), Ks, want_groups=False)
    assert rate_only["unique"] == got["unique"] and rate_only["max_count"] == got["max_count"]
