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
    got = hip.ops.linear_forward(t(x), t(W), t(b), t(sc), t(sh), relu=relu).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


@pytest.mark.parametrize("n,e,Ks", [
    (64, 32, [256] * 4),
    (1000, 32, [256] * 4),
    (70001, 32, [256] * 4),
    (515, 16, [256] * 4),
    (300, 32, [256] * 3),
    (2049, 32, [1024] * 8),
    (513, 64, [256, 128]),
    (1, 32, [32]),
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
