"""CPU: pin the two oracles to the golden vectors produced by the real reference
(oracle/make_golden.py -> tests/golden/).  No GPU, no /root/reference at run time."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as fh:
        return json.load(fh)


def test_manifest_lists_every_fixture_and_hashes_match():
    import hashlib
    m = manifest()
    files = sorted(f for f in os.listdir(GOLD) if f != "manifest.json")
    assert files == sorted(m["fixtures"])
    for f in files:
        with open(os.path.join(GOLD, f), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == m["fixtures"][f]["sha256"], f
    # what make_golden.py observed when it replayed the fixtures through both oracles
    for f, meta in m["fixtures"].items():
        assert meta.get("c_oracle_idx_mismatch_rows", 0) == 0, f
        assert meta.get("torch_ref_bit_identical", True) is True, f


@pytest.mark.parametrize("levels,codes", [(4, 256), (8, 1024)])
def test_c_oracle_rq_matches_reference(oracle, levels, codes):
    g = gold(f"f1_rq_{levels}x{codes}.npz")
    z, cbs = gi.rq_kat(levels, codes)
    o = oracle.rq_assign(z, cbs)
    assert np.array_equal(o["idx"], g["idx"].astype(np.int64))          # bit-exact indices
    np.testing.assert_allclose(o["xq"], g["xq"], rtol=0, atol=1e-6)
    n, e = z.shape
    loss = np.mean([(1 + 0.25) * s / (n * e) for s in o["sse"]])         # vq.py:90-92, rq.py:53
    assert abs(loss - float(g["rq_loss"])) <= 1e-5 * abs(float(g["rq_loss"]))


def test_c_oracle_tie_break_matches_reference(oracle):
    z, cb = gi.tie_case()
    o = oracle.rq_assign(z, [cb, cb])
    want = gold("f2_ties.npz")["idx"].astype(np.int64)
    assert np.array_equal(o["idx"], want)
    assert list(want[:8, 0]) == [5, 5, 5, 36, 36, 0, 0, 5]     # duplicates resolve to the lowest index
    assert want[8, 0] == 10                                      # exact midpoint -> lower index


@pytest.mark.parametrize("name,in_dim,n,bn", [("f8_encode_768_bn0.npz", 768, 2048, False),
                                              ("f8_encode_768_bn1.npz", 768, 1024, True),
                                              ("f8_encode_4096_bn0.npz", 4096, 512, False)])
def test_c_oracle_encode_assign_matches_reference(oracle, name, in_dim, n, bn):
    g = gold(name)
    dims, Ws, bs, bns, x = gi.encoder_case(in_dim, n, bn=bn)
    scs, shs = zip(*[gi.fold_bn(b) for b in bns])
    o = oracle.encode_assign(x, Ws, bs, list(g["codebooks"]), list(scs), list(shs), threads=8)
    assert np.array_equal(o["idx"], g["idx"].astype(np.int64))
    # north_star tolerance: 1e-5 fp32 on floats (latents are O(1))
    np.testing.assert_allclose(o["latent"], g["latent"], rtol=1e-5, atol=1e-5)
    # the reference agrees with itself across batch sizes on this fixture (it need not in general)
    assert np.array_equal(g["idx_batch64"], g["idx"])


@pytest.mark.parametrize("bn", [0, 1])
def test_c_oracle_tiny_model_eval(oracle, bn):
    g = gold(f"f4_step_bn{bn}.npz")
    sd = {k[4:]: g[k] for k in g.files if k.startswith("sd__")}
    names = gi.state_dict_names(3, bool(bn), 4)
    x = gi.f32(gi.rs(400 + bn).standard_normal((256, 128)))
    Ws = [sd[n + ".weight"] for n in names["encoder"]]
    bs = [sd[n + ".bias"] for n in names["encoder"]]
    scs, shs = [], []
    for l in range(3):
        if bn and l < 2:
            b = names["bn"]["encoder"][l]
            sc, sh = gi.fold_bn({k: sd[f"{b}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")})
        else:
            sc = sh = None
        scs.append(sc)
        shs.append(sh)
    o = oracle.encode_assign(x, Ws, bs, [sd[n] for n in names["codebooks"]], scs, shs)
    assert np.array_equal(o["idx"], g["eval_idx"].astype(np.int64))
    np.testing.assert_allclose(o["latent"], g["eval_latent"], rtol=1e-5, atol=1e-6)
    loss = np.mean([(1 + 0.25) * s / (256 * 16) for s in o["sse"]])
    assert abs(loss - float(g["eval_rq_loss"])) <= 1e-5 * abs(float(g["eval_rq_loss"]))


def test_c_oracle_ema_matches_reference(oracle):
    g = gold("f5_ema.npz")
    r = gi.rs(500)
    z = gi.f32(r.standard_normal((512, 32)))
    cb = gi.f32(r.standard_normal((256, 32)) * 0.9)
    cb[200:] *= 40.0
    idx = oracle.rq_assign(z, [cb])["idx"][:, 0]
    assert np.array_equal(idx, g["idx"].astype(np.int64))
    cnt, s = oracle.code_stats(idx, z, 256)
    assert np.array_equal(cnt, g["count"]) and np.array_equal(s, g["sum"])
    en, ew, ecb = oracle.ema_update(g["ema_count0"], g["ema_sum0"], cb, cnt, s, 0.99, 1e-5)
    np.testing.assert_allclose(en, g["ema_count1"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ew, g["ema_sum1"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ecb, g["codebook1"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ torch_ref (the timed CPU baseline)
def test_torch_ref_rq_and_sinkhorn_match_reference():
    from oracle import torch_ref
    z, cbs = gi.rq_kat(4, 256)
    g = gold("f1_rq_4x256.npz")
    xq, loss, idx = torch_ref.rq(torch.from_numpy(z), [torch.from_numpy(c) for c in cbs], 0.25, False, [0.0] * 4, 50)
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))
    np.testing.assert_allclose(xq.numpy(), g["xq"], rtol=0, atol=1e-6)
    for B in (8, 2048):
        g = gold(f"f3_sinkhorn_{B}.npz")
        z, cb = gi.sinkhorn_case(B)
        _, _, idx = torch_ref.vq(torch.from_numpy(z), torch.from_numpy(cb), 0.25, True, 0.003, 50)
        got, want = idx.numpy(), g["idx"].astype(np.int64)
        # rows may legitimately flip only where the reference's own top-2 margin is at rounding level
        bad = got != want
        assert not (bad & (g["margin"] > 1e-9)).any()


@pytest.mark.parametrize("bn", [0, 1])
def test_torch_ref_train_step_matches_reference(bn):
    from oracle import torch_ref
    g = gold(f"f4_step_bn{bn}.npz")
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd__")}
    x = torch.from_numpy(gi.f32(gi.rs(400 + bn).standard_normal((256, 128))))
    spec = torch_ref.Spec(128, [256] * 4, 16, [64, 32], bn=bool(bn), sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    leaf = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
            for k, v in sd.items()}
    out, rq_loss, idx = torch_ref.forward(spec, leaf, x, use_sk=True, training=True)
    loss, recon = torch_ref.compute_loss(spec, out, rq_loss, x)
    loss.backward()
    assert np.array_equal(idx.numpy(), g["train_idx"].astype(np.int64))
    np.testing.assert_allclose(out.detach().numpy(), g["train_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(loss), float(g["train_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(loss), g["trajectory"][0, 0], rtol=1e-5)
    for k in g.files:
        if k.startswith("grad__"):
            np.testing.assert_allclose(leaf[k[6:]].grad.numpy(), g[k], rtol=1e-4, atol=1e-7, err_msg=k)


def test_neartie_fixture_c2_is_what_the_oracle_computes(oracle):
    """F9 (oracle/neartie_audit.py): on the Games-shaped inputs the oracle reproduces the recorded index-matrix hash
    and the recorded number of near-tie rows, and that hash is also the reference's (0 differing rows at this shape).
    The C3 fixture (1 M rows) is replayed on the GPU only (tests/test_gpu_neartie.py)."""
    import hashlib
    f = gold("f9_neartie_c2.npz")
    n, in_dim = gi.NEARTIE_CASES["c2"]
    x = gi.neartie_items(n, in_dim)
    assert hashlib.sha256(x[:65536].tobytes()).hexdigest() == str(f["sha_x_head"])
    dims, Ws, bs = gi.neartie_encoder(in_dim)
    o = oracle.encode_assign(x, Ws, bs, list(f["codebooks"]), threads=8, want_margin=True)
    sha = hashlib.sha256(np.ascontiguousarray(o["idx"], dtype=np.int16).tobytes()).hexdigest()
    assert sha == str(f["sha_oracle"]) == str(f["sha_ref4096"]) == str(f["sha_ref64"])
    flagged = (o["margin"] <= np.float32(f["tau"]) * o["scale"]).any(1)
    assert int(flagged.sum()) == int(f["flagged_count"]) and len(f["rows"]) == 0


def test_torch_ref_training_step_at_the_run_sh_width_against_fp64_reference():
    """F11: one trainer.py:111-120 step of the run.sh model (BatchNorm, Sinkhorn level, batch 1024) through the oracle's
    torch restatement in fp32, judged against the reference's fp64 gradients with the reference's own fp32 run as the
    measure of what fp32 can deliver (tests/f11_check.py)."""
    import f11_check
    from oracle import torch_ref
    g = gold("f11_run_sh_step.npz")
    sd_np, x = gi.run_sh_train_case()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    for l in range(4):
        sd[f"rq.vq_layers.{l}.embedding.weight"] = torch.from_numpy(g["codebooks"][l].copy())
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
            for k, v in sd.items()}
    spec = torch_ref.Spec(768, [256] * 4, 32, gi.RUN_SH_LAYERS, bn=True, sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    xt = torch.from_numpy(x)
    out, rq_loss, idx = torch_ref.forward(spec, leaf, xt, use_sk=True, training=True)
    loss, recon = torch_ref.compute_loss(spec, out, rq_loss, xt)
    loss.backward()
    assert np.array_equal(idx.numpy(), g["idx"].astype(np.int64))
    assert np.array_equal(g["idx"], g["idx_f64"])                # fp64 assigns the same codes: the gradients are comparable
    want = g["f64__scalars"]
    np.testing.assert_allclose([loss.item(), recon.item(), rq_loss.item()], want[:3], rtol=1e-5)
    grads = {k: v.grad.numpy() for k, v in leaf.items() if v.requires_grad}
    rows, bad = f11_check.report(g, grads)
    assert not bad, f11_check.table(bad)
    norm = np.sqrt(sum(float((v.astype(np.float64) ** 2).sum()) for v in grads.values()))
    np.testing.assert_allclose(norm, want[3], rtol=1e-4)
    # the fp64 evaluation with forced codes (what the GPU tests judge against when a near-tied row went to another code): with
    # the fixture's own codes it is the reference's fp64 run
    scal, g64 = f11_check.f64_gradients_for(g, g["idx"])
    np.testing.assert_allclose(scal, want, rtol=1e-11)
    for k in f11_check.tensors(g):
        a, b = gi.strided_sample(g64[k]), g["f64__sample__" + k]
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(b).max(), 1e-12) + 1e-16, k
    # what the fixture says about fp32 itself: the first encoder layers lose 3-4 digits, the rest sits at ~1e-6
    m = manifest()["fixtures"]["f11_run_sh_step.npz"]["reference_f32_vs_f64_gradient_rel_err"]
    assert m["encoder.mlp_layers.1.weight"] > 1e-4 and m["decoder.mlp_layers.25.weight"] < 1e-5
