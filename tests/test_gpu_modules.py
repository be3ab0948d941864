"""Module API (RQVAE / ResidualVectorQuantizer / VectorQuantizer / MLPLayers) on the MI355X
against the golden vectors produced by the real reference (tests/golden/f4_*, f8_*)."""
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _tiny(hip, bn):
    g = np.load(os.path.join(GOLD, f"f4_step_bn{bn}.npz"))
    model = hip.RQVAE(in_dim=128, num_emb_list=[256] * 4, e_dim=16, layers=[64, 32], dropout_prob=0.0, bn=bool(bn),
                      loss_type="mse", quant_loss_weight=1.0, beta=0.25, kmeans_init=False, kmeans_iters=100,
                      sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd__")}
    model.load_state_dict(sd, strict=True)          # same keys, shapes, dtypes as the reference
    x = torch.from_numpy(gi.f32(gi.rs(400 + bn).standard_normal((256, 128)))).to(DEV)
    return g, model.to(DEV), x


@pytest.mark.parametrize("bn", [0, 1])
def test_eval_forward_matches_reference(hip, bn):
    g, model, x = _tiny(hip, bn)
    model.eval()
    with torch.no_grad():
        out, rq_loss, idx = model(x, use_sk=False)
        loss, recon = model.compute_loss(out, rq_loss, xs=x)
        idx2 = model.get_indices(x)
    assert np.array_equal(idx.cpu().numpy(), g["eval_idx"].astype(np.int64))       # bit-exact indices
    assert np.array_equal(idx2.cpu().numpy(), g["eval_idx"].astype(np.int64))
    np.testing.assert_allclose(out.cpu().numpy(), g["eval_out"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rq_loss.item(), float(g["eval_rq_loss"]), rtol=1e-5)
    np.testing.assert_allclose(loss.item(), float(g["eval_loss_total"]), rtol=1e-5)
    np.testing.assert_allclose(recon.item(), float(g["eval_loss_recon"]), rtol=1e-5)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("bn", [0, 1])
def test_training_steps_match_reference(hip, bn, fused):
    """trainer.py:111-120 three times: forward (Sinkhorn on the last level), loss, backward,
    clip 1.0, AdamW, linear warm-up -- against the reference's recorded trajectory."""
    from lcrec_amd.trainer import linear_schedule_with_warmup
    g, model, x = _tiny(hip, bn)
    model.train()
    # fused=True is what lcrec_amd.trainer.Trainer builds on a HIP device: same rule, one kernel
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=fused)
    sched = linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    traj = g["trajectory"]
    for step in range(3):
        opt.zero_grad()
        out, rq_loss, idx = model(x)
        loss, recon = model.compute_loss(out, rq_loss, xs=x)
        loss.backward()
        if step == 0:
            assert np.array_equal(idx.cpu().numpy(), g["train_idx"].astype(np.int64))
            np.testing.assert_allclose(out.detach().cpu().numpy(), g["train_out"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(loss.item(), float(g["train_loss"]), rtol=1e-5)
            np.testing.assert_allclose(rq_loss.item(), float(g["train_rq_loss"]), rtol=1e-5)
            # absolute floor relative to the largest gradient entry of the whole model: the bias of a
            # Linear that feeds BatchNorm has an exactly-zero gradient, i.e. pure rounding noise
            gmax = max(np.abs(g[f]).max() for f in g.files if f.startswith("grad__"))
            for k, p in model.named_parameters():
                ref = g["grad__" + k]
                np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-6 * gmax, err_msg=k)
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        sched.step()
        np.testing.assert_allclose([loss.item(), recon.item(), rq_loss.item(), float(gn), sched.get_last_lr()[0]],
                                   traj[step], rtol=2e-5, err_msg=f"step {step}")
        if step == 0:
            for k, v in model.state_dict().items():
                ref = g["step1__" + k]
                if np.issubdtype(ref.dtype, np.floating):
                    np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-4, atol=2e-6, err_msg=k)
                else:
                    assert np.array_equal(v.cpu().numpy(), ref), k


@pytest.mark.parametrize("name,in_dim,n,bn", [("f8_encode_768_bn0.npz", 768, 2048, False),
                                              ("f8_encode_768_bn1.npz", 768, 1024, True),
                                              ("f8_encode_4096_bn0.npz", 4096, 512, False)])
def test_get_indices_full_size_matches_reference(hip, name, in_dim, n, bn):
    g = np.load(os.path.join(GOLD, name))
    dims, Ws, bs, bns, x = gi.encoder_case(in_dim, n, bn=bn)
    model = hip.RQVAE(in_dim=in_dim, num_emb_list=[256] * 4, e_dim=32, layers=gi.RUN_SH_LAYERS, bn=bn,
                      kmeans_init=False, sk_epsilons=[0.0] * 4, sk_iters=50)
    names = gi.state_dict_names(len(Ws), bn, 4)
    sd = model.state_dict()
    for l, nme in enumerate(names["encoder"]):
        sd[nme + ".weight"] = torch.from_numpy(Ws[l])
        sd[nme + ".bias"] = torch.from_numpy(bs[l])
    if bn:
        for l, nme in enumerate(names["bn"]["encoder"]):
            for k, v in bns[l].items():
                sd[f"{nme}.{k}"] = torch.from_numpy(v)
    for l, nme in enumerate(names["codebooks"]):
        sd[nme] = torch.from_numpy(g["codebooks"][l])
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    xd = torch.from_numpy(x).to(DEV)
    idx = model.get_indices(xd)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (n, 4)
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int64))          # bit-exact vs the reference
    with torch.no_grad():
        lat = model.encoder(xd)
    np.testing.assert_allclose(lat.cpu().numpy(), g["latent"], rtol=1e-5, atol=1e-5)
    # batch 64 (generate_indices.py:78) gives the same rows: the fma-chain order does not depend on the batch
    idx64 = torch.cat([model.get_indices(xd[i:i + 64]) for i in range(0, n, 64)])
    assert torch.equal(idx64, idx)


def test_no_cpu_path(hip):
    model = hip.RQVAE(in_dim=128, num_emb_list=[256] * 2, e_dim=16, layers=[64], sk_epsilons=[0.0, 0.0])
    with pytest.raises(hip.LcrecError):
        model.eval().get_indices(torch.zeros(4, 128))


def test_device_kmeans_matches_oracle_lloyd_and_is_reproducible(hip):
    """layers.kmeans_device (replaces the sklearn call of layers.py:69-82, which
    is itself not bit-pinned): Lloyd iterations equal oracle/cpu_oracle.kmeans_lloyd bit for bit from
    the same initial centres; seeding is reproducible from its generator and improves on random rows."""
    from lcrec_amd import layers
    from oracle import cpu_oracle
    rs = gi.rs(77)
    blobs = rs.standard_normal((24, 32)) * 3.0
    x = gi.f32(blobs[rs.randint(0, 24, size=3000)] + 0.3 * rs.standard_normal((3000, 32)))
    xd = torch.from_numpy(x).to(DEV)
    init = x[rs.choice(3000, size=64, replace=False)]
    got = layers.kmeans_device(xd, 64, num_iters=7, tol=0.0, init=torch.from_numpy(init), relocate_empty=False)
    want, iters = cpu_oracle.kmeans_lloyd(x, init, 7, tol=0.0)
    assert np.array_equal(got.cpu().numpy(), want)

    def inertia(c):
        return float(torch.cdist(xd, c).min(1).values.pow(2).sum())

    g1 = torch.Generator(device=DEV).manual_seed(5)
    g2 = torch.Generator(device=DEV).manual_seed(5)
    c1 = layers.kmeans_device(xd, 24, num_iters=100, generator=g1)
    c2 = layers.kmeans_device(xd, 24, num_iters=100, generator=g2)
    assert torch.equal(c1, c2) and bool(torch.isfinite(c1).all())
    seed_only = layers.kmeans_pp_seed(xd, 24, torch.Generator(device=DEV).manual_seed(5))
    assert inertia(c1) <= inertia(seed_only) < inertia(xd[:24])
    # quality against the reference's own initialiser on the F10 data (sklearn's centres are in the fixture): the device
    # k-means++ / Lloyd run, empty clusters relocated as sklearn does, reaches a comparable inertia
    g10 = np.load(os.path.join(GOLD, "f10_kmeans.npz"))
    x10 = torch.from_numpy(gi.kmeans_case()).to(DEV)
    for K, iters in ((256, 10), (64, 100)):
        ref_c = torch.from_numpy(g10[f"centres_{K}_{iters}"]).to(DEV)
        mine = layers.kmeans_device(x10, K, num_iters=iters, generator=torch.Generator(device=DEV).manual_seed(1))
        i_ref = float(torch.cdist(x10, ref_c).min(1).values.pow(2).sum())
        i_mine = float(torch.cdist(x10, mine).min(1).values.pow(2).sum())
        assert i_mine <= 1.10 * i_ref, (K, i_mine, i_ref)
        assert int((hip.ops.code_stats(hip.ops.rq_assign(x10, mine.reshape(-1), [K])[0][:, 0], x10, K)[0] == 0).sum()) <= K // 50
    # same entry point the quantiser calls when main.py is given --kmeans_impl device
    layers.KMEANS_IMPL = "device"
    try:
        torch.manual_seed(3)
        a = layers.kmeans(xd, 16, 20)
        torch.manual_seed(3)
        b = layers.kmeans(xd, 16, 20)
        assert torch.equal(a, b) and a.shape == (16, 32)
    finally:
        layers.KMEANS_IMPL = "sklearn"


@pytest.mark.parametrize("batch", [13, 100, 475])
def test_ragged_batches_train_and_match_the_reference_ops(hip, batch):
    """The last batch of an epoch is whatever is left (Games: 16 859 % 1024 = 475 rows).  One training step on
    batches that are not multiples of 8 / 32, with the e_dim-16 layer (narrower than a K slice): forward values
    and every parameter gradient against the reference's op sequence on CPU (oracle/torch_ref + autograd)."""
    from oracle import torch_ref
    g, model, _ = _tiny(hip, 0)
    x = torch.from_numpy(gi.f32(gi.rs(900 + batch).standard_normal((batch, 128))))
    spec = torch_ref.Spec(128, [256] * 4, 16, [64, 32], bn=False, sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    leaf = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd__")}
    leaf = {k: (v.requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in leaf.items()}
    out_ref, rq_ref, idx_ref = torch_ref.forward(spec, leaf, x, use_sk=True, training=True)
    loss_ref, _ = torch_ref.compute_loss(spec, out_ref, rq_ref, x)
    loss_ref.backward()
    model.train()
    out, rq_loss, idx = model(x.to(DEV))
    loss, _ = model.compute_loss(out, rq_loss, xs=x.to(DEV))
    loss.backward()
    assert np.array_equal(idx.cpu().numpy()[:, :3], idx_ref.numpy()[:, :3])          # argmin levels: exact
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-5)
    gmax = max(float(v.grad.abs().max()) for v in leaf.values() if v.requires_grad and v.grad is not None)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), leaf[k].grad.numpy(), rtol=2e-4, atol=2e-6 * gmax, err_msg=k)
