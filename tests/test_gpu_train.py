"""The training step's own kernels (csrc/train_ops.hip) and the graph-captured step (lcrec_amd/engine.py).

Floating-point kernels: checked against plain torch fp32/fp64 ops on the CPU (the reference's ops: nn.BatchNorm1d in
train(), F.mse_loss / F.l1_loss, clip_grad_norm_, torch.optim.AdamW -- index/models/layers.py:25-30,
index/models/rqvae.py:74-85, index/trainer.py:118-120) within 1e-5, and the whole step against the reference's recorded
three-step trajectory (tests/golden/f4_step_bn{0,1}.npz)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def _rs(seed):
    return np.random.RandomState(seed)


@pytest.mark.parametrize("n,feat,relu", [(1024, 2048, True), (475, 96, True), (2, 32, True), (2048, 64, False), (333, 100, True),
                                          (4096, 64, True), (5000, 36, True), (300, 66, True), (1025, 24, False)])
def test_bn_relu_forward_backward_match_torch(hip, n, feat, relu):
    rs = _rs(n + feat)
    t = gi.f32(rs.standard_normal((n, feat)) * 2.0 + rs.standard_normal(feat) * 3.0)       # means far from 0
    gamma, beta = gi.f32(1 + 0.2 * rs.standard_normal(feat)), gi.f32(0.3 * rs.standard_normal(feat))
    rm, rv = gi.f32(0.1 * rs.standard_normal(feat)), gi.f32(0.5 + rs.uniform(size=feat))
    gy = gi.f32(rs.standard_normal((n, feat)))
    # reference: torch ops in fp64 on the CPU (what nn.BatchNorm1d + ReLU + autograd compute, without fp32 noise)
    td = torch.from_numpy(t).double().requires_grad_(True)
    gd, bd = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    rmd, rvd = torch.from_numpy(rm).double(), torch.from_numpy(rv).double()
    yd = F.batch_norm(td, rmd, rvd, gd, bd, training=True, momentum=0.1, eps=1e-5)
    if relu:
        yd = F.relu(yd)
    yd.backward(torch.from_numpy(gy).double())
    dev = torch.device(DEV)
    d = lambda a: torch.from_numpy(a).to(dev)
    rm_d, rv_d = d(rm), d(rv)
    y, mean, rstd = hip.ops.bn_relu_forward(d(t), d(gamma), d(beta), 1e-5, 0.1, rm_d, rv_d, relu=relu)
    np.testing.assert_allclose(y.cpu().numpy(), yd.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rmd.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rvd.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mean.cpu().numpy(), t.astype(np.float64).mean(0), rtol=1e-6, atol=3e-6)      # fp32 sums of 1 k values of size ~5
    dt, dg, db, dbias = hip.ops.bn_relu_backward(d(gy), d(t), y, d(gamma), mean, rstd, relu=relu)
    scale = float(np.abs(td.grad.numpy()).max())
    np.testing.assert_allclose(dt.cpu().numpy(), td.grad.numpy(), rtol=1e-4, atol=2e-6 * max(scale, 1.0))
    np.testing.assert_allclose(dg.cpu().numpy(), gd.grad.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bd.grad.numpy(), rtol=1e-5, atol=1e-4)
    # the Linear bias in front of a BatchNorm has gradient sum(dt) = 0 up to rounding
    assert float(dbias.abs().max()) <= 1e-5 * n * max(scale, 1.0)
    # without y: the mask recomputed from t by the forward's own expression -- the same bits
    if relu:
        again = hip.ops.bn_relu_backward(d(gy), d(t), None, d(gamma), mean, rstd, relu=True, beta=d(beta))
        for a, b in zip(again, (dt, dg, db, dbias)):
            assert torch.equal(a, b)
    # deterministic: same bits on a second run
    y2, _, _ = hip.ops.bn_relu_forward(d(t), d(gamma), d(beta), 1e-5, 0.1, None, None, relu=relu)
    assert torch.equal(y2, y)


def test_quantizer_input_grad_with_bias_gradient(hip):
    """lcrec_quantizer_input_grad_bias: the values of lcrec_quantizer_input_grad bit for bit, and their column sums (the encoder's
    last bias gradient) against fp64; batch sizes the one-workgroup kernel takes and one it hands to the two-launch form."""
    dev = torch.device(DEV)
    g = torch.Generator(device=dev).manual_seed(3)
    for n, e in ((1024, 32), (475, 32), (2048, 16), (70000, 32), (1000, 64)):
        z = torch.randn((n, e), generator=g, device=dev)
        cb = torch.randn((256, e), generator=g, device=dev)
        idx = torch.randint(0, 256, (n, 4), generator=g, device=dev)
        gx = torch.randn((n, e), generator=g, device=dev) * 1e-3
        plain = hip.ops.quantizer_input_grad(z, cb, idx[:, 0], 0.25 * 2e-5, 1.0, gx)
        dbias = torch.full((e,), float("nan"), device=dev)
        both = hip.ops.quantizer_input_grad(z, cb, idx[:, 0], 0.25 * 2e-5, 1.0, gx, dbias_out=dbias)
        assert torch.equal(both, plain)
        want = plain.double().sum(0)
        np.testing.assert_allclose(dbias.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-6 * float(plain.abs().sum(0).max()))


@pytest.mark.parametrize("cuts,feat,relu", [((0, 60, 120, 175), 96, True), ((0, 1, 9, 1024), 2048, True), ((0, 500, 500, 777), 100, False)])
def test_sharded_batchnorm_kernels_give_the_whole_batch_result(hip, cuts, feat, relu):
    """The data-parallel split of the BatchNorm step (lcrec_bn_stats -> exchange -> lcrec_bn_merge_stats -> lcrec_bn_relu_apply;
    lcrec_bn_backward_reduce -> all-reduce -> lcrec_bn_backward_apply) on three shards of a batch (one of them possibly
    one row, or empty-adjacent): the union must be nn.BatchNorm1d's result on the whole batch (torch fp64 on the CPU),
    the parameter-gradient shares must sum to the whole-batch gradients, and the sharded loss/gradient of the
    reconstruction term must sum to the global mean's."""
    n = cuts[-1]
    rs = _rs(n + feat)
    t = gi.f32(rs.standard_normal((n, feat)) * 2.0 + rs.standard_normal(feat) * 3.0)
    gamma, beta = gi.f32(1 + 0.2 * rs.standard_normal(feat)), gi.f32(0.3 * rs.standard_normal(feat))
    rm, rv = gi.f32(0.1 * rs.standard_normal(feat)), gi.f32(0.5 + rs.uniform(size=feat))
    gy = gi.f32(rs.standard_normal((n, feat)))
    td = torch.from_numpy(t).double().requires_grad_(True)
    gd, bd = torch.from_numpy(gamma).double().requires_grad_(True), torch.from_numpy(beta).double().requires_grad_(True)
    rmd, rvd = torch.from_numpy(rm).double(), torch.from_numpy(rv).double()
    yd = F.batch_norm(td, rmd, rvd, gd, bd, training=True, momentum=0.1, eps=1e-5)
    if relu:
        yd = F.relu(yd)
    yd.backward(torch.from_numpy(gy).double())
    dev = torch.device(DEV)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shards = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    rows = torch.zeros((len(shards), 2 * feat + 1), dtype=torch.float32, device=dev)
    for r, (a, b) in enumerate(shards):
        rows[r, 0] = b - a
        hip.ops.bn_stats(d(t[a:b]), row_out=rows[r])
    rm_d, rv_d = d(rm), d(rv)
    mean, rstd = hip.ops.bn_merge_stats(rows, 1e-5, 0.1, rm_d, rv_d)
    np.testing.assert_allclose(mean.cpu().numpy(), t.astype(np.float64).mean(0), rtol=1e-6, atol=3e-6)      # fp32 sums of 1 k values of size ~5
    np.testing.assert_allclose(rstd.cpu().numpy(), 1 / np.sqrt(t.astype(np.float64).var(0) + 1e-5), rtol=1e-5)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rmd.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rvd.numpy(), rtol=1e-5, atol=1e-6)
    ys = [hip.ops.bn_relu_apply(d(t[a:b]), d(gamma), d(beta), mean, rstd, relu) for a, b in shards]
    np.testing.assert_allclose(torch.cat(ys).cpu().numpy(), yd.detach().numpy(), rtol=1e-5, atol=1e-5)
    dbeta = [torch.zeros(feat, device=dev) for _ in shards]
    dgamma = [torch.zeros(feat, device=dev) for _ in shards]
    local = [hip.ops.bn_backward_reduce(d(gy[a:b]), d(t[a:b]), y, mean, rstd, relu, dbeta_out=dbeta[r], dgamma_out=dgamma[r])
             for r, ((a, b), y) in enumerate(zip(shards, ys))]
    for r in range(len(shards)):
        assert torch.equal(local[r][0], dbeta[r]) and torch.equal(local[r][1], dgamma[r])
    total = torch.stack(local).sum(0)
    np.testing.assert_allclose(total[1].cpu().numpy(), gd.grad.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(total[0].cpu().numpy(), bd.grad.numpy(), rtol=1e-5, atol=1e-4)
    dts = []
    for (a, b), y in zip(shards, ys):
        bias_grad = torch.full((feat,), 7.0, device=dev)
        dt, db = hip.ops.bn_backward_apply(d(gy[a:b]), d(t[a:b]), y, d(gamma), mean, rstd, total, n, relu, dbias_out=bias_grad)
        assert db.data_ptr() == bias_grad.data_ptr()
        np.testing.assert_allclose(db.cpu().numpy(), dt.double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-4)
        dts.append(dt)
    scale = float(np.abs(td.grad.numpy()).max())
    np.testing.assert_allclose(torch.cat(dts).cpu().numpy(), td.grad.numpy(), rtol=1e-4, atol=2e-6 * max(scale, 1.0))
    # reconstruction term: shares of the global mean and of its gradient
    for kind, fn in (("mse", F.mse_loss), ("l1", F.l1_loss)):
        out = torch.from_numpy(gy).requires_grad_(True)
        ref = fn(out, torch.from_numpy(t), reduction="mean")
        ref.backward()
        parts = [hip.ops.recon_loss_grad(d(gy[a:b]), d(t[a:b]), kind, global_rows=n) for a, b in shards]
        np.testing.assert_allclose(sum(float(p[0]) for p in parts), ref.item(), rtol=1e-6)
        np.testing.assert_allclose(torch.cat([p[1] for p in parts]).cpu().numpy(), out.grad.numpy(), rtol=1e-6, atol=1e-12)


def test_flatten_codebooks_aliases_one_storage_only(hip):
    """ops.flatten_codebooks hands lcrec_rq_assign a view when the codebooks lie back to back in ONE storage (the training
    engine's flat parameter buffer) and a copy otherwise -- in particular for separately allocated parameters that happen to
    be neighbours in memory: a view past the end of the first one's storage would resize it (move the parameter)."""
    dev = torch.device(DEV)
    a, b = torch.randn(32, 32, device=dev), torch.randn(32, 32, device=dev)       # two allocations, usually adjacent
    size, ptr = a.untyped_storage().nbytes(), a.data_ptr()
    flat, ks = hip.ops.flatten_codebooks([a, b])
    assert ks == [32, 32] and torch.equal(flat, torch.cat([a.reshape(-1), b.reshape(-1)]))
    assert a.untyped_storage().nbytes() == size and a.data_ptr() == ptr and flat.data_ptr() != ptr
    buf = torch.randn(3 * 1024, device=dev)
    v = [buf[i * 1024:(i + 1) * 1024].view(32, 32) for i in range(3)]
    flat, _ = hip.ops.flatten_codebooks(v[:2])
    assert flat.data_ptr() == v[0].data_ptr() and torch.equal(flat, buf[:2048]) and buf.untyped_storage().nbytes() == 3 * 4096
    flat, _ = hip.ops.flatten_codebooks(v[1:])
    assert flat.data_ptr() == v[1].data_ptr() and torch.equal(flat, buf[1024:])
    flat, _ = hip.ops.flatten_codebooks([v[0], v[2]])                                # same storage, not neighbours: a copy
    assert flat.data_ptr() != v[0].data_ptr() and torch.equal(flat, torch.cat([buf[:1024], buf[2048:]]))


def test_relu_bias_backward_and_losses_match_torch(hip):
    rs = _rs(11)
    dev = torch.device(DEV)
    d = lambda a: torch.from_numpy(a).to(dev)
    gy, y = gi.f32(rs.standard_normal((777, 200))), gi.f32(np.maximum(rs.standard_normal((777, 200)), 0))
    g, db = hip.ops.relu_bias_backward(d(gy), d(y), relu=True)
    want = gy * (y > 0)
    assert np.array_equal(g.cpu().numpy(), want)
    np.testing.assert_allclose(db.cpu().numpy(), want.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    g2, db2 = hip.ops.relu_bias_backward(d(gy), None, relu=False)
    assert np.array_equal(g2.cpu().numpy(), gy)
    np.testing.assert_allclose(db2.cpu().numpy(), gy.astype(np.float64).sum(0), rtol=1e-5, atol=1e-5)
    for kind, fn in (("mse", F.mse_loss), ("l1", F.l1_loss)):
        out = torch.from_numpy(gi.f32(rs.standard_normal((513, 768)))).requires_grad_(True)
        x = torch.from_numpy(gi.f32(rs.standard_normal((513, 768))))
        ref = fn(out, x, reduction="mean")
        ref.backward()
        loss, grad = hip.ops.recon_loss_grad(out.detach().to(dev), x.to(dev), kind)
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-6)
        np.testing.assert_allclose(grad.cpu().numpy(), out.grad.numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("decoupled,schedule", [(True, "linear"), (True, "constant"), (False, None)])
def test_clip_and_adamw_match_torch(hip, decoupled, schedule):
    """Five steps of clip_grad_norm_(1.0) + AdamW/Adam + warm-up schedule on a flat buffer against torch's own
    (single-tensor, CPU) implementation and transformers' multipliers as restated in lcrec_amd.trainer."""
    from lcrec_amd.trainer import constant_schedule_with_warmup, linear_schedule_with_warmup
    rs = _rs(5)
    n = 100_003
    p0 = gi.f32(rs.standard_normal(n) * 0.1)
    grads = [gi.f32(rs.standard_normal(n) * s) for s in (0.001, 0.01, 0.02, 1e-4, 0.003)]
    pr = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    cls = torch.optim.AdamW if decoupled else torch.optim.Adam
    opt = cls([pr], lr=1e-3, weight_decay=1e-4)
    sched = {"linear": lambda: linear_schedule_with_warmup(opt, 2, 10), "constant": lambda: constant_schedule_with_warmup(opt, 2),
             None: lambda: None}[schedule]()
    dev = torch.device(DEV)
    p = torch.from_numpy(p0.copy()).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    lr_used = torch.zeros((), dtype=torch.float32, device=dev)
    for i, g in enumerate(grads):
        pr.grad = torch.from_numpy(g.copy())
        lr_ref = opt.param_groups[0]["lr"]
        norm_ref = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        if sched is not None:
            sched.step()
        gd = torch.from_numpy(g.copy()).to(dev)
        clip = hip.ops.grad_norm_clip(gd, 1.0)
        hip.ops.adamw_step(p, gd, m, v, step, 1e-3, (0.9, 0.999), 1e-8, 1e-4, decoupled, clip=clip,
                           schedule={"linear": 1, "constant": 0, None: -1}[schedule], warmup_steps=2, total_steps=10,
                           lr_out=lr_used)
        np.testing.assert_allclose(clip[0].item(), float(norm_ref), rtol=1e-6)
        np.testing.assert_allclose(lr_used.item(), lr_ref, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(gd.cpu().numpy(), pr.grad.numpy(), rtol=1e-6, atol=1e-12)          # clipped in place
        np.testing.assert_allclose(p.cpu().numpy(), pr.detach().numpy(), rtol=2e-5, atol=1e-7, err_msg=f"step {i}")
    assert int(step.item()) == len(grads)
    st = opt.state[pr]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("n,k,f", [(1024, 768, 2048), (1024, 2048, 1024), (1000, 256, 128), (475, 128, 64), (1024, 64, 32),
                                   (2048, 1024, 512), (130, 4096, 2048)])
def test_linear_bn_forward_is_the_oracle_chain_with_batch_statistics(hip, oracle, n, k, f):
    """lcrec_linear_bn_forward: the previous layer's BatchNorm + ReLU applied to the operand on its way into LDS
    (layers.py:25-30 never materialised), the product bit for bit the oracle's fma chains on that operand, and this layer's
    batch statistics from the epilogue (per-tile partials merged by the last tile of each column strip) against fp64."""
    ops = hip.ops
    dev = torch.device(DEV)
    rs = _rs(40 + n + k + f)
    tp = gi.f32(rs.standard_normal((n, k)) * 1.5 + 0.3)
    sc, sh = gi.f32(0.5 + rs.uniform(size=k)), gi.f32(0.3 * rs.standard_normal(k))
    W, b = gi.f32(rs.standard_normal((f, k)) * (2.0 / (k + f)) ** 0.5), gi.f32(0.1 * rs.standard_normal(f))
    gamma, beta = gi.f32(1 + 0.1 * rs.standard_normal(f)), gi.f32(0.1 * rs.standard_normal(f))
    rm0, rv0 = gi.f32(0.1 * rs.standard_normal(f)), gi.f32(0.5 + rs.uniform(size=f))
    d = lambda a: torch.from_numpy(a.copy()).to(dev)
    u = oracle.affine_relu(tp, sc, sh, relu=True)
    want = oracle.linear(u, W, b, threads=8)
    assert ops.linear_bn_supported(n, k, f)
    for pro, stats in ((True, True), (False, True), (True, False)):
        rm, rv = d(rm0), d(rv0)
        t, st = ops.linear_bn_forward(d(tp) if pro else d(u), d(W), d(b), in_fold=(d(sc), d(sh)) if pro else None, in_relu=True,
                                      bn=(d(gamma), d(beta), 1e-5, 0.1, rm, rv) if stats else None)
        assert np.array_equal(t.cpu().numpy(), want), (pro, stats)
        if not stats:
            assert st is None
            continue
        w64 = want.astype(np.float64)
        mean, var = w64.mean(0), w64.var(0)
        rstd = 1.0 / np.sqrt(var + 1e-5)
        np.testing.assert_allclose(st[0].cpu().numpy(), mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st[1].cpu().numpy(), rstd, rtol=1e-5)
        np.testing.assert_allclose(st[2].cpu().numpy(), gamma * rstd, rtol=1e-5)
        np.testing.assert_allclose(st[3].cpu().numpy(), beta - mean * gamma * rstd, rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(rm.cpu().numpy(), 0.9 * rm0 + 0.1 * mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rv.cpu().numpy(), 0.9 * rv0 + 0.1 * var * n / (n - 1), rtol=1e-5)
    assert int(ops._ticket(dev, force=True).abs().sum()) == 0          # every strip's ticket is back at zero
    # a second pair of calls gives the same bits (the merge order does not depend on which tile arrives last)
    a = ops.linear_bn_forward(d(tp), d(W), d(b), in_fold=(d(sc), d(sh)), bn=(d(gamma), d(beta), 1e-5, 0.1, None, None))[1]
    b2 = ops.linear_bn_forward(d(tp), d(W), d(b), in_fold=(d(sc), d(sh)), bn=(d(gamma), d(beta), 1e-5, 0.1, None, None))[1]
    assert all(torch.equal(x, y) for x, y in zip(a, b2))


def test_folded_batchnorm_operands_in_the_backward_products(hip, oracle):
    """The consumers of a never-materialised activation: lcrec_linear_backward_weights with (x_scale, x_shift, relu) equals
    the same launch on the materialised input bit for bit -- also on a ragged batch, whose rows past the end must stay
    zero -- and lcrec_bn_relu_backward with the mask recomputed from t equals the mask from the stored activation."""
    ops = hip.ops
    dev = torch.device(DEV)
    rs = _rs(77)
    d = lambda a: torch.from_numpy(a.copy()).to(dev)
    for n in (1024, 475):
        probs_f, probs_m = [], []
        outs_f, outs_m = [], []
        for k, f in ((768, 2048), (2048, 1024), (256, 128), (64, 32)):
            x = gi.f32(rs.standard_normal((n, k)))
            sc, sh = gi.f32(0.5 + rs.uniform(size=k)), gi.f32(0.3 * rs.standard_normal(k))
            gy = gi.f32(rs.standard_normal((n, f)) * 0.01)
            u = oracle.affine_relu(x, sc, sh, relu=True)
            gf, gm = torch.empty((f, k), device=dev), torch.empty((f, k), device=dev)
            probs_f.append((d(gy), d(x), gf, (d(sc), d(sh), True)))
            probs_m.append((d(gy), d(u), gm))
            outs_f.append(gf)
            outs_m.append(gm)
        probs_f.append((probs_m[0][0], probs_m[0][1], torch.empty_like(outs_m[0])))          # a problem without a fold in the same launch
        ops.linear_backward_weights(probs_f)
        ops.linear_backward_weights(probs_m)
        for a, b in zip(outs_f, outs_m):
            assert torch.equal(a, b), n
        assert torch.equal(probs_f[-1][2], outs_m[0])
    n, f = 1000, 512
    t = gi.f32(rs.standard_normal((n, f)))
    gy = gi.f32(rs.standard_normal((n, f)))
    gamma = gi.f32(1 + 0.1 * rs.standard_normal(f))
    mean, var = t.mean(0), t.var(0)
    rstd = gi.f32(1 / np.sqrt(var + 1e-5))
    sc = gi.f32(gamma * rstd)
    sh = gi.f32(0.05 - mean * sc)
    y = oracle.affine_relu(t, sc, sh, relu=True)
    a = ops.bn_relu_backward(d(gy), d(t), d(y), d(gamma), d(gi.f32(mean)), d(rstd), True)
    b = ops.bn_relu_backward(d(gy), d(t), None, d(gamma), d(gi.f32(mean)), d(rstd), True, fold=(d(sc), d(sh)))
    assert all(torch.equal(p, q) for p, q in zip(a, b))


def test_ticket_forms_give_the_bits_of_the_two_launch_forms(hip):
    """include/lcrec.h, `ticket` arguments: with the caller's 4-byte word the last workgroup to arrive finishes a partial-sum
    reduction inside the launch -- same order of additions as the finishing launch it replaces, hence the same bits -- and the
    word is left zero.  Per-level SSE of lcrec_rq_assign / lcrec_rq_apply_level, the reconstruction loss, the gradient norm,
    AdamW's step counter, and the Sinkhorn set-up (global min/max by the distance launch's last workgroup)."""
    ops = hip.ops
    dev = torch.device(DEV)
    rs = _rs(31)
    z = torch.from_numpy(gi.f32(rs.standard_normal((5000, 32)))).to(dev)
    cbs = [torch.from_numpy(gi.f32(rs.standard_normal((256, 32)) * 0.6 ** l)).to(dev) for l in range(3)]
    flat, ks = ops.flatten_codebooks(cbs)
    out = torch.from_numpy(gi.f32(rs.standard_normal((1027, 768)))).to(dev)
    x = torch.from_numpy(gi.f32(rs.standard_normal((1027, 768)))).to(dev)
    g = torch.from_numpy(gi.f32(rs.standard_normal(3_000_003) * 0.01)).to(dev)
    zb = z[:1024].contiguous()
    res = {}
    for use in (False, True):
        ops.USE_TICKETS = use
        try:
            idx, xq, sse, resid = ops.rq_assign(z, flat, ks, want_xq=True, want_sse=True, want_resid=True)
            wide = torch.full((5000, 5), -7, dtype=torch.int64, device=dev)
            ops.rq_assign(z, flat, ks, want_sse=True, idx_into=(wide, 1))
            col = ops.sinkhorn_assign(zb, cbs[0], 0.003, 50)
            _, _, sse1 = ops.rq_apply_level(zb, cbs[0], col, want_sse=True)
            loss, grad = ops.recon_loss_grad(out, x, "mse")
            clip = ops.grad_norm_clip(g, 1.0)
            p = g.clone()
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            step = torch.zeros((), dtype=torch.int64, device=dev)
            for _ in range(3):
                ops.adamw_step(p, g.clone(), m, v, step, 1e-3, clip=clip)
            res[use] = [idx, sse, wide, col, sse1, loss, grad, clip, p, step]
        finally:
            ops.USE_TICKETS = True
    assert int(res[True][9]) == 3 and int(res[False][9]) == 3
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    wide = res[True][2]
    assert torch.equal(wide[:, 1:4], res[True][0]) and bool((wide[:, 0] == -7).all()) and bool((wide[:, 4] == -7).all())
    assert int(ops._ticket(dev).abs().sum()) == 0                 # every call left the word zero


def test_a_nan_loss_freezes_the_optimizer_and_ema_state(hip):
    """The sticky NaN flag of lcrec_step_losses as skip_flag of lcrec_adamw_step / lcrec_ema_update: nothing is updated
    and the step counter stays, i.e. the state a caller finds is that of the last good step -- the effect of the
    reference raising before the offending step's backward (index/trainer.py:116)."""
    ops = hip.ops
    dev = torch.device(DEV)
    rs = _rs(32)
    p = torch.from_numpy(gi.f32(rs.standard_normal(10_000))).to(dev)
    g = torch.from_numpy(gi.f32(rs.standard_normal(10_000))).to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    flag = torch.zeros(2, dtype=torch.bool, device=dev)
    sse = torch.tensor([1.0, float("nan")], dtype=torch.float64, device=dev)
    recon = torch.ones((), dtype=torch.float32, device=dev)
    last = torch.zeros(3, dtype=torch.float32, device=dev)
    ops.adamw_step(p, g, m, v, step, 1e-3, skip_flag=flag[0])
    good = [t.clone() for t in (p, m, v, step)]
    ops.step_losses(sse, 8, 32, 0.25, 1.0, recon, last, nan_flag=flag[0])
    assert bool(flag[0]) and not bool(flag[1]) and bool(torch.isnan(last[0]))
    ops.adamw_step(p, g, m, v, step, 1e-3, skip_flag=flag[0])
    assert all(torch.equal(a, b) for a, b in zip(good, (p, m, v, step))) and int(step) == 1
    en, ew = torch.ones(16, device=dev), torch.ones((16, 32), device=dev)
    cb = torch.ones((16, 32), device=dev)
    ops.ema_update(en, ew, cb, torch.full((16,), 3.0, device=dev), torch.full((16, 32), 5.0, device=dev), 0.9, 1e-5, skip_flag=flag[0])
    assert bool((en == 1).all()) and bool((ew == 1).all()) and bool((cb == 1).all())
    # the poison probe: a negative first assignment sets the second flag
    probe = torch.tensor([-1], dtype=torch.int64, device=dev)
    ops.step_losses(sse, 8, 32, 0.25, 1.0, recon, last, nan_flag=flag[0], poison_probe=probe, poison_flag=flag[1])
    assert bool(flag[1])


def _tiny(hip, bn):
    g = np.load(os.path.join(GOLD, f"f4_step_bn{bn}.npz"))
    model = hip.RQVAE(in_dim=128, num_emb_list=[256] * 4, e_dim=16, layers=[64, 32], dropout_prob=0.0, bn=bool(bn),
                      loss_type="mse", quant_loss_weight=1.0, beta=0.25, kmeans_init=False, kmeans_iters=100,
                      sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    sd = {k[4:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith("sd__")}
    model.load_state_dict(sd, strict=True)
    x = torch.from_numpy(gi.f32(gi.rs(400 + bn).standard_normal((256, 128)))).to(DEV)
    return g, model.to(DEV), x


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("bn", [0, 1])
def test_engine_reproduces_reference_trajectory(hip, bn, use_graph):
    """F4: trainer.py:111-120 three times through the graph-captured step (step 1 eager, step 2 captured + replayed,
    step 3 replayed) against the REFERENCE's recorded losses, gradient norm, learning rate, gradients and post-step
    state -- the same assertions tests/test_gpu_modules.py makes for the autograd path."""
    from lcrec_amd.engine import TrainEngine
    from lcrec_amd.trainer import linear_schedule_with_warmup
    g, model, x = _tiny(hip, bn)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    sched = linear_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    assert TrainEngine.unsupported_reason(model, opt) is None
    eng = TrainEngine(model, opt, "linear", 2, 10, use_graph=use_graph)
    traj = g["trajectory"]
    for step in range(3):
        eng.step(x)
        loss, recon, rq_loss = eng.last.tolist()
        got = [loss, recon, rq_loss, eng.clip[0].item()]
        np.testing.assert_allclose(got, traj[step][:4], rtol=2e-5, atol=1e-12, err_msg=f"step {step}")
        # traj[s][4] is the learning rate AFTER scheduler.step() of step s, i.e. the one step s + 1 uses
        np.testing.assert_allclose(eng.lr_used.item(), 0.0 if step == 0 else traj[step - 1][4], rtol=1e-6, atol=1e-12)
        if step == 0:
            gmax = max(np.abs(g[f]).max() for f in g.files if f.startswith("grad__"))
            coef = eng.clip[1].item()
            for k, p in model.named_parameters():
                np.testing.assert_allclose(p.grad.cpu().numpy() / coef, g["grad__" + k], rtol=1e-4, atol=1e-6 * gmax, err_msg=k)
            for k, v in model.state_dict().items():
                ref = g["step1__" + k]
                if np.issubdtype(ref.dtype, np.floating):
                    np.testing.assert_allclose(v.cpu().numpy(), ref, rtol=1e-4, atol=2e-6, err_msg=k)
                else:
                    assert np.array_equal(v.cpu().numpy(), ref), k
    total, recon_total = eng.end_epoch(sched)
    np.testing.assert_allclose(total, traj[:, 0].astype(np.float64).sum(), rtol=2e-5)
    assert eng.graph_replays == (2 if use_graph else 0)
    assert sched.last_epoch == 3 and abs(sched.get_last_lr()[0] - 1e-3 * (10 - 3) / 8) < 1e-12
    assert float(opt.state[next(model.parameters())]["step"]) == 3.0


@pytest.mark.parametrize("bn", [False, True])
def test_engine_equals_autograd_path_on_the_run_sh_architecture(hip, bn):
    """Same model, same batches, four steps: the autograd path (torch BatchNorm / AdamW / clip) and the engine (own
    kernels, one hipGraph) agree -- bit for bit in the forward GEMMs, to rounding where the op order differs."""
    from lcrec_amd.engine import TrainEngine
    torch.manual_seed(5)
    xs = [torch.randn(512, 768, device=DEV) for _ in range(2)]
    x_init = torch.randn(1024, 768, device=DEV)

    def build():
        torch.manual_seed(7)
        m = hip.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=gi.RUN_SH_LAYERS, bn=bn, kmeans_init=False,
                      sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50).to(DEV)
        with torch.no_grad():
            z = m.eval().encoder(x_init)
            for l, q in enumerate(m.rq.vq_layers):
                q.embedding.weight.copy_(z[l * 256:(l + 1) * 256] * (0.6 ** l))
        return m.train()

    a, b = build(), build()
    opt_a = torch.optim.AdamW(a.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    opt_b = torch.optim.AdamW(b.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    eng = TrainEngine(b, opt_b, None, 0, 0)
    losses = []
    for step in range(4):
        x = xs[step % 2]
        opt_a.zero_grad()
        out, rq_loss, idx = a(x)
        loss, _ = a.compute_loss(out, rq_loss, xs=x)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(a.parameters(), 1.0)
        opt_a.step()
        eng.step(x)
        losses.append((loss.item(), eng.last[0].item()))
    assert eng.graph_replays == 3
    # step 0 starts from identical state; from then on the two runs carry their own rounding (BatchNorm statistics in
    # a different order can flip a near-tied Sinkhorn assignment), so later losses are compared more loosely
    for i, (la, lb) in enumerate(losses):
        np.testing.assert_allclose(lb, la, rtol=1e-5 if i == 0 else 3e-4, err_msg=f"step {i}")
    sa, sb = a.state_dict(), b.state_dict()
    # Adam divides by sqrt(v): where a gradient entry is itself rounding noise (dead units) the update direction is too,
    # so a small fraction of entries may sit a fraction of one step (lr = 1e-3) apart; everything else agrees closely.
    # With BatchNorm the first encoder layers' gradients survive a 3-4 digit cancellation (fixture F11 measures it on the
    # reference itself), so after four Adam steps the two paths' parameters are only bounded here; what pins each path at
    # this width is test_run_sh_step_gradients_against_the_fp64_reference below.
    for k in sa:
        if bn and sa[k].dtype.is_floating_point:
            assert np.abs(sb[k].cpu().numpy() - sa[k].cpu().numpy()).max() < 5e-3, k
            continue
        if sa[k].dtype.is_floating_point:
            va, vb = sa[k].cpu().numpy(), sb[k].cpu().numpy()
            # (a bias whose gradient is rounding noise of a column sum gets lr-sized Adam steps of either sign: a handful of
            # elements per tensor can sit a few 1e-5 apart after four steps, whatever the two summation orders are)
            off = ~np.isclose(vb, va, rtol=1e-3, atol=5e-6)
            assert off.sum() <= max(8, 0.2 * off.size) and np.abs(vb - va).max() < 2e-3, (k, off.sum(), np.abs(vb - va).max())
        else:
            assert torch.equal(sa[k], sb[k]), k


def _run_sh_model(hip, g):
    sd_np, x = gi.run_sh_train_case()
    model = hip.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=gi.RUN_SH_LAYERS, dropout_prob=0.0, bn=True,
                      loss_type="mse", quant_loss_weight=1.0, beta=0.25, kmeans_init=False, kmeans_iters=100,
                      sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    for l in range(4):
        sd[f"rq.vq_layers.{l}.embedding.weight"] = torch.from_numpy(g["codebooks"][l].copy())
    model.load_state_dict(sd, strict=True)
    return model.to(DEV).train(), torch.from_numpy(x).to(DEV)


@pytest.mark.parametrize("path", ["engine", "engine-fused-bn", "autograd"])
def test_run_sh_step_gradients_against_the_fp64_reference(hip, path):
    """F11: the step index/run.sh actually trains (768 -> 2048-...-64 -> 32, BatchNorm, 4 x 256 codes, Sinkhorn on the last
    level, batch 1024; trainer.py:111-120, layers.py:19-30) against the imported reference run in fp64, tensor by tensor.
    The reference's own fp32 run is the measure of what fp32 can deliver: each of our paths must sit within
    f11_check.SLACK x that distance of the fp64 gradient (the first encoder layers lose 3-4 digits in BatchNorm's
    backward in ANY fp32 evaluation; everything else is pinned to ~1e-6).  The engine is checked on its eager first step
    and again on the captured-and-replayed second one (the warm-up schedule's first learning rate is 0, so both steps
    differentiate the same parameters)."""
    import f11_check
    from lcrec_amd.engine import TrainEngine
    g = np.load(os.path.join(GOLD, "f11_run_sh_step.npz"))
    model, x = _run_sh_model(hip, g)
    want = g["f64__scalars"]
    names = [k for k, _ in model.named_parameters()]
    assert sorted(names) == sorted(f11_check.tensors(g))

    def judge(grads, scalars, idx, what):
        flipped = int((idx != g["idx"].astype(np.int64)).any(1).sum())
        # An evaluation whose latents differ in the last bit (BatchNorm statistics summed in another order; the folded form's
        # single fma) may send a near-tied row or two of the 1024 to another code on the Sinkhorn level: a different, equally
        # valid problem for those rows, not different arithmetic.  Then the yardstick is the fp64 evaluation of THAT problem
        # (f11_check.f64_gradients_for: the oracle's torch restatement in fp64 with the codes forced; with the fixture's codes
        # it reproduces the reference's fp64 run to 1e-9, tests/test_oracle_golden.py) -- same bounds, nothing loosened.
        assert flipped <= 3, (what, flipped)
        target, f64 = want, None
        if flipped:
            target, f64 = f11_check.f64_gradients_for(g, idx)
        np.testing.assert_allclose(scalars[:3], target[:3], rtol=1e-5, err_msg=what)      # loss, recon, rq_loss
        np.testing.assert_allclose(scalars[3], target[3], rtol=1e-4, err_msg=what)        # gradient norm before clipping
        rows, bad = f11_check.report(g, grads, f64)
        print(f"\n[{what}] rows assigned differently: {flipped}\n" + f11_check.table(rows))
        assert not bad, what + "\n" + f11_check.table(bad)

    if path == "autograd":
        out, rq_loss, idx = model(x)
        loss, recon = model.compute_loss(out, rq_loss, xs=x)
        loss.backward()
        grads = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        judge(grads, [loss.item(), recon.item(), rq_loss.item(), float(norm)], idx.cpu().numpy(), "autograd path")
        return
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
    # "engine-fused-bn": BatchNorm folded into the GEMMs on either side of it (lcrec_linear_bn_forward; opt-in, see engine.py)
    eng = TrainEngine(model, opt, "linear", 2, 10, fuse_bn=path == "engine-fused-bn")
    before = eng.flat_p.clone()
    for step, what in enumerate(("engine, eager step", "engine, captured step")):
        eng.step(x)
        coef = eng.clip[1].item()
        grads = {k: (p.grad / coef).cpu().numpy() for k, p in model.named_parameters()}
        loss, recon, rq_loss = eng.last.tolist()
        judge(grads, [loss, recon, rq_loss, eng.clip[0].item()], eng.last_idx.cpu().numpy(), what)
        if step == 0:
            assert torch.equal(eng.flat_p, before)           # learning rate 0 on the first step: same parameters again
    assert eng.graph_replays == 1


def test_trainer_uses_the_engine_and_matches_the_autograd_epochs(hip, tmp_path):
    """Trainer.fit over a small dataset with a ragged last batch, engine on vs off: same epoch losses and collision
    rate; the engine captured one graph per batch size and replayed it."""
    from lcrec_amd import main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    data = torch.from_numpy(gi.toy_items(3, n=3000, d=128)).to(DEV)
    results = {}
    for mode in ("auto", "off"):
        argv = ["--data_path", "unused", "--ckpt_dir", str(tmp_path / mode), "--device", DEV, "--batch_size", "768",
                "--epochs", "3", "--eval_step", "3", "--no_kmeans_init", "--num_emb_list", "32", "32", "32", "--e_dim", "32",
                "--layers", "64", "--sk_epsilons", "0.0", "0.0", "0.003", "--train_engine", mode, "--bn", "True",
                "--lr_scheduler_type", "linear", "--warmup_epochs", "1"]
        args = cli.parse_args(argv)
        cli.seed_everything(2024)
        model = cli.build_model(args, 128)
        loader = DeviceLoader(data, 768, True, DEV)
        tr = Trainer(args, model, len(loader))
        per_epoch = [tr._train_epoch(loader, e) for e in range(3)]
        results[mode] = (per_epoch, tr._valid_epoch(loader), tr)
    on, off = results["auto"], results["off"]
    assert on[2].engine is not None and off[2].engine is None
    assert on[2].engine.graph_replays == 3 * 4 - 2              # 4 batches/epoch (3 x 768 + 696); the first step at each of the two sizes is eager
    for (la, ra), (lb, rb) in zip(on[0], off[0]):
        np.testing.assert_allclose([la, ra], [lb, rb], rtol=1e-3)
    assert abs(on[1] - off[1]) < 0.02
    # a checkpoint written while the engine owns the parameters has the reference's layout and reloads
    path = on[2]._save_checkpoint(epoch=2, ckpt_file="e.pth")
    from lcrec_amd import generate_indices as gen
    ck = gen.load_checkpoint(path)
    assert sorted(ck) == ["args", "best_collision_rate", "best_loss", "epoch", "optimizer", "state_dict"]
    fresh = cli.build_model(ck["args"], 128)
    fresh.load_state_dict(ck["state_dict"])
    for k, v in on[2].model.state_dict().items():
        assert torch.equal(v.cpu(), fresh.state_dict()[k]), k
    st = ck["optimizer"]["state"]
    assert len(st) == len(list(model.parameters())) and float(st[0]["step"]) == 12.0


def test_engine_covers_the_ema_codebook_update(hip, tmp_path):
    """index_improve's EMA path (use_ema): the captured step includes lcrec_ema_update; steps on which a dead-code reset
    is due run eagerly.  Same epochs as the autograd path without resets; with resets the replay count shows which
    steps left the graph."""
    from lcrec_amd import main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    data = torch.from_numpy(gi.toy_items(4, n=3000, d=128)).to(DEV)

    def run(mode, reset_interval):
        argv = ["--data_path", "unused", "--ckpt_dir", str(tmp_path / f"{mode}{reset_interval}"), "--device", DEV,
                "--batch_size", "1000", "--epochs", "4", "--no_kmeans_init", "--num_emb_list", "32", "32", "--e_dim", "32",
                "--layers", "64", "--sk_epsilons", "0.0", "0.0", "--train_engine", mode, "--no_bn", "--ema_decay", "0.9",
                "--reset_interval", str(reset_interval), "--reset_threshold", "0.01", "--reset_seed", "5"]
        args = cli.parse_args(argv)
        cli.seed_everything(2024)
        model = cli.build_model(args, 128)
        for l, q in enumerate(model.rq.vq_layers):
            q.reset_generator = torch.Generator(device=DEV).manual_seed(5 + l)
        loader = DeviceLoader(data, 1000, True, DEV)
        tr = Trainer(args, model, len(loader))
        losses = [tr._train_epoch(loader, e) for e in range(4)]
        return tr, losses

    on, lo = run("auto", 10_000)
    off, lf = run("off", 10_000)
    assert on.engine is not None and on.engine.graph_replays == 12 - 1 and off.engine is None
    np.testing.assert_allclose(np.array(lo), np.array(lf), rtol=5e-4)
    for qa, qb in zip(on.model.rq.vq_layers, off.model.rq.vq_layers):
        assert qa.step_count == qb.step_count == 12
        np.testing.assert_allclose(qa._ema_cluster_size.cpu().numpy(), qb._ema_cluster_size.cpu().numpy(), rtol=2e-3, atol=1e-3)
        np.testing.assert_allclose(qa._ema_w.cpu().numpy(), qb._ema_w.cpu().numpy(), rtol=5e-3, atol=5e-3)
    # resets every 4th step: steps 4, 8, 12 run eagerly (and step 1): 12 - 4 replays; the run is reproducible
    r1, l1 = run("auto", 4)
    r2, l2 = run("auto", 4)
    assert r1.engine.graph_replays == 12 - 4 and l1 == l2
    assert all(torch.equal(a, b) for a, b in zip(r1.model.state_dict().values(), r2.model.state_dict().values()))


def test_engine_small_batches_keep_the_sinkhorn_level_capturable(hip, tmp_path):
    """Batches of <= 64 rows put the Sinkhorn level into the one-workgroup LDS class, whose group table must not travel by
    a host copy inside a captured graph (it is a kernel argument for <= 4 groups): engine == autograd path at batch 48,
    with BatchNorm, ragged last batch of 24 rows included."""
    from lcrec_amd import main as cli
    from lcrec_amd.datasets import DeviceLoader
    from lcrec_amd.trainer import Trainer
    data = torch.from_numpy(gi.toy_items(8, n=3000, d=128))[:600].to(DEV)
    res = {}
    for mode in ("auto", "off"):
        argv = ["--data_path", "unused", "--ckpt_dir", str(tmp_path / mode), "--device", DEV, "--batch_size", "48", "--epochs", "2",
                "--no_kmeans_init", "--num_emb_list", "64", "64", "--e_dim", "32", "--layers", "64", "--sk_epsilons", "0.0", "0.003",
                "--train_engine", mode, "--bn", "True"]
        args = cli.parse_args(argv)
        cli.seed_everything(2024)
        loader = DeviceLoader(data, 48, True, DEV)
        tr = Trainer(args, cli.build_model(args, 128), len(loader))
        res[mode] = (tr, [tr._train_epoch(loader, e) for e in range(2)])
    assert res["auto"][0].engine.graph_replays == 2 * 13 - 2
    np.testing.assert_allclose(np.array(res["auto"][1]), np.array(res["off"][1]), rtol=2e-3)
