#!/usr/bin/env python3
"""One training step through the autograd path and through the engine from identical state: per-parameter gradient agreement."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lcrec_amd as hip  # noqa: E402
from lcrec_amd.engine import TrainEngine  # noqa: E402

DEV = "cuda:0"
bn = True
torch.manual_seed(5)
x = torch.randn(512, 768, device=DEV)
x_init = torch.randn(1024, 768, device=DEV)


def build():
    torch.manual_seed(7)
    m = hip.RQVAE(in_dim=768, num_emb_list=[256] * 4, e_dim=32, layers=[2048, 1024, 512, 256, 128, 64], bn=bn, kmeans_init=False,
                  sk_epsilons=[0.0, 0.0, 0.0, 0.003], sk_iters=50).to(DEV)
    with torch.no_grad():
        z = m.eval().encoder(x_init)
        for l, q in enumerate(m.rq.vq_layers):
            q.embedding.weight.copy_(z[l * 256:(l + 1) * 256] * (0.6 ** l))
    return m.train()


a, b = build(), build()
opt_b = torch.optim.AdamW(b.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
eng = TrainEngine(b, opt_b, None, 0, 0, use_graph=False)
out, rq_loss, idx = a(x)
loss, _ = a.compute_loss(out, rq_loss, xs=x)
loss.backward()
eng.step(x)
coef = eng.clip[1].item()
print("loss", loss.item(), eng.last[0].item(), "clip coef", coef, "norm", eng.clip[0].item())
gn = torch.sqrt(sum((p.grad ** 2).sum() for p in a.parameters())).item()
print("autograd grad norm", gn)
for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
    ga, gb = pa.grad, pb.grad / coef
    denom = ga.abs().max().item() + 1e-30
    print(f"{k:40s} |g|max {denom:.3e}  max abs diff / |g|max {((ga - gb).abs().max().item() / denom):.3e}")

# ---- four steps: autograd vs engine (eager) vs engine (graph)
print("---- 4 steps")
xs = [torch.randn(512, 768, device=DEV) for _ in range(2)]
ma, me, mg = build(), build(), build()
oa = torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
ee = TrainEngine(me, torch.optim.AdamW(me.parameters(), lr=1e-3, weight_decay=1e-4, fused=True), None, 0, 0, use_graph=False)
eg = TrainEngine(mg, torch.optim.AdamW(mg.parameters(), lr=1e-3, weight_decay=1e-4, fused=True), None, 0, 0, use_graph=True)
for step in range(4):
    xb = xs[step % 2]
    oa.zero_grad()
    out, rq_loss, idx = ma(xb)
    loss, _ = ma.compute_loss(out, rq_loss, xs=xb)
    loss.backward()
    gna = torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
    oa.step()
    ee.step(xb)
    eg.step(xb)
    if step >= 1:
        worst = []
        for (k, pa), (_, pb) in zip(ma.named_parameters(), me.named_parameters()):
            ga, gb = pa.grad, pb.grad
            worst.append(((ga - gb).abs().max().item() / (ga.abs().max().item() + 1e-30), k))
            sa = oa.state[pa]
            sb = ee.optimizer.state[pb]
            if k == "encoder.mlp_layers.1.weight":
                print(f"   {k}: clipped-grad rel diff {worst[-1][0]:.2e}; exp_avg rel diff "
                      f"{((sa['exp_avg'] - sb['exp_avg']).abs().max() / sa['exp_avg'].abs().max()).item():.2e}; exp_avg_sq rel diff "
                      f"{((sa['exp_avg_sq'] - sb['exp_avg_sq']).abs().max() / sa['exp_avg_sq'].abs().max()).item():.2e}; step {float(sa['step'])}")
        worst.sort(reverse=True)
        if step == 2:
            for w, k in sorted(worst, key=lambda t: t[1]):
                if "weight" in k:
                    print(f"      {k:36s} {w:.2e}")
            for (k, pa), (_, pb) in zip(ma.named_parameters(), me.named_parameters()):
                if k.endswith("bias"):
                    print(f"      param {k:30s} max|a| {pa.abs().max().item():.2e} max|a-b| {(pa - pb).abs().max().item():.2e}")
    wa = dict(ma.named_parameters())["encoder.mlp_layers.1.weight"]
    we = dict(me.named_parameters())["encoder.mlp_layers.1.weight"]
    wg = dict(mg.named_parameters())["encoder.mlp_layers.1.weight"]
    print(f"step {step}: loss autograd {loss.item():.6f} eager {ee.last[0].item():.6f} graph {eg.last[0].item():.6f}; "
          f"grad norm {float(gna):.5f} / {ee.clip[0].item():.5f} / {eg.clip[0].item():.5f}; "
          f"w1 max|autograd-eager| {(wa - we).abs().max().item():.3e}  max|eager-graph| {(we - wg).abs().max().item():.3e}")

# ---- which backward call diverges first?  record the (gy, x) arguments of every lcrec_linear_backward of step index 2
print("---- linear_backward arguments at the third step")
from lcrec_amd import ops as _ops  # noqa: E402
ma, me = build(), build()
oa = torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
ee = TrainEngine(me, torch.optim.AdamW(me.parameters(), lr=1e-3, weight_decay=1e-4, fused=True), None, 0, 0, use_graph=False)
orig_lb = _ops.linear_backward
rec = None


def spy(gy, x, w, *a, **k):
    if rec is not None:
        rec.append((gy.detach().clone(), x.detach().clone(), w.detach().clone()))
    return orig_lb(gy, x, w, *a, **k)


_ops.linear_backward = spy
orig_bb = _ops.bn_relu_backward
rec_bn = None


def spy_bn(gy, t, y, gamma, mean, rstd, relu=True, **k):
    out = orig_bb(gy, t, y, gamma, mean, rstd, relu, **k)
    if rec_bn is not None:
        rec_bn.append([v.detach().clone() for v in (gy, t, y, gamma, mean, rstd, out[0], out[1], out[2])])
    return out


_ops.bn_relu_backward = spy_bn
import lcrec_amd.layers as _layers  # noqa: E402
for step in range(3):
    xb = xs[step % 2]
    ra, re_ = [], []
    ba, be = [], []
    rec = ra if step == 2 else None
    rec_bn = ba if step == 2 else None
    oa.zero_grad()
    out, rq_loss, idx = ma(xb)
    loss, _ = ma.compute_loss(out, rq_loss, xs=xb)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(ma.parameters(), 1.0)
    oa.step()
    rec = re_ if step == 2 else None
    rec_bn = be if step == 2 else None
    ee.step(xb)
rec = None
rec_bn = None
names = ["gy", "t", "y", "gamma", "mean", "rstd", "dt", "dgamma", "dbeta"]
for i, (A, B) in enumerate(zip(ba, be)):
    r = lambda p, q: ((p - q).abs().max() / (p.abs().max() + 1e-30)).item()
    print(f"  bn call {i:2d} F={A[1].shape[1]:5d}: " + "  ".join(f"{n} {r(p, q):.1e}" for n, p, q in zip(names, A, B)))
    if i == 6:
        ya, yb = A[2], B[2]
        flips = (ya > 0) != (yb > 0)
        pre_a = (A[1] - A[4]) * A[5]
        print(f"     mask flips between the paths: {int(flips.sum())} of {flips.numel()}; elements with 0 < y < 1e-4: {int(((ya > 0) & (ya < 1e-4)).sum())};"
              f" columns with batch std of t below 1e-3: {int((A[1].std(0) < 1e-3).sum())}; min rstd {A[5].min().item():.2f} max rstd {A[5].max().item():.2f};"
              f" |gy| at flipped elements max {A[0][flips].abs().max().item() if flips.any() else 0:.2e} vs max|gy| {A[0].abs().max().item():.2e};"
              f" xhat at flips {pre_a[flips][:5].tolist()}")
        gy, t, y, gamma, mean, rstd = [v.double() for v in A[:6]]
        g = gy * (y > 0)
        xh = (t - mean) * rstd
        dt_ref = gamma * rstd * (g - g.mean(0) - xh * (g * xh).mean(0))
        print("     autograd-path dt vs fp64 formula:", r(dt_ref.float(), A[6]), " engine-path dt vs its own inputs:",
              r((B[3].double() * B[5].double() * ((B[0].double() * (B[2] > 0)) - (B[0].double() * (B[2] > 0)).mean(0) - ((B[1].double() - B[4].double()) * B[5].double()) * ((B[0].double() * (B[2] > 0)) * ((B[1].double() - B[4].double()) * B[5].double())).mean(0))).float(), B[6]))
for i, ((ga, xa, wa_), (gb, xb_, wb_)) in enumerate(zip(ra, re_)):
    r = lambda p, q: ((p - q).abs().max() / (p.abs().max() + 1e-30)).item()
    print(f"  call {i:2d} gy {tuple(ga.shape)}: gy rel diff {r(ga, gb):.2e}   x rel diff {r(xa, xb_):.2e}   W rel diff {r(wa_, wb_):.2e}")
