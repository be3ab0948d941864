"""The training step of the reference's index/trainer.py:111-123 as ONE hipGraph.

`Trainer._train_epoch` in the reference is, per batch: zero_grad -> model(data) -> compute_loss -> isnan check ->
backward -> clip_grad_norm_(1.0) -> optimizer.step -> scheduler.step -> two loss.item().  Driven
from Python through autograd that is ~170 kernel launches and ~2 ms of host work per step; at the reference's batch
size (1024, index/run.sh:2) the GPU needs less than that, so the step is host-bound.

This module runs the same arithmetic as a straight line of library calls -- no autograd graph, every product through
the same lcrec_* entry points the module API uses, so forward values and GEMM gradients are bit-identical to it:

    encoder  [Linear -> (BatchNorm batch statistics) -> ReLU] x 7      lcrec_linear_forward, lcrec_bn_relu_forward
    quantiser                                                          quantize.quantize_values (rq_assign / sinkhorn / code_stats)
    decoder  same
    loss     mse|l1 + quant_loss_weight * rq_loss, and d loss / d out   lcrec_recon_loss_grad
    backward decoder, closed-form quantiser gradients, encoder         lcrec_bn_relu_backward / lcrec_relu_bias_backward,
                                                                       lcrec_linear_backward (dX), and ONE grouped launch
                                                                       for all weight gradients, written straight into the
                                                                       flat gradient buffer (lcrec_linear_backward_weights)
    clip 1.0 + AdamW + warm-up schedule                                lcrec_grad_norm_clip, lcrec_adamw_step (learning
                                                                       rate and bias corrections from a device step counter)
    loss sums, NaN flag                                                device accumulators, read once per epoch

and captures it with torch.cuda.CUDAGraph (a hipGraph) once per batch size: a step is then one index_select, one copy
and one graph launch on the host.  Parameters, gradients and both Adam moments live in four flat fp32 buffers; the
module's nn.Parameters (and the torch optimizer's state entries, so checkpoints keep the reference's layout) are views
into them.

Measured and dropped: weight gradients (leaves of the dependency graph) and the per-code statistics on a second stream,
i.e. parallel branches in the captured graph, so that the narrow layers' GEMMs run beside the dX chain: 26 fork/join
pairs per step cost more than the overlap gains at batch 1024 (1.80 -> 1.88 ms/step, bn=True 1.99 -> 2.08) and barely pay
at 2048 (2.51 -> 2.45).  With ONE fork/join (the decoder's seven weight gradients queued and run beside the encoder's
chain) nothing changes at 1024 (1.80 -> 1.82) and little at 2048 (2.51 -> 2.46): the replayed graph does not run the two
branches' under-filled kernels side by side to any useful degree.

The EMA codebook update of index_improve/ is part of the captured step (lcrec_ema_update); the steps on which a dead-code
reset is due run eagerly.  Data-parallel runs use the same line with the exchanges in it.  What the
engine does not cover falls back to the autograd path in trainer.py, unchanged: dropout > 0, activations other than ReLU,
optimisers other than Adam/AdamW, --strict_nan_check (the reference's per-step host sync).
"""
import os

import torch
from torch import nn

from . import ops
from .quantize import level_plan, quantize_values

_ALIGN = 64     # floats: every parameter starts on a 256-byte boundary of the flat buffers


# K-runs per weight gradient in the step's grouped launch: 1 = one chain over the batch (0 = lcrec_linear_backward_splits' per-layer
# rule; LCREC_DW_SPLITS, tuning)
DW_SPLITS = int(os.environ.get("LCREC_DW_SPLITS", "1"))


class TrainEngine:
    def __init__(self, model, optimizer, schedule, warmup_steps, total_steps, max_norm=1.0, use_graph=True, use_ema=False,
                 dist=None, dp_graph="auto", fuse_bn=None):
        """schedule: "linear" | "constant" (index/trainer.py:83-92) or None (fixed learning rate).
        use_ema: the improve fork's EMA codebook update after every step (index_improve/trainer.py:119).
        dist: an enabled dist.DistContext for item-sharded data parallel (the caller sets dist.set_batch before a step).
        dp_graph: "on" captures the data-parallel step with its RCCL collectives; "off" launches the same line eagerly;
        "auto" = on for a one-rank group (run in the GPU suite), off for more ranks -- UNVERIFIED on more than one GPU."""
        self.model = model
        # BatchNorm folded into the GEMMs on either side of it (lcrec_linear_bn_forward: 56 launches per step instead of 68).
        # OFF by default: measured slower on MI355X (1.28 vs 1.18 ms per step at the run.sh shape) -- a hand-over between
        # workgroups inside a launch costs as much as the launch it saves.  LCREC_FUSE_BN=1 turns it on.
        self.fuse_bn = (__import__("os").environ.get("LCREC_FUSE_BN", "0") == "1") if fuse_bn is None else bool(fuse_bn)
        self.dist = dist if (dist is not None and dist.enabled) else None
        if self.dist is not None:
            import torch.distributed as tdist
            use_graph = use_graph and tdist.get_backend() == "nccl"      # gloo collectives synchronise the host
            if dp_graph == "off" or (dp_graph == "auto" and self.dist.world_size > 1):
                use_graph = False
        self.collectives = 0                                                 # collectives issued or captured (tests, logs)
        self.ema_levels = [q for q in model.rq.vq_layers if use_ema and q.ema_decay is not None]
        self.optimizer = optimizer
        self.max_norm = float(max_norm)
        self.use_graph = use_graph
        self.schedule = {"linear": 1, "constant": 0, None: -1}[schedule]
        self.warmup_steps, self.total_steps = int(warmup_steps), int(total_steps)
        group = optimizer.param_groups[0]
        self.base_lr = float(group.get("initial_lr", group["lr"]))
        self.betas = tuple(group["betas"])
        self.eps = float(group["eps"])
        self.weight_decay = float(group["weight_decay"])
        self.decoupled = isinstance(optimizer, torch.optim.AdamW)
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        self.device = self.params[0].device
        self._prior_steps = 0
        self._flatten()
        dev = self.device
        self.step_count = torch.full((), self._prior_steps, dtype=torch.int64, device=dev)   # optimiser steps taken (device side)
        self.host_steps = self._prior_steps
        self.clip = torch.zeros(2, dtype=torch.float32, device=dev)          # (grad norm, clip coefficient) of the last step
        self.lr_used = torch.zeros((), dtype=torch.float32, device=dev)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)          # sum of losses / of reconstruction losses (epoch)
        self.last = torch.zeros(3, dtype=torch.float32, device=dev)          # loss, recon, rq_loss of the last step
        self.bad = torch.zeros(2, dtype=torch.bool, device=dev)              # [loss was NaN, Sinkhorn solver gave up]
        self._graphs = {}                                                    # batch rows -> (graph, static input)
        self._seen = {}                                                      # batch rows -> eager steps done at that size
        self.graph_replays = 0
        self.last_idx = None
        self._bn_counters = None                                             # BatchNorm1d.num_batches_tracked tensors ...
        self._bn_pending = 0                                                 # ... and the steps not yet added to them
        # whoever reads the state dict (a checkpoint, a test) sees the host-side bookkeeping brought up to date first
        model.register_state_dict_pre_hook(lambda _m, _prefix, _keep: self.sync_host_state())

    # ------------------------------------------------------------------ support matrix
    @staticmethod
    def unsupported_reason(model, optimizer, args=None, dist=None, use_ema=False):
        if not isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)):
            return f"optimizer {type(optimizer).__name__}"
        if len(optimizer.param_groups) != 1 or optimizer.param_groups[0].get("amsgrad") or optimizer.param_groups[0].get("maximize"):
            return "optimizer options"
        if args is not None and getattr(args, "strict_nan_check", False):
            return "--strict_nan_check"
        p0 = next(model.parameters())
        if not p0.is_cuda:
            return "model is not on a HIP device"
        if model.loss_type not in ("mse", "l1"):
            return f"loss_type {model.loss_type}"
        if model.e_dim not in (16, 32, 64):
            return f"e_dim {model.e_dim}"
        for mlp in (model.encoder, model.decoder):
            if mlp.dropout > 0:
                return "dropout"
            if not mlp.fusable():
                return "non-ReLU activation"
            for g in mlp._groups:
                lin = mlp.mlp_layers[g["linear"]]
                if lin.bias is None or lin.in_features % 4 or lin.out_features % 4:
                    return "Linear shape"
                if "bn" in g:
                    bn = mlp.mlp_layers[g["bn"]]
                    if type(bn) is not nn.BatchNorm1d or bn.momentum is None or not bn.track_running_stats or not bn.affine:
                        return "BatchNorm options"
        return None

    # ------------------------------------------------------------------ flat buffers
    def _flatten(self):
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        dev = self.device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._step_f32 = torch.zeros((), dtype=torch.float32, device=dev)
        self.grad_view = {}
        # spans of the flat buffers for the data-parallel gradient exchange: the decoder's parameters (contiguous in
        # parameter order) and whatever lies before / after them
        dec = {id(p) for p in self.model.decoder.parameters()}
        where = [i for i, p in enumerate(self.params) if id(p) in dec]
        if where and where == list(range(where[0], where[-1] + 1)):
            lo = offs[where[0]]
            hi = offs[where[-1] + 1] if where[-1] + 1 < len(offs) else total
            self._late_span = (lo, hi)
            self._early_spans = [(a, b) for a, b in ((0, lo), (hi, total)) if b > a]
        else:
            self._late_span, self._early_spans = None, [(0, total)]
        with torch.no_grad():
            for p, off in zip(self.params, offs):
                n = p.numel()
                view = lambda flat: flat[off:off + n].view(p.shape)
                view(self.flat_p).copy_(p.data)
                p.data = view(self.flat_p)
                p.grad = view(self.flat_g)
                self.grad_view[p] = p.grad
                state = self.optimizer.state[p]
                if "exp_avg" in state:                                   # steps were taken before the engine took over
                    view(self.flat_m).copy_(state["exp_avg"])
                    view(self.flat_v).copy_(state["exp_avg_sq"])
                    self._prior_steps = int(float(state["step"]))
                state["step"] = self._step_f32
                state["exp_avg"] = view(self.flat_m)
                state["exp_avg_sq"] = view(self.flat_v)

    # ------------------------------------------------------------------ the step, as library calls
    def _mlp_forward(self, mlp, h):
        """One MLP going forward.  Per layer `saved` gets (layer input, Linear, BatchNorm | None, relu, t, y, mean, rstd,
        in_fold, out_fold): with FUSED BatchNorm (single process, batch-sized launches -- lcrec_linear_bn_forward) no
        activation y is ever written: a layer's input is the previous layer's pre-BatchNorm output t plus that BatchNorm's
        folded (scale, shift) = in_fold, applied by the consumers' operand staging; out_fold is this layer's own."""
        saved = []
        mods = mlp.mlp_layers
        groups = mlp._groups
        fused = self.fuse_bn and self.dist is None and all(
            ops.linear_bn_supported(h.shape[0], mods[g["linear"]].in_features, mods[g["linear"]].out_features) for g in groups)
        fold = None                                                          # (scale, shift, relu) of the tensor `h` stands for
        for g in groups:
            lin = mods[g["linear"]]
            relu = "act" in g
            if fused:
                bn = mods[g["bn"]] if "bn" in g else None
                bn_args = None if bn is None else (bn.weight.data, bn.bias.data, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
                if fold is None and bn is None:
                    t = ops.linear_forward(h, lin.weight.data, lin.bias.data, relu=relu)       # a plain layer: the ordinary kernel
                    stats = None
                else:
                    if bn is None and relu:
                        raise ops._lib.LcrecError("engine: a ReLU without BatchNorm after a BatchNorm layer is not a shape MLPLayers builds")
                    t, stats = ops.linear_bn_forward(h, lin.weight.data, lin.bias.data, in_fold=None if fold is None else fold[:2],
                                                     in_relu=bool(fold and fold[2]), bn=bn_args)
                out_fold = None if stats is None else (stats[2], stats[3], relu)
                saved.append((h, lin, bn, relu, t, None if bn is not None else t, None if stats is None else stats[0],
                              None if stats is None else stats[1], fold, out_fold))
                h, fold = t, out_fold
                continue
            if "bn" in g:
                bn = mods[g["bn"]]
                t = ops.linear_forward(h, lin.weight.data, lin.bias.data, relu=False)
                if self.dist is not None:
                    from .layers import sharded_bn_statistics
                    mean, rstd, _ = sharded_bn_statistics(self.dist, t, bn)          # statistics of the GLOBAL batch
                    y = ops.bn_relu_apply(t, bn.weight.data, bn.bias.data, mean, rstd, relu)
                    self.collectives += 1
                else:
                    y, mean, rstd = ops.bn_relu_forward(t, bn.weight.data, bn.bias.data, bn.eps, bn.momentum, bn.running_mean,
                                                        bn.running_var, relu=relu)
                saved.append((h, lin, bn, relu, t, y, mean, rstd, None, None))
            else:
                y = ops.linear_forward(h, lin.weight.data, lin.bias.data, relu=relu)
                saved.append((h, lin, None, relu, None, y, None, None, None, None))
            h = y
        if fold is not None:
            raise ops._lib.LcrecError("engine: the last layer of an MLP has no BatchNorm (layers.py:19-30)")
        return h, saved

    def _mlp_backward(self, saved, g, need_input_grad, dw, last_bias_done=False):
        """The dX chain of one MLP: BatchNorm/ReLU backward -> dX GEMM -> next layer.  Weight gradients are leaves of the
        dependency graph and most of them are a handful of tiles: they are queued in `dw` as (dt, layer input, gradient
        view[, input fold]) and computed by ONE grouped launch at the end of the step (lcrec_linear_backward_weights)."""
        gv = self.grad_view
        for i in range(len(saved) - 1, -1, -1):
            h, lin, bn, relu, t, y, mean, rstd, in_fold, out_fold = saved[i]
            if bn is not None and self.dist is not None:
                sums = ops.bn_backward_reduce(g, t, y, mean, rstd, relu, dbeta_out=gv[bn.bias], dgamma_out=gv[bn.weight])
                self.dist.all_reduce_(sums)
                self.collectives += 1
                dt, _ = ops.bn_backward_apply(g, t, y, bn.weight.data, mean, rstd, sums, self.dist.batch_rows[1], relu,
                                              dbias_out=gv[lin.bias])
            elif bn is not None:
                # (the ReLU mask is recomputed from t with the forward kernel's own expression: y is not read again)
                recompute = relu and y is not None and bn.bias is not None
                dt, _, _, _ = ops.bn_relu_backward(g, t, None if recompute else y, bn.weight.data, mean, rstd, relu,
                                                   dgamma_out=gv[bn.weight], dbeta_out=gv[bn.bias], dbias_out=gv[lin.bias],
                                                   fold=None if y is not None else out_fold[:2],
                                                   beta=bn.bias.data if recompute else None)
            elif last_bias_done and i == len(saved) - 1 and not relu:
                dt = g                                   # g is the pre-activation gradient already and the bias gradient is written
            else:
                dt, _ = ops.relu_bias_backward(g, y, relu, dbias_out=gv[lin.bias], inplace=True)
            dw.append((dt, h, gv[lin.weight]) if in_fold is None else (dt, h, gv[lin.weight], in_fold))
            if i > 0 or need_input_grad:
                w = lin.weight.data
                pad = (-w.shape[0]) % 32                # layers._LinearAct.backward: K slice of the k-major dX kernel
                if pad:
                    g = ops.linear_backward(torch.nn.functional.pad(dt, (0, pad)), h, torch.nn.functional.pad(w, (0, 0, 0, pad)),
                                            True, False)[0]
                else:
                    g = ops.linear_backward(dt, h, w, True, False)[0]
            else:
                g = None
        return g

    def _run(self, x, eager):
        m = self.model
        levels = list(m.rq.vq_layers)
        z, enc = self._mlp_forward(m.encoder, x)
        if eager and any(not q.initted for q in levels):
            m.rq._lazy_kmeans(z.reshape(-1, m.e_dim), True)                 # vq.py:67-68, first training batch only
        cbs = [q.embedding.weight.data for q in levels]
        q = quantize_values(z, cbs, float(m.rq.beta), level_plan(levels, True), False, True, want_loss=False)
        self.last_idx = q["idx"]                                             # [rows, L] of the last step (a graph's static output)
        out, dec = self._mlp_forward(m.decoder, q["xq"])
        # (BatchNorm1d.num_batches_tracked is only ever read by checkpoints: counted on the host, written by sync_host_state)
        if self._bn_counters is None:
            self._bn_counters = [s[2].num_batches_tracked for s in enc + dec if s[2] is not None]
        world = self.dist
        n, e = z.shape
        works = []
        # the Sinkhorn solver's give-up marker (a -1 assignment): tested by lcrec_step_losses itself, no launch of its own
        probes = ops.deferred_checks.drain()
        probe = probes[0][1] if probes else None
        for _msg, extra in probes[1:]:                                       # (more than one Sinkhorn level: rare)
            self.bad[1].logical_or_(extra[0] < 0)
        if world is None:
            recon, g_out = ops.recon_loss_grad(out, x, m.loss_type)
            sse = q["sse"]
        else:
            # this rank's share of the global-batch losses; one all-reduce of L+1 doubles makes them the global ones
            n = world.batch_rows[1]
            share, g_out = ops.recon_loss_grad(out, x, m.loss_type, global_rows=n)
            both = torch.cat([q["sse"], share.double().reshape(1)])
            world.all_reduce_(both)
            self.collectives += 1
            sse, recon = both[:len(levels)], both[len(levels)].float()
        # level losses, their mean, the total loss, the epoch's running sums and the NaN flag: one launch
        ops.step_losses(sse, n, e, float(m.rq.beta), m.quant_loss_weight, recon, self.last, self.sums, self.bad[0],
                        poison_probe=probe, poison_flag=self.bad[1])
        dw = []
        g_xq = self._mlp_backward(dec, g_out, True, dw)
        if world is not None and self._late_span is not None:
            ops.linear_backward_weights(dw, splits=DW_SPLITS)                # the decoder's weight gradients, one launch ...
            dw = []
            lo, hi = self._late_span                                         # ... and its span of the flat buffer is on its way
            works.append(world.all_reduce_(self.flat_g[lo:hi], async_op=True))
            self.collectives += 1
        scale = 2.0 / (len(levels) * n * e)                                  # quantize.py: d mean-level-loss / d (sum of squares)
        # (the encoder's last Linear has neither BatchNorm nor activation behind it: what reaches z is the gradient of its
        # pre-activation, and its bias gradient -- the column sums -- comes out of the same launch)
        last = enc[-1]
        bias_here = last[2] is None and not last[3]
        gz = ops.quantizer_input_grad(z, cbs[0], q["idx"][:, 0], float(m.rq.beta) * scale, m.quant_loss_weight, g_xq,
                                      dbias_out=self.grad_view[last[1].bias] if bias_here else None)
        # per-code (count, sum) of every level and the codebook gradients (scale * (cnt*C - sum)) * g_loss: one launch
        stats = ops.code_stats_levels(q["idx"], q["resid_in"], [c.shape[0] for c in cbs], cbs,
                                      [self.grad_view[lvl.embedding.weight] for lvl in levels], scale, m.quant_loss_weight)
        self._mlp_backward(enc, gz, False, dw, last_bias_done=bias_here)
        ops.linear_backward_weights(dw, splits=DW_SPLITS)                    # all 14 weight gradients, one launch: 2 000 tiles, one chain each
        del dw
        if world is not None:
            for lo, hi in self._early_spans:
                works.append(world.all_reduce_(self.flat_g[lo:hi], async_op=True))
                self.collectives += 1
            if not eager:
                for lvl in self.ema_levels:                                  # (eager: VectorQuantizer.ema_step does it)
                    for stat in stats[levels.index(lvl)]:
                        world.all_reduce_(stat)
                        self.collectives += 1
            for w in works:
                w.wait()
        # improve fork: EMA statistics and codebook blend (index_improve/models/vq.py:147-193).  The reference does this
        # inside the forward; nothing between there and the optimizer reads the codebooks again (the gradients above were
        # formed from the pre-update values, as autograd's saved tensors are), so here -- after them -- is equivalent.
        for lvl in self.ema_levels:
            t = levels.index(lvl)
            if eager:
                lvl.ema_step(stats[t], q["resid_in"][t])                    # counts the step, re-seeds dead codes when due
            else:
                ops.ema_update(lvl._ema_cluster_size, lvl._ema_w, lvl.embedding.weight.data, stats[t][0], stats[t][1],
                               lvl.ema_decay, lvl.epsilon, skip_flag=self.bad[0])
        ops.grad_norm_clip(self.flat_g, self.max_norm, out=self.clip)
        ops.adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.step_count, self.base_lr, self.betas, self.eps,
                       self.weight_decay, self.decoupled, clip=self.clip, schedule=self.schedule,
                       warmup_steps=self.warmup_steps, total_steps=self.total_steps, lr_out=self.lr_used,
                       skip_flag=self.bad[0])            # a NaN loss (sticky) leaves parameters, moments and step at the last good step

    # ------------------------------------------------------------------ driving it
    def step_selected(self, data, index):
        """One training step on the rows `index` (an int64 device vector) of the device-resident matrix `data`: on a replayed
        step the gather writes into the captured graph's input buffer (one launch instead of a gather and a copy)."""
        rows = int(index.shape[0])
        key = rows if self.dist is None else (rows, getattr(self.dist, "batch_rows", (None, None))[1])
        if key in self._graphs:
            return self.step(None, source=(data, index))
        return self.step(data.index_select(0, index))

    def step(self, batch, source=None):
        """One training step on `batch` ([rows, in_dim] on the engine's device)."""
        rows = int(batch.shape[0]) if batch is not None else int(source[1].shape[0])
        if self.dist is not None:
            held = getattr(self.dist, "batch_rows", None)
            if held is None or held[0] != rows:     # noqa: E129
                raise ops._lib.LcrecError("data-parallel step: call dist.set_batch(local rows, global rows) first")
            rows = (rows, held[1])                                           # a graph per (local, global) batch shape
        self.host_steps += 1
        self._bn_pending += 1
        # a step on which a level's dead-code reset is due (host logic, random draws, data-dependent shapes) runs eagerly
        reset_due = any((q.step_count + 1) % q.reset_interval == 0 for q in self.ema_levels)
        entry = self._graphs.get(rows)
        if batch is None and (entry is None or reset_due):
            batch, source = source[0].index_select(0, source[1]), None
        if entry is not None and not reset_due:
            if source is not None:
                torch.index_select(source[0], 0, source[1], out=entry[1])    # the batch is gathered straight into the graph's input
            else:
                entry[1].copy_(batch)
            entry[0].replay()
            self.graph_replays += 1
            for q in self.ema_levels:
                q.step_count += 1
            return
        done = self._seen.get(rows, 0)
        lazy = any(not q.initted for q in self.model.rq.vq_layers)
        if not self.use_graph or done < 1 or lazy or reset_due:
            # the first step at a batch size runs eagerly: it creates the stream's workspaces, runs the one-off k-means
            # initialisation (host sklearn), and it is a real training step
            with torch.no_grad(), ops.deferred_checks(raw=True):
                self._run(batch.contiguous(), eager=True)
            self._seen[rows] = done + 1
            return
        static = batch.clone()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            ops._ticket(self.device, force=True)         # the capture stream's ticket word (zeroed here, outside the capture)
        # Data parallel: the process group's watchdog THREAD polls the events of collectives issued before the capture
        # (hipEventQuery); in the default "global" capture mode such a call from any thread is an error that invalidates the
        # capture -- and kills the watchdog, which takes the process down (seen once in ~10 runs).  "thread_local" confines the
        # check to this thread, whose own calls are all capturable.
        mode = "thread_local" if self.dist is not None else "global"
        if self.dist is not None:
            torch.cuda.synchronize(self.device)          # (and nothing of the eager steps' collectives is left for it to poll)
        failure = None
        try:
            with torch.no_grad(), ops.deferred_checks(raw=True):
                with torch.cuda.graph(graph, stream=side, capture_error_mode=mode):
                    self._run(static, eager=False)
        except RuntimeError as err:
            if self.dist is None:
                raise
            failure = err
        if self.dist is not None:
            # a collective backend that cannot be captured: keep the straight line, launched eagerly.  Every rank must
            # take the same path -- a rank replaying a graph while a peer launches eagerly would still pair up collective
            # for collective, but one rank falling back alone is not a state worth reasoning about -- so the ranks agree
            # on the outcome (a capture records, it does not execute: nobody is inside a collective here)
            torch.cuda.synchronize(self.device)
            ok = torch.tensor([0.0 if failure is not None else 1.0], dtype=torch.float32, device=self.device)
            self.dist.all_reduce_(ok)
            if float(ok.item()) < self.dist.world_size:
                import logging
                logging.getLogger().warning("lcrec_amd.engine: data-parallel step not capturable on every rank (%s); running "
                                            "it eagerly", failure if failure is not None else "a peer's capture failed")
                self.use_graph = False
                del graph
                with torch.no_grad(), ops.deferred_checks(raw=True):
                    self._run(batch.contiguous(), eager=True)
                self._seen[rows] = done + 1
                return
        self._graphs[rows] = (graph, static)
        graph.replay()                                                       # capture records, replay executes: this step
        self.graph_replays += 1
        for q in self.ema_levels:
            q.step_count += 1

    def release(self):
        """Drop the captured graphs (and their private memory pools) now, while the runtime is fully alive, rather than leaving
        them to interpreter teardown; the next step at a batch size captures again."""
        torch.cuda.synchronize(self.device)
        self._graphs.clear()

    def begin_epoch(self):
        self.sums.zero_()

    def end_epoch(self, scheduler=None):
        """(sum of losses, sum of reconstruction losses) over the epoch's steps -- trainer.py:122-125 -- after checking
        the device-side flags; brings the host-side scheduler and optimizer bookkeeping up to date."""
        vals = self.sums.cpu()
        bad = self.bad.cpu()
        if bool(bad[1]):
            raise ops._lib.LcrecError("lcrec_sinkhorn_assign: grid barrier timed out (device oversubscribed?); "
                                      "set LCREC_SINKHORN_PERSISTENT=0 to use the multi-launch solver")
        if bool(bad[0]):
            raise ValueError("Training loss is nan")
        self.sync_host_state(scheduler)
        return float(vals[0]), float(vals[1])

    def sync_host_state(self, scheduler=None):
        """Make what the host can see agree with the device: the optimizer's per-parameter `step` entries (checkpoints)
        and the LambdaLR scheduler's counters and `lr` (logging, get_last_lr)."""
        self._step_f32.fill_(float(self.host_steps))
        if self._bn_pending and self._bn_counters:
            torch._foreach_add_(self._bn_counters, self._bn_pending)
        self._bn_pending = 0
        if scheduler is not None and scheduler.last_epoch != self.host_steps:
            scheduler.last_epoch = self.host_steps
            scheduler._step_count = self.host_steps + 1
            lrs = [base * lmbda(self.host_steps) for base, lmbda in zip(scheduler.base_lrs, scheduler.lr_lambdas)]
            for group, lr in zip(self.optimizer.param_groups, lrs):
                group["lr"] = lr
            scheduler._last_lr = lrs
