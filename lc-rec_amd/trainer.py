"""RQ-VAE training harness -- host-side mirror of the reference's index/trainer.py
(Trainer :14-251) and of index_improve/trainer.py's additions (use_ema forward, codebook
utilisation in the evaluation line).  Behaviour kept on purpose, because people and scripts
depend on it:
  * `train loss` / `reconstruction loss` are SUMS over the epoch's batches (trainer.py:122-125);
  * best_loss is only compared on evaluation epochs (:207), cur_eval_step is counted but never
    stops training (:189,213,217);
  * checkpoint dict keys, file names and the newest-N + best-N retention (:154-172, :231-247);
  * log line formats (:174-184, :221-230).
What differs is where the work happens: batches are slices of an HBM-resident matrix, the step
runs the HIP kernels through the module API, and the collision rate is a device sort
(lcrec_collision_groups) instead of a Python string set.
"""
import heapq
import logging
import os
from time import time

import numpy as np
import torch
from torch import optim
from tqdm import tqdm

from . import ops
from .utils import delete_file, ensure_dir, get_local_time, set_color


def linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """The multiplier the reference gets from transformers.get_linear_schedule_with_warmup
    (trainer.py:85-87): ramp 0 -> 1 over the warm-up, then linearly to 0 at num_training_steps."""
    def factor(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch)


def constant_schedule_with_warmup(optimizer, num_warmup_steps, last_epoch=-1):
    """transformers.get_constant_schedule_with_warmup (trainer.py:89-90): ramp 0 -> 1, then 1."""
    def factor(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1.0, num_warmup_steps))
        return 1.0
    return optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch)


class CheckpointKeeper:
    """Retention rule of trainer.py:231-247: keep the `limit` newest evaluation checkpoints plus the
    `limit` with the lowest collision rate; a file leaves the disk when it is in neither set."""

    def __init__(self, limit, remove=delete_file):
        self.limit = limit
        self.remove = remove
        self.best = []      # min-heap of (-collision_rate, path): root = worst of the kept best
        self.newest = []    # FIFO

    def add(self, collision_rate, path):
        entry = (-collision_rate, path)
        if len(self.newest) < self.limit:
            self.newest.append(entry)
            heapq.heappush(self.best, entry)
            return
        oldest = self.newest.pop(0)
        self.newest.append(entry)
        if collision_rate < -self.best[0][0]:
            dropped = heapq.heappop(self.best)
            heapq.heappush(self.best, entry)
            if dropped not in self.newest:
                self.remove(dropped[1])
        if oldest not in self.best:
            self.remove(oldest[1])


class Trainer(object):

    def __init__(self, args, model, data_num):
        self.args = args
        self.model = model
        self.logger = logging.getLogger()

        self.lr = args.lr
        self.learner = args.learner
        self.lr_scheduler_type = args.lr_scheduler_type
        self.weight_decay = args.weight_decay
        self.epochs = args.epochs
        self.warmup_steps = args.warmup_epochs * data_num
        self.max_steps = args.epochs * data_num

        self.save_limit = args.save_limit
        self.keeper = CheckpointKeeper(self.save_limit)
        self.eval_step = min(args.eval_step, self.epochs)
        self.device = torch.device(args.device)
        self.ckpt_dir = os.path.join(args.ckpt_dir, "{}".format(get_local_time()))
        ensure_dir(self.ckpt_dir)

        self.best_loss = np.inf
        self.best_collision_rate = np.inf
        self.best_loss_ckpt = "best_loss_model.pth"
        self.best_collision_ckpt = "best_collision_model.pth"
        self.use_ema = getattr(args, "ema_decay", None) is not None
        self.optimizer = self._build_optimizer()
        self.scheduler = self._get_scheduler()
        self.model = self.model.to(self.device)
        self.dist = None            # set by lcrec_amd.dist.attach() for item-sharded data parallel
        self.engine = None          # engine.TrainEngine once the first epoch has decided whether it applies
        self._engine_decided = False

    # reference attribute names, for scripts that poke at them
    @property
    def best_save_heap(self):
        return self.keeper.best

    @property
    def newest_save_queue(self):
        return self.keeper.newest

    def _build_optimizer(self):
        params = self.model.parameters()
        name = self.learner.lower()
        lr, wd = self.lr, self.weight_decay
        # same update rule as the reference's default (foreach) implementation, in one kernel per step
        fused = {"fused": True} if torch.device(self.args.device).type == "cuda" else {}
        if name == "adam":
            return optim.Adam(params, lr=lr, weight_decay=wd, **fused)
        if name == "sgd":
            return optim.SGD(params, lr=lr, weight_decay=wd)
        if name == "adagrad":
            opt = optim.Adagrad(params, lr=lr, weight_decay=wd)
            for state in opt.state.values():
                for k, v in state.items():
                    if torch.is_tensor(v):
                        state[k] = v.to(self.device)
            return opt
        if name == "rmsprop":
            return optim.RMSprop(params, lr=lr, weight_decay=wd)
        if name == "adamw":
            return optim.AdamW(params, lr=lr, weight_decay=wd, **fused)
        self.logger.warning("Received unrecognized optimizer, set default Adam optimizer")
        return optim.Adam(params, lr=lr)

    def _get_scheduler(self):
        if self.lr_scheduler_type.lower() == "linear":
            return linear_schedule_with_warmup(self.optimizer, self.warmup_steps, self.max_steps)
        return constant_schedule_with_warmup(self.optimizer, self.warmup_steps)

    def _check_nan(self, loss):
        if torch.isnan(loss):
            raise ValueError("Training loss is nan")

    def _check_nan_async(self, loss):
        """trainer.py:40-42's check without stalling the launch pipeline: the flag of step t is copied to pinned
        host memory asynchronously and read when step t+1 gets here, so the ValueError is raised one step later
        than the reference raises it (the epoch loop's exit checks the last step)."""
        if self.device.type != "cuda" or getattr(self.args, "strict_nan_check", False):
            return self._check_nan(loss)           # the reference's timing: raise before this step's backward
        if getattr(self, "_nan_host", None) is None:
            self._nan_host = torch.zeros((), dtype=torch.bool).pin_memory()
            self._nan_event = torch.cuda.Event()
            self._nan_pending = False
        if self._nan_pending:
            self._nan_event.synchronize()          # recorded a whole step ago: already complete
            if bool(self._nan_host):
                self._nan_pending = False
                raise ValueError("Training loss is nan")
        self._nan_host.copy_(torch.isnan(loss.detach()), non_blocking=True)
        self._nan_event.record()
        self._nan_pending = True

    def _check_nan_drain(self):
        if getattr(self, "_nan_pending", False):
            self._nan_pending = False
            self._nan_event.synchronize()
            if bool(self._nan_host):
                raise ValueError("Training loss is nan")

    def _get_engine(self):
        """The graph-captured step (engine.py) when the configuration allows it; None -> the autograd path below."""
        if not self._engine_decided:
            self._engine_decided = True
            from .engine import TrainEngine
            reason = TrainEngine.unsupported_reason(self.model, self.optimizer, self.args, self.dist, self.use_ema)
            if reason is None and (os.environ.get("LCREC_TRAIN_ENGINE", "1") == "0"
                                   or getattr(self.args, "train_engine", "auto") == "off"):
                reason = "switched off (--train_engine off / LCREC_TRAIN_ENGINE=0)"
            if reason is None:
                kind = "linear" if self.lr_scheduler_type.lower() == "linear" else "constant"
                self.engine = TrainEngine(self.model, self.optimizer, kind, self.warmup_steps, self.max_steps,
                                          use_ema=self.use_ema, dist=self.dist,
                                          dp_graph=getattr(self.args, "dp_graph", "auto"))
                self.logger.info("training step: %s (lcrec_amd.engine)", "one captured hipGraph per batch size"
                                 if self.engine.use_graph else "the engine's straight line, launched eagerly")
            else:
                self.logger.info("training step: autograd path (%s)", reason)
        return self.engine

    def _reducer(self):
        """The autograd path's bucketed gradient all-reduce (dist.GradReducer), built on first use."""
        if getattr(self, "grad_reducer", None) is None:
            from .dist import GradReducer
            self.grad_reducer = GradReducer(self.dist, list(self.model.parameters()))
        return self.grad_reducer

    def _set_global_batch(self, loader, data):
        """Tell the distributed context this rank's and the global batch's row counts (the loader knows both without a
        collective).  False: a global batch with fewer rows than ranks, which leaves some rank empty -- skipped on every
        rank (the only deviation from the single-process epoch; the reference itself cannot train BatchNorm on 1 row)."""
        n_local = int(data.shape[0])
        n_global = getattr(loader, "last_global_rows", None)
        if n_global is None or self.dist.world_size == 1:
            n_global = self.dist.sum_int(n_local)
        if n_global < self.dist.world_size:
            if not getattr(self, "_warned_small", False):
                self._warned_small = True
                self.logger.warning("global batch of %d rows on %d ranks: skipped", n_global, self.dist.world_size)
            return False
        self.dist.set_batch(n_local, n_global)
        return True

    def _train_epoch(self, train_data, epoch_idx):
        self.model.train()
        engine = self._get_engine()
        if engine is not None:
            # a device-resident loader hands out (matrix, row indices): the captured step gathers straight into its input
            selections = getattr(train_data, "iter_selections", None)
            iter_data = tqdm(selections() if selections else train_data, total=len(train_data), ncols=100,
                             desc=set_color(f"Train {epoch_idx}", "pink"), disable=not self._is_main())
            engine.begin_epoch()
            for data in iter_data:
                rows = data[1] if selections else data
                if self.dist is not None and not self._set_global_batch(train_data, rows):
                    continue
                if selections:
                    engine.step_selected(data[0], data[1])
                else:
                    engine.step(data.to(self.device))
            return engine.end_epoch(self.scheduler)      # raises "Training loss is nan" / solver errors of the epoch
        total_loss = torch.zeros((), dtype=torch.float64, device=self.device)
        total_recon = torch.zeros((), dtype=torch.float64, device=self.device)
        iter_data = tqdm(train_data, total=len(train_data), ncols=100, desc=set_color(f"Train {epoch_idx}", "pink"),
                         disable=not self._is_main())
        params = list(self.model.parameters())          # one walk of the module tree per epoch, not per step
        # no host synchronisation inside a step: the launch queue stays a step ahead of the GPU
        reducer = self._reducer() if self.dist is not None else None
        with ops.deferred_checks() as checks:
            for data in iter_data:
                data = data.to(self.device)
                if reducer is not None:
                    if not self._set_global_batch(train_data, data):
                        continue
                    n_local, n_global = self.dist.batch_rows
                    reducer.begin()                   # zeroes the flat gradient buffer the .grad views live in
                else:
                    self.optimizer.zero_grad()
                if self.use_ema:
                    out, rq_loss, _ = self.model(data, use_ema=True)
                else:
                    out, rq_loss, _ = self.model(data)
                loss, loss_recon = self.model.compute_loss(out, rq_loss, xs=data)
                if self.dist is not None:
                    # losses of the GLOBAL batch (what gets logged), and the NaN check on them: every rank sees the same
                    # value, so every rank raises in the same step instead of leaving the others inside a collective
                    both = self.dist.global_means(torch.stack([loss.detach(), loss_recon.detach()]), data.shape[0])
                    self._check_nan_async(both[0])
                    if reducer is not None:
                        (loss * (n_local / n_global)).backward()      # sum over ranks = gradient of the global mean loss
                        reducer.finish()
                    else:
                        loss.backward()
                        self.dist.reduce_gradients(self.model, n_local=data.shape[0])
                    total_loss += both[0]
                    total_recon += both[1]
                else:
                    self._check_nan_async(loss)
                    loss.backward()
                    # same values as `+= loss.item()` (fp32 -> double, summed in order) without a host sync per step
                    total_loss += loss.detach().double()
                    total_recon += loss_recon.detach().double()
                torch.nn.utils.clip_grad_norm_(params, 1.0)
                self.optimizer.step()
                self.scheduler.step()
                checks.poll()
            self._check_nan_drain()
        return total_loss.item(), total_recon.item()

    @torch.no_grad()
    def _valid_epoch(self, valid_data):
        self.model.eval()
        iter_data = tqdm(valid_data, total=len(valid_data), ncols=100, desc=set_color("Evaluate   ", "pink"),
                         disable=not self._is_main())
        chunks = []
        for data in iter_data:
            data = data.to(self.device)
            indices = self.model.get_indices(data)
            chunks.append(indices.view(-1, indices.shape[-1]))
        indices = torch.cat(chunks)
        if self.dist is not None:
            indices = self.dist.gather_rows(indices)
        ks = [q.n_e for q in self.model.rq.vq_layers]
        num_sample = indices.shape[0]
        unique = ops.collision_groups(indices, ks, want_groups=False)["unique"]
        return (num_sample - unique) / num_sample

    def _get_codebook_utilization(self):
        """index_improve/trainer.py:162-171."""
        try:
            stats = self.model.get_codebook_usage()
            return np.mean([s["utilization"] for s in stats]), stats
        except AttributeError:
            return None, None

    def _save_checkpoint(self, epoch, collision_rate=1, ckpt_file=None):
        ckpt_path = os.path.join(self.ckpt_dir, ckpt_file) if ckpt_file \
            else os.path.join(self.ckpt_dir, "epoch_%d_collision_%.4f_model.pth" % (epoch, collision_rate))
        if self._is_main():
            state = {
                "args": self.args,
                "epoch": epoch,
                "best_loss": self.best_loss,
                "best_collision_rate": self.best_collision_rate,
                "state_dict": self.model.state_dict(),
                "optimizer": self.optimizer.state_dict(),
            }
            torch.save(state, ckpt_path, pickle_protocol=4)
            self.logger.info(set_color("Saving current", "blue") + f": {ckpt_path}")
        return ckpt_path

    def _generate_train_loss_output(self, epoch_idx, s_time, e_time, loss, recon_loss):
        out = (set_color("epoch %d training", "green") + " [" + set_color("time", "blue") + ": %.2fs, ") \
            % (epoch_idx, e_time - s_time)
        out += set_color("train loss", "blue") + ": %.4f" % loss
        out += ", "
        out += set_color("reconstruction loss", "blue") + ": %.4f" % recon_loss
        return out + "]"

    def _is_main(self):
        return self.dist is None or self.dist.rank == 0

    def fit(self, data):
        cur_eval_step = 0
        for epoch_idx in range(self.epochs):
            t0 = time()
            train_loss, train_recon_loss = self._train_epoch(data, epoch_idx)
            t1 = time()
            self.logger.info(self._generate_train_loss_output(epoch_idx, t0, t1, train_loss, train_recon_loss))

            if (epoch_idx + 1) % self.eval_step == 0:
                v0 = time()
                collision_rate = self._valid_epoch(data)
                avg_util, usage = self._get_codebook_utilization() if self.use_ema else (None, None)

                if train_loss < self.best_loss:
                    self.best_loss = train_loss
                    self._save_checkpoint(epoch=epoch_idx, ckpt_file=self.best_loss_ckpt)
                if collision_rate < self.best_collision_rate:
                    self.best_collision_rate = collision_rate
                    cur_eval_step = 0
                    self._save_checkpoint(epoch_idx, collision_rate=collision_rate, ckpt_file=self.best_collision_ckpt)
                else:
                    cur_eval_step += 1

                v1 = time()
                if avg_util is None:
                    line = (set_color("epoch %d evaluating", "green") + " [" + set_color("time", "blue") + ": %.2fs, "
                            + set_color("collision_rate", "blue") + ": %f]") % (epoch_idx, v1 - v0, collision_rate)
                else:   # index_improve/trainer.py:239-253
                    line = (set_color("epoch %d evaluating", "green") + " [" + set_color("time", "blue") + ": %.2fs, "
                            + set_color("collision_rate", "blue") + ": %.4f") % (epoch_idx, v1 - v0, collision_rate)
                    line += ", " + set_color("codebook_utilization", "blue") + ": %.4f" % avg_util
                    for s in usage:
                        line += (f"\n  Quantizer {s['quantizer_id']}: {s['utilization']:.4f} "
                                 f"({s['used_codes']}/{s['total_codes']})")
                    line += "]"
                self.logger.info(line)

                ckpt_path = self._save_checkpoint(epoch_idx, collision_rate=collision_rate)
                if self._is_main():
                    self.keeper.add(collision_rate, ckpt_path)
        if self.engine is not None:
            self.engine.release()            # captured graphs go now, in order, not at interpreter teardown
        return self.best_loss, self.best_collision_rate
