"""`Trainer().fit(model, tasks)`: the user-facing training entry point (reference: engine/trainer.py:60-140 `Trainer.fit`,
:737-979 `train_step`; scripts/trainer_api.py:1-27).  A thin facade over what the hot path needs:

    dictionary <- tasks           task.initialize(global_dict)                      (trainer.py:118-121)
    adaptors   <- instructions    Task.upgrade_model_adaptor_cfg(tasks, model.cfg)  (trainer.py:124)
    model.initialize(global_dict) ; model -> GPU, bf16 (fp32 on request)            (trainer.py:125, 214-221)
    every update: one micro-batch list per task -> TrainStep.train_step             (trainer.py:747-884)
    lr: polynomial decay with linear warm-up (default_trainer.yaml:27-28, lr_scheduler/ofa_polynomial_decay)

Out of scope: CLI / hydra config tree, checkpoints, metrics / logging back-ends, validation loops, EMA,
FSDP / BMUF.  One process per GPU: when WORLD_SIZE > 1 (torch.distributed.run) the process group is initialised on RCCL.
"""
import logging
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .preprocessor import Dictionary, to_device
from .task import Task
from .trainer import TrainStep

logger = logging.getLogger("ofasys_amd")


@dataclass
class CommonConfig:
    seed: int = 1
    bf16: bool = True              # the MI355X build defaults to bf16 (same 8-bit exponent as fp32: no loss scaler needed);
    fp32: bool = False             # fp32 on request;
    fp16: bool = False             # fp16 + the reference's dynamic loss scaler: its own default (default_trainer.yaml:7-25)
    fp16_init_scale: int = 128     # (default_trainer.yaml:9-13, engine/optim/dynamic_loss_scaler.py)
    fp16_scale_window: Optional[int] = None
    fp16_scale_tolerance: float = 0.0
    min_loss_scale: float = 1e-4
    threshold_loss_scale: Optional[float] = None
    log_interval: int = 10
    use_graph: bool = True
    graph_pad_to_multiple: int = 16    # with use_graph: text batches are padded to a multiple of this (collate_tokens' own
                                       # pad_to_multiple, preprocessor/utils.py:75-113) so that a few hipGraphs cover the length
                                       # distribution and every DP rank meets the same few structures; 1 = the reference's padding


@dataclass
class OptimizationConfig:          # default_trainer.yaml:16-28
    max_update: int = 10000
    clip_norm: float = 1.0
    lr: List[float] = field(default_factory=lambda: [1e-5])
    warmup_ratio: float = 0.06
    end_learning_rate: float = 0.0
    power: float = 1.0


@dataclass
class OptimizerConfig:
    adam_betas: tuple = (0.9, 0.999)
    adam_eps: float = 1e-8
    weight_decay: float = 0.01


class _SkipPoll:
    """Host-side view of the device's skipped-update counter without a per-step sync: after every step the 2-element `skipped`
    tensor is copied to pinned memory behind an event; a read returns the newest copy whose event has completed."""

    def __init__(self):
        self._host = [torch.zeros(2, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._ev = [torch.cuda.Event() for _ in range(2)]
        self._busy = [False, False]
        self._value = 0
        self._turn = 0

    def push_and_read(self, skipped_dev):
        for i in range(2):
            if self._busy[i] and self._ev[i].query():
                self._value = max(self._value, int(self._host[i][1]))
                self._busy[i] = False
        i = self._turn
        if not self._busy[i]:
            self._host[i].copy_(skipped_dev, non_blocking=True)
            self._ev[i].record()
            self._busy[i] = True
            self._turn ^= 1
        return self._value


@dataclass
class TrainerConfig:
    common: CommonConfig = field(default_factory=CommonConfig)
    optimization: OptimizationConfig = field(default_factory=OptimizationConfig)
    optimizer: OptimizerConfig = field(default_factory=OptimizerConfig)


def polynomial_decay_lr(num_updates, base_lr, max_update, warmup_ratio, end_lr=0.0, power=1.0):
    """The learning rate the NEXT update runs with after `num_updates` completed ones (engine/lr/polynomial_decay_schedule.py):
    warmup_updates = max(int(total * ratio), 1) (`reinit`, :96-101); before the first update the rate is lr / warmup_updates
    (:104-106); then `step_update(num_updates)` (:81-93): linear warm-up while num_updates <= warmup_updates, polynomial decay to
    end_lr at total.  The trainer calls step_update AFTER an update (engine/trainer.py:932, 1091-1094), so update u uses
    step_update(u - 1) -- and a skipped update (overflow) does not advance num_updates."""
    if warmup_ratio > 0:
        warm = max(int(max_update * warmup_ratio), 1)
    else:
        warm = 0
    if num_updates <= 0:
        return base_lr / warm if warm > 0 else base_lr
    if warm > 0 and num_updates <= warm:
        return base_lr * num_updates / float(warm)
    if num_updates >= max_update:
        return end_lr
    frac = 1.0 - (num_updates - warm) / (max_update - warm)
    return (base_lr - end_lr) * frac ** power + end_lr


class Trainer:
    def __init__(self, cfg: Optional[TrainerConfig] = None, **overrides):
        """overrides: max_update=, lr=, clip_norm=, seed=, fp32=, use_graph=, log_interval=, weight_decay= (flat shortcuts)."""
        self.cfg = cfg or TrainerConfig()
        for k, v in overrides.items():
            for section in (self.cfg.common, self.cfg.optimization, self.cfg.optimizer):
                if hasattr(section, k):
                    setattr(section, k, [v] if k == "lr" and not isinstance(v, (list, tuple)) else v)
                    break
            else:
                raise TypeError(f"unknown trainer option {k}")
        self.global_dict = None
        self.step_engine: Optional[TrainStep] = None
        self.history = []

    # ------------------------------------------------------------------ set-up
    def _init_distributed(self):
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if world > 1 and not dist.is_initialized():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))      # nccl == RCCL on ROCm
        return rank, world, torch.device("cuda", local)

    def setup(self, model, tasks: List[Task]):
        if not torch.cuda.is_available():
            raise RuntimeError("ofasys_amd.Trainer needs an MI355X: the HIP path is the only path (no CPU fallback)")
        cfg = self.cfg
        rank, world, device = self._init_distributed()
        torch.manual_seed(cfg.common.seed)
        self.global_dict = Dictionary()
        for task in tasks:
            task.initialize(self.global_dict, is_train=True)
        Task.upgrade_model_adaptor_cfg(tasks, model.cfg)
        model.initialize(self.global_dict)
        model.to(device)
        half = torch.float16 if cfg.common.fp16 else torch.bfloat16
        if not cfg.common.fp32:
            model.to(half)
        if world > 1:
            import torch.distributed as dist
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        from . import ops
        ops.manual_seed(cfg.common.seed + rank)                     # dropout streams differ per rank (fairseq: seed + rank)
        if cfg.common.use_graph and cfg.common.graph_pad_to_multiple > 1:
            for task in tasks:
                if task.cfg.text.pad_to_multiple == 1:
                    task.cfg.text.pad_to_multiple = cfg.common.graph_pad_to_multiple
        crit = tasks[0].cfg.criterion
        for task in tasks[1:]:
            # the reference builds one criterion per task (task/base.py:213-216); the step engine holds one: refuse a mix it would
            # silently flatten
            c = task.cfg.criterion
            if (c.label_smoothing, c.drop_worst_ratio) != (crit.label_smoothing, crit.drop_worst_ratio):
                raise NotImplementedError("tasks with different criterion settings (label_smoothing / drop_worst_ratio) in one "
                                          "Trainer are not supported: the step engine applies one criterion to every micro-batch")
        self._update_freq = max(int(t.cfg.dataset.update_freq) for t in tasks)
        self.step_engine = TrainStep(model, lr=cfg.optimization.lr[0], betas=tuple(cfg.optimizer.adam_betas), eps=cfg.optimizer.adam_eps,
                                     weight_decay=cfg.optimizer.weight_decay, clip_norm=cfg.optimization.clip_norm,
                                     use_graph=cfg.common.use_graph, label_smoothing=crit.label_smoothing,
                                     drop_worst_ratio=crit.drop_worst_ratio, loss_scale=self._loss_scale_cfg(world))
        for task in tasks:
            task.init_data_iterator("train", rank, world)
        self._device, self._rank, self._world = device, rank, world
        self._float_dtype = torch.float32 if cfg.common.fp32 else half
        return self.step_engine

    def _loss_scale_cfg(self, world):
        """fp16: the reference builds its FP16Optimizer with a DynamicLossScaler (fp16_optimizer.py:262-285): scale_window defaults
        to 2**14 / world / update_freq; fp32 / bf16 train unscaled."""
        c = self.cfg.common
        if not c.fp16 or c.fp32:
            return None
        window = c.fp16_scale_window if c.fp16_scale_window is not None else \
            max(1, int(2 ** 14 / world / getattr(self, "_update_freq", 1)))
        return {"init_scale": float(c.fp16_init_scale), "scale_factor": 2.0, "scale_window": window,
                "tolerance": c.fp16_scale_tolerance, "threshold": c.threshold_loss_scale, "min_loss_scale": c.min_loss_scale}

    def _micro_batches(self, tasks):
        """samples[task][i] of engine/trainer.py:747-766, flattened: update_freq micro-batches of every task, on the device."""
        out = []
        for task in tasks:
            for _ in range(task.cfg.dataset.update_freq):
                s = to_device(task.get_sample("train"), self._device, self._float_dtype)
                mb = {"slots": s["net_input"]["slots"], "target": s["target"], "task": task.name}
                if s.get("constraint_masks") is not None:
                    mb["constraint_masks"] = s["constraint_masks"]
                out.append(mb)
        return out

    # ------------------------------------------------------------------ the loop
    def fit(self, model, tasks: List[Task]):
        cfg = self.cfg
        engine = self.setup(model, tasks)
        base_lr = cfg.optimization.lr[0]
        # The reference runs until num_updates -- which an overflow-skipped step does not advance -- reaches max_update
        # (engine/trainer.py:447-452, 957-960), so skipped steps are made up.  The device counts them (`skipped`[1]); the host follows
        # that counter through a pinned, event-guarded copy enqueued after every step (no sync: the schedule lags a skip by at most
        # one step) and reads it exactly at log time and before leaving the loop.
        poll = _SkipPoll() if self._device.type == "cuda" else None
        skipped, attempts, max_update = 0, 0, cfg.optimization.max_update
        while attempts - skipped < max_update:
            attempts += 1
            update = attempts - skipped                              # the update this attempt will be if it is not skipped
            engine.lr = polynomial_decay_lr(update - 1, base_lr, max_update, cfg.optimization.warmup_ratio,
                                            cfg.optimization.end_learning_rate, cfg.optimization.power)
            out = engine.train_step(self._micro_batches(tasks))
            if poll is not None:
                skipped = max(skipped, poll.push_and_read(out["skipped"]))
            last = attempts - skipped >= max_update
            if attempts % cfg.common.log_interval == 0 or last:
                skipped = int(float(out["skipped"][1]))              # (one sync, with the log line's own)
                engine.check()                                       # FloatingPointError on Nan/Inf gradients (trainer.py:866-876)
                n, loss = float(out["stats"][0]), float(out["stats"][1])
                rec = {"update": attempts - skipped, "loss": loss / max(n, 1.0) / 0.6931471805599453, "sample_size": n,
                       "gnorm": float(out["gnorm"]), "lr": engine.lr}
                self.history.append(rec)
                if self._rank == 0:
                    logger.info("update %(update)d | loss %(loss).4f (base 2 per token) | sample_size %(sample_size)d | "
                                "gnorm %(gnorm).3f | lr %(lr).3g", rec)
            if attempts > 4 * max_update + 64:                       # every step skipped and no scaler floor to stop at
                raise FloatingPointError(f"{skipped} of {attempts} updates were skipped on the device (non-finite gradients)")
        torch.cuda.synchronize()
        return self.history
