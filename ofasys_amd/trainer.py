"""Train-step harness around the hot path (reference: engine/trainer.py:737-979 `Trainer.train_step`,
engine/criterion/cross_entropy.py:50-67, engine/optim/fp16_optimizer.py + adam.py).  Only the arithmetic the measured
step needs is here -- the reference's CLI, checkpointing, logging and data loading are out of scope.

Step arithmetic reproduced:
    for task in tasks: for micro-batch:  loss = CE_sum(model(slots), target); loss.backward()   (grads accumulate)
    grads   <- all_reduce_sum(grads)                      (DDP: bucketed, overlapped with the last backward)
    n       <- all_reduce_sum(sum of sample_size)         (sample_size = non-pad target tokens)
    grads   *= 1 / n                                      (== world/n after DDP's division by world, trainer.py:857-860)
    gnorm    = ||grads||_2 ; grads *= min(1, clip/(gnorm+1e-6))       (module/utils.py:342-384)
    Adam on the fp32 master copy, cast back to the model dtype        (fp16_optimizer.py:32-71, adam.py:192-212)
The multiply, the clip coefficient and Adam are ONE kernel pass over the flat arena (coef stays on the device: no host
sync anywhere in the step).
"""
import os
import warnings
from typing import List

import torch
import torch.utils._python_dispatch          # noqa: F401  (TorchDispatchMode: the capture audit's view of torch-native ops)

from . import kernels as K
from . import ops
from .distributed import GradBucketReducer, all_reduce_scalars
from .lib import capture_audit


class FlatParams:
    """All trainable parameters as views of one flat buffer (model dtype), with a matching flat gradient arena."""

    def __init__(self, model: torch.nn.Module):
        from .module.multihead_attention import MultiheadAttention
        # arena order = registration order, except that each attention's k|v|q (self) or k|v (cross) projection weights
        # and biases are made adjacent so ONE packed GEMM can use them (and their gradients) without any copy
        mha_of, mhas = {}, []
        for m in model.modules():
            if isinstance(m, MultiheadAttention) and m.q_proj.bias is not None:
                mhas.append(m)
                for p in (m.k_proj.weight, m.v_proj.weight, m.q_proj.weight, m.k_proj.bias, m.v_proj.bias, m.q_proj.bias):
                    mha_of[id(p)] = m
        # ... and the k|v projections of ALL the decoder layers' encoder-decoder attentions are made adjacent across layers: they
        # all read the encoder output, so the stack runs them as ONE GEMM (N = layers * 2D; ops.CrossKVShared)
        cross = [m for m in mhas if getattr(m, "encoder_decoder_attention", False) and not m.self_attention]
        cross_ok = len(cross) > 1 and len({tuple(m.k_proj.weight.shape) for m in cross}) == 1 and \
            all(m.kdim == m.vdim and m.k_proj.weight.shape == m.v_proj.weight.shape for m in cross) and \
            all(q.requires_grad for m in cross for q in (m.k_proj.weight, m.v_proj.weight, m.k_proj.bias, m.v_proj.bias))
        cross_group = []
        if cross_ok:
            cross_group = [w for m in cross for w in (m.k_proj.weight, m.v_proj.weight)] + \
                [b for m in cross for b in (m.k_proj.bias, m.v_proj.bias)]
        cross_ids = {id(q) for q in cross_group}
        seen, params = set(), []
        for p in model.parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            m = mha_of.get(id(p))
            group = [p]
            if id(p) in cross_ids:
                group = cross_group
            elif m is not None:
                if m.self_attention:
                    group = [m.k_proj.weight, m.v_proj.weight, m.q_proj.weight, m.k_proj.bias, m.v_proj.bias, m.q_proj.bias]
                elif cross_ok and m in cross:
                    group = [m.q_proj.weight, m.q_proj.bias]
                else:
                    group = [m.k_proj.weight, m.v_proj.weight, m.k_proj.bias, m.v_proj.bias, m.q_proj.weight, m.q_proj.bias]
                if not all(g.requires_grad for g in group):
                    group = [p]
            for g in group:
                if id(g) not in seen:
                    seen.add(id(g))
                    params.append(g)
        assert params, "no trainable parameters"
        self.params = params
        # 4-D weights NOT consumed by ops.Conv2dFn keep torch's contiguous layout (the patch-embedding projection: ops.PatchEmbedFn
        # reshapes it to [D, C*p*p] as a view and its im2col emits columns in (c, ph, pw) order -- ADVICE r3)
        self._plain4d = {id(w) for m in model.modules() for w in getattr(m, "_ofa_plain_conv_weights", lambda: ())()}
        dtype, device = params[0].dtype, params[0].device
        self.offsets, off = [], 0
        for p in params:
            assert p.dtype == dtype and p.device == device
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8          # keep every view 16-byte aligned
        self.numel = off
        self.flat = torch.zeros(off, dtype=dtype, device=device)
        self.grad = torch.zeros(off, dtype=dtype, device=device)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                self._view(self.flat, o, p).copy_(p.data)
                p.data = self._view(self.flat, o, p)
                p.grad = self._view(self.grad, o, p)
                p._ofa_grad = p.grad          # backward kernels accumulate straight into the arena (ops._sink)
        off_of = {id(p): o for p, o in zip(params, self.offsets)}
        for m in mhas:
            n = 3 if m.self_attention else 2
            kw, kb = m.k_proj.weight, m.k_proj.bias
            if id(kw) not in off_of or kw.shape[0] != kw.shape[1] and not m.self_attention and m.kdim != m.embed_dim:
                continue
            D, Kin = kw.shape
            ow, ob = off_of[id(kw)], off_of[id(kb)]
            ws = [m.k_proj.weight, m.v_proj.weight, m.q_proj.weight][:n]
            bs = [m.k_proj.bias, m.v_proj.bias, m.q_proj.bias][:n]
            ok = all(off_of.get(id(w)) == ow + i * D * Kin for i, w in enumerate(ws)) and \
                all(off_of.get(id(b)) == ob + i * D for i, b in enumerate(bs)) and (D * Kin) % 8 == 0 and D % 8 == 0
            if ok:
                m._pack = {"w": self.flat[ow:ow + n * D * Kin].view(n * D, Kin), "b": self.flat[ob:ob + n * D],
                           "gw": self.grad[ow:ow + n * D * Kin].view(n * D, Kin), "gb": self.grad[ob:ob + n * D]}
        if cross_ok:
            L = len(cross)
            D, Kin = cross[0].k_proj.weight.shape
            ow, ob = off_of[id(cross[0].k_proj.weight)], off_of[id(cross[0].k_proj.bias)]
            ok = all(off_of.get(id(w)) == ow + i * D * Kin for i, w in enumerate(cross_group[:2 * L])) and \
                all(off_of.get(id(b)) == ob + i * D for i, b in enumerate(cross_group[2 * L:])) and (D * Kin) % 8 == 0 and D % 8 == 0
            if ok:
                pack = {"w": self.flat[ow:ow + 2 * L * D * Kin].view(2 * L * D, Kin), "b": self.flat[ob:ob + 2 * L * D],
                        "gw": self.grad[ow:ow + 2 * L * D * Kin].view(2 * L * D, Kin), "gb": self.grad[ob:ob + 2 * L * D],
                        "params": tuple(cross_group), "layers": L, "D": D}
                for i, m in enumerate(cross):
                    m._cross_all = (pack, i)

    def _view(self, buf, o, p):
        """Parameter p's window of an arena.  Spatial convolution weights ([Cout, Cin, kh, kw], kh * kw > 1) live in the arena in
        the order the im2col GEMM reads them -- [Cout][kh][kw][Cin], i.e. torch's channels_last strides for the same logical
        shape -- so the forward needs no permuted copy of the weight and the weight-gradient GEMM writes the arena directly
        (state dicts hold values, not strides: checkpoints interchange unchanged).  Only weights that ops.Conv2dFn consumes:
        a module lists the others in `_ofa_plain_conv_weights()`."""
        n = p.numel()
        if p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and (p.shape[1] * p.shape[2] * p.shape[3]) % 8 == 0 \
                and id(p) not in self._plain4d:
            Cout, Cin, kh, kw = p.shape
            return buf[o:o + n].view(Cout, kh, kw, Cin).permute(0, 3, 1, 2)
        return buf[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-point it at the arena
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                p.grad = self._view(self.grad, o, p)
                p._ofa_grad = p.grad


class _AtenAudit(torch.utils._python_dispatch.TorchDispatchMode):
    """Shows the capture audit the tensor arguments of torch-native ops too (masks, cached index tensors consumed by aten indexing or
    `cat`): the C ABI sees only its own arguments.  Thread-local, i.e. the forward pass and whatever backward work runs on the calling
    thread; the autograd engine's device thread runs this package's Functions, whose launches the audit sees through lib.ptr()."""

    def __init__(self, audit):
        super().__init__()
        self.audit = audit

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        for a in args:
            if torch.is_tensor(a):
                if a.is_cuda:
                    self.audit.see(a)
            elif isinstance(a, (list, tuple)):
                for b in a:
                    if torch.is_tensor(b) and b.is_cuda:
                        self.audit.see(b)
        self.audit.label("aten::" + getattr(func, "__name__", str(func)))
        return func(*args, **(kwargs or {}))


# Capture mode: other threads of the process (the RCCL watchdog of torch.distributed polls events) must not invalidate a
# capture in progress -- only this thread's stream is being recorded.
_CAPTURE_MODE = "thread_local"


_KEEP_ALIVE = []        # CUDAGraph objects are never destroyed: the destructor of one whose capture failed aborts the process


class _Uncapturable(Exception):
    """A sample holds something the static-input walker does not understand: the step must stay eager."""


def _walk(obj, path, out):
    """Collect every tensor of a sample (slot values may be tensors, dicts or lists of tensors -- audio slots carry
    {fbank, fbank_lengths, mask_indices}) as (path, tensor), depth first, in a deterministic order."""
    if torch.is_tensor(obj):
        out.append((path, obj))
    elif isinstance(obj, dict):
        for k in obj:                                  # insertion order: the same code builds every batch of a structure
            _walk(obj[k], path + (k,), out)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _walk(v, path + (i,), out)
    elif hasattr(obj, "modality") and hasattr(obj, "is_src"):           # a Slot
        _walk(obj.value, path + ("value",), out)
    elif hasattr(obj, "tensors") and hasattr(obj, "structure"):         # a packing.PackPlan: index / segment tables
        for i, t in enumerate(obj.tensors()):
            out.append((path + (i,), t))
    elif obj is None or isinstance(obj, (str, int, float, bool)):
        pass
    else:
        raise _Uncapturable(f"{'/'.join(map(str, path))}: {type(obj).__name__}")


def sample_tensors(samples):
    """[(path, tensor)] over every micro-batch: all slot values, `target`, `constraint_masks` and any other tensor-valued
    entry of the sample dict (what a replay must copy into the graph's static inputs)."""
    out = []
    for i, s in enumerate(samples):
        for k in s:
            _walk(s[k], (i, k), out)
    return out


def sample_structure(samples):
    """Hashable structure of a step: per micro-batch the task key, per slot (modality, side, attributes), and per tensor
    (path, shape, dtype).  Two steps with the same structure run the same kernels on the same shapes -- the key of the graph
    cache and of the reducer's learned contribution counts."""
    sig = []
    for i, s in enumerate(samples):
        slots = tuple((getattr(sl.modality, "name", str(sl.modality)), bool(sl.is_src), tuple(sl.attributes or ()))
                      for sl in s["slots"])
        scal = tuple((k, s[k]) for k in sorted(s) if isinstance(s[k], (str, int, float, bool)))
        plan = s["pack"].structure() if s.get("pack") is not None else None          # launch bounds of a ragged batch
        sig.append((slots, scal, plan))
    tens = tuple((p, tuple(t.shape), str(t.dtype)) for p, t in sample_tensors(samples))
    return (tuple(sig), tens)


class TrainStep:
    """One optimisation step = `train_step(samples)`.

    use_graph: capture the step into a hipGraph (torch.cuda.graphs) and replay it.  The eager step is HOST-bound on an
    MI355X (~600 kernel launches + autograd bookkeeping take as long as the kernels themselves); a replay costs one
    launch.  Everything a replay must see fresh lives on the device: the Philox stream position (ops._Rng.base), the
    Adam step counter / bias corrections / lr (self._sched), the clip coefficient.  Each distinct batch structure
    (`sample_structure`: slot modalities / attributes, every tensor's shape and dtype) gets its own graph after
    `graph_warmup` eager steps; new batches of a captured structure are copied into the graph's static input tensors (every
    tensor of the sample: slot values incl. dict-valued audio slots, target, constraint masks).

    Data parallel (world > 1), `dp_graph`:
      "full"  (default) ONE graph holds forward, backward, the bucketed gradient all-reduces -- launched from inside backward
              as each bucket completes, so RCCL traffic over xGMI overlaps the remaining backward kernels -- the scalar
              all-reduce, clip and Adam.  RCCL collectives are captured like any other stream work.
      "split" two graphs (forward+backward, clip+Adam) around an eager bucket-by-bucket all-reduce: no overlap.  Chosen
              automatically when the process group's backend is not nccl (gloo's host-side collectives cannot be captured), or
              with dp_graph="split" / OFA_DP_GRAPH=split.
      Do not keep an autograd graph of this model alive from BEFORE the capture (outputs of an earlier forward on another
      stream): it pins the parameters' AccumulateGrad nodes to that stream and torch routes the captured backward through
      it, which invalidates the capture (ROCm: a crash in hipStreamEndCapture).  `del` such outputs first.
      A capture that fails cannot be retried in the same process (torch leaves the CUDA generator in its capturing state:
      "Cannot register the state during capturing stage" on every later capture), so there is no full -> split retry: the step
      engine warns, keeps the half-built graph objects alive (their destructor would abort the process) and runs eagerly from
      then on -- identically on every rank, the failure being structural."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=1.0, process_group=None,
                 bucket_bytes: int = 64 << 20, use_graph: bool = False, graph_warmup: int = 2,
                 label_smoothing: float = 0.0, constraint_range=None, drop_worst_ratio: float = 0.0, dp_graph: str = None,
                 loss_scale=None, max_graphs: int = 16, shard_optimizer: bool = False):
        """shard_optimizer (data parallel, opt-in): gradients are reduce-SCATTERED, every rank clips and updates only the 1 / world of
        the arena it owns, and the updated 16-bit parameters are all-gathered (distributed.GradBucketReducer, "sharded exchange") --
        the optimizer pass (0.6 ms of an 11 ms cfg-2 step) shrinks by the world size at the all-reduce's wire volume.  The optimizer
        state arrays keep their full length (what a rank does not own is never touched); the global gradient norm is the sum of the
        ranks' partial sums of squares.  Default off: the all-reduce form is what has run on hardware."""
        self.model = model
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.master = self.fp.flat.float().clone() if self.fp.flat.dtype != torch.float32 else self.fp.flat
        self.exp_avg = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.betas, self.eps, self.weight_decay, self.clip_norm = betas, eps, weight_decay, clip_norm
        self.num_updates = 0
        self.group = process_group
        self.shard_optimizer = bool(shard_optimizer)
        self.reducer = GradBucketReducer(self.fp.params, self.fp.grad, self.fp.offsets, process_group, bucket_bytes,
                                         shard=self.shard_optimizer)
        self.world = self.reducer.world
        self.pad = model.global_dict.pad()
        # criterion: plain cross entropy (engine/criterion/cross_entropy.py) or, with any of these set, the label-smoothed
        # one (label_smoothed_cross_entropy.py); a sample may carry its own "constraint_masks" [B, Tt, V]
        self.label_smoothing, self.constraint_range, self.drop_worst_ratio = label_smoothing, constraint_range, drop_worst_ratio
        self._stats = torch.zeros(3, dtype=torch.float64, device=dev)      # [sample_size, loss_sum, ntokens]
        self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._gnorm_t = torch.zeros(1, dtype=torch.float32, device=dev)
        # device-resident schedule: _step_t = number of updates done; _lr_t = learning rate; _sched = [grad multiplier,
        # lr*sqrt(1-b2^t)/(1-b1^t), lr, skip flag of this update, count of skipped updates] consumed by ofa_adam_step(step = 0)
        self._step_t = torch.zeros(1, dtype=torch.float64, device=dev)
        self._lr_t = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self._sched = torch.zeros(5, dtype=torch.float32, device=dev)
        self._lr = float(lr)
        self.use_graph = bool(use_graph) and dev.type == "cuda"
        # LayerDrop: the kept layers -- hence the autograd graph and every parameter's contribution count -- change per step and
        # per rank, so the step structure key says nothing about them: never captured, and the bucket reducer is never armed
        self.dynamic_autograd = any(float(getattr(m, "layerdrop", 0.0) or 0.0) > 0.0 for m in model.modules())
        if self.use_graph and self.dynamic_autograd:
            warnings.warn("ofasys_amd.TrainStep: LayerDrop draws the kept layers on the host every step; the step is not captured")
            self.use_graph = False
        # SyncBatchNorm layers exchange statistics in the middle of forward AND backward.  They do so on their own communicator
        # (ops.sync_bn_process_group, created here while every rank is at the same point), and with them in the model the bucket
        # reducer is never armed: ranks learn step structures at different times (padded lengths are part of the key), and a rank
        # that launches buckets from inside backward next to one that launches them at finish() would interleave the two kinds of
        # collectives differently -- harmless across communicators only as long as nobody's stream waits on both.  Every bucket
        # goes out at finish(), in index order, on every rank.
        self.sync_collectives = any(getattr(m, "_ofa_sync", None) not in (None, False) for m in model.modules())
        if self.sync_collectives and self.world > 1 and any(getattr(m, "_ofa_sync", None) is True for m in model.modules()):
            ops.sync_bn_process_group()
        self.graph_warmup = graph_warmup
        default_dp = "full"
        if self.world > 1:
            import torch.distributed as dist
            if dist.get_backend(process_group) != "nccl":
                default_dp = "split"
        self.dp_graph = dp_graph or os.environ.get("OFA_DP_GRAPH") or default_dp
        assert self.dp_graph in ("full", "split"), self.dp_graph
        # loss_scale: None, or a dict for the reference's DYNAMIC loss scaler (engine/optim/dynamic_loss_scaler.py; the fp16
        # trainer default, default_trainer.yaml:7-9): {"init_scale": 128, "scale_factor": 2, "scale_window": 2000, "tolerance": 0,
        # "threshold": None, "min_loss_scale": 1e-4}.  The whole state machine runs on the device (ofa_step_schedule_scaled):
        # the loss gradient is seeded with the scale, an overflowing update is skipped and the scale halved, and it grows back
        # every scale_window clean updates.  (The compute dtypes of this build are bf16 / fp32, whose exponent range makes the
        # scale numerically idle; it is here so that a reference fp16 recipe runs unchanged.)
        self.loss_scale_cfg = None
        if loss_scale is not None:
            cfg = {"init_scale": 2.0 ** 7, "scale_factor": 2.0, "scale_window": 2000, "tolerance": 0.0, "threshold": None,
                   "min_loss_scale": 1e-4}
            cfg.update(loss_scale if isinstance(loss_scale, dict) else {})
            self.loss_scale_cfg = cfg
            self._ls = torch.tensor([cfg["init_scale"], 0, -1, -1, 0, 0, 0, 0], dtype=torch.float64, device=dev)
        self._graphs = {}                           # batch structure -> dict(graphs, static samples, seen count)
        # Every captured structure pins its activations: the captures share ONE memory pool (step graphs replay one at a time and
        # consume nothing of each other, so the pool's blocks are reused across them) and at most `max_graphs` structures are
        # captured -- the rest run eagerly.  Variable-length data should arrive bucketed (pad_to_multiple / packing buckets).
        self.max_graphs = int(max_graphs)
        self._pool = None
        self._warned_cap = False
        self._skipped_seen = 0.0
        self._seed = torch.ones((), dtype=torch.float32, device=dev)   # the backward seed of a micro-batch's loss (autograd's default: one fill per call)
        self.last = {}

    # ------------------------------------------------------------------ learning rate (host -> device scalar)
    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self._lr_t.fill_(self._lr)                  # outside any captured region: replays read the new value

    # ------------------------------------------------------------------ the three phases of a step
    def _fwd_bwd(self, samples, overlap_reduce, structure=None):
        model = self.model
        model.train()
        ops.drop_pending()
        self.fp.zero_grad()
        self._stats.zero_()
        self.reducer.overlap = overlap_reduce
        # gradients are only read after backward unless buckets are all-reduced from inside it: fold lazily, in batches
        ops.defer_reductions(self.world == 1 or not overlap_reduce)
        self.reducer.begin_step(structure, dynamic=self.dynamic_autograd or self.sync_collectives)
        for s in samples:
            plan = s.get("pack")
            target = s["target"]
            if plan is None:
                logits = model(s["slots"])[0]
            else:                                    # ragged batch (packing.py): logits and targets by packed decoder row
                logits = model(s["slots"], pack=plan)[0]
                target = s.get("target_packed")
                if target is None:
                    idx = plan.dec_index
                    target = s["target"].reshape(-1)[idx.clamp_min(0)].masked_fill(idx < 0, self.pad).view(1, -1)
            cm = s.get("constraint_masks")
            if plan is not None and cm is not None:
                raise NotImplementedError("constraint masks are laid out by padded position: run such samples without a pack plan")
            counted = False
            if self.label_smoothing > 0 or self.constraint_range is not None or self.drop_worst_ratio > 0 or cm is not None:
                loss, _, n = ops.label_smoothed_cross_entropy(logits, target, self.pad, self.label_smoothing,
                                                              self.constraint_range, cm, self.drop_worst_ratio)
            else:
                # (the backward seed is known here: the criterion kernel writes the logits' gradient in its one pass over them)
                if self.loss_scale_cfg is None and (self._seed is None or self._seed.device != logits.device):
                    self._seed = torch.ones((), dtype=torch.float32, device=logits.device)
                loss = ops.cross_entropy_sum(logits, target, self.pad, seed=self._seed if self.loss_scale_cfg is None else None)
                n = None
                if loss.is_cuda and loss.dtype == torch.float32 and target.is_contiguous() and target.dtype == torch.int64:
                    # [sample_size, loss_sum, ntokens] += (non-pad targets, loss, non-pad targets) in one launch
                    K.step_stats_add(self._stats, loss.detach(), target, self.pad)
                    counted = True
                else:
                    n = target.ne(self.pad).sum()
            if self.loss_scale_cfg is None:
                if self._seed is None or self._seed.shape != loss.shape or self._seed.dtype != loss.dtype or self._seed.device != loss.device:
                    self._seed = torch.ones_like(loss)
                loss.backward(self._seed)
            else:                                    # FP16Optimizer.backward: loss * loss_scale (fp16_optimizer.py:92-102)
                loss.backward(self._ls[0].to(loss.dtype))
            if not counted:
                self._stats[0] += n
                self._stats[1] += loss.detach().double()
                self._stats[2] += n
        ops.flush_folds()
        ops.rng_advance()

    def _reduce(self):
        self.reducer.finish()
        all_reduce_scalars(self._stats, self.group)

    def _update(self):
        # coef = (1/sample_size) * min(1, clip / (||g/sample_size|| + 1e-6)), all on the device; a non-finite norm or an empty
        # batch sets the skip flag instead (ofa_step_schedule) and ofa_adam_step leaves weights and moments untouched
        self._gsq.zero_()
        owned = self.reducer.owned_ranges()              # the whole arena, or (sharded optimizer) this rank's pieces of the buckets
        for lo, hi, counted in owned:
            if counted:
                K.sumsq(self.fp.grad[lo:hi], self._gsq)
        if self.shard_optimizer and self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self._gsq, op=dist.ReduceOp.SUM, group=self.group)      # the global norm: every element counted once
        if self.loss_scale_cfg is None:
            K.step_schedule(self._gsq, self._stats, self._step_t, self._lr_t, self._sched, self._gnorm_t, self.clip_norm,
                            self.betas[0], self.betas[1])   # adam.py:205-207 + the clip coefficient, one device thread
        else:
            c = self.loss_scale_cfg
            K.step_schedule_scaled(self._gsq, self._stats, self._step_t, self._lr_t, self._sched, self._gnorm_t, self._ls,
                                   self.clip_norm, self.betas[0], self.betas[1], c["scale_factor"], c["scale_window"],
                                   c["tolerance"], c["threshold"], c["min_loss_scale"])
        for lo, hi, _ in owned:
            K.adam_step(self.master[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.fp.grad[lo:hi], self.fp.flat[lo:hi],
                        self._sched, 0.0, self.betas[0], self.betas[1], self.eps, self.weight_decay, 0)
        self.reducer.gather_params(self.fp.flat)         # (sharded optimizer: the other ranks' updated parameters)
        self._gnorm = self._gnorm_t

    def check(self):
        """Host-side poll of the device guard (ONE sync): raises FloatingPointError as engine/trainer.py:866-876 does when an
        update since the last call saw a non-finite gradient norm or no target token (that update was skipped on the device)."""
        if self.loss_scale_cfg is not None:          # overflows are the scaler's business (trainer.py:957-960 logs and goes on) ...
            if float(self._ls[5]) != 0.0:            # ... until the scale hits the floor (dynamic_loss_scaler.py:58-66)
                raise FloatingPointError(f"Minimum loss scale reached ({self.loss_scale_cfg['min_loss_scale']}). Your loss is "
                                         "probably exploding. Try lowering the learning rate, using gradient clipping or "
                                         "increasing the batch size.")
            return
        skipped = float(self._sched[4])
        if skipped > self._skipped_seen:
            n = int(skipped - self._skipped_seen)
            self._skipped_seen = skipped
            raise FloatingPointError(f"gradients are Nan/Inf (or sample_size == 0) in {n} update(s): skipped on the device")

    # ------------------------------------------------------------------ graph plumbing
    def captured_graphs(self):
        return sum(1 for e in self._graphs.values() if "graphs" in e)

    def _capture(self, samples, structure, mode):
        import gc
        gc.collect()                      # drop unreachable autograd graphs (see the class docstring) before recording
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        pool = self._pool
        entry = {"static": samples, "graphs": [], "mode": mode}
        # A captured kernel node holds raw addresses.  What the capture allocates lives in the graphs' private pool; everything older
        # lives in the default pool and stays mapped only while Python holds it.  lib.capture_audit records every such tensor a
        # C-ABI call of the capture addresses (and _AtenAudit the inputs of the forward's torch-native ops); the entry PINS them, so
        # that no cache that later drops or regrows a tensor (ops.cached_index, SegmentPlan, a grown mask) can unmap memory a graph
        # still reads -- the allocator flush at the entry of the NEXT capture would otherwise hand it back to the driver.
        with capture_audit() as audit, _AtenAudit(audit):
            if self.world == 1 or mode == "full":
                g = torch.cuda.CUDAGraph()
                _KEEP_ALIVE.append(g)
                with torch.cuda.graph(g, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    self._fwd_bwd(samples, overlap_reduce=self.world > 1, structure=structure)
                    if self.world > 1:
                        self._reduce()
                    self._update()
                entry["graphs"] = [g]
            else:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                _KEEP_ALIVE.extend((ga, gb))
                with torch.cuda.graph(ga, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    self._fwd_bwd(samples, overlap_reduce=False)
                with torch.cuda.graph(gb, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    self._update()
                entry["graphs"] = [ga, gb]
        entry["audit"] = audit
        if os.environ.get("OFA_CAPTURE_PINS", "1") != "0":          # (0: the audit tool's negative control -- reproduces a dangling pointer)
            entry["pins"] = audit.tensors()
        return entry

    def owned_tensors(self):
        """Tensors whose lifetime is the step engine's / the model's own (what a capture may address without a pin)."""
        own = list(self.model.parameters()) + list(self.model.buffers())
        own += [self.fp.flat, self.fp.grad, self.master, self.exp_avg, self.exp_avg_sq, self._stats, self._gsq, self._gnorm_t,
                self._step_t, self._lr_t, self._sched, self._seed, getattr(self, "_ls", None)]
        own += list(ops._Rng.base.values())
        return own

    def audit_report(self, entry):
        """[(first C-ABI call, shape, dtype, storage bytes)]: default-pool tensors the captured graphs of `entry` address that neither
        the engine, the model nor the entry's static inputs own -- each one is kept alive by entry["pins"] only."""
        own = self.owned_tensors() + [t for _, t in sample_tensors(entry["static"])]
        return entry["audit"].foreign(own)

    def _replay(self, entry, samples):
        if samples is not entry["static"]:
            # ONE batched copy launch (K.copy_batched) instead of one ~3.5 us copy launch per tensor (34 per cfg-2 step, 77 per cfg-2b step)
            K.copy_batched([(dst, src) for (_, dst), (_, src) in zip(entry["static_tensors"], sample_tensors(samples)) if dst is not src])
        entry["graphs"][0].replay()
        if len(entry["graphs"]) == 2:
            self.reducer.overlap = False
            self.reducer.begin_step(None)
            self._reduce()
            entry["graphs"][1].replay()

    def train_step(self, samples: List[dict], eager: bool = False):
        """samples: one dict per (task, micro-batch): {"slots": [...], "target": LongTensor[B,Tt], optional
        "constraint_masks", "task"}."""
        done = False
        structure = None
        if (self.use_graph and not eager) or self.world > 1:
            try:
                structure = sample_structure(samples)
            except _Uncapturable as e:
                if self.use_graph:
                    warnings.warn(f"ofasys_amd.TrainStep: a sample holds an object the static-input walker does not know ({e}); "
                                  "running eagerly")
                    self.use_graph = False
        if self.use_graph and not eager and structure is not None:
            entry = self._graphs.get(structure)
            if entry is None:
                if len(self._graphs) >= 64 * max(self.max_graphs, 1):       # bound the bookkeeping of never-captured structures too
                    self._graphs = {k: e for k, e in self._graphs.items() if "graphs" in e}
                entry = self._graphs[structure] = {"seen": 0}
            if "graphs" not in entry and entry["seen"] >= self.graph_warmup and self.captured_graphs() >= self.max_graphs:
                if not self._warned_cap:
                    warnings.warn(f"ofasys_amd.TrainStep: {self.max_graphs} batch structures are captured already; further ones run "
                                  "eagerly (bucket the padded lengths -- pad_to_multiple / packing buckets -- or raise max_graphs)")
                    self._warned_cap = True
            elif "graphs" in entry:
                self._replay(entry, samples)
                done = True
            elif entry["seen"] >= self.graph_warmup and (self.world == 1 or self.dp_graph != "full"
                                                         or self.reducer.knows(structure) or self.graph_warmup == 0):
                mode = "full" if (self.world > 1 and self.dp_graph == "full") else "split"
                rng_state = ops._Rng.offset
                try:
                    torch.cuda.synchronize()
                    cap = self._capture(samples, structure, mode)
                except Exception as e:              # anything uncapturable (a custom adaptor, a collective): eager from now on
                    warnings.warn(f"ofasys_amd.TrainStep: hipGraph capture ({mode}) failed ({type(e).__name__}: {e}); a failed "
                                  "capture cannot be retried in this process -- running eagerly")
                    ops._Rng.offset = rng_state
                    ops.defer_reductions(False)
                    self.use_graph = False
                else:
                    cap["static_tensors"] = sample_tensors(cap["static"])
                    entry.update(cap)
                    self._replay(entry, samples)
                    done = True
            else:
                entry["seen"] += 1
        if not done:
            self._fwd_bwd(samples, overlap_reduce=True, structure=structure)
            self._reduce()
            self._update()
        self.num_updates += 1
        self.last = {"stats": self._stats, "gnorm": self._gnorm, "skipped": self._sched[3:5]}
        if self.loss_scale_cfg is not None:
            self.last["loss_scale"] = self._ls[0:1]
        return self.last


Trainer = TrainStep            # the name round-1 code and tests use; `ofasys_amd.Trainer` is the fit() facade (engine.py)
