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
from typing import List, Optional

import torch

from . import kernels as K
from . import ops
from .distributed import GradBucketReducer, all_reduce_scalars


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
        seen, params = set(), []
        for p in model.parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            m = mha_of.get(id(p))
            group = [p]
            if m is not None:
                if m.self_attention:
                    group = [m.k_proj.weight, m.v_proj.weight, m.q_proj.weight, m.k_proj.bias, m.v_proj.bias, m.q_proj.bias]
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
                self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
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

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-point it at the arena
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + o * self.grad.element_size():
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._ofa_grad = p.grad


# Capture mode: other threads of the process (the RCCL watchdog of torch.distributed polls events) must not invalidate a
# capture in progress -- only this thread's stream is being recorded.
_CAPTURE_MODE = "thread_local"


class Trainer:
    """One optimisation step = `train_step(samples)`.

    use_graph: capture the step into a hipGraph (torch.cuda.graphs) and replay it.  The eager step is HOST-bound on an
    MI355X (~600 kernel launches + autograd bookkeeping take as long as the kernels themselves); a replay costs one
    launch.  Everything a replay must see fresh lives on the device: the Philox stream position (ops._Rng.base), the
    Adam step counter / bias corrections / lr (self._sched), the clip coefficient.  Each distinct batch structure
    (slot and target shapes) gets its own graph after `graph_warmup` eager steps; new batches of a captured structure are
    copied into the graph's static input tensors.  With world_size > 1 the step is two graphs (forward+backward,
    clip+Adam) around an eager all-reduce of the gradient arena: collectives are kept out of the captured region."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=1.0, process_group=None,
                 bucket_bytes: int = 64 << 20, use_graph: bool = False, graph_warmup: int = 2,
                 label_smoothing: float = 0.0, constraint_range=None, drop_worst_ratio: float = 0.0):
        self.model = model
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.master = self.fp.flat.float().clone() if self.fp.flat.dtype != torch.float32 else self.fp.flat
        self.exp_avg = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.betas, self.eps, self.weight_decay, self.clip_norm = betas, eps, weight_decay, clip_norm
        self.num_updates = 0
        self.group = process_group
        self.reducer = GradBucketReducer(self.fp.params, self.fp.grad, self.fp.offsets, process_group, bucket_bytes)
        self.world = self.reducer.world
        self.pad = model.global_dict.pad()
        # criterion: plain cross entropy (engine/criterion/cross_entropy.py) or, with any of these set, the label-smoothed
        # one (label_smoothed_cross_entropy.py); a sample may carry its own "constraint_masks" [B, Tt, V]
        self.label_smoothing, self.constraint_range, self.drop_worst_ratio = label_smoothing, constraint_range, drop_worst_ratio
        self._stats = torch.zeros(3, dtype=torch.float64, device=dev)      # [sample_size, loss_sum, ntokens]
        self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._gnorm_t = torch.zeros(1, dtype=torch.float32, device=dev)
        # device-resident schedule: _step_t = number of updates done; _lr_t = learning rate; _sched = [grad multiplier,
        # lr*sqrt(1-b2^t)/(1-b1^t), lr] consumed by ofa_adam_step(step = 0)
        self._step_t = torch.zeros(1, dtype=torch.float64, device=dev)
        self._lr_t = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self._sched = torch.zeros(3, dtype=torch.float32, device=dev)
        self._lr = float(lr)
        self.use_graph = bool(use_graph) and dev.type == "cuda"
        self.graph_warmup = graph_warmup
        self._graphs = {}                           # batch structure -> dict(graphs, static samples, seen count)
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
    def _fwd_bwd(self, samples, overlap_reduce):
        model = self.model
        model.train()
        self.fp.zero_grad()
        self._stats.zero_()
        self.reducer.overlap = overlap_reduce
        # gradients are only read after backward unless buckets are all-reduced from inside it: fold lazily, in batches
        ops.defer_reductions(self.world == 1 or not overlap_reduce)
        self.reducer.begin_step(tuple(s.get("task", len(s["slots"])) for s in samples))
        for s in samples:
            logits = model(s["slots"])[0]
            cm = s.get("constraint_masks")
            if self.label_smoothing > 0 or self.constraint_range is not None or self.drop_worst_ratio > 0 or cm is not None:
                loss, _, n = ops.label_smoothed_cross_entropy(logits, s["target"], self.pad, self.label_smoothing,
                                                              self.constraint_range, cm, self.drop_worst_ratio)
            else:
                loss = ops.cross_entropy_sum(logits, s["target"], self.pad)
                n = s["target"].ne(self.pad).sum()
            loss.backward()
            self._stats[0] += n
            self._stats[1] += loss.detach().double()
            self._stats[2] += n
        ops.flush_folds()
        ops.side_join()                             # side-stream weight gradients are complete beyond this point
        ops.rng_advance()

    def _reduce(self):
        self.reducer.finish()
        all_reduce_scalars(self._stats, self.group)

    def _update(self):
        # coef = (1/sample_size) * min(1, clip / (||g/sample_size|| + 1e-6)), all on the device
        self._gsq.zero_()
        K.sumsq(self.fp.grad, self._gsq)
        K.step_schedule(self._gsq, self._stats, self._step_t, self._lr_t, self._sched, self._gnorm_t, self.clip_norm,
                        self.betas[0], self.betas[1])       # adam.py:205-207 + the clip coefficient, one device thread
        gnorm = self._gnorm_t
        K.adam_step(self.master, self.exp_avg, self.exp_avg_sq, self.fp.grad, self.fp.flat, self._sched, 0.0,
                    self.betas[0], self.betas[1], self.eps, self.weight_decay, 0)
        self._gnorm = gnorm

    # ------------------------------------------------------------------ graph plumbing
    @staticmethod
    def _tensors_of(samples):
        out = []
        for s in samples:
            for sl in s["slots"]:
                v = getattr(sl, "value", None)
                if torch.is_tensor(v):
                    out.append(v)
            out.append(s["target"])
        return out

    def _signature(self, samples):
        return tuple((s.get("task", len(s["slots"])),) + tuple((tuple(t.shape), t.dtype) for t in self._tensors_of([s]))
                     for s in samples)

    def _capture(self, samples):
        ops.side_stream()                           # streams / RNG bases must exist before capture starts
        pool = torch.cuda.graph_pool_handle()
        entry = {"static": samples, "graphs": []}
        if self.world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self._fwd_bwd(samples, overlap_reduce=False)
                self._update()
            entry["graphs"] = [g]
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self._fwd_bwd(samples, overlap_reduce=False)
            with torch.cuda.graph(gb, pool=pool, capture_error_mode=_CAPTURE_MODE):
                self._update()
            entry["graphs"] = [ga, gb]
        return entry

    def _replay(self, entry, samples):
        if samples is not entry["static"]:
            for dst, src in zip(self._tensors_of(entry["static"]), self._tensors_of(samples)):
                if dst is not src:
                    dst.copy_(src, non_blocking=True)
        entry["graphs"][0].replay()
        if self.world > 1:
            self.reducer.overlap = False
            self.reducer.begin_step(None)
            self._reduce()
            entry["graphs"][1].replay()

    def train_step(self, samples: List[dict], eager: bool = False):
        """samples: one dict per (task, micro-batch): {"slots": [...], "target": LongTensor[B,Tt]}."""
        done = False
        if self.use_graph and not eager:
            sig = self._signature(samples)
            entry = self._graphs.get(sig)
            if entry is None:
                entry = self._graphs[sig] = {"seen": 0}
            if "graphs" in entry:
                self._replay(entry, samples)
                done = True
            elif entry["seen"] >= self.graph_warmup:
                rng_state = (ops._Rng.offset,)
                try:
                    torch.cuda.synchronize()
                    cap = self._capture(samples)
                except Exception as e:              # anything uncapturable in a custom adaptor: stay eager, loudly
                    import warnings
                    warnings.warn(f"ofasys_amd.Trainer: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly")
                    self.use_graph = False
                    ops._Rng.offset = rng_state[0]
                    ops.side_join()
                else:
                    entry.update(cap)
                    self._replay(entry, samples)
                    done = True
            else:
                entry["seen"] += 1
        if not done:
            self._fwd_bwd(samples, overlap_reduce=True)
            self._reduce()
            self._update()
        self.num_updates += 1
        self.last = {"stats": self._stats, "gnorm": self._gnorm}
        return self.last
