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


class Trainer:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_norm=1.0, process_group=None,
                 bucket_bytes: int = 64 << 20):
        self.model = model
        self.fp = FlatParams(model)
        dev = self.fp.flat.device
        self.master = self.fp.flat.float().clone() if self.fp.flat.dtype != torch.float32 else self.fp.flat
        self.exp_avg = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.fp.numel, dtype=torch.float32, device=dev)
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        self.num_updates = 0
        self.group = process_group
        self.reducer = GradBucketReducer(self.fp.params, self.fp.grad, self.fp.offsets, process_group, bucket_bytes)
        self.pad = model.global_dict.pad()
        self._stats = torch.zeros(3, dtype=torch.float64, device=dev)      # [sample_size, loss_sum, ntokens]
        self._gsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last = {}

    def train_step(self, samples: List[dict]):
        """samples: one dict per (task, micro-batch): {"slots": [...], "target": LongTensor[B,Tt]}."""
        model = self.model
        model.train()
        self.fp.zero_grad()
        self._stats.zero_()
        self.reducer.begin_step(tuple(s.get("task", len(s["slots"])) for s in samples))
        for i, s in enumerate(samples):
            logits = model(s["slots"])[0]
            loss = ops.cross_entropy_sum(logits, s["target"], self.pad)
            loss.backward()
            n = s["target"].ne(self.pad).sum()
            self._stats[0] += n
            self._stats[1] += loss.detach().double()
            self._stats[2] += n
        self.reducer.finish()
        all_reduce_scalars(self._stats, self.group)
        # coef = (1/sample_size) * min(1, clip / (||g/sample_size|| + 1e-6)), all on the device
        self._gsq.zero_()
        K.sumsq(self.fp.grad, self._gsq)
        inv_n = (1.0 / self._stats[0]).float()
        gnorm = self._gsq.sqrt() * inv_n
        coef = inv_n.reshape(1)
        if self.clip_norm > 0:
            coef = coef * (self.clip_norm / (gnorm + 1e-6)).clamp(max=1.0)
        self.num_updates += 1
        K.adam_step(self.master, self.exp_avg, self.exp_avg_sq, self.fp.grad, self.fp.flat, coef.contiguous(), self.lr,
                    self.betas[0], self.betas[1], self.eps, self.weight_decay, self.num_updates)
        self.last = {"stats": self._stats, "gnorm": gnorm}
        return self.last
