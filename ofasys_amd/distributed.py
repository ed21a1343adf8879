"""Data-parallel gradient reduction for the hot path (reference contract: distributed/distributed_model_dispatcher.py:49-75
-> torch DDP buckets of bucket_cap_mb=25 (configure/configs.py:250), all-reduce(SUM) of every trainable parameter's
grad overlapped with backward; unused parameters contribute zeros, default_trainer.yaml:13-14).

MI355X design: gradients already live in ONE flat arena (ofasys_amd/trainer.py), so a "bucket" is just a contiguous
slice of it -- no gather/scatter copies.  Buckets are cut in reverse parameter order (the order backward produces
them); a post-accumulate hook counts parameters down and fires `all_reduce(async_op=True)` on the slice as soon as
its last gradient lands, so RCCL traffic over xGMI overlaps the remaining backward kernels.  Slices that backward
never touches (unused parameters) are reduced at `finish()`.  Bucket size defaults to 64 MiB: xGMI rings are
per-link bound (~153 GB/s/link), large messages amortise the per-collective latency, and 288 GB of HBM makes the
arena free.
"""
from typing import List

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: List[torch.nn.Parameter], flat_grad: torch.Tensor, offsets: List[int],
                 process_group=None, bucket_bytes: int = 64 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.flat_grad = flat_grad
        self.params = params
        elem = flat_grad.element_size()
        # cut buckets walking the arena from the END (backward order ~ reverse registration order)
        self.buckets = []   # (start, end, [param indices])
        end = flat_grad.numel()
        cur_hi, cur_members = end, []
        for i in reversed(range(len(params))):
            lo = offsets[i]
            cur_members.append(i)
            if (cur_hi - lo) * elem >= bucket_bytes or i == 0:
                self.buckets.append((lo, cur_hi, list(cur_members)))
                cur_hi, cur_members = lo, []
        self.param_bucket = {}
        for b, (_, _, members) in enumerate(self.buckets):
            for i in members:
                self.param_bucket[i] = b
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._enabled = True
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(params):
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def _make_hook(self, i):
        def hook(param):
            if not self._enabled:
                return
            b = self.param_bucket[i]
            self._pending[b] -= 1
            if self._pending[b] == 0 and not self._launched[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self._launched[b] = True
        self._handles.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reset(self):
        for b, (_, _, members) in enumerate(self.buckets):
            self._pending[b] = sum(1 for i in members if self.params[i].requires_grad)
            self._launched[b] = False
        self._handles = []

    def no_sync(self, flag=True):
        """Skip reduction for this backward (earlier micro-batches / tasks accumulate locally, engine/trainer.py:766-784)."""
        self._enabled = not flag

    def finish(self):
        """Reduce whatever backward did not trigger (unused parameters), then wait for every bucket."""
        if self.world > 1:
            for b in range(len(self.buckets)):
                if not self._launched[b]:
                    self._launch(b)
            for h in self._handles:
                h.wait()
        self.reset()


def all_reduce_scalars(t: torch.Tensor, group=None):
    """Sum of the packed logging scalars + sample_size across ranks (engine/trainer.py:1267 -> distributed/utils.py:598-644)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
