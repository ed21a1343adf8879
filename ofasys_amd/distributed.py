"""Data-parallel gradient reduction for the hot path (reference contract: distributed/distributed_model_dispatcher.py:49-75
-> torch DDP buckets of bucket_cap_mb=25 (configure/configs.py:250), all-reduce(SUM) of every trainable parameter's
grad overlapped with backward; unused parameters contribute zeros, default_trainer.yaml:13-14).

MI355X design: gradients already live in ONE flat arena (ofasys_amd/trainer.py), so a "bucket" is just a contiguous
slice of it -- no gather/scatter copies.  Buckets are cut in reverse parameter order (the order backward produces
them).  Every gradient contribution -- whether a HIP kernel accumulated it straight into the arena (ops._sink) or
autograd's AccumulateGrad did -- calls `notify(i)`; once a parameter has received as many contributions as it did in
the learning step of the SAME step structure its bucket counts down, and a full bucket fires `all_reduce(async_op=True)` on
its slice, so RCCL traffic over xGMI overlaps the remaining backward kernels.  Buckets are launched STRICTLY IN INDEX ORDER (bucket b
only once 0..b-1 are in flight): under DP every rank pads its own batch, so ranks can be in different modes in the same step
(learning: everything at finish(); armed: from inside backward; replaying a graph) and completion order is rank-local -- index
order is the one sequence of collectives every rank issues in every mode.  Counts are keyed on the step's full structure
(trainer.sample_structure: slot modalities / attributes, every tensor shape) -- two steps with the same key run the same
autograd graph, so the counts are exact; a contribution that still arrives for a bucket already in flight raises instead of
racing the collective.  Whatever is left (unused parameters, the learning step itself) is reduced at `finish()`.
Bucket size defaults to 64 MiB: xGMI rings are per-link bound (~153 GB/s/link), large messages amortise the
per-collective latency, and 288 GB of HBM makes the arena free.
"""
from typing import List

import torch
import torch.distributed as dist


class GradBucketReducer:
    def __init__(self, params: List[torch.nn.Parameter], flat_grad: torch.Tensor, offsets: List[int],
                 process_group=None, bucket_bytes: int = 64 << 20, shard: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_available() and dist.is_initialized() else 0
        # shard: the SHARDED-OPTIMIZER exchange (reduce-scatter -> every rank updates 1 / world of the arena -> all-gather of the
        # parameters) instead of the all-reduce: see the section "sharded exchange" below
        self.shard = bool(shard)
        self.flat_grad = flat_grad
        self.params = params
        elem = flat_grad.element_size()
        # cut buckets walking the arena from the END (backward order ~ reverse registration order)
        self.buckets = []   # (start, end, [param indices])
        cur_hi, cur_members = flat_grad.numel(), []
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
        self.overlap = True                       # False: never launch from inside backward (captured steps)
        self.profile = False                      # bench.py: time the post-backward wait of eager steps (exposed_events)
        self.exposed_events = []
        self.last_launch_order, self.last_early = [], 0
        self.expected = None                      # contributions per parameter per step for the current signature
        self._learned = {}                        # step signature -> learned contribution counts
        self._sig = None
        self._count = [0] * len(params)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self.launch_order = []
        self._handles = []
        self._hooks = []
        for i, p in enumerate(params):
            if p.requires_grad:
                p._ofa_grad_ready = self._make_notify(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self.notify(i)))
        self._reset()

    def _make_notify(self, i):
        return lambda: self.notify(i)

    def knows(self, signature):
        return signature is not None and signature in self._learned

    def begin_step(self, signature=None, dynamic=False):
        """`signature` identifies the step's structure (trainer.sample_structure); early bucket launches are only armed for a
        structure whose contribution counts were learned on an earlier, identical step (None: never armed, never learned).
        `dynamic`: the autograd graph of this step is NOT a function of the structure key (LayerDrop draws the kept layers per step
        and per rank, module/layer_drop.py:37-41): nothing is armed and nothing is learned -- every bucket goes out at finish(),
        in index order, dropped layers contributing zeros (the reference's DDP does the same through find_unused_parameters,
        default_trainer.yaml:13-14)."""
        if dynamic:
            signature = None
        self._sig = signature
        self.expected = self._learned.get(signature) if (self.overlap and signature is not None) else None
        self._reset()

    def notify(self, i):
        """One gradient contribution for parameter i has been enqueued on the compute stream."""
        self._count[i] += 1
        if self.world == 1 or self.expected is None:
            return
        if self._count[i] > self.expected[i] or self._launched[self.param_bucket[i]]:
            raise RuntimeError(
                f"GradBucketReducer: parameter {i} received contribution #{self._count[i]} but {self.expected[i]} were learned "
                "for this step structure (its bucket's all-reduce may already be in flight): the structure key does not "
                "determine the autograd graph -- pass a distinguishing 'task' entry in the samples")
        if self._count[i] == self.expected[i]:
            b = self.param_bucket[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self._launch_ready()

    def _launch_ready(self):
        """Launch the longest prefix of complete buckets: the order of collectives is the bucket index on every rank."""
        while self._next < len(self.buckets) and self._ready[self._next]:
            self._launch(self._next)

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        assert b == self._next, (b, self._next)
        self._launched[b] = True
        self._next = b + 1
        self.launch_order.append(b)
        if not self.shard:
            self._handles.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        plo, phi, tail = self.piece(b)
        if phi > plo:       # in place: the output is this rank's piece of the input (ncclReduceScatter's in-place form)
            self._handles.append(dist.reduce_scatter_tensor(self.flat_grad[plo:phi], self.flat_grad[lo:tail], op=dist.ReduceOp.SUM,
                                                            group=self.group, async_op=True))
        if hi > tail:
            self._handles.append(dist.all_reduce(self.flat_grad[tail:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ------------------------------------------------------------------ sharded exchange
    # A bucket [lo, hi) of the arena is cut into `world` equal pieces of a multiple of SHARD_ALIGN elements -- rank r owns piece r -- and
    # a short tail (< world * SHARD_ALIGN elements) that every rank keeps (all-reduced, updated identically everywhere: no gather).
    # Buckets stay slices of the ONE arena (no packing copies); both collectives run in place on it.  Per step a rank then reads and
    # writes 1 / world of the optimizer state (Adam is 28 bytes per parameter: 0.6 ms of the cfg-2 step at world = 1) and receives the
    # other ranks' updated 16-bit parameters -- 2 bytes per parameter per rank pair instead of the all-reduce's second half, which moved
    # the same 2 bytes: the wire volume is that of the all-reduce (reduce-scatter + all-gather IS a ring all-reduce), only the
    # all-gather now sits behind the optimizer instead of in front of it.
    SHARD_ALIGN = 8          # elements: 16-byte vectors of the 16-bit arena, 32-byte ones of the fp32 state

    def piece(self, b, rank=None):
        """(piece_lo, piece_hi, tail_lo) of bucket b for `rank` (default: this rank): its piece and where the replicated tail starts."""
        lo, hi, _ = self.buckets[b]
        q = self.world * self.SHARD_ALIGN
        per = (hi - lo) // q * self.SHARD_ALIGN
        r = self.rank if rank is None else rank
        return lo + r * per, lo + (r + 1) * per, lo + self.world * per

    def owned_ranges(self):
        """[(lo, hi, counted)]: the arena ranges this rank's optimizer updates -- its piece of every bucket (counted once in a global
        norm) and every bucket's replicated tail (counted on rank 0 only).  Not sharded: the whole arena."""
        if not self.shard:
            return [(0, self.flat_grad.numel(), True)]
        out = []
        for b, (lo, hi, _) in enumerate(self.buckets):
            plo, phi, tail = self.piece(b)
            if phi > plo:
                out.append((plo, phi, True))
            if hi > tail:
                out.append((tail, hi, self.rank == 0))
        return out

    def gather_params(self, flat_param):
        """After the owners' update: every bucket's pieces of the 16-bit parameter arena to every rank (in place: the input is this rank's
        piece of the output), bucket by bucket in index order; waits for all of them."""
        if not self.shard or self.world == 1:
            return
        handles = []
        for b, (lo, hi, _) in enumerate(self.buckets):
            plo, phi, tail = self.piece(b)
            if phi > plo:
                handles.append(dist.all_gather_into_tensor(flat_param[lo:tail], flat_param[plo:phi], group=self.group, async_op=True))
        for h in handles:
            h.wait()

    def _reset(self):
        self._count = [0] * len(self.params)
        for b, (_, _, members) in enumerate(self.buckets):
            if self.expected is None:
                self._pending[b] = -1
            else:
                self._pending[b] = sum(1 for i in members if self.expected[i] > 0)
                if self._pending[b] == 0:
                    self._pending[b] = -1                   # nothing will ever notify: left for finish()
            self._launched[b] = False
        self._ready = [False] * len(self.buckets)
        self._next = 0
        self.launch_order = []                              # (tests: the sequence of collectives of the last step)
        self._handles = []

    def finish(self):
        """Reduce whatever backward did not trigger (unused parameters, learning step), then wait for every bucket."""
        if self.world > 1:
            early = self._next                                  # buckets that went out from inside backward
            for b in range(self._next, len(self.buckets)):
                self._launch(b)
            ev = None
            if self.profile and self.flat_grad.is_cuda:         # how long the compute stream stalls for the exchange after backward
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for h in self._handles:
                h.wait()
            if ev is not None:
                ev[1].record()
                self.exposed_events.append(ev)
            self.last_early = early
        self.last_launch_order = list(self.launch_order)
        if self.expected is None and self.overlap and self._sig is not None:
            self._learned[self._sig] = list(self._count)
        self._reset()


    def bucket_sizes(self):
        return [(hi - lo) * self.flat_grad.element_size() for lo, hi, _ in self.buckets]

    def time_buckets_alone(self, reps=3):
        """ms per bucket of a blocking all-reduce of that slice with nothing else running (bench.py: the un-overlapped price of the
        exchange; run on every rank, outside any step -- the gradient arena is scratch between steps)."""
        out = []
        if self.world == 1 or not self.flat_grad.is_cuda:
            return out
        for lo, hi, _ in self.buckets:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)      # warm
            e0.record()
            for _ in range(reps):
                dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
        return out


def all_reduce_scalars(t: torch.Tensor, group=None):
    """Sum of the packed logging scalars + sample_size across ranks (engine/trainer.py:1267 -> distributed/utils.py:598-644)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
