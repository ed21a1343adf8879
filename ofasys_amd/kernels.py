"""Thin functional wrappers over the C ABI (no autograd here -- see ofasys_amd/ops.py).

Each wrapper allocates outputs with torch (plumbing: device memory + streams), checks layout, and enqueues the HIP
kernel on torch's current stream.  Everything here requires GPU tensors; nothing falls back to torch math.
"""
import ctypes

import torch

from .lib import (BF16, F32, GEMM_ACCUM, GEMM_BIAS_COL, GEMM_BIAS_ROW, GEMM_FORCE_SIMPLE, GEMM_OUT_F32, OfaError,
                  dptr, dtype_code, lib, ptr, stream)

_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """A grow-only scratch buffer (fp32 words) per (tag, device, HIP stream): kernels on different streams never share one."""
    key = (tag, device, torch.cuda.current_stream().cuda_stream if device.type == "cuda" else 0)
    buf = _ws_cache.get(key)
    n = (nbytes + 3) // 4
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1 << 20), dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


class _FoldJob(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("cols", ctypes.c_int64), ("stride", ctypes.c_int64),
                ("nslots", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("alpha", ctypes.c_float),
                ("out_dtype", ctypes.c_int32)]


class _CopyJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_int64)]


def copy_batched(pairs):
    """dst.copy_(src) for every (dst, src) pair of same-shape, same-dtype, contiguous device tensors -- in ONE launch per 96 pairs
    (ofa_copy_batched).  Pairs that do not qualify are copied by torch."""
    jobs = []
    for dst, src in pairs:
        if (src.dtype == dst.dtype and src.shape == dst.shape and src.device == dst.device and dst.is_cuda
                and src.is_contiguous() and dst.is_contiguous()):
            jobs.append((dptr(src), dptr(dst), src.numel() * src.element_size()))
        else:
            dst.copy_(src, non_blocking=True)
    if jobs:
        arr = (_CopyJob * len(jobs))()
        for it, (s_, d_, n) in zip(arr, jobs):
            it.src, it.dst, it.bytes = s_, d_, n
        lib().call("ofa_copy_batched", ctypes.addressof(arr), len(jobs), stream())


DEFER_FOLD = 2            # OFA_DEFER_FOLD
GEMM_DEFER_REDUCE = 128   # OFA_GEMM_DEFER_REDUCE


class FoldQueue:
    """Deferred `out (+)= alpha * sum_s part[s]` reductions (LayerNorm dgamma/dbeta/dbias, bias column sums, split-K weight
    gradients): producers leave their fp32 partial rows in a private buffer and register a job; `flush()` folds up to 56
    jobs per launch (csrc/fold.hip) instead of one 5-9 us launch each."""

    # partial rows are folded while they still sit in the 256 MiB Infinity Cache: deferring a whole backward pass worth
    # of split-K slabs (3.6 GB on cfg-2) sends them to HBM and back and costs more than the saved launches
    MAX_PENDING_BYTES = 48 << 20        # (one layer's grouped weight-gradient slabs: 12.20 ms/step; 96 MiB 12.25, 200 MiB 12.35, 16 MiB 12.29)

    def __init__(self):
        self.jobs, self.keep, self.bytes, self.outs, self.want_flush = [], [], 0, set(), False

    def add(self, part, part_off, out, cols, stride, nslots, alpha=1.0, accumulate=True, flush_ok=True):
        assert part.dtype == torch.float32 and out.is_contiguous()
        if out.data_ptr() in self.outs:
            # a second contribution to the same gradient (a module applied to several slots): jobs of one launch run
            # concurrently, so two read-modify-writes of one output must not share a launch
            self.flush()
        self.outs.add(out.data_ptr())
        self.jobs.append(_FoldJob(dptr(part) + part_off * 4, dptr(out), cols, stride, nslots, int(accumulate),
                                  float(alpha), dtype_code(out)))
        self.keep.append((part, out))
        self.bytes += nslots * stride * 4
        self.want_flush = self.bytes > self.MAX_PENDING_BYTES

    def flush_if_large(self):
        """Called by a producer AFTER it has launched the kernel that writes the partial rows it just registered."""
        if getattr(self, "want_flush", False):
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        arr = (_FoldJob * len(self.jobs))(*self.jobs)
        lib().call("ofa_fold_batched", ctypes.addressof(arr), len(self.jobs), stream())
        self.jobs, self.keep, self.bytes, self.want_flush = [], [], 0, False   # (same stream: buffers may be reused)
        self.outs = set()


class ImmediateFold:
    """The FoldQueue's `add` for a caller whose partial rows are already written and that has no queue to defer to (gradients consumed
    from inside backward: data-parallel bucket all-reduces): one fold launch per job, same arithmetic and summation order."""

    def add(self, part, part_off, out, cols, stride, nslots, alpha=1.0, accumulate=True, flush_ok=True):
        assert part.dtype == torch.float32 and out.is_contiguous()
        job = (_FoldJob * 1)(_FoldJob(dptr(part) + part_off * 4, dptr(out), cols, stride, nslots, int(accumulate), float(alpha),
                                      dtype_code(out)))
        lib().call("ofa_fold_batched", ctypes.addressof(job), 1, stream())

    def flush_if_large(self):
        pass


def _u8(mask):
    """bool/uint8 mask as a contiguous uint8 tensor without a copy when possible."""
    mask = mask.contiguous()
    return mask.view(torch.uint8) if mask.dtype == torch.bool else mask.to(torch.uint8)


def _rows_cols(x):
    cols = x.shape[-1]
    return x.numel() // cols, cols


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps=1e-5, fuse_gelu=False):
    x = x.contiguous()
    rows, cols = _rows_cols(x)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib().call("ofa_gelu_layernorm_fwd" if fuse_gelu else "ofa_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y),
               ptr(mean), ptr(rstd), rows, cols, eps, dtype_code(x), stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, fuse_gelu=False, dgamma=None, dbeta=None, dbias=None, want_dbias=False,
                  fold=None, dres=None):
    """dgamma/dbeta given: the parameter gradients are ACCUMULATED into them (gradient arena); else fresh tensors.
    fuse_gelu + (dbias or want_dbias): also the column sums of dx (gradient of the bias of the Linear feeding the GELU)."""
    dy = dy.contiguous()
    x = x.contiguous()
    rows, cols = _rows_cols(x)
    dx = torch.empty_like(x)
    acc = dgamma is not None
    if not acc:
        dgamma = torch.empty(cols, dtype=gamma.dtype, device=x.device)
        dbeta = torch.empty(cols, dtype=gamma.dtype, device=x.device)
        if want_dbias:
            dbias = torch.empty(cols, dtype=gamma.dtype, device=x.device)
    assert dgamma.dtype == x.dtype and dbeta.dtype == x.dtype and (dbias is None or dbias.dtype == x.dtype)
    wsr = lib().cdll.ofa_layernorm_bwd_ws_rows()
    if fold is not None and acc:
        # deferred: keep this call's partial rows [q][nslots][cols] in a private buffer; FoldQueue.flush() reduces them
        ns = lib().cdll.ofa_layernorm_bwd_slots(rows, cols, dtype_code(x), int(fuse_gelu))
        nq = 3 if (fuse_gelu and dbias is not None) else 2
        ws = torch.empty(nq * ns * cols, dtype=torch.float32, device=x.device)
        for q, o in enumerate((dgamma, dbeta, dbias)[:nq]):
            fold.add(ws, q * ns * cols, o, cols, cols, ns)
        acc = DEFER_FOLD
    else:
        ws = workspace(wsr * cols * 4, x.device, "ln")
    if fuse_gelu:
        lib().call("ofa_gelu_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma),
                   ptr(dbeta), ptr(dbias), ptr(ws), rows, cols, int(acc), dtype_code(x), stream())
    else:
        if dres is not None:
            dres = dres.contiguous()
            assert dres.shape == x.shape and dres.dtype == x.dtype
        lib().call("ofa_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dgamma),
                   ptr(dbeta), ptr(ws), rows, cols, int(acc), dtype_code(x), stream())
    if acc == DEFER_FOLD:
        fold.flush_if_large()
    return dx, dgamma, dbeta, (dbias if fuse_gelu else None)


# ------------------------------------------------------------------ GEMM
def gemm(a, b, trans_a=False, trans_b=False, bias=None, bias_row=False, alpha=1.0, out=None, accumulate=False,
         out_f32=False, force_simple=False, a_kpad_zero=False, fold=None):
    """C = alpha * (op(a) @ op(b) + bias) [+ C].  a, b: 2-D (or 3-D batched) tensors whose last dim is contiguous."""
    assert a.dim() == b.dim() and a.dim() in (2, 3)
    batched = a.dim() == 3
    if a.stride(-1) != 1:
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    batch = a.shape[0] if batched else 1
    ar, ac = a.shape[-2], a.shape[-1]
    br, bc = b.shape[-2], b.shape[-1]
    M, K = (ac, ar) if trans_a else (ar, ac)
    N, Kb = (br, bc) if trans_b else (bc, br)
    if K != Kb:
        raise OfaError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    dt = dtype_code(a)
    if b.dtype != a.dtype:
        raise OfaError("gemm: operand dtypes differ")
    if out is None:
        odt = torch.float32 if (out_f32 or a.dtype == torch.float32) else a.dtype
        out = torch.empty((batch, M, N) if batched else (M, N), dtype=odt, device=a.device)
    flags = 0
    if bias is not None:
        flags |= GEMM_BIAS_ROW if bias_row else GEMM_BIAS_COL
    if accumulate:
        flags |= GEMM_ACCUM
    if force_simple:
        flags |= GEMM_FORCE_SIMPLE
    if dt != F32 and out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    if a_kpad_zero:
        flags |= 32   # OFA_GEMM_A_KPAD_ZERO
    lda, ldb, ldc = a.stride(-2), b.stride(-2), out.stride(-2)
    sa = a.stride(0) if batched else 0
    sb = b.stride(0) if batched else 0
    sc = out.stride(0) if batched else 0
    ws = None
    if fold is not None and bias is None and not batched and ldc == N and N % 4 == 0 and _prof is None:
        # split-K product whose result nobody reads yet (a weight gradient): leave the fp32 slabs [splits][M][N] in a
        # private buffer and let the FoldQueue reduce them (+ alpha / accumulate) in its batched launch
        nsp = lib().cdll.ofa_gemm_splits(M, N, K, int(trans_a), int(trans_b), 1, flags, dt, 256 << 20)
        if nsp > 1:
            ws = torch.empty(nsp * M * N, dtype=torch.float32, device=a.device)
            fold.add(ws, 0, out, M * N, M * N, nsp, alpha, accumulate)
            flags |= GEMM_DEFER_REDUCE
            ws_bytes = 256 << 20            # (the planner saw this budget; the slabs themselves fit by construction)
    if ws is None:
        ws = workspace(256 << 20, a.device, "gemm")
        ws_bytes = ws.numel() * 4
    args = ("ofa_gemm", ptr(a), ptr(b), ptr(out), ptr(bias), M, N, K, int(trans_a), int(trans_b), lda, ldb, ldc, batch,
            sa, sb, sc, 0, 0, 0, 0, float(alpha), flags, dt, ptr(ws), ws_bytes, stream())
    if _prof is not None:
        # roofline timing, IN SITU: two HIP events on the launch stream around THIS launch, inside the running step -- operands
        # as cold / warm as the step leaves them, no re-launch (an eager step is host-bound, so the stream is idle when the kernel
        # starts and the events bracket the kernel alone, + ~1 us of event overhead billed to it)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib().call(*args)
        e1.record()
        _prof.append((2.0 * M * N * K * batch, e0, e1, batch * ((M * K + K * N) * a.element_size() + M * N * out.element_size())))
    else:
        lib().call(*args)
    if flags & GEMM_DEFER_REDUCE:
        fold.flush_if_large()
    return out


def gemm_colstat(a, b, bias=None, max_groups=512):
    """out = a @ b^T (+ bias) like gemm(a, b, False, True), plus the per-column (sum, sum of squares) of the rounded output as partial
    rows [groups, 2, N] fp64 left by the kernel's epilogue (ofa_gemm_colstat), or None when the selected plan cannot produce them."""
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1] and a.dtype == b.dtype
    if a.stride(-1) != 1:
        a = a.contiguous()
    if b.stride(-1) != 1:
        b = b.contiguous()
    M, Kd, N = a.shape[0], a.shape[1], b.shape[0]
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if a.dtype == torch.float32 or N % 8:
        return gemm(a, b, False, True, bias=bias, out=out), None
    flags = GEMM_BIAS_COL if bias is not None else 0
    ws = workspace(256 << 20, a.device, "gemm")
    cap = min(max_groups, (M + 31) // 32)
    partial = torch.empty(cap, 2, N, dtype=torch.float64, device=a.device)
    groups = ctypes.c_int(0)
    args = ("ofa_gemm_colstat", ptr(a), ptr(b), ptr(out), ptr(bias), M, N, Kd, 0, 1, a.stride(0), b.stride(0), out.stride(0), 1.0,
            flags, dtype_code(a), ptr(ws), ws.numel() * 4, ptr(partial), cap, ctypes.addressof(groups), stream())
    if _prof is not None:                        # roofline timing in situ (see gemm)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib().call(*args)
        e1.record()
        _prof.append((2.0 * M * N * Kd, e0, e1, (M * Kd + Kd * N + M * N) * a.element_size()))
    else:
        lib().call(*args)
    return out, (partial[:groups.value] if groups.value > 0 else None)


class _GroupItem(ctypes.Structure):     # ofa_gemm_group_item
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p), ("slabs", ctypes.c_void_p), ("lda", ctypes.c_int64),
                ("ldb", ctypes.c_int64), ("m", ctypes.c_int32), ("n", ctypes.c_int32), ("k", ctypes.c_int32),
                ("splits", ctypes.c_int32), ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64), ("out_alpha", ctypes.c_float)]


GROUP_MAX = 16
GROUP_DIRECT = True      # (tests / A-B tools may clear it: every grouped product then goes through an fp32 slab and the fold)


def gemm_group_ok(dy, x, out):
    """Can out[M,N] (+)= dy[K,M]^T x[K,N] ride in a grouped launch (ofa_gemm_group_tn)?"""
    return (dy.dim() == 2 and x.dim() == 2 and dy.dtype in (torch.bfloat16, torch.float16) and x.dtype == dy.dtype
            and dy.stride(1) == 1 and x.stride(1) == 1 and dy.shape[0] == x.shape[0] and dy.shape[0] >= 1
            and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and out.is_contiguous())


def gemm_group_tn(products, fold):
    """products: [(dy [K,M], x [K,N], out [M,N], alpha)], each passing gemm_group_ok: out += alpha * dy^T x for all of them in
    ONE launch of 256 x 256 tiles (fp32 K-slice slabs) + the FoldQueue's batched reduce (`fold`: the queue to register the
    slabs with; the caller flushes it)."""
    assert 1 <= len(products) <= GROUP_MAX
    arr = (_GroupItem * len(products))()
    for it, (dy, x, out, alpha) in zip(arr, products):
        it.a, it.b, it.lda, it.ldb = dptr(dy), dptr(x), dy.stride(0), x.stride(0)
        it.m, it.n, it.k = dy.shape[1], x.shape[1], dy.shape[0]
    dt = dtype_code(products[0][0])
    lib().call("ofa_gemm_group_plan", ctypes.addressof(arr), len(products), dt)
    # a product the plan leaves ONE K-slice (a small micro-batch's weight gradients) is accumulated straight onto its 16-bit arena
    # gradient in the kernel's epilogue: no fp32 slab, no fold launch (the fold of a one-slab job only rounds and accumulates).
    # Arithmetic: round(alpha * dY^T X) to 16 bits, add the old 16-bit gradient, round again -- that is the REFERENCE's accumulation
    # over micro-batches (autograd hands AccumulateGrad a 16-bit gradient, `p.grad += g` adds in 16 bits: engine/trainer.py:766-784
    # under model.half() / bfloat16()); the slab + fold path (fp32 sum of the slices and the old gradient, ONE rounding) is tighter than
    # the reference, not the other way round.  tests/test_kernels_gpu.py::test_gemm_group_tn_direct_accumulation_over_micro_batches
    # bounds both against an fp64 sum.
    slabs, direct_outs = [], set()
    for it, (dy, x, out, alpha) in zip(arr, products):
        direct = (GROUP_DIRECT and it.splits == 1 and out.dtype == dy.dtype and out.dim() == 2 and out.stride(1) == 1 and out.stride(0) % 8 == 0
                  and out.data_ptr() % 16 == 0 and out.data_ptr() not in direct_outs)     # (two read-modify-writes of one output
        if direct:                                                                         #  must not share a launch)
            direct_outs.add(out.data_ptr())
            it.out, it.ldo, it.out_alpha = dptr(out), out.stride(0), float(alpha)
            slabs.append(None)
        else:
            sl = torch.empty(it.splits * it.m * it.n, dtype=torch.float32, device=dy.device)
            it.slabs = dptr(sl)
            slabs.append(sl)
    if _prof is not None:
        # roofline timing in situ (see gemm): the grouped launch alone; the fold of its K-slice slabs is a FoldQueue launch later on
        # and is accounted for in bench.py from the fold kernel's own share (rocprof), not here
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lib().call("ofa_gemm_group_tn", ctypes.addressof(arr), len(products), dt, stream())
        e1.record()
        es = products[0][0].element_size()
        _prof.append((sum(2.0 * it.m * it.n * it.k for it in arr), e0, e1,
                      sum((it.m + it.n) * it.k * es + (it.splits * it.m * it.n * 4 if sl is not None else 2 * it.m * it.n * es)
                          for it, sl in zip(arr, slabs))))
    else:
        lib().call("ofa_gemm_group_tn", ctypes.addressof(arr), len(products), dt, stream())
    for it, sl, (dy, x, out, alpha) in zip(arr, slabs, products):   # (registered after the launch: add() may flush the queue)
        if sl is not None:
            fold.add(sl, 0, out, it.m * it.n, it.m * it.n, it.splits, alpha, True)


# ---- optional per-launch timing of the GEMM kernel family (bench.py roofline): HIP events on the launch stream
_prof = None
_PROF_REPS = 1            # (round 1-2 re-launched every call 5x back to back with warm caches; since round 3 the launch itself is timed)


def gemm_profile_begin():
    global _prof
    _prof = []


def gemm_profile_end():
    global _prof
    torch.cuda.synchronize()
    rec, _prof = _prof, None
    return {"launches": len(rec), "flops": sum(r[0] for r in rec), "bytes": sum(r[3] for r in rec),
            "time_ms": sum(r[1].elapsed_time(r[2]) for r in rec) / _PROF_REPS, "reps": _PROF_REPS}


def gemm_heads(a, b, out, M, N, K, trans_a, trans_b, lda, ldb, ldc, B, heads, sa, sa2, sb, sb2, sc, sc2, alpha=1.0,
               accumulate=False):
    """Per-(batch, head) GEMMs straight on [B, T, heads*hd] rows (no permute copies): operand X of product (b, h)
    starts at X + b*sX2 + h*sX."""
    dt = dtype_code(a)
    flags = GEMM_ACCUM if accumulate else 0
    if dt != F32 and out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    ws = workspace(256 << 20, a.device, "gemm")
    lib().call("ofa_gemm", ptr(a), ptr(b), ptr(out), None, M, N, K, int(trans_a), int(trans_b), lda, ldb, ldc, B * heads,
               sa, sb, sc, heads, sa2, sb2, sc2, float(alpha), flags, dt, ptr(ws), ws.numel() * 4, stream())
    return out


# ------------------------------------------------------------------ softmax family
# (the reference's three pybind extensions, fused_kernels/fused_softmax.py:12-92: forward(inputs, [mask,] scale) and
#  backward(output_grads, softmax_results, scale), fp32 / bf16 / fp16)
def scaled_softmax(x, scale):
    x = x.contiguous()
    b, np_, sq, sk = x.shape
    y = torch.empty_like(x)
    lib().call("ofa_scaled_softmax_fwd", ptr(x), ptr(y), float(scale), b, np_, sq, sk, dtype_code(x), stream())
    return y


def scaled_softmax_bwd(dy, y, scale, inplace=False):
    dy = dy.contiguous()
    y = y.contiguous()
    rows = y.numel() // y.shape[-1]
    dx = dy if inplace else torch.empty_like(dy)
    lib().call("ofa_scaled_softmax_bwd", ptr(dy), ptr(y), ptr(dx), float(scale), 1, 1, rows, y.shape[-1], dtype_code(y),
               stream())
    return dx


def scaled_masked_softmax(x, mask, scale):
    x = x.contiguous()
    mask = mask.to(torch.uint8).contiguous()
    b, np_, sq, sk = x.shape
    y = torch.empty_like(x)
    lib().call("ofa_scaled_masked_softmax_fwd", ptr(x), ptr(mask), ptr(y), float(scale), b, np_, sq, sk, mask.shape[0],
               dtype_code(x), stream())
    return y


def scaled_masked_softmax_bwd(dy, y, scale, inplace=True):
    """In place on dy by default, as the reference's extension (scaled_masked_softmax_cuda.cu:98-117)."""
    assert dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape and y.dim() == 4
    b, np_, sq, sk = y.shape
    dx = dy if inplace else torch.empty_like(dy)
    lib().call("ofa_scaled_masked_softmax_bwd", ptr(dy), ptr(y), ptr(dx), float(scale), b, np_, sq, sk, dtype_code(y),
               stream())
    return dx


def scaled_upper_triang_masked_softmax(x, scale):
    x = x.contiguous()
    ab, sq, sk = x.shape
    assert sq == sk
    y = torch.empty_like(x)
    lib().call("ofa_scaled_upper_triang_masked_softmax_fwd", ptr(x), ptr(y), float(scale), ab, sq, dtype_code(x), stream())
    return y


def scaled_upper_triang_masked_softmax_bwd(dy, y, scale, inplace=True):
    """In place on dy by default (scaled_upper_triang_masked_softmax_cuda.cu:68-95); dx is zero above the diagonal."""
    assert dy.is_contiguous() and y.is_contiguous() and dy.shape == y.shape and y.dim() == 3 and y.shape[1] == y.shape[2]
    dx = dy if inplace else torch.empty_like(dy)
    lib().call("ofa_scaled_upper_triang_masked_softmax_bwd", ptr(dy), ptr(y), ptr(dx), float(scale), y.shape[0], y.shape[1],
               dtype_code(y), stream())
    return dx


def get_batch_per_block(sq, sk, b, np_):
    return lib().cdll.ofa_get_batch_per_block(sq, sk, b, np_)


def attn_softmax(x, bias, kpm, scale, heads, causal):
    x = x.contiguous()
    BA, T, S = x.shape
    if bias is not None:
        bias = bias.contiguous()
    if kpm is not None:
        kpm = _u8(kpm)
    p = torch.empty_like(x)
    lib().call("ofa_attn_softmax_fwd", ptr(x), ptr(bias), ptr(kpm), ptr(p), float(scale), BA, heads, T, S, int(causal),
               dtype_code(x), stream())
    return p


# ------------------------------------------------------------------ fused attention (bf16)
def pad32(n):
    return (n + 31) // 32 * 32


def transpose_heads(x, T_pad):
    """x: [B, T, C] (last dim contiguous) -> [B, C, T_pad], zero for t >= T."""
    B, T, C = x.shape
    if x.stride(-1) != 1 or x.stride(0) != T * x.stride(1):
        x = x.contiguous()
    xt = torch.empty(B, C, T_pad, dtype=x.dtype, device=x.device)
    lib().call("ofa_transpose_heads", ptr(x), ptr(xt), B, T, C, T_pad, x.stride(1), dtype_code(x), stream())
    return xt


def _rows3(x):
    """[B, T, D] view with contiguous last dim and dense batch stride; returns (tensor, ld)."""
    if x.stride(-1) != 1 or x.stride(0) != x.shape[1] * x.stride(1):
        x = x.contiguous()
    return x, x.stride(1)


def _same_ld(k, v):
    k, ldk = _rows3(k)
    v, ldv = _rows3(v)
    if ldv != ldk:
        k, v = k.contiguous(), v.contiguous()
        ldk = k.stride(1)
    return k, v, ldk


def _c_dtype(c_attn):
    """dtype code of the per-head scale (fp32 or bf16 [heads], contiguous); fp32 when absent."""
    if c_attn is None:
        return dtype_code(torch.empty(0, dtype=torch.float32))
    assert c_attn.is_contiguous()
    return dtype_code(c_attn)


def c_attn_grad(delta, c_attn, B, heads, T, out=None, accumulate=False):
    """d c_attn[h] = sum_{b,t<T} delta[b*heads+h, t] / c_attn[h]  (delta: fp32 [B*heads, ld] from attn_bwd)."""
    if out is None:
        out = torch.empty_like(c_attn)
        accumulate = False
    lib().call("ofa_c_attn_grad", ptr(delta), ptr(c_attn), ptr(out), B, heads, T, delta.stride(0), int(accumulate),
               dtype_code(c_attn), stream())
    return out


def _shared_bias(bias, heads, q):
    """A batch-shared position bias: [heads, Tb, Sb] (the reference's [B*A, T, S] tensor is B copies of it), contiguous, q's dtype."""
    assert bias.dim() == 3 and bias.shape[0] == heads, (tuple(bias.shape), heads)
    if bias.dtype != q.dtype:
        bias = bias.to(q.dtype)
    return bias.contiguous()


class _BiasSlots(ctypes.Structure):
    _fields_ = [("values", ctypes.c_void_p * 8), ("values2", ctypes.c_void_p * 8), ("start", ctypes.c_int32 * 8), ("n", ctypes.c_int32 * 8),
                ("inner", ctypes.c_int32 * 8), ("count", ctypes.c_int32)]


def bias_build(abs_bias, starts=(), values=(), heads=None, want_out=True):
    """The batch-shared position bias of one layer (ofa_bias_build): out = abs_bias [A, Tb, Sb] (None: zeros) with values_k
    [n_k, n_k, A] added on the diagonal block at starts[k] -> (out [A, Tb, Sb] or None, (row image, column image)): the two
    tile-swizzled images the shared-bias attention kernels read."""
    live = [(s, v) for s, v in zip(starts, values) if v is not None]
    ref = abs_bias if abs_bias is not None else (live[0][1][0] if isinstance(live[0][1], tuple) else live[0][1])
    slots = _BiasSlots()
    assert len(live) <= 8, "at most 8 slots per bias"
    keep, ends = [], []
    for i, (s, v) in enumerate(live):
        if isinstance(v, tuple):             # an OUTER slot (frames [F,F,A], patches [P,P,A]): value(i, j) = frames[i/P][j/P] + patches[i%P][j%P]
            vf, vi = (t.to(ref.dtype).contiguous() for t in v)
            keep += [vf, vi]
            n = vf.shape[0] * vi.shape[0]
            slots.values[i], slots.values2[i], slots.inner[i] = dptr(vf), dptr(vi), vi.shape[0]
        else:
            v = v.to(ref.dtype).contiguous()
            keep.append(v)
            n = v.shape[0]
            slots.values[i], slots.values2[i], slots.inner[i] = dptr(v), None, 0
        slots.start[i], slots.n[i] = int(s), n
        ends.append(int(s) + n)
    slots.count = len(live)
    if abs_bias is not None:
        abs_bias = abs_bias.contiguous()
        A, Tb, Sb = abs_bias.shape
    else:
        A = heads
        Tb = Sb = max(ends)
    n = lib().cdll.ofa_bias_swz_elems(A, Tb, Sb)
    swz = torch.empty(2, n, dtype=ref.dtype, device=ref.device)
    out = torch.empty(A, Tb, Sb, dtype=ref.dtype, device=ref.device) if want_out else None
    lib().call("ofa_bias_build", ptr(abs_bias), ctypes.addressof(slots), ptr(out), ptr(swz[0]), ptr(swz[1]), A, Tb, Sb,
               dtype_code(ref), stream())
    return out, (swz[0], swz[1])


def _shared_swz(bias, bias_shared):
    """bias_shared: True (the images are built here, one extra launch) or the (row image, column image) pair of bias_build."""
    if isinstance(bias_shared, tuple):
        return bias_shared
    return bias_build(bias, want_out=False)[1]


def attn_fwd(q, k, v, heads, scale, bias=None, kpm=None, c_attn=None, causal=False, seg=None, bias_shared=False):
    """q [B,T,D], k, v [B,S,D] (row views of a packed buffer are fine) -> out [B,T,D], lse [B*heads, Tpad].
    seg (packing.Segments): ragged mode -- q [1,rows_q,D], k, v [1,rows_k,D] hold the samples back to back, out rows outside
    every segment are zero, lse is [heads, pad32(rows_q)] by packed row.
    bias_shared: bias is [heads, Tb, Sb], the same for every sample, indexed by the position inside the sample (ofa_attn_sbias_*)."""
    q, ldq = _rows3(q)
    k, v, ldk = _same_ld(k, v)
    B, T, D = q.shape
    S = k.shape[1]
    Tpad = pad32(T)
    if bias_shared:
        bias = _shared_bias(bias, heads, q)
        Tb, Sb = bias.shape[1], bias.shape[2]
        bias = _shared_swz(bias, bias_shared)[0]            # the kernel reads the row image
        if seg is not None:
            assert B == 1 and kpm is None and seg.rows_q == T and seg.rows_k == S and Tb >= seg.max_q - 31 and Sb >= seg.max_k - 31
            out = torch.empty(1, T, D, dtype=q.dtype, device=q.device)
            lse = torch.empty(heads, Tpad, dtype=torch.float32, device=q.device)
            lib().call("ofa_attn_sbias_fwd", ptr(q), ptr(k), ptr(v), ptr(bias), Tb, Sb, None, ptr(c_attn), _c_dtype(c_attn), ptr(out),
                       ptr(lse), seg.batch, heads, min(seg.max_q, Tb), min(seg.max_k, Sb), Tpad, ldq, ldk, D, float(scale), int(causal),
                       ptr(seg.table), T, S, dtype_code(q), stream())
            return out, lse
        out = torch.empty(B, T, D, dtype=q.dtype, device=q.device)
        lse = torch.empty(B * heads, Tpad, dtype=torch.float32, device=q.device)
        if kpm is not None:
            kpm = _u8(kpm)
        lib().call("ofa_attn_sbias_fwd", ptr(q), ptr(k), ptr(v), ptr(bias), Tb, Sb, ptr(kpm), ptr(c_attn), _c_dtype(c_attn), ptr(out),
                   ptr(lse), B, heads, T, S, Tpad, ldq, ldk, D, float(scale), int(causal), None, 0, 0, dtype_code(q), stream())
        return out, lse
    if seg is not None:
        assert B == 1 and bias is None and kpm is None and seg.rows_q == T and seg.rows_k == S, (B, T, S, seg.rows_q, seg.rows_k)
        out = torch.empty(1, T, D, dtype=q.dtype, device=q.device)          # filler rows are zeroed by the kernel
        lse = torch.empty(heads, Tpad, dtype=torch.float32, device=q.device)
        lib().call("ofa_attn_fwd", ptr(q), ptr(k), ptr(v), None, None, ptr(c_attn), _c_dtype(c_attn), ptr(out), ptr(lse),
                   seg.batch, heads, seg.max_q, seg.max_k, Tpad, ldq, ldk, D, float(scale), int(causal), ptr(seg.table),
                   T, S, dtype_code(q), stream())
        return out, lse
    out = torch.empty(B, T, D, dtype=q.dtype, device=q.device)
    lse = torch.empty(B * heads, Tpad, dtype=torch.float32, device=q.device)
    if bias is not None:
        bias = bias.contiguous()
    if kpm is not None:
        kpm = _u8(kpm)
    lib().call("ofa_attn_fwd", ptr(q), ptr(k), ptr(v), ptr(bias), ptr(kpm), ptr(c_attn), _c_dtype(c_attn), ptr(out), ptr(lse),
               B, heads, T, S, Tpad, ldq, ldk, D, float(scale), int(causal), None, 0, 0, dtype_code(q), stream())
    return out, lse


def attn_cs_slots(B, T, seg=None, k_side=False):
    """Partial rows of the column sums the attention backward kernels leave per call (ofa_attn_bwd_cs): one per (sample, 128-row tile)."""
    if seg is not None:
        B, T = seg.batch, (seg.max_k if k_side else seg.max_q)
    return lib().cdll.ofa_attn_cs_slots(int(B), int(T))


def _cs_args(cs, heads):
    """cs = dict(q=, k=, v=, c=): fp32 2-D views [slots, heads * 64] (row stride free, k and v the same; c: [slots, heads] contiguous)."""
    q, k, v, c = cs.get("q"), cs.get("k"), cs.get("v"), cs.get("c")
    for t in (q, k, v):
        assert t is None or (t.dtype == torch.float32 and t.stride(1) == 1 and t.shape[1] == heads * 64)
    assert c is None or (c.dtype == torch.float32 and c.is_contiguous() and c.shape[1] == heads)
    assert k is None or v is None or k.stride(0) == v.stride(0)
    kv = k if k is not None else v
    return (ptr(q), q.stride(0) if q is not None else 0, ptr(k), ptr(v), kv.stride(0) if kv is not None else 0, ptr(c))


def attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=None, kpm=None, c_attn=None, causal=False, need_dbias=False,
             outs=None, seg=None, bias_shared=False, dbias_dtype=torch.float32, cs=None):
    """outs=(dq, dk, dv): caller-provided gradient views with the SAME row strides as q / k (e.g. column slices of one
    packed [B,T,3D] buffer next to a packed qkv input) -- the kernels write them in place.
    cs: optional dict of fp32 partial-row buffers (see _cs_args / attn_cs_slots) that receive the column sums of dq / dk / dv and
    the per-head sums of delta / c_attn -- the projections' bias gradients and the c_attn gradient, finished by the FoldQueue."""
    q, ldq = _rows3(q)
    k, v, ldk = _same_ld(k, v)
    dout, ldo = _rows3(dout)
    out, ldo2 = _rows3(out)
    if ldo != ldo2:
        dout, out = dout.contiguous(), out.contiguous()
        ldo = dout.stride(1)
    B, T, D = q.shape
    S = k.shape[1]
    Tpad = pad32(T)
    delta = torch.empty(B * heads, Tpad, dtype=torch.float32, device=q.device)   # rowsum(dO*O): written by the dQ kernel
    dbias = torch.empty(B * heads, T, S, dtype=q.dtype, device=q.device) if (need_dbias and not bias_shared) else None
    if bias is not None and not bias_shared:
        bias = bias.contiguous()
    if kpm is not None:
        kpm = _u8(kpm)
    if outs is not None:
        dq, dk, dv = outs
        assert dq.stride(1) == ldq and dk.stride(1) == ldk and dv.stride(1) == ldk
    else:
        # dq is written with ld = ldq, dk/dv with ld = ldk: give the kernel dense outputs by passing dense strides
        if ldq != D:
            q = q.contiguous()
            ldq = D
        if ldk != D:
            k, v = k.contiguous(), v.contiguous()
            ldk = D
        dq = torch.empty(B, T, D, dtype=q.dtype, device=q.device)     # (ragged mode: the kernels zero the filler rows)
        dk = torch.empty(B, S, D, dtype=q.dtype, device=q.device)
        dv = torch.empty(B, S, D, dtype=q.dtype, device=q.device)
    csa = _cs_args(cs, heads) if cs else ()
    fn = "ofa_attn_bwd_cs" if cs else "ofa_attn_bwd"
    if bias_shared:
        # dbias (when asked for): fp32 [heads, Tb, Sb] = the sum over the batch of dS, from the batch-walking third kernel
        bias = _shared_bias(bias, heads, q)
        Tb, Sb = bias.shape[1], bias.shape[2]
        swz_row, swz_col = _shared_swz(bias, bias_shared)
        dbias = torch.empty(heads, Tb, Sb, dtype=dbias_dtype, device=q.device) if need_dbias else None   # (16-bit: cast by the chunk fold)
        ws, ws_bytes = None, 0
        if need_dbias:
            nchunk = lib().cdll.ofa_attn_sbias_chunks(B if seg is None else seg.batch, heads, Tb, Sb)
            if nchunk > 1 or dbias_dtype not in (torch.float32, q.dtype):
                ws = torch.empty(nchunk, heads, Tb, Sb, dtype=torch.float32, device=q.device)
                ws_bytes = ws.numel() * 4
        if seg is not None:
            assert B == 1 and kpm is None and seg.rows_q == T and seg.rows_k == S
            dims = (seg.batch, heads, min(seg.max_q, Tb), min(seg.max_k, Sb), Tpad, ldq, ldk, ldo, float(scale), int(causal), ptr(seg.table), T, S)
            kp = None
        else:
            dims = (B, heads, T, S, Tpad, ldq, ldk, ldo, float(scale), int(causal), None, 0, 0)
            kp = ptr(kpm)
        lib().call("ofa_attn_sbias_bwd_cs" if cs else "ofa_attn_sbias_bwd", ptr(q), ptr(k), ptr(v), ptr(dout), ptr(bias), ptr(swz_row),
                   ptr(swz_col), Tb, Sb, kp, ptr(c_attn), _c_dtype(c_attn), ptr(lse),
                   ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), ptr(dbias), dtype_code(dbias) if dbias is not None else F32, ptr(ws),
                   ws_bytes, *dims, dtype_code(q), *csa, stream())
        return dq, dk, dv, dbias, delta
    if seg is not None:
        assert B == 1 and bias is None and kpm is None and not need_dbias and seg.rows_q == T and seg.rows_k == S
        lib().call(fn, ptr(q), ptr(k), ptr(v), ptr(dout), None, None, ptr(c_attn), _c_dtype(c_attn), ptr(lse),
                   ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), None, seg.batch, heads, seg.max_q, seg.max_k, Tpad, ldq, ldk, ldo,
                   float(scale), int(causal), ptr(seg.table), T, S, dtype_code(q), *csa, stream())
        return dq, dk, dv, None, delta
    lib().call(fn, ptr(q), ptr(k), ptr(v), ptr(dout), ptr(bias), ptr(kpm), ptr(c_attn), _c_dtype(c_attn), ptr(lse),
               ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), ptr(dbias), B, heads, T, S, Tpad, ldq, ldk, ldo, float(scale),
               int(causal), None, 0, 0, dtype_code(q), *csa, stream())
    return dq, dk, dv, dbias, delta


def attn_decode(q, k_cache, v_cache, S, heads, scale, bias=None, kpm=None, c_attn=None, need_probs=False):
    """One query row per batch entry against a KV cache (csrc/attention_decode.hip).
    q: [B, D]; k_cache, v_cache: [B, capacity, D] (same strides, last dim contiguous), the first S rows valid;
    bias: [B*heads, S] or None; kpm: bool/uint8 [B, >= S] or None.  Returns out [B, D] and probs [B*heads, S] (or None)."""
    B, D = q.shape
    q = q.contiguous()
    assert k_cache.dim() == 3 and k_cache.shape == v_cache.shape and k_cache.stride() == v_cache.stride()
    assert k_cache.stride(2) == 1 and k_cache.shape[0] == B and k_cache.shape[1] >= S and k_cache.dtype == q.dtype
    out = torch.empty(B, D, dtype=q.dtype, device=q.device)
    probs = torch.empty(B * heads, S, dtype=q.dtype, device=q.device) if need_probs else None
    if bias is not None:
        bias = bias.reshape(B * heads, S).to(q.dtype).contiguous()
    kpm_ld = 0
    if kpm is not None:
        kpm = _u8(kpm)
        kpm_ld = kpm.stride(0)
    lib().call("ofa_attn_decode", ptr(q), ptr(k_cache), ptr(v_cache), ptr(bias), ptr(kpm), ptr(c_attn), _c_dtype(c_attn),
               ptr(out), ptr(probs), B, heads, D // heads, S, k_cache.stride(1), k_cache.stride(0), kpm_ld, float(scale),
               dtype_code(q), stream())
    return out, probs


# ------------------------------------------------------------------ elementwise / embedding
def gelu_fwd(x):
    x = x.contiguous()
    y = torch.empty_like(x)
    lib().call("ofa_gelu_fwd", ptr(x), ptr(y), x.numel(), dtype_code(x), stream())
    return y


def gelu_bwd(dy, x):
    dy, x = dy.contiguous(), x.contiguous()
    dx = torch.empty_like(x)
    lib().call("ofa_gelu_bwd", ptr(dy), ptr(x), ptr(dx), x.numel(), dtype_code(x), stream())
    return dx


def dropout_add(x, residual, p, seed, offset, offset_base=None):
    """offset_base: optional device int64[1] added to `offset` on the device (graph-safe Philox stream position)."""
    x = x.contiguous()
    if residual is not None:
        residual = residual.contiguous()
    y = torch.empty_like(x)
    lib().call("ofa_dropout_add_fwd", ptr(x), ptr(residual), ptr(y), x.numel(), float(p), seed, offset, ptr(offset_base),
               dtype_code(x), stream())
    return y


def dropout_bwd(dy, p, seed, offset, offset_base=None):
    dy = dy.contiguous()
    dx = torch.empty_like(dy)
    lib().call("ofa_dropout_bwd", ptr(dy), ptr(dx), dy.numel(), float(p), seed, offset, ptr(offset_base), dtype_code(dy),
               stream())
    return dx


def add_rowvec_mask(a, b=None, vec=None, rowmask=None):
    """b: same rows as a, or the rows of ONE sample (a's row count a multiple of b's: positions shared by the batch)."""
    a = a.contiguous()
    rows, cols = _rows_cols(a)
    period = 0
    if b is not None:
        b = b.contiguous()
        brows = b.numel() // cols
        if brows != rows:
            assert brows > 0 and rows % brows == 0, (rows, brows)
            period = brows
    if rowmask is not None:
        rowmask = _u8(rowmask)
    y = torch.empty_like(a)
    lib().call("ofa_add_rowvec_mask", ptr(a), ptr(b), ptr(vec), ptr(rowmask), ptr(y), rows, cols, period, dtype_code(a), stream())
    return y


def batch_sum(x, batch, out=None, accumulate=False):
    """x [batch * n] (contiguous) -> out [n] = sum over the batch (fp32 accumulation)."""
    x = x.contiguous()
    n = x.numel() // batch
    if out is None:
        out = torch.empty(n, dtype=x.dtype, device=x.device)
        accumulate = False
    lib().call("ofa_batch_sum", ptr(x), ptr(out), int(batch), n, int(bool(accumulate)), dtype_code(x), stream())
    return out


def embedding_fwd(weight, ids, pad_mask_of=None):
    """pad_mask_of: a token id -- also returns the bool mask ids == pad_mask_of, written by the same kernel."""
    ids = ids.contiguous()
    V, D = weight.shape
    out = torch.empty(*ids.shape, D, dtype=weight.dtype, device=weight.device)
    mask = None
    if pad_mask_of is not None and D % (4 if weight.dtype == torch.float32 else 8) == 0:
        mask = torch.empty(ids.shape, dtype=torch.bool, device=weight.device)
    lib().call("ofa_embedding_fwd", ptr(weight), ptr(ids), ptr(out), ids.numel(), D, V, ptr(mask), int(pad_mask_of) if mask is not None else 0,
               dtype_code(weight), stream())
    if pad_mask_of is None:
        return out
    return out, (mask if mask is not None else ids.eq(pad_mask_of))


def gather_rows(src2d, index):
    """out[r] = index[r] >= 0 ? src2d[index[r]] : 0  (index: int64 [n] on the device)."""
    assert src2d.dim() == 2 and src2d.is_contiguous() and index.dtype == torch.int64 and index.is_contiguous()
    out = torch.empty(index.numel(), src2d.shape[1], dtype=src2d.dtype, device=src2d.device)
    lib().call("ofa_gather_rows", ptr(src2d), ptr(index), ptr(out), index.numel(), src2d.shape[1], src2d.shape[0],
               dtype_code(src2d), stream())
    return out


def gather_rows_parts(parts, index):
    """Packed rows of the virtual concatenation torch.cat(parts, dim=1) (parts: contiguous [B, n_k, D], <= 8 of them): index[r] = b * T + t."""
    assert 1 <= len(parts) <= 8 and index.dtype == torch.int64 and index.is_contiguous()
    B, D = parts[0].shape[0], parts[0].shape[2]
    assert all(t.dim() == 3 and t.is_contiguous() and t.shape[0] == B and t.shape[2] == D and t.dtype == parts[0].dtype for t in parts)
    out = torch.empty(index.numel(), D, dtype=parts[0].dtype, device=parts[0].device)
    srcs = (ctypes.c_void_p * len(parts))(*[dptr(t) for t in parts])
    lens = (ctypes.c_int * len(parts))(*[int(t.shape[1]) for t in parts])
    lib().call("ofa_gather_rows_parts", ctypes.addressof(srcs), ctypes.addressof(lens), len(parts), ptr(index), ptr(out), index.numel(), D, B,
               dtype_code(parts[0]), stream())
    return out


def scatter_rows_part(dpacked, inverse, B, nk, Ttot, start):
    """[B, nk, D] gradient of one part: row (b, j) = dpacked[inverse[b * Ttot + start + j]] or 0."""
    assert dpacked.dim() == 2 and dpacked.is_contiguous() and inverse.dtype == torch.int64 and inverse.is_contiguous()
    D = dpacked.shape[1]
    out = torch.empty(B, nk, D, dtype=dpacked.dtype, device=dpacked.device)
    lib().call("ofa_scatter_rows_part", ptr(dpacked), ptr(inverse), ptr(out), B, nk, Ttot, start, D, dpacked.shape[0], dtype_code(dpacked), stream())
    return out


def embedding_bwd(dout, ids, V, padding_idx=-1, dweight=None):
    dout = dout.contiguous()
    ids = ids.contiguous()
    D = dout.shape[-1]
    if dweight is None:
        dweight = torch.zeros(V, D, dtype=dout.dtype, device=dout.device)
    present = workspace(V, dout.device, "embed_present") if V > 4096 else None
    S = lib().cdll.ofa_embedding_bwd_slices(V, D)
    slices = workspace(S * V * D * 4, dout.device, "embed_slices") if S > 1 else None
    lib().call("ofa_embedding_bwd", ptr(dout), ptr(ids), ptr(dweight), ids.numel(), D, V,
               -1 if padding_idx is None else padding_idx, ptr(present), ptr(slices), dtype_code(dout), stream())
    return dweight


def segment_rowsum(dout2d, plan, dweight, accumulate):
    """dweight[plan.rows[s]] (+)= sum of the rows of dout2d in segment s (ofa_segment_rowsum); plan: ops.SegmentPlan."""
    assert dout2d.dim() == 2 and dout2d.is_contiguous() and dweight.is_contiguous() and dout2d.dtype == dweight.dtype
    lib().call("ofa_segment_rowsum", ptr(dout2d), ptr(plan.order), ptr(plan.seg_off), ptr(plan.seg_row), ptr(dweight), plan.nseg,
               dout2d.shape[1], int(accumulate), dtype_code(dout2d), stream())
    return dweight


def im2col_patch(img, p, Kpad, lead=0):
    """[B, C, H, W] -> [B * (lead + N), Kpad]: every sample's N patch rows behind `lead` zero rows."""
    img = img.contiguous()
    B, C, H, W = img.shape
    col = torch.empty(B * (lead + (H // p) * (W // p)), Kpad, dtype=img.dtype, device=img.device)
    lib().call("ofa_im2col_patch", ptr(img), ptr(col), B, C, H, W, p, Kpad, int(lead), dtype_code(img), stream())
    return col


# ------------------------------------------------------------------ criterion / optimizer
def step_stats_add(stats, loss, target, pad):
    """stats (fp64 [3]: sample_size, loss_sum, ntokens) += (non-pad targets, loss, non-pad targets): one launch."""
    assert stats.dtype == torch.float64 and stats.numel() >= 3 and loss.dtype == torch.float32 and loss.numel() == 1
    t = target.reshape(-1)
    assert t.dtype == torch.int64 and t.is_contiguous()
    lib().call("ofa_step_stats_add", ptr(stats), ptr(loss), ptr(t), t.numel(), int(pad), stream())


def cross_entropy_fwd(logits2d, target, V, ignore_index):
    """logits2d: [rows, ld] storage (ld >= V, multiple of the vector width); returns lse [rows], row_loss [rows]."""
    rows, ld = logits2d.shape[0], logits2d.stride(0)
    lse = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    lib().call("ofa_cross_entropy_fwd", ptr(logits2d), ptr(target), ptr(lse), ptr(row_loss), rows, V, ld, ignore_index,
               dtype_code(logits2d), stream())
    return lse, row_loss


def cross_entropy_fwd_grad_ok(logits2d, V):
    return logits2d.is_cuda and bool(lib().cdll.ofa_cross_entropy_fwd_grad_ok(int(V), int(logits2d.stride(0)), dtype_code(logits2d)))


def cross_entropy_fwd_grad(logits2d, target, grad_scale, V, ignore_index):
    """lse, row_loss AND dlogits = (softmax - onehot) * grad_scale[0] in one pass over the logits (ofa_cross_entropy_fwd_grad):
    grad_scale is the fp32 device scalar the backward pass will be seeded with (None: 1)."""
    rows, ld = logits2d.shape[0], logits2d.stride(0)
    lse = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    dlogits = torch.empty(rows, ld, dtype=logits2d.dtype, device=logits2d.device)
    lib().call("ofa_cross_entropy_fwd_grad", ptr(logits2d), ptr(target), ptr(grad_scale), ptr(lse), ptr(row_loss), ptr(dlogits), rows, V,
               ld, ignore_index, dtype_code(logits2d), stream())
    return lse, row_loss, dlogits


def cross_entropy_bwd(logits2d, target, lse, grad_scale, V, ignore_index, dlogits=None):
    rows, ld = logits2d.shape[0], logits2d.stride(0)
    if dlogits is None:
        dlogits = torch.empty(rows, ld, dtype=logits2d.dtype, device=logits2d.device)
    lib().call("ofa_cross_entropy_bwd", ptr(logits2d), ptr(target), ptr(lse), ptr(grad_scale), ptr(dlogits), rows, V, ld,
               ignore_index, dtype_code(logits2d), stream())
    return dlogits


def ls_cross_entropy_fwd(logits2d, target, V, ignore_index, eps, cstart=-1, cend=-1, cmask=None):
    """label-smoothed CE rows: returns lse, row_loss, row_nll, row_cnt (fp32 [rows])."""
    rows, ld = logits2d.shape[0], logits2d.stride(0)
    out = [torch.empty(rows, dtype=torch.float32, device=logits2d.device) for _ in range(4)]
    if cmask is not None:
        cmask = _u8(cmask)
    lib().call("ofa_ls_cross_entropy_fwd", ptr(logits2d), ptr(target), *[ptr(o) for o in out], rows, V, ld, ignore_index,
               float(eps), int(cstart), int(cend), ptr(cmask), dtype_code(logits2d), stream())
    return out


def ls_cross_entropy_bwd(logits2d, target, lse, row_cnt, row_w, grad_scale, V, ignore_index, eps, cstart=-1, cend=-1,
                         cmask=None):
    rows, ld = logits2d.shape[0], logits2d.stride(0)
    dlogits = torch.empty(rows, ld, dtype=logits2d.dtype, device=logits2d.device)
    if cmask is not None:
        cmask = _u8(cmask)
    lib().call("ofa_ls_cross_entropy_bwd", ptr(logits2d), ptr(target), ptr(lse), ptr(row_cnt), ptr(row_w), ptr(grad_scale),
               ptr(dlogits), rows, V, ld, ignore_index, float(eps), int(cstart), int(cend), ptr(cmask),
               dtype_code(logits2d), stream())
    return dlogits


def probs_fwd(logits2d, V, ld, log_probs):
    rows = logits2d.shape[0]
    y = torch.empty(rows, V, dtype=torch.float32, device=logits2d.device)
    lib().call("ofa_probs_fwd", ptr(logits2d), ptr(y), rows, V, ld, int(log_probs), dtype_code(logits2d), stream())
    return y


def probs_bwd(dy, y, V, dtype, log_probs):
    rows = y.shape[0]
    ld = (V + 7) // 8 * 8
    d = torch.empty(rows, ld, dtype=dtype, device=y.device)
    lib().call("ofa_probs_bwd", ptr(dy), ptr(y), ptr(d), rows, V, ld, int(log_probs), dtype_code(d), stream())
    return d


def sumsq(x, out):
    """out (fp32[1]) += sum(x*x)."""
    x = x.contiguous()
    ws = workspace(lib().cdll.ofa_sumsq_ws_floats() * 4, x.device, "sumsq")
    lib().call("ofa_sumsq", ptr(x), ptr(out), ptr(ws), x.numel(), dtype_code(x), stream())
    return out


def step_schedule(gsq, sample_size, step, lr, sched, gnorm, clip_norm, beta1, beta2):
    """Device-side scalar schedule of one update (see ofa_step_schedule); all arguments are 1-element device tensors
    (gsq fp32, sample_size / step / lr fp64, sched fp32[3], gnorm fp32)."""
    lib().call("ofa_step_schedule", ptr(gsq), ptr(sample_size), ptr(step), ptr(lr), ptr(sched), ptr(gnorm), float(clip_norm),
               float(beta1), float(beta2), stream())


def step_schedule_scaled(gsq, sample_size, step, lr, sched, gnorm, loss_scaler, clip_norm, beta1, beta2, scale_factor=2.0,
                         scale_window=2000, tolerance=0.0, threshold=None, min_loss_scale=1e-4):
    """step_schedule with the reference's dynamic loss scaler (see ofa_step_schedule_scaled); loss_scaler: fp64[8] device state."""
    lib().call("ofa_step_schedule_scaled", ptr(gsq), ptr(sample_size), ptr(step), ptr(lr), ptr(sched), ptr(gnorm), ptr(loss_scaler),
               float(clip_norm), float(beta1), float(beta2), float(scale_factor), float(scale_window), float(tolerance),
               float(threshold or 0.0), float(min_loss_scale), stream())


def adam_step(master, exp_avg, exp_avg_sq, grad, model_param, coef, lr, beta1, beta2, eps, weight_decay, step):
    lib().call("ofa_adam_step", ptr(master), ptr(exp_avg), ptr(exp_avg_sq), ptr(grad), ptr(model_param), ptr(coef),
               master.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
               dtype_code(grad), stream())


def colsum(x, alpha=1.0, out=None, accumulate=False, out_dtype=torch.float32, fold=None):
    """Column sums of a 2-D tensor (last dim contiguous): fresh tensor of `out_dtype`, or (accumulated) into `out`.
    fold (with out): only the row-group partials are computed now, the FoldQueue finishes the sum at its flush."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=out_dtype, device=x.device)
        fold = None
    if fold is not None and out.is_contiguous():
        ns = lib().cdll.ofa_colsum_slots(rows)
        ws = torch.empty(ns * cols, dtype=torch.float32, device=x.device)
        fold.add(ws, 0, out, cols, cols, ns, alpha, accumulate)
        lib().call("ofa_colsum", ptr(x), ptr(out), ptr(ws), rows, cols, x.stride(0), float(alpha), DEFER_FOLD,
                   dtype_code(x), dtype_code(out), stream())
        fold.flush_if_large()
        return out
    ws = workspace(lib().cdll.ofa_colsum_ws_floats(cols) * 4, x.device, "colsum")
    lib().call("ofa_colsum", ptr(x), ptr(out), ptr(ws), rows, cols, x.stride(0), float(alpha), int(accumulate),
               dtype_code(x), dtype_code(out), stream())
    return out


def mul(a, b):
    a, b = a.contiguous(), b.contiguous()
    rows, cols = _rows_cols(a)
    y = torch.empty_like(a)
    lib().call("ofa_mul", ptr(a), ptr(b), ptr(y), rows, cols, 0, dtype_code(a), stream())
    return y


def add_n(tensors):
    """sum of up to 16 same-shaped tensors in one launch (fp32 accumulation, one rounding)."""
    ts = [t.contiguous() for t in tensors]
    assert 1 <= len(ts) <= 16 and all(t.shape == ts[0].shape and t.dtype == ts[0].dtype for t in ts)
    if len(ts) == 1:
        return ts[0]
    out = torch.empty_like(ts[0])
    arr = (ctypes.c_void_p * len(ts))(*[dptr(t) for t in ts])
    lib().call("ofa_add_n", ctypes.addressof(arr), len(ts), ptr(out), out.numel(), dtype_code(out), stream())
    return out


def mul_rowvec(a, vec):
    a, vec = a.contiguous(), vec.contiguous()
    rows, cols = _rows_cols(a)
    y = torch.empty_like(a)
    lib().call("ofa_mul", ptr(a), ptr(vec), ptr(y), rows, cols, 1, dtype_code(a), stream())
    return y


def scale_row_groups(x, scale, group):
    """y[r, :] = x[r, :] * scale[r // group]; x: [rows, cols] rows, scale: fp32 [rows // group]."""
    x = x.contiguous()
    rows, cols = _rows_cols(x)
    assert scale.dtype == torch.float32 and scale.numel() * group == rows
    y = torch.empty_like(x)
    lib().call("ofa_scale_row_groups", ptr(x), ptr(scale.contiguous()), ptr(y), rows, cols, group, dtype_code(x), stream())
    return y


def sum_f32(x):
    x = x.contiguous()
    out = torch.empty((), dtype=torch.float32, device=x.device)
    lib().call("ofa_reduce_sum_f32", ptr(x), ptr(out), x.numel(), stream())
    return out


def bias_block_add_(bias, values, start):
    """In place: bias[B,A,T,T][:, :, s:s+n, s:s+n] += values[n,n,A] (broadcast over batch)."""
    B, A, T, _ = bias.shape
    n = values.shape[0]
    values = values.contiguous()
    assert bias.is_contiguous() and values.dtype == bias.dtype
    lib().call("ofa_bias_block_add", ptr(bias), ptr(values), B, A, T, start, n, dtype_code(bias), stream())
    return bias


def bias_block_add_batch_(bias, values, start):
    """In place: bias[B,A,T,T][:, :, s:s+n, s:s+n] += values[B,A,n,n] (one matrix per sample and head)."""
    B, A, T, _ = bias.shape
    n = values.shape[-1]
    values = values.contiguous()
    assert bias.is_contiguous() and values.dtype == bias.dtype and tuple(values.shape) == (B, A, n, n)
    lib().call("ofa_bias_block_add_batch", ptr(bias), ptr(values), B, A, T, start, n, dtype_code(bias), stream())
    return bias


def bias_block_slice(dbias, start, n):
    """dbias[B,A,T,T][:, :, s:s+n, s:s+n] as a contiguous [B,A,n,n] tensor (the gradient of bias_block_add_batch_'s values)."""
    dbias = dbias.contiguous()
    B, A, T, _ = dbias.shape
    out = torch.empty(B, A, n, n, dtype=dbias.dtype, device=dbias.device)
    lib().call("ofa_bias_block_slice", ptr(dbias), ptr(out), B, A, T, start, n, dtype_code(dbias), stream())
    return out


def bias_outer_grad(dbias, start, F, P):
    """dbias [1|.., A, T, T] (one matrix) -> (d_frames [F,F,A], d_patches [P,P,A]) of an OUTER slot at `start` (ofa_bias_outer_grad)."""
    dbias = dbias.contiguous()
    A, T = dbias.shape[-3], dbias.shape[-1]
    assert dbias.numel() == A * T * T
    dvf = torch.empty(F, F, A, dtype=dbias.dtype, device=dbias.device)
    dvi = torch.empty(P, P, A, dtype=dbias.dtype, device=dbias.device)
    lib().call("ofa_bias_outer_grad", ptr(dbias), ptr(dvf), ptr(dvi), A, T, start, F, P, dtype_code(dbias), stream())
    return dvf, dvi


def bias_block_grad(dbias, start, n):
    dbias = dbias.contiguous()
    B, A, T, _ = dbias.shape
    dvalues = torch.empty(n, n, A, dtype=dbias.dtype, device=dbias.device)
    lib().call("ofa_bias_block_grad", ptr(dbias), ptr(dvalues), B, A, T, start, n, dtype_code(dbias), stream())
    return dvalues


def mean_heads(p, B, heads):
    """p: [B*heads, T, S] -> [B, T, S] mean over heads."""
    p = p.contiguous()
    T, S = p.shape[1], p.shape[2]
    out = torch.empty(B, T, S, dtype=p.dtype, device=p.device)
    lib().call("ofa_mean_heads", ptr(p), ptr(out), B, heads, T * S, dtype_code(p), stream())
    return out


def head_sum(x, B, heads, T):
    """x: fp32 [B*heads, ld] -> [heads] sums over batch and the first T columns."""
    out = torch.empty(heads, dtype=torch.float32, device=x.device)
    lib().call("ofa_head_sum_f32", ptr(x), ptr(out), B, heads, T, x.stride(0), stream())
    return out


# ------------------------------------------------------------------ convolution stack (NHWC rows), csrc/conv.hip
def conv_out_size(n, k, stride, pad):
    return (n + 2 * pad - k) // stride + 1


def im2col(x, B, H, W, C, kh, kw, stride, pad, nchw=False):
    """x: NHWC rows [B*H*W, C] (or the [B,C,H,W] image when nchw) -> col [B*Ho*Wo, Kpad], taps ordered (kh, kw, c)."""
    x = x.contiguous()
    Ho, Wo = conv_out_size(H, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    K = kh * kw * C
    Kpad = (K + 7) // 8 * 8
    col = torch.empty(B * Ho * Wo, Kpad, dtype=x.dtype, device=x.device)
    lib().call("ofa_im2col", ptr(x), ptr(col), B, H, W, C, kh, kw, stride, pad, Kpad, int(nchw), dtype_code(x), stream())
    return col, Ho, Wo


def col2im(dcol, B, H, W, C, kh, kw, stride, pad):
    dcol = dcol.contiguous()
    dx = torch.empty(B * H * W, C, dtype=dcol.dtype, device=dcol.device)
    lib().call("ofa_col2im", ptr(dcol), ptr(dx), B, H, W, C, kh, kw, stride, pad, dcol.shape[1], dtype_code(dcol), stream())
    return dx


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, training, momentum, eps, relu=False, residual=None):
    """x: [rows, C].  Returns y, mean, rstd (fp32 [C]); updates the running buffers in training mode."""
    x = x.contiguous()
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(lib().cdll.ofa_batchnorm_ws_floats(C) * 4, x.device, "bn")
    if residual is not None:
        residual = residual.contiguous()
    lib().call("ofa_batchnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(y), ptr(mean), ptr(rstd),
               ptr(running_mean), ptr(running_var), ptr(ws), rows, C, float(eps), float(momentum), int(not training),
               int(relu), dtype_code(x), stream())
    return y, mean, rstd


def batchnorm_bwd(dy, y, x, gamma, mean, rstd, batch_stats, relu, want_dres=False, dgamma=None, dbeta=None, beta=None):
    """beta (relu layers without a residual only): the ReLU gate is recomputed from x instead of read from y."""
    dy = dy.contiguous()
    rows, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    acc = dgamma is not None
    if not acc:
        dgamma = torch.empty(C, dtype=gamma.dtype, device=x.device)
        dbeta = torch.empty(C, dtype=gamma.dtype, device=x.device)
    ws = workspace(lib().cdll.ofa_batchnorm_ws_floats(C) * 4, x.device, "bn")
    lib().call("ofa_batchnorm_bwd", ptr(dy), ptr(y), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dres), ptr(dgamma),
               ptr(dbeta), ptr(ws), rows, C, int(batch_stats), int(relu), int(acc), ptr(beta) if (relu and not want_dres) else None,
               dtype_code(x), stream())
    return dx, dres, dgamma, dbeta


def batchnorm_fwd_stats(x):
    """SyncBatchNorm, forward phase 1: this rank's [2*C + 1] fp64 sums (sum x, sum x^2, row count) -- to be all-reduced."""
    x = x.contiguous()
    rows, C = x.shape
    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=x.device)
    ws = workspace(lib().cdll.ofa_batchnorm_ws_floats(C) * 4, x.device, "bn")
    lib().call("ofa_batchnorm_fwd_stats", ptr(x), ptr(sums), ptr(ws), rows, C, dtype_code(x), stream())
    return sums


def batchnorm_fwd_apply(x, gamma, beta, running_mean, running_var, sums, momentum, eps, relu=False, residual=None, groups=0):
    """Forward from statistics that exist already; returns y, mean, rstd.  groups = 0: SyncBatchNorm phase 2 (`sums` = the
    all-reduced [2C + 1] sums); groups > 0: `sums` = [groups, 2, C] partial rows over x's rows (a convolution's GEMM epilogue)."""
    x = x.contiguous()
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    if residual is not None:
        residual = residual.contiguous()
    lib().call("ofa_batchnorm_fwd_apply", ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(y), ptr(mean), ptr(rstd),
               ptr(running_mean), ptr(running_var), ptr(sums), int(groups), rows, C, float(eps), float(momentum), int(relu),
               dtype_code(x), stream())
    return y, mean, rstd


def batchnorm_bwd_stats(dy, y, x, gamma, mean, rstd, relu, regate_beta=None, dgamma=None, dbeta=None):
    """SyncBatchNorm, backward phase 1: this rank's [2, C] fp32 sums (sum g, sum g*xhat) -- to be all-reduced -- and the rank-local
    parameter gradients (accumulated into dgamma / dbeta when given)."""
    dy = dy.contiguous()
    rows, C = x.shape
    acc = dgamma is not None
    if not acc:
        dgamma = torch.empty(C, dtype=gamma.dtype, device=x.device)
        dbeta = torch.empty(C, dtype=gamma.dtype, device=x.device)
    sums = torch.empty(2, C, dtype=torch.float32, device=x.device)
    ws = workspace(lib().cdll.ofa_batchnorm_ws_floats(C) * 4, x.device, "bn")
    lib().call("ofa_batchnorm_bwd_stats", ptr(dy), ptr(y), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(dgamma), ptr(dbeta),
               ptr(ws), rows, C, int(relu), int(acc), ptr(regate_beta), dtype_code(x), stream())
    return sums, dgamma, dbeta


def batchnorm_bwd_dx(dy, y, x, gamma, mean, rstd, sums, total_rows, relu, want_dres=False, regate_beta=None):
    """SyncBatchNorm, backward phase 2 (sums all-reduced; total_rows: 1-element fp64 device tensor)."""
    dy = dy.contiguous()
    rows, C = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    assert total_rows.dtype == torch.float64 and total_rows.numel() == 1
    lib().call("ofa_batchnorm_bwd_dx", ptr(dy), ptr(y), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(sums), ptr(dx), ptr(dres), rows,
               ptr(total_rows), C, int(relu), ptr(regate_beta), dtype_code(x), stream())
    return dx, dres


def maxpool_fwd(x, B, H, W, C, k, stride, pad):
    x = x.contiguous()
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    y = torch.empty(B * Ho * Wo, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(B * Ho * Wo, C, dtype=torch.uint8, device=x.device)
    lib().call("ofa_maxpool_fwd", ptr(x), ptr(y), ptr(arg), B, H, W, C, k, stride, pad, dtype_code(x), stream())
    return y, arg, Ho, Wo


def maxpool_bwd(dy, arg, B, H, W, C, k, stride, pad):
    dy = dy.contiguous()
    dx = torch.empty(B * H * W, C, dtype=dy.dtype, device=dy.device)
    lib().call("ofa_maxpool_bwd", ptr(dy), ptr(arg), ptr(dx), B, H, W, C, k, stride, pad, dtype_code(dy), stream())
    return dx


def relu(x, gate=None):
    x = x.contiguous()
    y = torch.empty_like(x)
    lib().call("ofa_relu", ptr(x), ptr(gate), ptr(y), x.numel(), dtype_code(x), stream())
    return y


# ------------------------------------------------------------------ residual join (csrc/join.hip)
def join_fwd(x, residual, ln_a, ln_b, eps, p, seed, offset, offset_base):
    """y = residual + dropout_p(LN_a(x)); z = LN_b(y).  ln_a / ln_b: (gamma, beta) or None.  Returns y, z (or None), stats, keep_bits
    (the dropout decisions for join_bwd, or None: this shape regenerates them)."""
    x = x.contiguous()
    residual = residual.contiguous() if residual is not None else None       # None: y = dropout_p(LN_a(x))
    rows, cols = _rows_cols(x)
    y = torch.empty_like(x)
    z = torch.empty_like(x) if ln_b is not None else None
    stats = torch.empty(4, rows, dtype=torch.float32, device=x.device)
    nkeep = lib().cdll.ofa_join_keep_bytes(rows, cols, dtype_code(x)) if p > 0 else 0
    keep = torch.empty(nkeep, dtype=torch.uint8, device=x.device) if nkeep else None
    ga, ba = ln_a if ln_a is not None else (None, None)
    gb, bb = ln_b if ln_b is not None else (None, None)
    lib().call("ofa_join_fwd", ptr(x), ptr(residual), ptr(ga), ptr(ba), ptr(gb), ptr(bb), ptr(y), ptr(z), ptr(stats), ptr(keep), rows, cols,
               float(eps), float(p), seed, offset, ptr(offset_base), dtype_code(x), stream())
    return y, z, stats, keep


def join_bwd(dy, dz, x, y, ga, gb, stats, p, seed, offset, offset_base, grads, fold=None, want_dres=True, keep=None):
    """grads = (dgamma_a, dbeta_a, dgamma_b, dbeta_b[, dx_colsum]) output tensors (accumulated into; None for an absent
    LayerNorm; dx_colsum: the bias gradient of the Linear that produced x, optional).
    Returns dres, dx.  The column partials are folded by `fold` (a FoldQueue) or immediately."""
    like = dy if dy is not None else dz
    rows, cols = _rows_cols(like)
    dres, dx = (torch.empty_like(like) if want_dres else None), torch.empty_like(like)
    ns = lib().cdll.ofa_join_bwd_slots(rows, cols, dtype_code(like))
    ws = torch.empty(5 * ns * cols, dtype=torch.float32, device=like.device)
    q = fold if fold is not None else FoldQueue()
    for i, o in enumerate(grads):
        if o is not None:
            q.add(ws, i * ns * cols, o, cols, cols, ns)
    want_xsum = len(grads) > 4 and grads[4] is not None
    lib().call("ofa_join_bwd", ptr(dy.contiguous() if dy is not None else None), ptr(dz.contiguous() if dz is not None else None),
               ptr(x), ptr(y), ptr(ga), ptr(gb), ptr(stats), ptr(keep), ptr(dres), ptr(dx), ptr(ws), rows, cols, float(p), seed, offset,
               ptr(offset_base), int(want_xsum), dtype_code(like), stream())
    if fold is None:
        q.flush()
    else:
        fold.flush_if_large()
    return dres, dx
