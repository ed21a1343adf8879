"""ctypes binding of libofasys_amd.so (the C ABI declared in include/ofasys_amd.h).

The prototypes are parsed from the header itself, so Python can never drift from the C declaration.
There is NO CPU fallback: if the shared library is missing, or a call is made on non-GPU tensors, this
module raises -- the product path is the HIP path or nothing.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must be imported first: the .so resolves libamdhip64.so.7 to the copy torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "ofasys_amd.h")
LIB_PATH = os.environ.get("OFASYS_AMD_LIB") or os.path.join(_HERE, "libofasys_amd.so")   # override: kernel experiments

F32, BF16, F16 = 0, 1, 2
GEMM_BIAS_COL, GEMM_BIAS_ROW, GEMM_ACCUM, GEMM_FORCE_SIMPLE, GEMM_OUT_F32 = 1, 2, 4, 8, 16

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "void": None,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes])} for every function prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char\s*\*)\s+(ofa_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int)
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.replace("const", " ").split()[0]
                    argtypes.append(_CTYPES[base])
        protos[name] = (restype, argtypes)
    return protos


class OfaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise OfaError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C ofasys_amd/csrc). ofasys_amd has no CPU/eager fallback.")
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (restype, argtypes) in self.protos.items():
            fn = getattr(self.cdll, name)   # AttributeError here == header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes

    def call(self, name, *args):
        if _audit is not None:
            _audit.label(name)
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            msg = self.cdll.ofa_last_error()
            raise OfaError(f"{name} failed (status {rc}): {msg.decode() if msg else ''}")


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib


def dtype_code(t, allow_f16=True):
    """ofa_dtype of a tensor: fp32, bf16 or fp16 (every entry point of the C ABI takes the three)."""
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if allow_f16 and t.dtype == torch.float16:
        return F16
    raise OfaError(f"ofasys_amd kernels take float32, bfloat16 or float16 tensors, got {t.dtype}")


class PointerAudit:
    """Which device memory does a hipGraph being captured address that its private memory pool does NOT own?

    A captured kernel node holds raw addresses.  Memory the capture itself allocates lives in the graph's private pool and stays
    mapped for as long as the graph does; memory allocated BEFORE the capture (parameters, arenas, caches filled by the eager warm-up
    steps, grown scratch buffers) lives in torch's default pool, and nothing but a Python reference keeps it there -- once the last
    reference goes, the block returns to the allocator and the `torch.cuda.empty_cache()` at the entry of the NEXT capture hands it
    back to the driver: the older graph then faults on its next replay.

    While an audit is active, every tensor whose address goes through `ptr()` / `dptr()` and lies in a default-pool segment (as of
    the snapshot taken at the start of the capture) is recorded with the C-ABI call that first used it, and KEPT ALIVE: the
    graph's entry stores `pins`, so a cache that later drops or replaces such a tensor cannot unmap what a graph still reads.
    `foreign(owned)` reports the pins nobody else owns -- the latent dangling references."""

    def __init__(self, segments):
        # segments: [(start, end)] of the default pool, sorted
        self.starts = [a for a, _ in segments]
        self.ends = [b for _, b in segments]
        self.pins = {}            # storage address -> [tensor, first call]
        self._pending = []

    @classmethod
    def from_snapshot(cls, snapshot):
        """From torch.cuda.memory_snapshot(): the segments of the default pool (segment_pool_id (0, 0))."""
        segs = sorted((s["address"], s["address"] + s["total_size"]) for s in snapshot
                      if tuple(s.get("segment_pool_id", (0, 0))) == (0, 0))
        return cls(segs)

    def in_default_pool(self, addr):
        import bisect
        i = bisect.bisect_right(self.starts, addr) - 1
        return i >= 0 and addr < self.ends[i]

    def see(self, t):
        try:
            sp = t.untyped_storage().data_ptr()
        except Exception:                                 # noqa: BLE001  (a tensor subclass without storage: its data_ptr will do)
            sp = t.data_ptr()
        if sp in self.pins or not self.in_default_pool(sp):
            return
        rec = [t, None]
        self.pins[sp] = rec
        self._pending.append(rec)

    def label(self, name):
        for rec in self._pending:
            rec[1] = name
        self._pending = []

    def tensors(self):
        return [rec[0] for rec in self.pins.values()]

    def foreign(self, owned):
        """[(first call, shape, dtype, bytes)] of the pinned tensors whose storage is not among `owned` (an iterable of tensors
        whose lifetime is the step engine's own: parameters, buffers, arenas, optimizer state, static inputs)."""
        have = set()
        for o in owned:
            if o is not None and o.is_cuda:
                have.add(o.untyped_storage().data_ptr())
        out = []
        for sp, (t, name) in self.pins.items():
            if sp not in have:
                out.append((name or "?", tuple(t.shape), str(t.dtype).replace("torch.", ""), t.untyped_storage().nbytes()))
        return out


_audit = None


class capture_audit:
    """Context manager: audit (and pin) the pre-existing device memory a capture addresses.  Usage (trainer.TrainStep._capture):
        with capture_audit() as audit:
            with torch.cuda.graph(g, pool=pool): ...
        entry["pins"] = audit.tensors()
    The allocator cache is flushed BEFORE the snapshot is taken, so that torch.cuda.graph()'s own flush at capture entry releases
    nothing more: an address range the snapshot calls "default pool" cannot come back as private-pool memory during the capture."""

    def __enter__(self):
        global _audit
        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        self.audit = PointerAudit.from_snapshot(torch.cuda.memory_snapshot())
        self._prev, _audit = _audit, self.audit
        return self.audit

    def __exit__(self, *exc):
        global _audit
        _audit = self._prev
        return False


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OfaError("ofasys_amd: tensor is not on a GPU -- the HIP path is the only path (no CPU fallback)")
    if _audit is not None:
        _audit.see(t)
    return t.data_ptr()


def dptr(t):
    """t.data_ptr() for the call sites that put device addresses into host-side tables (fold jobs, grouped GEMM items, bias slots)
    instead of passing them as arguments: the same capture audit sees them."""
    if _audit is not None and t.is_cuda:
        _audit.see(t)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
