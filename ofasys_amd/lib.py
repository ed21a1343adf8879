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


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OfaError("ofasys_amd: tensor is not on a GPU -- the HIP path is the only path (no CPU fallback)")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
