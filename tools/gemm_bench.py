"""Per-shape timing of the MFMA GEMM entry point (ofa_gemm, as the step calls it) on the products of the cfg-2 step, with
torch.matmul (hipBLASLt / rocBLAS) on the same operands as a YARD-STICK only -- the product never calls it.  Back-to-back launches of
one shape, HIP events on the launch stream; the grouped weight-gradient launch (ofa_gemm_group_tn) is timed on one encoder layer's
four products against the same four products through torch.matmul.
  python tools/gemm_bench.py   ->  gpurun_out/profiles/round<ROUND>_gemm_microbench.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"
PEAK = 2500.0   # dense bf16 MFMA, TFLOP/s (MI355X_MICROARCH.md)


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


R, Rd = 13312, 1536          # encoder / decoder rows of the packed cfg-2 batch (B = 32; row buckets of 512 / 256)
shapes = [
    # forward (y = x W^T): rows x N x K
    ("NT", R, 2304, 768, "qkv forward"), ("NT", R, 768, 768, "out_proj forward"), ("NT", R, 3072, 768, "fc1 forward"),
    ("NT", R, 768, 3072, "fc2 forward"), ("NT", R, 9216, 768, "cross k|v of 6 layers"),
    ("NT", Rd, 2304, 768, "decoder qkv"), ("NT", Rd, 3072, 768, "decoder fc1"), ("NT", Rd, 768, 3072, "decoder fc2"),
    ("NT", Rd, 51272, 768, "output projection"),
    # input gradients (dx = dy W)
    ("NN", R, 768, 2304, "qkv dgrad"), ("NN", R, 768, 768, "out_proj dgrad"), ("NN", R, 768, 3072, "fc1 dgrad"),
    ("NN", R, 3072, 768, "fc2 dgrad"), ("NN", Rd, 768, 51272, "output projection dgrad"),
    # single weight gradients (dW = dy^T x) that do not ride in a group
    ("TN", 51272, 768, Rd, "embedding / output weight gradient"),
    ("NT", 8192, 8192, 8192, "large square"),
]
lines = []
for kind, M, N, Kk, what in shapes:
    ta, tb = {"NT": (False, True), "NN": (False, False), "TN": (True, False)}[kind]
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kpad = False
    if kind == "NN" and Kk % 64:
        # the vocabulary projection's input gradient as the step runs it (ops.LinearFn): the logit gradient lives in rows padded to a
        # multiple of 64 elements with a zero tail, so the contraction runs on the LDS-DMA loop over the padded K (V = 51265 -> 51328)
        store = torch.zeros(M, (Kk + 63) // 64 * 64, device=dev, dtype=torch.bfloat16)
        store[:, :Kk] = a
        a, kpad = store[:, :Kk], True
    t = bench(lambda: K.gemm(a, b, ta, tb, out=out, a_kpad_zero=kpad))
    A = a.t() if ta else a.contiguous()
    Bm = b.t() if tb else b
    o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tt = bench(lambda: torch.matmul(A, Bm, out=o2))
    fl = 2.0 * M * N * Kk
    lines.append(f"{kind} {M:6d} x {N:6d} x K {Kk:6d}  {what:36s} ofa {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TFLOP/s ({fl / t / 1e9 / PEAK * 100:4.1f} %)"
                 f" | torch.matmul {tt * 1e3:8.1f} us {fl / tt / 1e9:7.1f} TFLOP/s")
    print(lines[-1], flush=True)

# one encoder layer's four weight gradients, grouped (the step's form) against four library calls
dys = [torch.randn(R, n, device=dev).bfloat16() for n in (2304, 768, 3072, 768)]
xs = [torch.randn(R, k, device=dev).bfloat16() for k in (768, 768, 768, 3072)]
outs = [torch.zeros(dy.shape[1], x.shape[1], device=dev, dtype=torch.bfloat16) for dy, x in zip(dys, xs)]


def grouped():
    f = K.FoldQueue()
    K.gemm_group_tn([(dy, x, o, 1.0) for dy, x, o in zip(dys, xs, outs)], f)
    f.flush()


try:
    tg = bench(grouped)
    tl = bench(lambda: [torch.matmul(dy.t(), x) for dy, x in zip(dys, xs)])
    fl = sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x in zip(dys, xs))
    lines.append(f"grouped weight gradients of one encoder layer (4 products, K = {R}): ofa {tg * 1e3:.1f} us {fl / tg / 1e9:.1f} TFLOP/s "
                 f"({fl / tg / 1e9 / PEAK * 100:.1f} %, slab folds included) | 4 x torch.matmul {tl * 1e3:.1f} us {fl / tl / 1e9:.1f} TFLOP/s")
    print(lines[-1], flush=True)
except Exception as e:                                     # (the yard-stick must not take the per-shape table with it)
    lines.append(f"grouped weight gradients: not timed ({type(e).__name__}: {e})")
    print(lines[-1])

rnd = os.environ.get("ROUND", "6")
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profiles")
os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, f"round{rnd}_gemm_microbench.txt"), "w") as f:
    f.write("# tools/gemm_bench.py: ofa_gemm on the cfg-2 step's products, back-to-back launches, HIP events; torch.matmul (hipBLASLt) on the "
            "same operands as a yard-stick\n# (% = of the 2.5 PFLOP/s dense bf16 MFMA peak)\n" + "\n".join(lines) + "\n")
