"""Tile choice on the ResNet trunk's convolution products (forward NT: [rows, K] x [N, K]^T; input gradient NN: [rows, N] x [N, K]):
    for t in 0 22 12 11; do OFA_GEMM_TILE=$t python tools/gemm_conv_sweep.py [cfg2b|cfg4]; done
prints one line per shape: the planner's choice (OFA_GEMM_TILE unset / 0) or the forced 128x128 / 64x128 / 64x64 tile.
(The OFA_GEMM_* planner overrides exist in the DEBUG library only: make -C ofasys_amd/csrc debug, then run with
OFASYS_AMD_LIB=ofasys_amd/libofasys_amd_dbg.so; the shipped library ignores them.)"""
import os, sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
w = sys.argv[1] if len(sys.argv) > 1 else "cfg2b"
r1, r2, r3 = ((32 * 96 * 96, 32 * 48 * 48, 32 * 24 * 24) if w == "cfg2b" else (32 * 56 * 56, 32 * 28 * 28, 32 * 14 * 14))
convs = [(r1, 64, 64), (r1, 576, 64), (r1, 64, 256), (r1, 256, 64), (r2, 256, 128), (r2, 1152, 128), (r2, 128, 512), (r2, 512, 128),
         (r3, 512, 256), (r3, 2304, 256), (r3, 256, 1024), (r3, 1024, 256)]
print("# OFA_GEMM_TILE =", os.environ.get("OFA_GEMM_TILE", "(planner)"), w)
tot = 0.0
for rows, Kk, N in convs:
    for kind in ("NT", "NN"):
        if kind == "NT":
            a = torch.randn(rows, Kk, device=dev).bfloat16(); b = torch.randn(N, Kk, device=dev).bfloat16()
            out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
            t = bench(lambda: K.gemm(a, b, False, True, out=out)); M_, N_, K_ = rows, N, Kk
        else:                                   # dX = dY [rows, N] @ W [N, K]
            a = torch.randn(rows, N, device=dev).bfloat16(); b = torch.randn(N, Kk, device=dev).bfloat16()
            out = torch.empty(rows, Kk, device=dev, dtype=torch.bfloat16)
            t = bench(lambda: K.gemm(a, b, False, False, out=out)); M_, N_, K_ = rows, Kk, N
        tot += t
        print(f"{kind} M={M_:7d} N={N_:5d} K={K_:5d}  {t*1e3:8.1f} us {2.0*M_*N_*K_/t/1e9:7.1f} TF")
print(f"sum {tot*1e3:.1f} us")
