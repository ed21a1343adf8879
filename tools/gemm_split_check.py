"""Split-K (+ reduce launch) vs an unsplit launch of smaller tiles on the ResNet trunk's skinny products (cfg-2b / cfg-4:
B = 32 images of 384^2 -> layer2 rows 73728, layer3 rows 18432; conv as [rows, K] x [N, K]^T).  Run twice:
    python tools/gemm_split_check.py                       # the planner's choice (split when K >= 1024 and tiles < 384)
    OFA_GEMM_SPLIT_MIN_K=1000000 python tools/gemm_split_check.py    # never split
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
rows3, rows2 = 32 * 24 * 24, 32 * 48 * 48
shapes = [('NT', rows3, 256, 1024), ('NT', rows3, 256, 2304), ('NT', rows3, 1024, 256), ('NN', rows3, 256, 1024), ('NN', rows3, 2304, 256),
          ('NN', rows3, 1024, 256), ('NT', rows2, 128, 512), ('NT', rows2, 128, 1152), ('NT', rows2, 512, 128), ('NN', rows2, 128, 512),
          ('NN', rows2, 1152, 128), ('NT', 6272, 256, 1024), ('NT', 6272, 256, 2304), ('NN', 6272, 256, 1024)]
print("# OFA_GEMM_SPLIT_MIN_K =", os.environ.get("OFA_GEMM_SPLIT_MIN_K", "(default 1024)"))
for kind, M, N, Kk in shapes:
    ta, tb = {'NT': (False, True), 'NN': (False, False)}[kind]
    a = torch.randn(M, Kk, device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    sp = K.lib().cdll.ofa_gemm_splits(M, N, Kk, int(ta), int(tb), 1, 0, K.dtype_code(a), 1 << 30)
    t = bench(lambda: K.gemm(a, b, ta, tb, out=out))
    fl = 2.0 * M * N * Kk
    print(f"{kind} M={M:6d} N={N:5d} K={Kk:5d} splits={sp}  {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF")
