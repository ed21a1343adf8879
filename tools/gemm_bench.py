import sys, torch, time
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [
 ('NT', 14336, 768, 768), ('NT', 14336, 2304, 768), ('NT', 14336, 3072, 768), ('NT', 14336, 768, 3072),
 ('NT', 2048, 768, 768), ('NT', 2048, 3072, 768), ('NT', 2048, 768, 3072), ('NT', 2048, 51264, 768),
 ('NN', 14336, 768, 768), ('NN', 14336, 768, 3072), ('NN', 14336, 3072, 768), ('NN', 2048, 768, 768), ('NN', 2048, 768, 51264),
 ('TN', 768, 768, 14336), ('TN', 3072, 768, 14336), ('TN', 768, 3072, 14336), ('TN', 768, 768, 2048), ('TN', 51264, 768, 2048),
 ('NT', 8192, 8192, 8192), ('NT', 4096, 4096, 4096),
]
for kind, M, N, Kk in shapes:
    ta, tb = {'NT': (False, True), 'NN': (False, False), 'TN': (True, False)}[kind]
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(a, b, ta, tb, out=out))
    A = a.t() if ta else a; Bm = b.t() if tb else b
    tt = bench(lambda: torch.matmul(A, Bm))
    fl = 2.0 * M * N * Kk
    print(f"{kind} M={M:6d} N={N:6d} K={Kk:6d}  ofa {t*1e3:8.1f} us {fl/t/1e9:7.1f} TF | hipblaslt {tt*1e3:8.1f} us {fl/tt/1e9:7.1f} TF")
