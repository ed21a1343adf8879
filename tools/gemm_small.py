import sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [('NT', 2048, 768, 768), ('NT', 2048, 2304, 768), ('NT', 2048, 3072, 768), ('NT', 2048, 768, 3072), ('NT', 2048, 1536, 768),
          ('NN', 2048, 768, 768), ('NN', 2048, 768, 2304), ('NN', 2048, 768, 3072), ('NN', 2048, 3072, 768),
          ('TN', 768, 768, 2048), ('TN', 3072, 768, 2048), ('TN', 768, 3072, 2048), ('TN', 2304, 768, 2048)]
for kind, M, N, Kk in shapes:
    ta, tb = {'NT': (False, True), 'NN': (False, False), 'TN': (True, False)}[kind]
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: K.gemm(a, b, ta, tb, out=out))
    print(f"{kind} M={M:6d} N={N:6d} K={Kk:6d}  {t:8.1f} us {2.0*M*N*Kk/t/1e6:7.1f} TF")
