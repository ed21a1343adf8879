import sys, itertools, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
torch.manual_seed(0)
bad = 0; n = 0
for kind, M, N, Kk in itertools.product(['NT', 'NN', 'TN'], [24, 64, 96, 128, 256, 384, 1536, 6144], [64, 128, 256, 576, 1024], [256, 320, 512, 576, 640, 960, 1152]):
    ta, tb = {'NT': (False, True), 'NN': (False, False), 'TN': (True, False)}[kind]
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    for acc in (False, True):
        out = torch.randn(M, N, device=dev).bfloat16() if acc else None
        base = out.float().clone() if acc else 0
        try:
            y = K.gemm(a, b, ta, tb, out=out, accumulate=acc)
        except Exception as e:
            print("ERR", kind, M, N, Kk, acc, str(e)[:80]); bad += 1; continue
        A = (a.t() if ta else a).float(); Bm = (b.t() if tb else b).float()
        ref = A @ Bm + base
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        n += 1
        if err > 1.5e-2:
            bad += 1
            print(f"BAD {kind} M={M} N={N} K={Kk} acc={acc} err={err:.3e}")
print("checked", n, "bad", bad)
