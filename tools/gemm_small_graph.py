"""Decoder-side products (M = 1536 rows) inside a replayed graph: per-launch time of the planner's choice and of forced tiles
(OFA_GEMM_TILE = 22 / 12 / 11, DEBUG library) -- an eager loop is host-bound at ~14 us per call and cannot see these kernels."""
import os, sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
shapes = [('NT', 1536, 2304, 768, 'dec qkv'), ('NT', 1536, 768, 768, 'dec out/q'), ('NT', 1536, 3072, 768, 'dec fc1'), ('NT', 1536, 768, 3072, 'dec fc2'),
          ('NN', 1536, 768, 2304, 'dec qkv dgrad'), ('NN', 1536, 768, 768, 'dec out dgrad'), ('NN', 1536, 768, 3072, 'dec fc1 dgrad'), ('NN', 1536, 3072, 768, 'dec fc2 dgrad')]
N_CALLS = 40
def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N_CALLS): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(7):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / N_CALLS * 1e3)
    return sorted(ts)[3]
arms = [a for a in os.environ.get("ARMS", "0,22,12,11").split(",")]
for kind, M, N, Kk, what in shapes:
    ta, tb = {'NT': (False, True), 'NN': (False, False)}[kind]
    a = torch.randn((M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16() if kind == 'NT' else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for arm in arms:
        if arm == "0": os.environ.pop("OFA_GEMM_TILE", None)
        else: os.environ["OFA_GEMM_TILE"] = arm
        try:
            t = graph_time(lambda: K.gemm(a, b, ta, tb, bias=bias, out=out))
            res.append(f"{arm}: {t:5.1f}")
        except Exception as e:
            res.append(f"{arm}: err {type(e).__name__}")
    os.environ.pop("OFA_GEMM_TILE", None)
    fl = 2.0 * M * N * Kk
    print(f"{kind} {M} x {N:5d} x K {Kk:5d}  {what:16s} " + " | ".join(res), flush=True)
