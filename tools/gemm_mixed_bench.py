"""Two tile heights in one big-tile launch (gemm_big_mixed_kernel, csrc/gemm_mfma.hip) against the plain plan: per-launch time inside
replayed graphs, arms interleaved (OFA_GEMM_MIXED = 0 never / 1 wherever eligible / -1 the planner's model decides; DEBUG library:
OFASYS_AMD_LIB=ofasys_amd/libofasys_amd_dbg.so python tools/gemm_mixed_bench.py), and bit-equality of the three results."""
import os, sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
shapes = [('NT', 13312, 3072, 768, 'fc1 fwd'), ('NT', 13312, 2304, 768, 'qkv fwd'), ('NT', 13312, 9216, 768, 'cross kv'), ('NT', 13312, 768, 768, 'out fwd'),
          ('NT', 13312, 768, 3072, 'fc2 fwd'), ('NN', 13312, 3072, 768, 'fc2 dgrad'), ('NN', 13312, 768, 768, 'out dgrad'), ('NT', 12800, 3072, 768, 'fc1 fwd M=12800'),
          ('NT', 13000, 3072, 768, 'fc1 fwd M=13000'), ('NT', 1536, 51272, 768, 'vocab proj'), ('NT', 8192, 8192, 8192, 'square')]
N_CALLS = 20
def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N_CALLS): fn()
    for _ in range(2): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / N_CALLS * 1e3)
    return sorted(ts)[2]
for kind, M, N, Kk, what in shapes:
    tb = kind == 'NT'
    a = torch.randn((M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    outs, graphs = {}, {}
    def build(arm):
        os.environ["OFA_GEMM_MIXED"] = arm
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda: K.gemm(a, b, False, tb, bias=bias, out=out)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            fn(); fn()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(N_CALLS): fn()
        return g, out
    for arm in ("0", "1", "-1"):
        graphs[arm], outs[arm] = build(arm)
    ts = {arm: [] for arm in graphs}
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for rnd in range(7):
        for arm, g in graphs.items():
            g.replay(); torch.cuda.synchronize()
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts[arm].append(e0.elapsed_time(e1) / N_CALLS * 1e3)
    eq = torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["-1"])
    print(f"{kind} {M:6d} x {N:5d} x K {Kk:5d}  {what:18s} " + " | ".join(f"{arm}: {sorted(v)[3]:6.1f}" for arm, v in ts.items()) + f" | equal {eq}", flush=True)
