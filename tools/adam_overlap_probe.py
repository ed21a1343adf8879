"""Can the optimizer pass (ofa_adam_step: 28 B per parameter of pure HBM streaming, no MFMA) hide under MFMA-bound GEMMs?  
Three hipGraphs on the cfg-2 sizes: (a) the four forward GEMMs of six encoder layers back to back, (b) Adam over the 141.7 M-parameter arena,
(c) both, as two parallel branches of ONE graph (Adam on a side stream, forked before the first GEMM and joined after the last) -- the form a
deferred update under the next step's forward would take.  Also (d): Adam cut into 24 slices, one launched behind every GEMM's predecessor
(so that a slice and a GEMM always start together).  Prints the replay times; overlap = a + b - c.
  python tools/adam_overlap_probe.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
N = 141_680_064
R = 13312
master = torch.randn(N, device=dev) * 0.02
m, v = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
grad = (torch.randn(N, device=dev) * 1e-3).bfloat16()
param = master.bfloat16()
coef = torch.tensor([1.0, 1e-4, 1e-4, 0.0], device=dev)
x768 = torch.randn(R, 768, device=dev).bfloat16()
x3072 = torch.randn(R, 3072, device=dev).bfloat16()
Ws = [torch.randn(n, k, device=dev).bfloat16() * 0.02 for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072))]
outs = [torch.empty(R, n, device=dev, dtype=torch.bfloat16) for n in (2304, 768, 3072, 768)]


def gemms():
    for _ in range(6):
        K.gemm(x768, Ws[0], False, True, out=outs[0])
        K.gemm(x768, Ws[1], False, True, out=outs[1])
        K.gemm(x768, Ws[2], False, True, out=outs[2])
        K.gemm(x3072, Ws[3], False, True, out=outs[3])


def adam(lo=0, hi=N):
    K.adam_step(master[lo:hi], m[lo:hi], v[lo:hi], grad[lo:hi], param[lo:hi], coef, 0.0, 0.9, 0.999, 1e-8, 0.0, 0)


def both_parallel():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        adam()
    gemms()
    torch.cuda.current_stream().wait_stream(side)


def both_sliced():
    side = torch.cuda.Stream()
    cuts = [int(N * i / 24) // 8 * 8 for i in range(25)]
    i = 0
    for _ in range(6):
        for w, xin, o in ((Ws[0], x768, outs[0]), (Ws[1], x768, outs[1]), (Ws[2], x768, outs[2]), (Ws[3], x3072, outs[3])):
            side.wait_stream(torch.cuda.current_stream())          # the slice starts when this GEMM's predecessor is done
            with torch.cuda.stream(side):
                adam(cuts[i], cuts[i + 1])
            K.gemm(xin, w, False, True, out=o)
            i += 1
    torch.cuda.current_stream().wait_stream(side)


def capture(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def timed(g, n=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


ga, gb, gc, gd = capture(gemms), capture(adam), capture(both_parallel), capture(both_sliced)
ta, tb, tc, td = timed(ga), timed(gb), timed(gc), timed(gd)
print(f"(a) 24 forward GEMMs alone          {ta:8.1f} us")
print(f"(b) Adam over {N / 1e6:.1f} M parameters alone {tb:8.1f} us")
print(f"(c) both, parallel graph branches    {tc:8.1f} us   (a + b = {ta + tb:.1f}: {ta + tb - tc:+.1f} us hidden, {(ta + tb - tc) / tb * 100:.0f} % of Adam)")
print(f"(d) both, Adam in 24 slices beside the GEMMs {td:8.1f} us   ({ta + tb - td:+.1f} us hidden, {(ta + tb - td) / tb * 100:.0f} % of Adam)")
