"""Timing-only ablations of the ping-pong GEMM loop (csrc/gemm_pp.hip, debug library): variant 21 with pieces of its K-loop removed
(ABL bit 1: no fragment reads, 2: no LDS-DMA, 4: no MFMAs -- results are WRONG by construction, only the times mean anything).
  python tools/gemm_pp_ablate.py"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OFASYS_AMD_LIB", os.path.join(ROOT, "ofasys_amd", "libofasys_amd_dbg.so"))
from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


names = {0: "lockstep", 21: "pp21 full", 121: "no reads", 221: "no DMA", 321: "no reads, no DMA", 421: "no MFMA", 521: "no reads, no MFMA",
         621: "no DMA, no MFMA", 721: "barriers only"}
for (M, N, Kk) in [(8192, 8192, 8192), (13312, 2304, 768), (13312, 3072, 768)]:
    a = torch.randn(M, Kk, device=dev).bfloat16()
    b = torch.randn(N, Kk, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    os.environ["OFA_GEMM_TILE"] = "84"
    ts = {v: [] for v in names}
    for v in names:
        os.environ["OFA_GEMM_PP"] = str(v)
        for _ in range(3):
            K.gemm(a, b, False, True, out=o)
    for _ in range(3):
        for v in names:
            os.environ["OFA_GEMM_PP"] = str(v)
            ts[v].append(timed(lambda: K.gemm(a, b, False, True, out=o), 10))
    fl = 2.0 * M * N * Kk
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = (tiles + 255) // 256
    print(f"NT {M}x{N}x{Kk}: {tiles} tiles = {rounds} round(s), {Kk // 64} K-tiles each")
    for v in names:
        t = statistics.median(ts[v])
        print(f"   {names[v]:22s} {t:8.1f} us  {fl / t / 1e6:6.0f} TF-equivalent   {t / rounds / (Kk // 64) * 1e3:7.0f} ns per K-tile and round")
os.environ["OFA_GEMM_PP"] = "0"
os.environ["OFA_GEMM_TILE"] = "0"
