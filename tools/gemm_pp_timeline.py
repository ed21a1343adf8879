"""Per-segment shader clocks of the ping-pong GEMM loop (csrc/gemm_pp.hip built with -DOFA_PP_TIMELINE: make -C ofasys_amd/csrc timeline).
Waves 0 (group 0) and 4 (group 1) of every workgroup sum, over the phases of their K loop:
   load   load-segment start -> last issue (fragment reads + this wave's LDS-DMA pieces + counted vmcnt)
   bar1   -> barrier passed
   frag   -> fragments landed (lgkmcnt(0))
   mfma   -> MFMA segment issued (+ the counted vmcnt of a tile's last phase)
   bar2   -> next load-segment start (barrier passed)
and the kernel's total clocks beside the 100 MHz real-time counter (= the shader clock the loop actually ran at).
  python tools/gemm_pp_timeline.py [variant]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OFASYS_AMD_LIB"] = os.path.join(ROOT, "ofasys_amd", "libofasys_amd_tl.so")
from ofasys_amd import kernels as K  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "21"
dev = "cuda"
for (M, N, Kk) in [(8192, 8192, 8192), (13312, 2304, 768), (13312, 768, 3072)]:
    a = torch.randn(M, Kk, device=dev).bfloat16()
    b = torch.randn(N, Kk, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    os.environ["OFA_GEMM_TILE"] = "84"
    os.environ["OFA_GEMM_PP"] = variant
    for _ in range(3):
        K.gemm(a, b, False, True, out=o)
    torch.cuda.synchronize()
    ws = K.workspace(256 << 20, a.device, "gemm")
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rec = ws.view(torch.int64)[: tiles * 2 * 16].view(tiles, 2, 16).cpu().double()
    print(f"NT {M}x{N}x{Kk}  variant {variant}: {tiles} workgroups, {Kk // 64} K-tiles, 2 phases per K-tile")
    for g in (0, 1):
        r = rec[:, g]
        n = r[:, 5].clamp_min(1)
        seg = [float((r[:, i] / n).mean()) for i in range(5)]
        tot, real = float(r[:, 6].mean()), float(r[:, 7].mean())
        print(f"   group {g}: load {seg[0]:6.0f}  bar1 {seg[1]:6.0f}  frag {seg[2]:6.0f}  mfma {seg[3]:6.0f}  bar2 {seg[4]:6.0f}   = {sum(seg):6.0f} clocks per phase"
              f" | kernel {tot:9.0f} clocks in {real * 10:8.0f} ns -> {tot / (real * 10):.2f} GHz")
os.environ["OFA_GEMM_PP"] = "0"
os.environ["OFA_GEMM_TILE"] = "0"
