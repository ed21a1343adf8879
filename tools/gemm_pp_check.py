"""Race / correctness screen of the ping-pong GEMM loop (csrc/gemm_pp.hip) against the lockstep loop on the same tile: every product is
computed by both and compared BIT FOR BIT (same MFMA order per accumulator), repeatedly and with the chip busy (a missing DMA wait or a
too-early buffer refill only shows when pieces land late).  Shapes cover K = 1 .. 7 tiles (prologue / tail paths of the counted vmcnt
scheme), ragged M / N edges, all three operand layouts, split-K slabs, batched products, bias / alpha / accumulate, ragged weight-gradient
row counts, the grouped launch.  Needs the DEBUG library (OFASYS_AMD_LIB=ofasys_amd/libofasys_amd_dbg.so, set below by default).
  python tools/gemm_pp_check.py [variant ...]        default variants: 23 (shipped) 21 20 22 11"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OFASYS_AMD_LIB", os.path.join(ROOT, "ofasys_amd", "libofasys_amd_dbg.so"))
from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"
variants = [int(v) for v in sys.argv[1:]] or [23, 21, 20, 22, 11]


def setenv(pp, tile):
    os.environ["OFA_GEMM_PP"] = str(pp)
    os.environ["OFA_GEMM_TILE"] = str(tile)


bad = 0
nchk = 0


def check(tag, fn, tile, reps=3):
    global bad, nchk
    setenv(0, tile)
    ref = fn()
    torch.cuda.synchronize()
    for v in variants:
        setenv(v, tile)
        for r in range(reps):
            out = fn()
            nchk += 1
            same = all(torch.equal(x, y) for x, y in zip(out, ref))
            if not same:
                bad += 1
                d = max(float((x.float() - y.float()).abs().max()) for x, y in zip(out, ref))
                print(f"MISMATCH {tag} variant {v} tile {tile} rep {r}: max abs diff {d:.4g}", flush=True)
                break
    setenv(0, 0)


torch.manual_seed(7)
LAY = {"NT": (False, True), "NN": (False, False), "TN": (True, False)}
# (M, N, K): K tiles 1..7 and long; ragged edges; the step's shapes
small = [(256, 256, 64), (256, 512, 128), (512, 256, 192), (600, 520, 256), (1000, 768, 320), (2050, 1288, 448), (192, 256, 64),
         (3333, 264, 448), (384, 1024, 384)]
big = [(13312, 2304, 768), (13312, 768, 3072), (13312, 768, 768), (4096, 4096, 4096), (1536, 51272, 768)]
for kind, (ta, tb) in LAY.items():
    for (M, N, Kk) in small + big:
        if kind == "TN" and M > 8192:
            M, N = N, 3072            # weight-gradient orientation: [N_out, N_in] = dY^T X over the rows
        a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
        b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
        bias = torch.randn(N, device=dev).bfloat16()
        for tile in (84, 83):
            if ta and tile == 83:
                continue
            check(f"{kind} {M}x{N}x{Kk} plain", lambda: [K.gemm(a, b, ta, tb)], tile)
            check(f"{kind} {M}x{N}x{Kk} bias alpha", lambda: [K.gemm(a, b, ta, tb, bias=bias, alpha=0.5)], tile, reps=1)

            def accf():
                acc = torch.ones(M, N, device=dev, dtype=torch.bfloat16)
                K.gemm(a, b, ta, tb, out=acc, accumulate=True)
                return [acc]
            check(f"{kind} {M}x{N}x{Kk} accumulate", accf, tile, reps=1)
# batched products (attention-shaped strides are exercised by the model tests; here plain 3-D)
for (Bt, M, N, Kk) in [(3, 512, 512, 256), (5, 300, 260, 128)]:
    a = torch.randn(Bt, M, Kk, device=dev).bfloat16()
    b = torch.randn(Bt, N, Kk, device=dev).bfloat16()
    check(f"batched NT {Bt}x{M}x{N}x{Kk}", lambda: [K.gemm(a, b, False, True)], 84)
# ragged weight-gradient row counts (zero source for A's missing rows), single products and the grouped launch
for (M, N, Kk) in [(256, 1024, 1568), (1024, 256, 1568), (768, 768, 40), (264, 200, 1025), (768, 3072, 13312)]:
    dy = torch.randn(Kk, M, device=dev).bfloat16()
    x = torch.randn(Kk, N, device=dev).bfloat16()
    check(f"TN ragged rows {M}x{N}x{Kk}", lambda: [K.gemm(dy, x, True, False)], 84)
groups = [
    [(768, 3072, 1024), (3072, 768, 1024), (2304, 768, 1024), (768, 768, 1024)],
    [(768, 768, 512), (1536, 768, 2048), (200, 776, 512), (8, 8, 64), (264, 256, 1088)],
    [(256, 1024, 1568), (1024, 256, 1568), (256, 2304, 1568), (768, 768, 40), (264, 200, 1025)],
    [(2304, 768, 13312), (768, 768, 13312), (3072, 768, 13312), (768, 3072, 13312)],
]
for gi, shapes in enumerate(groups):
    prods = []
    for i, (M, N, Kk) in enumerate(shapes):
        dy = torch.randn(Kk, M + 8 * (i % 2), device=dev).bfloat16()[:, :M]
        x = torch.randn(Kk, N, device=dev).bfloat16()
        prods.append((dy, x))

    def grp():
        outs = [torch.ones(dy.shape[1], x.shape[1], device=dev, dtype=torch.bfloat16) for dy, x in prods]
        q = K.FoldQueue()
        K.gemm_group_tn([(dy, x, o, 0.5 + 0.25 * i) for i, ((dy, x), o) in enumerate(zip(prods, outs))], q)
        q.flush()
        return outs
    check(f"group {gi}", grp, 0)
# against an fp32 reference too (the lockstep loop is not the oracle of anything: both could be wrong the same way)
setenv(23 if 23 in variants else variants[0], 84)
for kind, (ta, tb) in LAY.items():
    M, N, Kk = 1000, 776, 320
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    ref = (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
    out = K.gemm(a, b, ta, tb)
    e = float((out.float() - ref).abs().max() / ref.abs().max())
    nchk += 1
    if e > 1e-2:
        bad += 1
        print(f"MISMATCH vs fp32 reference {kind}: rel {e:.3g}")
setenv(0, 0)
print(f"gemm_pp_check: {nchk} comparisons, {bad} mismatching products, variants {variants}")
sys.exit(1 if bad else 0)
