"""A/B of the two main loops of the 256 x 256 / 192 x 256 GEMM tile on the products of the cfg-2 step: the compiler-scheduled lockstep loop
(gemm_big_kernel, csrc/gemm_mfma.hip) against the ping-pong loop (gemm_pp_kernel, csrc/gemm_pp.hip) and its variants, interleaved rounds in
ONE process (cdna_hip_programming.md section 5.4 rule 24), every variant's output compared bit for bit with the lockstep loop's (same MFMA
order per accumulator), torch.matmul (hipBLASLt) on the same operands as a yard-stick only.  Needs the DEBUG library:
  OFASYS_AMD_LIB=ofasys_amd/libofasys_amd_dbg.so python tools/gemm_pp_ab.py [quick]
OFA_GEMM_PP (variant) and OFA_GEMM_TILE (forced tile) are read per call by the debug library."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OFASYS_AMD_LIB", os.path.join(ROOT, "ofasys_amd", "libofasys_amd_dbg.so"))
from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"
PEAK = 2500.0
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
# variant = NKS*10 + {1: stagger + setprio, 0: stagger, 2: lockstep}; + 1000 + 100*NDL: only NDL LDS-DMA pieces per phase in the load segment
VARIANTS = [int(v) for v in sys.argv[2:]] or [21, 20, 22, 11]


def setenv(pp, tile):
    os.environ["OFA_GEMM_PP"] = str(pp)
    os.environ["OFA_GEMM_TILE"] = str(tile)


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


R, Rd = 13312, 1536
shapes = [
    ("NT", R, 2304, 768, 84, "qkv forward"), ("NT", R, 768, 768, 83, "out_proj forward"), ("NT", R, 3072, 768, 84, "fc1 forward"),
    ("NT", R, 768, 3072, 83, "fc2 forward"), ("NT", R, 9216, 768, 84, "cross k|v of 6 layers"),
    ("NT", Rd, 51272, 768, 84, "output projection"),
    ("NN", R, 768, 2304, 83, "qkv dgrad"), ("NN", R, 768, 768, 83, "out_proj dgrad"), ("NN", R, 768, 3072, 83, "fc1 dgrad"),
    ("NN", R, 3072, 768, 84, "fc2 dgrad"), ("NN", R, 768, 9216, 83, "cross k|v dgrad"),
    ("NT", 8192, 8192, 8192, 84, "large square"), ("NT", 4096, 4096, 4096, 84, "4096 cube"),
]
if quick:
    shapes = [shapes[0], shapes[2], shapes[3], shapes[6], shapes[8], shapes[11]]
NROUND, NL = int(os.environ.get('AB_ROUNDS', '3')), 20
print(f"# rounds {NROUND} x {NL} launches, median us; eq = bit-identical to the lockstep loop on the same tile", flush=True)
for kind, M, N, Kk, tile, what in shapes:
    ta, tb = {"NT": (False, True), "NN": (False, False), "TN": (True, False)}[kind]
    torch.manual_seed(0)
    a = torch.randn((Kk, M) if ta else (M, Kk), device=dev).bfloat16()
    b = torch.randn((N, Kk) if tb else (Kk, N), device=dev).bfloat16()
    outs = {}
    arms = [("plan", 0, 0), ("lock", 0, tile)] + [(f"pp{v}", v, tile) for v in VARIANTS]
    for name, pp, tl in arms:
        setenv(pp, tl)
        o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            K.gemm(a, b, ta, tb, out=o)
        outs[name] = o
    torch.cuda.synchronize()
    eq = {n: bool(torch.equal(outs[n], outs["lock"])) for n, _, _ in arms}
    A = a.t() if ta else a
    Bm = b.t() if tb else b
    o2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, Bm, out=o2)
    ts = {n: [] for n, _, _ in arms}
    ts["blaslt"] = []
    for _ in range(NROUND):
        for name, pp, tl in arms:
            setenv(pp, tl)
            o = outs[name]
            ts[name].append(timed(lambda: K.gemm(a, b, ta, tb, out=o), NL))
        ts["blaslt"].append(timed(lambda: torch.matmul(A, Bm, out=o2), NL))
    fl = 2.0 * M * N * Kk
    parts = []
    for name in [n for n, _, _ in arms] + ["blaslt"]:
        t = statistics.median(ts[name])
        parts.append(f"{name} {t:7.1f} ({fl / t / 1e6:6.0f} TF{'' if name == 'blaslt' or eq.get(name, True) else ' NEQ!'})")
    print(f"{kind} {M:6d}x{N:6d}x{Kk:6d} {what:24s} " + " | ".join(parts), flush=True)

# one encoder layer's four weight gradients, grouped
setenv(0, 0)
dys = [torch.randn(R, n, device=dev).bfloat16() for n in (2304, 768, 3072, 768)]
xs = [torch.randn(R, k, device=dev).bfloat16() for k in (768, 768, 768, 3072)]


def grouped(outs):
    f = K.FoldQueue()
    K.gemm_group_tn([(dy, x, o, 1.0) for dy, x, o in zip(dys, xs, outs)], f)
    f.flush()


res = {}
ts = {}
for v in [0] + VARIANTS:
    setenv(v, 0)
    outs = [torch.zeros(dy.shape[1], x.shape[1], device=dev, dtype=torch.bfloat16) for dy, x in zip(dys, xs)]
    grouped(outs)
    res[v] = outs
    ts[v] = []
torch.cuda.synchronize()
for _ in range(NROUND):
    for v in [0] + VARIANTS:
        setenv(v, 0)
        scratch = [torch.zeros_like(o) for o in res[v]]
        ts[v].append(timed(lambda: grouped(scratch), 10))
fl = sum(2.0 * R * dy.shape[1] * x.shape[1] for dy, x in zip(dys, xs))
parts = []
for v in [0] + VARIANTS:
    t = statistics.median(ts[v])
    same = all(torch.equal(x, y) for x, y in zip(res[v], res[0]))
    parts.append(f"{'lock' if v == 0 else 'pp%d' % v} {t:7.1f} ({fl / t / 1e6:6.0f} TF{'' if same else ' NEQ!'})")
print("grouped weight gradients of one encoder layer (folds included): " + " | ".join(parts), flush=True)
setenv(0, 0)
