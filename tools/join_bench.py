"""Residual-join kernels (csrc/join.hip): the row-per-wave forms against the split-row forms they replace -- results and time.
Needs the DEBUG library (the switches OFA_JOIN_FWD / OFA_JOIN_BWD select the kernel per call).
  python tools/join_bench.py           results + time per call
  python tools/join_bench.py check     results only, more shapes (tests/test_kernels_gpu.py runs this); exit status 1 on a mismatch"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OFASYS_AMD_LIB", os.path.join(ROOT, "ofasys_amd", "libofasys_amd_dbg.so"))
import torch  # noqa: E402

from ofasys_amd import kernels as K  # noqa: E402

dev = "cuda"
CHECK = len(sys.argv) > 1 and sys.argv[1] == "check"
BAD = []


def timed(fn, n=20):
    """us per call inside a replayed hipGraph of n calls (no host launch cost in the figure)"""
    if CHECK:
        return 0.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def case(rows, cols, has_a, has_b, p, dtype=torch.bfloat16):
    torch.manual_seed(rows + cols)
    x = torch.randn(rows, cols, device=dev).to(dtype)
    r = torch.randn(rows, cols, device=dev).to(dtype)
    mk = lambda s, o: (torch.randn(cols, device=dev) * s + o).to(dtype)  # noqa: E731
    lna = (mk(0.1, 1.0), mk(0.1, 0.0)) if has_a else None
    lnb = (mk(0.1, 1.0), mk(0.1, 0.0)) if has_b else None
    dy = torch.randn(rows, cols, device=dev).to(dtype)
    dz = torch.randn(rows, cols, device=dev).to(dtype) if has_b else None
    out = {}
    for name, fv, bv in (("split", "0", "0"), ("row8x1", "1", "1"), ("row4x2", "1", "2"), ("row8x2", "1", "3")):
        os.environ["OFA_JOIN_FWD"], os.environ["OFA_JOIN_BWD"] = fv, bv
        y, z, stats, keep = K.join_fwd(x, r, lna, lnb, 1e-5, p, 1234, 77, None)
        grads = [torch.zeros(cols, device=dev) if on else None for on in (has_a, has_a, has_b, has_b)] + [torch.zeros(cols, device=dev)]
        torch.cuda.synchronize()
        dres, dx = K.join_bwd(dy, dz, x if has_a else None, y if has_b else None, lna[0] if has_a else None, lnb[0] if has_b else None,
                              stats, p, 1234, 77, None, tuple(grads), keep=keep)
        torch.cuda.synchronize()
        tf = timed(lambda: K.join_fwd(x, r, lna, lnb, 1e-5, p, 1234, 77, None))
        tb = timed(lambda: K.join_bwd(dy, dz, x if has_a else None, y if has_b else None, lna[0] if has_a else None,
                                      lnb[0] if has_b else None, stats, p, 1234, 77, None, tuple(grads), keep=keep))   # (+ its fold launch)
        out[name] = (y, z, dres, dx, grads, tf, tb)
    ref = out["split"]
    line = f"rows {rows:6d} cols {cols:5d} LN_a {int(has_a)} LN_b {int(has_b)} p {p:.1f}:"
    for name in ("split", "row8x1", "row4x2", "row8x2"):
        y, z, dres, dx, grads, tf, tb = out[name]
        line += f"  {name} fwd {tf:6.1f} bwd {tb:6.1f} us"
    print(line)
    for name in ("row8x1", "row4x2", "row8x2"):
        y, z, dres, dx, grads, tf, tb = out[name]
        ney = float((y != ref[0]).float().mean())
        errs = [rel(y, ref[0]), rel(z, ref[1]) if has_b else 0.0, rel(dres, ref[2]), rel(dx, ref[3])] + \
               [rel(g, h) for g, h in zip(grads, ref[4]) if g is not None]
        ok = ney < 1e-3 and max(errs) < 3e-3 and (has_a or ney == 0.0)
        if not ok:
            BAD.append((rows, cols, has_a, has_b, p, name))
        print(f"    {name}: y differs on {ney:.2e} of the elements; rel err y/z/dres/dx/grads = " + " ".join(f"{e:.1e}" for e in errs) + ("  OK" if ok else "  MISMATCH"))


if CHECK:
    for rows, cols in ((13312, 768), (1, 768), (7, 256), (2049, 512), (3000, 1024), (33, 768)):
        for has_a, has_b in ((True, True), (False, True), (True, False), (False, False)):
            for p in (0.1, 0.0):
                case(rows, cols, has_a, has_b, p)
    case(3072, 768, True, True, 0.1, torch.float16)
    case(100, 1024, False, True, 0.3, torch.float16)
    print(f"{len(BAD)} mismatching cases", BAD)
    sys.exit(1 if BAD else 0)
for rows, cols in ((13312, 768), (3072, 768), (13312, 1024), (5000, 256), (4099, 512)):
    for has_a, has_b in ((True, True), (False, True), (True, False), (False, False)):
        if (has_a and has_b) or rows == 13312:
            case(rows, cols, has_a, has_b, 0.1)
case(13312, 768, True, True, 0.0)
case(3072, 768, True, True, 0.1, torch.float16)
