"""GPU parity tests of every C-ABI kernel against a plain PyTorch fp32 reference of the same op (and the oracle's
restatements where the op is OFASys-specific).  Run with: pytest -m gpu"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ofasys_amd import kernels
    return kernels


DEV = "cuda"


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(5, 256), (130, 768), (67, 3072), (9, 1024), (3, 64), (5000, 768), (4100, 3072)])
@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm(K, dtype, rows, cols, gelu):
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device=DEV).to(dtype)
    g = (1 + 0.1 * torch.randn(cols, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(cols, device=DEV)).to(dtype)
    dy = torch.randn(rows, cols, device=DEV).to(dtype)
    xr = x.float().requires_grad_(True)
    gr = g.float().requires_grad_(True)
    br = b.float().requires_grad_(True)
    yr = F.layer_norm(F.gelu(xr) if gelu else xr, (cols,), gr, br, 1e-5)
    yr.backward(dy.float())
    y, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, fuse_gelu=gelu)
    dx, dg, db, dbias = K.layernorm_bwd(dy, x, g, mean, rstd, fuse_gelu=gelu, want_dbias=gelu)
    t = tol(dtype)
    big = 4 if rows > 1000 else 1                     # bf16 outputs of long column sums round at ~2^-8 of a larger value
    assert rel(y, yr) < t
    assert rel(dx, xr.grad) < 2 * t
    assert rel(dg, gr.grad) < 2 * t * big
    assert rel(db, br.grad) < 2 * t * big
    if not gelu:                                      # residual-branch gradient added inside the kernel
        dres = torch.randn_like(dy)
        dx2 = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres)[0]
        assert rel(dx2, xr.grad + dres.float()) < 2 * t
    if gelu:
        assert rel(dbias, xr.grad.sum(0)) < 2 * t * big
        # accumulate mode adds onto existing gradients (the train step's arena)
        acc = [torch.ones(cols, device=DEV, dtype=dtype) for _ in range(3)]
        K.layernorm_bwd(dy, x, g, mean, rstd, fuse_gelu=True, dgamma=acc[0], dbeta=acc[1], dbias=acc[2])
        assert rel(acc[0].float() - 1, gr.grad) < 4 * t * big and rel(acc[2].float() - 1, xr.grad.sum(0)) < 4 * t * big


GEMM_SHAPES = [(64, 64, 64), (128, 128, 128), (200, 136, 72), (130, 768, 256), (534, 264, 768), (77, 64, 1032),
               (256, 3072, 768), (1000, 208, 264), (1024, 256, 1568), (256, 2304, 40)]      # (the last two: ragged-K weight-gradient shapes)


@pytest.mark.parametrize("M,N,K_", [(3584, 4096, 4096), (3320, 3848, 4160), (4032, 3072, 256), (13312, 768, 768)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_big_tile(K, M, N, K_, ta, tb):
    """Products the planner sends to the 256x256 / 192x256 eight-wave tiles (csrc/gemm_mfma.hip, gemm_big_kernel): long K, one
    round of 192 x 256 tiles, and the packed train step's 13312-row shape."""
    torch.manual_seed(11)
    a = torch.randn((K_, M) if ta else (M, K_), device=DEV).bfloat16()
    b = torch.randn((N, K_) if tb else (K_, N), device=DEV).bfloat16()
    bias = torch.randn(N, device=DEV).bfloat16()
    ref = ((a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float()) + bias.float()) * 0.5
    for _ in range(3):            # repeat: a missing DMA wait only shows under load (warm caches, all CUs streaming)
        out = K.gemm(a, b, ta, tb, bias=bias, alpha=0.5)
        assert rel(out, ref) < 1e-2
    acc = torch.ones(M, N, device=DEV, dtype=torch.bfloat16)
    K.gemm(a, b, ta, tb, out=acc, accumulate=True)
    assert rel(acc.float() - 1, ref * 2 - bias.float()) < 2e-2
    assert rel(K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, out_f32=True), ref) < 2e-3


_FORCED_TILE_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from ofasys_amd import kernels as K
torch.manual_seed(5)
rel = lambda a, b: float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))
for M, N, K_ in [(600, 520, 192), (1000, 768, 256), (2050, 1288, 128), (192, 256, 64), (3333, 264, 448)]:
    for ta, tb in [(False, True), (False, False), (True, False), (True, True)]:
        a = torch.randn((K_, M) if ta else (M, K_), device="cuda").bfloat16()
        b = torch.randn((N, K_) if tb else (K_, N), device="cuda").bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16()
        brow = torch.randn(M, device="cuda").bfloat16()
        prod = (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
        for _ in range(2):
            assert rel(K.gemm(a, b, ta, tb, bias=bias, alpha=0.5), (prod + bias.float()) * 0.5) < 1e-2, (M, N, K_, ta, tb)
        assert rel(K.gemm(a, b, ta, tb, bias=brow, bias_row=True), prod + brow.float()[:, None]) < 1e-2, (M, N, K_, ta, tb, "row")
        assert rel(K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, out_f32=True), (prod + bias.float()) * 0.5) < 2e-3, (M, N, K_, ta, tb, "f32")
        acc = torch.ones(M, N, device="cuda", dtype=torch.bfloat16)
        K.gemm(a, b, ta, tb, out=acc, accumulate=True)
        assert rel(acc.float() - 1, prod) < 2e-2, (M, N, K_, ta, tb, "acc")
        wide = torch.zeros(M, N + 24, device="cuda", dtype=torch.bfloat16)         # ldc > N: nothing outside the view is written
        K.gemm(a, b, ta, tb, out=wide[:, :N])
        assert rel(wide[:, :N], prod) < 1e-2 and float(wide[:, N:].abs().max()) == 0.0, (M, N, K_, ta, tb, "ldc")
print("forced tiles ok")
"""


@pytest.mark.parametrize("pp", ["0", "23"])
@pytest.mark.parametrize("tile", ["83", "84"])
def test_gemm_eight_wave_tiles_forced(K, tile, pp):
    """Every product of the list on the eight-wave 192 x 256 / 256 x 256 kernels (gemm_big_kernel<3|4, 2, .., 2, 4>): ragged edges in M
    and N, all four operand layouts, column / row bias, alpha, fp32 output, accumulation, ldc > N.  OFA_GEMM_TILE exists in the DEBUG library only
    (libofasys_amd_dbg.so, OFASYS_AMD_LIB), hence the subprocess; the planner's own choice of these kernels is covered by test_gemm_big_tile.
    pp = 23: the same products through the ping-pong main loop (csrc/gemm_pp.hip) in every layout it is built for -- the shipped planner
    sends it the m-major-operand products only; pp = 0 forces the lockstep loop for all of them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = os.path.join(root, "ofasys_amd", "libofasys_amd_dbg.so")            # (make -C ofasys_amd/csrc debug; built by __graft_entry__.build)
    assert os.path.exists(dbg), "the debug library (planner overrides compiled in) is not built: make -C ofasys_amd/csrc debug"
    env = dict(os.environ, OFA_GEMM_TILE=tile, OFASYS_AMD_LIB=dbg, OFA_GEMM_PP=pp)
    r = subprocess.run([sys.executable, "-c", _FORCED_TILE_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced tiles ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_MIXED_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from ofasys_amd import kernels as K
torch.manual_seed(12)
def run(a, b, tb, mode, **kw):
    os.environ["OFA_GEMM_MIXED"] = mode
    return K.gemm(a, b, False, tb, **kw)
n = 0
for M, N, K_ in [(13312, 3072, 768), (12800, 3072, 768), (13001, 3080, 768), (2048, 9216, 256), (1100, 520, 1024), (5000, 1024, 512)]:
    for tb in (True, False):
        a = torch.randn(M, K_, device="cuda").bfloat16()
        b = torch.randn((N, K_) if tb else (K_, N), device="cuda").bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16()
        ref = a.float() @ (b.float().t() if tb else b.float())
        plain = run(a, b, tb, "0", bias=bias, alpha=0.5)
        mixed = run(a, b, tb, "1", bias=bias, alpha=0.5)
        assert torch.equal(plain, mixed), (M, N, K_, tb)
        assert float((mixed.float() - (ref + bias.float()) * 0.5).abs().max()) <= 1e-2 * float(ref.abs().max()), (M, N, K_, tb)
        f0, f1 = run(a, b, tb, "0", out_f32=True), run(a, b, tb, "1", out_f32=True)
        assert torch.equal(f0, f1), (M, N, K_, tb, "f32")
        acc0 = torch.ones(M, N, device="cuda", dtype=torch.bfloat16); acc1 = acc0.clone()
        run(a, b, tb, "0", out=acc0, accumulate=True); run(a, b, tb, "1", out=acc1, accumulate=True)
        assert torch.equal(acc0, acc1), (M, N, K_, tb, "acc")
        wide = torch.zeros(M, N + 24, device="cuda", dtype=torch.bfloat16)         # ldc > N: nothing outside the view is written
        run(a, b, tb, "1", out=wide[:, :N])
        assert float(wide[:, N:].abs().max()) == 0.0, (M, N, K_, tb, "ldc")
        n += 1
print("mixed tiles ok", n)
"""


def test_gemm_mixed_tile_heights_bit_identical_to_plain_plan(K):
    """gemm_big_mixed_kernel (256- and 192-row tiles in one launch against round quantisation, csrc/gemm_mfma.hip) forced wherever it is
    eligible (OFA_GEMM_MIXED=1, debug library) against the plain plan (=0): bit-identical outputs -- k-major and m-major B, column bias,
    alpha, fp32 output, 16-bit accumulation, ldc > N, ragged M and N -- and correct against an fp32 product."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = os.path.join(root, "ofasys_amd", "libofasys_amd_dbg.so")
    assert os.path.exists(dbg), "the debug library (planner overrides compiled in) is not built: make -C ofasys_amd/csrc debug"
    r = subprocess.run([sys.executable, "-c", _MIXED_SCRIPT, root], env=dict(os.environ, OFASYS_AMD_LIB=dbg), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "mixed tiles ok 12" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_gemm_ping_pong_loop_bit_identical_to_lockstep_loop():
    """tools/gemm_pp_check.py: every product (K = 1 .. 7 tiles and long, ragged M / N, NT / NN / TN, bias / alpha / accumulate, batched, ragged
    weight-gradient row counts, the grouped launch) through the ping-pong loop and through the lockstep loop on the same tile, compared bit
    for bit, repeatedly (a missing LDS-DMA wait or a too-early buffer refill shows as a rare mismatch); plus one fp32 reference per layout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_pp_check.py"), "23", "21"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " 0 mismatching products" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K_", GEMM_SHAPES)
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm(K, dtype, M, N, K_, ta, tb):
    torch.manual_seed(1)
    a = torch.randn((K_, M) if ta else (M, K_), device=DEV).to(dtype)
    b = torch.randn((N, K_) if tb else (K_, N), device=DEV).to(dtype)
    bias = torch.randn(N, device=DEV).to(dtype)
    ref = ((a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float()) + bias.float()) * 0.5
    out = K.gemm(a, b, ta, tb, bias=bias, alpha=0.5)
    assert out.dtype == dtype
    t = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel(out, ref) < t
    if dtype == torch.bfloat16:
        # exact-tier kernel on the same bf16 data, fp32 output and accumulate flag
        out32 = K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, out_f32=True)
        assert out32.dtype == torch.float32 and rel(out32, ref) < 2e-3
        outs = K.gemm(a, b, ta, tb, bias=bias, alpha=0.5, force_simple=True)
        assert rel(outs, ref) < 1e-2
        acc = out32.clone()
        K.gemm(a, b, ta, tb, alpha=1.0, out=acc, accumulate=True)
        ref2 = ref + (a.float().t() if ta else a.float()) @ (b.float().t() if tb else b.float())
        assert rel(acc, ref2) < 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shapes", [
    [(768, 3072, 1024), (3072, 768, 1024), (2304, 768, 1024), (768, 768, 1024)],          # one encoder layer's four products
    [(768, 768, 512), (1536, 768, 2048), (200, 776, 512), (8, 8, 64), (264, 256, 1088)],   # unequal K, ragged M / N, a tiny one
    [(256, 256, 64)] * 8,                                                                  # the most one launch takes
    # row counts that are not a multiple of the 64-row K tile (round 4: A's missing rows are read as zeros inside the kernel):
    # 1568 = 8 x 14 x 14 positions of the ResNet trunk at micro-batch 8, a contraction shorter than one tile, 64 k + 1 rows
    [(256, 1024, 1568), (1024, 256, 1568), (256, 2304, 1568), (768, 768, 40), (264, 200, 1025)],
])
def test_gemm_group_tn(K, dtype, shapes):
    """ofa_gemm_group_tn + ofa_fold_batched: out_p += alpha_p * dy_p^T x_p for a group of weight-gradient products, against fp32
    matmuls of the same 16-bit data (K-slices are fp32 partial sums: summation order is the only difference)."""
    torch.manual_seed(5)
    prods, refs = [], []
    for i, (M, N, Kk) in enumerate(shapes):
        dy = torch.randn(Kk, M + 8 * (i % 2), device=DEV).to(dtype)[:, :M]      # every other operand with lda > m
        x = torch.randn(Kk, N, device=DEV).to(dtype)
        out = torch.randn(M, N, device=DEV)
        alpha = 0.5 + 0.25 * i
        refs.append(out + alpha * (dy.float().t() @ x.float()))
        assert K.gemm_group_ok(dy, x, out)
        prods.append((dy, x, out, alpha))
    q = K.FoldQueue()
    K.gemm_group_tn(prods, q)
    q.flush()
    for (dy, x, out, alpha), ref in zip(prods, refs):
        assert rel(out, ref) < 2e-5 * max(1.0, math.sqrt(dy.shape[0] / 64)), (tuple(out.shape), rel(out, ref))
    # the same group again lands on top (accumulate): twice the product
    K.gemm_group_tn(prods, q)
    q.flush()
    dy, x, out, alpha = prods[0]
    assert rel(out, refs[0] + alpha * (dy.float().t() @ x.float())) < 1e-4
    # a ragged contraction is eligible (since round 4); fp32 operands are not
    assert K.gemm_group_ok(dy[:40], x[:40], out)
    assert not K.gemm_group_ok(dy.float(), x.float(), out)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_group_tn_direct_accumulation_over_micro_batches(K, dtype):
    """update_freq > 1 with small micro-batches: every micro-batch's one-slice weight gradient lands on the 16-bit arena gradient in the
    kernel's epilogue (round the product, add, round -- the reference's `p.grad += g` in 16 bits).  Four accumulations against the fp64
    sum: within the four half-ulp roundings of the 16-bit format, and no worse than torch's own 16-bit `+=` of the same four products."""
    torch.manual_seed(9)
    M, N, Kk, steps = 768, 1024, 256, 4
    out = torch.zeros(M, N, device=DEV, dtype=dtype)
    emul = torch.zeros(M, N, device=DEV, dtype=dtype)
    ref = torch.zeros(M, N, device=DEV, dtype=torch.float64)
    for _ in range(steps):
        dy = torch.randn(Kk, M, device=DEV).to(dtype)
        x = torch.randn(Kk, N, device=DEV).to(dtype)
        q = K.FoldQueue()
        K.gemm_group_tn([(dy, x, out, 0.5)], q)
        q.flush()
        prod = 0.5 * (dy.double().t() @ x.double())
        ref += prod
        emul += prod.to(dtype)                        # what the reference does: a 16-bit gradient added in 16 bits
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    scale = float(ref.abs().max())
    err, err_emul = float((out.double() - ref).abs().max()) / scale, float((emul.double() - ref).abs().max()) / scale
    print(f"MEASURED direct 16-bit accumulation over {steps} micro-batches ({dtype}): max err {err:.2e} of the largest entry (torch 16-bit += : {err_emul:.2e})")
    assert err <= steps * eps and err <= 1.5 * err_emul + eps / 4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K_,bias", [
    (6272, 256, 1024, False), (6272, 1024, 256, False), (1568, 256, 2304, False),      # ResNet layer3 products (B = 32 / 8), 1 x 1 and 3 x 3
    (25088, 128, 1152, False), (784, 1024, 256, True), (200, 136, 72, True),           # layer2; micro-batch 4; ragged M / N / K
    (13312, 768, 768, True), (13312, 2304, 768, False), (2048, 768, 768, True),        # the eight-wave tiles, the four-wave ring tiles
    (100352, 64, 64, False),                                                            # more wave blocks than max_groups: no statistics
])
def test_gemm_colstat(K, dtype, M, N, K_, bias):
    """ofa_gemm_colstat: the product is bit-identical to ofa_gemm's, and the partial rows its epilogue leaves sum to the column sums /
    sums of squares of the ROUNDED output (what the BatchNorm statistics kernel would read back from memory)."""
    torch.manual_seed(3)
    a = (torch.randn(M, K_, device=DEV) * 0.5).to(dtype)
    b = (torch.randn(N, K_, device=DEV) * 0.5).to(dtype)
    bv = torch.randn(N, device=DEV).to(dtype) if bias else None
    out, part = K.gemm_colstat(a, b, bias=bv)
    ref = K.gemm(a, b, False, True, bias=bv)
    assert torch.equal(out, ref)
    from ofasys_amd.lib import lib
    split = lib().cdll.ofa_gemm_splits(M, N, K_, 0, 1, 1, 0, 1, 256 << 20) > 1        # a split-K plan finishes in the reduce kernel
    if M >= 100000 or split:
        assert part is None and (split == (K_ >= 2048 and M < 4096) or M >= 100000)
        return
    assert part is not None and part.dtype == torch.float64 and part.shape[1:] == (2, N) and part.shape[0] <= 512
    got = part.sum(0)
    o = out.double()
    want = torch.stack([o.sum(0), (o * o).sum(0)])
    scale = torch.stack([o.abs().sum(0), (o * o).sum(0)]).clamp_min(1e-30)
    assert float(((got - want).abs() / scale).max()) < 2e-6, float(((got - want).abs() / scale).max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(8, 14, 14, 256, 1024, 1, 1, 0), (8, 28, 28, 128, 128, 3, 2, 1), (2, 56, 56, 64, 64, 3, 1, 1)])
def test_conv_bn_statistics_from_the_gemm_epilogue(K, dtype, shape):
    """ops.conv_bn (BatchNorm statistics from the convolution's GEMM epilogue) against conv2d + batch_norm (statistics kernel over the
    stored output): same outputs to a rounding of the 16-bit type, same running statistics, same gradients."""
    from ofasys_amd import ops
    B, H, W, Cin, Cout, k, stride, pad = shape
    torch.manual_seed(4)
    x0 = torch.randn(B * H * W, Cin, device=DEV).to(dtype)
    conv = torch.nn.Conv2d(Cin, Cout, k, stride, pad, bias=False).to(DEV).to(dtype)
    res = {}
    for fused in (True, False):
        bn = torch.nn.BatchNorm2d(Cout).to(DEV).to(dtype).train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, Cout))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, Cout))
        conv.weight.grad = None
        x = x0.clone().requires_grad_(True)
        if fused:
            y, Ho, Wo = ops.conv_bn(x, conv, bn, B, H, W, relu=True)
        else:
            t, Ho, Wo = ops.conv2d(x, conv.weight, None, B, H, W, stride, pad)
            y = ops.batch_norm(t, bn, relu=True)
        dy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)).to(dtype)
        y.backward(dy)
        torch.cuda.synchronize()
        res[fused] = (y.detach().float(), bn.running_mean.float().clone(), bn.running_var.float().clone(), x.grad.float(),
                      conv.weight.grad.float().clone(), bn.weight.grad.float().clone())
    tol = 1.6e-2 if dtype == torch.bfloat16 else 2e-3
    for a, b, t in zip(res[True], res[False], (tol, 1e-4, 1e-4, 4 * tol, 4 * tol, 4 * tol)):
        assert rel(a, b) < t, (rel(a, b), t)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_group_tn_one_slice_products_accumulate_without_slabs(K, dtype):
    """A grouped weight-gradient launch whose products are ONE K-slice each (a small micro-batch: few rows) adds alpha * dy^T x straight
    onto the 16-bit gradient in the kernel's epilogue -- no fp32 slab, nothing queued for the fold -- also on top of an earlier
    contribution, with ragged M / N tiles, a row count that is not a multiple of 64, and the same output twice in one group (the
    second contribution then takes the slab path: two read-modify-writes of one output must not share a launch)."""
    torch.manual_seed(6)
    shapes = [(1024, 1024, 512), (3072, 1024, 512), (264, 200, 136), (1024, 4096, 448)]
    prods, refs = [], []
    for i, (M, N, Kk) in enumerate(shapes):
        dy = (torch.randn(Kk, M, device=DEV) * 0.2).to(dtype)
        x = (torch.randn(Kk, N, device=DEV) * 0.2).to(dtype)
        out = (torch.randn(M, N, device=DEV)).to(dtype)
        alpha = 1.0 if i % 2 == 0 else 0.5
        refs.append(out.float() + alpha * (dy.float().t() @ x.float()))
        prods.append((dy, x, out, alpha))
    dy0, x0, out0, a0 = prods[0]
    prods.append((dy0, x0, out0, a0))                       # the same output again
    refs[0] = refs[0] + a0 * (dy0.float().t() @ x0.float())
    q = K.FoldQueue()
    K.gemm_group_tn(prods, q)
    assert len(q.jobs) == 1                                 # only the duplicate went through a slab
    q.flush()
    tol = 1.6e-2 if dtype == torch.bfloat16 else 2e-3
    for (dy, x, out, alpha), ref in zip(prods[:4], refs):
        assert rel(out.float(), ref) < tol, (tuple(out.shape), rel(out.float(), ref))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,shape", [(2, (12, 64, 64)), (6, (4, 7, 7)), (12, (16, 33, 41)), (16, (1, 5))])
def test_add_n_and_fan_out(K, dtype, n, shape):
    """ofa_add_n: the sum of n tensors in one launch (fp32 accumulation, one rounding; a tail that is not a whole 16-byte vector), and
    ops.fan_out: n consumers of one tensor get their gradients summed by it."""
    from ofasys_amd import ops
    torch.manual_seed(8)
    ts = [torch.randn(*shape, device=DEV).to(dtype) for _ in range(n)]
    got = K.add_n(ts)
    want = torch.stack([t.float() for t in ts]).sum(0)
    assert got.dtype == dtype and rel(got.float(), want) < (1e-6 if dtype == torch.float32 else 8e-3)
    x = torch.randn(*shape, device=DEV).to(dtype).requires_grad_(True)
    views = ops.fan_out(x, n + 3)                                      # (19 consumers with n = 16: two launches)
    loss = sum((i + 1) * (v.float() * ts[i % n].float()).sum() for i, v in enumerate(views[:-1]))      # the last view stays unused
    loss.backward()
    wantg = sum((i + 1) * ts[i % n].float() for i in range(n + 2))
    assert rel(x.grad.float(), wantg) < (1e-6 if dtype == torch.float32 else 1.6e-2)


def test_step_stats_add(K):
    """ofa_step_stats_add: [sample_size, loss_sum, ntokens] += (non-pad targets, loss, non-pad targets), accumulated over calls."""
    torch.manual_seed(10)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    want = torch.zeros(3, dtype=torch.float64)
    for n in (1, 255, 256, 2048, 5000):
        t = torch.randint(0, 7, (n,), device=DEV)
        loss = torch.rand((), device=DEV) * 1000
        K.step_stats_add(stats, loss, t, 1)
        c = float((t != 1).sum())
        want += torch.tensor([c, float(loss), c], dtype=torch.float64)
    assert torch.equal(stats.cpu()[[0, 2]], want[[0, 2]]) and abs(float(stats[1]) - float(want[1])) < 1e-9 * float(want[1])


def test_fan_out_of_a_deep_stack(K):
    """More consumers than two launches of ofa_add_n hold (16 + 15 = 31): every gradient is in the sum exactly once."""
    from ofasys_amd import ops
    torch.manual_seed(9)
    x = torch.randn(3, 17, device=DEV).requires_grad_(True)
    for n in (31, 32, 47):
        x.grad = None
        ws = [torch.randn(3, 17, device=DEV) for _ in range(n)]
        sum((v * w).sum() for v, w in zip(ops.fan_out(x, n), ws)).backward()
        assert rel(x.grad, torch.stack(ws).sum(0)) < 1e-6, n


def test_gemm_splitk_and_batched(K):
    torch.manual_seed(2)
    # wgrad shape: skinny output, long contraction -> split-K path
    dy = torch.randn(4096, 768, device=DEV).bfloat16()
    x = torch.randn(4096, 256, device=DEV).bfloat16()
    dw = K.gemm(dy, x, True, False)
    assert rel(dw, dy.float().t() @ x.float()) < 1e-2
    # row-bias + batched (the V^T projection layout): C_b[N,T] = W[N,K] X_b[T,K]^T
    W = torch.randn(192, 128, device=DEV).bfloat16()
    X = torch.randn(3, 40, 128, device=DEV).bfloat16()
    bias = torch.randn(192, device=DEV).bfloat16()
    out = torch.zeros(3, 192, 64, device=DEV, dtype=torch.bfloat16)
    K.gemm(W.unsqueeze(0).expand(3, -1, -1), X, False, True, bias=bias, bias_row=True, out=out[:, :, :40])
    ref = torch.einsum("nk,btk->bnt", W.float(), X.float()) + bias.float()[None, :, None]
    assert rel(out[:, :, :40], ref) < 1e-2 and float(out[:, :, 40:].abs().max()) == 0.0


def test_fused_softmax_backward_golden(K, golden_dir):
    """The three backward entry points + the fp16 dtype against autograd through the reference's own forward_torch_softmax
    (tests/golden/fused_softmax_bwd.npz): in place on dy, zero above the diagonal for the causal variant."""
    import numpy as np
    g = np.load(golden_dir + "/fused_softmax_bwd.npz")
    scale = float(g["scale"][0])
    mask = torch.from_numpy(g["mask"]).to(DEV)
    b, np_, sq, _ = g["x"].shape
    for tag, dt, ytol, dtol in (("f32", torch.float32, 1e-6, 5e-6), ("f16", torch.float16, 1e-3, 2e-3)):
        x = torch.from_numpy(g["x"]).to(DEV).to(dt)
        dy = torch.from_numpy(g["dy"]).to(DEV).to(dt)
        for name in ("plain", "masked", "causal"):
            if name == "plain":
                y = K.scaled_softmax(x, scale)
            elif name == "masked":
                y = K.scaled_masked_softmax(x, mask, scale)
            else:
                y = K.scaled_upper_triang_masked_softmax(x.view(-1, sq, sq), scale).view(b, np_, sq, sq)
            assert y.dtype == dt
            assert rel(y.float().cpu(), torch.from_numpy(g[f"{tag}.{name}.y"].astype(np.float32))) < ytol, (tag, name)
            d = dy.clone()
            if name == "plain":
                dx = K.scaled_softmax_bwd(d, y, scale, inplace=True)
            elif name == "masked":
                dx = K.scaled_masked_softmax_bwd(d, y, scale)
            else:
                d3 = d.view(-1, sq, sq)
                d3 += torch.triu(torch.full_like(d3, float("nan")), 1)          # never read above the diagonal
                dx = K.scaled_upper_triang_masked_softmax_bwd(d3, y.view(-1, sq, sq), scale).view(b, np_, sq, sq)
                assert float(torch.triu(dx.float(), 1).abs().max()) == 0.0
            assert dx.data_ptr() == d.data_ptr()                                 # completely in place, as the reference
            assert rel(dx.float().cpu(), torch.from_numpy(g[f"{tag}.{name}.dx"].astype(np.float32))) < dtol, (tag, name)
            if name != "plain":                                                  # out-of-place form gives the same bits
                fn = K.scaled_masked_softmax_bwd if name == "masked" else K.scaled_upper_triang_masked_softmax_bwd
                shp = y.shape if name == "masked" else (-1, sq, sq)
                src = dy.clone().view(shp) if name == "masked" else dy.clone().view(shp)
                assert torch.equal(fn(src, y.view(shp), scale, inplace=False).view(dx.shape), dx)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("sk", [16, 40, 129, 448, 1000])
def test_softmax_family(K, dtype, sk):
    from oracle import restate
    torch.manual_seed(3)
    x = (2 * torch.randn(2, 3, 8, sk, device=DEV)).to(dtype)
    t = 1e-6 if dtype == torch.float32 else (1e-2 if dtype == torch.bfloat16 else 2e-3)
    y = K.scaled_softmax(x, 0.37)
    assert rel(y, restate.scaled_softmax(x.cpu(), 0.37).to(DEV)) < max(t, 1e-6)
    mask = torch.rand(2, 1, 8, sk, device=DEV) > 0.8
    ym = K.scaled_masked_softmax(x, mask, 0.37)
    assert rel(ym, restate.scaled_masked_softmax(x.cpu(), mask.cpu(), 0.37).to(DEV)) < max(t, 1e-6)
    ym1 = K.scaled_masked_softmax(x, mask[:1], 0.37)
    assert rel(ym1, restate.scaled_masked_softmax(x.cpu(), mask[:1].cpu(), 0.37).to(DEV)) < max(t, 1e-6)
    dy = torch.randn_like(x)
    dx = K.scaled_softmax_bwd(dy, y, 0.37)
    assert rel(dx, restate.scaled_softmax_bwd(dy.cpu(), y.cpu(), 0.37).to(DEV)) < max(10 * t, 1e-5)
    dyc = dy.clone()
    dxi = K.scaled_softmax_bwd(dyc, y, 0.37, inplace=True)
    assert dxi.data_ptr() == dyc.data_ptr() and torch.equal(dxi, dx)
    if sk <= 448:
        xc = x[0, :, :, :8].contiguous() if sk >= 8 else None
        sq = torch.randn(6, sk, sk, device=DEV).to(dtype)
        yc = K.scaled_upper_triang_masked_softmax(sq, 0.5)
        assert rel(yc, restate.scaled_upper_triang_masked_softmax(sq.cpu(), 0.5).to(DEV)) < max(t, 1e-6)
        assert float(torch.triu(yc.float(), 1).abs().max()) == 0.0


def test_fused_softmax_golden(K, golden_dir):
    import numpy as np
    g = np.load(golden_dir + "/fused_softmax.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    s = float(g["scale"][0])
    assert rel(K.scaled_softmax(x, s).cpu(), torch.from_numpy(g["y"])) < 1e-6
    assert rel(K.scaled_masked_softmax(x, torch.from_numpy(g["mask"]).to(DEV), s).cpu(), torch.from_numpy(g["y_masked"])) < 1e-6
    from oracle import restate
    for a in [(128, 128, 2, 4), (64, 448, 2, 4), (4, 16, 1, 1), (8, 4096, 1, 1)]:
        assert K.get_batch_per_block(*a) == restate.get_batch_per_block(*a)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_softmax(K, dtype):
    torch.manual_seed(4)
    B, A, T, S = 2, 3, 9, 21
    x = torch.randn(B * A, T, S, device=DEV).to(dtype)
    bias = torch.randn(B * A, T, S, device=DEV).to(dtype)
    kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
    kpm[1, 15:] = True
    for causal in (False, True):
        if causal:
            xs, bs, kp = x[:, :, :T].contiguous(), bias[:, :, :T].contiguous(), kpm[:, :T].contiguous()
        else:
            xs, bs, kp = x, bias, kpm
        w = xs.float() * 0.3 + bs.float()
        if causal:
            w = w + torch.triu(torch.full((T, T), float("-inf"), device=DEV), 1)
        w = w.view(B, A, T, -1).masked_fill(kp[:, None, None, :], float("-inf")).view(B * A, T, -1)
        ref = torch.softmax(w, -1)
        p = K.attn_softmax(xs, bs, kp, 0.3, A, causal)
        assert rel(p, ref) < (1e-6 if dtype == torch.float32 else 1e-2)


def _attn_ref(q, k, v, heads, scale, bias, kpm, c_attn, causal):
    B, T, D = q.shape
    S = k.shape[1]
    hd = D // heads
    qh = q.view(B, T, heads, hd).transpose(1, 2)
    kh = k.view(B, S, heads, hd).transpose(1, 2)
    vh = v.view(B, S, heads, hd).transpose(1, 2)
    w = qh @ kh.transpose(-1, -2) * scale
    if bias is not None:
        w = w + bias.view(B, heads, T, S)
    if causal:
        w = w + torch.triu(torch.full((T, S), float("-inf"), device=q.device), 1)
    if kpm is not None:
        w = w.masked_fill(kpm[:, None, None, :], float("-inf"))
    p = torch.softmax(w, -1)
    o = p @ vh
    if c_attn is not None:
        o = o * c_attn.view(1, heads, 1, 1)
    return o.transpose(1, 2).reshape(B, T, D)


@pytest.mark.parametrize("B,heads,T,S,causal,use_bias,use_kpm", [
    (2, 4, 32, 32, False, False, False),
    (2, 4, 45, 45, True, True, True),
    (1, 12, 130, 130, False, True, True),
    (2, 4, 20, 77, False, True, True),     # cross attention
    (2, 3, 64, 267, False, False, True),
    (1, 2, 200, 200, True, False, False),
    (2, 2, 300, 131, True, True, True),     # causal with T != S, ragged tails in both
    (3, 2, 448, 448, False, False, True),
    (1, 1, 1, 3, False, False, False),
])
def test_fused_attention(K, B, heads, T, S, causal, use_bias, use_kpm):
    torch.manual_seed(5)
    D = heads * 64
    q = torch.randn(B, T, D, device=DEV).bfloat16()
    k = torch.randn(B, S, D, device=DEV).bfloat16()
    v = torch.randn(B, S, D, device=DEV).bfloat16()
    bias = torch.randn(B * heads, T, S, device=DEV).bfloat16() if use_bias else None
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        kpm[-1, S - 5:] = True
    c = (1 + 0.2 * torch.randn(heads, device=DEV)).float()
    scale = (64 * 2) ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    br = bias.float().requires_grad_(True) if use_bias else None
    cr = c.clone().requires_grad_(True)
    ref = _attn_ref(qr, kr, vr, heads, scale, br, kpm, cr, causal)
    dout = torch.randn(B, T, D, device=DEV).bfloat16()
    ref.backward(dout.float())
    out, lse = K.attn_fwd(q, k, v, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal)
    assert rel(out, ref) < 2e-2
    dq, dk, dv, dbias, delta = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm, c_attn=c,
                                           causal=causal, need_dbias=use_bias)
    assert rel(dq, qr.grad) < 3e-2
    assert rel(dk, kr.grad) < 3e-2
    assert rel(dv, vr.grad) < 3e-2
    if use_bias:
        assert rel(dbias, br.grad) < 3e-2
    # dc_attn[h] = sum(delta[:, h, :T]) / c[h]
    dc = delta.view(B, heads, -1)[:, :, :T].sum((0, 2)) / c
    assert rel(dc, cr.grad) < 3e-2
    # delta = rowsum(dO * O): written by the dQ kernel; the two-call form (ofa_attn_bwd_prep, then ofa_attn_bwd with out == NULL)
    # gives the same rows and the same gradients
    want = (dout.float() * out.float()).view(B, T, heads, 64).sum(-1).permute(0, 2, 1)
    assert rel(delta.view(B, heads, -1)[:, :, :T], want) < 1e-5
    if not use_bias:
        from ofasys_amd.lib import lib, ptr, stream
        Tp = K.pad32(T)
        d2 = torch.empty(B * heads, Tp, device=DEV)
        lib().call("ofa_attn_bwd_prep", ptr(dout), ptr(out), ptr(d2), B, heads, T, Tp, D, 1, stream())
        assert rel(d2.view(B, heads, -1)[:, :, :T], want) < 1e-5
        dq2, dk2, dv2 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        lib().call("ofa_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(dout), None, ptr(kpm.view(torch.uint8)) if use_kpm else None, ptr(c), 0,
                   ptr(lse), ptr(d2), None, ptr(dq2), ptr(dk2), ptr(dv2), None, B, heads, T, S, Tp, D, D, D, scale, int(causal), None, 0,
                   0, 1, stream())
        assert rel(dq2, dq) < 1e-3 and rel(dk2, dk) < 1e-3 and rel(dv2, dv) < 1e-3


def _cs_bufs(K, B, T, S, heads, seg=None, ld_extra=0):
    D = heads * 64
    nq, nk = K.attn_cs_slots(B, T, seg), K.attn_cs_slots(B, S, seg, k_side=True)
    wq = torch.full((nq, D + ld_extra), float("nan"), device=DEV)          # (every partial row must be written: NaN would survive)
    wkv = torch.full((nk, 2 * D + ld_extra), float("nan"), device=DEV)
    wc = torch.full((nq, heads), float("nan"), device=DEV)
    return dict(q=wq[:, :D], k=wkv[:, :D], v=wkv[:, D:2 * D], c=wc)


def _cs_check(cs, dq, dk, dv, delta_rows, c):
    """The partial rows sum to the column sums of the stored tensors up to their 16-bit rounding noise (the kernels sum the fp32 values
    before rounding: |error| ~ 2^-9 |x| sqrt(rows) per column) and to sum(delta) / c."""
    for name, g in (("q", dq), ("k", dk), ("v", dv)):
        g2 = g.float().reshape(-1, g.shape[-1])
        want = g2.sum(0)
        got = cs[name].sum(0)
        assert torch.isfinite(got).all(), name
        noise = 2.0 ** -9 * g2.pow(2).sum(0).sqrt()                    # one sigma of the summed rounding errors, per column
        assert ((got - want).abs() <= 4 * noise + 1e-3 * want.abs().max()).all(), name
    want_c = delta_rows / c
    got_c = cs["c"].sum(0)
    assert (got_c - want_c).abs().max() <= 1e-3 * want_c.abs().max().clamp_min(1.0)


@pytest.mark.parametrize("B,heads,T,S,causal,use_kpm,shared", [(2, 4, 200, 200, False, False, False), (3, 2, 130, 77, False, True, False),
                                                             (2, 2, 96, 96, True, False, False), (2, 3, 257, 300, False, False, True),
                                                             (1, 12, 448, 448, False, False, False)])
def test_attention_backward_column_sums(K, B, heads, T, S, causal, use_kpm, shared):
    """ofa_attn_bwd_cs / ofa_attn_sbias_bwd_cs: the bias gradients of the q / k / v projections and the c_attn gradient as partial rows
    out of the backward kernels' epilogues; dq / dk / dv themselves are bit-identical to the plain call."""
    torch.manual_seed(11)
    D = heads * 64
    q, k, v = (torch.randn(B, n, D, device=DEV).bfloat16() for n in (T, S, S))
    dout = torch.randn(B, T, D, device=DEV).bfloat16()
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        kpm[-1, S - 9:] = True
    bias = (0.5 * torch.randn(heads, T, S, device=DEV)).bfloat16() if shared else None
    c = (1 + 0.2 * torch.randn(heads, device=DEV)).float()
    kw = dict(bias=bias, kpm=kpm, c_attn=c, causal=causal, bias_shared=shared)
    out, lse = K.attn_fwd(q, k, v, heads, 0.125, **kw)
    ref = K.attn_bwd(q, k, v, out, dout, lse, heads, 0.125, **kw)
    cs = _cs_bufs(K, B, T, S, heads, ld_extra=8)
    got = K.attn_bwd(q, k, v, out, dout, lse, heads, 0.125, cs=cs, **kw)
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a, b)
    delta_rows = got[4].view(B, heads, -1)[:, :, :T].sum((0, 2))
    _cs_check(cs, got[0], got[1], got[2], delta_rows, c)


def test_attention_backward_column_sums_ragged(K):
    """... in ragged mode: tiles beyond a sample's length write zero rows, filler rows count as zeros."""
    from ofasys_amd.packing import Segments
    heads, D = 4, 256
    qlens, klens = [150, 24, 300, 1], [40, 260, 129, 7]
    qo = [0, 160, 192, 512]; ko = [0, 64, 352, 512]
    Rq, Rk = 576, 576
    table = torch.tensor([[qo[i], qlens[i], ko[i], klens[i]] for i in range(4)], dtype=torch.int32, device=DEV)
    seg = Segments(table, 4, Rq, Rk, max(qlens), max(klens))
    g = torch.Generator().manual_seed(5)
    q, do = (torch.randn(1, Rq, D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(2))
    k, v = (torch.randn(1, Rk, D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(2))
    c = (1 + 0.2 * torch.randn(heads, generator=g)).to(DEV)
    out, lse = K.attn_fwd(q, k, v, heads, 0.125, seg=seg, c_attn=c)
    ref = K.attn_bwd(q, k, v, out, do, lse, heads, 0.125, seg=seg, c_attn=c)
    cs = _cs_bufs(K, 1, Rq, Rk, heads, seg=seg)
    assert cs["q"].shape[0] == 4 * 3 * 4 and cs["k"].shape[0] == 4 * 3 * 4           # (sample, tile, wave)
    got = K.attn_bwd(q, k, v, out, do, lse, heads, 0.125, seg=seg, c_attn=c, cs=cs)
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a, b)
    delta = got[4].view(heads, -1)
    rows = torch.cat([torch.arange(qo[i], qo[i] + qlens[i]) for i in range(4)]).to(DEV)
    _cs_check(cs, got[0], got[1], got[2], delta[:, rows].sum(1), c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_elementwise(K, dtype):
    torch.manual_seed(6)
    x = torch.randn(1000, 264, device=DEV).to(dtype)
    dy = torch.randn_like(x)
    xr = x.float().requires_grad_(True)
    yr = F.gelu(xr)
    yr.backward(dy.float())
    t = tol(dtype)
    assert rel(K.gelu_fwd(x), yr) < t
    assert rel(K.gelu_bwd(dy, x), xr.grad) < t
    # dropout: keep-rate statistics, scaling, determinism in (seed, offset), backward uses the same mask
    res = torch.randn_like(x)
    y = K.dropout_add(x, res, 0.1, 1234, 77)
    y2 = K.dropout_add(x, res, 0.1, 1234, 77)
    assert torch.equal(y, y2)
    ones = torch.ones_like(x)
    kept = K.dropout_add(ones, None, 0.1, 1234, 77).float() > 0       # the mask depends on (seed, offset, index) only
    rate = kept.float().mean().item()
    assert abs(rate - 0.9) < 0.01
    want = torch.where(kept, x.float() / 0.9, torch.zeros_like(x.float())) + res.float()
    assert rel(y, want) < (1e-6 if dtype == torch.float32 else 2e-2)
    g = K.dropout_bwd(dy, 0.1, 1234, 77)
    assert rel(g, torch.where(kept, dy.float() / 0.9, torch.zeros_like(dy.float()))) < (1e-6 if dtype == torch.float32 else 1e-2)
    y3 = K.dropout_add(x, res, 0.1, 1234, 78)
    assert not torch.equal(y, y3)
    base = torch.tensor([7], dtype=torch.int64, device=DEV)          # device-side stream position: 70 + 7 == 77
    assert torch.equal(K.dropout_add(x, res, 0.1, 1234, 70, base), y)
    assert torch.equal(K.dropout_bwd(dy, 0.1, 1234, 70, base), g)
    # p = 0 is the identity + residual
    assert rel(K.dropout_add(x, res, 0.0, 1, 0), x.float() + res.float()) < t
    # add + row vector + row mask
    vec = torch.randn(264, device=DEV).to(dtype)
    mask = torch.rand(1000, device=DEV) > 0.7
    ref = (x.float() + res.float() + vec.float()) * (~mask).float()[:, None]
    assert rel(K.add_rowvec_mask(x, res, vec, mask), ref) < t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding(K, dtype):
    torch.manual_seed(7)
    V, D = 500, 256
    w = torch.randn(V, D, device=DEV).to(dtype)
    ids = torch.randint(0, V, (7, 33), device=DEV)
    ids[0, :10] = 1
    ids[3, 5:9] = 42
    out = K.embedding_fwd(w, ids)
    assert torch.equal(out, w[ids])
    dout = torch.randn(7, 33, D, device=DEV).to(dtype)
    dw = K.embedding_bwd(dout, ids, V, padding_idx=1)
    wr = w.float().requires_grad_(True)
    F.embedding(ids, wr, padding_idx=1).backward(dout.float())
    assert rel(dw, wr.grad) < (1e-6 if dtype == torch.float32 else 1e-2)
    assert float(dw[1].float().abs().max()) == 0.0
    dw2 = K.embedding_bwd(dout, ids, V, padding_idx=1)
    assert torch.equal(dw, dw2)      # deterministic


@pytest.mark.parametrize("V,n,D", [(2, 5000, 768), (1026, 6112, 768), (51265, 6112, 768), (9000, 300, 264), (511, 40000, 12), (70, 900, 4)])
def test_embedding_bwd_tables(K, V, n, D):
    """Vocabulary-sized tables (few hits per row, presence flags) and tiny ones (token types / positions: thousands of
    hits per row, reduced in slices); accumulates into an existing gradient."""
    torch.manual_seed(17)
    ids = torch.randint(0, V, (n,), device=DEV)
    dout = torch.randn(n, D, device=DEV).bfloat16()
    base = torch.randn(V, D, device=DEV).bfloat16()
    dw = K.embedding_bwd(dout, ids, V, padding_idx=None, dweight=base.clone())
    ref = base.float().index_add(0, ids, dout.float())
    assert rel(dw, ref) < 2e-2
    assert torch.equal(dw, K.embedding_bwd(dout, ids, V, padding_idx=None, dweight=base.clone()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_embed_im2col(K, dtype):
    torch.manual_seed(8)
    img = torch.randn(2, 3, 28, 42, device=DEV).to(dtype)
    w = torch.randn(64, 3, 14, 14, device=DEV).to(dtype)
    Kp = 592
    col = K.im2col_patch(img, 14, Kp)
    assert col.shape == (2 * 2 * 3, Kp) and float(col[:, 588:].float().abs().max()) == 0.0
    wp = torch.zeros(64, Kp, device=DEV, dtype=dtype)
    wp[:, :588] = w.view(64, -1)
    out = K.gemm(col, wp, False, True)
    ref = F.conv2d(img.float(), w.float(), stride=14).flatten(2).transpose(1, 2).reshape(-1, 64)
    assert rel(out, ref) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [204, 1001, 51265])
def test_cross_entropy(K, dtype, V):
    torch.manual_seed(9)
    rows = 37
    ld = (V + 7) // 8 * 8
    store = torch.randn(rows, ld, device=DEV).to(dtype) * 3
    logits = store[:, :V]
    target = torch.randint(0, V, (rows,), device=DEV)
    target[::5] = 1
    lr = logits.float().requires_grad_(True)
    loss_ref = F.nll_loss(F.log_softmax(lr, -1), target, ignore_index=1, reduction="sum")
    loss_ref.backward()
    lse, row_loss = K.cross_entropy_fwd(store, target, V, 1)
    assert rel(row_loss.sum(), loss_ref.detach()) < 1e-5
    gs = torch.tensor([1.0], device=DEV)
    d = K.cross_entropy_bwd(store, target, lse, gs, V, 1)
    assert rel(d[:, :V], lr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    if ld > V:
        assert float(d[:, V:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V,ld", [(204, 208), (1001, 1024), (8192, 8192), (20011, 20032), (51265, 51328), (65536, 65536)])
def test_cross_entropy_forward_and_gradient_in_one_pass(K, dtype, V, ld):
    """ofa_cross_entropy_fwd_grad against the two-kernel route: lse to the last bits (a 1024- instead of a 256-wide reduction tree), the
    same row losses, the same gradient up to one rounding of the 16-bit output; ignored rows and padding columns are zeros."""
    torch.manual_seed(9 + V)
    rows = 41
    store = (torch.randn(rows, ld, device=DEV) * 3).to(dtype)
    target = torch.randint(0, V, (rows,), device=DEV)
    target[::5] = 1
    gs = torch.tensor([0.37], device=DEV)
    assert K.cross_entropy_fwd_grad_ok(store[:, :V], V)
    lse, row_loss = K.cross_entropy_fwd(store, target, V, 1)
    d = K.cross_entropy_bwd(store, target, lse, gs, V, 1)
    lse2, row_loss2, d2 = K.cross_entropy_fwd_grad(store[:, :V], target, gs, V, 1)
    assert (lse2 - lse).abs().max() <= 4e-6 * lse.abs().max()
    assert (row_loss2 - row_loss).abs().max() <= 4e-6 * lse.abs().max()
    assert float(row_loss2[::5].abs().max()) == 0.0 and float(d2[::5].float().abs().max()) == 0.0
    assert rel(d2, d) < 1e-2 and (d2.float() - d.float()).abs().max() <= 2.0 ** -7 * d.float().abs().max()
    if ld > V:
        assert float(d2[:, V:].float().abs().max()) == 0.0
    assert not K.cross_entropy_fwd_grad_ok(torch.empty(2, 65544, device=DEV, dtype=dtype), 65540)       # longer than a block's registers
    assert not K.cross_entropy_fwd_grad_ok(torch.empty(2, 64, device=DEV), 64)                           # fp32: the two-kernel route


def test_cross_entropy_function_hands_back_the_forward_gradient_only_for_the_promised_seed():
    from ofasys_amd import ops
    torch.manual_seed(3)
    V, rows = 1001, 19
    store = (torch.randn(rows, 1008, device=DEV) * 2).bfloat16()
    target = torch.randint(0, V, (rows,), device=DEV)
    target[3] = 1
    seed = torch.ones((), device=DEV)
    grads = []
    for mode in ("plain", "seeded", "other"):
        x = store[:, :V].clone().requires_grad_(True)       # (a dense [rows, V] tensor: the function pads it itself)
        loss = ops.cross_entropy_sum(x, target, 1, seed=None if mode == "plain" else seed)
        loss.backward(seed if mode != "other" else torch.full((), 2.0, device=DEV))
        grads.append((float(loss), x.grad.float()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-5 * abs(grads[0][0]) and abs(grads[0][0] - grads[2][0]) <= 1e-5 * abs(grads[0][0])
    assert rel(grads[1][1], grads[0][1]) < 1e-2
    assert rel(grads[2][1], 2 * grads[0][1]) < 1e-2          # seeded with something else after all: the fallback scales correctly


def test_adam_and_sumsq(K):
    torch.manual_seed(10)
    n = 100003
    p0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV).bfloat16()
    # restatement of engine/optim/adam.py:192-212 (eps is added to sqrt(v) BEFORE the bias correction, unlike torch.optim.Adam)
    ref_p, rm, rv = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    master = p0.clone()
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    model = p0.bfloat16()
    coef = torch.tensor([0.5], device=DEV)
    for step in (1, 2, 3):
        gg = g.float() * 0.5
        rm.mul_(0.9).add_(gg, alpha=0.1)
        rv.mul_(0.999).addcmul_(gg, gg, value=0.001)
        step_size = 1e-3 * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        ref_p.add_(ref_p, alpha=-0.01 * 1e-3)
        ref_p.addcdiv_(rm, rv.sqrt().add_(1e-8), value=-step_size)
        K.adam_step(master, m, v, g, model, coef, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
    assert rel(master, ref_p) < 1e-6
    assert torch.equal(model, master.bfloat16())
    out = torch.zeros(1, device=DEV)
    K.sumsq(g, out)
    K.sumsq(master, out)
    assert rel(out, (g.float() ** 2).sum() + (master ** 2).sum()) < 1e-5


def test_deferred_folds_match_immediate_reductions(K):
    """FoldQueue (csrc/fold.hip): LayerNorm dgamma/dbeta/dbias, bias column sums and split-K weight gradients left as fp32
    partial rows and reduced in ONE batched launch must equal the per-call reductions."""
    torch.manual_seed(21)
    dt = torch.bfloat16
    rows, cols = 5000, 768
    x = torch.randn(rows, cols, device=DEV).to(dt)
    dy = torch.randn(rows, cols, device=DEV).to(dt)
    g = (1 + 0.1 * torch.randn(cols, device=DEV)).to(dt)
    b = torch.zeros(cols, device=DEV, dtype=dt)
    _, mean, rstd = K.layernorm_fwd(x, g, b, 1e-5, fuse_gelu=True)
    a = torch.randn(4096, 768, device=DEV).to(dt)          # wgrad: dW[768, 520] += a^T @ c   (K = 4096 -> split-K)
    c = torch.randn(4096, 520, device=DEV).to(dt)

    def run(fold):
        outs = [torch.full((cols,), 0.5, device=DEV, dtype=dt) for _ in range(4)]
        gw = torch.full((768, 520), 0.25, device=DEV, dtype=dt)
        dx = K.layernorm_bwd(dy, x, g, mean, rstd, fuse_gelu=True, dgamma=outs[0], dbeta=outs[1], dbias=outs[2], fold=fold)[0]
        K.colsum(dy, alpha=0.5, out=outs[3], accumulate=True, fold=fold)
        K.gemm(a, c, True, False, alpha=2.0, out=gw, accumulate=True, fold=fold)
        if fold is not None:
            assert len(fold.jobs) == 5                      # 3 LayerNorm quantities + bias + one split-K product
            assert float((gw.float() - 0.25).abs().max()) == 0.0      # nothing folded yet
            fold.flush()
            assert not fold.jobs
        return [dx] + outs + [gw]

    want = run(None)
    got = run(K.FoldQueue())
    assert torch.equal(want[0], got[0])
    for w, o in zip(want[1:], got[1:]):
        assert rel(o, w.float()) < 1e-2
    ref = 0.25 + 2.0 * (a.float().t() @ c.float())
    assert rel(got[-1], ref) < 1e-2


# ---------------------------------------------------------------------------------------------- convolution stack
def _nhwc_rows(t):          # [B,C,H,W] -> [B*H*W, C]
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H,W,nchw", [
    (3, 64, 7, 2, 3, 30, 26, True),       # the stem convolution reads the NCHW image directly
    (64, 64, 1, 1, 0, 9, 7, False),       # 1x1: plain GEMM on the rows
    (64, 128, 3, 1, 1, 9, 7, False),
    (128, 128, 3, 2, 1, 10, 8, False),
    (256, 512, 1, 2, 0, 8, 8, False),     # strided 1x1 (downsample branch)
    (8, 16, 3, 2, 0, 11, 9, False),       # no padding (audio subsampling convs)
])
def test_conv2d(K, dtype, Cin, Cout, k, stride, pad, H, W, nchw):
    from ofasys_amd import ops
    torch.manual_seed(31)
    B = 2
    x = torch.randn(B, Cin, H, W, device=DEV).to(dtype)
    w = (0.1 * torch.randn(Cout, Cin, k, k, device=DEV)).to(dtype)
    b = torch.randn(Cout, device=DEV).to(dtype) if not nchw else None
    xr, wr = x.float().clone().requires_grad_(True), w.float().clone().requires_grad_(True)
    br = b.float().clone().requires_grad_(True) if b is not None else None
    ref = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    xin = (x if nchw else _nhwc_rows(x)).detach().clone().requires_grad_(not nchw)
    wp = w.clone().requires_grad_(True)
    bp = b.clone().requires_grad_(True) if b is not None else None
    y, Ho, Wo = ops.conv2d(xin, wp, bp, B, H, W, stride, pad, nchw)
    assert (Ho, Wo) == tuple(ref.shape[-2:])
    t = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(y, _nhwc_rows(ref)) < t
    y.backward(_nhwc_rows(dy).to(dtype))
    assert rel(wp.grad, wr.grad) < t
    if not nchw:
        assert rel(xin.grad, _nhwc_rows(xr.grad)) < t
    if b is not None:
        assert rel(bp.grad, br.grad) < t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training,relu,with_res", [(True, True, True), (True, False, False), (False, True, False), (True, True, False)])
# one and several row groups / column chunks of the statistics kernels, a channel count that is not a multiple of the 256-channel block
@pytest.mark.parametrize("shape", [(3, 64, 7, 5), (8, 256, 28, 28), (2, 1032, 20, 21), (2, 64, 96, 96)])
def test_batchnorm(K, dtype, training, relu, with_res, shape):
    from ofasys_amd import ops
    torch.manual_seed(32)
    B, C, H, W = shape
    x = (1.5 * torch.randn(B, C, H, W, device=DEV) + 0.3).to(dtype)
    res = torch.randn(B, C, H, W, device=DEV).to(dtype) if with_res else None
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * torch.randn(C)); bn.bias.copy_(0.1 * torch.randn(C))
        bn.running_mean.copy_(0.1 * torch.randn(C)); bn.running_var.copy_(1 + 0.1 * torch.rand(C))
    import copy
    bnr = copy.deepcopy(bn).float()
    bn = bn.to(dtype)
    with torch.no_grad():                          # same (rounded) parameters on both sides: a 0.4% gain difference would
        for pr, p in zip(bnr.parameters(), bn.parameters()):     # flip ReLU gates of near-zero outputs
            pr.copy_(p.float())
        for br_, b_ in zip(bnr.buffers(), bn.buffers()):
            br_.copy_(b_.to(br_.dtype))
    bn.train(training); bnr.train(training)
    xr = x.float().clone().requires_grad_(True)
    rr = res.float().clone().requires_grad_(True) if with_res else None
    yr = bnr(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    dy = torch.randn_like(yr)
    yr.backward(dy)
    xin = _nhwc_rows(x).requires_grad_(True)
    rin = _nhwc_rows(res).requires_grad_(True) if with_res else None
    y = ops.batch_norm(xin, bn, relu=relu, residual=rin)
    t = 2e-5 if dtype == torch.float32 else 3e-2
    assert rel(y, _nhwc_rows(yr)) < t
    y.backward(_nhwc_rows(dy).to(dtype))
    assert rel(xin.grad, _nhwc_rows(xr.grad)) < 2 * t
    assert rel(bn.weight.grad, bnr.weight.grad) < 2 * t
    assert rel(bn.bias.grad, bnr.bias.grad) < 2 * t
    if with_res:
        assert rel(rin.grad, _nhwc_rows(rr.grad)) < 2 * t
    assert rel(bn.running_mean, bnr.running_mean) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert rel(bn.running_var, bnr.running_var) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert int(bn.num_batches_tracked) == int(bnr.num_batches_tracked)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_relu(K, dtype):
    from ofasys_amd import ops
    torch.manual_seed(33)
    B, C, H, W = 2, 16, 13, 10
    x = torch.randn(B, C, H, W, device=DEV).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    xin = _nhwc_rows(x).requires_grad_(True)
    y, Ho, Wo = ops.max_pool(xin, B, H, W, 3, 2, 1)
    assert (Ho, Wo) == tuple(ref.shape[-2:]) and torch.equal(y.float(), _nhwc_rows(ref))
    y.backward(_nhwc_rows(dy).to(dtype))
    assert rel(xin.grad, _nhwc_rows(xr.grad)) < (1e-6 if dtype == torch.float32 else 1e-2)
    z = x.clone().requires_grad_(True)
    r = ops.relu(z)
    assert torch.equal(r, F.relu(x))
    r.backward(torch.ones_like(r))
    assert torch.equal(z.grad, (x > 0).to(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["plain", "range", "range_mask", "drop"])
def test_label_smoothed_cross_entropy(K, dtype, name):
    """ops.label_smoothed_cross_entropy (fused log-softmax + smoothing + constraint masks + drop_worst) against golden
    vectors produced by the reference's label_smoothed_nll_loss (tests/golden/ls_cross_entropy.npz)."""
    from ofasys_amd import ops
    from tests.golden_util import load_golden
    g = load_golden("ls_cross_entropy")
    eps, cs, ce, dw = [float(v) for v in g[name + ".cfg"]]
    x = torch.from_numpy(g["logits"]).to(DEV).to(dtype).requires_grad_(True)
    tg = torch.from_numpy(g[name + ".target"]).to(DEV)
    crange = None if cs < 0 else (int(cs), int(ce))
    sm = torch.from_numpy(g[name + ".sample_mask"]).bool().to(DEV) if (name + ".sample_mask") in g else None
    loss, nll, ntok = ops.label_smoothed_cross_entropy(x, tg, 1, eps, crange, sm, dw)
    loss.backward()
    t = 1e-5 if dtype == torch.float32 else 2e-2
    assert int(ntok) == int(g[name + ".ntokens"][0])
    assert abs(float(loss) - float(g[name + ".loss"][0])) <= t * abs(float(g[name + ".loss"][0]))
    assert abs(float(nll) - float(g[name + ".nll"][0])) <= t * abs(float(g[name + ".nll"][0]))
    assert rel(x.grad, torch.from_numpy(g[name + ".dlogits"]).to(DEV)) < (1e-5 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols,has_a,has_b,p", [(300, 768, True, True, 0.1), (70, 256, False, True, 0.1),
                                                      (129, 768, True, False, 0.0), (64, 1024, True, True, 0.3),
                                                      (33, 512, False, False, 0.1), (5000, 768, False, True, 0.1),
                                                      (2051, 1024, True, False, 0.1), (17, 1280, True, True, 0.1),
                                                      (40, 640, True, True, 0.2)])
def test_residual_join_equals_unfused_chain(K, dtype, rows, cols, has_a, has_b, p):
    """csrc/join.hip: y = residual + dropout(LN_a(x)), z = LN_b(y) and its backward must equal the op-by-op kernels
    (LayerNorm, dropout+add, LayerNorm-with-residual-gradient): y bit for bit without LN_a (same Philox positions, same rounding
    points; with LN_a the row statistics are summed in another lane grouping, which may move a rounding on a few elements in 10^5),
    everything else to rounding noise.  The backward reads the keep bits the forward left (16-bit rows of 256 k columns)."""
    from ofasys_amd import ops
    torch.manual_seed(41)
    x = torch.randn(rows, cols, device=DEV).to(dtype)
    r = torch.randn(rows, cols, device=DEV).to(dtype)
    lna = torch.nn.LayerNorm(cols).to(DEV).to(dtype) if has_a else None
    lnb = torch.nn.LayerNorm(cols).to(DEV).to(dtype) if has_b else None
    for ln in (lna, lnb):
        if ln is not None:
            with torch.no_grad():
                ln.weight.copy_(1 + 0.1 * torch.randn(cols)); ln.bias.copy_(0.1 * torch.randn(cols))
    dy = torch.randn(rows, cols, device=DEV).to(dtype)
    dz = torch.randn(rows, cols, device=DEV).to(dtype)

    def run(fused):
        for ln in (lna, lnb):
            if ln is not None:
                ln.zero_grad()
        xx, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        ops.manual_seed(99)
        if fused:
            y, z = ops.residual_join(xx, rr, lna, p, True, lnb)
        else:
            h = ops.layer_norm(xx, lna.weight, lna.bias, lna.eps) if has_a else xx
            y = ops.dropout_add(h, rr, p, True)
            z = None
            if has_b:
                y, z = ops.layer_norm_fork(y, lnb.weight, lnb.bias, lnb.eps)
        outs = [y] + ([z] if z is not None else [])
        torch.autograd.backward(outs, [dy] + ([dz] if z is not None else []))
        g = [xx.grad, rr.grad] + [t.grad for ln in (lna, lnb) if ln is not None for t in (ln.weight, ln.bias)]
        return [y.detach()] + ([z.detach()] if z is not None else []) + g

    a, b = run(True), run(False)
    assert len(a) == len(b)
    if has_a and dtype != torch.float32:
        assert float((a[0] != b[0]).float().mean()) < 1e-3    # y: the same dropout mask; LN_a statistics in another summation order
    else:
        assert torch.equal(a[0], b[0])                        # y: same Philox positions, same rounding points -> bit-exact
    for i, (u, v) in enumerate(zip(a, b)):                    # the rest: same maths, different summation grouping / FMA
        assert rel(u, v.float()) < (2e-5 if dtype == torch.float32 else 2e-2), i      # contraction (rows split over waves)


def test_residual_join_row_per_wave_forms_against_split_row_forms():
    """tools/join_bench.py check: the row-per-wave residual-join kernels (every waves-per-block / rows-in-flight form of the backward, keep
    bits read back from the forward) against the split-row kernels on the same inputs (debug library: OFA_JOIN_FWD / OFA_JOIN_BWD): y equal
    bit for bit without LN_a, everything else to fp32 rounding noise; row counts of 1, a partial block, several sweeps of the grid."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "join_bench.py"), "check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 mismatching cases" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,S,cap,use_bias,use_kpm,use_c", [
    (3, 4, 1, 64, False, False, False),          # the first decoding step: one key
    (2, 12, 7, 64, True, False, True),
    (5, 4, 64, 64, True, True, True),            # cache exactly full
    (2, 12, 449, 512, False, True, True),        # encoder-decoder cache of the cfg-2 source length + 1
    (1, 16, 1500, 2048, True, True, False),
])
def test_attn_decode_matches_reference(K, B, H, S, cap, use_bias, use_kpm, use_c, dtype):
    """csrc/attention_decode.hip: one query row per (batch, head) against the first S rows of a [B, capacity, D] cache --
    softmax(q.k*scale + bias, masked) @ v * c_attn, with the probabilities as a second output."""
    torch.manual_seed(S)
    D = H * 64
    q = torch.randn(B, D, device=DEV).to(dtype)
    kc = torch.randn(B, cap, D, device=DEV).to(dtype)
    vc = torch.randn(B, cap, D, device=DEV).to(dtype)
    bias = (torch.randn(B * H, S, device=DEV)).to(dtype) if use_bias else None
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, cap, dtype=torch.bool, device=DEV)
        kpm[:, S // 2:S // 2 + max(S // 5, 0)] = True
        kpm[0, S - 1:] = S > 1                                            # row 0: last key padded too (never all of them)
    c = (torch.rand(H, device=DEV) + 0.5).to(dtype) if use_c else None
    scale = 0.0884
    out, probs = K.attn_decode(q, kc, vc, S, H, scale, bias=bias, kpm=kpm, c_attn=c, need_probs=True)
    qf = q.float().view(B, H, 1, 64)
    kf = kc[:, :S].float().view(B, S, H, 64).transpose(1, 2)
    vf = vc[:, :S].float().view(B, S, H, 64).transpose(1, 2)
    w = (qf @ kf.transpose(2, 3)) * scale
    if bias is not None:
        w = w + bias.float().view(B, H, 1, S)
    if kpm is not None:
        w = w.masked_fill(kpm[:, :S].view(B, 1, 1, S), float("-inf"))
    p = torch.softmax(w, dim=-1)
    o = p @ vf
    if c is not None:
        o = o * c.float().view(1, H, 1, 1)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert rel(out.float(), o.transpose(1, 2).reshape(B, D)) < tol
    assert rel(probs.float(), p.reshape(B * H, S)) < tol
    # a strided cache view (ld > D) reads the same rows
    big = torch.zeros(B, cap, D + 64, device=DEV, dtype=dtype)
    big[:, :, :D] = kc
    big2 = torch.zeros(B, cap, D + 64, device=DEV, dtype=dtype)
    big2[:, :, :D] = vc
    out2, _ = K.attn_decode(q, big[:, :, :D], big2[:, :, :D], S, H, scale, bias=bias, kpm=kpm, c_attn=c)
    assert torch.equal(out2, out)


def test_attn_decode_fully_masked_row_is_zero(K):
    q = torch.randn(1, 64, device=DEV)
    kc, vc = torch.randn(1, 64, 64, device=DEV), torch.randn(1, 64, 64, device=DEV)
    kpm = torch.ones(1, 64, dtype=torch.bool, device=DEV)
    out, probs = K.attn_decode(q, kc, vc, 5, 1, 1.0, kpm=kpm, need_probs=True)
    assert float(out.abs().max()) == 0.0 and float(probs.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["tbc_view", "tbc_contig", "btc"])
def test_drop_path_per_sample(K, dtype, layout):
    """module/droppath.py:40-60: every sample is either dropped entirely or scaled by 1/keep; the gradient takes the same
    factors; identity in eval mode."""
    from ofasys_amd.module.layers import DropPath
    torch.manual_seed(3)
    B, T, C = 64, 5, 256
    base = torch.randn(B, T, C, device=DEV).to(dtype)
    if layout == "tbc_view":
        x, axis = base.transpose(0, 1), 1                     # [T,B,C] view of batch-major storage (the model's layout)
    elif layout == "tbc_contig":
        x, axis = base.transpose(0, 1).contiguous(), 1
    else:
        x, axis = base, 0
    x = x.detach().requires_grad_(True)
    m = DropPath(0.25, batch_axis=axis).train()
    y = m(x)
    assert y.shape == x.shape
    yb = (y if axis == 0 else y.transpose(0, 1)).float()
    xb = (x if axis == 0 else x.transpose(0, 1)).detach().float()
    kept = 0
    for b in range(B):
        if float(yb[b].abs().max()) == 0.0:
            continue
        kept += 1
        assert rel(yb[b], xb[b] / 0.75) < tol(dtype)
    assert 30 <= kept <= 60                                   # keep = 0.75 of 64 samples
    g = torch.randn_like(y)
    (dx,) = torch.autograd.grad(y, x, g)
    gb = (g if axis == 0 else g.transpose(0, 1)).float()
    dxb = (dx if axis == 0 else dx.transpose(0, 1)).float()
    for b in range(B):
        want = gb[b] / 0.75 if float(yb[b].abs().max()) > 0 else torch.zeros_like(gb[b])
        assert float((dxb[b] - want).abs().max()) <= tol(dtype) * float(gb[b].abs().max())
    assert m.eval()(x) is x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c_attn_grad_kernel(K, dtype):
    """ofa_c_attn_grad: dc[h] (+)= sum_{b, t < T} delta[b*heads+h, t] / c[h]; the padded tail of each delta row is ignored."""
    torch.manual_seed(5)
    B, H, T, ld = 7, 12, 45, 64
    delta = torch.randn(B * H, ld, device=DEV)
    c = (torch.rand(H, device=DEV) + 0.5).to(dtype)
    want = delta.view(B, H, ld)[:, :, :T].double().sum(dim=(0, 2)) / c.double()
    got = K.c_attn_grad(delta, c, B, H, T)
    assert got.dtype == dtype and rel(got.float(), want.float()) < (1e-5 if dtype == torch.float32 else 8e-3)
    acc = torch.randn(H, device=DEV).to(dtype)
    base = acc.double().clone()
    K.c_attn_grad(delta, c, B, H, T, out=acc, accumulate=True)
    assert rel(acc.float(), (base + want).float()) < (1e-5 if dtype == torch.float32 else 1.6e-2)


@pytest.mark.parametrize("clip", [0.0, 1.0, 100.0])
def test_step_schedule_kernel(K, clip):
    """ofa_step_schedule against the host arithmetic of the reference (engine/trainer.py:857-884, optim/adam.py:205-207)."""
    gsq = torch.tensor([37.5], device=DEV)
    stats = torch.tensor([24.0, 3.0, 24.0], dtype=torch.float64, device=DEV)
    step = torch.tensor([4.0], dtype=torch.float64, device=DEV)
    lr = torch.tensor([3e-4], dtype=torch.float64, device=DEV)
    sched, gnorm = torch.zeros(5, device=DEV), torch.zeros(1, device=DEV)
    K.step_schedule(gsq, stats, step, lr, sched, gnorm, clip, 0.9, 0.999)
    gn = math.sqrt(37.5) / 24.0
    coef = (1 / 24.0) * (min(1.0, clip / (gn + 1e-6)) if clip > 0 else 1.0)
    t = 5.0
    want = [coef, 3e-4 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t), 3e-4]
    assert float(step) == 5.0 and abs(float(gnorm) - gn) < 1e-6 * gn
    for a, b in zip(sched.tolist(), want):
        assert abs(a - b) <= 1e-6 * abs(b), (sched.tolist(), want)
    assert sched[3:].tolist() == [0.0, 0.0]
    # guard: non-finite norm / empty batch -> skip flag, step counter untouched, running count of skipped updates
    for bad_gsq, bad_n in ((float("inf"), 24.0), (float("nan"), 24.0), (37.5, 0.0)):
        K.step_schedule(torch.tensor([bad_gsq], device=DEV), torch.tensor([bad_n, 0.0, 0.0], dtype=torch.float64, device=DEV),
                        step, lr, sched, gnorm, clip, 0.9, 0.999)
        assert float(step) == 5.0 and sched[:2].tolist() == [0.0, 0.0] and float(sched[3]) == 1.0
    assert float(sched[4]) == 3.0
    master = torch.randn(1000, device=DEV)
    m, v, g = torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV), torch.randn(1000, device=DEV)
    w0 = master.clone()
    model = master.clone()
    K.adam_step(master, m, v, g, model, sched, 0.0, 0.9, 0.999, 1e-8, 0.01, 0)          # skip flag set: nothing moves
    assert torch.equal(master, w0) and float(m.abs().sum()) == 0.0 and float(v.abs().sum()) == 0.0


def test_copy_batched(K):
    """ofa_copy_batched: any number of (dst, src) pairs in one launch per 96 -- sizes from one byte to several chunks, unaligned
    storage offsets, a dtype-converting pair left to torch."""
    torch.manual_seed(5)
    pairs = []
    for i, n in enumerate([1, 3, 17, 4096, 32768 // 2 + 5, 100003, 7, 250000] * 14):          # 112 pairs: two launches
        base = torch.randn(n + 3, device=DEV).to(torch.bfloat16 if i % 2 else torch.float32)
        src = base[i % 3:i % 3 + n]                                                        # (2- / 4-byte steps off the 16-byte grid)
        dst = torch.zeros(n + 3, device=DEV, dtype=src.dtype)[(i + 1) % 3:(i + 1) % 3 + n]
        pairs.append((dst, src))
    i64 = torch.arange(1000, device=DEV)
    pairs.append((torch.zeros(1000, device=DEV, dtype=torch.int64), i64))
    conv = (torch.zeros(64, device=DEV, dtype=torch.bfloat16), torch.randn(64, device=DEV))   # dtypes differ: torch's copy_
    K.copy_batched(pairs + [conv])
    torch.cuda.synchronize()
    for dst, src in pairs:
        assert torch.equal(dst, src)
    assert torch.equal(conv[0], conv[1].to(torch.bfloat16))
