"""GPU: attention with a batch-SHARED position bias (ofa_attn_sbias_fwd / _bwd, csrc/attention.hip): the reference's dense [B*A, T, S]
bias (abs-pos + rel-pos, adaptor/general.py:223-282, model/transformer.py:280-299) is B copies of one [A, T, S] matrix, which the
kernels take once and index by (head, position, position) for every sample; its gradient -- the sum over the batch of dS -- comes from
a third backward kernel that walks the batch per tile.  Checked against a plain PyTorch fp32 reference that EXPANDS the bias over the
batch (autograd then reduces the expand: exactly the reference's arithmetic), incl. causal masks, key padding, ragged (segment) mode
with position-local indexing, and bitwise reproducibility.  Tolerances as tests/test_kernels_gpu.py::test_fused_attention."""
import pytest
import torch

from tests.test_kernels_gpu import _attn_ref, rel

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from ofasys_amd import kernels
    return kernels


def _mk(B, heads, T, S, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    D = heads * 64
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV).bfloat16()                  # noqa: E731
    return mk(B, T, D), mk(B, S, D), mk(B, S, D), mk(heads, T, S), mk(B, T, D)


@pytest.mark.parametrize("B,heads,T,S,causal,use_kpm", [
    (2, 4, 64, 64, False, False),
    (3, 2, 45, 45, True, True),              # decoder self-attention: causal, ragged tails, odd S (unaligned bias rows)
    (2, 12, 130, 130, False, True),
    (4, 4, 20, 77, False, True),             # cross attention
    (5, 2, 300, 300, False, True),
    (2, 3, 448, 448, False, False),          # the cfg-2b encoder shape (196 + 252)
    (2, 2, 64, 1600, False, True),           # long keys: the 128-key tile form of the batch-sum kernel
])
def test_shared_bias_attention_matches_reference(K, B, heads, T, S, causal, use_kpm):
    q, k, v, bias, dout = _mk(B, heads, T, S, 7 + T + S)
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        kpm[-1, S - 5:] = True
        kpm[0, S - 2:] = True
    c = (1 + 0.2 * torch.randn(heads, device=DEV)).float()
    scale = (64 * 2) ** -0.5
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    br = bias.float().requires_grad_(True)
    cr = c.clone().requires_grad_(True)
    ref = _attn_ref(qr, kr, vr, heads, scale, br.unsqueeze(0).expand(B, heads, T, S).reshape(B * heads, T, S), kpm, cr, causal)
    ref.backward(dout.float())
    out, lse = K.attn_fwd(q, k, v, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal, bias_shared=True)
    assert rel(out, ref) < 2e-2
    dq, dk, dv, G, delta = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal,
                                      need_dbias=True, bias_shared=True)
    assert rel(dq, qr.grad) < 3e-2 and rel(dk, kr.grad) < 3e-2 and rel(dv, vr.grad) < 3e-2
    assert G.dtype == torch.float32 and G.shape == (heads, T, S)
    assert rel(G, br.grad) < 3e-2                                               # sum over the batch of dS
    dc = delta.view(B, heads, -1)[:, :, :T].sum((0, 2)) / c
    assert rel(dc, cr.grad) < 3e-2
    # the same numbers as this build's dense-bias kernels fed B copies of the bias (up to the last bf16 digit: the shared form feeds
    # the bias to the score MFMAs as their initial accumulator, the dense one adds it afterwards); and bitwise reproducible
    dense = bias.unsqueeze(0).expand(B, heads, T, S).reshape(B * heads, T, S).contiguous()
    out_d, _ = K.attn_fwd(q, k, v, heads, scale, bias=dense, kpm=kpm, c_attn=c, causal=causal)
    assert rel(out_d, out) < 4e-3
    again = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal, need_dbias=True, bias_shared=True)
    assert torch.equal(again[3], G) and torch.equal(again[0], dq)
    # the gradient in the bias' own dtype (what autograd gets): the fp32 sum rounded once -- by the kernel (one chunk) or the chunk fold
    g16 = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal, need_dbias=True, bias_shared=True,
                     dbias_dtype=torch.bfloat16)[3]
    assert g16.dtype == torch.bfloat16 and torch.equal(g16, G.to(torch.bfloat16))


def test_shared_bias_over_ragged_segments(K):
    """Ragged mode: samples packed back to back, the bias indexed by the position INSIDE the sample -- equal to the padded call at
    every valid row (forward, dq / dk / dv, the batch-summed bias gradient), filler rows of the outputs exactly zero."""
    from ofasys_amd.packing import build_pack_plan
    B, heads, T = 4, 2, 96
    lens = [96, 41, 70, 9]
    q, k, v, bias, dout = _mk(B, heads, T, T, 21)
    scale = (64 * 2) ** -0.5
    kpm = torch.zeros(B, T, dtype=torch.bool)
    for b, n in enumerate(lens):
        kpm[b, n:] = True
    plan = build_pack_plan(kpm, kpm, bucket=64).to(DEV)
    assert plan.enc_prefix
    idx = plan.enc_index
    pack = lambda t: K.gather_rows(t.reshape(B * T, -1).contiguous(), idx).view(1, -1, t.shape[-1])      # noqa: E731
    c = (1 + 0.1 * torch.randn(heads, device=DEV)).float()
    # padded rows carry garbage in a padded run; zero dout there so that the two runs see the same gradient signal
    dout = dout.masked_fill(kpm.to(DEV).unsqueeze(-1), 0.0)
    for causal in (False, True):
        out, lse = K.attn_fwd(q, k, v, heads, scale, bias=bias, kpm=kpm.to(DEV), c_attn=c, causal=causal, bias_shared=True)
        g = K.attn_bwd(q, k, v, out, dout, lse, heads, scale, bias=bias, kpm=kpm.to(DEV), c_attn=c, causal=causal, need_dbias=True,
                       bias_shared=True)
        pout, plse = K.attn_fwd(pack(q), pack(k), pack(v), heads, scale, bias=bias, c_attn=c, causal=causal, seg=plan.enc_self,
                                bias_shared=True)
        pg = K.attn_bwd(pack(q), pack(k), pack(v), pout, pack(dout), plse, heads, scale, bias=bias, c_attn=c, causal=causal,
                        need_dbias=True, seg=plan.enc_self, bias_shared=True)
        rows = torch.nonzero(idx >= 0).squeeze(1)
        filler = torch.nonzero(idx < 0).squeeze(1)
        src = idx[rows]

        def same(packed, padded, name, tol=2e-2):
            pr = packed.reshape(-1, packed.shape[-1])
            assert rel(pr[rows], padded.reshape(B * T, -1)[src].float()) < tol, name
            assert float(pr[filler].float().abs().max()) == 0.0, name + " filler rows"
        same(pout, out, "out", 1e-2)
        same(pg[0], g[0], "dq")
        same(pg[1], g[1], "dk")
        same(pg[2], g[2], "dv")
        # the batch-summed bias gradient: padded query rows carry dO = 0 (so dS = 0), padded keys are masked: the two runs agree everywhere
        assert rel(pg[3], g[3]) < 3e-2


def test_shared_bias_through_autograd_functions():
    """ops.attention with a [A, T, S] bias: the fused path (shared kernels) and the exact tier (bias expanded over the batch, autograd
    reducing it) give the same gradients, incl. the bias'."""
    from ofasys_amd import ops
    B, heads, T = 3, 2, 48
    q, k, v, bias, dout = _mk(B, heads, T, T, 5)
    scale = (64 * 2) ** -0.5
    grads = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (q, k, v, bias)]
        if fused:
            out, _ = ops.attention(leaves[0], leaves[1], leaves[2], heads, scale, bias=leaves[3])
        else:
            out, _ = ops.attention(leaves[0].float(), leaves[1].float(), leaves[2].float(), heads, scale, bias=leaves[3].float())
        (out.float() * dout.float()).sum().backward()
        grads.append([t.grad.float() for t in leaves])
    for a, b, name in zip(grads[0], grads[1], ("dq", "dk", "dv", "dbias")):
        assert rel(a, b) < 4e-2, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,V,D", [(252, 511, 12), (196, 6892, 12), (37, 511, 4), (64, 2047, 16), (20, 30, 40)])
def test_planned_table_gradient_matches_scan_kernel_and_torch(K, dtype, n, V, D):
    """ops.embedding(..., plan_key=): the rel-pos table lookup `table[bucket[i][j]]` whose gradient uses a cached sort of the ids
    (ofa_segment_rowsum) instead of the scan kernel -- same sums as torch's index_add in fp32 and as the scan kernel, reproducible,
    also when accumulating into an existing gradient."""
    from ofasys_amd import ops
    g = torch.Generator(device="cpu").manual_seed(n + V)
    ids = torch.randint(0, V, (n, n), generator=g).to(DEV)
    ids[0, 0], ids[-1, -1] = 0, V - 1
    w = torch.randn(V, D, generator=g).to(DEV).to(dtype)
    dout = torch.randn(n, n, D, generator=g).to(DEV).to(dtype)
    want = torch.zeros(V, D, device=DEV).index_add_(0, ids.reshape(-1), dout.reshape(-1, D).float())
    res = []
    for key in (("test", n, V, D, str(dtype)), None):
        wl = w.clone().requires_grad_(True)
        out = ops.embedding(ids, wl, plan_key=key)
        assert torch.equal(out, w[ids])
        out.backward(dout)
        res.append(wl.grad.float())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert rel(res[0], want) < tol and rel(res[1], want) < tol
    plan = ops.SegmentPlan.get((("test", n, V, D, str(dtype)), tuple(ids.shape), str(ids.device)), ids)
    acc = torch.ones(V, D, device=DEV, dtype=dtype)
    K.segment_rowsum(dout.reshape(-1, D).contiguous(), plan, acc, True)
    assert rel(acc.float(), want + 1.0) < tol
    again = torch.zeros(V, D, device=DEV, dtype=dtype)
    K.segment_rowsum(dout.reshape(-1, D).contiguous(), plan, again, False)
    assert torch.equal(again.float(), res[0])


@pytest.mark.parametrize("A,T,S,slots", [(12, 70, 70, ((0, 30), (30, 40))), (4, 64, 64, ((0, 64),)), (3, 45, 131, ()), (16, 33, 33, ((5, 20),)), (40, 40, 40, ((8, 32),))])   # (40 heads: two launches of <= 24)
@pytest.mark.parametrize("with_abs", [True, False])
def test_bias_build_assembles_and_swizzles(K, A, T, S, slots, with_abs):
    """ofa_bias_build: the row-major result is general.py:265-280's clone + diagonal block adds (bit-exact: one bf16 rounding per
    element), and the two swizzled images hold exactly that matrix in the MFMA lane order documented in include/ofasys_amd.h."""
    if not with_abs and (not slots or T != S or slots[-1][0] + slots[-1][1] != T):
        pytest.skip("without an abs-pos matrix the slots define the size")
    g = torch.Generator(device="cpu").manual_seed(A + T + S)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV).bfloat16()                  # noqa: E731
    abs_b = mk(A, T, S) if with_abs else None
    starts = [s for s, _ in slots]
    values = [mk(n, n, A) for _, n in slots]
    if with_abs:
        out, (sr, sc) = K.bias_build(abs_b, starts, values)
    else:
        out, (sr, sc) = K.bias_build(None, starts, values, heads=A)
    ref = abs_b.clone() if with_abs else torch.zeros(A, T, S, device=DEV, dtype=torch.bfloat16)
    for s, v in zip(starts, values):
        n = v.shape[0]
        ref[:, s:s + n, s:s + n] += v.permute(2, 0, 1)
    assert torch.equal(out, ref)
    nqt, nkt = (T + 31) // 32, (S + 31) // 32
    pad = torch.zeros(A, nqt * 32, nkt * 32, device=DEV, dtype=torch.bfloat16)
    pad[:, :T, :S] = ref
    lane = torch.arange(64, device=DEV)
    i, hi = lane & 31, lane >> 5
    r = torch.arange(16, device=DEV)
    col = (r & 3)[None, :] + 8 * (r >> 2)[None, :] + 4 * hi[:, None]                # crowl(r, hi): [64, 16]
    tiles = pad.view(A, nqt, 32, nkt, 32).permute(0, 1, 3, 2, 4)                  # [A, qt, kt, row, col]
    row_img = tiles[:, :, :, i[:, None].expand(64, 16), col]                       # [A, qt, kt, 64, 16]
    col_img = tiles[:, :, :, col, i[:, None].expand(64, 16)].permute(0, 2, 1, 3, 4)   # [A, kt, qt, 64, 16]
    assert torch.equal(sr.view(A, nqt, nkt, 64, 16), row_img)
    assert torch.equal(sc.view(A, nkt, nqt, 64, 16), col_img)


@pytest.mark.parametrize("Fr,P,s0,Tt,nt", [(3, 13, 5, 50, 6), (2, 12, 4, 32, 4), (8, 196, 0, 1600, 32)])   # (element-wise / 8-byte pieces / cfg-4)
def test_bias_build_outer_slot_and_its_gradient(K, Fr, P, s0, Tt, nt):
    """A video slot's rel-pos values are frames[i // P][j // P] + patches[i % P][j % P] (video_image_sequence.py:187-204): the assembly
    reads the two tables (bit-exact against the reference's broadcast add + block add in bf16), the gradient kernels sum the bias
    gradient's block straight into them (fp32 accumulation, against torch's two reductions)."""
    from ofasys_amd import ops
    A = 6                                                    # the outer block at s0, a dense text slot in the last nt positions
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV).bfloat16()                  # noqa: E731
    abs_b, vf, vi, vt = mk(A, Tt, Tt), mk(Fr, Fr, A), mk(P, P, A), mk(nt, nt, A)
    assert s0 + Fr * P <= Tt - nt
    out, (sr, sc) = K.bias_build(abs_b, [s0, Tt - nt], [(vf, vi), vt])
    dense = ops.OuterRelPos(vf, vi).dense()                                          # bf16 broadcast add, as the reference
    ref = abs_b.clone()
    ref[:, s0:s0 + Fr * P, s0:s0 + Fr * P] += dense.permute(2, 0, 1)
    ref[:, Tt - nt:, Tt - nt:] += vt.permute(2, 0, 1)
    assert torch.equal(out, ref)
    G = mk(1, A, Tt, Tt)
    dvf, dvi = K.bias_outer_grad(G, s0, Fr, P)
    blk = G[0, :, s0:s0 + Fr * P, s0:s0 + Fr * P].float().view(A, Fr, P, Fr, P)
    assert rel(dvf, blk.sum((2, 4)).permute(1, 2, 0)) < 4e-3 and rel(dvi, blk.sum((1, 3)).permute(1, 2, 0)) < 4e-3
    # through autograd: BiasAssembleFn with an outer slot == the dense formulation
    a1 = abs_b.unsqueeze(0).clone().requires_grad_(True)
    f1, p1 = vf.clone().requires_grad_(True), vi.clone().requires_grad_(True)
    b1, _, _ = ops.BiasAssembleFn.apply(a1, [s0], ["outer"], f1, p1)
    a2 = abs_b.unsqueeze(0).clone().requires_grad_(True)
    f2, p2 = vf.clone().requires_grad_(True), vi.clone().requires_grad_(True)
    b2, _, _ = ops.BiasAssembleFn.apply(a2, [s0], ["dense"], ops.OuterRelPos(f2, p2).dense())
    assert torch.equal(b1, b2)
    b1.backward(G)
    b2.backward(G)
    assert rel(f1.grad, f2.grad) < 1e-2 and rel(p1.grad, p2.grad) < 1e-2 and torch.equal(a1.grad, a2.grad)
