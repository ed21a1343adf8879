"""GPU: the position bias computed INSIDE the fused attention kernels (csrc/attention.hip MODE 2, ofa_attn_pos_fwd / _bwd,
ofa_relpos_table_grad) against a plain PyTorch fp32 reference of the reference's arithmetic: scores = scale * q k^T + the dense
[B,A,T,S] bias (abs-pos pos_q pos_k^T per head, adaptor/general.py:223-243 + table[bucket] on the slot blocks, :265-280), softmax,
PV, c_attn.  Tolerances as tests/test_kernels_gpu.py::test_fused_attention (bf16 in, fp32 accumulate).  The model-level goldens
(tests/test_model_gpu.py bf16, test_fp16_gpu.py) go through this path too; here every piece is checked on its own incl. both
id planes, causal masks, key padding, ragged (segment) mode and bitwise reproducibility of the table gradient."""
import pytest
import torch

from tests.test_kernels_gpu import _attn_ref, rel

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="no GPU")]
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from ofasys_amd import kernels
    return kernels


def _case(B, heads, T, S, planes, nbuckets, seed, self_attn=True, dtype=torch.bfloat16):
    from ofasys_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    D = heads * 64
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV).to(dtype)                  # noqa: E731
    q, k, v = mk(B, T, D), mk(B, S, D), mk(B, S, D)
    pq, pk = mk(B, T, D) * 0.5, mk(B, S, D) * 0.5
    rel_map, tables = None, []
    if planes:
        # two "slots": [0, n1) and [n1, T) carry rel-pos ids, everything off the diagonal blocks none (id 0)
        n1 = T // 3
        blocks = []
        for start, n in ((0, n1), (n1, T - n1)):
            pl = [(torch.randint(0, nbuckets[p], (n, n), generator=g), p) for p in range(planes)]
            blocks.append((start, n, pl))
        rel_map = ops.RelMap(blocks, T, DEV)
        tables = [(torch.randn(nbuckets[p], heads, generator=g) * 0.7).to(DEV).to(dtype) for p in range(planes)]
        dense = torch.zeros(heads, T, T)
        for start, n, pl in blocks:
            for ids, p in pl:
                dense[:, start:start + n, start:start + n] += tables[p].float().cpu()[ids].permute(2, 0, 1)
        rel_dense = dense.to(DEV)
    else:
        rel_dense = None
    return q, k, v, pq, pk, rel_map, tables, rel_dense


def _ref(q, k, v, pq, pk, tables, rel_map, heads, scale, kpm, c, causal, blocks_fn=None):
    """fp32 torch reference with autograd over q, k, v, pos_q, pos_k and the tables."""
    B, T, D = q.shape
    S = k.shape[1]
    leaves = [t.float().requires_grad_(True) for t in (q, k, v, pq, pk)]
    tl = [t.float().requires_grad_(True) for t in tables]
    qr, kr, vr, pqr, pkr = leaves
    ab = (pqr.view(B, T, heads, 64).transpose(1, 2) @ pkr.view(B, S, heads, 64).transpose(1, 2).transpose(-1, -2)) * scale
    bias = ab
    if rel_map is not None:
        ids = rel_map.ids.long()[:, :T, :S]                                         # [planes, T, S] compact ids
        used = rel_map.used.long()
        relb = torch.zeros(heads, T, S, device=q.device)
        for p in range(ids.shape[0]):
            u = used[ids[p]]
            slot, row = u >> 20, u & 0xfffff
            for s_, t in enumerate(tl):
                m = ((slot == s_) & (ids[p] > 0)).float()
                relb = relb + (t[row.clamp(max=t.shape[0] - 1)].permute(2, 0, 1) * m)
        bias = bias + relb.unsqueeze(0)
    cr = c.clone().requires_grad_(True)
    out = _attn_ref(qr, kr, vr, heads, scale, bias.reshape(B * heads, T, S), kpm, cr, causal)
    return out, leaves, tl, cr


@pytest.mark.parametrize("B,heads,T,S,planes,causal,use_kpm", [
    (2, 4, 64, 64, 1, False, False),
    (2, 2, 45, 45, 1, True, True),            # decoder self-attention: causal + rel-pos, ragged tail
    (1, 12, 130, 130, 2, False, True),        # two id planes (video: frame + image tables)
    (2, 4, 20, 77, 0, False, True),           # cross attention: abs-pos only
    (2, 2, 300, 300, 1, False, True),
    (1, 3, 448, 448, 1, False, False),        # the cfg-2b encoder shape (196 + 252)
])
def test_attn_pos_matches_dense_reference(K, B, heads, T, S, planes, causal, use_kpm):
    from ofasys_amd import ops
    nb = (37, 23)
    q, k, v, pq, pk, rel_map, tables, _ = _case(B, heads, T, S, planes, nb, seed=11 + T + planes)
    kpm = None
    if use_kpm:
        kpm = torch.zeros(B, S, dtype=torch.bool, device=DEV)
        kpm[-1, S - 5:] = True
    c = (1 + 0.2 * torch.randn(heads, device=DEV)).float()
    scale = (64 * 2) ** -0.5
    ref, leaves, tl, cr = _ref(q, k, v, pq, pk, tables, rel_map, heads, scale, kpm, c, causal)
    dout = torch.randn(B, T, heads * 64, device=DEV).bfloat16()
    ref.backward(dout.float())
    pos = ops._PosCall(pq, pk, rel_map, tables, heads, True)
    out, lse = K.attn_pos_fwd(q, k, v, heads, scale, pos, kpm=kpm, c_attn=c, causal=causal)
    assert rel(out, ref) < 2e-2
    dq, dk, dv, dpq, dpk, slab, delta = K.attn_pos_bwd(q, k, v, out, dout, lse, heads, scale, pos, kpm=kpm, c_attn=c, causal=causal)
    for got, want, name in ((dq, leaves[0].grad, "dq"), (dk, leaves[1].grad, "dk"), (dv, leaves[2].grad, "dv"),
                            (dpq, leaves[3].grad, "dpos_q"), (dpk, leaves[4].grad, "dpos_k")):
        assert rel(got, want) < 3e-2, name
    dc = delta.view(B, heads, -1)[:, :, :T].sum((0, 2)) / c
    assert rel(dc, cr.grad) < 3e-2
    if planes:
        assert slab.shape == (B * heads, (T + 127) // 128, rel_map.ncompact)
        dt = [torch.zeros_like(t) for t in tables]
        K.relpos_table_grad(slab, rel_map, heads, dt, False)
        for p in range(planes):
            assert rel(dt[p], tl[p].grad) < 3e-2, f"dtable{p}"
        # accumulate form + bitwise reproducibility of the whole backward (per-wave LDS histograms, fixed fold order)
        dt2 = [t.clone() for t in dt]
        K.relpos_table_grad(slab, rel_map, heads, dt2, True)
        for p in range(planes):
            assert rel(dt2[p], 2 * dt[p].float()) < 1e-2
        again = K.attn_pos_bwd(q, k, v, out, dout, lse, heads, scale, pos, kpm=kpm, c_attn=c, causal=causal)
        assert torch.equal(again[5], slab) and torch.equal(again[3], dpq) and torch.equal(again[4], dpk)


def test_attn_pos_equals_dense_bias_kernels(K):
    """The positional kernels against THIS BUILD's dense-bias kernels fed the materialised [B*A,T,S] tensor (fp32-built, rounded to
    bf16 once): the two must agree to bf16 rounding of the bias itself."""
    from ofasys_amd import ops
    B, heads, T = 2, 4, 160
    q, k, v, pq, pk, rel_map, tables, rel_dense = _case(B, heads, T, T, 1, (61,), seed=3)
    scale = (64 * 2) ** -0.5
    ab = (pq.float().view(B, T, heads, 64).transpose(1, 2) @ pk.float().view(B, T, heads, 64).transpose(1, 2).transpose(-1, -2)) * scale
    bias = (ab + rel_dense.unsqueeze(0)).reshape(B * heads, T, T).bfloat16()
    out_d, lse_d = K.attn_fwd(q, k, v, heads, scale, bias=bias)
    out_p, lse_p = K.attn_pos_fwd(q, k, v, heads, scale, ops._PosCall(pq, pk, rel_map, tables, heads, True))
    assert rel(out_p, out_d.float()) < 2e-2
    assert rel(lse_p, lse_d) < 2e-2


def test_attn_pos_ragged_segments_match_padded(K):
    """Ragged mode: samples packed back to back with a segment table, ids indexed by the position INSIDE the sample -- equal to the
    padded call at every valid row (forward and every gradient), filler rows of the outputs exactly zero."""
    from ofasys_amd import ops
    from ofasys_amd.packing import build_pack_plan
    B, heads, T = 3, 2, 96
    lens = [96, 41, 70]
    q, k, v, pq, pk, rel_map, tables, _ = _case(B, heads, T, T, 1, (29,), seed=21)
    scale = (64 * 2) ** -0.5
    kpm = torch.zeros(B, T, dtype=torch.bool)
    for b, n in enumerate(lens):
        kpm[b, n:] = True
    plan = build_pack_plan(kpm, kpm, bucket=64).to(DEV)
    assert plan.enc_prefix
    idx, inv = plan.enc_index, plan.enc_inverse
    pack = lambda t: K.gather_rows(t.reshape(B * T, -1).contiguous(), idx).view(1, -1, t.shape[-1])      # noqa: E731
    c = (1 + 0.1 * torch.randn(heads, device=DEV)).float()
    dout = torch.randn(B, T, heads * 64, device=DEV).bfloat16()
    pos = ops._PosCall(pq, pk, rel_map, tables, heads, True)
    out, lse = K.attn_pos_fwd(q, k, v, heads, scale, pos, kpm=kpm.to(DEV), c_attn=c, causal=True)
    g = K.attn_pos_bwd(q, k, v, out, dout, lse, heads, scale, pos, kpm=kpm.to(DEV), c_attn=c, causal=True)
    ppos = ops._PosCall(pack(pq), pack(pk), rel_map, tables, heads, True)
    pout, plse = K.attn_pos_fwd(pack(q), pack(k), pack(v), heads, scale, ppos, c_attn=c, causal=True, seg=plan.enc_self)
    pg = K.attn_pos_bwd(pack(q), pack(k), pack(v), pout, pack(dout), plse, heads, scale, ppos, c_attn=c, causal=True, seg=plan.enc_self)
    rows = torch.nonzero(idx >= 0).squeeze(1)
    filler = torch.nonzero(idx < 0).squeeze(1)
    src = idx[rows]

    def same(packed, padded, name, tol=1e-2):
        pr = packed.reshape(-1, packed.shape[-1])
        assert rel(pr[rows], padded.reshape(B * T, -1)[src].float()) < tol, name
        assert float(pr[filler].float().abs().max()) == 0.0, name + " filler rows"
    same(pout, out, "out")
    for i, name in ((0, "dq"), (1, "dk"), (2, "dv"), (3, "dpos_q"), (4, "dpos_k")):
        same(pg[i], g[i], name, 2e-2)
    dt_a = [torch.zeros_like(t) for t in tables]
    dt_b = [torch.zeros_like(t) for t in tables]
    K.relpos_table_grad(g[5], rel_map, heads, dt_a, False)
    K.relpos_table_grad(pg[5], rel_map, heads, dt_b, False)
    assert rel(dt_b[0], dt_a[0].float()) < 2e-2


def test_posbias_autograd_through_ops_attention():
    """ops.attention with an ops.PosBias: gradients reach pos_q, pos_k and the tables through autograd (fused path) and agree with the
    dense fallback (`PosBias.dense()` + the exact-tier attention)."""
    from ofasys_amd import ops
    B, heads, T = 2, 2, 48
    q, k, v, pq, pk, rel_map, tables, _ = _case(B, heads, T, T, 1, (19,), seed=5)
    scale = (64 * 2) ** -0.5

    class _Lazy:                                                         # the dense blocks of PosBias.dense(): values [n, n, A]
        def __init__(self, ids, tab):
            self.ids, self.tab = ids, tab

        def values(self):
            return ops.embedding(self.ids, self.tab)
    n1 = T // 3
    used = rel_map.used.long()
    ids = rel_map.ids.long()[0]
    blocks = [(0, _Lazy((used[ids[:n1, :n1]] & 0xfffff), None)), (n1, _Lazy((used[ids[n1:T, n1:T]] & 0xfffff), None))]
    grads = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (q, k, v, pq, pk)]
        tab = tables[0].clone().requires_grad_(True)
        for _, lz in blocks:
            lz.tab = tab
        pb = ops.PosBias(leaves[3], leaves[4], heads, scale, rel_map, (tab,), blocks)
        if fused:
            out, _ = ops.attention(leaves[0], leaves[1], leaves[2], heads, scale, bias=pb)
        else:
            out, _ = ops.attention(leaves[0].float(), leaves[1].float(), leaves[2].float(), heads, scale,
                                   bias=ops.PosBias(leaves[3].float(), leaves[4].float(), heads, scale, rel_map, (tab,), blocks))
        out.float().square().sum().backward()
        grads.append([t.grad.float() for t in leaves] + [tab.grad.float()])
    for a, b, name in zip(grads[0], grads[1], ("dq", "dk", "dv", "dpos_q", "dpos_k", "dtable")):
        assert rel(a, b) < 4e-2, name
