"""Attention with the position bias: dense [B*A,T,S] bias kernels (round 2: the bias tensor is an INPUT here -- building it and reducing its
gradient cost extra kernels in the step) vs the positional kernels (round 3: pos_q / pos_k MFMAs + table gather in the kernel) vs no bias.
usage: python tools/attn_pos_bench.py [cfg2b|cfg4]      (env OFA_ATTN_POS_DQ_W2=1: the dQ kernel at two waves per SIMD, 39 spills)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K, ops  # noqa: E402


def bench(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


which = sys.argv[1] if len(sys.argv) > 1 else "cfg2b"
if which == "cfg2b":           # image 196 + text 252, base, batch 32: one id plane per slot
    B, A, T = 32, 12, 448
    slots = [(0, 196, [(torch.randint(0, 729, (196, 196)), 1)]), (196, 252, [((torch.arange(252)[:, None] - torch.arange(252)[None, :]).clamp(-170, 170) + 255, 0)])]
    rows = (511, 6892)
else:                          # 8 frames x 196 + text 32, batch 4: two planes on the video block
    B, A, T = 4, 12, 1600
    P, Fr = 196, 8
    fb = (torch.arange(Fr)[:, None] - torch.arange(Fr)[None, :]) + 255
    ib = torch.randint(0, 729, (P, P))
    slots = [(0, Fr * P, [(fb.view(Fr, 1, Fr, 1).expand(Fr, P, Fr, P).reshape(Fr * P, Fr * P), 1), (ib.view(1, P, 1, P).expand(Fr, P, Fr, P).reshape(Fr * P, Fr * P), 2)]),
             (Fr * P, 32, [((torch.arange(32)[:, None] - torch.arange(32)[None, :]) + 255, 0)])]
    rows = (511, 511, 6892)
D = A * 64
dev = "cuda"
q, k, v, pq, pk = (torch.randn(B, T, D, device=dev).bfloat16() for _ in range(5))
rel = ops.RelMap(slots, T, dev)
tables = [torch.randn(r, A, device=dev).bfloat16() for r in rows][:rel.ntables]
kpm = torch.zeros(B, T, dtype=torch.bool, device=dev)
kpm[:, T - 7:] = True
c = torch.ones(A, device=dev)
scale = 128 ** -0.5
print(f"{which}: B={B} heads={A} T=S={T}, {rel.planes} id plane(s), {rel.ncompact} compact ids; dQ kernel waves/SIMD = {2 if os.environ.get('OFA_ATTN_POS_DQ_W2') else 1}")
flops = 4.0 * B * A * T * T * 64
out, lse = K.attn_fwd(q, k, v, A, scale, kpm=kpm, c_attn=c)
dout = torch.randn_like(out)
t = bench(lambda: K.attn_fwd(q, k, v, A, scale, kpm=kpm, c_attn=c))
tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, kpm=kpm, c_attn=c))
print(f"no bias          fwd {t:8.1f} us ({flops / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * flops / tb / 1e6:6.0f} TF/s)")
if B * A * T * T * 2 < 8e9:
    bias = torch.randn(B * A, T, T, device=dev).bfloat16()
    out, lse = K.attn_fwd(q, k, v, A, scale, bias=bias, kpm=kpm, c_attn=c)
    t = bench(lambda: K.attn_fwd(q, k, v, A, scale, bias=bias, kpm=kpm, c_attn=c))
    tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, bias=bias, kpm=kpm, c_attn=c, need_dbias=True))
    print(f"dense bias       fwd {t:8.1f} us ({flops / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * flops / tb / 1e6:6.0f} TF/s)   (+ building the bias / reducing dbias elsewhere)")
    del bias
for name, r, tabs in (("pos, abs only", None, ()), ("pos, abs + rel", rel, tables)):
    pos = ops._PosCall(pq, pk, r, tabs, A, True)
    out, lse = K.attn_pos_fwd(q, k, v, A, scale, pos, kpm=kpm, c_attn=c)
    t = bench(lambda: K.attn_pos_fwd(q, k, v, A, scale, pos, kpm=kpm, c_attn=c))
    tb = bench(lambda: K.attn_pos_bwd(q, k, v, out, dout, lse, A, scale, pos, kpm=kpm, c_attn=c))
    f2 = flops * 1.5          # the abs-pos contraction: 128 instead of 64 in QK^T (and its three gradient products)
    print(f"{name:16s} fwd {t:8.1f} us ({f2 / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * f2 / tb / 1e6:6.0f} TF/s)   incl. d pos_q / d pos_k / table-gradient slab")
