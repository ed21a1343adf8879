"""Attention with the position bias: the reference's dense [B*A,T,S] tensor (round 2) vs ONE [A,T,S] matrix shared by the batch
(round 3: ofa_attn_sbias_*, the gradient summed over the batch by a third kernel) vs no bias.
usage: python tools/attn_sbias_bench.py [cfg2b|cfg4|dec]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K  # noqa: E402


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
B, A, T, S, causal = {"cfg2b": (32, 12, 448, 448, False), "cfg4": (4, 12, 1600, 1600, False), "dec": (32, 12, 64, 64, True),
                      "cross": (32, 12, 64, 448, False)}[which]
D = A * 64
dev = "cuda"
q = torch.randn(B, T, D, device=dev).bfloat16()
k, v = (torch.randn(B, S, D, device=dev).bfloat16() for _ in range(2))
kpm = torch.zeros(B, S, dtype=torch.bool, device=dev)
kpm[:, S - 7:] = True
c = torch.ones(A, device=dev)
scale = 128 ** -0.5
flops = 4.0 * B * A * T * S * 64
print(f"{which}: B={B} heads={A} T={T} S={S} causal={causal}")
out, lse = K.attn_fwd(q, k, v, A, scale, kpm=kpm, c_attn=c, causal=causal)
dout = torch.randn_like(out)
t = bench(lambda: K.attn_fwd(q, k, v, A, scale, kpm=kpm, c_attn=c, causal=causal))
tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, kpm=kpm, c_attn=c, causal=causal))
print(f"no bias             fwd {t:8.1f} us ({flops / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * flops / tb / 1e6:6.0f} TF/s)")
if B * A * T * S * 2 < 8e9:
    bias = torch.randn(B * A, T, S, device=dev).bfloat16()
    out, lse = K.attn_fwd(q, k, v, A, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal)
    t = bench(lambda: K.attn_fwd(q, k, v, A, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal))
    tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, bias=bias, kpm=kpm, c_attn=c, causal=causal, need_dbias=True))
    print(f"dense [B*A,T,S]     fwd {t:8.1f} us ({flops / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * flops / tb / 1e6:6.0f} TF/s)   + building the bias / reducing dbias over the batch elsewhere")
    del bias
sb = torch.randn(A, T, S, device=dev).bfloat16()
sw = K.bias_build(sb, want_out=False)[1]            # the swizzled images: built once per layer by the bias assembly
t_build = bench(lambda: K.bias_build(sb, want_out=False))
out, lse = K.attn_fwd(q, k, v, A, scale, bias=sb, kpm=kpm, c_attn=c, causal=causal, bias_shared=sw)
t = bench(lambda: K.attn_fwd(q, k, v, A, scale, bias=sb, kpm=kpm, c_attn=c, causal=causal, bias_shared=sw))
tb0 = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, bias=sb, kpm=kpm, c_attn=c, causal=causal, need_dbias=False, bias_shared=sw))
tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, scale, bias=sb, kpm=kpm, c_attn=c, causal=causal, need_dbias=True, bias_shared=sw))
print(f"shared [A,T,S]      fwd {t:8.1f} us ({flops / t / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * flops / tb / 1e6:6.0f} TF/s)   of which the batch-summed dS kernel {tb - tb0:6.1f} us; building the swizzled images {t_build:5.1f} us per layer")
