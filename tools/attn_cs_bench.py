"""What the column-sum epilogues of the attention backward kernels cost (ofa_attn_bwd_cs against ofa_attn_bwd, same build, interleaved):
encoder self-attention, decoder self-attention and cross-attention shapes of cfg-2; arms: all partial rows, q only, c_attn only, k only, k and v."""
import sys, os, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
from ofasys_amd.lib import lib, ptr, stream
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
has_cs = hasattr(lib().cdll, "ofa_attn_bwd_cs")
for (B, A, T, S, causal) in [(32, 12, 448, 448, False), (32, 12, 64, 64, True), (32, 12, 64, 448, False)]:
    D = A * 64
    q = torch.randn(B, T, D, device='cuda').bfloat16(); k = torch.randn(B, S, D, device='cuda').bfloat16(); v = torch.randn(B, S, D, device='cuda').bfloat16()
    c = torch.ones(A, device='cuda')
    out, lse = K.attn_fwd(q, k, v, A, 0.125, c_attn=c, causal=causal)
    dout = torch.randn_like(out)
    Tp = K.pad32(T)
    delta = torch.zeros(B * A, Tp, device='cuda')
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    nq, nk = 4 * B * ((T + 127) // 128), 4 * B * ((S + 127) // 128)
    wq = torch.empty(nq, D, device='cuda'); wk = torch.empty(nk, D, device='cuda'); wv = torch.empty(nk, D, device='cuda'); wc = torch.empty(nq, A, device='cuda')
    def plain():
        lib().call("ofa_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(dout), None, None, ptr(c), 0, ptr(lse), ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), None, B, A, T, S, Tp, D, D, D, 0.125, int(causal), None, 0, 0, 1, stream())
    def cs(a, b, cc, d):
        def f():
            lib().call("ofa_attn_bwd_cs", ptr(q), ptr(k), ptr(v), ptr(dout), None, None, ptr(c), 0, ptr(lse), ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), None, B, A, T, S, Tp, D, D, D, 0.125, int(causal), None, 0, 0, 1,
                       ptr(wq) if a else None, D, ptr(wk) if b else None, ptr(wv) if cc else None, D, ptr(wc) if d else None, stream())
        return f
    arms = [("plain", plain)]
    if has_cs:
        arms += [("cs all", cs(1, 1, 1, 1)), ("cs q", cs(1, 0, 0, 0)), ("cs c", cs(0, 0, 0, 1)), ("cs k", cs(0, 1, 0, 0)), ("cs kv", cs(0, 1, 1, 0))]
    res = {n: [] for n, _ in arms}
    for r in range(5):
        for n, f in arms:
            res[n].append(bench(f))
    print(f"B{B} A{A} T{T} S{S} causal={causal}: " + " | ".join(f"{n} {sorted(t)[len(t)//2]:.1f}" for n, t in res.items()), flush=True)
