import sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, A, T, S, causal) in [(32, 12, 448, 448, False), (32, 12, 64, 64, True), (32, 12, 64, 448, False)]:
    D = A * 64
    q = torch.randn(B, T, D, device='cuda').bfloat16(); k = torch.randn(B, S, D, device='cuda').bfloat16(); v = torch.randn(B, S, D, device='cuda').bfloat16()
    kpm = torch.zeros(B, S, dtype=torch.bool, device='cuda'); kpm[:, S - 7:] = True
    c = torch.ones(A, device='cuda')
    out, lse = K.attn_fwd(q, k, v, A, 0.125, kpm=kpm, c_attn=c, causal=causal)
    dout = torch.randn_like(out)
    tf = bench(lambda: K.attn_fwd(q, k, v, A, 0.125, kpm=kpm, c_attn=c, causal=causal))
    Tp, Sp = K.pad32(T), K.pad32(S)
    delta = torch.zeros(B * A, Tp, device='cuda')
    dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    from ofasys_amd.lib import lib, ptr, stream
    def bwd():
        lib().call("ofa_attn_bwd", ptr(q), ptr(k), ptr(v), ptr(dout), None, ptr(kpm.view(torch.uint8)),
                   ptr(c), 0, ptr(lse), ptr(delta), ptr(out), ptr(dq), ptr(dk), ptr(dv), None, B, A, T, S, Tp, D, D, D, 0.125, int(causal), None, 0, 0, 1, stream())
    tb = bench(bwd)
    fl = 4.0 * B * A * T * S * 64 * (0.5 if causal else 1)
    print(f"B{B} A{A} T{T} S{S} causal={causal}: fwd {tf:7.1f} us {fl/tf/1e6:6.1f} TF | bwd(dq+dkv) {tb:7.1f} us {2.5*fl/tb/1e6:6.1f} TF")
