import sys, torch
sys.path.insert(0, '/root/repo')
from ofasys_amd import kernels as K
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, A, T, S = 32, 12, 448, 448
D = A * 64
q = torch.randn(B, T, D, device='cuda').bfloat16(); k = torch.randn(B, S, D, device='cuda').bfloat16(); v = torch.randn(B, S, D, device='cuda').bfloat16()
bias = torch.randn(B * A, T, S, device='cuda').bfloat16()
kpm = torch.zeros(B, S, dtype=torch.bool, device='cuda'); kpm[:, S - 7:] = True
c = torch.ones(A, device='cuda')
for bz in (None, bias):
    out, lse = K.attn_fwd(q, k, v, A, 0.125, bias=bz, kpm=kpm, c_attn=c)
    dout = torch.randn_like(out)
    tf = bench(lambda: K.attn_fwd(q, k, v, A, 0.125, bias=bz, kpm=kpm, c_attn=c))
    tb = bench(lambda: K.attn_bwd(q, k, v, out, dout, lse, A, 0.125, bias=bz, kpm=kpm, c_attn=c, need_dbias=bz is not None))
    print("bias" if bz is not None else "no bias", f"fwd {tf:.1f} us  bwd(prep+dq+dkv+alloc) {tb:.1f} us")
