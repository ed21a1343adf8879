"""Decode attention kernel (csrc/attention_decode.hip) against its HBM roofline: bytes = 2 * B * S * D * sizeof(T) (K and V
once each); prints us and TB/s at beam-search shapes of the cfg-2 model (D = 768, 12 heads)."""
import sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
dev = 'cuda'
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
H, D = 12, 768
for B, S, cap in [(160, 448, 448), (160, 64, 64), (160, 256, 256), (32, 448, 448), (640, 448, 448), (160, 1024, 1024)]:
    q = torch.randn(B, D, device=dev).bfloat16()
    kc = torch.randn(B, cap, D, device=dev).bfloat16()
    vc = torch.randn(B, cap, D, device=dev).bfloat16()
    c = torch.ones(H, device=dev).bfloat16()
    bias = torch.randn(B * H, S, device=dev).bfloat16()
    t = bench(lambda: K.attn_decode(q, kc, vc, S, H, 0.088, bias=bias, c_attn=c))
    by = 2 * B * S * D * 2
    print(f"attn_decode B={B:4d} S={S:5d}: {t:7.1f} us  {by/t/1e6:5.2f} TB/s  ({by/1e6:.0f} MB)")
