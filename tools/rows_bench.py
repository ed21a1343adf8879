"""Microbenchmarks of the row kernels (GELU+LayerNorm, residual join) at the cfg-2 shapes; prints us and effective TB/s."""
import sys, torch
sys.path.insert(0, '.')
from ofasys_amd import ops
from ofasys_amd.module.layers import LayerNorm
dev = 'cuda'
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows in (14336, 2048):
    D, F = 768, 3072
    h = torch.randn(rows, F, device=dev).bfloat16().requires_grad_()
    ln = LayerNorm(F).to(dev).bfloat16()
    y = ops.layer_norm(h, ln.weight, ln.bias, 1e-5, fuse_gelu=True)
    dy = torch.randn_like(y)
    tf = bench(lambda: ops.layer_norm(h, ln.weight, ln.bias, 1e-5, fuse_gelu=True))
    tb = bench(lambda: torch.autograd.grad(y, h, dy, retain_graph=True))
    by = rows * F * 2
    print(f"gelu_ln rows={rows}: fwd {tf:7.1f} us {2*by/tf/1e6:5.2f} TB/s | bwd {tb:7.1f} us {3*by/tb/1e6:5.2f} TB/s")
    x = torch.randn(rows, D, device=dev).bfloat16().requires_grad_()
    r = torch.randn(rows, D, device=dev).bfloat16().requires_grad_()
    la, lb = LayerNorm(D).to(dev).bfloat16(), LayerNorm(D).to(dev).bfloat16()
    yy, zz = ops.residual_join(x, r, la, 0.1, True, lb)
    g1, g2 = torch.randn_like(yy), torch.randn_like(zz)
    tf = bench(lambda: ops.residual_join(x, r, la, 0.1, True, lb))
    tb = bench(lambda: torch.autograd.grad([yy, zz], [x, r], [g1, g2], retain_graph=True))
    by = rows * D * 2
    print(f"join    rows={rows}: fwd {tf:7.1f} us {4*by/tf/1e6:5.2f} TB/s | bwd {tb:7.1f} us {6*by/tb/1e6:5.2f} TB/s")
