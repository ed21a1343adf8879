"""The criterion on the cfg-2 logits (1536 x 51265, padded to 51328): ofa_cross_entropy_fwd + ofa_cross_entropy_bwd (two kernels, three passes over
the logits) against ofa_cross_entropy_fwd_grad (one kernel, one read and one write), eager launches, medians of 5 x 30."""
import sys, torch
sys.path.insert(0, '.')
from ofasys_amd import kernels as K
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, V, ld in [(1536, 51265, 51328), (1536, 8192, 8192), (256, 51265, 51328)]:
    store = (torch.randn(rows, ld, device='cuda') * 3).bfloat16()
    t = torch.randint(0, V, (rows,), device='cuda')
    gs = torch.ones(1, device='cuda')
    lse, rl = K.cross_entropy_fwd(store, t, V, 1)
    d = torch.empty_like(store)
    def two():
        l, r = K.cross_entropy_fwd(store, t, V, 1)
        K.cross_entropy_bwd(store, t, l, gs, V, 1, dlogits=d)
    def one():
        K.cross_entropy_fwd_grad(store[:, :V], t, gs, V, 1)
    res = {"two": [], "one": []}
    for r in range(5):
        res["two"].append(bench(two)); res["one"].append(bench(one))
    print(rows, V, {k: round(sorted(v)[2], 1) for k, v in res.items()}, flush=True)
