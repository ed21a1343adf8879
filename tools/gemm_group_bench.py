"""Grouped weight-gradient launch (ofa_gemm_group_tn) at the cfg-2 layer shapes: microseconds per launch, TFLOP/s."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from ofasys_amd import kernels as K   # noqa: E402

GROUPS = {
    "encoder layer (K=13312)": [(2304, 768, 13312), (768, 768, 13312), (3072, 768, 13312), (768, 3072, 13312)],
    "decoder layer (K=3072, cross k|v K=13312)": [(2304, 768, 3072), (768, 768, 3072), (768, 768, 3072), (1536, 768, 13312),
                                                  (768, 768, 3072), (3072, 768, 3072), (768, 3072, 3072)],
}
dev = torch.device("cuda", 0)
for name, shapes in GROUPS.items():
    prods = []
    for M, N, Kk in shapes:
        prods.append((torch.randn(Kk, M, device=dev).bfloat16(), torch.randn(Kk, N, device=dev).bfloat16(),
                      torch.zeros(M, N, device=dev), 1.0))
    q = K.FoldQueue()
    for _ in range(3):
        K.gemm_group_tn(prods, q)
        q.__init__()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        K.gemm_group_tn(prods, q)
        q.__init__()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = sum(2.0 * M * N * Kk for M, N, Kk in shapes)
    print(f"{name}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TF")
