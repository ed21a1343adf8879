"""BatchNorm forward / backward at the ResNet-101 shapes of cfg-2b (batch 32, 224 x 224): run under
  rocprofv3 --kernel-trace --stats -d /tmp/bn -o p -- python tools/bn_bench.py ; python tools/prof_by_grid.py /tmp/bn/p_results.db
for microseconds per (kernel, grid); prints the algorithmic bytes per call next to the event-timed whole calls."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from ofasys_amd import kernels as K   # noqa: E402

SHAPES = [(401408, 64), (100352, 64), (100352, 256), (25088, 128), (25088, 512), (6272, 256), (6272, 1024)]
dev = torch.device("cuda", 0)
for rows, C in SHAPES:
    x = torch.randn(rows, C, device=dev).bfloat16()
    dy = torch.randn(rows, C, device=dev).bfloat16()
    g = torch.ones(C, device=dev).bfloat16()
    b = torch.zeros(C, device=dev).bfloat16()
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    def fwd():
        return K.batchnorm_fwd(x, g, b, rm, rv, True, 0.1, 1e-5, relu=True)
    y, mean, rstd = fwd()
    def bwd():
        return K.batchnorm_bwd(dy, y, x, g, mean, rstd, True, True, beta=b)
    bwd()
    torch.cuda.synchronize()
    out = []
    for fn, nbytes in ((fwd, 3 * rows * C * 2), (bwd, 7 * rows * C * 2)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        out.append(f"{us:7.1f} us {nbytes / us * 1e-6:5.2f} TB/s")
    print(f"rows {rows:7d} C {C:5d}: fwd {out[0]} | bwd {out[1]}")
