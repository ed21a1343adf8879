"""Time the train step's GEMM shapes under one forced tile configuration (OFA_GEMM_TILE is read once per process):
  OFA_GEMM_TILE=0|22|12|34|44 python tools/gemm_tile_sweep.py [M ...]      -> lines "kind M N K us TF" 
(The OFA_GEMM_* planner overrides exist in the DEBUG library only: make -C ofasys_amd/csrc debug, then run with
OFASYS_AMD_LIB=ofasys_amd/libofasys_amd_dbg.so; the shipped library ignores them.)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K
dev = 'cuda'
torch.manual_seed(0)
Ms = [int(x) for x in sys.argv[1:]] or [13312]
tile = os.environ.get("OFA_GEMM_TILE", "0")
for M in Ms:
    shapes = [("NT", M, 2304, 768), ("NT", M, 768, 768), ("NT", M, 3072, 768), ("NT", M, 768, 3072),
              ("NN", M, 768, 2304), ("NN", M, 768, 768), ("NN", M, 768, 3072), ("NN", M, 3072, 768),
              ("TN", 2304, 768, M), ("TN", 768, 768, M), ("TN", 3072, 768, M), ("TN", 768, 3072, M)]
    if os.environ.get("OFA_SWEEP_SHAPES"):            # "NT,3584,4096,4096;NN,..." replaces the train step's list
        shapes = [(t.split(",")[0],) + tuple(int(v) for v in t.split(",")[1:]) for t in os.environ["OFA_SWEEP_SHAPES"].split(";")]
    for kind, m, n, k in shapes:
        ta, tb = {'NT': (False, True), 'NN': (False, False), 'TN': (True, False)}[kind]
        a = torch.randn((k, m) if ta else (m, k), device=dev).bfloat16()
        b = torch.randn((n, k) if tb else (k, n), device=dev).bfloat16()
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        try:
            for _ in range(3):
                K.gemm(a, b, ta, tb, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.gemm(a, b, ta, tb, out=out)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            err = ""
            if os.environ.get("OFA_SWEEP_CHECK"):
                A = (a.t() if ta else a).float(); Bm = (b.t() if tb else b).float()
                ref = A @ Bm                       # (torch fp32 matmul: the yard-stick of this tool only)
                err = f" err {float((out.float() - ref).abs().max() / ref.abs().max()):.2e}"
            print(f"tile{tile} {kind} {m} {n} {k} {us:8.1f} us {2.0*m*n*k/us/1e6:7.1f} TF{err}", flush=True)
        except Exception as e:
            print(f"tile{tile} {kind} {m} {n} {k} ERR {str(e)[:60]}", flush=True)
