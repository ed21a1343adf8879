"""Phase timeline of the fused attention forward kernel (measurement build: tools/experiments/build_timeline_lib.sh):
  OFASYS_AMD_LIB=tools/experiments/_build/libofasys_amd_tl.so python tools/attn_timeline.py [B A T S causal]
Wave 0 of every workgroup stamps s_memtime at entry / first tiles landed / exit and, inside its fifth 32-key block, at each
phase boundary of fwd_block; this prints where a key block's time goes."""
import ctypes
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ofasys_amd import kernels as K
from ofasys_amd.lib import lib

B, A, T, S, causal = (int(x) for x in sys.argv[1:6]) if len(sys.argv) >= 6 else (32, 12, 448, 448, 0)
D = A * 64
torch.manual_seed(0)
q = torch.randn(B, T, D, device='cuda').bfloat16(); k = torch.randn(B, S, D, device='cuda').bfloat16(); v = torch.randn(B, S, D, device='cuda').bfloat16()
c = torch.ones(A, device='cuda')
for _ in range(3):
    K.attn_fwd(q, k, v, A, 0.125, c_attn=c, causal=bool(causal))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.attn_fwd(q, k, v, A, 0.125, c_attn=c, causal=bool(causal))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
nwg = ((T + 127) // 128) * B * A
buf = np.zeros((8192, 16), dtype=np.uint64)
cd = lib().cdll
cd.ofa_debug_attn_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int64]
rc = cd.ofa_debug_attn_timeline(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes)
assert rc == 0, rc
tl = buf[:min(nwg, 8192)].astype(np.int64)
tl = tl[tl[:, 8] != 0]
mhz = float(np.median((tl[:, 11] - tl[:, 8]) / np.maximum(tl[:, 15] - tl[:, 14], 1))) * 100.0
c2us = 1.0 / mhz
nkb = (S + 31) // 32
print(f"B{B} A{A} T{T} S{S} causal={causal}: {us:.1f} us/launch; {len(tl)} workgroups stamped, s_memtime {mhz:.0f} MHz, {nkb} key blocks per workgroup")
q_ = lambda x: f"mean {x.mean():7.1f}  p10 {np.percentile(x, 10):7.1f}  p50 {np.percentile(x, 50):7.1f}  p90 {np.percentile(x, 90):7.1f}"
print(f"  entry -> first K/V tiles landed [us] : {q_((tl[:, 9] - tl[:, 8]) * c2us)}")
print(f"  key-block loop [us]                  : {q_((tl[:, 11] - tl[:, 9]) * c2us)}   (= {((tl[:, 11] - tl[:, 9]) * c2us).mean() / nkb:.2f} us per block)")
print(f"  workgroup life [us]                  : {q_((tl[:, 11] - tl[:, 8]) * c2us)}")
rt0 = (tl[:, 14] - tl[:, 14].min()) * 0.01                 # s_memrealtime: 100 MHz
rt1 = (tl[:, 15] - tl[:, 14].min()) * 0.01
print(f"  workgroup ENTRY after the first one [us]: {q_(rt0)}  max {rt0.max():.1f}")
print(f"  workgroup EXIT  after the first entry   : {q_(rt1)}  max {rt1.max():.1f}   (device-side span of the launch)")
ok = tl[tl[:, 7] != 0]
names = ["K fragments: issue -> in registers", "4 S MFMAs + scale + row max (lane-local)", "cross-half max exchange (__shfl_xor)",
         "exp2 x16 + sum", "cross-half sum exchange", "rescale (if any) + wait for the V^T fragments", "pack + 4 PV MFMAs issued"]
if len(ok) == 0:
    sys.exit(0)
d = np.diff(ok[:, 0:8], axis=1)
print(f"  block 4 of wave 0, shader clocks ({len(ok)} workgroups):")
for i, nm in enumerate(names):
    print(f"    {nm:48s}: {q_(d[:, i])}")
print(f"    {'end of block -> through the stage barrier':48s}: {q_(ok[:, 10] - ok[:, 7])}")
print(f"    {'whole block incl. barrier':48s}: {q_(ok[:, 10] - ok[:, 0])}")
