#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 0 33; do OFA_SWEEP_CHECK=1 OFA_GEMM_TILE=$t timeout 300 python tools/gemm_tile_sweep.py 13312 14336 2>&1 | grep -v amdgpu.ids; done > gpurun_out/g_tri_sweep.txt
python - <<'PY'
import collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/g_tri_sweep.txt'):
    p=l.split()
    if len(p)<7 or p[5]=='ERR': print(l); continue
    rows[(p[1],int(p[2]),int(p[3]),int(p[4]))][p[0]]=(float(p[5]), p[-1])
for k,v in rows.items():
    print(k, ' '.join(f"{t}:{v[t][0]:6.1f}({v[t][1]})" for t in v), f"{v['tile0'][0]/v['tile33'][0]:.2f}x" if 'tile33' in v else '')
PY
OFA_GEMM_TILE=33 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | tail -2
OFA_GEMM_TILE=33 timeout 600 python tools/gemm_sweep.py 2>&1 | tail -2
