#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3s; mkdir -p $O
for rep in 1 2; do for v in 1 0; do
  OFA_BN_REGATE=$v timeout 600 python bench.py --workload cfg2b --steps 40 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/b_$v.json 2> $O/b_$v.log
  python -c "
import json;d=json.load(open('$O/b_$v.json'));print('cfg2b regate=$v', round(d['ms_per_step'],3))"
done; done
python tools/bn_bench.py 2>&1 | grep rows
OFA_BN_REGATE=0 python tools/bn_bench.py 2>&1 | grep rows
