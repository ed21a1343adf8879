#!/bin/bash
# round 3, call N: the swizzled bias through LDS in dK/dV (and dQ at three waves per SIMD, OFA_ATTN_DQ3=1): gate, microbench, benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_sbias_gpu.py tests/test_packing_gpu.py -q -m gpu -x > $O/t_sbias.log 2>&1; rc=$?; echo "sbias rc=$rc"; tail -3 $O/t_sbias.log; grep -E "^FAILED|^E  " $O/t_sbias.log | head -20
if [ $rc -ne 0 ]; then exit 1; fi
OFA_ATTN_DQ3=1 timeout 900 python -m pytest tests/test_attn_sbias_gpu.py tests/test_packing_gpu.py -q -m gpu -x > $O/t_sbias3.log 2>&1; rc=$?; echo "sbias dq3 rc=$rc"; tail -3 $O/t_sbias3.log; grep -E "^FAILED|^E  " $O/t_sbias3.log | head -20
for w in cfg2b cfg4 cross; do python tools/attn_sbias_bench.py $w 2>&1 | grep -E "shared"; done | tee $O/sbias_bench_dq2.txt
for w in cfg2b cfg4 cross; do OFA_ATTN_DQ3=1 python tools/attn_sbias_bench.py $w 2>&1 | grep -E "shared"; done | tee $O/sbias_bench_dq3.txt
for v in 0 1; do
for w in cfg2b cfg4; do
  if [ $v = 1 ]; then export OFA_ATTN_DQ3=1; else unset OFA_ATTN_DQ3; fi
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_${w}_$v.json 2> $O/bench_${w}_$v.log
  python -c "
import json;d=json.load(open('$O/bench_${w}_$v.json'));print('$w dq3=$v', round(d['ms_per_step'],3), round(d['value']))"
done; done
