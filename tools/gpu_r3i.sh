#!/bin/bash
# round 3, call I: swizzled shared-bias attention: parity gate, microbench (dq at two vs three waves per SIMD), model tests, benches
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_sbias_gpu.py -q -m gpu -x > $O/t_sbias.log 2>&1; rc=$?; echo "sbias rc=$rc"; tail -3 $O/t_sbias.log; grep -E "^FAILED|^E  " $O/t_sbias.log | head -20
if [ $rc -ne 0 ]; then exit 1; fi
for w in cfg2b cfg4 dec cross; do python tools/attn_sbias_bench.py $w 2>&1 | grep -v amdgpu.ids | grep -E "^cfg|^dec|^cross|shared|no bias"; done | tee $O/sbias_bench_dq2.txt
for w in cfg2b cfg4; do OFA_ATTN_DQ3=1 python tools/attn_sbias_bench.py $w 2>&1 | grep -E "shared"; done | tee $O/sbias_bench_dq3.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_packing_gpu.py tests/test_configs_gpu.py tests/test_trainstep_gpu.py -q -m gpu -x > $O/t_model.log 2>&1; echo "model rc=$?"; tail -3 $O/t_model.log; grep -E "^FAILED|^E  " $O/t_model.log | head -20
for w in cfg2 cfg2b cfg4; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_$w.json 2> $O/bench_$w.log
  python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', round(d['ms_per_step'],3), round(d['value']))"
done
