#!/bin/bash
# round 3, call F: gate (shared-bias + train-step tests), then the round-3 profile collection
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_attn_sbias_gpu.py tests/test_trainstep_gpu.py tests/test_packing_gpu.py -q -m gpu > $O/t_gate.log 2>&1; rc=$?; echo "gate rc=$rc"; tail -3 $O/t_gate.log
if [ $rc -ne 0 ]; then grep -E "^FAILED|^ERROR|^E  " $O/t_gate.log | head -20; exit 1; fi
timeout 1500 bash tools/collect_profiles.sh > $O/collect.log 2>&1; echo "collect rc=$?"
P=gpurun_out/profiles
head -30 $P/round3_rocprof_kernel_stats.txt; head -12 $P/round3_rocprof_cfg2b_kernel_stats.txt; head -12 $P/round3_rocprof_cfg4_kernel_stats.txt
tail -4 $P/round3_pmc_traffic.txt; cat $P/round3_attn_sbias_bench.txt
for f in round3_bench round3_bench_cfg2b round3_bench_cfg4 round3_bench_cfg2b_padded; do python -c "
import json;d=json.load(open('$P/$f.json'));r=d['roofline'];print('$f', round(d['ms_per_step'],2), round(d['value']), 'frac', round(r.get('frac',0),4), 'rocprof', (r.get('rocprof') or {}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
