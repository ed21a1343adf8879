#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r2fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > /tmp/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r2write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > /tmp/w.log 2>&1
ls /tmp/r2fetch | head; tail -3 /tmp/f.log
python $R/tools/pmc_traffic.py /tmp/r2fetch/p_results.db /tmp/r2write/p_results.db $R/gpurun_out/k_pmc_traffic.json 4 > $R/gpurun_out/k_pmc_traffic.txt 2>&1
head -16 $R/gpurun_out/k_pmc_traffic.txt; tail -1 $R/gpurun_out/k_pmc_traffic.txt
cd $R; python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']; print(round(o['ms_per_step'],3), round(o['value']), round(r['frac'],4), round(r['gemm_ms_per_step'],3))"
