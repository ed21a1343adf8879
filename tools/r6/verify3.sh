#!/bin/bash
# round 6, GPU call 3: the fix (stat_dma clamp) + the shipped row-per-wave join backward
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
run() { local name=$1; shift; echo "##### $name: $*"; timeout -k 10 ${T:-400} "$@" > $O/$name.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/$name.log | grep -E "${PAT:-.}" | tail -${TAIL:-12}; }
PAT="fault|round|TOTAL|COMPLETE|Error" run v_audit_cfg2b python tools/capture_audit.py --workload cfg2b --stress --rounds 3
PAT="fault|round|TOTAL|COMPLETE|Error" run v_audit_cfg3 python tools/capture_audit.py --workload cfg3 --stress
T=1500 PAT="MEASURED|passed|failed|Error|error" TAIL=40 run v_newtests python -m pytest tests/test_bench_parity_gpu.py tests/test_packing_gpu.py -q -s -x -k "cfg2b or behind_lse"
for w in cfg2 cfg2b cfg3 cfg4 cfg5; do
  timeout -k 10 420 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/q_$w.json 2> $O/q_$w.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/q_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],3), 'ms/step, gemm frac', round(d['roofline'].get('frac',0),4), 'step frac', round(d['roofline']['step_frac'],4))
except Exception as e:
    print('$w FAILED', e); print(open('$O/q_$w.err').read()[-1500:])
PY
done
T=2400 PAT="passed|failed|Error|error" TAIL=15 run v_gpusuite python -m pytest tests -m gpu -q -x
