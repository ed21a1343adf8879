#!/bin/bash
# round 6, GPU call 1: pointer audit of the replayed step graphs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
DBG=$R/ofasys_amd/libofasys_amd_dbg.so
run() { local name=$1; shift; echo "##### $name: $*"; timeout -k 10 ${T:-400} "$@" > $O/$name.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/$name.log | tail -${TAIL:-40}; }
TAIL=80 run audit_cfg2b_shipped python tools/capture_audit.py --workload cfg2b --frames --stress
OFASYS_AMD_LIB=$DBG OFA_JOIN_BWD=9 TAIL=30 run audit_cfg2b_rowbwd_pins python tools/capture_audit.py --workload cfg2b --stress --rounds 3
OFA_CAPTURE_PINS=0 OFASYS_AMD_LIB=$DBG OFA_JOIN_BWD=9 TAIL=30 run audit_cfg2b_rowbwd_nopins python tools/capture_audit.py --workload cfg2b --stress --rounds 3
OFA_CAPTURE_PINS=0 TAIL=30 run audit_cfg2b_shipped_nopins python tools/capture_audit.py --workload cfg2b --stress --rounds 3
TAIL=30 run audit_cfg2 python tools/capture_audit.py --workload cfg2 --stress
TAIL=40 run audit_cfg3 python tools/capture_audit.py --workload cfg3 --stress
TAIL=40 T=600 run audit_cfg5 python tools/capture_audit.py --workload cfg5 --stress
TAIL=30 run audit_cfg4 python tools/capture_audit.py --workload cfg4 --stress
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/quick_cfg2.json 2> $O/quick_cfg2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_cfg2.json').read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],3), 'ms/step, gemm frac', round(d['roofline'].get('frac',0),4))
PY
