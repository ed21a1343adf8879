#!/bin/bash
# Same-box A/B of one debug switch of the working tree's DEBUG library on a step: arm "a" runs with `$3`, arm "b" with `$4` (env assignments,
# e.g. OFA_GROUP_FUSE=0 / OFA_GROUP_FUSE=1), interleaved.  gpurun -- 'bash tools/ab_env.sh cfg2 3 OFA_GROUP_FUSE=0 OFA_GROUP_FUSE=1'
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-cfg2}; N=${2:-3}; A=$3; B=$4
cd $R
export OFASYS_AMD_LIB=$R/ofasys_amd/libofasys_amd_dbg.so
for i in $(seq 1 $N); do
  for arm in "$A" "$B"; do
    env $arm python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arm', '$W', round(d['ms_per_step'],3), 'ms/step')"
  done
done
