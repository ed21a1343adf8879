#!/bin/bash
# Same-box A/B of the cfg-2 (or $1) step: a snapshot of an earlier commit ($PREV, default .ab_prev/base: `git archive <commit> | tar -x -C .ab_prev/base`,
# then make -C .ab_prev/base/ofasys_amd/csrc) against the working tree, interleaved, N rounds.  gpurun -- 'bash tools/ab_step.sh [workload] [rounds]'
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-cfg2}; N=${2:-3}
cd $R
for i in $(seq 1 $N); do
  for arm in prev cur; do
    if [ $arm = prev ]; then D=${PREV:-$R/.ab_prev/base}; else D=$R; fi
    (cd $D && python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$arm', '$W', round(d['ms_per_step'],3), 'ms/step')")
  done
done
