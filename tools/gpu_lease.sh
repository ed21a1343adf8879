#!/bin/bash
# ONE parameterised script for a GPU lease (replaces the per-lease tools/gpu_*.sh of rounds 1-3):
#   gpurun --timeout 1500 -- 'bash tools/gpu_lease.sh <stage> [<stage> ...]'
# Stages (each writes under gpurun_out/<stage>.*; a failing stage does not stop the next one):
#   tests            the whole GPU suite:  python -m pytest tests -m gpu -q
#   tests:<expr>     a pytest -k expression, with -s (the MEASURED lines)
#   smoke            __graft_entry__.smoke()
#   bench[:<wl>]     python bench.py [--workload <wl>]  (cfg2 cfg2b cfg3 cfg4 cfg5), default steps
#   quick[:<wl>]     the same with --steps 10 --warmup 3 --no-cpu-baseline
#   profiles         tools/collect_profiles.sh (kernel traces, PMC passes, microbenchmarks of the round: ROUND=<n>)
#   py:<script>      python tools/<script>.py
#   sh:<file>        bash <file> (a scratch experiment kept OUT of the tree's tools/: scratch/*.sh)
# Every stage runs under its own watchdog (`timeout -k 15 <seconds>`; STAGE_TIMEOUT=<s> overrides the per-stage default): a hung
# process -- round 5 lost 25 GPU-minutes to a tracer stuck on a faulting graph -- costs its own limit, never the whole lease.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
W() { local t=${STAGE_TIMEOUT:-$1}; shift; timeout -k 15 "$t" "$@"; local rc=$?; [ $rc -eq 124 ] && echo "!!! watchdog: '$*' exceeded ${t}s and was killed"; return $rc; }
for st in "$@"; do
  name=${st%%:*}; arg=""; [[ "$st" == *:* ]] && arg=${st#*:}
  tag=$(echo "$st" | tr -c 'A-Za-z0-9_.\n' '_')
  echo "=== stage $st"
  case $name in
    tests) if [ -z "$arg" ]; then W 2400 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -5 $O/tests.log
           else W 1200 python -m pytest tests -m gpu -q -s -k "$arg" > $O/$tag.log 2>&1; grep -E "MEASURED|passed|failed|Error|error" $O/$tag.log | tail -60; fi ;;
    smoke) W 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
    bench) W 900 python bench.py ${arg:+--workload $arg} > $O/bench_${arg:-cfg2}.json 2> $O/bench_${arg:-cfg2}.err; tail -c 600 $O/bench_${arg:-cfg2}.json; tail -3 $O/bench_${arg:-cfg2}.err ;;
    quick) W 420 python bench.py ${arg:+--workload $arg} --steps 10 --warmup 3 --no-cpu-baseline > $O/quick_${arg:-cfg2}.json 2> $O/quick_${arg:-cfg2}.err
           python -c "import json,sys; d=json.loads(open('$O/quick_${arg:-cfg2}.json').read().strip().splitlines()[-1]); print('${arg:-cfg2}', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'tok/s, gemm frac', round(d['roofline'].get('frac',0),4), 'step frac', round(d['roofline']['step_frac'],4))" || tail -5 $O/quick_${arg:-cfg2}.err ;;
    profiles) W 3000 bash tools/collect_profiles.sh > $O/profiles.log 2>&1; tail -3 $O/profiles.log ;;
    py) W 900 python tools/$arg.py > $O/$tag.log 2>&1; tail -40 $O/$tag.log ;;
    sh) W 1200 bash $arg > $O/$tag.log 2>&1; tail -40 $O/$tag.log ;;
    *) echo "unknown stage $st" ;;
  esac
done
