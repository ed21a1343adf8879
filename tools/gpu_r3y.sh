#!/bin/bash
# Round 3: the fp16 bench lines (the reference trainer's default precision) of the headline and the default configuration.
set +x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/profiles
for w in cfg2 cfg2b cfg4; do
  timeout 100 python bench.py --dtype fp16 --workload $w --no-cpu-baseline 2> gpurun_out/bench_fp16_$w.err | tail -1 > gpurun_out/profiles/round3_bench_fp16_$w.json
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/profiles/round3_bench_fp16_$w.json"))
    print("$w fp16", round(d["ms_per_step"], 2), "ms", round(d["value"]), d["unit"], "frac", round(d["roofline"]["frac"], 4), d["dtype"])
except Exception as e:
    print("$w fp16 failed", e)
PY
done
tail -3 gpurun_out/bench_fp16_cfg2.err
