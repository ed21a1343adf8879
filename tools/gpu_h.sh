#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu --timeout 1200 > gpurun_out/h_all_gpu_tests.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/h_all_gpu_tests.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_profiles.sh > gpurun_out/h_profiles.log 2>&1
python - <<'PY'
import json
o=json.loads(open("gpurun_out/profiles/round2_bench.json").read().strip().splitlines()[-1]); r=o["roofline"]
print(round(o["ms_per_step"],3), round(o["value"]), r["frac"], r["gemm_ms_per_step"], r["traffic"], r["algorithmic_bytes_per_launch"], r.get("hbm"))
PY
