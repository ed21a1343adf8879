#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -q -m gpu --timeout 1200 > gpurun_out/h_all_gpu_tests.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed" gpurun_out/h_all_gpu_tests.log | head -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/collect_profiles.sh > gpurun_out/h_profiles.log 2>&1
tail -c 1200 gpurun_out/profiles/round2_bench.json
