#!/bin/bash
# Round-end check on a GPU box (gpurun -- 'bash tools/gpu_q.sh'): the -m gpu suite, the smoke call and the default bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.log; tail -c 400 gpurun_out/bench_final.json
