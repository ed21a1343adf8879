#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -x -q -m gpu > gpurun_out/q_tests_full.log 2>&1; grep -E "passed|failed" gpurun_out/q_tests_full.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
