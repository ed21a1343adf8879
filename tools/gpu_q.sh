#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5; grep -E "^FAILED|^ERROR" gpurun_out/pytest_gpu.log | head
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
