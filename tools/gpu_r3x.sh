#!/bin/bash
# Round 3, validation of the late additions: the full GPU suite with the measured parity figures shown (-s), then smoke().
set +x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 420 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3x_suite.log 2>&1
echo "suite rc=$?"
grep -E "passed|failed|error" gpurun_out/r3x_suite.log | tail -3
grep -E "^MEASURED" gpurun_out/r3x_suite.log > gpurun_out/r3x_measured.txt
wc -l gpurun_out/r3x_measured.txt
grep -E "droppath|chmask|cfg-2" gpurun_out/r3x_measured.txt
grep -B2 -A25 "^E  \|Error" gpurun_out/r3x_suite.log | head -60
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
