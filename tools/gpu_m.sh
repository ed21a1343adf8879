#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_trainstep_gpu.py -q -m gpu --timeout 900 > gpurun_out/m_tests.log 2>&1
grep -n "Fatal\|Error\|^E  \|passed\|failed" gpurun_out/m_tests.log | head -20
python -m pytest tests/test_configs_gpu.py -q -m gpu --timeout 900 > gpurun_out/m_tests2.log 2>&1
grep -n "Fatal\|Error\|^E  \|passed\|failed\|Current thread\|File \"/root/repo\|test_configs" gpurun_out/m_tests2.log | head -30
