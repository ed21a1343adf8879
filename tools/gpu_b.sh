#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_trainstep_gpu.py tests/test_configs_gpu.py -q -m gpu --timeout 900 > gpurun_out/b_new_tests.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " gpurun_out/b_new_tests.log | head -60
python -m pytest tests/test_model_gpu.py -q -m gpu --timeout 900 -k "large" > gpurun_out/b_sel_tests.log 2>&1
tail -3 gpurun_out/b_sel_tests.log
