#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_configs_gpu.py -q -m gpu --timeout 1200 -k "two_ranks or dp_step" > gpurun_out/o_dp2.log 2>&1
tail -12 gpurun_out/o_dp2.log | cut -c1-400
