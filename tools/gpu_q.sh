#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fp16_gpu.py tests/test_packing_gpu.py -x -q -m gpu -k "attn or attention or pack or ragged or c_attn" > gpurun_out/attn_tests.log 2>&1; grep -E "passed|failed" gpurun_out/attn_tests.log | tail -2; grep -E "^FAILED|Error" gpurun_out/attn_tests.log | head -5
timeout 300 python tools/attn_bench.py 2>&1 | tail -4
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>&1 | tail -1 | cut -c80-200
