#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_packing_gpu.py -q -m gpu --timeout 900 -k "attn or attention or packed" 2>&1 | tail -3
python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/i_attn_bench.txt; cat gpurun_out/i_attn_bench.txt
