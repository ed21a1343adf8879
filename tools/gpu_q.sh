#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "resnet or video or cfg4 or cfg3" 2>&1 | tail -2
echo "== cfg2b grouped"; timeout 600 python bench.py --workload cfg2b --steps 20 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>&1 | tail -1 | cut -c80-200
echo "== cfg2b ungrouped"; OFA_WGRAD_GROUP=0 timeout 600 python bench.py --workload cfg2b --steps 20 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>&1 | tail -1 | cut -c80-200
