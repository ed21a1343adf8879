#!/bin/bash
# round-2 GPU call A: the new parity tests + a short bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_trainstep_gpu.py tests/test_configs_gpu.py -q -m gpu -x --timeout 900 > gpurun_out/a_new_tests.log 2>&1
tail -30 gpurun_out/a_new_tests.log
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -m gpu --timeout 900 -k "softmax or step_schedule or large or two_tasks or adam" > gpurun_out/a_sel_tests.log 2>&1
tail -15 gpurun_out/a_sel_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
tail -c 1500 gpurun_out/a_bench.json; tail -5 gpurun_out/a_bench.err
