#!/bin/bash
# round 3, call G: BatchNorm fold change gate, Adam streaming microbench, torch-native glue attribution, split-K check
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "batchnorm or conv or resnet" > $O/t_bn.log 2>&1; echo "bn rc=$?"; tail -2 $O/t_bn.log
timeout 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "resnet" > $O/t_model.log 2>&1; echo "model rc=$?"; tail -2 $O/t_model.log
timeout 300 tools/experiments/_build/adam_stream_bench > $O/adam_stream.txt 2>&1; cat $O/adam_stream.txt
timeout 300 python tools/bn_bench.py > $O/bn_bench.txt 2>&1; tail -20 $O/bn_bench.txt
timeout 300 python tools/gemm_split_check.py 2>&1 | grep -v amdgpu.ids > $O/split_default.txt; cat $O/split_default.txt
OFA_GEMM_SPLIT_MIN_K=1000000 timeout 300 python tools/gemm_split_check.py 2>&1 | grep -v amdgpu.ids > $O/split_never.txt; cat $O/split_never.txt
timeout 600 python tools/native_glue_trace.py cfg2b > $O/glue_cfg2b.txt 2>&1; head -60 $O/glue_cfg2b.txt
timeout 600 python tools/native_glue_trace.py cfg2 > $O/glue_cfg2.txt 2>&1; head -40 $O/glue_cfg2.txt
