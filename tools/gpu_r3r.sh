#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_fp16_gpu.py -q -m gpu -x -k "batchnorm or resnet or conv or video or fp16" > $O/t.log 2>&1; rc=$?; echo "gate rc=$rc"; tail -3 $O/t.log; grep -E "^FAILED|^E  " $O/t.log | head
if [ $rc -ne 0 ]; then exit 1; fi
for w in cfg2b cfg4; do
timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 > $O/bench_$w.json 2> $O/bench_$w.log
python -c "
import json;d=json.load(open('$O/bench_$w.json'));print('$w', round(d['ms_per_step'],3), round(d['value']))"
done
