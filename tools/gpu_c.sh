#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_packing_gpu.py -q -m gpu --timeout 900 > gpurun_out/c_pack_tests.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " gpurun_out/c_pack_tests.log | head -40
python -m pytest tests/test_kernels_gpu.py -q -m gpu --timeout 900 -k "attn or attention" > gpurun_out/c_attn_tests.log 2>&1
tail -3 gpurun_out/c_attn_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c_bench_packed.json 2> gpurun_out/c_bench_packed.err
python - <<'PY'
import json
for f in ("gpurun_out/c_bench_packed.json",):
    try:
        o=json.loads(open(f).read().strip().splitlines()[-1]); r=o["roofline"]
        print(f, o["ms_per_step"], o["value"], r["step_frac"], r.get("frac"), r.get("gemm_ms_per_step"), o["config"]["step_mode"], o["config"]["ragged_row_packing"])
    except Exception as e: print(f, "ERR", e)
PY
tail -5 gpurun_out/c_bench_packed.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pack > gpurun_out/c_bench_padded.json 2> gpurun_out/c_bench_padded.err
python - <<'PY'
import json
o=json.loads(open("gpurun_out/c_bench_padded.json").read().strip().splitlines()[-1]); r=o["roofline"]
print("padded", o["ms_per_step"], o["value"], r["step_frac"], r.get("frac"), r.get("gemm_ms_per_step"))
PY
