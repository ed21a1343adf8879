#!/bin/bash
# round 6, GPU call 4: same-box A/B of the join backward and of the big-tile split-K plan; the GEMM microbenchmark
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 bash tools/ab_env.sh cfg2 3 OFA_JOIN_BWD=0 OFA_JOIN_BWD=9 2>&1 | tee $O/ab_join_bwd.txt
timeout 600 bash tools/ab_env.sh cfg2 3 OFA_GEMM_BIGSPLIT=0 OFA_GEMM_BIGSPLIT=1 2>&1 | tee $O/ab_bigsplit.txt
ROUND=6 timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm_bench.txt
OFASYS_AMD_LIB=$R/ofasys_amd/libofasys_amd_dbg.so OFA_GEMM_BIGSPLIT=0 ROUND=6x timeout 600 python tools/gemm_bench.py 2>&1 | grep -E "output projection dgrad" | tee $O/gemm_bench_nosplit.txt
