#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_WAIT[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CU_CYCLES" | sort -u | tr '\n' ' ' > $R/gpurun_out/p_counters.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/p1 -o p -- python $R/tools/gemm_tile_sweep.py 13312 > /tmp/p1.log 2>&1
python $R/tools/pmc_dump.py /tmp/p1/p_results.db gemm > $R/gpurun_out/p_gemm_stalls.txt 2>&1
cat $R/gpurun_out/p_counters.txt; echo; cat $R/gpurun_out/p_gemm_stalls.txt | cut -c1-330
tail -3 /tmp/p1.log
