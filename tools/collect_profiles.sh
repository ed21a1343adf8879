#!/bin/bash
# Round-3 evidence: kernel-trace stats of the bench commands (cfg-2 headline, cfg-2b, cfg-4), HBM PMC passes, MFMA / LDS PMC pass, microbenchmarks.
# Run on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/profiles/, copied into profiles/ afterwards)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 4 distinct batches x 3 set-up steps (2 eager + capture) + 2 warm-up + 10 timed = 24 train steps in the trace
rocprofv3 --kernel-trace --stats -d /tmp/r3stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r3stats/p_results.db 24 70 --json $O/round3_rocprof_kernel_stats.json > $O/round3_rocprof_kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r3stats/p_results.db > $O/round3_rocprof_by_grid.txt 2>&1
for w in cfg2b cfg4; do
  rocprofv3 --kernel-trace --stats -d /tmp/r3stats_$w -o p -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_${w}_run.log 2>&1
  python $R/tools/prof_summary.py /tmp/r3stats_$w/p_results.db 24 90 --json $O/round3_rocprof_${w}_kernel_stats.json > $O/round3_rocprof_${w}_kernel_stats.txt 2>&1
done
# HBM traffic: separate PMC passes (eager: 3 + 1 = 4 train steps each)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r3fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r3write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/write_run.log 2>&1
python $R/tools/pmc_traffic.py /tmp/r3fetch/p_results.db /tmp/r3write/p_results.db $O/round3_pmc_traffic.json 4 > $O/round3_pmc_traffic.txt 2>&1
# MFMA busy / LDS activity / bank conflicts per kernel
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/r3mfma -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/mfma_run.log 2>&1
(set +x; echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
 echo "#   -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph   (MI355X, round 3, packed cfg-2 step; tools/pmc_dump.py)"
 echo "# Averages per launch, grouped by (kernel, grid x).  MFMA utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs);"
 echo "# LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Appended per line as [mfma xx% | lds-conflict yy%]."
 python $R/tools/pmc_dump.py /tmp/r3mfma/p_results.db) > $O/round3_pmc_mfma_lds.txt 2>&1
cd $R
# the bench lines below quote the kernel-trace / PMC summaries of THIS run (bench.py reads them from profiles/)
cp $O/round3_rocprof_kernel_stats.json $O/round3_rocprof_cfg2b_kernel_stats.json $O/round3_rocprof_cfg4_kernel_stats.json $O/round3_pmc_traffic.json $R/profiles/
python bench.py > $O/round3_bench.json 2> $O/bench_run.log
tail -c 1200 $O/round3_bench.json
python bench.py --workload cfg2b --steps 30 --warmup 5 > $O/round3_bench_cfg2b.json 2> $O/bench_cfg2b_run.log
python bench.py --workload cfg4 --steps 30 --warmup 5 > $O/round3_bench_cfg4.json 2> $O/bench_cfg4_run.log
python bench.py --workload cfg2b --no-pack --steps 30 --warmup 5 --no-cpu-baseline > $O/round3_bench_cfg2b_padded.json 2> $O/bench_cfg2b_padded_run.log
python tools/attn_bench.py > $O/round3_attn_bench.txt 2>&1
for w in cfg2b cfg4 dec cross; do python tools/attn_sbias_bench.py $w 2>&1 | grep -v amdgpu.ids >> $O/round3_attn_sbias_bench.txt; done
for w in cfg2 cfg2b cfg4; do python tools/native_glue_trace.py $w 2>&1 | grep -v amdgpu.ids > $O/round3_native_glue_$w.txt; done
timeout 300 tools/experiments/_build/adam_stream_bench > $O/round3_adam_stream_bench.txt 2>&1 || true
(python tools/gemm_split_check.py; OFA_GEMM_SPLIT_MIN_K=1000000 python tools/gemm_split_check.py) 2>&1 | grep -v amdgpu.ids > $O/round3_gemm_split_check.txt
