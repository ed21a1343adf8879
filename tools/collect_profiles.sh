#!/bin/bash
# Evidence of a round (ROUND=<n>, default 6): kernel-trace stats of the bench commands (cfg-2 headline, cfg-2b, cfg-4), HBM PMC passes, MFMA / LDS PMC pass, microbenchmarks.
# Run on the GPU box:  bash tools/collect_profiles.sh   (outputs under gpurun_out/profiles/, copied into profiles/ afterwards)
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/profiles; mkdir -p $O
N=${ROUND:-6}
# every command under its own watchdog: a tracer stuck on a faulting process (round 5: 25 GPU-minutes) costs its limit, not the lease
T() { local t=$1; shift; timeout -k 15 "$t" "$@"; local rc=$?; [ $rc -eq 124 ] && echo "!!! watchdog: '$*' exceeded ${t}s"; return $rc; }
cd /tmp && export TMPDIR=/tmp
# 4 distinct batches x 3 set-up steps (2 eager + capture) + 2 warm-up + 10 timed = 24 train steps in the trace
T 600 rocprofv3 --kernel-trace --stats -d /tmp/r${N}stats -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_run.log 2>&1
python $R/tools/prof_summary.py /tmp/r${N}stats/p_results.db 24 70 --json $O/round${N}_rocprof_kernel_stats.json > $O/round${N}_rocprof_kernel_stats.txt 2>&1
python $R/tools/prof_by_grid.py /tmp/r${N}stats/p_results.db > $O/round${N}_rocprof_by_grid.txt 2>&1
for w in cfg2b cfg4; do
  T 600 rocprofv3 --kernel-trace --stats -d /tmp/r${N}stats_$w -o p -- python $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_${w}_run.log 2>&1
  python $R/tools/prof_summary.py /tmp/r${N}stats_$w/p_results.db 24 90 --json $O/round${N}_rocprof_${w}_kernel_stats.json > $O/round${N}_rocprof_${w}_kernel_stats.txt 2>&1
done
# the two multi-task workloads: 4 batches x 3 set-up steps + 2 warm-up + 6 timed = 20 train steps in the trace
for w in cfg3 cfg5; do
  T 600 rocprofv3 --kernel-trace --stats -d /tmp/r${N}stats_$w -o p -- python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --profile-gemm 0 > $O/stats_${w}_run.log 2>&1
  python $R/tools/prof_summary.py /tmp/r${N}stats_$w/p_results.db 20 90 --json $O/round${N}_rocprof_${w}_kernel_stats.json > $O/round${N}_rocprof_${w}_kernel_stats.txt 2>&1
done
# HBM traffic: separate PMC passes (eager: 3 + 1 = 4 train steps each)
T 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r${N}fetch -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/fetch_run.log 2>&1
T 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r${N}write -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/write_run.log 2>&1
python $R/tools/pmc_traffic.py /tmp/r${N}fetch/p_results.db /tmp/r${N}write/p_results.db $O/round${N}_pmc_traffic.json 4 > $O/round${N}_pmc_traffic.txt 2>&1
# ... and of the reference's default configuration (cfg-2b)
T 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/r${N}fetch_cfg2b -o p -- python $R/bench.py --workload cfg2b --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/fetch_cfg2b_run.log 2>&1
T 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/r${N}write_cfg2b -o p -- python $R/bench.py --workload cfg2b --steps 3 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/write_cfg2b_run.log 2>&1
python $R/tools/pmc_traffic.py /tmp/r${N}fetch_cfg2b/p_results.db /tmp/r${N}write_cfg2b/p_results.db $O/round${N}_pmc_traffic_cfg2b.json 4 > $O/round${N}_pmc_traffic_cfg2b.txt 2>&1
# MFMA busy / LDS activity / bank conflicts per kernel
T 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d /tmp/r${N}mfma -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph > $O/mfma_run.log 2>&1
(set +x; echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"
 echo "#   -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-gemm 0 --no-graph   (MI355X, round $N, packed cfg-2 step; tools/pmc_dump.py)"
 echo "# Averages per launch, grouped by (kernel, grid x).  MFMA utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs);"
 echo "# LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Appended per line as [mfma xx% | lds-conflict yy%]."
 python $R/tools/pmc_dump.py /tmp/r${N}mfma/p_results.db) > $O/round${N}_pmc_mfma_lds.txt 2>&1
cd $R
python tools/build_id.py > $O/round${N}_source_id.txt
# the bench lines below quote the kernel-trace / PMC summaries of THIS run (bench.py reads them from profiles/)
cp $O/round${N}_rocprof_kernel_stats.json $O/round${N}_rocprof_cfg2b_kernel_stats.json $O/round${N}_rocprof_cfg4_kernel_stats.json $O/round${N}_rocprof_cfg3_kernel_stats.json $O/round${N}_rocprof_cfg5_kernel_stats.json $O/round${N}_pmc_traffic.json $O/round${N}_pmc_traffic_cfg2b.json $R/profiles/
T 600 python bench.py > $O/round${N}_bench.json 2> $O/bench_run.log
tail -c 1200 $O/round${N}_bench.json
T 600 python bench.py --workload cfg2b --steps 30 --warmup 5 > $O/round${N}_bench_cfg2b.json 2> $O/bench_cfg2b_run.log
T 600 python bench.py --workload cfg4 --steps 30 --warmup 5 > $O/round${N}_bench_cfg4.json 2> $O/bench_cfg4_run.log
T 600 python bench.py --workload cfg3 --steps 30 --warmup 5 > $O/round${N}_bench_cfg3.json 2> $O/bench_cfg3_run.log
T 600 python bench.py --workload cfg5 --steps 10 --warmup 3 > $O/round${N}_bench_cfg5.json 2> $O/bench_cfg5_run.log
T 600 python bench.py --workload cfg2b --no-pack --steps 30 --warmup 5 --no-cpu-baseline > $O/round${N}_bench_cfg2b_padded.json 2> $O/bench_cfg2b_padded_run.log
T 300 python tools/attn_bench.py > $O/round${N}_attn_bench.txt 2>&1
for w in cfg2b cfg4 dec cross; do T 300 python tools/attn_sbias_bench.py $w 2>&1 | grep -v amdgpu.ids >> $O/round${N}_attn_sbias_bench.txt; done
for w in cfg2 cfg2b cfg4; do T 300 python tools/native_glue_trace.py $w 2>&1 | grep -v amdgpu.ids > $O/round${N}_native_glue_$w.txt; done
timeout 300 tools/experiments/_build/adam_stream_bench > $O/round${N}_adam_stream_bench.txt 2>&1 || true
(T 300 python tools/gemm_split_check.py; OFASYS_AMD_LIB=$R/ofasys_amd/libofasys_amd_dbg.so OFA_GEMM_SPLIT_MIN_K=1000000 T 300 python tools/gemm_split_check.py) 2>&1 | grep -v amdgpu.ids > $O/round${N}_gemm_split_check.txt
# what every parity test measures next to its bound (pytest -s prints MEASURED lines)
ROUND=$N T 600 python tools/gemm_bench.py > /dev/null 2>&1 || true
# round 6: the mixed-height plan, the fused criterion, the attention column-sum epilogues, the decoder-side products in a replayed graph
OFASYS_AMD_LIB=$R/ofasys_amd/libofasys_amd_dbg.so T 600 python tools/gemm_mixed_bench.py 2>&1 | grep -v amdgpu.ids > $O/round${N}_gemm_mixed_bench.txt
T 300 python tools/ce_bench.py 2>&1 | grep -v amdgpu.ids > $O/round${N}_ce_bench.txt
T 300 python tools/attn_cs_bench.py 2>&1 | grep -v amdgpu.ids > $O/round${N}_attn_cs_bench.txt
OFASYS_AMD_LIB=$R/ofasys_amd/libofasys_amd_dbg.so T 300 python tools/gemm_small_graph.py 2>&1 | grep -v amdgpu.ids > $O/round${N}_gemm_small_graph.txt
python tools/prof_last_step.py /tmp/r${N}stats/p_results.db > $O/round${N}_last_step_sequence.txt 2>&1 || true
T 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed" > $O/round${N}_parity_measured.txt
