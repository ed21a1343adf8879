#!/bin/bash
# round 3, call T: the full GPU suite, smoke(), and the N > 1 code path on one device (gloo, two ranks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3t; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/t_all.log 2>&1; echo "suite rc=$?"; tail -4 $O/t_all.log; grep -E "^FAILED|^E  " $O/t_all.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in cfg2 cfg2b; do
OFA_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --backend gloo --no-cpu-baseline --profile-gemm 0 --workload $w > $O/dp2_$w.json 2> $O/dp2_$w.log; echo "dp2 $w rc=$?"; tail -c 600 $O/dp2_$w.json; echo; grep -E "Error|error" $O/dp2_$w.log | head -5
done
