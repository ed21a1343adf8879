#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for mb in 96 40 16 200; do echo "== pending $mb MB"; OFA_FOLD_PENDING_MB=$mb timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-gemm 0 2>&1 | tail -1 | cut -c80-200; done
