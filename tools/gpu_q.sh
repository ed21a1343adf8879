#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in "" rpg2 rpg3; do
  if [ -z "$v" ]; then echo "== product (1 read per gap)"; unset OFASYS_AMD_LIB; else echo "== $v"; export OFASYS_AMD_LIB=$R/tools/experiments/_build/libofasys_amd_$v.so; fi
  timeout 300 python tools/gemm_group_bench.py 2>&1 | tail -2
done
