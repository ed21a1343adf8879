#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "one wave"; timeout 300 python tools/attn_bias_bench.py 2>&1 | grep -v amdgpu
echo "two waves (16 spilled registers)"; OFASYS_AMD_LIB=$GRAFT_REPO_ROOT/tools/experiments/_build/libofasys_amd_dkv2.so timeout 300 python tools/attn_bias_bench.py 2>&1 | grep -v amdgpu
