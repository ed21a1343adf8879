#!/bin/bash
# Measurement build for tools/gemm_timeline.py / tools/attn_timeline.py: the product objects, with gemm_mfma.hip compiled with
# -DOFA_GEMM_TIMELINE and attention.hip with -DOFA_ATTN_TIMELINE -> tools/experiments/_build/libofasys_amd_tl.so (git-ignored).
# Never loaded by the package itself.
set -e
cd "$(dirname "$0")/../.."
make -C ofasys_amd/csrc -j8 >/dev/null
mkdir -p tools/experiments/_build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
/opt/rocm/bin/hipcc $F -DOFA_GEMM_TIMELINE -c ofasys_amd/csrc/gemm_mfma.hip -o tools/experiments/_build/gemm_mfma_tl.o &
/opt/rocm/bin/hipcc $F -DOFA_ATTN_TIMELINE -c ofasys_amd/csrc/attention.hip -o tools/experiments/_build/attention_tl.o &
wait
objs=$(ls ofasys_amd/csrc/build/*.o | grep -v -e gemm_mfma.o -e attention.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/experiments/_build/gemm_mfma_tl.o tools/experiments/_build/attention_tl.o -o tools/experiments/_build/libofasys_amd_tl.so
ls -la tools/experiments/_build/libofasys_amd_tl.so
