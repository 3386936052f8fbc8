#!/bin/bash
# build a variant of libcvllm_hip.so with extra -D flags for ONE source file:  build_variant_file.sh FILE NAME -DFOO ...
set -e
R=/root/repo/compactor-vllm_amd
F=$1; N=$2; shift; shift
mkdir -p /tmp/var_$N
EXTRA=""
if [ "$F" = "prefill_attn" ]; then EXTRA="-fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize"; fi
if [ "$F" = "decode_attn" ]; then EXTRA="-mllvm -amdgpu-kernarg-preload-count=16"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $EXTRA "$@" -c $R/csrc/$F.hip -o /tmp/var_$N/$F.o
OBJS=$(ls $R/build/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/var_$N/$F.o -o /root/repo/tools/dbg/variants/lib_$N.so
echo built $N
