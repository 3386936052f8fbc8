#!/bin/bash
# build a variant of libcvllm_hip.so with extra -D flags for prefill_attn.hip:  build_variant.sh NAME -DFOO ...
set -e
R=/root/repo/compactor-vllm_amd
N=$1; shift
mkdir -p /tmp/var_$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize "$@" -c $R/csrc/prefill_attn.hip -o /tmp/var_$N/prefill_attn.o
OBJS=$(ls $R/build/*.o | grep -v prefill_attn.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/var_$N/prefill_attn.o -o /root/repo/tools/dbg/variants/lib_$N.so
echo built $N
