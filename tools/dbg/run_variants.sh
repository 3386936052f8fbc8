#!/bin/bash
# time every variant library on the 32K prefill microbench (numerics are NOT valid for the EXP_* variants)
mkdir -p gpurun_out/$1
for f in tools/dbg/variants/lib_*.so; do
  n=$(basename $f .so)
  echo -n "$n: "
  CVLLM_LIB_PATH=$PWD/$f python tools/microbench.py prefill --L ${2:-32768} 2>&1 | grep "prefill B"
done | tee gpurun_out/$1/variants.log
