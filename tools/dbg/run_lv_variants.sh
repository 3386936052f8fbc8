#!/bin/bash
mkdir -p gpurun_out/$1
for f in tools/dbg/variants/lib_lv_*.so; do
  n=$(basename $f .so)
  echo -n "$n: "
  CVLLM_LIB_PATH=$PWD/$f python tools/microbench.py scoring --L 32768 2>&1 | grep "leverage"
done | tee gpurun_out/$1/lv_variants.log
