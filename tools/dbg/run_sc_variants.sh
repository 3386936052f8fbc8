#!/bin/bash
# time the scoring microbench with every variant library:  run_sc_variants.sh <round> <grep pattern>
mkdir -p gpurun_out/$1
for f in tools/dbg/variants/lib_*.so; do
  n=$(basename $f .so)
  CVLLM_LIB_PATH=$PWD/$f python tools/microbench.py scoring --L 32768 2>&1 | grep "$2" | sed "s/^/$n: /"
done | tee gpurun_out/$1/sc_variants.log
