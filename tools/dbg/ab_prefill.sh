#!/bin/bash
# A/B of prefill kernel builds on ONE box: the product library against every tools/dbg/variants/lib_*.so, interleaved rounds
# (separate processes; each microbench run warms ~40 ms before timing).   bash tools/dbg/ab_prefill.sh OUTDIR [L] [rounds]
O=gpurun_out/$1; mkdir -p $O
L=${2:-32768}; R=${3:-3}
for r in $(seq 1 $R); do
  echo -n "round $r product: "; python tools/microbench.py prefill --L $L 2>&1 | grep "prefill B"
  for f in tools/dbg/variants/lib_*.so; do
    n=$(basename $f .so)
    echo -n "round $r $n: "; CVLLM_LIB_PATH=$PWD/$f python tools/microbench.py prefill --L $L 2>&1 | grep "prefill B"
  done
done | tee -a $O/ab_prefill_$L.log
