#!/bin/bash
# per-kernel durations of the scoring microbench for every variant library:  run_sc_kern.sh <round> <kernel regex>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
for f in tools/dbg/variants/lib_*.so; do
  n=$(basename $f .so)
  export CVLLM_LIB_PATH=$PWD/$f
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -- python3 tools/microbench.py scoring --L 32768 > $O/kt_$n.log 2>&1
  echo "== $n"; python3 tools/prof_summary.py $O/kt_$n "$2" | cut -c1-60,100-170
  rm -rf $O/kt_$n
done | tee $O/sc_kern.log
