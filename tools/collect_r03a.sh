#!/bin/bash
# Round-3 evidence that needs no code change (run on the GPU box from the repo root: bash tools/collect_r03a.sh):
#   1. rocprofv3 --kernel-trace of the decode kernel at the batch shapes the bench line only event-times
#      (4 x 32K after 50 % retention = B 4, 16 640 rows per head; one GPU's share of C5 = B 8, 65 536 rows per head);
#   2. FETCH_SIZE / WRITE_SIZE of every bandwidth-bound store-stream kernel (traffic <= 1.15 x algorithmic),
#      separate --pmc passes, collection limited to those kernels.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a
mkdir -p "$O"
for cfg in "4 16640 8" "8 65536 4" "1 16640 32"; do
  set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/kt_$1_$2" -- python3 tools/microbench.py decode --B $1 --L $2 --splits $3 > "$O/kt_$1_$2.log" 2>&1 || exit 1
  { echo "# rocprofv3 --kernel-trace -- python3 tools/microbench.py decode --B $1 --L $2 --splits $3"; grep "^decode" "$O/kt_$1_$2.log"; python3 tools/prof_summary.py "$O/kt_$1_$2" decode; } >> "$O/r03_decode_batch_kernel_durations.txt"
  rm -rf "$O/kt_$1_$2"
  echo "[decode trace B=$1 L=$2 done]"
done
RX='store_all|compact_store|leverage|snapkv|chunk_mass|qkv_rope|zscore|sj_|sh_|select_'
MB="python3 tools/microbench.py scoring --L 32768 --iters 6 --warmup 4 --with-producer"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --kernel-trace --output-format csv -d "$O/pf" -- $MB > "$O/pmc_fetch.log" 2>&1
echo "pmc FETCH_SIZE rc=$?" | tee -a "$O/pmc_fetch.log"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --kernel-trace --output-format csv -d "$O/pw" -- $MB > "$O/pmc_write.log" 2>&1
echo "pmc WRITE_SIZE rc=$?" | tee -a "$O/pmc_write.log"
python3 tools/pmc_summary.py "$O/pf" "$O/pw" "" > "$O/r03_scoring_pmc_raw.json"
rm -rf "$O/pf" "$O/pw"
ls -la "$O"
