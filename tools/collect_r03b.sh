#!/bin/bash
# Round-3 A/B evidence on ONE box (bash tools/collect_r03b.sh):
#  1. split merge of decode attention inside the engine's HIP graph: two-kernel (default) vs in-launch - tokens/s and the
#     kernel-trace durations of the decode kernels of the same command;
#  2. store-stream overlap vs the same chain on the main stream (--serial-store): tokens/s and the prefill kernel's mean.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b
mkdir -p "$O"
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-multi-seq --no-roofline"
for mode in two-kernel in-launch; do
  CVLLM_DECODE_MERGE=$mode $B > "$O/bench_merge_$mode.log" 2>&1 || exit 1
  grep '^{"metric"' "$O/bench_merge_$mode.log" | tail -1 > "$O/bench_merge_$mode.json"
  CVLLM_DECODE_MERGE=$mode timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$O/kt_$mode" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-multi-seq --no-roofline > "$O/kt_$mode.log" 2>&1 || exit 1
  { echo "# CVLLM_DECODE_MERGE=$mode rocprofv3 --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-roofline (C3 through LLM.generate, HIP-graph decode)"; python3 tools/prof_summary.py "$O/kt_$mode" decode_; python3 tools/prof_summary.py "$O/kt_$mode" prefill_attn | tail -1; } > "$O/kt_$mode.txt"
  rm -rf "$O/kt_$mode"
  echo "[merge $mode done]"
done
$B --serial-store > "$O/bench_serial_store.log" 2>&1 || exit 1
grep '^{"metric"' "$O/bench_serial_store.log" | tail -1 > "$O/bench_serial_store.json"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$O/kt_serial" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-multi-seq --no-roofline --serial-store > "$O/kt_serial.log" 2>&1 || exit 1
{ echo "# --serial-store: rocprofv3 --kernel-trace -- python3 bench.py --steps 1 --warmup 1 --no-roofline --serial-store"; python3 tools/prof_summary.py "$O/kt_serial" | head -30; } > "$O/kt_serial.txt"
{ echo "# default (store-stream overlap), same command without --serial-store (two-kernel merge run above)"; } >> "$O/kt_serial.txt"
rm -rf "$O/kt_serial"
python3 - <<'PY'
import json
O="gpurun_out/r03b"
for n in ("merge_two-kernel","merge_in-launch","serial_store"):
    d=json.load(open(f"{O}/bench_{n}.json"))
    print(n, d["value"], "tok/s", d["ms_per_step"], "ms/step")
PY
