#!/bin/bash
# Round evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh r01
# Writes small summaries into gpurun_out/<round>/ (copy the ones to keep into profiles/).
set -eo pipefail
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
mkdir -p "$O"
# 1. kernel trace + the tool's own stats of the bench command
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-multi-seq > "$O/bench_under_rocprof.log" 2>&1
python3 tools/prof_summary.py "$O/kt" > "$O/${R}_bench_c3_kernel_summary.txt"
cp "$(find "$O/kt" -name '*kernel_stats.csv' | head -1)" "$O/${R}_bench_c3_kernel_stats.csv"
grep "^{\"metric\"" "$O/bench_under_rocprof.log" | tail -1 > "$O/${R}_bench_c3_under_rocprof.json"
rm -rf "$O/kt"
echo "[1/6] kernel trace done"
# 2. HBM traffic of the decode kernel (separate --pmc passes; C3 per-layer decode shape)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pf" -- python3 tools/microbench.py decode --L 16640 --splits 32 > "$O/pmc_f.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pw" -- python3 tools/microbench.py decode --L 16640 --splits 32 > "$O/pmc_w.log" 2>&1
python3 tools/pmc_summary.py "$O/pf" "$O/pw" > "$O/${R}_bench_pmc_kernels.json"
rm -rf "$O/pf" "$O/pw"
echo "[2/6] decode PMC done"
# 3. SQ counters of the prefill kernel (two passes of 8 counters)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/s1" -- python3 tools/microbench.py prefill --L 32768 > "$O/sq1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$O/s2" -- python3 tools/microbench.py prefill --L 32768 > "$O/sq2.log" 2>&1
{ echo "prefill_attn_kernel<BF16,128,4>, 1 x 32768 tokens, per launch (SQ counters are in 4-cycle units summed over waves / SIMDs)"; python3 tools/pmc_sq.py "$O/s1" prefill_attn; python3 tools/pmc_sq.py "$O/s2" prefill_attn; } > "$O/${R}_prefill_sq_counters.txt"
rm -rf "$O/s1" "$O/s2"
echo "[3/6] prefill SQ counters done"
# 4. in-kernel phase timestamps of the decode kernel (debug build)
timeout -k 10 200 python3 tools/decode_ts.py > "$O/${R}_decode_phase_timestamps.txt" 2>&1
echo "[4/6] decode timestamps done"
# 5. stand-alone kernel timings
{ python3 tools/microbench.py prefill --L 16384; python3 tools/microbench.py prefill --L 32768; python3 tools/microbench.py decode --L 16640 --splits 32; python3 tools/microbench.py decode --L 65536 --B 8 --splits 4; python3 tools/microbench.py scoring --L 32768; } > "$O/${R}_microbench.txt" 2>&1
echo "[5/6] microbench done"
# 6. the bench lines: default run (C3, with cpu_baseline), C2, C4
python3 bench.py > "$O/bench_c3.log" 2>&1
grep "^{\"metric\"" "$O/bench_c3.log" | tail -1 > "$O/${R}_bench_c3.json"
python3 bench.py --workload C2 --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench_c2.log" 2>&1
grep "^{\"metric\"" "$O/bench_c2.log" | tail -1 > "$O/${R}_bench_c2.json"
python3 bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench_c4.log" 2>&1
grep "^{\"metric\"" "$O/bench_c4.log" | tail -1 > "$O/${R}_bench_c4.json"
echo "[6/6] bench lines done"
# 7. optional (COLLECT_C5=1, ~3 min): one GPU's share of BASELINE.json configs[4] - 8 sequences of 128K context
if [ "${COLLECT_C5:-0}" = "1" ]; then
  python3 bench.py --workload C5 --steps 1 --warmup 0 --no-cpu-baseline > "$O/bench_c5.log" 2>&1
  grep "^{\"metric\"" "$O/bench_c5.log" | tail -1 > "$O/${R}_bench_c5_per_gpu.json"
fi
rm -f "$O"/*.log
ls -la "$O"
