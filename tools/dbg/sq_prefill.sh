#!/bin/bash
# SQ counters of the prefill kernel (two passes of 8 counters) for the product library or CVLLM_LIB_PATH
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/s1" -- python3 tools/microbench.py prefill --L 32768 > "$O/sq1.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d "$O/s2" -- python3 tools/microbench.py prefill --L 32768 > "$O/sq2.log" 2>&1
{ echo "prefill attention (prefill_attn_w4_kernel<BF16,4>), 1 x 32768 tokens, per launch (SQ counters are in 4-cycle units summed over waves / SIMDs)"; python3 tools/pmc_sq.py "$O/s1" prefill_attn; python3 tools/pmc_sq.py "$O/s2" prefill_attn; } > "$O/prefill_sq_counters.txt"
rm -rf "$O/s1" "$O/s2"
cat "$O/prefill_sq_counters.txt"
