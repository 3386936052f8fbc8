#!/bin/bash
# SQ counters + clock of one variant library on the 32K prefill microbench:  sq_variant.sh <round> <variant name>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
export CVLLM_LIB_PATH=$PWD/tools/dbg/variants/lib_$2.so
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq_$2 -- python3 tools/microbench.py prefill --L 32768 > $O/sq_$2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq2_$2 -- python3 tools/microbench.py prefill --L 32768 > $O/sq2_$2.log 2>&1
{ echo "== $2"; python3 tools/pmc_sq.py $O/sq_$2 prefill_attn; python3 tools/pmc_sq.py $O/sq2_$2 prefill_attn; python3 tools/prof_summary.py $O/sq2_$2 prefill; } > $O/sq_$2.txt
rm -rf $O/sq_$2 $O/sq2_$2
cat $O/sq_$2.txt
