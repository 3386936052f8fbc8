#!/bin/bash
# SQ counters of the scoring kernels on the 32K scoring microbench:  sq_scoring.sh <round> [variant name]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
N=${2:-default}
if [ -n "$2" ]; then export CVLLM_LIB_PATH=$PWD/tools/dbg/variants/lib_$2.so; fi
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sqs_$N -- python3 tools/microbench.py scoring --L 32768 > $O/sqs_$N.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sqs2_$N -- python3 tools/microbench.py scoring --L 32768 > $O/sqs2_$N.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/sqs3_$N -- python3 tools/microbench.py scoring --L 32768 > $O/sqs3_$N.log 2>&1
{ for k in "snapkv_kernel<cvllm::BF16, 128, 4, false>" "snapkv_kernel<cvllm::BF16, 128, 4, true>" chunk_mass leverage_fused2; do echo "== $k"; python3 tools/pmc_sq.py $O/sqs_$N "$k"; python3 tools/pmc_sq.py $O/sqs2_$N "$k"; python3 tools/pmc_sq.py $O/sqs3_$N "$k"; done; } > $O/sqs_$N.txt
grep -h "Kernel_Name" -m1 $O/sqs_$N/*/*counter_collection.csv > /dev/null 2>&1
cut -d, -f1-20 $O/sqs_$N/*/*counter_collection.csv 2>/dev/null | grep -o "snapkv[^\"]*" | sort | uniq -c | head -5 >> $O/sqs_$N.txt
rm -rf $O/sqs_$N $O/sqs2_$N $O/sqs3_$N
cat $O/sqs_$N.txt
