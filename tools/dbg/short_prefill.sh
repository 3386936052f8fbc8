mkdir -p gpurun_out/r03_short
{ echo "# tools/microbench.py prefill --L <L> --B <B>  (HQ 32 / HKV 8 / D 128 / page 128, bf16, no cached prefix), one box;"; echo "# w4 = default dispatch (4-wave kernel, round 3: optimistic probabilities + LDS-DMA), w8 = CVLLM_PREFILL=8wave"
for L in 512 2048 4096 8192; do for B in 1 8; do
 echo -n "w4 "; python tools/microbench.py prefill --L $L --B $B 2>&1 | grep "prefill B"
 echo -n "w8 "; CVLLM_PREFILL=8wave python tools/microbench.py prefill --L $L --B $B 2>&1 | grep "prefill B"
done; done; } > gpurun_out/r03_short/r03_prefill_short_sequences.txt
cat gpurun_out/r03_short/r03_prefill_short_sequences.txt
