"""Constants shared with the reference (`compactor_vllm/config/constants.py:3`)."""

RESERVED_BATCH = 0  # batch row 0 of the page table is the padding row of captured decode batches
