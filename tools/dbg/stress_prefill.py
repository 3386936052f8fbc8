import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "compactor-vllm_amd")
import torch
import test_gpu_random_sweeps as T
dev = torch.device("cuda:0")
bad = 0
for seed in range(16, 112):
    try:
        T.test_prefill_random(dev, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL", seed, str(e)[:200])
print("prefill stress done, failures:", bad)
