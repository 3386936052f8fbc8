"""Build libcvllm_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python compactor-vllm_amd/build.py [--force]

Each csrc/*.hip is compiled to an object (in parallel), then linked into
compactor_vllm_amd/libcvllm_hip.so.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "compactor_vllm_amd", "libcvllm_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# prefill_attn.hip: row maxima over MFMA results; with NaNs honoured hipcc canonicalises every operand (v_max x,x)
# decode_attn.hip: first kernel arguments preloaded into SGPRs at wave launch (the decode kernel's fixed cost is a
# chain of dependent memory round trips; the kernarg fetch is the first of them)
# prefill_attn.hip, 4-wave kernel: MFMA results in arch VGPRs (hipcc otherwise selects the ACC-register form for a kernel
# allowed more than 256 registers and copies every logit back with v_accvgpr_read), and no SLP vectorisation (it pairs
# the row-sum adds of the two query blocks into v_pk_add_f32 and moves them out of their hand-placed MFMA shadows)
FILE_FLAGS = {"prefill_attn.hip": ["-fno-honor-nans", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
              "decode_attn.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=16"]}


AUDITED = {"prefill_attn.hip"}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "cvllm.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        if force or _newer([src] + hdrs, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        if os.path.basename(src) in AUDITED:
            # asm-owned accumulator registers (prefill_attn.hip, pv_mfma): the same compile to assembly, audited
            asm = obj[:-2] + ".s"
            subprocess.run([_hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "--cuda-device-only", "-S",
                            src, "-o", asm], check=True)
            subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "audit_acc_regs.py"), asm],
                           check=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(objs, LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
