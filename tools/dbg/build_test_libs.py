"""Debug builds that only tests load (never the product package): built by __graft_entry__.build(), in-tree so that they
travel to the GPU box.

    python tools/dbg/build_test_libs.py

* libcvllm_sel_withhold.so - select.hip alone with -DCVLLM_SEL_WITHHOLD: slice 0 of every column of the per-head
  ordered write never publishes its counts and the look-back wait gives up after 2 ms, so that the error path
  (sticky word -> cvllm_select_status, in-bounds clamped lists) can be driven on purpose
  (tests/test_gpu_round3_corners.py).
* libcvllm_dec_withhold.so - the WHOLE library with decode_attn.hip compiled -DCVLLM_DEC_WITHHOLD: in the in-launch split
  merge split 1 never sends its numerators and the merging workgroups give up after 2 ms (NaN slice + sticky error
  word), so that the engine's per-loop health check and its fallback to the two-kernel merge can be driven on purpose
  (tests/test_gpu_engine.py, in a child process started with CVLLM_LIB_PATH pointing here).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CS = os.path.join(ROOT, "compactor-vllm_amd", "csrc")
SEL_WITHHOLD = os.path.join(HERE, "libcvllm_sel_withhold.so")
DEC_WITHHOLD = os.path.join(HERE, "libcvllm_dec_withhold.so")
OBJ = os.path.join(ROOT, "compactor-vllm_amd", "build")


def build(force: bool = False) -> list:
    out = []
    src = os.path.join(CS, "select.hip")
    deps = [src, os.path.join(CS, "common.h"), os.path.join(ROOT, "include", "cvllm.h")]
    if force or not os.path.exists(SEL_WITHHOLD) or any(os.path.getmtime(d) > os.path.getmtime(SEL_WITHHOLD) for d in deps):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-DCVLLM_SEL_WITHHOLD", src, "-o", SEL_WITHHOLD]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    out.append(SEL_WITHHOLD)
    src = os.path.join(CS, "decode_attn.hip")
    deps = [src, os.path.join(CS, "common.h"), os.path.join(ROOT, "include", "cvllm.h")]
    objs = [os.path.join(OBJ, f) for f in sorted(os.listdir(OBJ)) if f.endswith(".o") and f != "decode_attn.o"]
    if force or not os.path.exists(DEC_WITHHOLD) or any(os.path.getmtime(d) > os.path.getmtime(DEC_WITHHOLD) for d in deps + objs):
        dbg_obj = os.path.join(HERE, "decode_attn_withhold.o")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DCVLLM_DEC_WITHHOLD",
               "-mllvm", "-amdgpu-kernarg-preload-count=16", "-c", src, "-o", dbg_obj]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, dbg_obj, "-o", DEC_WITHHOLD]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    out.append(DEC_WITHHOLD)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
