"""Audit of the asm-owned registers of prefill_attn_w4_kernel (csrc/prefill_attn.hip: pv_mfma_w, P4_VNM ...).

The kernel keeps its 128 output accumulators in a[128:255] and its softmax state (reference points, row sums, packed P
words) in v[222:255] by naming them literally inside inline asm; the compiler must never allocate those registers itself.
This script reads the assembly hipcc emits for the file (-S) and fails if any instruction OUTSIDE an ;;#ASMSTART /
;;#ASMEND pair of such a kernel names an AGPR >= 128 or an arch VGPR >= 222, if the kernel spills, or if it uses
scratch.  build.py runs it whenever prefill_attn.hip is rebuilt.

    python tools/audit_acc_regs.py file.s
"""
import re
import sys

ACC0 = 128
VGPR0 = 222


def audit(path: str) -> list:
    problems = []
    text = open(path).read()
    for m in re.finditer(r"^(_ZN5cvllm22prefill_attn_w4_kernel\w+):\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        in_asm = False
        for ln, line in enumerate(body.split("\n")):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm or not t or t.startswith(";") or t.startswith("."):
                continue
            code = t.split(";")[0]
            for r in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
                hi = int(r.group(2)) if r.group(2) is not None else int(r.group(3))
                if hi >= ACC0:
                    problems.append(f"{name}: compiler instruction touches an asm-owned AGPR: {t}")
            for r in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", code):
                hi = int(r.group(2)) if r.group(2) is not None else int(r.group(3))
                if hi >= VGPR0:
                    problems.append(f"{name}: compiler instruction touches an asm-owned VGPR: {t}")
    for m in re.finditer(r"\.amdhsa_kernel (_ZN5cvllm22prefill_attn_w4_kernel\w+)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", m.group(2)).group(1))
        if scratch != 0:
            problems.append(f"{m.group(1)}: {scratch} bytes of scratch (a spill could go through the owned registers)")
    return problems


if __name__ == "__main__":
    p = audit(sys.argv[1])
    for x in p[:20]:
        print("AUDIT FAIL:", x)
    print(f"audit_acc_regs: {len(p)} problem(s)")
    sys.exit(1 if p else 0)
