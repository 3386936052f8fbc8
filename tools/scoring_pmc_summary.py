"""HBM traffic of the bandwidth-bound store-stream kernels against their algorithmic bytes (<= 1.15 x).

    python tools/scoring_pmc_summary.py <raw.json from tools/pmc_summary.py> [--L 32768] > profiles/rNN_scoring_pmc.json

Raw counters: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `tools/microbench.py scoring --L 32768
--with-producer`, collection limited to the listed kernels; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 rule
of MI355X_MICROARCH.md).  Algorithmic bytes per launch at B = 1, HQ 32 / HKV 8 / D 128 / bf16 (SURVEY 8d formulas).
"""
import json
import sys


def main():
    raw = json.load(open(sys.argv[1]))
    L = int(sys.argv[sys.argv.index("--L") + 1]) if "--L" in sys.argv else 32768
    N, HQ, HKV, D, e, w = L, 32, 8, 128, 2, 32
    kept = round(0.5 * (L - 80) * HKV) + HKV * 64  # retained pairs + on average half a page of padding per head
    alg = {
        "store_all_kernel": (4 * N * HKV * D * e, "4 N HKV D elt (K and V rows read and written once)"),
        "compact_store_kernel": (4 * kept * D * e + 4 * kept, "4 N_kept D elt + 4 N_kept"),
        "leverage_fused2_kernel": (N * HKV * D * e + 4 * N * HKV, "N HKV D elt + 4 N HKV"),
        "chunk_mass_kernel": (N * (HQ + HKV) * D * e + 4 * N * HKV, "N (HQ + HKV) D elt + 4 N HKV"),
        "snapkv_kernel": (N * HKV * D * e + w * HQ * D * e + 4 * N * HKV, "N HKV D elt + w HQ D elt + 4 N HKV, BOTH passes together"),
        "qkv_producer_kernel": (2 * N * (HQ + HKV) * D * e + 4 * N * D, "2 N (HQ + HKV) D elt + 4 N D (cos / sin rows)"),
    }
    out = {"workload": f"tools/microbench.py scoring --L {L} --with-producer (B = 1, HQ 32, HKV 8, D 128, bf16)",
           "rule": "hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; Infinity-Cache hits are counted (guide, section HBM)",
           "kernels": {}}
    for key, (a, formula) in alg.items():
        rows = {k: v for k, v in raw.items() if key in k}
        if not rows:
            continue
        hbm = sum(v["hbm_bytes_per_launch"] for v in rows.values())
        out["kernels"][key] = {"algorithmic_bytes": int(a), "formula": formula, "hbm_bytes_per_launch": int(hbm),
                               "ratio": round(hbm / a, 3),
                               "fetch_KiB": {k.split("cvllm::")[-1]: v["FETCH_SIZE_KiB_avg"] for k, v in rows.items()},
                               "write_KiB": {k.split("cvllm::")[-1]: v["WRITE_SIZE_KiB_avg"] for k, v in rows.items()}}
    sel = {k.split("cvllm::")[-1]: v["hbm_bytes_per_launch"] for k, v in raw.items() if "::s" in k and "hist" in k or "sj_count" in k or "sh_write" in k}
    out["selection_kernels_hbm_bytes_per_launch"] = sel
    out["selection_note"] = "every selection pass re-reads the 1 MiB score tensor (4 N HKV); 8 launches, L2 / Infinity-Cache resident"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
