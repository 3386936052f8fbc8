#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on synthetic data: prefill+decode tokens/s @ 32K context, 50% KV retention
(Compactor scoring + eviction overlapped with prefill), one MI355X per rank; plus the decode-attention kernel's
achieved HBM GB/s against the roofline and a CPU baseline of the same hot path (oracle, host cores).

    python bench.py [--gpus N --steps K --warmup W] [--workload C3|C2|C4|tiny] [--ctx L --new T --ratio r]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one generate call (reference LLM.generate -> scheduler throughput, scheduler.py:203-205):
prefill of one `ctx`-token prompt per rank with scoring/selection/compaction on the store stream, then `new`
greedy decode tokens.  Sequences shard across GPUs with no data-path collective: weak scaling.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compactor-vllm_amd"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (shell config, ctx, new tokens, method, ratio)
    "C3": ("llama", 32768, 256, "COMPACTOR", 0.5),   # BASELINE.json configs[2] — the config the metric is quoted on
    "C2": ("llama", 16384, 256, "NONE", 1.0),        # configs[1]
    "C4": ("qwen3", 32768, 256, "SNAPKV", 0.25),     # configs[3]
    "tiny": ("tiny", 2048, 16, "COMPACTOR", 0.5),
    # BASELINE.json configs[4], ONE GPU's share: 8 sequences of 128K context (130816 + 256 tokens; the sequences are
    # prefilled one after the other, then decoded as a batch).  ~1 minute per step: not part of the default run.
    "C5": ("llama", 131072 - 256, 256, "COMPACTOR", 0.5),
}


def hip_events(n):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    evs = []
    for _ in range(n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        evs.append(e)
    return hip, evs


@torch.inference_mode()
def prefill_only(llm, prompts, bcp, ratio):
    """Prefill (with scoring / eviction / reclamation exactly as `ModelRunner.generate` does it) and KEEP the sequences'
    cache rows, for the roofline leg: returns {"bm": batch rows on the device, "ids": sequence ids to free}."""
    from compactor_vllm_amd import SamplingParams, SequenceCompressionParams
    from compactor_vllm_amd.utils.arguments import build_prefill_args
    from compactor_vllm_amd.utils.sequence import Sequence

    r = llm.master_model_runner
    bms, ids_all = [], []
    pending = [Sequence(p, sampling_params=SamplingParams(0.0, 8), compression_params=SequenceCompressionParams(ratio, 16, 64))
               for p in prompts]
    while pending:  # same packing rule as the scheduler: a prefill wave holds at most max_batched_tokens
        wave, used = [], 0
        while pending and (not wave or used + pending[0].prompt_len <= r.max_batched_tokens):
            used += pending[0].prompt_len
            wave.append(pending.pop(0))
        a = build_prefill_args(wave, bcp, r.num_kv_heads, r.PHI, r.device)
        ids = [s.seq_id for s in wave]
        ok, rows = r.kv_manager.allocate_sequences(ids, (a.max_new_tokens + a.context_lens).tolist())
        assert ok
        r.run_prefill(a, rows)
        r._join_store_stream()
        r.kv_manager.reclaim_pages(ids, a.max_new_tokens.tolist())
        bms.append(rows)
        ids_all += ids
    torch.cuda.synchronize()
    return {"bm": torch.cat(bms), "ids": ids_all}


@torch.inference_mode()  # like the engine: its graphs register RNG state as inference tensors
def roofline_decode_attn(model, state, workload, rounds=5, use_graph=True, live_pmc=False):
    """Achieved HBM GB/s of decode attention AS THE REFERENCE DEFINES IT (a2 = stage 1 + split merge,
    cv/attention/sparse_decode_kernel.py:246-435) on the REAL post-prefill cache of every layer (distinct memory per
    layer, 2.2 GB per pass => cold L2 / Infinity Cache, like inside a decode step).

    The product call `head_sparse_decode_attention` is captured for all layers back to back into one HIP graph -
    nothing is switched off: on this workload it is ONE kernel per layer (decode_fused_kernel streams K/V and merges
    its splits in the same launch); on grids that oversubscribe the chip it would be that kernel plus
    decode_stage2_kernel, and both would be inside the timed region.  `rounds` replays are queued on the current stream
    between ONE pair of HIP events recorded on that stream; per-launch time = elapsed / (rounds * layers).  That
    includes the dispatch gap between consecutive launches, so it is an upper bound of the rocprofv3 kernel durations
    (profiles/) and the reported fraction a lower bound.  (Event-record nodes captured INSIDE a graph do not refresh
    the events' timestamps on this ROCm, so the events stay outside.)"""
    from compactor_vllm_amd.attention import sparse_decode_kernel as dk

    cfg, dev = model.cfg, model.dev
    bm = state["bm"]
    B = bm.numel()
    q = torch.randn(B, cfg.heads, cfg.head_dim, device=dev, dtype=torch.bfloat16)
    nl = cfg.layers
    elt = 2
    bytes_alg = []
    lens_all = [a.bh_seq_lens.index_select(0, bm).contiguous() for a in model.attn]
    for lens in lens_all:
        rows = int(lens.sum().item())
        bytes_alg.append(2 * cfg.head_dim * elt * rows + 2 * B * cfg.heads * cfg.head_dim * elt)

    def one_pass():
        for li, a in enumerate(model.attn):
            dk.head_sparse_decode_attention(q, a.k_cache, a.v_cache, lens_all[li], a.page_table, bm, cfg.kv_heads,
                                            a.page_size)

    if use_graph:
        # warm-up AND capture on one side stream: the decode workspace is per stream, so it must exist on the capture
        # stream before the capture starts (nothing is allocated or zeroed inside the captured region)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            one_pass()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                one_pass()
        torch.cuda.current_stream(dev).wait_stream(side)
        replay = graph.replay
    else:  # --no-graph (the rocprofv3 --pmc passes): the same launches issued eagerly
        one_pass()  # warm-up (workspace allocation)
        torch.cuda.synchronize()
        replay = one_pass
    # warm replays for ~50 ms before the timed ones, so that the clocks have settled under this load
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    replay()
    e1.record()
    torch.cuda.synchronize()
    first = max(e0.elapsed_time(e1) * 1e-3, 1e-5)
    for _ in range(min(200, int(0.05 / first) + 1)):
        replay()
    rounds = max(rounds, min(50, int(0.02 / first) + 1))
    e0.record()
    for _ in range(rounds):
        replay()
    e1.record()
    torch.cuda.synchronize()
    merge_ok = dk.merge_status(dev) == 0
    avg_s = e0.elapsed_time(e1) * 1e-3 / (rounds * nl)
    avg_bytes = sum(bytes_alg) / len(bytes_alg)
    achieved = avg_bytes / avg_s / 1e9
    n_splits = dk.plan_internal_splits(B * cfg.kv_heads, model.attn[0].page_table.shape[-1] * model.attn[0].page_size,
                                       None)
    fits = B * cfg.kv_heads * n_splits <= dk._cus(dev.index)
    one_kernel = n_splits == 1 or (fits and not os.environ.get("CVLLM_DECODE_MERGE", "").startswith(("t", "2")))
    traffic, traffic_src = None, "live counter pass not requested"
    if live_pmc and cfg.kv_heads == 8 and cfg.head_dim == 128 and cfg.heads == 32:
        L_rows = int(round((avg_bytes - 2 * B * 32 * 128 * 2) / (2 * 128 * 2 * B * 8)))
        traffic, traffic_src = pmc_traffic_live(B, L_rows, n_splits, int(avg_bytes))
    if traffic is None:  # fall back to the committed record of the collection run (and say which of the two it is)
        live_why = traffic_src
        traffic, traffic_src = pmc_traffic(workload, int(avg_bytes))
        traffic_src = f"{traffic_src}  [no live figure: {live_why}]"
    out = {"bound": "hbm",
           "kernel": "decode_fused_kernel (K/V streaming + in-launch split merge: the whole of reference a2)"
                     if one_kernel else "decode_fused_kernel + decode_stage2_kernel (K/V streaming, then the split merge: "
                                        "together the whole of reference a2)",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
           "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
           "splits": n_splits, "launches_per_layer": 1 if one_kernel else 2, "merge_included": True,
           "merge_status_ok": merge_ok,
           "timing": f"one HIP event pair on the launch stream around {rounds} queued "
                     + ("replays of a HIP graph holding" if use_graph else "eager passes of")
                     + f" the product's decode-attention call of all {nl} layers back to back (each on its own cache), "
                     f"divided by {rounds * nl}; includes the inter-launch dispatch gaps; ~50 ms of untimed replays "
                     f"first (clocks settled)"}
    rp = rocprof_decode(workload, int(avg_bytes))
    if rp is not None:
        out["rocprof"] = rp
    return out


def rocprof_decode(workload, alg_bytes):
    """The same launch's duration as rocprofv3 --kernel-trace saw it over THIS command (the newest committed
    profiles/*_bench_decode_rocprof.json, written by tools/collect_profiles.sh from the kernel trace of bench.py): mean
    kernel duration(s) of one layer's decode attention, summed over its launches, and the fraction they give.  A trace
    cannot be taken from inside the benchmark; None when no matching record is committed."""
    try:
        names = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_bench_decode_rocprof.json"))
        d = json.load(open(os.path.join(ROOT, "profiles", names[-1])))
        if d.get("workload") != workload or abs(d["algorithmic_bytes_per_launch"] - alg_bytes) > 0.02 * alg_bytes:
            return None
        us = float(d["mean_us_per_layer"])
        return {"avg_launch_us": round(us, 2), "frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "kernels": d.get("kernels"), "source": f"profiles/{names[-1]} ({d.get('command', '')})"}
    except Exception:  # noqa: BLE001
        return None


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)


@torch.inference_mode()
def roofline_prefill_attn(model, ctx, rounds=3):
    """Second roofline object: the prefill attention kernel (MFMA-bound) at the workload's context length on
    synthetic q/k/v of the model's head shape with an empty cache; causal FLOPs = 4 * S^2 * D * HQ / 2 per launch,
    timed with events on the current stream (the stream the kernel is launched on) around `rounds` (or more) launches
    after a warm-up long enough for the clock to settle."""
    from compactor_vllm_amd.attention.sparse_varlen_kernel import causal_sparse_varlen_with_cache

    cfg, dev = model.cfg, model.dev
    a = model.attn[0]
    q = torch.randn(ctx, cfg.heads, cfg.head_dim, device=dev, dtype=torch.bfloat16)
    k = torch.randn(ctx, cfg.kv_heads, cfg.head_dim, device=dev, dtype=torch.bfloat16)
    v = torch.randn(ctx, cfg.kv_heads, cfg.head_dim, device=dev, dtype=torch.bfloat16)
    lens = torch.zeros(1, cfg.kv_heads, dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    cu = torch.tensor([0, ctx], dtype=torch.int32, device=dev)

    def run():
        return causal_sparse_varlen_with_cache(q, k, v, a.k_cache, a.v_cache, lens, a.page_table, bm, cu, ctx, 0,
                                               cfg.kv_heads, a.page_size)

    # Warm until ~50 ms of this kernel have run: the chip's clock takes that long to settle under a matrix-pipe load (a
    # 16 K launch timed after one warm launch reads 2.30 ms, 1.98 ms once settled; profiles/r02_prefill_workgroup_stamps.txt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    first = max(e0.elapsed_time(e1) * 1e-3, 1e-5)
    for _ in range(min(40, int(0.05 / first) + 1)):
        run()
    rounds = max(rounds, min(10, int(0.03 / first) + 1))
    e0.record()
    for _ in range(rounds):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / rounds
    flops = 4.0 * ctx * ctx * cfg.head_dim * cfg.heads / 2
    tf = flops / sec / 1e12
    import os
    kern = ("prefill_attn_w4_kernel (4 waves, one per SIMD)"
            if cfg.head_dim == 128 and a.page_size % 64 == 0 and not os.environ.get("CVLLM_PREFILL", "").startswith("8")
            else "prefill_attn_kernel (8 waves)")
    return {"bound": "mfma", "kernel": kern, "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "avg_launch_us": round(sec * 1e6, 1), "algorithmic_flops_per_launch": flops,
            "note": "peak = the guide's dense bf16 figure at 2.4 GHz; profiles/r03_prefill_workgroup_stamps.txt measures "
                    "1.83 GHz shader clock inside this kernel (s_memtime / s_memrealtime; the chip is power-limited here), "
                    "i.e. 1.90 PFLOP/s at the delivered clock"}


@torch.inference_mode()
def roofline_scoring(model, ctx):
    """Third roofline object: the bandwidth-bound store-stream kernels (cache write, compaction, Compactor / SnapKV
    scoring, the fused producer) at the workload's layer shape, each timed live through its boundary call with HIP events
    on the launch stream (~30 ms of warm calls first).  `achieved` = ALGORITHMIC bytes (SURVEY 8d formulas) / live time of
    the call - an upper bound of the kernel's own time where the call launches small helpers too (z-score, casts); the
    kernel-only rocprofv3 duration and the PMC traffic of the same shape are read from the newest committed
    profiles/*_scoring_kernel_durations.json / *_scoring_pmc.json (taken at 32768 tokens; omitted for other lengths)."""
    from compactor_vllm_amd.compression.common import extract_and_store_top_kv
    from compactor_vllm_amd.compression.compactor import approximate_leverage_scores, non_causal_attn_scores
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores
    from compactor_vllm_amd.kv_cache.store_kv_cache import prefill_store_all_kv
    from compactor_vllm_amd.layers.rotary_embedding import fused_qkv_rope

    cfg, dev = model.cfg, model.dev
    HQ, HKV, D, e, L, PS = cfg.heads, cfg.kv_heads, cfg.head_dim, 2, ctx, model.attn[0].page_size
    a = model.attn[0]
    qkv = (torch.randn(L, (HQ + 2 * HKV) * D, device=dev) * 0.3).to(torch.bfloat16)
    q = qkv[:, : HQ * D].view(L, HQ, D)
    k = qkv[:, HQ * D : (HQ + HKV) * D].view(L, HKV, D)
    v = qkv[:, (HQ + HKV) * D :].view(L, HKV, D)
    PHI = (torch.randn(D, 48, device=dev) / 48 ** 0.5).to(torch.bfloat16)
    cu = torch.tensor([0, L], dtype=torch.int32, device=dev)
    bm = torch.ones(1, dtype=torch.int32, device=dev)
    zero = torch.zeros(1, HKV, dtype=torch.int32, device=dev)
    sc = torch.randn(L, HKV, device=dev)
    kept = int(round(0.5 * (L - 80) * HKV))
    retain = torch.tensor([kept], dtype=torch.int32, device=dev)
    pos = torch.arange(L, device=dev)
    cs = getattr(model, "rope_cs", None)

    def t_us(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        first = max(e0.elapsed_time(e1) * 1e-3, 2e-5)
        for _ in range(min(400, int(0.03 / first) + 1)):
            fn()
        n = max(5, min(100, int(0.01 / first) + 1))
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    calls = {
        "store_all_kernel": (4 * L * HKV * D * e, lambda: prefill_store_all_kv(
            new_keys=k, new_values=v, cu_seqlens_k=cu, max_seqlen_k=L, k_cache=a.k_cache, v_cache=a.v_cache,
            page_table=a.page_table, bh_lens=zero.clone(), batch_mapping=bm, PAGE_SIZE=PS)),
        "select + compact_store_kernel": (4 * L * HKV + 4 * (kept + 64 * HKV) * D * e, lambda: extract_and_store_top_kv(
            sc, cu, L, L * HKV, HKV, k, v, retain, a.page_table, bm, zero.clone(), a.k_cache, a.v_cache, PS)),
        "leverage_fused2_kernel": (L * HKV * D * e + 4 * L * HKV, lambda: approximate_leverage_scores(
            k, [L], PHI, normalize=False, chunk_size=512)),
        "chunk_mass_kernel": (L * (HQ + HKV) * D * e + 4 * L * HKV, lambda: non_causal_attn_scores(
            q, k, v, cu, L, chunk_size=128, sm_scale=1.0, normalize=False)),
        "snapkv_kernel (two passes)": (L * HKV * D * e + 32 * HQ * D * e + 4 * L * HKV, lambda: query_aware_key_scores(
            q, k, cu, cu, w=32, max_seqlen_k=L)),
    }
    if cs is not None:
        calls["qkv_producer_kernel"] = (2 * L * (HQ + HKV) * D * e + 4 * L * D,
                                        lambda: fused_qkv_rope(qkv, pos, cs, HQ, HKV, D))
    dur, pmc = {}, {}
    try:
        names = sorted(os.listdir(os.path.join(ROOT, "profiles")))
        dn = [p for p in names if p.endswith("_scoring_kernel_durations.json")]
        pn = [p for p in names if p.endswith("_scoring_pmc.json")]
        if L == 32768 and dn:
            dur = json.load(open(os.path.join(ROOT, "profiles", dn[-1])))
        if L == 32768 and pn:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pn[-1])))["kernels"]
    except Exception:  # noqa: BLE001
        pass
    out = {}
    for name, (alg, fn) in calls.items():
        us = t_us(fn)
        key = name.split(" ")[-1] if name.startswith("select") else name.split(" ")[0]
        rp = sum(v["mean_us"] for kname, v in dur.items() if kname.startswith(key)) or None
        tr = pmc.get(key, {}).get("hbm_bytes_per_launch")
        out[name] = {"algorithmic_bytes": int(alg), "call_us": round(us, 1), "achieved": round(alg / us / 1e3, 1),
                     "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4), "rocprof_kernel_us": rp,
                     "frac_rocprof": None if not rp else round(alg / rp / 1e3 / HBM_PEAK_GBS, 4), "traffic": tr}
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "shape": f"1 x {L} tokens, HQ {HQ} / HKV {HKV} / D {D}, bf16",
            "kernels": out,
            "note": "achieved = algorithmic bytes / live time of the boundary call (HIP events on the launch stream, "
                    "helpers included); rocprof_kernel_us / traffic from the committed profiles of the same shape "
                    "(separate rocprofv3 --kernel-trace and --pmc passes over tools/microbench.py scoring)"}


@torch.inference_mode()
def copy_bandwidth_gbs(dev, nbytes=1 << 30, rounds=5):
    """Practical HBM roof next to the 8 TB/s spec: a device-to-device copy of 1 GiB (read + write =
    2 GiB moved) timed with events on the current stream."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * rounds / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_traffic(workload, alg_bytes):
    """(HBM bytes per launch, where the figure comes from).  Counters cannot be read from inside the benchmark; the
    figure is taken from the newest committed `profiles/*_bench_pmc.json`, which tools/collect_profiles.sh writes from
    separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 rule of MI355X_MICROARCH.md, + WRITE_SIZE) OVER
    THIS COMMAND - and only if that file says it profiled bench.py on the same workload and its algorithmic bytes agree
    within 2 %.  Anything else returns null with the reason."""
    try:
        names = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_bench_pmc.json"))
        if not names:
            return None, "no profiles/*_bench_pmc.json"
        d = json.load(open(os.path.join(ROOT, "profiles", names[-1])))
        if "bench.py" not in d.get("command", ""):
            return None, f"profiles/{names[-1]} was not taken over bench.py"
        if d.get("workload") != workload:
            return None, f"profiles/{names[-1]} is for workload {d.get('workload')}"
        k = next(v for n, v in d["kernels"].items() if "decode_fused_kernel" in n)
        if abs(d["algorithmic_bytes_per_launch"] - alg_bytes) > 0.02 * alg_bytes:
            return None, f"profiles/{names[-1]}: algorithmic bytes differ"
        return int(k["hbm_bytes_per_launch"]), f"profiles/{names[-1]} ({d['command']})"
    except Exception as exc:  # noqa: BLE001
        return None, f"unreadable: {type(exc).__name__}"


def pmc_traffic_live(B, L, splits, alg_bytes, timeout_s=240):
    """(HBM bytes per launch, source) measured NOW, on this box: two child runs of `rocprofv3 --pmc <counter>
    --kernel-include-regex decode_fused` (separate passes for FETCH_SIZE and WRITE_SIZE, collection limited to the roofline
    kernel) over tools/microbench.py's decode leg - the same kernel through the same C-ABI call at this run's per-layer shape
    (B sequences, L rows per kv-head, `splits` key splits).  hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 rule
    of MI355X_MICROARCH.md).  A process cannot read these counters about itself, hence the children; they run while this
    process idles.  Returns (None, reason) if the profiler is missing, fails, or the shapes' algorithmic bytes disagree."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "this process is itself being profiled"
    micro_alg = 2 * 128 * 2 * B * 8 * L + 2 * B * 32 * 128 * 2
    if abs(micro_alg - alg_bytes) > 0.02 * alg_bytes:
        return None, "microbench shape does not reproduce this run's algorithmic bytes"
    got = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                out_dir = os.path.join(td, ctr)
                cmd = [exe, "--pmc", ctr, "--kernel-include-regex", "decode_fused", "--kernel-trace", "--output-format",
                       "csv", "-d", out_dir, "--", sys.executable, os.path.join(ROOT, "tools", "microbench.py"), "decode",
                       "--B", str(B), "--L", str(L), "--splits", str(splits)]
                r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True,
                                   timeout=timeout_s)
                if r.returncode != 0:
                    return None, f"rocprofv3 --pmc {ctr} over tools/microbench.py: rc {r.returncode}"
                vals = []
                for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                    with open(f, newline="") as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] == ctr and "decode_fused" in row["Kernel_Name"]:
                                vals.append(float(row["Counter_Value"]))
                if not vals:
                    return None, f"rocprofv3 --pmc {ctr}: no rows for decode_fused_kernel"
                got[ctr] = (sum(vals) / len(vals), len(vals))
    except Exception as exc:  # noqa: BLE001
        return None, f"live counter pass failed: {type(exc).__name__}"
    hbm = int((2 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024)
    return hbm, (f"live, this run: rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-include-regex decode_fused --kernel-trace -- "
                 f"python3 tools/microbench.py decode --B {B} --L {L} --splits {splits} (two child passes, "
                 f"{got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches; hbm = (2 x FETCH_SIZE + WRITE_SIZE) x 1024)")


def cpu_baseline(budget_s=25.0):
    """The hot path on the host cores with the CPU oracle (kind 'port'), on BASELINE.json configs[0] (C1: HQ 32 / HKV 8 /
    D 128 / page 128, ONE 4 096-token sequence, dense attention, fp16 like the reference's test shapes): one layer's
    prefill attention + cache write, then decode-attention steps over the 4 096-row cache, all timed; tokens/s of the
    attention path for a 32-layer stack = (4096 + new) / (32 * (t_prefill + new * t_decode)).  No GEMMs: the oracle
    restates only the path.  The 32 K workload of the metric is NOT timed on the CPU: `extrapolated_32k` scales the
    measured times by the operation counts (prefill x 64, decode x 8 at 50 % retention) and is labelled as such."""
    from oracle import ref_cpu as O

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a 1-GPU box's CPU share is 16 cores; more threads than cores only thrash
    torch.set_num_threads(cores)
    HQ, HKV, D, PS, Lp, layers, new = 32, 8, 128, 128, 4096, 32, 256
    g = torch.Generator().manual_seed(1234)
    q = torch.randn(Lp, HQ, D, generator=g).to(torch.float16)
    k = torch.randn(Lp, HKV, D, generator=g).to(torch.float16)
    v = torch.randn(Lp, HKV, D, generator=g).to(torch.float16)
    P = Lp // PS + 1
    cu = torch.tensor([0, Lp], dtype=torch.int32)
    lens = torch.zeros(1, HKV, dtype=torch.int32)
    bm = torch.ones(1, dtype=torch.int32)
    pt = torch.arange(2 * HKV * P, dtype=torch.int32).view(2, HKV, P)
    kc = torch.zeros(2 * HKV * P * PS, D, dtype=torch.float16)
    vc = torch.zeros_like(kc)
    t0 = time.perf_counter()
    O.prefill_attention(q, k, v, kc, vc, lens, pt, bm, cu, HKV, PS)
    O.store_all_kv(k, v, cu, kc, vc, pt, lens, bm, PS)
    t_prefill = time.perf_counter() - t0
    q1 = torch.randn(1, HQ, D, generator=g).to(torch.float16)
    nd, t1 = 0, time.perf_counter()
    while nd < 16 and time.perf_counter() - t0 < budget_s:
        O.decode_attention(q1, kc, vc, lens, pt, bm, HKV, PS)
        nd += 1
    t_dec = (time.perf_counter() - t1) / max(nd, 1)
    tok_s = (Lp + new) / (layers * (t_prefill + new * t_dec))
    # the same prefill-attention call on ONE thread (SURVEY 8d asks for both), only if the budget allows
    t_one = None
    if time.perf_counter() - t0 + 8 * t_prefill < budget_s + 20:
        torch.set_num_threads(1)
        t2 = time.perf_counter()
        O.prefill_attention(q, k, v, kc, vc, torch.zeros_like(lens), pt, bm, cu, HKV, PS)
        t_one = time.perf_counter() - t2
        torch.set_num_threads(cores)
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        cpu_model = "unknown"
    ext = (32768 + new) / (layers * (64 * t_prefill + new * 4 * t_dec))  # S^2 prefill; decode over 16 K kept rows
    return {"value": round(tok_s, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "workload": "C1 (BASELINE.json configs[0]): HQ=32 HKV=8 D=128 page=128, one 4096-token sequence, dense, fp16",
            "extrapolated_32k": round(ext, 2),
            "sample": f"CPU oracle (torch fp32, {cores} threads) of the attention path only, C1 timed in full for ONE "
                      f"layer: prefill attention + cache write {t_prefill:.2f} s for {Lp} tokens, decode attention "
                      f"{t_dec * 1e3:.1f} ms/token over the {Lp}-row cache ({nd} steps timed); value = ({Lp}+{new}) tokens "
                      f"/ ({layers} layers x (prefill + {new} decode steps)); extrapolated_32k is NOT measured: the same "
                      f"times scaled by operation count to the metric's 32768-token / 50 %-retention workload; host CPU "
                      f"{cpu_model}, {avail} cores visible"
                      + ("" if t_one is None else f"; prefill attention alone on 1 thread: {t_one:.2f} s")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="C3", choices=list(WORKLOADS))
    ap.add_argument("--ctx", type=int, default=0)
    ap.add_argument("--new", type=int, default=0)
    ap.add_argument("--ratio", type=float, default=-1.0)
    ap.add_argument("--method", default="", choices=["", "COMPACTOR", "SNAPKV", "NONE"], help="override the workload's")
    ap.add_argument("--seqs", type=int, default=1, help="sequences per GPU (the metric's configs use 1)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-multi-seq", action="store_true", help="skip the extra 4-sequences-per-GPU leg (N=1 only)")
    ap.add_argument("--serial-store", action="store_true",
                    help="A/B: run the scoring / selection / compaction chain on the main stream instead of the store stream")
    ap.add_argument("--no-roofline", action="store_true", help="A/B runs: only the tokens/s line")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="roofline.traffic from the committed profiles/*_bench_pmc.json instead of two live rocprofv3 --pmc "
                         "child passes (~40 s)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    # One rank per GPU.  Rehearsal on a box with fewer GPUs than ranks (CVLLM_DIST_BACKEND=gloo): ranks wrap around the
    # visible devices and the two scalar collectives run on the host - RCCL refuses two ranks on one device.
    backend = os.environ.get("CVLLM_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit(f"{world} ranks but {ndev} visible GPU(s)")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL; used only for the timing barrier / max-reduce

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import bench_shell as bs
    from compactor_vllm_amd import (LLM, BatchCompressionParams, CompressionMethod, LLMConfig, SamplingParams,
                                    SequenceCompressionParams)
    shared_device = world > ndev
    if shared_device:
        # rehearsal only: ranks that share a GPU time-slice its CUs, so a launch's workgroups are not co-resident and the
        # in-launch split merge of decode attention would time out (and say so: cvllm_decode_merge_status) - use the
        # two-kernel merge, which has no such requirement
        import compactor_vllm_amd.attention.sparse_decode_kernel as _dk

        _dk.set_merge_mode("two-kernel")

    shape, ctx, new, method_name, ratio = WORKLOADS[args.workload]
    ctx = args.ctx or ctx
    new = args.new or new
    ratio = args.ratio if args.ratio >= 0 else ratio
    method_name = args.method or method_name
    method = CompressionMethod[method_name]
    cfg = {"llama": bs.LLAMA31_8B, "qwen3": bs.QWEN3_8B, "tiny": bs.TINY}[shape]
    nseq = max(1, args.seqs) if args.workload != "C5" or args.seqs > 1 else 8
    # N = 1 only: an extra, untimed-for-`value` leg repeats the workload with MULTI sequences per GPU (what a serving
    # batch looks like); the KV pool is sized for it up front, which does not change the 1-sequence timings
    MULTI = 4
    multi_leg = world == 1 and nseq == 1 and not args.no_multi_seq and args.workload != "tiny"
    max_seqs = MULTI if multi_leg else nseq
    model = bs.ModelShell(cfg, dev, max_model_len=ctx + new, seed=0)
    # The product's engine drives the shell model: scheduler, paged KV cache (pages for the FULL uncompressed length
    # are reserved before prefill and reclaimed after compaction, like the reference), prefill with scoring + eviction
    # on the store stream, continuous-batching decode on HIP graphs.  One packed prefill holds up to 256 K tokens;
    # longer batches (C5) are prefilled sequence by sequence and decoded together (scheduler.py:65-108).
    page = 128
    conf = LLMConfig(model=cfg.name, max_num_seqs=max_seqs, max_model_len=ctx + new, hf_config=model.hf_config, eos=-1,
                     kvcache_page_size=page, enforce_eager=args.no_graph, show_progress_bar=False)
    llm = LLM(conf, model, device=dev, num_pages=max_seqs * cfg.kv_heads * (-(-(ctx + new) // page)) + 8,
              max_batched_tokens=max(ctx, min(nseq * ctx, 262144)))
    if args.serial_store:
        llm.master_model_runner.store_stream = None  # maybe_execute_in_stream(STORE_STREAM=None) runs inline
    # The job's request list is GLOBAL (world x nseq requests, request i drawn from seed 1 + i) and every rank takes its
    # share by the longest-processing-time-first partition on L^2 + ratio * L * new (bench_dist.partition_lpt: the same
    # deterministic partition on every rank, nothing communicated) - sequences shard across replicas, no collective.
    import bench_dist

    n_req = world * nseq
    costs = [bench_dist.request_cost(ctx, new, ratio)] * n_req
    mine = bench_dist.partition_lpt(costs, world)[rank]
    assert len(mine) == nseq
    prompts = [torch.randint(0, cfg.vocab, (ctx,), generator=torch.Generator().manual_seed(1 + i)).tolist() for i in mine]
    g = torch.Generator().manual_seed(1000 + rank)
    # `new` tokens per sequence: the one sampled from the prefill logits + (new - 1) decode steps (quirk Q11 of the
    # reference's loop: max_new_tokens counts decode steps)
    sampling = SamplingParams(temperature=0.0, max_new_tokens=new - 1)
    bcp = BatchCompressionParams(compression_method=method)

    merge_fallbacks = []

    def run(ps):
        try:
            out = llm.generate(ps, sampling, bcp, per_sequence_compression_params=[
                SequenceCompressionParams(ratio, protected_first_tokens=16, protected_last_tokens=64) for _ in ps])
        except RuntimeError as exc:
            # The in-launch decode merge found its workgroups not co-resident (another process on this GPU): the engine has
            # raised - the call's tokens are invalid - and switched the process to the two-kernel merge.  Give the call's
            # pages back and run the step again, ONCE; the line reports it (config.decode_merge_fallback).
            if "in-launch split merge timed out" not in str(exc) or merge_fallbacks:
                raise
            merge_fallbacks.append(str(exc)[:120])
            kvm = llm.master_model_runner.kv_manager
            kvm.free_sequences(list(kvm.seq_id_to_batch))
            out = llm.generate(ps, sampling, bcp, per_sequence_compression_params=[
                SequenceCompressionParams(ratio, protected_first_tokens=16, protected_last_tokens=64) for _ in ps])
        assert all(len(o) == new for o in out)
        return out

    def step():
        return run(prompts)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tokens_per_rank = args.steps * nseq * (ctx + new)
    value = world * tokens_per_rank / elapsed
    result = {
        "metric": "prefill+decode tokens/sec @ 32K ctx, 50% KV retention, 1xMI355X; decode-attn HBM GB/s",
        "value": round(value, 1),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {cfg.name} random weights, {ctx}-token prefill + {new} decode per sequence, "
                        f"{method_name} ratio {ratio} (protected 16/64, chunk 512), {nseq} sequence(s) per GPU, "
                        f"{'scoring+eviction serial on the main stream (A/B)' if args.serial_store else 'store-stream overlapped scoring+eviction'}, "
                        f"{'eager' if args.no_graph else 'HIP-graph'} decode, driven by the product's engine "
                        f"(LLM.generate: scheduler, paged KV cache, continuous batching)",
            "ctx": ctx, "new_tokens": new, "method": method_name, "ratio": ratio, "sequences_per_gpu": nseq,
            "parallelism": f"replicas x{world} (a global list of {n_req} requests sharded by bench_dist.partition_lpt, "
                           f"no collectives)"
                           + (f"; REHEARSAL: {world} ranks share {ndev} GPU(s) (gloo), two-kernel decode merge" if shared_device else ""),
            "decode_merge_fallback": bool(merge_fallbacks),
        },
    }
    if rank == 0 and args.no_roofline:
        print(json.dumps(result), flush=True)
    elif rank == 0:
        state = prefill_only(llm, prompts, bcp, ratio)  # a prefill whose cache stays allocated: the real cache
        result["roofline"] = roofline_decode_attn(model, state, args.workload, use_graph=not args.no_graph,
                                                  live_pmc=(world == 1 and not args.no_live_pmc))
        copy_bw = copy_bandwidth_gbs(dev)
        result["roofline"]["copy_bw"] = round(copy_bw, 1)  # measured device copy rate: the practical HBM roof
        result["roofline"]["frac_of_copy_bw"] = round(result["roofline"]["achieved"] / copy_bw, 4)
        try:
            result["roofline_prefill"] = roofline_prefill_attn(model, ctx)
        except Exception as exc:  # noqa: BLE001 - secondary object, see above
            result["roofline_prefill"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        try:
            result["roofline_scoring"] = roofline_scoring(model, ctx)
        except Exception as exc:  # noqa: BLE001 - secondary object
            result["roofline_scoring"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        llm.master_model_runner.kv_manager.free_sequences(state["ids"])
        if multi_leg:
            # an extra leg must never cost the main line: anything going wrong here is reported, not raised
            try:
                mp = prompts + [torch.randint(0, cfg.vocab, (ctx,), generator=g).tolist() for _ in range(MULTI - 1)]
                torch.cuda.empty_cache()  # the 1-sequence legs leave the caching allocator fragmented for 4x tensors
                run(mp)
                torch.cuda.synchronize()
                dts = []  # two timed steps, the faster one is reported (the first 4x-sized step after the 1-sequence
                for _ in range(2):  # legs still pays caching-allocator growth)
                    t1 = time.perf_counter()
                    run(mp)
                    torch.cuda.synchronize()
                    dts.append(time.perf_counter() - t1)
                dt = min(dts)
                mstate = prefill_only(llm, mp, bcp, ratio)
                mr = roofline_decode_attn(model, mstate, "-", use_graph=not args.no_graph)
                llm.master_model_runner.kv_manager.free_sequences(mstate["ids"])
                result["multi_sequence"] = {
                    "sequences_per_gpu": MULTI, "value": round(MULTI * (ctx + new) / dt, 1), "unit": "tokens/s",
                    "ms_per_step": round(dt * 1e3, 2), "steps": 2, "ms_per_step_all": [round(x * 1e3, 1) for x in dts],
                    "roofline": {k: mr[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                    "avg_launch_us", "algorithmic_bytes_per_launch")},
                    "note": "same workload with 4 sequences per GPU (packed varlen prefill, batched decode): not the "
                            "metric's configuration, shown because the decode-attention launch is then 273 MB instead "
                            "of 68 MB",
                }
            except Exception as exc:  # noqa: BLE001
                result["multi_sequence"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline()
            except Exception as exc:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port",
                                          "sample": f"failed: {type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
