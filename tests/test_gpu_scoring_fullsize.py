"""GPU: the scoring kernels at BASELINE.json's full sizes against INDEPENDENT evaluations.

* configs[3] (C4): SnapKV at the Qwen3-8B head shape, one 32 768-token sequence - the 256-tile row-LSE reduce path.
* configs[4] (C5): Compactor leverage scores, chunk attention mass, z-scores and the blend at 130 816 tokens x 8 kv
  heads (256 leverage chunks, 1 022 attention chunks, one 1 M-element z-score segment).

Each result is checked (a) against a plain fp32 torch evaluation of the closed form ON THE GPU over the whole tensor
(torch matmul / softmax / linalg - none of this package's kernels), and (b) against the CPU oracle
(oracle/ref_cpu.py, pinned to the reference by tests/golden) on whole sampled chunks / the whole sequence where the
oracle finishes in seconds.  References: cv/compression/snapkv.py:160-276, cv/compression/compactor.py:113-221,
:338-599."""
import math

import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _torch_snapkv(q, k, w, pool=5, tile=128):
    """fp32 on the device: rows = last w queries x G heads per kv head; softmax over keys [0, L-w); column sums;
    trailing `pool`-tap mean clipped at `tile` boundaries; last w keys +inf."""
    N, HQ, D = q.shape
    HKV = k.shape[1]
    G = HQ // HKV
    keff = N - w
    out = torch.full((N, HKV), float("inf"), dtype=torch.float32, device=q.device)
    i = torch.arange(keff, device=q.device)
    lo = torch.maximum((i // tile) * tile, i - (pool - 1))
    for h in range(HKV):
        qq = q[N - w :, h * G : (h + 1) * G].float().reshape(w * G, D)
        kk = k[:keff, h].float()
        p = torch.softmax((qq @ kk.T) / math.sqrt(D), dim=-1)
        s = p.sum(0)
        cs = torch.cat([torch.zeros(1, device=q.device, dtype=torch.float64), s.double().cumsum(0)])
        win = (cs[i + 1] - cs[lo]).float()
        out[:keff, h] = win / (i - lo + 1).float()
    return out


def test_snapkv_c4_32k_qwen3_shape(dev):
    """configs[3]: 32 768 keys, HQ 32 / HKV 8 / D 128, w = 32 (query-aware scoring path of SnapKV 25 %)."""
    from compactor_vllm_amd.compression.snapkv import query_aware_key_scores

    N, HQ, HKV, D, w = 32768, 32, 8, 128, 32
    g = torch.Generator(device=dev).manual_seed(44)
    q = torch.randn(N, HQ, D, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(N, HKV, D, device=dev, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0, N], dtype=torch.int32, device=dev)
    out = query_aware_key_scores(q, k, cu, cu, w=w, max_seqlen_k=N)
    torch.cuda.synchronize()
    ref = _torch_snapkv(q, k, w)
    assert torch.isinf(out[N - w :]).all() and torch.isfinite(out[: N - w]).all()
    d = (out[: N - w] - ref[: N - w]).abs().max()
    assert torch.allclose(out[: N - w], ref[: N - w], rtol=2e-4, atol=2e-5), float(d)
    # every row of P sums to one: the scores of a head sum to w*G before pooling; pooling keeps that within the
    # clipped-window edge effect
    tot = out[: N - w].sum(0)
    assert torch.allclose(tot, torch.full_like(tot, float(w * HQ // HKV)), rtol=2e-2)
    # the CPU oracle on the whole sequence (pinned to the reference's vectors in tests/golden)
    orc = O.snapkv_scores(q.cpu(), k.cpu(), cu.cpu(), cu.cpu(), w)
    fin = torch.isfinite(orc)
    assert torch.equal(fin, torch.isfinite(out.cpu()))
    assert torch.allclose(out.cpu()[fin], orc[fin], rtol=2e-4, atol=2e-5), float((out.cpu()[fin] - orc[fin]).abs().max())


def _torch_leverage_raw(k, PHI, chunks, reg=5e-3):
    """fp32/fp64 closed form on the device for ALL chunks: x_i^T (Xc^T Xc + reg I)^-1 x_i, rows centred per chunk."""
    N, H, D = k.shape
    X = torch.matmul(k.float().transpose(0, 1), PHI.float())  # [H, N, r]
    r = X.shape[-1]
    out = torch.empty(N, H, dtype=torch.float32, device=k.device)
    eye = torch.eye(r, dtype=torch.float64, device=k.device)
    s = 0
    for L in chunks:
        Xc = X[:, s : s + L].double()
        Xc = Xc - Xc.mean(dim=-2, keepdim=True)
        Gm = Xc.transpose(-1, -2) @ Xc + reg * eye
        Gi = torch.linalg.inv(Gm.cpu()).to(k.device)  # 8 matrices of 48 x 48: host LAPACK, independent of rocSOLVER
        out[s : s + L] = ((Xc @ Gi) * Xc).sum(-1).clamp_min(0).float().T
        s += L
    return out


def _zscore_chunks(x, chunks):
    out = torch.empty_like(x, dtype=torch.float32)
    s = 0
    for L in chunks:
        seg = x[s : s + L].float()
        m = seg.mean()
        var = ((seg * seg).mean() - m * m).clamp_min(0)
        out[s : s + L] = (seg - m) / var.sqrt()
        s += L
    return out


def _torch_chunk_mass(q, k, chunk=128, q_tile=64):
    N, HQ, D = q.shape
    HKV = k.shape[1]
    G = HQ // HKV
    out = torch.empty(N, HKV, dtype=torch.float32, device=q.device)
    nfull = N // chunk
    step = 64  # chunks per batch
    for c0 in range(0, nfull, step):
        c1 = min(nfull, c0 + step)
        qq = q[c0 * chunk : c1 * chunk].float().view(c1 - c0, chunk, HKV, G, D).permute(0, 2, 1, 3, 4)
        qq = qq.reshape(c1 - c0, HKV, chunk * G, D)
        kk = k[c0 * chunk : c1 * chunk].float().view(c1 - c0, chunk, HKV, D).permute(0, 2, 1, 3)
        p = torch.softmax(qq @ kk.transpose(-1, -2), dim=-1)  # sm_scale = 1.0 (quirk Q3)
        out[c0 * chunk : c1 * chunk] = p.sum(-2).permute(0, 2, 1).reshape((c1 - c0) * chunk, HKV)
    M = N - nfull * chunk
    if M:
        s = nfull * chunk
        pad_rows = -(-M // q_tile) * q_tile - M
        for h in range(HKV):
            qq = q[s:, h * G : (h + 1) * G].float().reshape(M * G, D)
            p = torch.softmax(qq @ k[s:, h].float().T, dim=-1)
            out[s:, h] = p.sum(0) + G * pad_rows / float(chunk)
    return out


def test_compactor_scoring_c5_128k(dev):
    """configs[4] per-sequence size: 130 816 tokens x 8 kv heads, bf16.  Leverage (raw and z-scored per chunk), chunk
    attention mass, z-score per sequence + 0.5 * pre-scores, protected fills."""
    from compactor_vllm_amd.compression.compactor import approximate_leverage_scores, non_causal_attn_scores

    N, HQ, HKV, D = 131072 - 256, 32, 8, 128
    dtype = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(55)
    # keys with a per-token scale so that leverage scores are far from uniform (a flat score field would make the
    # z-scored comparison vacuous)
    scale = (0.25 + torch.rand(N, 1, 1, device=dev, generator=g) * 1.5)
    k = (torch.randn(N, HKV, D, device=dev, generator=g) * scale).to(dtype)
    q = (torch.randn(N, HQ, D, device=dev, generator=g) * 0.3).to(dtype)
    PHI = (torch.randn(D, 48, device=dev, generator=g) / math.sqrt(48)).to(dtype)
    chunks = O.split_into_chunks([N], 512)
    assert len(chunks) == 256 and chunks[-1] == 256

    # ---- a5 raw, fp32 out of the kernel's pipeline is rounded to bf16 once
    raw = approximate_leverage_scores(k, [N], PHI, normalize=False, chunk_size=512)
    pre = approximate_leverage_scores(k, [N], PHI, normalize=True, chunk_size=512)
    torch.cuda.synchronize()
    assert raw.dtype == dtype and pre.dtype == dtype and raw.shape == (N, HKV)
    ref_raw = _torch_leverage_raw(k, PHI, chunks)
    d_raw = (raw.float() - ref_raw).abs().max()
    assert d_raw <= 2.0 ** -8, float(d_raw)  # scores in [0, 1): half a bf16 ulp at 1 = 2^-9 plus fp32-vs-fp64 slack
    ref_pre = _zscore_chunks(ref_raw.to(dtype), chunks)  # the reference z-scores the ROUNDED scores (compactor.py:216-220)
    d_pre = (pre.float() - ref_pre).abs().max()
    assert d_pre <= 0.1, float(d_pre)  # SURVEY P3 bar for 16-bit pre-scores
    # the CPU oracle on whole chunks: first, one in the middle, the 256-row tail
    for ci in (0, 137, 255):
        s = ci * 512
        L = chunks[ci]
        orc = O.leverage_scores(k[s : s + L].cpu(), [L], PHI.cpu(), normalize=True, chunk_size=512)
        assert torch.allclose(pre[s : s + L].cpu().float(), orc.float(), rtol=0, atol=0.1), ci
        orc_raw = O.leverage_scores(k[s : s + L].cpu(), [L], PHI.cpu(), normalize=False, chunk_size=512)
        assert torch.allclose(raw[s : s + L].cpu().float(), orc_raw.float(), rtol=0, atol=0.01), ci

    # ---- a7 raw mass, then the engine's call (z-score per sequence + 0.5 * pre, protected 16 / 64)
    cu = torch.tensor([0, N], dtype=torch.int32, device=dev)
    mass = non_causal_attn_scores(q, k, k, cu, N, chunk_size=128, sm_scale=1.0, normalize=False)
    torch.cuda.synchronize()
    ref_mass = _torch_chunk_mass(q, k)
    d_m = (mass - ref_mass).abs().max()
    assert torch.allclose(mass, ref_mass, rtol=2e-4, atol=2e-4), float(d_m)
    for c0 in (0, 500 * 128, 1021 * 128):  # CPU oracle on whole 128-token chunks (chunks are independent)
        sl = slice(c0, c0 + 128)
        orc = O.chunk_attn_mass(q[sl].cpu(), k[sl].cpu(), [0, 128], 128, 1.0)
        assert torch.allclose(mass[sl].cpu(), orc, rtol=2e-4, atol=2e-4), c0
    out = non_causal_attn_scores(q, k, k, cu, N, chunk_size=128, sm_scale=1.0, normalize=True, accum_scores=pre,
                                 context_lens=[N], protected_first_tokens=[16], protected_last_tokens=[64],
                                 accum_blending=0.5)
    torch.cuda.synchronize()
    assert torch.isinf(out[:16]).all() and torch.isinf(out[-64:]).all()
    seg = ref_mass.double()
    m = seg.mean()
    var = ((seg * seg).mean() - m * m).clamp_min(0)
    ref_out = ((seg - m) / var.sqrt()).float() + 0.5 * pre.float()  # blended with the kernel's own pre-scores (P3)
    d_o = (out[16:-64] - ref_out[16:-64]).abs().max()
    assert torch.allclose(out[16:-64], ref_out[16:-64], rtol=2e-4, atol=2e-3), float(d_o)


def test_zscore_one_million_element_segment(dev):
    """a6 at the C5 segment size (130 816 rows x 8 heads in ONE segment, multi-workgroup path) and a ragged mix, fp32:
    against a float64 evaluation on the device; deterministic across runs."""
    from compactor_vllm_amd.compression.compactor import zscore_segments_

    g = torch.Generator(device=dev).manual_seed(6)
    lens = [131072 - 256, 513, 1, 40000]
    cu = [0]
    for n in lens:
        cu.append(cu[-1] + n)
    x = torch.randn(cu[-1], 8, device=dev, generator=g) * 3 + 1
    cud = torch.tensor(cu, dtype=torch.int32, device=dev)
    a = zscore_segments_(x.clone(), cud)
    b = zscore_segments_(x.clone(), cud)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    for i, n in enumerate(lens):
        seg = x[cu[i] : cu[i + 1]].double()
        m = seg.mean()
        var = ((seg * seg).mean() - m * m).clamp_min(0)
        if n == 1:
            continue  # one row x 8 heads still has var > 0; compare below
        ref = ((seg - m) / var.sqrt()).float()
        assert torch.allclose(a[cu[i] : cu[i + 1]], ref, rtol=0, atol=5e-5), i
