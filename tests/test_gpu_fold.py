"""LayerNorm folded into the projection it feeds (include/d3pm_hip.h: d3pm_fold_block, d3pm_tuning.ln_fold;
csrc/d3pm_mfma_tile.h EPI_LNF / EPI_STATS): the reference applies every LayerNorm of a DiT block directly in front of a Linear
(/root/reference/vall_e/vall_e/ar_discrete.py:131-132, 136-142, 145-159), so LN(x) W^T + b = rstd (x W'^T - mean s) + b'.

Checked here, kernel by kernel and then through the denoiser:
  * the row moments a producing projection leaves behind are the moments of the rows it stored, bit for bit, whatever tile
    geometry ran (so a result never depends on the schedule, i.e. on the batch size);
  * the folded projection against an fp32 evaluation of LayerNorm -> Linear (torch), beside the unfolded HIP launches: it must be at
    least as close (it skips one 16-bit rounding), with FiLM + GELU as in fc1, for rows with a large common offset too;
  * one denoiser evaluation folded vs unfolded (logits within the storage-type noise both are allowed against the oracle), and a
    shard of a batch against the unsplit batch bit for bit (regime_batch).
"""
import pytest
import torch
import torch.nn.functional as F

from util import REPORT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GEMM_VARIANTS = (0, 2, 3, 4, 5, 6, 7, 8)


def _rows(M, d, dtype, seed, offset=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, d, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 3.0) + offset * torch.randn(M, 1, generator=g)
    return x.to(dtype).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_row_stats_are_the_part_moments(built_lib, dtype):
    from vall_e.vall_e import _hip
    x = _rows(777, 512, dtype, 1, offset=2.0)
    raw = _hip.op_row_stats(x)
    st = _hip.stats_rows(raw, 777).cpu()
    xf = x.float().cpu().reshape(777, 16, 32)
    assert torch.allclose(st[..., 0], xf.sum(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(st[..., 1], (xf * xf).sum(-1), rtol=1e-5, atol=1e-4)
    # a view with a row stride (the rows of a wider buffer)
    wide = torch.zeros(777, 1024, dtype=dtype, device=DEV)
    wide[:, :512] = x
    assert torch.equal(_hip.stats_rows(_hip.op_row_stats(wide[:, :512]), 777).cpu(), st)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,form", [(1536, 512, "r1"), (768, 512, "r2"), (768, 2048, "mask"), (24576, 512, "r1"), (24576, 2048, "mask"),
                                      (12288, 512, "r2"), (1000, 512, "r1")])
def test_producer_moments_are_schedule_independent(built_lib, dtype, M, K, form):
    """out-projection / fc2 + residual with EPI_STATS: the stored rows equal the plain launch's, the moments equal
    d3pm_op_row_stats of the stored rows, both bit for bit, under every GEMM schedule that applies to the shape."""
    from vall_e.vall_e import _hip
    g = torch.Generator().manual_seed(M + K)
    x = _rows(M, K, dtype, 2)
    w = (torch.randn(512, K, generator=g) / K ** 0.5).to(dtype).to(DEV)
    b = (torch.randn(512, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = _rows(M, 512, dtype, 3, offset=1.0)
    r2 = _rows(M, 512, dtype, 4) if form == "r2" else None
    mask = (torch.rand(96, generator=g) > 0.2).to(torch.uint8).to(DEV) if form == "mask" else None
    period = 96 if form == "mask" else 1
    ref_y = ref_st = None
    for v in GEMM_VARIANTS:
        with _hip.tuning(gemm_variant=v):
            y, st = _hip.op_linear_stats(x, w, b, r1, r2=r2, row_mask=mask, mask_period=period)
            plain = _hip.op_linear(x, w, b, r1=r1, r2=r2, row_mask=mask, mask_period=period)
        assert torch.equal(y, plain), f"variant {v}: the stored rows differ from the plain launch"
        assert torch.equal(_hip.stats_rows(st, M), _hip.stats_rows(_hip.op_row_stats(y), M)), f"variant {v}: moments are not those of the stored rows"
        if ref_y is None:
            ref_y, ref_st = y, st
        assert torch.equal(y, ref_y) and torch.equal(_hip.stats_rows(st, M), _hip.stats_rows(ref_st, M)), f"variant {v} differs from the automatic schedule"


def _ln_linear_ref(x, gamma, beta, w, b, film=None, gelu=False):
    """fp32 evaluation of LayerNorm -> [FiLM] -> Linear -> [GELU] on the 16-bit operands."""
    xf = x.float()
    h = F.layer_norm(xf, (xf.shape[1],), gamma.float(), beta.float(), 1e-6)
    if film is not None:
        d = xf.shape[1]
        h = h * (1.0 + film[:d].float()) + film[d:].float()
    y = F.linear(h, w.float(), b.float())
    return F.gelu(y) if gelu else y


@pytest.mark.parametrize("dtype,quantum", [(torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)])
@pytest.mark.parametrize("M,N,film,gelu,offset", [(768, 1536, False, False, 0.0), (1536, 1024, False, False, 3.0), (768, 2048, True, True, 0.0),
                                                   (24576, 1536, False, False, 1.0), (24576, 2048, True, True, 0.5), (12288, 1024, False, False, 10.0),
                                                   (1000, 1536, False, False, 0.0)])
def test_folded_projection_against_fp32_layernorm_linear(built_lib, dtype, quantum, M, N, film, gelu, offset):
    from vall_e.vall_e import _hip
    d = 512
    g = torch.Generator().manual_seed(N + M)
    x = _rows(M, d, dtype, 5, offset=offset)
    gamma = (1.0 + 0.3 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    beta = (0.2 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    w = (torch.randn(N, d, generator=g) / d ** 0.5).to(dtype).to(DEV)
    b = (0.1 * torch.randn(N, generator=g)).to(dtype).to(DEV)
    fv = (0.3 * torch.randn(2 * d, generator=g)).to(dtype).to(DEV) if film else None
    ref = _ln_linear_ref(x, gamma, beta, w, b, fv, gelu)
    wf, fs, fb = _hip.op_fold_weights(w, b, gamma, beta, fv)
    st = _hip.op_row_stats(x)
    outs = []
    for v in GEMM_VARIANTS:
        with _hip.tuning(gemm_variant=v):
            outs.append(_hip.op_linear_fold(x, wf, fs, fb, st, act=1 if gelu else 0))
        assert torch.equal(outs[-1], outs[0]), f"GEMM schedule {v} changes the folded projection"
    unfolded = _hip.op_linear(_hip.op_layernorm(x, gamma, beta, film=fv), w, b, act=1 if gelu else 0)
    scale = ref.abs().max().item()
    err_f = (outs[0].float() - ref).abs().max().item() / scale
    err_u = (unfolded.float() - ref).abs().max().item() / scale
    rms_f = (outs[0].float() - ref).pow(2).mean().sqrt().item() / scale
    rms_u = (unfolded.float() - ref).pow(2).mean().sqrt().item() / scale
    REPORT[f"fold_{str(dtype)[6:]}_{M}x{N}_film{int(film)}_offset{offset}"] = {"max_rel_err_folded": err_f, "max_rel_err_unfolded": err_u,
                                                                           "rms_rel_err_folded": rms_f, "rms_rel_err_unfolded": rms_u}
    # output rounding alone costs half a quantum of the largest value; W' = rn(W gamma) and the 16-bit rows add a few more
    assert err_f < 4 * quantum, (err_f, err_u)
    assert rms_f < 1.25 * rms_u + 1e-6, f"the folded form is noisier than the launches it replaces: rms {rms_f} vs {rms_u}"


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 8e-3), (torch.bfloat16, 7e-2)])
def test_denoiser_folded_vs_unfolded_and_shard_invariance(built_lib, dtype, tol):
    """One denoiser evaluation at the libritts shape with the LayerNorms folded (default) and unfolded: the logits agree to the
    storage-type noise both carry against the oracle (tests/test_gpu_bench_path.py holds each of them to it); and four utterances
    taken out of a 32-utterance batch reproduce their logits and ids bit for bit once the shard is told the global batch."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    assert smp.folded, "the sampler did not build the folded tables for a 16-bit d = 512 model"
    texts, proms = synth.make_inputs(cfg, 32, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(32)
    gen = torch.Generator().manual_seed(3)
    live = torch.rand(32, cfg.canvas, generator=gen) < 0.5
    ids = torch.randint(0, 1024, (32, cfg.canvas), generator=gen, dtype=torch.int32)
    x = torch.where(live.to(DEV) & (x != 0), ids.to(DEV), x)
    t = 40
    lg_f = smp.denoise(x, fm, t, kv_t, kv_p)[0].clone()
    with _hip.tuning(ln_fold=0):
        lg_u = smp.denoise(x, fm, t, kv_t, kv_p)[0].clone()
    diff = (lg_f.float() - lg_u.float()).abs()
    REPORT[f"fold_denoise_{str(dtype)[6:]}_folded_vs_unfolded"] = {"max_abs": diff.max().item(), "mean_abs": diff.mean().item()}
    assert diff.max().item() < tol and diff.mean().item() < tol / 5
    assert not torch.equal(lg_f, lg_u)
    # utterances 8..11 alone, told that they are a shard of 32: same kernels, same bits
    sl = slice(8, 12)
    kv_ts, kv_ps = smp.cond_kv(ct[sl].contiguous(), cp[sl].contiguous())
    with _hip.tuning(regime_batch=32):
        lg_s = smp.denoise(x[sl].contiguous(), fm, t, kv_ts, kv_ps)[0].clone()
    assert torch.equal(lg_s, lg_f[sl]), "a shard of the batch does not reproduce the unsplit batch's logits"
    # ... and without the hint the shard runs the small-batch attention kernels: rounding-level differences, not equality
    lg_n = smp.denoise(x[sl].contiguous(), fm, t, kv_ts, kv_ps)[0]
    assert (lg_n.float() - lg_f[sl].float()).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_loop_equals_step_by_step_ops_and_the_general_sampler(built_lib, dtype):
    """Inside d3pm_sample_loop the sampler launch of iteration t also prepares iteration t - 1 (embedding rows + their moments, fc1
    under norm3 + FiLM(t - 1): posterior_sample_prep_rows), and for 1025 classes the row routine is the predicate-free one
    (sample_row_1025).  Both must be invisible: the loop's trace equals a chain of d3pm_denoise_step + d3pm_posterior_sample calls
    (stand-alone embedding / fold launches), and the ids of the specialised routine equal those of the general routine (taken when
    the posterior is asked for) on the same logits."""
    from vall_e.vall_e import AR, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    texts, proms = synth.make_inputs(cfg, 3, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(3)
    xs = x.clone()
    trace = smp.sample_loop(x, fm, 99, 92, kv_t, kv_p, seed=77, trace=True)
    for i, t in enumerate(range(99, 92, -1)):
        lg, _ = smp.denoise(xs, fm, t, kv_t, kv_p)
        nxt, _ = smp.posterior_sample(lg, xs, t, seed=77)
        nxt_general, post = smp.posterior_sample(lg, xs, t, seed=77, want_posterior=True)
        assert torch.equal(nxt, nxt_general), f"t={t}: the 1025-class routine and the general routine draw different ids"
        assert torch.equal(trace[i], nxt), f"t={t}: the loop (fused preparation) and the step-by-step ops disagree on {(trace[i] != nxt).sum().item()} ids"
        xs = nxt
    assert torch.equal(x, xs)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_sampler_early_out_on_revealed_rows_returns_the_full_routines_ids(built_lib, dtype):
    """sample_row_1025 returns the kept token of a revealed row (x_t != mask) without the per-class posterior when that token's exact
    score clears an upper bound of every other class's (d3pm_sample_row.h).  The ids must be those of the full routine -- taken by the
    general sample_row when the posterior is asked for -- on every kind of row: mostly masked (t = 99), half revealed, almost all
    revealed, and on logits made hostile to the test (peaked on ANOTHER class than the kept one, so that the kept token loses the race
    and the fallback has to run; flat; peaked on the kept token)."""
    from vall_e.vall_e import AR, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    B = 4
    texts, proms = synth.make_inputs(cfg, B, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(B)
    g = torch.Generator(device="cpu").manual_seed(5)
    rows, K = x.numel(), 1025
    n_revealed_total, n_changed_total, t_prev = 0, 0, 99
    for t in (99, 60, 30, 10, 2):
        if t != 99:                      # advance the canvas of the previous case to x_t along the build's own trajectory
            smp.sample_loop(x, fm, t_prev, t, kv_t, kv_p, seed=11 + t)
        t_prev = t
        xs = x.clone()
        lg, _ = smp.denoise(xs, fm, t, kv_t, kv_p)
        variants = [lg]
        flat = lg.reshape(rows, K).float()
        kept = xs.reshape(rows).long().clamp(max=K - 1)
        other = (kept + 17) % 1024
        peaked_other = flat.clone(); peaked_other[torch.arange(rows), other] += 12.0
        peaked_kept = flat.clone(); peaked_kept[torch.arange(rows), kept] += 12.0
        variants += [peaked_other.to(dtype).reshape(lg.shape), peaked_kept.to(dtype).reshape(lg.shape), torch.zeros_like(lg),
                     (flat * 6.0).to(dtype).reshape(lg.shape)]
        for vi, logits in enumerate(variants):
            fast, _ = smp.posterior_sample(logits, xs, t, seed=1000 + t)
            full, _ = smp.posterior_sample(logits, xs, t, seed=1000 + t, want_posterior=True)
            assert torch.equal(fast, full), f"t={t} variant {vi}: {(fast != full).sum().item()} ids differ between the early-out and the full routine"
            revealed = xs.reshape(-1) != cfg.mask_id
            n_revealed_total += int(revealed.sum())
            n_changed_total += int((full.reshape(-1)[revealed] != xs.reshape(-1)[revealed]).sum())
    assert n_revealed_total > 10000, "the cases must contain revealed rows"
    assert n_changed_total > 100, "and rows whose kept token loses the race (the fallback path)"


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2.5e-2), (torch.bfloat16, 1.5e-1)])
def test_condition_encoders_at_d512_on_padded_heads_track_the_torch_modules(built_lib, dtype, tol):
    """At d_model = 512 the encoders' 16 heads are 32 wide; since round 4 their self-attention runs on the 64-wide MFMA kernels
    through zero-padded heads (csrc/d3pm_headpad.hip) instead of the generic FMA kernel.  The conditions must track the torch modules
    of the same weights and dtype (the eager path the generic kernel mirrors) as closely as 16-bit rounding allows, for both
    encoders, at batch 1 and at batch 3 with the utterances of a batch equal to their batch-1 runs."""
    from vall_e.vall_e import AR, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    texts, proms = synth.make_inputs(cfg, 3, 1)
    ct, cp = m.encode_conditions(texts, proms)
    rt, rp = m.encode_conditions_torch(texts, proms)
    for name, a, b in (("text", ct, rt), ("prompt", cp, rp)):
        a, b = a.float().cpu(), b.float().cpu()
        err = (a - b).abs().max().item()
        scale = b.abs().max().item()
        assert torch.isfinite(a).all() and err < tol * max(scale, 1.0), (name, err, scale)
    for b_ in range(3):
        ct1, cp1 = m.encode_conditions(texts[b_:b_ + 1], proms[b_:b_ + 1])
        assert torch.equal(ct1[0], ct[b_]) and torch.equal(cp1[0], cp[b_]), f"utterance {b_}: the conditions depend on the batch"
