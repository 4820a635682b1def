"""Parity of the HIP path (through the C ABI) against the reference's golden vectors and the CPU
oracle.  Run on a real MI355X: python -m pytest tests -m gpu.

Tolerances (BASELINE.json north_star): token ids bit-exact wherever the arithmetic is integer or
the logits are given; denoiser logits within 1e-3 of the fp32 reference in F32 mode.  In F16 mode
(the only dtype the reference sampler runs in) every op output is rounded to fp16 like the eager
reference, but GEMM accumulation order differs from the CPU BLAS, so activations may differ by a few
fp16 ulps; a sampled id may then differ only where the reference's own Gumbel race was a near-tie,
which the teacher-forced test audits position by position.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import d3pm_oracle as O
from oracle import philox
from util import REPORT, bits, f16, load, native_setup, ulp16_diff

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def make_model(cfg, sd32, dtype):
    from vall_e.vall_e import AR
    m = AR.from_config(cfg)
    m.load_state_dict(sd32)
    return m.to(dtype).to(DEV)


class Native:
    def __init__(self, dtype):
        self.cfg, self.sd32, self.texts, self.proms, self.orc = native_setup(dtype)
        self.model = make_model(self.cfg, self.sd32, dtype)
        self.smp = self.model.sampler()
        with torch.no_grad():
            self.cp, self.ct = self.orc.conditions(self.texts[0], self.proms[0])      # oracle conditions
        self.kv_t, self.kv_p = self.smp.cond_kv(self.ct[None].to(DEV), self.cp[None].to(DEV))
        self.mask = torch.zeros(self.cfg.canvas, dtype=torch.bool)
        self.mask[: self.cfg.n_frames] = True
        self.fm = self.mask.to(torch.uint8).to(DEV)

    def logits(self, x_t, t, **kw):
        x = torch.as_tensor(np.asarray(x_t), dtype=torch.int32, device=DEV).reshape(1, -1)
        lg, hid = self.smp.denoise(x, self.fm, t, self.kv_t, self.kv_p, **kw)
        return (None if lg is None else lg[0].cpu()), (None if hid is None else hid[0].cpu())


@pytest.fixture(scope="module")
def n16():
    return Native(torch.float16)


@pytest.fixture(scope="module")
def n32():
    return Native(torch.float32)


# ---------------------------------------------------------------------------------------------
def test_uniform_stream_is_the_oracles(built_lib):
    from vall_e.vall_e import _hip
    for stream in (0, 1):
        u = _hip.uniform(123, 40, 448 * 3, 448, 1025, stream, DEV).cpu().numpy()
        assert np.array_equal(u, philox.uniform_rows(123, 40, 448 * 3, 448, 1025, stream))


def test_posterior_and_sample_on_reference_logits(n16):
    """Given the reference's own fp16 logits, x_t and the shared noise, the ids must be the reference's."""
    g = load("native_step.npz")
    logits = f16(g["logits_full_f16"])[None].to(DEV)
    x_t = torch.from_numpy(g["x_t"].astype(np.int32))[None].to(DEV)
    t = int(g["t"])
    x_next, post = n16.smp.posterior_sample(logits, x_t, t, seed=123, want_posterior=True)
    post = post[0].cpu().view(torch.float16)
    ref_rows = f16(g["posterior_rows_f16"])
    du = ulp16_diff(post[g["rows"]], ref_rows)
    REPORT["posterior_rows_ulp_max"] = int(du.max())
    REPORT["posterior_rows_mismatch_frac"] = float((du > 0).float().mean())
    assert du.max() <= 1 and (du > 0).float().mean() < 2e-3
    ids = x_next[0].cpu().numpy()
    mism = int((ids != g["x_next_seed123"]).sum())
    REPORT["sample_given_logits_mismatches"] = mism
    assert mism == 0


def test_wrong_shapes_are_rejected_before_the_c_abi(n16):
    """Upstream's p_sample / q_sample accept any [B, W]; here W must be the canvas the kernels index with."""
    from vall_e.vall_e import _hip
    cfg = n16.cfg
    logits = torch.zeros(1, cfg.canvas, cfg.n_classes, dtype=torch.float16, device=DEV)
    t = torch.tensor([40])
    with pytest.raises(_hip.D3PMError):
        n16.model.p_sample(logits, t, torch.zeros(1, cfg.canvas - 48, dtype=torch.int64, device=DEV))
    with pytest.raises(_hip.D3PMError):
        n16.model.p_sample(logits[:, :, :1000], t, torch.zeros(1, cfg.canvas, dtype=torch.int64, device=DEV))
    with pytest.raises(_hip.D3PMError):
        n16.model.q_sample(torch.zeros(1, cfg.canvas, dtype=torch.int64, device=DEV), t, torch.ones(cfg.canvas - 1, dtype=torch.bool, device=DEV))
    with pytest.raises(_hip.D3PMError):
        n16.smp.denoise(torch.zeros(2, cfg.canvas, dtype=torch.int32, device=DEV), n16.fm, 40, n16.kv_t, n16.kv_p)   # K/V of one utterance
    with pytest.raises(_hip.D3PMError):      # fp8 entry points refuse shapes their kernels cannot run instead of running 16-bit ones
        n16.smp.denoise(torch.zeros(1, cfg.canvas, dtype=torch.int32, device=DEV), n16.fm, 40, n16.kv_t, n16.kv_p, fp8=True)


def test_q_sample_matches_reference(n16):
    g = load("native_step.npz")
    x0 = torch.from_numpy(g["x_t"].astype(np.int32))[None].to(DEV)
    out = n16.smp.q_sample(x0, n16.fm, int(g["t"]), seed=123)
    assert np.array_equal(out[0].cpu().numpy(), g["q_sample_seed123"].astype(np.int32))
    m = n16.model.q_sample(x0.long(), torch.tensor([int(g["t"])]), n16.mask.to(DEV), seed=123)
    assert np.array_equal(m[0].cpu().numpy(), g["q_sample_seed123"].astype(np.int64))


def test_logits_fp32_within_1e3_of_reference(n32):
    g = load("native_step.npz")
    t = int(g["t"])
    lg, hid = n32.logits(g["x_t"], t, want_hidden=True)
    ref_rows = torch.from_numpy(g["logits_rows_f32"])
    err = (lg[g["rows"]] - ref_rows).abs().max().item()
    REPORT["native_f32_logits_max_abs_err_vs_reference"] = err
    assert err < 1e-3
    with torch.no_grad():
        full = n32.orc.logits(torch.from_numpy(g["x_t"].astype(np.int64)), t, n32.cp, n32.ct, n32.mask)
    err_full = (lg - full).abs().max().item()
    REPORT["native_f32_logits_max_abs_err_vs_oracle_full"] = err_full
    assert err_full < 1e-3
    assert (hid - torch.from_numpy(g["hidden_f32"])).abs().max().item() < 1e-3


def test_block0_matches_reference(n32, n16):
    g = load("native_step.npz")
    t = int(g["t"])
    _, h32 = n32.logits(g["x_t"], t, want_logits=False, want_hidden=True, only_layers=1)
    e32 = (h32 - torch.from_numpy(g["block0_out_f32"])).abs().max().item()
    _, h16 = n16.logits(g["x_t"], t, want_logits=False, want_hidden=True, only_layers=1)
    ref16 = f16(g["block0_out_f16"])
    d16 = (h16.float() - ref16.float()).abs()
    REPORT["block0_f32_max_abs_err"] = e32
    REPORT["block0_f16_max_abs_err"] = float(d16.max())
    REPORT["block0_f16_exact_frac"] = float((h16 == ref16).float().mean())
    REPORT["block0_f16_absmax"] = float(ref16.abs().max())
    assert e32 < 1e-4
    # fp16 eager reference vs fp16 HIP: same rounding points, different accumulation order ->
    # at most a couple of fp16 quanta of the activation range (|x| < 8 -> quantum 2^-8 .. 2^-7)
    assert d16.max() <= 2.0 ** -6 and d16.mean() < 5e-4


def test_logits_fp16_close_to_reference(n16):
    g = load("native_step.npz")
    lg, _ = n16.logits(g["x_t"], int(g["t"]))
    ref = f16(g["logits_full_f16"])
    diff = (lg.float() - ref.float()).abs()
    du = ulp16_diff(lg, ref)
    REPORT["native_f16_logits_max_abs_err"] = float(diff.max())
    REPORT["native_f16_logits_exact_frac"] = float((du == 0).float().mean())
    REPORT["native_f16_logits_mean_abs_err"] = float(diff.mean())
    REPORT["native_f16_logits_frac_within_2e-3"] = float((diff <= 2e-3).float().mean())
    # |logits| < 4 -> fp16 quantum up to 2^-9 = 1.95e-3; the reference's own fp16-vs-fp32 gap is 4.8e-3
    assert diff.max() < 8e-3 and diff.mean() < 6e-4
    assert (diff <= 2e-3).float().mean() > 0.99


def _audit(n, golden_traj, seed, utt=0):
    """Teacher-forced: feed the reference's x_t at every step, compare the sampled x_{t-1}.
    Returns (mismatches, audited_ok, worst_gap)."""
    cfg = n.cfg
    x_init, _ = n.orc.canvas_init()
    prev = x_init.numpy()
    mism, audited, worst = 0, 0, 0.0
    for i, t in enumerate(range(cfg.timesteps - 1, 0, -1)):
        ref_next = golden_traj[i].astype(np.int64)
        x = torch.from_numpy(prev.astype(np.int32))[None].to(DEV)
        lg, _ = n.smp.denoise(x, n.fm, t, n.kv_t, n.kv_p)
        nxt, _ = n.smp.posterior_sample(lg, x, t, seed=seed, utt0=utt)
        got = nxt[0].cpu().numpy()
        bad = np.nonzero(got != ref_next)[0]
        if len(bad):
            with torch.no_grad():
                post = n.orc.posterior(n.orc.logits(torch.from_numpy(prev.astype(np.int64)), t, n.cp, n.ct, n.mask),
                                       torch.from_numpy(prev.astype(np.int64)), t)
            u = torch.from_numpy(philox.uniform_batch(seed, t, utt, 1, cfg.canvas)[0])
            gum = -torch.log(-torch.log(torch.clamp(u, min=torch.finfo(torch.float32).tiny, max=1.0)))
            v = post.float() + gum
            for r in bad:
                gap = (v[r, ref_next[r]] - v[r, got[r]]).item()
                worst = max(worst, gap)
                audited += int(gap < 0.05)       # < ~3 fp16 quanta of a posterior logit near -20
            mism += len(bad)
        prev = ref_next
    return mism, audited, worst


def test_teacher_forced_loop_ids(n16):
    """Every one of the 99 x 448 sampled ids equals the reference's, or the reference's own race
    between the two candidates was decided by less than the fp16 quantum of the posterior logits."""
    g = load("native_loop.npz")
    mism, audited, worst = _audit(n16, g["traj_utt0_seed123"], 123)
    total = 99 * n16.cfg.canvas
    REPORT["teacher_forced_mismatches"] = mism
    REPORT["teacher_forced_total"] = total
    REPORT["teacher_forced_worst_gap"] = worst
    assert mism == audited, f"{mism - audited} mismatches are not near-ties (worst gap {worst})"
    assert mism / total < 2e-3


def test_free_running_loop_agreement(n16):
    g = load("native_loop.npz")
    x, fm = n16.model.canvas_init(1)
    n16.smp.sample_loop(x, fm, 99, 0, n16.kv_t, n16.kv_p, seed=123)
    got = x[0].cpu().numpy()
    ref = g["traj_utt0_seed123"][-1]
    agree = float((got == ref).mean())
    REPORT["free_running_agreement"] = agree
    REPORT["free_running_live_agreement"] = float((got[:350] == ref[:350]).mean())
    assert agree > 0.9


def test_greedy_is_bit_identical(n16):
    g = load("native_loop.npz")
    from vall_e.vall_e import _hip
    x, fm = n16.model.canvas_init(1)
    n16.smp.sample_loop(x, fm, 99, 0, n16.kv_t, n16.kv_p, seed=0, flags=_hip.FLAG_GREEDY)
    assert np.array_equal(x[0].cpu().numpy(), g["greedy"].astype(np.int32))
    # final.weight x 30: every live frame unmasks at t=1 to the same id
    from vall_e.vall_e import synth
    sdg = synth.make_state_dict(n16.cfg, 0, logit_gain=30.0)
    m = make_model(n16.cfg, sdg, torch.float16)
    smp = m.sampler()
    kv_t, kv_p = smp.cond_kv(n16.ct[None].to(DEV), n16.cp[None].to(DEV))
    x, fm = m.canvas_init(1)
    smp.sample_loop(x, fm, 99, 0, kv_t, kv_p, seed=0, flags=_hip.FLAG_GREEDY)
    assert np.array_equal(x[0].cpu().numpy(), g["greedy_gain30"].astype(np.int32))


def test_plumbing_config_10_steps(n16):
    """BASELINE.json configs[0]: 1 utterance, 10 diffusion steps (timesteps attribute lowered at call time)."""
    g = load("native_loop.npz")
    n16.model.timesteps = 11
    try:
        out = n16.model.generate_audio([n16.texts[0]], [n16.proms[0]], seed=123)
    finally:
        n16.model.timesteps = 100
    assert out.shape == (448,) and out.dtype == torch.int64
    agree = float((out.cpu().numpy() == g["plumbing10_utt0_seed123"]).mean())
    REPORT["plumbing10_agreement"] = agree
    assert agree > 0.99


def test_condition_encoders_match_reference(n16, n32):
    """HIP condition encoders (d3pm_encode_conditions: embeddings + PE, 2 post-norm encoder layers, SiLU Mlp)
    against the reference's conditions (golden, via the oracle) -- and the torch-ROCm module path for comparison."""
    g = load("native_step.npz")
    for n, tag, tol in ((n32, "f32", 2e-4), (n16, "f16", 1.2e-2)):
        conv = (lambda a: torch.from_numpy(a)) if tag == "f32" else f16
        ref_t, ref_p = conv(g[f"cond_text_{tag}"]).float(), conv(g[f"cond_prompt_{tag}"]).float()
        for name, fn in (("hip", n.model.encode_conditions), ("torch", n.model.encode_conditions_torch)):
            ct, cp = fn(n.texts[:2], n.proms[:2])
            et = (ct[0].cpu().float() - ref_t).abs().max().item()
            ep = (cp[0].cpu().float() - ref_p).abs().max().item()
            REPORT[f"cond_{name}_{tag}_text_max_abs_err"] = et
            REPORT[f"cond_{name}_{tag}_prompt_max_abs_err"] = ep
            assert et < tol and ep < tol, (name, tag, et, ep)
    # a prompt with fewer quantizer levels: the absent levels contribute nothing (zero one-hot rows upstream)
    short = [p[:, :3] for p in n32.proms[:1]]
    with torch.no_grad():
        cp_ref, _ = n32.orc.conditions(n32.texts[0], short[0])
    _, cp = n32.model.encode_conditions(n32.texts[:1], short)
    assert (cp[0].cpu() - cp_ref).abs().max().item() < 2e-4


def test_batch_is_independent_runs(n16):
    texts, proms = n16.texts[:2] + [n16.texts[0]], n16.proms[:2] + [n16.proms[0]]
    both = n16.model.generate_audio(texts, proms, steps=12, seed=5, utt0=4).cpu()
    for b in range(3):
        one = n16.model.generate_audio([texts[b]], [proms[b]], steps=12, seed=5, utt0=4 + b).cpu()
        assert torch.equal(both[b], one), b
    again = n16.model.generate_audio(texts, proms, steps=12, seed=5, utt0=4).cpu()
    assert torch.equal(both, again)
    chunked = n16.model.generate_audio(texts, proms, steps=12, seed=5, utt0=4, streams=2).cpu()   # 2 HIP streams
    assert torch.equal(both, chunked)
    assert not torch.equal(both[0], both[2])      # same inputs, different noise rows


def test_masked_out_frames_never_influence_live_frames(n16):
    g = load("native_step.npz")
    x = g["x_t"].copy()
    a, _ = n16.logits(x, 40)
    x[n16.cfg.n_frames:] = np.random.default_rng(0).integers(0, 1025, size=n16.cfg.canvas - n16.cfg.n_frames)
    b, _ = n16.logits(x, 40)
    assert torch.equal(a, b)


def test_bf16_mode_runs_and_tracks_fp32(n32):
    cfg, sd32 = n32.cfg, n32.sd32
    m = make_model(cfg, sd32, torch.bfloat16)
    smp = m.sampler()
    kv_t, kv_p = smp.cond_kv(n32.ct[None].to(DEV), n32.cp[None].to(DEV))
    g = load("native_step.npz")
    x = torch.from_numpy(g["x_t"].astype(np.int32))[None].to(DEV)
    lg, _ = smp.denoise(x, n32.fm, 40, kv_t, kv_p)
    ref, _ = n32.logits(g["x_t"], 40)
    err = (lg[0].cpu().float() - ref).abs().max().item()
    REPORT["native_bf16_vs_f32_logits_max_abs_err"] = err
    assert err < 0.15


# ---- the widths get_model asks for (d=512, H=8, L=6), canvas 448: reference's own classes ----------
@pytest.fixture(scope="module")
def wide():
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig(d_model=512, n_heads=8, n_layers=6)
    sd32 = synth.make_state_dict(cfg, 0)
    texts, proms = synth.make_inputs(cfg, 1, 1)
    return cfg, sd32, texts, proms


@pytest.mark.parametrize("tag,dtype,force_generic", [("f32", torch.float32, False), ("f16", torch.float16, True),
                                                      ("f16", torch.float16, False)])
def test_wide_model_logits(wide, tag, dtype, force_generic):
    from vall_e.vall_e import _hip
    cfg, sd32, texts, proms = wide
    g = load("wide_step.npz")
    orc = O.Oracle({k: v.to(dtype) for k, v in sd32.items()}, O.Shape.of(cfg))
    with torch.no_grad():
        cp, ct = orc.conditions(texts[0], proms[0])
    conv = (lambda a: torch.from_numpy(a)) if dtype == torch.float32 else f16
    assert (cp[:16].float() - conv(g[f"cond_prompt_rows_{tag}"]).float()).abs().max() < 2e-2
    m = make_model(cfg, sd32, dtype)
    smp = m.sampler()
    ct_hip, cp_hip = m.encode_conditions(texts, proms)              # HIP condition encoders at d=512 (16 heads of 32)
    ctol = 1e-3 if dtype == torch.float32 else 3e-2
    ec = (cp_hip[0, :16].cpu().float() - conv(g[f"cond_prompt_rows_{tag}"]).float()).abs().max().item()
    et = (ct_hip[0, :16].cpu().float() - conv(g[f"cond_text_rows_{tag}"]).float()).abs().max().item()
    REPORT[f"wide_{tag}_cond_hip_max_abs_err"] = max(ec, et)
    assert ec < ctol and et < ctol
    kv_t, kv_p = smp.cond_kv(ct[None].to(DEV), cp[None].to(DEV))
    x = torch.from_numpy(g["x_t"].astype(np.int32))[None].to(DEV)
    fm = torch.zeros(cfg.canvas, dtype=torch.uint8, device=DEV)
    fm[: cfg.n_frames] = 1
    flags = _hip.FLAG_FORCE_GENERIC if force_generic else 0
    lg, hid = smp.denoise(x, fm, int(g["t"]), kv_t, kv_p, want_hidden=True, flags=flags)
    ref = conv(g[f"logits_rows_{tag}"]).float()
    err = (lg[0].cpu()[g["rows"]].float() - ref).abs().max().item()
    herr = (hid[0].cpu()[g["rows"]].float() - conv(g[f"hidden_rows_{tag}"]).float()).abs().max().item()
    key = f"wide_{tag}_{'generic' if force_generic else 'auto'}"
    REPORT[key + "_logits_max_abs_err"] = err
    REPORT[key + "_hidden_max_abs_err"] = herr
    REPORT[key + "_logits_absmax"] = float(ref.abs().max())
    assert err < (1e-3 if dtype == torch.float32 else 2e-2)
    if dtype == torch.float16:
        nxt, _ = smp.posterior_sample(lg, x, int(g["t"]), seed=123)
        agree = float((nxt[0].cpu().numpy() == g["x_next_seed123"]).mean())
        REPORT[key + "_sample_agreement"] = agree
        assert agree >= 0.99, f"{key}: only {agree:.4f} of the ids sampled from the HIP logits equal the reference's"


# ---- BASELINE.json configs[1] at full size: size-independent properties -----------------------------------
def test_full_size_libritts_properties():
    """32 utterances, d=512/H=8/L=6, 768-frame canvas, bf16 (the bench workload), 3 reverse steps:
    determinism, batch-split invariance (B=32 run == B=16 run on the same global utterance indices),
    stream-chunk invariance, id range, masked-frame independence -- no oracle needed at this size."""
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig.libritts()
    sd32 = synth.make_state_dict(cfg, 0)
    m = make_model(cfg, sd32, torch.bfloat16)
    texts, proms = synth.make_inputs(cfg, 32, 1)
    a = m.generate_audio(texts, proms, steps=3, seed=9).cpu()
    assert a.shape == (32, cfg.canvas) and a.dtype == torch.int64
    assert int(a.min()) >= 0 and int(a.max()) <= 1024
    assert torch.equal(a, m.generate_audio(texts, proms, steps=3, seed=9).cpu())
    assert not torch.equal(a, m.generate_audio(texts, proms, steps=3, seed=10).cpu())
    # batch-split invariance of the HIP loop: given the same conditions, utterances 3..18 run alone (with their
    # global noise rows) reproduce rows 3..18 of the 32-utterance run bit for bit -- also across stream chunks.
    # (End to end the torch-ROCm condition encoders may pick batch-size dependent BLAS kernels, so the
    # generate_audio-level comparison is on ids with a tolerance.)
    smp = m.sampler()
    ct_all, cp_all = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct_all, cp_all)
    x, fm = m.canvas_init(32)
    smp.sample_loop(x, fm, 3, 0, kv_t, kv_p, seed=9)
    # (16 utterances: the same self-attention kernel as the 32 -- tests/test_gpu_batch_sweep.py)
    kv_t5, kv_p5 = smp.cond_kv(ct_all[3:19].contiguous(), cp_all[3:19].contiguous())
    x5, _ = m.canvas_init(16)
    smp.sample_loop(x5, fm, 3, 0, kv_t5, kv_p5, seed=9, utt0=3)
    assert torch.equal(x[3:19], x5)
    five = m.generate_audio(texts[3:8], proms[3:8], steps=3, seed=9, utt0=3).cpu()
    assert (five == a[3:8]).float().mean().item() > 0.99
    chunked = m.generate_audio(texts, proms, steps=3, seed=9, streams=4).cpu()
    assert (chunked == a).float().mean().item() > 0.99
    # logits never depend on the ids parked in masked-out frames
    ct, cp = m.encode_conditions(texts[:2], proms[:2])
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(2)
    la, _ = smp.denoise(x, fm, 50, kv_t, kv_p)
    x[:, cfg.n_frames:] = 77
    lb, _ = smp.denoise(x, fm, 50, kv_t, kv_p)
    assert torch.equal(la, lb)
    # MFMA family vs generic family on the same inputs (different accumulation order / flash softmax only)
    from vall_e.vall_e import _hip
    lg, _ = smp.denoise(x, fm, 50, kv_t, kv_p, flags=_hip.FLAG_FORCE_GENERIC)
    err = (la.float() - lg.float()).abs().max().item()
    REPORT["libritts_bf16_mfma_vs_generic_logits_max_abs_err"] = err
    assert err < 0.15


def test_vctk_long_prompt_config_runs():
    """BASELINE.json configs[3]: 750 prompt keys, 375 live frames on a 384 canvas, 200-step
    schedule (the closed-form scalars for 200 steps are pinned to the reference in tests/golden/tables_t200.npz).
    Determinism, id range, MFMA-vs-generic logits, and the first iteration starts at t = 199."""
    from vall_e.vall_e import _hip, synth
    cfg = synth.D3PMConfig.vctk_long_prompt()
    m = make_model(cfg, synth.make_state_dict(cfg, 0), torch.bfloat16)
    texts, proms = synth.make_inputs(cfg, 2, 1)
    a, trace = m.generate_audio(texts, proms, steps=None, seed=3, return_trace=True)
    assert a.shape == (2, cfg.canvas) and trace.shape == (199, 2, cfg.canvas)
    assert int(a.min()) >= 0 and int(a.max()) <= 1024
    assert (a[:, cfg.n_frames:] == trace[-1].long()[:, cfg.n_frames:]).all()
    assert torch.equal(a, m.generate_audio(texts, proms, seed=3))
    masked = (trace[:, :, :cfg.n_frames] == cfg.mask_id).sum(dim=(1, 2)).cpu()
    assert masked[0] > masked[-1]                      # frames unmask over the 199 iterations
    smp = m.sampler()
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(2)
    la, _ = smp.denoise(x, fm, 150, kv_t, kv_p)
    lg, _ = smp.denoise(x, fm, 150, kv_t, kv_p, flags=_hip.FLAG_FORCE_GENERIC)
    assert (la.float() - lg.float()).abs().max().item() < 0.15


def test_sample_loop_is_hip_graph_capturable(n16):
    """include/d3pm_hip.h promises that nothing in the library allocates or synchronises: the whole reverse process
    can be captured into a HIP graph on a side stream and replayed."""
    x_eager, fm = n16.model.canvas_init(1)
    n16.smp.sample_loop(x_eager, fm, 8, 0, n16.kv_t, n16.kv_p, seed=21)
    x_graph, _ = n16.model.canvas_init(1)
    x0 = x_graph.clone()
    n16.smp.workspace(1)                        # allocate outside the capture
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        n16.smp.sample_loop(x_graph, fm, 8, 0, n16.kv_t, n16.kv_p, seed=21)
    for _ in range(2):
        x_graph.copy_(x0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(x_graph, x_eager)


def test_generate_audio_graph_replay_equals_eager_launches(n16):
    """generate_audio(graph=True) replays the loop from a HIP graph whose seed lives in HBM (D3PM_FLAG_SEED_IN_HBM):
    same ids as the eager launches, for every seed, from ONE captured graph."""
    m = n16.model
    texts, proms = n16.texts[:1], n16.proms[:1]
    outs, n_graphs = {}, None
    for seed in (3, 4, 3):
        eager = m.generate_audio(texts, proms, steps=12, seed=seed, graph=False)
        graphed = m.generate_audio(texts, proms, steps=12, seed=seed, graph=True)
        assert torch.equal(eager, graphed), seed
        outs.setdefault(seed, graphed)
        n_graphs = n_graphs or len(m.sampler()._graphs)
    assert not torch.equal(outs[3], outs[4])
    assert len(m.sampler()._graphs) == n_graphs                      # new seeds did not capture new graphs


@pytest.mark.parametrize("tag", ["f16", "f32"])
def test_training_forward_loss_matches_the_reference(tag, n16, n32):
    """SURVEY §8f row 3, forward half: AR.forward = q_sample (Philox stream 1) -> denoiser -> masked cross-entropy,
    99 steps, against the loss the reference's own forward() produced on these inputs (tests/golden/native_forward.npz)
    and against the oracle's last-step logits."""
    n = n16 if tag == "f16" else n32
    g = load("native_forward.npz")
    resps = torch.from_numpy(g["resps"].astype(np.int64))
    seed = int(g["seed"])
    last = n.model([n.texts[0]], [n.proms[0]], [resps], seed=seed)
    loss = float(n.model.loss)
    ref = float(g[f"loss_{tag}"])
    REPORT[f"training_forward_loss_{tag}"] = {"hip": loss, "reference": ref}
    assert abs(loss - ref) < (2e-4 if tag == "f32" else 4e-3), (loss, ref)
    rows = load("native_step.npz")["rows"]
    ref_rows = torch.from_numpy(g["last_logits_rows_f32"]) if tag == "f32" else f16(g["last_logits_rows_f16"])
    got = last[rows].float().cpu()
    assert last.shape == (n.cfg.canvas, n.cfg.n_classes) and bool((last[300:] == 0).all())
    # the noised canvas of the last step must be the reference's (integer path), so the logits agree to rounding
    assert (got - ref_rows.float()).abs().max().item() < (1e-3 if tag == "f32" else 2e-2)
    # two utterances = two independent single-utterance runs, averaged
    two = n.model([n.texts[0], n.texts[1]], [n.proms[0], n.proms[1]], [resps, resps[:200]], seed=seed)
    assert two.shape == last.shape and torch.isfinite(n.model.loss)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_ce_loss_rows_against_torch(n32, dtype):
    """d3pm_ce_loss_rows = F.cross_entropy(logits * mask, targets * mask, reduction='none') in fp32 (ar_discrete.py:683-690)."""
    cfg, smp = n32.cfg, n32.smp
    g = torch.Generator(device="cpu").manual_seed(3)
    logits = (torch.randn(2, cfg.canvas, cfg.n_classes, generator=g) * 3).to(dtype).to(DEV)
    x0 = torch.randint(0, cfg.n_classes, (2, cfg.canvas), generator=g).to(torch.int32).to(DEV)
    mask = (torch.rand(cfg.canvas, generator=g) < 0.7).to(torch.uint8).to(DEV)
    targets = (x0 * mask[None].to(torch.int32)).contiguous()
    got = smp.ce_loss_rows(logits, targets, mask)
    x = logits.float() * mask[None, :, None].float()
    ref = torch.nn.functional.cross_entropy(x.reshape(-1, cfg.n_classes), targets.reshape(-1).long(), reduction="none").reshape(2, -1)
    assert (got - ref).abs().max().item() < 2e-5
    assert torch.allclose(got[:, mask == 0], torch.full_like(got[:, mask == 0], float(np.log(cfg.n_classes))), atol=1e-6)


def test_fp8_fast_path_agreement_with_the_16_bit_path():
    """BASELINE.json configs[4] (block-scaled fp8 QKV / cross-query / fc1 / fc2 operands, 50-step schedule): no reference
    counterpart exists, so the test reports agreement against the bf16 path on the same inputs -- logits, teacher-forced
    sampled ids (same x_t, same noise) and the free-running 49-iteration loop -- and checks the mode is deterministic and
    really engaged.  Both forms of the mode are measured: fc2 left 16-bit, and fc2 on MX operands too (the shipped fast path).
    Asserted quality: teacher-forced id agreement over 12 000 draws >= 0.998 (measured 0.99933 / 0.99925; single draws of 3 000
    measured 0.9987 .. 0.9997: one to four near-tie flips); logits relative L2 error <= 0.033 / 0.04 (measured 0.0308 / 0.0367: the error of rounding three / four operand pairs per block to e4m3 -- 3 mantissa bits --
    which no choice of scales changes: per-row fp32 scales measured 0.0297 in round 2)."""
    import dataclasses
    from vall_e.vall_e import synth
    cfg = dataclasses.replace(synth.D3PMConfig.libritts(), timesteps=50)
    m = make_model(cfg, synth.make_state_dict(cfg, 0), torch.bfloat16)
    smp = m.sampler()
    texts, proms = synth.make_inputs(cfg, 4, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(4)
    gen = torch.Generator(device="cpu").manual_seed(17)      # fixed inputs: the agreement rates are statistics of a few flips per thousand
    x[:, ::2] = torch.randint(0, 1024, x[:, ::2].shape, generator=gen, dtype=x.dtype).to(x.device)      # a half-denoised canvas
    t = 25
    l16, _ = smp.denoise(x, fm, t, kv_t, kv_p)
    live = slice(0, cfg.n_frames)
    a16 = m.generate_audio(texts, proms, seed=9)[:, live]
    for tag, fc2 in (("fp8_fast_path_fc2_16bit", False), ("fp8_fast_path", True)):
        smp.fp8_fc2, smp._fp8 = fc2, None                              # (re-)quantise the weights for this form
        l8, _ = smp.denoise(x, fm, t, kv_t, kv_p, fp8=True)
        assert not torch.equal(l16, l8)                                   # the fp8 kernels ran
        d = (l8.float() - l16.float())[:, live]
        rel = d.norm().item() / l16.float()[:, live].norm().item()
        top1 = (l8[:, live].float().argmax(-1) == l16[:, live].float().argmax(-1)).float().mean().item()
        same = total = 0
        for tt, sd in ((40, 5), (25, 6), (25, 7), (10, 8)):          # teacher-forced: the same x_t and the same noise through both paths
            la, _ = smp.denoise(x, fm, tt, kv_t, kv_p)
            lb, _ = smp.denoise(x, fm, tt, kv_t, kv_p, fp8=True)
            na, _ = smp.posterior_sample(la, x, tt, seed=sd)
            nb, _ = smp.posterior_sample(lb, x, tt, seed=sd)
            same += int((na[:, live] == nb[:, live]).sum())
            total += na[:, live].numel()
        forced = same / total
        a8 = m.generate_audio(texts, proms, seed=9, fp8=True)[:, live]
        free = (a16 == a8).float().mean().item()
        REPORT[tag] = {"logits_rel_l2_err": rel, "logits_top1_agreement": top1, "teacher_forced_id_agreement": forced,
                       "free_running_49_iterations_id_agreement": free}
    for tag, lim in (("fp8_fast_path_fc2_16bit", 0.033), ("fp8_fast_path", 0.04)):
        r = REPORT[tag]
        assert r["logits_rel_l2_err"] <= lim and r["teacher_forced_id_agreement"] >= 0.998, (tag, r)
    assert torch.equal(a8, m.generate_audio(texts, proms, seed=9, fp8=True)[:, live])
    assert int(a8.min()) >= 0 and int(a8.max()) <= 1024


def test_fp8_with_16_bit_fc2_at_the_throughput_batch():
    """The fp8 mode with fc2 left a 16-bit GEMM, at 32 utterances: the persistent block-scaled GEMM re-reads the LayerNorm block
    scales for every tile it walks, and fc1's 16-bit output [n][2048] covers the whole shared workspace region -- the scales
    once lived inside it and were overwritten from the second round of tiles on (at 4 utterances every workgroup had read them
    before any store, so the small test passed by timing).  They have a workspace slot of their own now (csrc/d3pm_api.hip: ws.mxs)
    and d3pm_op_linear_mx refuses operands that overlap its outputs.  Checked: the logits error against the 16-bit path is the
    4-utterance test's, and four utterances of the batch run alone (told the global batch) give the same bits."""
    import dataclasses
    from vall_e.vall_e import _hip, synth
    cfg = dataclasses.replace(synth.D3PMConfig.libritts(), timesteps=50)
    m = make_model(cfg, synth.make_state_dict(cfg, 0), torch.bfloat16)
    smp = m.sampler()
    smp.fp8_fc2, smp._fp8 = False, None
    texts, proms = synth.make_inputs(cfg, 32, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(32)
    gen = torch.Generator(device="cpu").manual_seed(17)
    x[:, ::2] = torch.randint(0, 1024, x[:, ::2].shape, generator=gen, dtype=x.dtype).to(x.device)
    live = slice(0, cfg.n_frames)
    l16 = smp.denoise(x, fm, 25, kv_t, kv_p)[0].clone()
    l8 = smp.denoise(x, fm, 25, kv_t, kv_p, fp8=True)[0].clone()
    rel = (l8.float() - l16.float())[:, live].norm().item() / l16.float()[:, live].norm().item()
    per_utt = ((l8.float() - l16.float())[:, live].flatten(1).norm(dim=1) / l16.float()[:, live].flatten(1).norm(dim=1)).cpu()
    REPORT["fp8_fc2_16bit_batch32"] = {"logits_rel_l2_err": rel, "worst_utterance_rel_l2_err": per_utt.max().item()}
    assert rel <= 0.036 and per_utt.max().item() <= 0.05, (rel, per_utt)
    sl = slice(20, 24)                                   # rows the later tile rounds handle
    kv_ts, kv_ps = smp.cond_kv(ct[sl].contiguous(), cp[sl].contiguous())
    with _hip.tuning(regime_batch=32):
        l8s = smp.denoise(x[sl].contiguous(), fm, 25, kv_ts, kv_ps, fp8=True)[0]
    assert torch.equal(l8s, l8[sl]), "fp8 logits of a shard differ from the same utterances inside the 32-utterance batch"
    # the overlap check of the single op: scales placed inside the output range are refused
    M, N, K = 192, 2048, 512
    x8 = torch.zeros((M, K), dtype=torch.uint8, device=DEV)
    w8 = torch.zeros((N, K), dtype=torch.uint8, device=DEV)
    sw = torch.full((N, 4, K // 128), 127, dtype=torch.uint8, device=DEV)
    buf = torch.zeros(M * N * 2 + 4096, dtype=torch.uint8, device=DEV)
    y = buf[: M * N * 2].view(torch.bfloat16).view(M, N)
    sx_inside = buf[1024: 1024 + M * 16].view(M, 4, 4)
    with pytest.raises(_hip.D3PMError):
        import ctypes as C
        _hip.check(_hip.lib().d3pm_op_linear_mx(_hip.BF16, _hip._p(x8), K, _hip._p(sx_inside), _hip._p(w8), _hip._p(sw), None, _hip._p(y), N, None, 0,
                                                None, 1, None, None, M, N, K, 0, C.byref(_hip.TUNING), _hip.stream_ptr()), "d3pm_op_linear_mx")
