"""Bit-identity of the EXPERIMENTS against the shipped launches they would replace -- libd3pm_hip_ab.so only.

Not collected by `pytest tests/` (the file name does not match test_*.py): run it explicitly on the GPU box,

    python -m pytest tests/ab_bit_identity.py -q -m gpu

It builds the A/B library on demand (`__graft_entry__.build_ab()`, -DD3PM_ABLATIONS, ~2 min) and routes vall_e.vall_e._hip to
it for this process.  Every arm here was measured slower than the shipped path and is therefore not
in libd3pm_hip.so; the tests keep the claim "same results" checkable: the fused final + sampler kernel, the bf16 GELU
table, the LayerNorm-prologue GEMM, the compiler-placed / deferred-store / non-temporal big-tile schedules and the
five-slab ring GEMM.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session")
def ab_lib():
    import __graft_entry__ as g
    g.build_ab()
    from vall_e.vall_e import _hip
    return _hip.use_ab_library()


@pytest.fixture(autouse=True)
def _defaults(ab_lib):
    from vall_e.vall_e import _hip

    def reset():
        _hip.reset_tuning()
        _hip.set_gemm_big_mode(1); _hip.set_attn_arm(0); _hip.set_gemm_ring(False); _hip.set_gelu_table(False)
        _hip.set_ln_prologue(False); _hip.set_fused_final_sample(False)
    reset()
    yield
    reset()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "r1", "r1r2", "r1mask", "nobias"])
@pytest.mark.parametrize("M,N,K", [(9600, 1536, 512), (2496, 512, 2048), (192, 512, 256)])
def test_experimental_gemm_schedules_match_the_one_tile_kernel(ab_lib, dtype, epi, M, N, K):
    """compiler-placed fragment reads (mode 0), deferred stores (9), non-temporal stores (513), all DMA pieces at the top of a
    k-step (2049) on the three big-tile geometries, and the five-slab ring GEMM, against the 128 x 128 one-tile kernel."""
    from vall_e.vall_e import _hip
    T = 96
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    b = None if epi == "nobias" else (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi.startswith("r1") else None
    r2 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi == "r1r2" else None
    mask = (torch.rand(T, generator=g) < 0.8).to(torch.uint8).to(DEV) if epi == "r1mask" else None
    arms = [(5, 1, False), (6, 0, False), (6, 1, False), (7, 0, False), (8, 0, False), (6, 9, False), (6, 513, False), (0, 1, True)]
    if epi == "bias":
        arms.append((6, 2049, False))
    outs = []
    for v, mode, ring in arms:
        _hip.set_gemm_variant(v); _hip.set_gemm_big_mode(mode); _hip.set_gemm_ring(ring)
        for rep in range(2):
            outs.append(_hip.op_linear(x, w, b, act=1 if epi == "gelu" else 0, r1=r1, r2=r2, row_mask=mask, mask_period=T,
                                       family=_hip.FAMILY_MFMA).clone())
    for i, o in enumerate(outs[1:]):
        assert torch.equal(outs[0], o), f"arm {arms[(i + 1) // 2]} differs on {(outs[0] != o).float().mean().item():.2e} of the elements"


def test_attention_arms_with_the_same_results(ab_lib):
    """three query groups per wave, hand-placed fragment reads (164), K / V tiles by direct-to-LDS DMA (228) against the shipped kernel"""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(5)
    B, T, d, H = 4, 768, 512, 8
    qkv = (torch.randn(B, T, 3 * d, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    outs = []
    for arm in (0, 3, 164, 228):
        _hip.set_attn_arm(arm)
        outs.append(_hip.op_attention(q, k, v, H, 0.125, family=_hip.FAMILY_MFMA).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_final_sample_equals_the_two_launch_path(ab_lib, dtype):
    """final projection + posterior + Gumbel-max draw in one kernel (d3pm_final_sample.hip; ar_discrete.py:773-779,401-420)
    against the final GEMM followed by the stand-alone sampler: the logits are the same fp32 sums rounded the same way and
    the draw is the same code over the same lane grouping, so the ids must be identical -- for masked and revealed tokens,
    early and late steps, with and without noise."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0, logit_gain=4.0))      # sharper logits: revealed tokens flip sometimes
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    g = torch.Generator(device="cpu").manual_seed(3)
    B = 3                                                                   # 2304 rows: 72 workgroups of 32 rows
    hidden = torch.randn(B, cfg.canvas, cfg.d_model, generator=g).to(dtype).to(DEV)
    x = torch.full((B, cfg.canvas), cfg.mask_id, dtype=torch.int32)
    x[:, ::3] = torch.randint(0, 1024, x[:, ::3].shape, generator=g, dtype=torch.int32)
    x[:, cfg.n_frames:] = 0
    x = x.to(DEV)
    for t, flags in ((99, 0), (50, 0), (1, 0), (40, _hip.FLAG_GREEDY)):
        logits = _hip.op_linear(hidden.view(-1, cfg.d_model), m.final.weight, m.final.bias, family=_hip.FAMILY_MFMA, ldy=1032)
        want, _ = smp.posterior_sample(logits.reshape(B, cfg.canvas, cfg.n_classes).contiguous(), x, t, seed=17, utt0=5, flags=flags)
        got = smp.final_sample(hidden, x, t, seed=17, utt0=5, flags=flags)
        assert torch.equal(got, want), f"t={t}: {(got != want).sum().item()} of {got.numel()} ids differ"
    assert int(want.max()) <= 1024 and int(want.min()) >= 0


def test_sample_loop_is_the_same_with_and_without_the_fused_final_kernel(ab_lib):
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(torch.bfloat16).to(DEV)
    texts, proms = synth.make_inputs(cfg, 2, 1)
    try:
        _hip.set_fused_final_sample(False)
        a = m.generate_audio(texts, proms, steps=6, seed=4)
        _hip.set_fused_final_sample(True)
        b = m.generate_audio(texts, proms, steps=6, seed=4)
    finally:
        _hip.set_fused_final_sample(False)
    assert torch.equal(a, b)


def test_gelu_table_lookup_is_bit_identical_to_the_arithmetic_epilogue(ab_lib):
    """D3PM_TUNE_GELU_TABLE (opt-in): rn_bf16(gelu(v)) from the LDS table the device fills with the arithmetic path itself."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(2)
    x = (torch.randn(1536, 512, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(2048, 512, generator=g) / math.sqrt(512)).to(torch.bfloat16).to(DEV)
    b = torch.randn(2048, generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    try:
        for variant in (6, 4):
            _hip.set_gemm_variant(variant)
            for tab in (False, True):
                _hip.set_gelu_table(tab)
                outs.append(_hip.op_linear(x, w, b, act=1, family=_hip.FAMILY_MFMA).clone())
    finally:
        _hip.set_gemm_variant(0)
        _hip.set_gelu_table(False)
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("form,m,N", [("norm1_qkv", 768, 1536), ("norm2_norm22_q", 768, 512), ("norm3_film_fc1", 768, 2048),
                                      ("norm1_qkv", 100, 1536), ("norm2_norm22_q", 128, 512), ("norm3_film_fc1", 1500, 2048)])
def test_layernorm_prologue_projection_equals_layernorm_then_linear(ab_lib, dtype, form, m, N):
    """d3pm_op_linear_lnpro (the latency GEMM normalising its operand rows in LDS) against d3pm_op_layernorm followed by
    d3pm_op_linear: same bits, for the three LayerNorm-fed projections of a block, whole and ragged row counts."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(m + N)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype).to(DEV)
    x = mk(m, 512, sc=2.0) + 0.25
    w, b = mk(N, 512, sc=1.0 / math.sqrt(512)), mk(N, sc=0.3)
    lw, lb, lw2, lb2, film = mk(512, sc=0.5) + 1, mk(512, sc=0.2), mk(512, sc=0.5) + 1, mk(512, sc=0.2), mk(1024, sc=0.3)
    try:
        _hip.set_gemm_variant(4)          # the reference launches on the same schedule family (all are bit-identical anyway)
        if form == "norm1_qkv":
            y = _hip.op_linear_lnpro(x, w, b, lw, lb)
            ref = _hip.op_linear(_hip.op_layernorm(x, lw, lb), w, b, family=_hip.FAMILY_MFMA)
        elif form == "norm2_norm22_q":
            y = _hip.op_linear_lnpro(x, w, b, lw, lb, ln2_w=lw2, ln2_b=lb2)
            h = torch.cat([_hip.op_layernorm(x, lw, lb), _hip.op_layernorm(x, lw2, lb2)])
            ref = _hip.op_linear(h, w, b, family=_hip.FAMILY_MFMA)
        else:
            y = _hip.op_linear_lnpro(x, w, b, lw, lb, film=film, act=1)
            ref = _hip.op_linear(_hip.op_layernorm(x, lw, lb, film=film), w, b, act=1, family=_hip.FAMILY_MFMA)
    finally:
        _hip.set_gemm_variant(0)
    assert y.shape == ref.shape
    assert torch.equal(y, ref), f"{form}: {(y != ref).sum().item()} of {y.numel()} elements differ"


def test_generate_audio_is_the_same_with_and_without_layernorm_prologues(ab_lib):
    """D3PM_TUNE_LN_PROLOGUE inside the loop at one and two utterances (the regime it applies to)."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(torch.bfloat16).to(DEV)
    for batch in (1, 2):
        texts, proms = synth.make_inputs(cfg, batch, 1)
        try:
            _hip.set_ln_prologue(False)
            a = m.generate_audio(texts, proms, steps=4, seed=4)
            _hip.set_ln_prologue(True)
            b = m.generate_audio(texts, proms, steps=4, seed=4)
        finally:
            _hip.set_ln_prologue(False)          # the library default (measured slower, include/d3pm_hip.h)
        assert torch.equal(a, b)
