"""Kernel-level numerics on the GPU: each HIP kernel (through the d3pm_op_* C entry points) against a
plain PyTorch fp32 reference of the same op, and the MFMA family against the generic family.

Tolerances: fp32 kernels 1e-4 relative to the output scale; 16-bit kernels must agree with the
fp32 reference to within 2 units in the last place of the *storage* type on 99.5 % of the elements
(fp32 accumulation, one rounding per op) and with each other (generic vs MFMA: different summation
order only) likewise.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eps(dtype):
    return {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2.0 ** -23}[dtype]


def assert_close_lp(got, ref32, dtype, what, ulps=2.0, frac=0.995):
    got = got.float()
    scale = ref32.abs().clamp_min(ref32.abs().max() * 1e-3)
    rel = (got - ref32).abs() / scale
    ok = (rel <= ulps * _eps(dtype)).float().mean().item()
    worst = rel.max().item() / _eps(dtype)
    assert ok >= frac, f"{what}: only {ok:.4f} within {ulps} ulp (worst {worst:.1f} ulp)"
    assert worst < 64, f"{what}: worst element off by {worst:.1f} ulp"


def gelu32(v):
    return 0.5 * v * (1.0 + torch.erf(v * 0.7071067811865476))


@pytest.fixture(autouse=True)
def _default_tuning():
    """Every test starts from and leaves behind the library's default schedule choices (d3pm_tuning_default): a knob left
    set by one test must not change which schedules the later tests -- or the parity report -- exercise."""
    from vall_e.vall_e import _hip
    _hip.reset_tuning()
    yield
    _hip.reset_tuning()


@pytest.fixture(params=[2, 3, 4, 5, 6, 7, 8], ids=lambda v: {2: "gemm_throughput", 3: "gemm_latency_r1", 4: "gemm_latency", 5: "gemm_one_tile",
                                                          6: "gemm_big_192x256", 7: "gemm_big_96x512", 8: "gemm_big_192x128"}[v])
def gemm_variant(request, built_lib):
    """Every shipped schedule of the MFMA GEMM must pass the same numerics (auto selection is restored afterwards)."""
    from vall_e.vall_e import _hip
    with _hip.tuning(gemm_variant=request.param):
        yield request.param


@pytest.fixture(params=[1, 2, 4, 32, 33], ids=lambda v: f"qg{v}")      # 4: the key-split latency kernel; 32 / 33: the 32 x 32 x 16 kernel (pipelined / plain) where it applies
def attn_qg(request, built_lib):
    from vall_e.vall_e import _hip
    with _hip.tuning(attn_query_groups=request.param):
        yield request.param


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,ldy", [(448, 512, 512, None), (1536, 1536, 512, None), (1536, 512, 2048, None),
                                       (448, 1025, 512, 1032), (200, 96, 64, None), (448, 2048, 512, None),
                                       (49344, 512, 256, None), (576, 1024, 384, 1040)])
def test_linear_mfma_vs_generic_vs_torch(gemm_variant, dtype, M, N, K, ldy):
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g)).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    ref = x.float() @ w.float().T + b.float()
    for fam in (_hip.FAMILY_GENERIC, _hip.FAMILY_MFMA):
        y = _hip.op_linear(x, w, b, family=fam, ldy=ldy)
        assert_close_lp(y, ref, dtype, f"linear fam{fam} {M}x{N}x{K}")
    yg = _hip.op_linear(x, w, b, family=_hip.FAMILY_GENERIC, ldy=ldy).float()
    ym = _hip.op_linear(x, w, b, family=_hip.FAMILY_MFMA, ldy=ldy).float()
    assert (yg == ym).float().mean().item() > 0.97           # same rounding points, different sum order


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_epilogues(gemm_variant, dtype):
    """bias + exact-erf GELU; double residual with the eager rounding order; in-place residual; row mask."""
    from vall_e.vall_e import _hip
    M, N, K, T = 896, 512, 512, 448
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    r2 = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    mask = (torch.arange(T) < 350).to(torch.uint8).to(DEV)
    lin = (x.float() @ w.float().T + b.float()).to(dtype).float()
    outs = {}
    for fam in (_hip.FAMILY_GENERIC, _hip.FAMILY_MFMA):
        y = _hip.op_linear(x, w, b, act=1, family=fam)
        assert_close_lp(y, gelu32(lin), dtype, f"gelu fam{fam}", ulps=3.0)
        y2 = _hip.op_linear(x, w, b, r1=r1, r2=r2, family=fam)
        ref2 = ((r1.float() + r2.float()).to(dtype).float() + lin)
        assert_close_lp(y2, ref2, dtype, f"double residual fam{fam}")
        xin = r1.clone()
        y3 = _hip.op_linear(x, w, b, r1=xin, row_mask=mask, mask_period=T, family=fam, out=xin)
        ref3 = (r1.float() + lin) * mask.float().repeat(M // T)[:, None]
        assert_close_lp(y3, ref3, dtype, f"in-place residual + mask fam{fam}")
        assert (y3[350:448] == 0).all() and (y3[448 + 350:] == 0).all()
        outs[fam] = (y.float(), y2.float(), y3.float())
    for a, c in zip(outs[_hip.FAMILY_GENERIC], outs[_hip.FAMILY_MFMA]):
        assert (a == c).float().mean().item() > 0.97


def test_gemm_schedules_are_bit_identical(built_lib):
    """The latency and throughput schedules accumulate in the same order: choosing by M never changes a result."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(1536, 512, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(1536, 512, generator=g) / math.sqrt(512)).to(torch.bfloat16).to(DEV)
    b = torch.randn(1536, generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    for v in (2, 3, 4, 5):
        with _hip.tuning(gemm_variant=v):
            outs.append(_hip.op_linear(x, w, b, act=1, family=_hip.FAMILY_MFMA).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "r1", "r1r2", "r1mask", "nobias"])
@pytest.mark.parametrize("M,N,K", [(9600, 1536, 512), (2496, 512, 2048), (192, 512, 256)])
def test_gemm_big_tiles_match_one_tile_kernel(built_lib, dtype, epi, M, N, K):
    """The big-tile persistent schedules (192 x 256 / 96 x 512 tiles of eight waves, 192 x 128 tiles of four;
    d3pm_mfma_gemm_big.hip; the experimental arms are compared the same way in tests/ab_bit_identity.py) against the
    128 x 128 one-tile-per-workgroup kernel, bit for bit, for every epilogue: 300 / 300 tiles (a second tile for some
    workgroups, stores in flight into it), a long-K single round, and the smallest legal shape (K = 4 k-steps)."""
    from vall_e.vall_e import _hip
    T = 96
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    b = None if epi == "nobias" else (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi.startswith("r1") else None
    r2 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi == "r1r2" else None
    mask = (torch.rand(T, generator=g) < 0.8).to(torch.uint8).to(DEV) if epi == "r1mask" else None
    outs, arms = [], [5, 6, 7, 8]
    for v in arms:
        with _hip.tuning(gemm_variant=v):
            for rep in range(2):        # twice: a race between the DMA pieces and the fragment reads would not repeat
                outs.append(_hip.op_linear(x, w, b, act=1 if epi == "gelu" else 0, r1=r1, r2=r2, row_mask=mask, mask_period=T,
                                           family=_hip.FAMILY_MFMA).clone())
    for i, o in enumerate(outs[1:]):
        assert torch.equal(outs[0], o), f"arm {arms[(i + 1) // 2]} differs on {(outs[0] != o).float().mean().item():.2e} of the elements"
    if epi in ("bias", "nobias"):
        ref = x.float() @ w.float().T + (b.float() if b is not None else 0)
        assert_close_lp(outs[0], ref, dtype, f"big tile reference {M}x{N}x{K}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi", ["bias", "gelu", "r1", "r1r2", "r1mask", "nobias"])
def test_gemm_persistent_matches_one_tile_kernel(built_lib, dtype, epi):
    """>= 512 whole tiles: the throughput schedule runs its persistent kernel; every epilogue must equal the
    one-tile-per-workgroup launch bit for bit."""
    from vall_e.vall_e import _hip
    M, N, K, T = 40 * 128, 14 * 128, 512, 512
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    b = None if epi == "nobias" else (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi.startswith("r1") else None
    r2 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi == "r1r2" else None
    mask = (torch.rand(T, generator=g) < 0.8).to(torch.uint8).to(DEV) if epi == "r1mask" else None
    outs = []
    for v in (5, 2, 0, 3, 4):
        with _hip.tuning(gemm_variant=v):
            outs.append(_hip.op_linear(x, w, b, act=1 if epi == "gelu" else 0, r1=r1, r2=r2, row_mask=mask, mask_period=T,
                                       family=_hip.FAMILY_MFMA).clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = x.float() @ w.float().T + (b.float() if b is not None else 0)
    if epi == "bias" or epi == "nobias":
        assert_close_lp(outs[1], ref, dtype, f"persistent {epi}")


def test_linear_fp32_generic(built_lib):
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(448, 32, generator=g).to(DEV)
    w = torch.randn(96, 32, generator=g).to(DEV)
    b = torch.randn(96, generator=g).to(DEV)
    y = _hip.op_linear(x, w, b)
    assert (y - (x @ w.T + b)).abs().max().item() < 1e-4


def torch_attention(q, k, v, H, scale):
    B, Tq, d = q.shape
    S, hd = k.shape[1], d // H
    qh = (q.float() * scale).view(B, Tq, H, hd).transpose(1, 2)
    kh = k.float().view(B, S, H, hd).transpose(1, 2)
    vh = v.float().view(B, S, H, hd).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, d)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Tq,S", [(448, 448), (448, 50), (448, 398), (768, 768), (768, 225), (128, 1), (64, 65), (384, 384), (128, 64)])
def test_attention_mfma_vs_generic_vs_torch(attn_qg, dtype, Tq, S):
    from vall_e.vall_e import _hip
    B, H, hd = 2, 8, 64
    d = H * hd
    g = torch.Generator(device="cpu").manual_seed(Tq * 1000 + S)
    q = torch.randn(B, Tq, d, generator=g).to(dtype).to(DEV)
    kv = torch.randn(B, S, 2 * d, generator=g).to(dtype).to(DEV)        # packed K|V rows like the cond cache
    k, v = kv[..., :d], kv[..., d:]
    scale = math.sqrt(1.0 / hd)
    ref = torch_attention(q, k, v, H, scale)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    for fam in (_hip.FAMILY_GENERIC, _hip.FAMILY_MFMA):
        o = _hip.op_attention(q, k, v, H, scale, family=fam).float()
        err = (o - ref).abs().max().item()
        assert err < tol, f"attention fam{fam} Tq={Tq} S={S}: max abs err {err}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("resident", [0, 4, 5, -4], ids=["tile_by_tile", "resident16", "resident32", "key_split"])
@pytest.mark.parametrize("Tq,S1,S2", [(768, 50, 225), (448, 50, 200), (384, 64, 256), (300, 1, 33), (768, 33, 97), (256, 32, 1), (512, 17, 160)])
def test_attention_pair_vs_torch(built_lib, dtype, resident, Tq, S1, S2):
    """The text / prompt cross-attention pair of a block (d3pm_op_attention_pair) on every schedule -- tile by tile, both K / V
    images resident on the 16 x 16 x 32 instruction, the opt-in key-split kernel, resident on the 32 x 32 x 16 instruction with the
    software-pipelined 32-key block (key counts that end inside a block are masked through the product's C operand; blocks past the last key are not
    walked) -- against torch fp32 on the same 16-bit inputs, K / V as views of packed cache rows."""
    from vall_e.vall_e import _hip
    B, H, hd = 3, 8, 64
    d = H * hd
    g = torch.Generator(device="cpu").manual_seed(Tq * 7 + S1 * 3 + S2)
    q1 = torch.randn(B, Tq, d, generator=g).to(dtype).to(DEV)
    q2 = torch.randn(B, Tq, d, generator=g).to(dtype).to(DEV)
    kv1 = (1.5 * torch.randn(B, S1, 2 * d, generator=g)).to(dtype).to(DEV)
    kv2 = (1.5 * torch.randn(B, S2, 2 * d, generator=g)).to(dtype).to(DEV)
    scale = math.sqrt(1.0 / hd)
    ref1 = torch_attention(q1, kv1[..., :d], kv1[..., d:], H, scale)
    ref2 = torch_attention(q2, kv2[..., :d], kv2[..., d:], H, scale)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    # -4: the opt-in key-split kernel (attn_query_groups = 4), pair = the two halves of its grid
    knobs = {"attn_cross_resident": 0, "attn_query_groups": 4} if resident == -4 else {"attn_cross_resident": resident}
    with _hip.tuning(**knobs):
        o1, o2 = _hip.op_attention_pair(q1, kv1[..., :d], kv1[..., d:], q2, kv2[..., :d], kv2[..., d:], H, scale)
    for name, o, ref in (("text", o1, ref1), ("prompt", o2, ref2)):
        assert torch.isfinite(o).all()
        err = (o.float() - ref).abs().max().item()
        assert err < tol, f"{name} problem Tq={Tq} S={S1}/{S2} resident={resident}: max abs err {err}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attention_running_reference_moves_late(built_lib, dtype):
    """Scores that keep growing along the keys (and a burst in the last tile): the deferred running maximum of the flash kernels
    has to be raised in the middle of the walk -- in the pipelined 32 x 32 x 16 kernel that is the rare path behind P.V(j) and
    in front of the product of block j + 2.  Against torch fp32 on the same 16-bit inputs; the softmax is close to one-hot here,
    so the error is that of the 16-bit scores: the bound is the generic kernel's error (eager rounding points) times two."""
    from vall_e.vall_e import _hip
    B, H, hd, T = 2, 8, 64, 384
    d = H * hd
    g = torch.Generator(device="cpu").manual_seed(5)
    q = torch.randn(B, T, d, generator=g)
    k = torch.randn(B, T, d, generator=g) * torch.linspace(0.5, 4.0, T).view(1, T, 1)
    k[:, -40:] *= 2.5
    v = torch.randn(B, T, d, generator=g)
    q, k, v = (t.to(dtype).to(DEV) for t in (q, k, v))
    scale = math.sqrt(1.0 / hd)
    ref = torch_attention(q, k, v, H, scale)
    bound = 2.0 * (_hip.op_attention(q, k, v, H, scale, family=_hip.FAMILY_GENERIC).float() - ref).abs().max().item()
    for qg in (2, 4, 32, 33):
        with _hip.tuning(attn_query_groups=qg):
            o = _hip.op_attention(q, k, v, H, scale).float()
        assert torch.isfinite(o).all()
        err = (o - ref).abs().max().item()
        assert err < bound, f"qg{qg} {dtype}: max abs err {err} (generic kernel x 2: {bound})"


def test_attention_self_packed_qkv_and_tiny_heads(built_lib):
    """The self-attention call reads q/k/v out of one [N,3d] projection; head_dim 2 is the upstream shape."""
    from vall_e.vall_e import _hip
    for dtype, H, hd, tol in ((torch.float32, 16, 2, 2e-5), (torch.float16, 16, 2, 3e-3), (torch.float16, 8, 64, 4e-3)):
        B, T, d = 2, 448, H * hd
        g = torch.Generator(device="cpu").manual_seed(hd)
        qkv = torch.randn(B, T, 3 * d, generator=g).to(dtype).to(DEV)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        scale = math.sqrt(1.0 / hd)
        o = _hip.op_attention(q, k, v, H, scale).float()
        assert (o - torch_attention(q, k, v, H, scale)).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_layernorm_and_film(built_lib, dtype):
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(3)
    for d in (32, 512):
        x = torch.randn(448, d, generator=g).to(dtype).to(DEV)
        w = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
        b = (0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
        film = (0.2 * torch.randn(2 * d, generator=g)).to(dtype).to(DEV)
        ref = torch.nn.functional.layer_norm(x.float(), (d,), w.float(), b.float(), 1e-6)
        y = _hip.op_layernorm(x, w, b)
        tol = 1e-5 if dtype == torch.float32 else 4 * _eps(dtype) * 4
        assert (y.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
        yf = _hip.op_layernorm(x, w, b, film=film).float()
        reff = ref.to(dtype).float() * (1 + film[:d].float()).to(dtype).float()
        reff = reff.to(dtype).float() + film[d:].float()
        assert (yf - reff).abs().max().item() < tol * 2 * max(1.0, reff.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 5, 33, 450, 1027, 4100])
def test_layernorm_row_counts_that_do_not_fill_the_xcd_ranges(built_lib, dtype, M):
    """layernorm_vec deals its 4-row workgroups to the 8 XCDs in contiguous ranges (blockIdx -> row remap): every row count --
    fewer workgroups than XCDs, a remainder of 1 .. 7 workgroups, a ragged last workgroup -- must still write every row once."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(M)
    d = 512
    x = torch.randn(M, d, generator=g).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    b = (0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    ref = torch.nn.functional.layer_norm(x.float(), (d,), w.float(), b.float(), 1e-6)
    y = _hip.op_layernorm(x, w, b)
    assert (y.float() - ref).abs().max().item() < 16 * _eps(dtype) * max(1.0, ref.abs().max().item())


# ---- fp8 (OCP e4m3) fast path, BASELINE.json configs[4]: no reference counterpart, pinned to torch on the SAME codes ----
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_quantize_mx_is_the_block_rule(built_lib, dtype):
    """d3pm_op_quantize_mx (blocks of 32 along K, power-of-two scales, e4m3 codes) against the same rule in torch
    (_hip.quantize_mx): scales and codes bit for bit, incl. blocks of very different magnitude, all-zero blocks and a block
    whose maximum sits exactly on the 448 / 1.75 boundary of the exponent choice."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(3)
    M, K = 384, 2048
    x = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-12, 12, (M, K // 32), generator=g).float()).repeat_interleave(32, dim=1)
    x[3] = 0
    x[4, :32] = 0
    x[5, :32] = torch.tensor([1.75 * 2.0 ** -3] + [0.01] * 31)        # amax = 1.75 x 2^E exactly: scale 2^(E - 8), code 448
    x[6, :32] = torch.tensor([1.8125 * 2.0 ** -3] + [0.01] * 31)      # just above: the next power of two
    x = x.to(dtype).to(DEV)
    x8, sx = _hip.op_quantize_mx(x)
    r8, rs = _hip.quantize_mx(x)
    assert sx.shape == (M, 4, K // 128) and torch.equal(sx, rs)
    assert torch.equal(x8, r8), f"{(x8 != r8).float().mean().item():.2e} of the codes differ"
    deq = _hip.dequantize_mx(x8, sx)
    xf = x.float()
    blockmax = xf.abs().reshape(M, K // 32, 32).amax(dim=-1).repeat_interleave(32, dim=1)
    assert ((deq - xf).abs() <= 0.0625 * xf.abs() + 2.0 ** -9 * blockmax + 1e-30).all()      # e4m3: 3 mantissa bits, 2^-9 of the block scale x 448
    assert x8[5, 0].item() == 0x7E and x8[6, 0].item() < 0x7E                               # 448 = 0x7E; never the NaN code 0x7F
    assert (x8 & 0x7F).max().item() <= 0x7E


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_mx_is_the_layernorm_then_the_block_rule(built_lib, dtype):
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(21)
    M, d = 1024, 512
    x = (torch.randn(M, d, generator=g) * 2).to(dtype).to(DEV)
    x[5] = 0                                                           # a masked frame: LN(0) = bias
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    b = (0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    film = (0.2 * torch.randn(2 * d, generator=g)).to(dtype).to(DEV)
    for fl in (None, film):
        y16 = _hip.op_layernorm(x, w, b, film=fl)                      # the 16-bit result the MX row is derived from
        y8, sx = _hip.op_layernorm_mx(x, w, b, film=fl)
        r8, rs = _hip.quantize_mx(y16)
        assert torch.equal(sx, rs) and torch.equal(y8, r8)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,epi", [(192, 256, 512, "bias"), (384, 128, 512, "bias"), (1920, 1536, 512, "bias"), (2304, 2048, 512, "gelu"),
                                       (768, 512, 2048, "r1mask"), (960, 512, 2048, "r1"), (24576, 1536, 512, "bias"), (576, 384, 1024, "nobias")])
def test_linear_mx_against_fp32_on_the_same_codes(built_lib, dtype, M, N, K, epi):
    """Products of two e4m3 numbers and power-of-two scales are exact in fp32, so the block-scaled kernel must reproduce a torch
    fp32 evaluation of the de-quantised operands up to summation order and the final 16-bit rounding.  The operands carry
    blocks of very different magnitude (2^-6 .. 2^6 per 32 elements): a scale applied to the wrong block, row or k-step -- the
    lane / op_sel map of v_mfma_scale_f32_16x16x128_f8f6f4 -- shows up as an O(1) error.  Twice: a race between the DMA pieces
    and the fragment reads would not repeat."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    T = 96

    def blocky(r, c, sc):
        return torch.randn(r, c, generator=g) * sc * torch.exp2(torch.randint(-6, 7, (r, c // 32), generator=g).float()).repeat_interleave(32, dim=1)
    x8, sx = _hip.quantize_mx(blocky(M, K, 1.0).to(DEV))
    w8, sw = _hip.quantize_mx(blocky(N, K, 1.0 / math.sqrt(K)).to(DEV))
    b = None if epi == "nobias" else (torch.randn(N, generator=g) * 0.1).to(dtype).to(DEV)
    r1 = torch.randn(M, N, generator=g).to(dtype).to(DEV) if epi.startswith("r1") else None
    mask = (torch.rand(T, generator=g) < 0.8).to(torch.uint8).to(DEV) if epi == "r1mask" else None
    y = _hip.op_linear_mx(x8, sx, w8, sw, b, dtype, act=1 if epi == "gelu" else 0, r1=r1, row_mask=mask, mask_period=T)
    y2 = _hip.op_linear_mx(x8, sx, w8, sw, b, dtype, act=1 if epi == "gelu" else 0, r1=r1, row_mask=mask, mask_period=T)
    assert torch.equal(y, y2)
    ref = _hip.dequantize_mx(x8, sx) @ _hip.dequantize_mx(w8, sw).T + (b.float() if b is not None else 0)
    if epi == "gelu":
        ref = torch.nn.functional.gelu(ref.to(dtype).float())
    if r1 is not None:
        ref = (ref.to(dtype).float() + r1.float())
    if mask is not None:
        ref = ref * mask.float().repeat(M // T)[:, None]
    assert_close_lp(y, ref, dtype, f"mx linear {M}x{N}x{K} {epi}")
    if epi in ("bias", "gelu", "nobias"):
        # the MX-output epilogue (fc1 -> fc2): fp32 -> e4m3 directly (no 16-bit rounding in between) and the tanh form of GELU
        # (|gelu_tanh - gelu_erf| < 5e-4): every element within the e4m3 grid of the fp32 reference -- half a step of 2^-3
        # relative, 2^-9 of the block maximum for the subnormal range -- plus that slack; block scales = the rule on the block's
        # own maximum (a maximum on the edge of a power of two may land on either side: one exponent step allowed)
        y8, sy = _hip.op_linear_mx(x8, sx, w8, sw, b, dtype, act=1 if epi == "gelu" else 0, mx_out=True)
        ref32 = _hip.dequantize_mx(x8, sx) @ _hip.dequantize_mx(w8, sw).T + (b.float() if b is not None else 0)
        if epi == "gelu":
            ref32 = torch.nn.functional.gelu(ref32)
        deq = _hip.dequantize_mx(y8, sy)
        bm = ref32.abs().reshape(M, N // 32, 32).amax(dim=-1)
        tol = 0.0625 * ref32.abs() + (2.0 ** -9 * 1.01) * bm.repeat_interleave(32, dim=1) + 1e-3
        assert ((deq - ref32).abs() <= tol).all(), f"{((deq - ref32).abs() - tol).max().item()}"
        rs = _hip.mx_scale_bytes(bm).reshape(M, N // 128, 4).permute(0, 2, 1)
        assert ((sy.int() - rs.int()).abs() <= 1).all() and (sy == rs).float().mean().item() > 0.98
        assert (y8 & 0x7F).max().item() <= 0x7E


def test_linear_mx_rejects_what_it_cannot_run(built_lib):
    from vall_e.vall_e import _hip
    x8 = torch.zeros(100, 512, dtype=torch.uint8, device=DEV)
    sx = torch.zeros(100, 4, 4, dtype=torch.uint8, device=DEV)
    w8 = torch.zeros(256, 512, dtype=torch.uint8, device=DEV)
    sw = torch.zeros(256, 4, 4, dtype=torch.uint8, device=DEV)
    with pytest.raises(_hip.D3PMError):
        _hip.op_linear_mx(x8, sx, w8, sw, None, torch.bfloat16)            # M = 100 is not a multiple of 192









@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("config", ["libritts", "wide448"])
def test_cross_attention_pair_resident_kernel_equals_the_tile_by_tile_kernel(built_lib, dtype, config):
    """attn_cross_hd64 (all K / V tiles of the text and prompt problems resident in LDS; attn_cross_resident = 4) runs the per-tile
    arithmetic of attn_mfma_hd64 in the same order: a denoise step must give the same hidden state and logits bit for bit, for whole
    and ragged key tiles (50 / 225 keys; 50 / 398 keys does not fit and must fall back) and a canvas that is not a multiple
    of the 256-query block (448)."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts() if config == "libritts" else synth.D3PMConfig(d_model=512, n_heads=8, n_layers=2, s_prompt=200)
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    texts, proms = synth.make_inputs(cfg, 3, 1)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(3)
    x[:, ::2] = torch.randint(0, 1024, x[:, ::2].shape, device=x.device, dtype=x.dtype)
    outs = []
    for on in (0, 4):                       # 4: resident on the 16 x 16 x 32 instruction (the 32 x 32 x 16 form accumulates in another order)
        with _hip.tuning(attn_cross_resident=on):
            lg, hid = smp.denoise(x, fm, 30, kv_t, kv_p, want_hidden=True)
        outs.append((lg.clone(), hid.clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0])




@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("form,M,K", [("self_out", 96 * 300, 512), ("cross_out", 96 * 300, 512), ("mlp_down", 96 * 258, 2048),
                                      ("self_out", 96, 256), ("cross_out", 192, 512), ("mlp_down", 96 * 3, 256)])
def test_row_panel_projection_equals_linear_then_layernorm(built_lib, dtype, form, M, K):
    """d3pm_op_linear_rowpanel (projection onto the residual stream + the LayerNorms of the new rows, one launch) against the
    launches it replaces inside the block -- d3pm_op_linear, then d3pm_op_layernorm: same bits in x and in every LayerNorm
    output, for one round of tiles, more tiles than CUs (persistent walk) and a handful of tiles."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(M + K)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype).to(DEV)
    x, x2, r1 = mk(M, K), mk(M, K), mk(M, 512, sc=2.0)
    w, b = mk(512, K, sc=1.0 / math.sqrt(K)), mk(512, sc=0.3)
    lw, lb, lw2, lb2 = mk(512, sc=0.5) + 1, mk(512, sc=0.2), mk(512, sc=0.5) + 1, mk(512, sc=0.2)
    film = mk(1024, sc=0.3)
    period = 96 * 3 if M % (96 * 3) == 0 else M
    mask = (torch.rand(period, generator=g) < 0.9).to(torch.uint8).to(DEV)
    if form == "self_out":
        y, l1, l2 = _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, ln2_w=lw2, ln2_b=lb2)
        y_ref = _hip.op_linear(x, w, b, r1=r1, family=_hip.FAMILY_MFMA)
        refs = [_hip.op_layernorm(y_ref, lw, lb), _hip.op_layernorm(y_ref, lw2, lb2)]
        outs = [l1, l2]
    elif form == "cross_out":
        y, l1, l2 = _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, x2=x2, film=film)
        h = _hip.op_linear(x, w, b, family=_hip.FAMILY_MFMA)
        y_ref = _hip.op_linear(x2, w, b, r1=r1, r2=h, family=_hip.FAMILY_MFMA)
        refs, outs = [_hip.op_layernorm(y_ref, lw, lb, film=film)], [l1]
        assert l2 is None
    else:
        y, l1, l2 = _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, row_mask=mask)
        y_ref = _hip.op_linear(x, w, b, r1=r1, row_mask=mask, mask_period=period, family=_hip.FAMILY_MFMA)
        refs, outs = [_hip.op_layernorm(y_ref, lw, lb)], [l1]
    assert torch.equal(y, y_ref), f"{form}: residual stream differs in {(y != y_ref).sum().item()} elements"
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o, r), f"{form}: LayerNorm output {i} differs in {(o != r).sum().item()} elements"
    if form != "mlp_down":
        # fp8 fast path: the same launch with its LayerNorm rows leaving as MX codes + block scales = the block rule applied
        # to the 16-bit LayerNorm result, bit for bit (torch-side rule: _hip.quantize_mx)
        kw = dict(ln2_w=lw2, ln2_b=lb2) if form == "self_out" else dict(x2=x2, film=film)
        y8, m1, m2 = _hip.op_linear_rowpanel(x, w, b, r1, lw, lb, mx=True, **kw)
        assert torch.equal(y8, y_ref)
        for (codes, scales), r in zip([m for m in (m1, m2) if m is not None], refs):
            rc, rs = _hip.quantize_mx(r)
            assert torch.equal(scales, rs) and torch.equal(codes, rc), f"{form}: MX LayerNorm rows differ"


def test_row_panel_rejects_other_combinations(built_lib):
    from vall_e.vall_e import _hip
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):      # 100 rows: not whole 96-row tiles
        _hip.op_linear_rowpanel(z(100, 512), z(512, 512), z(512), z(100, 512), z(512), z(512), ln2_w=z(512), ln2_b=z(512))
    with pytest.raises(RuntimeError):      # a single LayerNorm without mask / FiLM / second operand is not one of the three forms
        _hip.op_linear_rowpanel(z(96, 512), z(512, 512), z(512), z(96, 512), z(512), z(512))


def test_sample_loop_is_the_same_with_and_without_row_panel_launches(built_lib):
    """D3PM_TUNE_ROW_PANEL inside the loop: ids after a few reverse steps with every fused form on and off (bench shape rows:
    32 x 768 = 256 tiles of 96 rows)."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(torch.bfloat16).to(DEV)
    texts, proms = synth.make_inputs(cfg, 32, 1)
    outs = []
    for maskbits in (0, 7, 1, 2, 4):           # the stand-alone LayerNorm structure (ln_fold = 0): what the row panels fuse
        with _hip.tuning(row_panel=maskbits, ln_fold=0):
            outs.append(m.generate_audio(texts, proms, steps=3, seed=4).clone())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    # with the LayerNorms folded into the projections (default) the only row_panel choice left is the dual out-projection launch
    # (bit 1: big tiles, bit 3: latency GEMM) against two launches: same bits again
    folded = []
    for maskbits in (10, 0):
        with _hip.tuning(row_panel=maskbits):
            folded.append(m.generate_audio(texts, proms, steps=3, seed=4).clone())
    assert torch.equal(folded[0], folded[1])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("batch", [1, 2])
def test_dual_out_projection_of_the_latency_regime_is_bit_identical(built_lib, dtype, batch):
    """row_panel bit 8: at one or two utterances both cross-attention out-projections run as ONE launch of the latency GEMM (two
    products through one resident weight panel).  Hidden state and logits of a denoise step and the ids of a short loop must be
    the bits of the two-launch form."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.libritts()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(DEV)
    smp = m.sampler()
    texts, proms = synth.make_inputs(cfg, batch, 3)
    ct, cp = m.encode_conditions(texts, proms)
    kv_t, kv_p = smp.cond_kv(ct, cp)
    x, fm = m.canvas_init(batch)
    x[:, ::2] = torch.randint(0, 1024, x[:, ::2].shape, device=x.device, dtype=x.dtype)
    outs = []
    for maskbits, fold in ((3, 0), (11, 0), (2, 1), (10, 1)):      # unfolded pair, then the folded pair (two launches vs the dual launch)
        with _hip.tuning(row_panel=maskbits, ln_fold=fold):
            lg, hid = smp.denoise(x, fm, 30, kv_t, kv_p, want_hidden=True)
            ids = m.generate_audio(texts, proms, steps=3, seed=4)
        outs.append((lg.clone(), hid.clone(), ids.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[2], outs[3]):
        assert torch.equal(a, b)






@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("epi,M,N,K", [("bias", 768, 1536, 512), ("gelu", 768, 2048, 512), ("r1", 768, 512, 512), ("r1r2", 768, 512, 512),
                                       ("r1mask", 768, 512, 2048), ("bias", 750, 1025, 512), ("gelu", 100, 2048, 256), ("r1mask", 1500, 512, 1024)])
def test_latency_gemm_tile_geometries_are_bit_identical(built_lib, dtype, epi, M, N, K):
    """D3PM_TUNE_LAT_TILE: the 64 x 64, 96 x 64 and 32 x 64 tiles of the latency GEMM against the one-tile 128 x 128 kernel (same k
    order, same epilogue code), whole and ragged shapes."""
    from vall_e.vall_e import _hip
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dtype).to(DEV)
    x, w, b = mk(M, K), mk(N, K, sc=1.0 / math.sqrt(K)), mk(N, sc=0.3)
    kw = {}
    if epi == "gelu":
        kw["act"] = 1
    if epi in ("r1", "r1r2", "r1mask"):
        kw["r1"] = mk(M, N)
    if epi == "r1r2":
        kw["r2"] = mk(M, N)
    if epi == "r1mask":
        kw["row_mask"] = (torch.rand(250, generator=g) < 0.9).to(torch.uint8).to(DEV)
        kw["mask_period"] = 250
    if N % 8:
        kw["ldy"] = (N + 7) & ~7             # the final projection's padded logits rows
    outs = []
    with _hip.tuning(gemm_variant=5):
        outs.append(_hip.op_linear(x, w, b, family=_hip.FAMILY_MFMA, **kw).clone())
    for tile in (1, 2, 3, 0):
        with _hip.tuning(gemm_variant=4, lat_tile=tile):
            outs.append(_hip.op_linear(x, w, b, family=_hip.FAMILY_MFMA, **kw).clone())
    for i, o in enumerate(outs[1:]):
        assert torch.equal(outs[0], o), f"lat tile arm {i}: {(outs[0] != o).sum().item()} elements differ"
