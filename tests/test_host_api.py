"""Host-side contract of the drop-in package (no GPU): registry, state-dict layout, loud failure
without a HIP device, C-ABI symbol table, schedule arithmetic, Philox known answers."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import philox
from util import load


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
            "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, exp in kat:
        out = philox.philox4x32_10(*ctr, *key)
        assert " ".join("%08x" % int(v) for v in out) == exp


def test_uniform_stream_layout():
    u = philox.uniform_batch(123, 40, 2, 2, 448)
    assert u.shape == (2, 448, 1025) and u.dtype == np.float32
    assert (u >= 0).all() and (u < 1).all()
    # utterance 3 of a batch starting at utt 2 == utterance 0 of a batch starting at utt 3
    assert np.array_equal(u[1], philox.uniform_batch(123, 40, 3, 1, 448)[0])


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "d3pm_hip.h")).read()
    declared = set(re.findall(r"\b(d3pm_[a-z_0-9]+)\s*\(", header))
    from vall_e.vall_e import _hip
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    for name in declared:
        assert hasattr(built_lib, name), name
    from vall_e.vall_e import _hip as _h
    assert built_lib.d3pm_abi_version() == _h.ABI_VERSION == 6


def test_product_library_holds_no_experiments_and_no_tuning_state(built_lib):
    """VERDICT round 2 / ADVICE: the timing-only ablation builds (wrong results by construction) and the process-wide
    d3pm_set_tuning are gone from libd3pm_hip.so.  What is left of them lives in libd3pm_hip_ab.so (include/d3pm_hip_ab.h),
    which only tests/ab_*.py load; the product's schedule choices travel in d3pm_tuning, a plain struct the caller owns."""
    import ctypes as C
    from vall_e.vall_e import _hip
    ab_header = open(os.path.join(ROOT, "include", "d3pm_hip_ab.h")).read()
    ab_only = set(re.findall(r"\b(d3pm_[a-z_0-9]+)\s*\(", ab_header)) - {"d3pm_hip"}
    assert ab_only == set(_hip.AB_SIGNATURES), ab_only ^ set(_hip.AB_SIGNATURES)
    for name in ab_only | {"d3pm_set_tuning"}:
        assert not hasattr(built_lib, name), f"{name} is exported by the product library"
    header = open(os.path.join(ROOT, "include", "d3pm_hip.h")).read()
    assert "WRONG" not in header and "d3pm_set_tuning(" not in header.replace("(d3pm_set_tuning)", "")
    # the kernels of the experiments are not even in the code object: no ablation / ring / fused-final instantiation
    lib_bytes = open(_hip.LIB_PATH, "rb").read()
    for marker in (b"gemm_mfma_ring", b"final_sample_fused", b"fill_gelu_table"):
        assert marker not in lib_bytes, marker
    # defaults come from the library and a tuning is a value: two of them are independent
    a, b = _hip.Tuning(), _hip.Tuning()
    built_lib.d3pm_tuning_default(C.byref(a))
    built_lib.d3pm_tuning_default(C.byref(b))
    b.gemm_variant = 5
    assert (a.gemm_variant, a.gemm_persist_slots, a.row_panel, a.workspace_alias, a.attn_cross_resident) == (0, 1024, 10, 1, 1)
    # workspace_alias is read from the shape's tuning, not from a global
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig.libritts()
    sh_a, sh_b = _hip.make_shape(cfg, torch.bfloat16, a), _hip.make_shape(cfg, torch.bfloat16, b)
    b.workspace_alias = 0
    assert built_lib.d3pm_workspace_bytes(C.byref(sh_b), 4) > built_lib.d3pm_workspace_bytes(C.byref(sh_a), 4) > 0


@pytest.mark.parametrize("timesteps", [100, 200])
def test_schedule_build_matches_reference_tables(built_lib, timesteps):
    """d3pm_schedule_build (host C) against the scalars read out of the reference's own dense tables;
    200 steps = SURVEY §8d config 4 (reference subclass with timesteps pinned, see make_golden.py)."""
    from vall_e.vall_e import _hip
    s = _hip.Schedule(timesteps)
    g = load(f"tables_t{timesteps}.npz")
    assert bool(g["structured"])
    for name in ("betas", "d", "c", "dbar", "cbar"):
        assert np.array_equal(getattr(s, name), g[name]), name


def test_schedule_rejects_bad_arguments(built_lib):
    from vall_e.vall_e import _hip
    with pytest.raises(_hip.D3PMError):
        _hip.Schedule(1)
    assert b"bad arguments" in built_lib.d3pm_last_error()


def test_workspace_query_needs_no_gpu(built_lib):
    from vall_e.vall_e import _hip, synth
    sh = _hip.make_shape(synth.D3PMConfig.libritts(), torch.bfloat16)
    n = built_lib.d3pm_workspace_bytes(ctypes.byref(sh), 32)
    assert 150e6 < n < 260e6          # x, h, h2 | att2, att (25 MB each) + the shared qkv / mlp / logits region (100 MB)
    bad = _hip.make_shape(synth.D3PMConfig(d_model=30, n_heads=16), torch.float16)
    assert built_lib.d3pm_workspace_bytes(ctypes.byref(bad), 1) == 0


def test_state_dict_layout_is_the_references():
    from vall_e.vall_e import AR, synth
    m = AR.reference_native()
    spec = synth.state_dict_spec(synth.D3PMConfig.native())
    sd = m.state_dict()
    assert len(sd) == 271 and set(sd) == set(spec)
    assert all(tuple(sd[k].shape) == spec[k] for k in spec)
    assert sum(p.numel() for p in m.parameters()) == 1145473        # SURVEY.md §8b [probe]
    m.load_state_dict(synth.make_state_dict(synth.D3PMConfig.native()), strict=True)


def test_nar_state_dict_layout_is_the_references():
    from vall_e.vall_e import NAR, synth
    cfg = synth.NARConfig(d_model=128, n_heads=2, n_layers=2)
    m = NAR(cfg.n_tokens, cfg.d_model, cfg.n_heads, cfg.n_layers)
    spec = synth.nar_state_dict_spec(cfg)
    assert set(m.state_dict()) == set(spec) and all(tuple(m.state_dict()[k].shape) == spec[k] for k in spec)
    m.load_state_dict(synth.make_nar_state_dict(cfg), strict=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m([torch.tensor([1, 2])], [torch.zeros(3, 8, dtype=torch.long)], [torch.zeros(4, 1, dtype=torch.long)])


def test_registry_surface():
    import vall_e.vall_e as vv
    with pytest.raises(ValueError):
        vv.get_model("something")
    with pytest.raises(NotImplementedError):
        vv.get_model("ar")                    # the stock causal AR model is out of scope
    with pytest.raises(NotImplementedError):
        vv.get_model("nar-bogus")
    from vall_e.vall_e.ar import AR as AR2
    assert AR2 is vv.AR


def test_no_cpu_fallback():
    from vall_e.vall_e import AR
    m = AR.reference_native()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.generate_audio([torch.tensor([1, 2, 3])], [torch.zeros(4, 8, dtype=torch.long)])
    with pytest.raises(RuntimeError, match="no CPU path"):       # the training-side forward has none either
        m([torch.tensor([1, 2, 3])], [torch.zeros(4, 8, dtype=torch.long)], [torch.tensor([5, 6, 7])])
    with pytest.raises(ValueError):
        m([torch.tensor([1])], [torch.zeros(4, 8, dtype=torch.long)])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tts-with-diffusion-model_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace(
                    "oracle/philox.py", ""), f


def test_wrapper_rejects_tensors_the_c_side_would_overrun():
    """The C ABI derives every extent from the shape struct; the Python wrappers must refuse a tensor of another shape,
    dtype or layout instead of handing its pointer over."""
    import torch
    from vall_e.vall_e import _hip
    dev = torch.device("cpu")
    ok = torch.zeros(2, 448, dtype=torch.int32)
    _hip._require(ok, "x_t", (2, 448), (torch.int32,), dev)
    for bad, why in ((torch.zeros(2, 300, dtype=torch.int32), "width"), (torch.zeros(2, 448, dtype=torch.int64), "dtype"),
                     (torch.zeros(448, 2, dtype=torch.int32).t(), "layout"), ([1, 2], "type")):
        with pytest.raises(_hip.D3PMError):
            _hip._require(bad, "x_t", (2, 448), (torch.int32,), dev)


def test_inputs_are_padded_truncated_and_validated_like_upstream():
    """ar_discrete.py:711-735 zero-pads short and truncates over-long phoneme / prompt sequences to the model's fixed key
    counts; an empty or mismatched batch is an error before anything touches the GPU."""
    from vall_e.vall_e import AR
    short, exact, long_ = torch.arange(1, 4), torch.arange(1, 51), torch.arange(1, 80)
    assert AR._pad_rows(short, 50).tolist() == [1, 2, 3] + [0] * 47
    assert torch.equal(AR._pad_rows(exact, 50), exact)
    assert torch.equal(AR._pad_rows(long_, 50), long_[:50])
    prom = torch.ones(7, 8, dtype=torch.long)
    assert AR._pad_rows(prom, 398).shape == (398, 8) and AR._pad_rows(prom, 398)[7:].abs().sum() == 0
    m = AR.reference_native()
    with pytest.raises(ValueError):
        m.generate_audio([], [])
    with pytest.raises(ValueError):
        m.generate_audio([short], [prom, prom])


def test_mx_quantiser_layout_and_error_bound():
    """_hip.quantize_mx / dequantize_mx (the host side of the block-scaled fp8 path): codes [N, K], scales [N, 4, K / 128] with
    scales[n][g][s] <-> elements [128 s + 32 g, +32); the scale is the smallest power of two that keeps the block inside e4m3."""
    from vall_e.vall_e import _hip
    g = torch.Generator().manual_seed(1)
    w = torch.randn(6, 256, generator=g) * torch.exp2(torch.randint(-5, 6, (6, 8), generator=g).float()).repeat_interleave(32, dim=1)
    w[0, 32:64] = 0
    codes, sc = _hip.quantize_mx(w)
    assert codes.shape == (6, 256) and codes.dtype == torch.uint8 and sc.shape == (6, 4, 2) and sc.dtype == torch.uint8
    for n in range(6):
        for blk in range(8):
            amax = w[n, 32 * blk: 32 * blk + 32].abs().max().item()
            byte = int(sc[n, blk % 4, blk // 4])
            if amax == 0:
                assert byte == 1
                continue
            scale = 2.0 ** (byte - 127)
            assert amax / scale <= 448.0 < amax / (scale / 2)
    deq = _hip.dequantize_mx(codes, sc)
    bm = w.abs().reshape(6, 8, 32).amax(dim=-1).repeat_interleave(32, dim=1)
    assert ((deq - w).abs() <= 0.0625 * w.abs() + 2.0 ** -9 * bm).all()
    assert (codes & 0x7F).max().item() <= 0x7E


def test_nar_grid_is_padded_to_whole_gemm_tiles_only_when_cheap():
    """NAR._pack rounds the padded grid up so that batch * t_max is a multiple of 192 (the projections then run as big-tile
    GEMMs), but never by more than 3 %: small batches keep their exact t_max.  Host logic only (the packer reads no device memory)."""
    from vall_e.vall_e import NAR
    m = NAR(1024, 64, 1, 1)                     # parameters on the CPU: the packer's grids stay there too

    def t_max(batch, t, p, r):
        lists = [torch.zeros(t, dtype=torch.int64)] * batch, [torch.zeros(p, 8, dtype=torch.int64)] * batch, [torch.zeros(r, 1, dtype=torch.int64)] * batch
        return m._pack(*lists)[4]

    assert t_max(32, 50, 225, 750) == 1032 and (32 * 1032) % 192 == 0          # 1027 -> 1032: + 0.5 %
    assert t_max(2, 10, 20, 30) == 62                                          # 62 -> 96 would be + 55 %: untouched
    assert t_max(3, 50, 225, 750) == 1027                                      # 1027 -> 1088 would be + 5.9 %: untouched
    try:
        NAR.pad_rows_to_tiles = False
        assert t_max(32, 50, 225, 750) == 1027
    finally:
        NAR.pad_rows_to_tiles = True


def test_kernels_with_asynchronous_asm_loads_do_not_spill():
    """gemm_mfma_big's folded-LayerNorm instantiations (EPI_LNF = 64) request the row moments and the per-column vectors with
    `global_load_dword*` from inline asm and wait for them by hand (counted vmcnt), because hipcc would otherwise drain the DMA
    stream in front of every use.  hipcc believes the destination registers are valid as soon as the asm statement has been
    issued: if it ever SPILLS one of them it stores the stale register and the load later lands in a register that holds
    something else -- an address, say.  A round-4 experiment that pushed those kernels to 256 registers + 28 bytes of scratch
    ended in a memory access fault on the GPU.  So: those instantiations must compile without scratch (cross-compiles, no GPU)."""
    import subprocess
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), "d3pm_mfma_gemm_big.hip"],
                         capture_output=True, text=True, timeout=900).stdout
    rows = [l for l in out.splitlines() if "gemm_mfma_big" in l]
    assert len(rows) >= 40, out[-2000:]
    lnf = [l for l in rows if re.search(r"gemm_mfma_bigID(F16_|F16b)Li(64|65)E", l)]
    assert len(lnf) >= 6, "the LNF / LNF + GELU instantiations of three geometries and two dtypes"
    for l in lnf:
        m = re.search(r"VGPR\s+(\d+).*scratch\s+(\d+)", l)
        assert m and int(m.group(2)) == 0 and int(m.group(1)) <= 256, f"spilling kernel with asynchronous asm loads: {l}"
