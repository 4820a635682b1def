"""Stock NAR levels 1..7 through d3pm_nar_level against the reference fixture and the oracle."""
import numpy as np
import pytest
import torch

from oracle import nar_oracle as N
from util import f16, load

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(dtype, cfg=None):
    from vall_e.vall_e import NAR, synth
    cfg = cfg or synth.NARConfig(d_model=128, n_heads=2, n_layers=2)
    sd32 = synth.make_nar_state_dict(cfg, 0)
    m = NAR(cfg.n_tokens, cfg.d_model, cfg.n_heads, cfg.n_layers)
    m.load_state_dict(sd32)
    return cfg, sd32, m.to(dtype).to(DEV)


@pytest.mark.parametrize("tag,dtype,tol", [("f32", torch.float32, 1e-3), ("f16", torch.float16, 1.5e-2)])
def test_nar_logits_match_reference(built_lib, tag, dtype, tol):
    from vall_e.vall_e import synth
    g = load("nar_small.npz")
    cfg, sd32, m = build(dtype)
    conv = (lambda a: torch.from_numpy(a)) if dtype == torch.float32 else f16
    for lvl, n_lv in ((0, 1), (3, 4)):
        texts, proms, resps = synth.make_nar_inputs(2, 1, n_levels=n_lv)
        _, logits, lens, t_max = m(texts, proms, resps, return_logits_level=lvl, greedy=True)
        for b in range(2):
            tt, tp, tr = (int(v) for v in lens[b])
            rows = logits[b, tt + tp + 2: tt + tp + 2 + tr].cpu().float()
            ref = conv(g[f"logits_l{lvl}_utt{b}_{tag}"]).float()
            err = (torch.cat([rows[:8], rows[-8:]]) - ref).abs().max().item()
            assert err < tol, (tag, lvl, b, err)


def test_nar_sampling_matches_oracle_on_the_shared_stream(built_lib):
    """fp32: the seven levels sampled by the HIP path equal the oracle's Gumbel-max over the same Philox stream
    (a rare near-tie may flip an id: >= 99 % of the ids must agree); greedy likewise; ragged batch of two."""
    from vall_e.vall_e import synth
    cfg, sd32, m = build(torch.float32)
    texts, proms, resps = synth.make_nar_inputs(2, 1)
    for mode in ("gumbel", "greedy"):
        out = m(texts, proms, resps, sampling_temperature=0.2, seed=11, greedy=(mode == "greedy"))
        ref = N.generate(sd32, cfg.n_heads, cfg.n_layers, texts, proms, resps, 0.2, sampler=mode, seed=11)
        for b in range(2):
            assert out[b].shape == (len(resps[b]), 8) and out[b].dtype == torch.int64
            assert torch.equal(out[b][:, 0].cpu(), resps[b][:, 0])
            agree = (out[b].cpu() == ref[b]).float().mean().item()
            assert agree > 0.99, (mode, b, agree)
    again = m(texts, proms, resps, sampling_temperature=0.2, seed=11)
    other = m(texts, proms, resps, sampling_temperature=0.2, seed=12)
    first = m(texts, proms, resps, sampling_temperature=0.2, seed=11)
    assert all(torch.equal(a, b) for a, b in zip(again, first)) and not all(torch.equal(a, b) for a, b in zip(again, other))


def test_nar_mfma_family_and_batch_independence(built_lib):
    """d=512 / 8 heads (the `-half` size, 2 layers): MFMA GEMM + masked MFMA attention vs the generic family, and a
    ragged batch vs the utterances run one by one (padding rows and key masks must not leak)."""
    from vall_e.vall_e import _hip, synth
    cfg, sd32, m = build(torch.bfloat16, synth.NARConfig(d_model=512, n_heads=8, n_layers=2))
    texts, proms, resps = synth.make_nar_inputs(3, 2, t_text=(20, 50), t_prom=(100, 225), t_resp=(300, 750))
    _, lg, lens, t_max = m(texts, proms, resps, return_logits_level=0, greedy=True)
    run = m.runner(t_max)
    lens_d, text, prom, resp, _, _ = m._pack(texts, proms, resps)
    lg_gen = run.level(lens_d, text, prom, resp, t_max, 0, 0.2, 0, flags=_hip.FLAG_FORCE_GENERIC | _hip.FLAG_GREEDY,
                       want_logits=True)
    for b in range(3):
        tt, tp, tr = (int(v) for v in lens[b])
        a = lg[b, tt + tp + 2: tt + tp + 2 + tr].float()
        c = lg_gen[b, tt + tp + 2: tt + tp + 2 + tr].float()
        assert (a - c).abs().max().item() < 0.08          # measured 0.017 (one bf16 quantum at |logit| ~ 4); was 0.25
    batch = m(texts, proms, resps, seed=3)
    for b in range(3):
        one = m([texts[b]], [proms[b]], [resps[b]], seed=3, utt0=b)
        assert (one[0] == batch[b]).float().mean().item() > 0.98


def test_cli_two_stages_write_a_qnt_file(built_lib, tmp_path):
    """`python -m vall_e` with pre-tokenised inputs (formats.py): D3PM stage -> NAR stage -> `.qnt.pt` [1, 8, t] in the
    layout EnCodec's decoder reads (/root/reference/vall_e/__main__.py:44-73, emb/qnt.py:68,93)."""
    import json
    from vall_e import __main__ as cli, formats
    (tmp_path / "u.phn.txt").write_text("HH AH0 L OW1 _ W ER1 L D", encoding="utf8")
    symmap = formats.build_symmap([formats.read_phones(tmp_path / "u.phn.txt")])
    (tmp_path / "symmap.json").write_text(json.dumps(symmap), encoding="utf8")
    torch.save(torch.randint(0, 1024, (1, 8, 120), dtype=torch.int64), tmp_path / "prompt.qnt.pt")
    argv = [str(tmp_path / "out.qnt.pt"), "--phn-file", str(tmp_path / "u.phn.txt"), "--symmap", str(tmp_path / "symmap.json"),
            "--prompt-qnt", str(tmp_path / "prompt.qnt.pt"), "--native", "--seed", "5", "--nar-model", "nar-quarter"]
    torch.manual_seed(0)
    cli.main(argv)                                                   # level 0 only: no --nar-ckpt
    lvl0 = torch.load(tmp_path / "out.qnt.pt")
    assert lvl0.shape == (1, 1, 350) and lvl0.dtype == torch.int64 and 0 <= lvl0.min() and lvl0.max() <= 1024
    from vall_e.vall_e import get_model
    torch.manual_seed(1)
    torch.save(get_model("nar-quarter").state_dict(), tmp_path / "nar.pt")
    torch.manual_seed(0)
    cli.main(argv + ["--nar-ckpt", str(tmp_path / "nar.pt")])
    full = torch.load(tmp_path / "out.qnt.pt")
    assert full.shape == (1, 8, 350) and full.dtype == torch.int64
    assert torch.equal(full[:, :1], lvl0)                             # the NAR stage keeps level 0
    assert 0 <= full[:, 1:].min() and full[:, 1:].max() < 1024


def test_two_stage_dp_helper_on_one_gpu(built_lib):
    """dp.generate_codes_dp with the real models (world size 1): equals running the two stages by hand, and sub-batches
    started at their global utterance index reproduce the rows of the full batch (what the sharding relies on)."""
    from vall_e.vall_e import AR, NAR, dp, synth
    cfg = synth.D3PMConfig.native()
    ar = AR.from_config(cfg)
    ar.load_state_dict(synth.make_state_dict(cfg, 0))
    ar = ar.half().to(DEV)
    ncfg = synth.NARConfig(d_model=256, n_heads=4, n_layers=2)
    nar = NAR(ncfg.n_tokens, ncfg.d_model, ncfg.n_heads, ncfg.n_layers)
    nar.load_state_dict(synth.make_nar_state_dict(ncfg, 0))
    nar = nar.half().to(DEV)
    texts, proms = synth.make_inputs(cfg, 3, 1)
    codes = dp.generate_codes_dp(ar, nar, texts, proms, seed=5, steps=8)
    assert codes.shape == (3, cfg.n_frames, 8) and codes.dtype == torch.int64
    lvl0 = ar.generate_audio(texts, proms, seed=5, steps=8)[:, : cfg.n_frames]
    assert torch.equal(codes[..., 0], lvl0.clamp(max=1023))
    tail = dp.generate_codes_dp(ar, nar, texts[1:], proms[1:], seed=5, steps=8, ar_fn=lambda t, p, **k: ar.generate_audio(t, p, **{**k, "utt0": k["utt0"] + 1}),
                                nar_fn=lambda t, p, r, **k: nar(t, p, r, **{**k, "utt0": k["utt0"] + 1}))
    assert (tail == codes[1:]).float().mean().item() > 0.98


def test_nar_padding_and_batch_do_not_change_an_utterance(built_lib):
    """NAR.pad_rows_to_tiles pads the [B, t_max] grid so that B * t_max is a multiple of 192 (big-tile GEMMs); the padded
    row count depends on the batch size, so an utterance's rows sit in different tiles in different batches.  Every GEMM
    schedule accumulates in the same order and rows / keys past an utterance's own length are masked, so neither the padding
    nor the batch may change a logit or an id: padded vs unpadded grid, and one utterance alone vs inside a ragged batch of
    32 (what dp.generate_codes_dp's world-size independence rests on for the NAR stage), bit for bit."""
    from vall_e.vall_e import synth
    cfg, sd32, m = build(torch.bfloat16, synth.NARConfig(d_model=512, n_heads=8, n_layers=2))
    texts, proms, resps = synth.make_nar_inputs(32, 5, t_text=(20, 50), t_prom=(100, 225), t_resp=(300, 750))
    outs = {}
    for pad in (True, False):
        m.pad_rows_to_tiles = pad
        ids, lg, lens, t_max = m(texts, proms, resps, seed=3, return_logits_level=3)
        outs[pad] = (ids, lg, lens, t_max)
    assert outs[True][3] != outs[False][3], "the padding did not change the grid: nothing was tested"
    for b in range(32):
        tt, tp, tr = (int(v) for v in outs[True][2][b])
        rows = slice(tt + tp + 2, tt + tp + 2 + tr)
        assert torch.equal(outs[True][1][b, rows], outs[False][1][b, rows]), f"utterance {b}: logits differ between the padded and the plain grid"
        assert torch.equal(outs[True][0][b], outs[False][0][b]), f"utterance {b}: ids differ between the padded and the plain grid"
    m.pad_rows_to_tiles = True
    for b in (0, 13, 31):
        one = m([texts[b]], [proms[b]], [resps[b]], seed=3, utt0=b)
        assert torch.equal(one[0], outs[True][0][b]), f"utterance {b}: {(one[0] != outs[True][0][b]).sum().item()} ids differ alone vs inside the batch of 32"
