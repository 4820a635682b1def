"""Live pinning: the oracle against the upstream reference imported in this container
(skipped wherever /root/reference does not exist, e.g. on the GPU box)."""
import numpy as np
import pytest
import torch

import ref_harness as rh
from oracle import d3pm_oracle as O
from util import native_setup

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference sources not mounted")


@pytest.fixture(scope="module")
def ref_model():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return rh.build_reference_native()


def test_tables(ref_model):
    betas = O.cosine_betas(100)
    assert torch.equal(betas, ref_model.betas)
    d, c, db, cb = O.scalar_tables(betas, 100)
    for t in (0, 1, 50, 99):
        assert ref_model.q_onestep_mats[t][3, 3].item() == float(d[t])
        assert ref_model.q_onestep_mats[t][3, 512].item() == float(c[t])
        assert ref_model.q_mats[t][7, 7].item() == float(db[t])
        assert ref_model.q_mats[t][7, 512].item() == float(cb[t])
        assert ref_model.q_mats[t][512, 512].item() == 1.0 and ref_model.q_mats[t][512, 7].item() == 0.0


def test_short_loop_bit_identical(ref_model):
    """3 reverse steps, fp16, shared Philox noise: reference generate_audio vs oracle.generate."""
    from make_golden import SharedNoise
    cfg, sd32, texts, proms, orc = native_setup()
    ref_model.float().load_state_dict(sd32)
    m = ref_model.half()
    m.timesteps = 4
    try:
        with rh.cuda_strings_as_cpu(), SharedNoise(99, cfg.canvas, 3):
            y = m.generate_audio(text_list=[texts[0]], proms_list=[proms[0]])
    finally:
        m.timesteps = 100
    assert torch.equal(y, orc.generate(texts[0], proms[0], O.philox_noise(99, cfg.canvas), t_start=3))


def test_upstream_whole_module_pickle_converts_and_loads(ref_model, tmp_path):
    """export.py:14-20 ships `torch.save(model)` with the symmaps as attributes.  tools/convert_upstream_pickle.py (run where
    upstream's classes import, i.e. here) turns it into tensors + dicts; AR.load_exported takes that strictly."""
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("convert_upstream_pickle", os.path.join(root, "tools", "convert_upstream_pickle.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    from vall_e.vall_e import AR
    cfg, sd32, *_ = native_setup()
    m = ref_model.float()
    m.load_state_dict(sd32)
    m.phone_symmap = {"<s>": 2, "</s>": 1, "AH0": 3}
    m.spkr_symmap = {"p225": 0, "p226": 1}
    tables = {k: m.__dict__.pop(k) for k in [k for k, v in list(m.__dict__.items()) if isinstance(v, (list, torch.Tensor)) and k.startswith("q_")]}
    try:                                                       # (the 630 MB of dense tables are plain attributes: keep the test file small)
        torch.save(m, tmp_path / "ar.pt")                       # the reference's export format: a whole-module pickle
    finally:
        m.__dict__.update(tables)
    conv.main([str(tmp_path / "ar.pt"), str(tmp_path / "ar_export.pt")])
    mine = AR.load_exported(tmp_path / "ar_export.pt")
    assert mine.phone_symmap == m.phone_symmap and mine.spkr_symmap == m.spkr_symmap
    ref_sd = m.state_dict()
    got = mine.state_dict()
    assert set(got) - {"_symmaps"} == set(ref_sd)
    assert all(torch.equal(got[k], ref_sd[k]) for k in ref_sd)


def test_formats_against_the_references_loaders_live(tmp_path):
    """formats.py next to data.py:_load_quants / _get_phones executed from the reference's source on the same files."""
    from vall_e import formats
    data = rh.load_reference_data_module()
    p = tmp_path / "spk" / "u.wav"
    p.parent.mkdir()
    codes = torch.randint(0, 1024, (1, 8, 23), dtype=torch.int64)
    torch.save(codes, data._replace_file_extension(p, ".qnt.pt"))
    data._replace_file_extension(p, ".phn.txt").write_text("HH AH0 L OW1 _ W ER1 L D", encoding="utf8")
    assert torch.equal(formats.load_quants(tmp_path / "spk" / "u.qnt.pt"), data._load_quants(p))
    assert formats.read_phones(tmp_path / "spk" / "u.phn.txt") == data._get_phones(p)
    ds = data.VALLEDatset([p])
    assert formats.build_symmap([formats.read_phones(tmp_path / "spk" / "u.phn.txt")]) == ds.phone_symmap
