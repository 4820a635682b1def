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
