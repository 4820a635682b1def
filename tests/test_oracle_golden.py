"""The CPU oracle against the fixtures the *reference itself* produced (tests/golden/make_golden.py).

Integer / table / sampler outputs must be bit-exact everywhere.  Bit-exact equality of transformer
activations additionally needs the CPU the fixtures were generated on (BLAS accumulation order);
elsewhere a 2-ulp fp16 / 1e-5 fp32 tolerance applies.
"""
import numpy as np
import pytest
import torch

from conftest import same_platform_as_golden
from oracle import d3pm_oracle as O
from oracle import philox
from util import bits, f16, load, native_setup, ulp16_diff


@pytest.mark.parametrize("timesteps", [100, 200])
def test_schedule_scalars_match_reference_tables(timesteps):
    g = load(f"tables_t{timesteps}.npz")
    assert bool(g["structured"])
    betas = O.cosine_betas(timesteps)
    assert np.array_equal(bits(betas), g["betas"])
    for mine, name in zip(O.scalar_tables(betas, timesteps), ("d", "c", "dbar", "cbar")):
        assert np.array_equal(mine.view(np.uint16), g[name]), name
    if timesteps == 100:
        assert g["eps16"][0] == bits(torch.tensor([1e-6], dtype=torch.float16))[0]


def test_dense_tables_have_closed_form_structure():
    """Independent of the fixture: rebuild the reference-style dense tables for a short schedule and
    check d*I + c*1e_M^T with row M = e_M against the scalar recurrence."""
    T, K, M = 12, 65, 32
    betas = O.cosine_betas(T)
    one, qbar, one_t = O.dense_tables(betas, T, K, M)
    d, c, db, cb = O.scalar_tables(betas, T)
    for t in range(T):
        for tab, dd, cc in ((one[t], d[t], c[t]), (qbar[t], db[t], cb[t])):
            exp = torch.zeros(K, K, dtype=torch.float16)
            exp.fill_diagonal_(float(dd))
            exp[:, M] = float(cc)
            exp[M, :] = 0
            exp[M, M] = 1
            assert torch.equal(tab, exp), t
        assert torch.equal(one_t[t], one[t].T)


def test_posterior_closed_form_equals_dense_matmul():
    torch.manual_seed(3)
    T = 100
    orc_tabs = O.scalar_tables(O.cosine_betas(T), T)
    dense = O.dense_tables(O.cosine_betas(T), T)
    logits = (torch.randn(64, 1025) * 1.5).half()
    x_t = torch.randint(0, 1025, (64,))
    x_t[::2] = 512
    for t in (1, 2, 40, 99):
        a = O.posterior_logits_dense(logits, x_t, t, dense[2], dense[1])
        b = O.posterior_logits_closed(logits, x_t, t, orc_tabs)
        assert torch.equal(a, b), t
    assert torch.equal(O.posterior_logits_closed(logits, x_t, 0, orc_tabs), logits)


def test_sampler_on_reference_logits():
    g = load("native_step.npz")
    cfg, _, _, _, orc = native_setup()
    logits = f16(g["logits_full_f16"])
    x_t = torch.from_numpy(g["x_t"].astype(np.int64))
    t = int(g["t"])
    post = orc.posterior(logits, x_t, t)
    assert np.array_equal(bits(post[g["rows"]]), g["posterior_rows_f16"])
    u = torch.from_numpy(philox.uniform_batch(123, t, 0, 1, cfg.canvas)[0])
    assert np.array_equal(O.gumbel_argmax(post, u, t).numpy(), g["x_next_seed123"].astype(np.int64))
    uq = torch.from_numpy(philox.uniform_batch(123, t, 0, 1, cfg.canvas, stream=philox.STREAM_Q_SAMPLE)[0])
    mask = torch.zeros(cfg.canvas, dtype=torch.bool)
    mask[: cfg.n_frames] = True
    assert np.array_equal(O.q_sample(x_t, t, orc.tabs, uq, mask).numpy(), g["q_sample_seed123"].astype(np.int64))


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f16", torch.float16)])
def test_denoiser_tensors(tag, dtype):
    g = load("native_step.npz")
    cfg, _, texts, proms, orc = native_setup(dtype)
    exact = same_platform_as_golden()
    conv = (lambda a: torch.from_numpy(a)) if dtype == torch.float32 else f16

    def close(mine, ref, what):
        ref = conv(ref)
        if exact:
            assert torch.equal(mine, ref), what
        elif dtype == torch.float32:
            assert (mine - ref).abs().max() < 1e-4, what
        else:
            assert ulp16_diff(mine, ref).max() <= 4, what

    with torch.no_grad():
        cp, ct = orc.conditions(texts[0], proms[0])
        close(cp, g[f"cond_prompt_{tag}"], "cond_prompt")
        close(ct, g[f"cond_text_{tag}"], "cond_text")
        x_t = torch.from_numpy(g["x_t"].astype(np.int64))
        mask = torch.zeros(cfg.canvas, dtype=torch.bool)
        mask[: cfg.n_frames] = True
        t = int(g["t"])
        x0 = conv(g[f"block0_in_{tag}"])
        temb = orc.sd["time_emb.weight"][t][None]
        y0 = O.dit_block(orc.sd, 0, x0[None], cp[None], ct[None], temb, mask, orc.shape)[0]
        close(y0, g[f"block0_out_{tag}"], "block0_out")
        close(O.denoiser_hidden(orc.sd, orc.shape, x_t, t, cp, ct, mask)[0], g[f"hidden_{tag}"], "hidden")
        close(orc.logits(x_t, t, cp, ct, mask)[g["rows"]], g[f"logits_rows_{tag}"], "logits")


def test_plumbing_10_steps_and_greedy():
    """BASELINE.json configs[0] (10 diffusion steps, CPU) and the degenerate greedy mode."""
    g = load("native_loop.npz")
    cfg, sd32, texts, proms, orc = native_setup()
    y = orc.generate(texts[0], proms[0], O.philox_noise(123, cfg.canvas), t_start=10)
    if same_platform_as_golden():
        assert np.array_equal(y.numpy(), g["plumbing10_utt0_seed123"].astype(np.int64))
    else:
        assert (y.numpy() == g["plumbing10_utt0_seed123"]).mean() > 0.98
    # greedy never unmasks with these weights: every live frame stays 512, pad rows stay 0
    yg = orc.generate(texts[0], proms[0], None, greedy=True, t_start=5)
    assert (yg[: cfg.n_frames] == 512).all() and np.array_equal(g["greedy"][: cfg.n_frames], np.full(cfg.n_frames, 512))


@pytest.mark.skipif(not same_platform_as_golden(), reason="full trajectories are only bit-stable on the fixture CPU")
def test_full_loop_trajectory_utt1():
    g = load("native_loop.npz")
    cfg, sd32, texts, proms, orc = native_setup()
    tr = []
    orc.generate(texts[1], proms[1], O.philox_noise(7, cfg.canvas), trace=tr)
    assert np.array_equal(torch.stack(tr).numpy(), g["traj_utt1_seed7"].astype(np.int64))


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f16", torch.float16)])
def test_nar_oracle_matches_reference_fixture(tag, dtype):
    """Stock NAR (levels 1..7): response-row logits at levels 0 and 3 and the full generation under torch seed 0
    against what the reference produced (tests/golden/nar_small.npz)."""
    from oracle import nar_oracle as N
    from vall_e.vall_e import synth
    g = load("nar_small.npz")
    cfg = synth.NARConfig(d_model=128, n_heads=2, n_layers=2)
    sd = {k: v.to(dtype) for k, v in synth.make_nar_state_dict(cfg, 0).items()}
    conv = (lambda a: torch.from_numpy(a)) if dtype == torch.float32 else f16
    exact = same_platform_as_golden()
    for lvl, n_lv in ((0, 1), (3, 4)):
        texts, proms, resps = synth.make_nar_inputs(2, 1, n_levels=n_lv)
        with torch.no_grad():
            lg = N.level_logits(sd, cfg.n_heads, cfg.n_layers, texts, proms, resps, lvl)
        for b in range(2):
            mine, ref = torch.cat([lg[b][:8], lg[b][-8:]]), conv(g[f"logits_l{lvl}_utt{b}_{tag}"])
            if exact:
                assert torch.equal(mine, ref), (lvl, b)
            else:
                assert (mine.float() - ref.float()).abs().max() < (1e-4 if dtype == torch.float32 else 2e-2)
    if exact:
        texts, proms, resps = synth.make_nar_inputs(2, 1)
        torch.manual_seed(0)
        y = N.generate(sd, cfg.n_heads, cfg.n_layers, texts, proms, resps, 0.2, sampler="torch")
        for b in range(2):
            assert np.array_equal(y[b].numpy(), g[f"generated_seed0_utt{b}_{tag}"].astype(np.int64))


def test_gumbel_max_is_the_categorical_distribution():
    """The build samples NAR levels with a Philox Gumbel-max instead of torch's multinomial: same distribution."""
    from oracle import nar_oracle as N
    logits = torch.tensor([[2.0, 0.5, -1.0, 0.0, 1.0]]).repeat(20000, 1)
    ids = N.sample_gumbel([logits], 1.0, seed=5, level=0)[0]
    freq = torch.bincount(ids, minlength=5).float() / len(ids)
    assert (freq - torch.softmax(logits[0], -1)).abs().max() < 0.012


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f16", torch.float16)])
def test_training_forward_loss(tag, dtype):
    """SURVEY §8f row 3, forward half: the reference's own AR.forward (ar_discrete.py:588-694) ran on these inputs
    with its q_sample draws on Philox stream 1; make_golden.py asserted bit-equality of loss and last logits with
    this oracle function at generation time."""
    g = load("native_forward.npz")
    cfg, _, texts, proms, orc = native_setup(dtype)
    resps = torch.from_numpy(g["resps"].astype(np.int64))
    seed = int(g["seed"])

    def q_noise(t):
        return torch.from_numpy(philox.uniform_batch(seed, t, 0, 1, cfg.canvas, stream=philox.STREAM_Q_SAMPLE))[0]

    with torch.no_grad():
        loss, x = O.training_forward(orc.sd, orc.shape, texts[0], proms[0], resps, q_noise)
    ref_loss = float(g[f"loss_{tag}"])
    ref_rows = torch.from_numpy(g[f"last_logits_rows_{tag}"]) if dtype == torch.float32 else f16(g[f"last_logits_rows_{tag}"])
    if same_platform_as_golden():
        assert float(loss) == ref_loss and torch.equal(x[g_rows()], ref_rows)
    else:
        assert abs(float(loss) - ref_loss) < (1e-4 if dtype == torch.float32 else 5e-3)
    assert 1.0 < ref_loss < 7.0                      # ~ log(1025) * (1 + masked fraction) / live frames ... order of magnitude
    assert x.shape == (cfg.canvas, 1025) and bool((x[300:] == 0).all())   # logits of padded frames are zeroed (:683)


def g_rows():
    return load("native_step.npz")["rows"]


def test_training_gradients_of_the_oracle_match_the_references_autograd():
    """tests/golden/native_grads.npz: per-parameter gradient summaries from `loss.backward()` on the reference's own forward
    (make_golden.py:gen_grads, three timesteps, fp32).  torch.autograd over the oracle must reproduce them -- this is the
    checker the HIP backward kernels are compared with on the GPU (tests/test_gpu_train.py)."""
    g = load("native_grads.npz")
    cfg, sd32, texts, proms, _ = native_setup(torch.float32)
    resps = torch.from_numpy(load("native_forward.npz")["resps"].astype(np.int64))
    seed, T = int(g["seed"]), int(g["timesteps"])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd32.items()}

    def q_noise(t):
        return torch.from_numpy(philox.uniform_batch(seed, t, 0, 1, cfg.canvas, stream=philox.STREAM_Q_SAMPLE))[0]

    loss, _ = O.training_forward(sd, O.Shape.of(cfg), texts[0], proms[0], resps, q_noise, timesteps=T)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    names = [k[2:] for k in g.files if k.startswith("g/")]
    assert len(names) >= 230
    for name in names:
        grad = sd[name].grad.clone()
        if name in ("text_emb.weight", "resps_emb.weight"):
            grad[0] = 0                                   # nn.Embedding(padding_idx=0) upstream
        flat = grad.reshape(-1).double()
        idx = torch.linspace(0, flat.numel() - 1, 8).long()
        want = g["g/" + name]
        scale = max(np.abs(want[2:]).max(), want[1] / flat.numel(), 1e-12)
        assert np.abs(flat[idx].numpy() - want[2:]).max() <= 1e-4 * scale + 1e-10, (name, flat[idx].numpy(), want[2:])
        assert abs(flat.abs().sum().item() - want[1]) <= 1e-4 * want[1] + 1e-12, (name, flat.abs().sum().item(), want[1])


def test_n_q_extension_oracle_reduces_to_the_pinned_oracle_at_n_q_1():
    """oracle/d3pm_nq_oracle.py (the build's definition of the n_q > 1 extension) with a level-0-only state dict: the same
    logits and the same sampled ids as oracle/d3pm_oracle.py, bit for bit."""
    from oracle import d3pm_nq_oracle as NQ
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig.native()
    sd = {k: v.half() for k, v in synth.make_state_dict(cfg, 0).items()}
    texts, proms = synth.make_inputs(cfg, 1, 1)
    shape = O.Shape.of(cfg)
    orc = O.Oracle(sd, shape)
    a = orc.generate(texts[0], proms[0], O.philox_noise(9, cfg.canvas), t_start=3)
    b = NQ.generate(sd, shape, texts[0], proms[0], 9, t_start=3)
    assert b.shape == (cfg.canvas, 1) and torch.equal(b[:, 0], a)
    x, mask = orc.canvas_init()
    with torch.no_grad():
        cp, ct = orc.conditions(texts[0], proms[0])
        assert torch.equal(NQ.logits(sd, shape, x[:, None], 50, cp, ct, mask)[:, 0], orc.logits(x, 50, cp, ct, mask))



def test_wide_trajectory_fixture_is_the_oracles_and_ends_on_the_references_ids():
    """tests/golden/wide_f16_trajectory.npz (make_wide_trajectory.py) is what the GPU suite teacher-forces the HIP path along: it
    must be the oracle's current d = 512 fp16 loop row by row, and end on the ids the reference produced (wide_step.npz)."""
    from vall_e.vall_e import synth
    cfg = synth.D3PMConfig(d_model=512, n_heads=8, n_layers=6)
    sd32 = synth.make_state_dict(cfg, 0)
    texts, proms = synth.make_inputs(cfg, 1, 1)
    fx = load("wide_f16_trajectory.npz")
    ref_final = load("wide_step.npz")["loop_seed123"].astype(np.int64)
    assert np.array_equal(fx["x_0"].astype(np.int64), ref_final)
    orc = O.Oracle({k: v.half() for k, v in sd32.items()}, O.Shape.of(cfg))
    trace = []
    with torch.no_grad():
        end = orc.generate(texts[0], proms[0], O.philox_noise(123, cfg.canvas), trace=trace)
    assert np.array_equal(end.numpy(), ref_final)
    assert len(trace) == fx["x"].shape[0] == 99
    for i, row in enumerate(trace):
        assert np.array_equal(row.numpy(), fx["x"][i].astype(np.int64)), f"row {i} of the fixture is not the oracle's"
