"""The N>1 path on CPU: world_size-2 (and 3, uneven) gloo groups around the data-parallel sharding.
The HIP generate call is replaced by a deterministic stand-in keyed by the *global* utterance index,
so the test checks exactly what the DP layer owns: partitioning, utt0 offsets, gather order."""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeModel:
    class cfg:
        canvas = 16
    device = torch.device("cpu")


def _fake_generate(texts, proms, *, seed, utt0):
    rows = []
    for b, t in enumerate(texts):
        g = torch.Generator().manual_seed(seed * 1000 + utt0 + b)
        rows.append(torch.randint(0, 1025, (16,), generator=g) + int(t[0]))
    return torch.stack(rows) if len(rows) > 1 else rows[0]


class _FakeAR:
    class cfg:
        canvas, n_frames = 16, 12
    device = torch.device("cpu")


class _FakeNAR:
    n_resp_levels, n_tokens = 7, 1024


def _fake_nar(texts, proms, resps, *, seed, utt0):
    out = []
    for b, r in enumerate(resps):
        g = torch.Generator().manual_seed(seed * 7919 + utt0 + b)
        rest = torch.randint(0, 1024, (r.shape[0], 7), generator=g)
        out.append(torch.cat([r.long(), rest], dim=-1))
    return out


def _two_stage(dp, n_utts):
    texts = [torch.tensor([i]) for i in range(n_utts)]
    return dp.generate_codes_dp(_FakeAR(), _FakeNAR(), texts, texts, seed=3, ar_fn=_fake_generate, nar_fn=_fake_nar)


def _worker(rank, world, port, n_utts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from vall_e.vall_e import dp
    texts = [torch.tensor([i]) for i in range(n_utts)]
    out = dp.generate_audio_dp(_FakeModel(), texts, texts, seed=3, generate_fn=_fake_generate)
    q.put((rank, out, _two_stage(dp, n_utts)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_utts", [(2, 8), (2, 5), (3, 4)])
def test_dp_gather_equals_single_process(world, n_utts):
    from vall_e.vall_e import dp
    texts = [torch.tensor([i]) for i in range(n_utts)]
    single = dp.generate_audio_dp(_FakeModel(), texts, texts, seed=3, generate_fn=_fake_generate)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_utts, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    outs = {r: a for r, a, _ in got}
    codes = {r: c for r, _, c in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single_codes = _two_stage(dp, n_utts)
    assert single_codes.shape == (n_utts, 12, 8) and single_codes.dtype == torch.int64
    for r in range(world):
        assert torch.equal(outs[r], single), r
        assert torch.equal(codes[r], single_codes), r       # D3PM stage -> NAR stage -> one gather, any rank count


def test_shard_bounds_cover_everything():
    from vall_e.vall_e import dp
    for n in (1, 7, 8, 33, 256):
        for w in (1, 2, 3, 8):
            spans = [dp.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _grad_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from vall_e.vall_e.train import all_reduce_gradients
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 2))
    for i, p in enumerate(model.parameters()):
        if rank == 1 and i == 4:
            continue                                      # a parameter that got no gradient on this rank: counts as zeros
        if i >= 6:
            continue                                      # no gradient on ANY rank (upstream's dead cross_attn2 / token_emb): stays None
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    n = all_reduce_gradients(model, bucket_bytes=80)     # tiny buckets: several collectives, tensors never split
    out = [None if p.grad is None else p.grad.flatten().tolist() for p in model.parameters()]     # plain lists: no shared-memory handles through the queue
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, n, out))


def test_gradient_all_reduce_world2():
    """The data-parallel gradient reduction of the training step (reference: DeepSpeed inside engine.backward,
    utils/engines.py:144-147): bucketed SUM all-reduce / world size, identical collectives on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, g0), (_, n1, g1) = res
    assert n0 == n1 and n0 > 1
    for i, (a, b) in enumerate(zip(g0, g1)):
        assert a == b
        if i >= 6:
            assert a is None, "a parameter without a gradient on every rank must keep .grad = None (as at world size 1)"
            continue
        want = ((1 * (i + 1)) + (0 if i == 4 else 2 * (i + 1))) / 2.0
        assert all(abs(v - want) < 1e-6 for v in a), (i, a[0], want)


def test_bench_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks itself (a child torch.distributed.run,
    launched before the parent touches a GPU) and forward rank 0's JSON line: n_gpus == n_ranks_seen == 2.  The sampler is the
    CPU stand-in of bench.py (--stand-in), everything around it -- launcher, env rendezvous on 127.0.0.1, sharding by rank,
    all-gather, barrier + max-over-ranks timing, the JSON contract -- is the code the 8-GPU run goes through."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "native",
                          "--cpu-steps", "0", "--stand-in", "--batch", "3", "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["config"]["global_batch"] == 6
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["stand_in"] is True
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in out


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """--gpus N on a box with fewer than N GPUs must fail non-zero instead of printing an n_gpus = 1 line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() >= 64:
        pytest.skip("more GPUs than the test asks for")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode != 0 and not any(ln.startswith("{") for ln in res.stdout.splitlines())
