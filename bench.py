"""bench.py -- EnCodec codec-tokens/sec of the 100-step D3PM sampler on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no torchrun environment (no WORLD_SIZE) the script starts the N ranks itself: a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` of this same command line, launched
before this process makes any GPU call; rank 0's JSON line and the child's exit code pass through.

One "step" = one full reverse process (condition encoders + 99 denoise/sample iterations) over this
rank's batch of utterances, inputs resident in HBM.  Workload at every N: BASELINE.json configs[1]
("LibriTTS"): d=512, 8 heads, 6 blocks, 750 live frames on a 768 canvas,
50 phoneme + 225 prompt keys, 32 utterances per GPU (weak scaling), synthetic random-init weights.
Tokens counted are the live level-0 codec frames the D3PM produces (n_q = 1, as upstream).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "tts-with-diffusion-model_amd"), ROOT]

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}    # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="libritts", choices=["libritts", "native", "vctk"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--cpu-steps", type=int, default=4, help="diffusion iterations timed for cpu_baseline (0 = skip)")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 path with several ranks sharing one GPU (NCCL refuses that)")
    ap.add_argument("--no-nar", action="store_true", help="skip the extra NAR (levels 1..7) measurement")
    ap.add_argument("--no-nq8", action="store_true", help="skip the extra n_q = 8 extension measurement")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra fp8 fast-path measurement (BASELINE.json configs[4])")
    ap.add_argument("--no-vctk", action="store_true", help="skip the extra VCTK long-prompt measurement (BASELINE.json configs[3])")
    ap.add_argument("--streams", type=int, default=1, help="independent batch chunks on separate HIP streams")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket GEMM launches with HIP events (roofline.achieved is then 0): measures what the "
                         "event pairs cost the timed region")
    ap.add_argument("--tune", default="", help="A/B ONLY: comma-separated field=value pairs of d3pm_tuning "
                    "(include/d3pm_hip.h), e.g. row_panel=0; the JSON line records them and is not the headline configuration")
    ap.add_argument("--cpu-only", action="store_true",
                    help="time only the CPU port (no GPU needed) and print its JSON: the container calibration under profiles/")
    ap.add_argument("--stand-in", action="store_true",
                    help="REHEARSAL ONLY (no GPU needed): the launcher / sharding / all-gather / timing path of this script around a "
                         "deterministic CPU stand-in for the sampler, with --backend gloo; the JSON line is marked invalid_for_headline")
    ap.add_argument("--profile-iters", type=int, default=0,
                    help="PROFILING ONLY: run this many diffusion iterations instead of all 99 (the JSON line is then "
                         "marked invalid_for_headline)")
    return ap.parse_args()


def algorithmic_flops_per_step(cfg, batch):
    """F_tok = L(32 d^2 + 4 T d + 4 d (S_t+S_p)) + 2 d K per canvas row and iteration."""
    d, T = cfg.d_model, cfg.canvas
    f_tok = cfg.n_layers * (32 * d * d + 4 * T * d + 4 * d * (cfg.s_text + cfg.s_prompt)) + 2 * d * cfg.n_classes
    return float(f_tok) * batch * T * (cfg.timesteps - 1)


def note(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cores() -> int:
    """Threads this process may really use: scheduler affinity capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def build_id() -> str:
    """Identity of the kernel sources this process runs (sha1 over csrc/ and the C header): rocprofv3 PMC summaries
    under profiles/ carry the id of the tree they were collected on, so a traffic figure is only ever reported beside
    timings of the same build."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "tts-with-diffusion-model_amd", "csrc", "*")) +
                   [os.path.join(ROOT, "include", "d3pm_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(kernel_class="gemm"):
    """HBM-side bytes per launch of a kernel class from the committed rocprofv3 PMC summary of THIS build
    (profiles/*pmc_traffic.json, written by profiles/summarize_pmc.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE
    passes over this same script: counters cannot be collected from inside the timed run).  A summary of another build
    (its `_build_id` differs from build_id()) is not reported: returns (None, reason)."""
    import glob
    me = build_id()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        if data.get("_build_id") == me and kernel_class in data:
            return data[kernel_class]["traffic_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, f"no profiles/*pmc_traffic.json was collected on build {me}"


def nar_stage(dev, dtype, batch, cfg, level0):
    """The step right after the D3PM path (reference __main__.py:36-38): the stock NAR model (registry size
    d=1024 / 16 heads / 12 layers) fills quantizer levels 1..7 for the same batch.  Reported beside, never inside,
    `value` (the BASELINE.json metric counts the D3PM stage).  Inputs are resident in HBM like the D3PM stage's output is
    in the real pipeline; the stage is timed twice over the same repetitions -- host wall clock around a synchronise and
    HIP events on the stream -- so that a host-bound run shows up as a gap between the two."""
    from vall_e.vall_e import NAR, synth
    ncfg = synth.NARConfig()
    nar = NAR(ncfg.n_tokens, ncfg.d_model, ncfg.n_heads, ncfg.n_layers)
    nar.load_state_dict(synth.make_nar_state_dict(ncfg, 0))
    nar = nar.to(dtype).to(dev)
    texts, proms = synth.make_inputs(cfg, batch, 1)
    texts = [t.to(dev) for t in texts]
    proms = [p.to(dev) for p in proms]
    resps = [level0[b, : cfg.n_frames].clamp(max=ncfg.n_tokens - 1).reshape(-1, 1) for b in range(batch)]   # on the device
    for i in range(2):
        nar(texts, proms, resps, seed=i)                   # warm-up
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(reps):
        full = nar(texts, proms, resps, seed=2 + i)
    e1.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    dt_ev = e0.elapsed_time(e1) * 1e-3 / reps
    assert full[0].shape == (cfg.n_frames, 8)
    rows = sum(len(t) + len(p) + len(r) + 2 for t, p, r in zip(texts, proms, resps))
    flops = 7.0 * rows * ncfg.n_layers * 24 * ncfg.d_model ** 2       # GEMM flops only (QKV, out, FFN), 7 levels
    return {"model": f"NAR d={ncfg.d_model} H={ncfg.n_heads} L={ncfg.n_layers}", "seconds_per_batch": dt,
            "seconds_per_batch_hip_events": dt_ev, "repetitions": reps,
            "codec_tokens_per_s": 7 * batch * cfg.n_frames / dt, "gemm_tflops": flops / dt / 1e12}


def fp8_fast_path(dev, dtype, batch, cfg, sd32, texts, proms):
    """BASELINE.json configs[4] on one GPU: e4m3 operands for the QKV / cross-query / fc1 projections and the 50-step
    schedule (49 iterations).  Reported beside `value`, never inside it (the headline is the bf16 100-step config);
    the reference has no such mode, so the quality number is id agreement with the 16-bit path on the same schedule."""
    import dataclasses
    from vall_e.vall_e import AR
    fast = dataclasses.replace(cfg, timesteps=50)
    m = AR.from_config(fast)
    m.load_state_dict({k: v for k, v in sd32.items() if k != "time_emb.weight"}, strict=False)
    m = m.to(dtype).to(dev)
    out = {}
    ids = {}
    for name, fp8 in (("bf16_50_steps", False), ("fp8_50_steps", True)):
        m.generate_audio(texts, proms, seed=1, fp8=fp8)
        torch.cuda.synchronize()
        reps = 6
        t0 = time.perf_counter()
        for i in range(reps):
            ids[name] = m.generate_audio(texts, proms, seed=7, fp8=fp8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"seconds_per_batch": dt, "codec_tokens_per_s": batch * cfg.n_frames / dt, "repetitions": reps}
    live = slice(0, cfg.n_frames)
    a, b = (ids[k].reshape(batch, -1)[:, live] for k in ("bf16_50_steps", "fp8_50_steps"))     # one utterance comes back 1-D
    out["id_agreement_fp8_vs_bf16"] = float((a == b).float().mean())
    out["speedup_fp8_vs_bf16_same_schedule"] = out["fp8_50_steps"]["codec_tokens_per_s"] / out["bf16_50_steps"]["codec_tokens_per_s"]
    out["note"] = ("49 iterations (timesteps = 50), same synthetic weights except time_emb (random init for the shorter "
                   "schedule); fp8 = block-scaled e4m3 (v_mfma_scale_f32_16x16x128_f8f6f4, one power-of-two scale per 32 elements) "
                   "for norm1->QKV, norm2|22->cross q, norm3->fc1; the LayerNorm rows leave the row-panel out-projections in that format; "
                   "free-running id agreement compounds single near-tie flips over 49 iterations (teacher-forced agreement and "
                   "logits error: tests/test_gpu_parity.py::test_fp8_fast_path_agreement_with_the_16_bit_path)")
    return out


def class_table(per_class, key):
    """{class: launches, us per launch, TFLOP/s | GB/s, fraction of the roof that bounds it, share of the timed kernel time}
    from d3pm_prof_read_class tuples (launches, ms, flops, bytes)."""
    table = {}
    for name, (n, ms, fl, by) in per_class.items():
        if n == 0:
            continue
        sec = ms * 1e-3
        row = {"launches_timed": n, "avg_launch_us": ms * 1e3 / n, "share_of_timed_kernel_time": None,
               "algorithmic_gflop_per_launch": fl / n / 1e9, "algorithmic_mb_per_launch": by / n / 1e6,
               "tflops": fl / sec / 1e12 if sec > 0 else 0.0, "gbs": by / sec / 1e9 if sec > 0 else 0.0}
        bound = "mfma" if name in ("gemm", "attention", "gemm_layernorm") and fl > 0 else "hbm"
        if name == "gemm_layernorm":
            bound = "hbm"      # 2 x 125 MB through HBM around 12.9 + 25.8 GFLOP per block: priced against both roofs
            row["frac_of_mfma_peak"] = row["tflops"] / MFMA_PEAK_TFLOPS[key]
        row["bound"] = bound
        row["frac_of_bound"] = (row["tflops"] / MFMA_PEAK_TFLOPS[key]) if bound == "mfma" else (row["gbs"] / HBM_PEAK_GBS)
        table[name] = row
    total_ms = sum(v[1] for v in per_class.values())
    for name, row in table.items():
        row["share_of_timed_kernel_time"] = per_class[name][1] / total_ms if total_ms > 0 else None
    return table


CLASSES = (("gemm", 0), ("attention", 1), ("layernorm", 3), ("sample", 2), ("gemm_layernorm", 4))


def vctk_long_prompt(dev, dtype, batch, key):
    """BASELINE.json configs[3]: 10 s prompt (750 keys), 5 s target (375 frames on a 384 canvas),
    200-step schedule (199 iterations), the same d = 512 model family.  Reported beside `value` (the headline stays configs[1]):
    tokens/s of a batch, the p50 of one utterance, and the same per-class table as the headline."""
    from vall_e.vall_e import AR, _hip, synth
    cfg = synth.D3PMConfig.vctk_long_prompt()
    m = AR.from_config(cfg)
    m.load_state_dict(synth.make_state_dict(cfg, 0))
    m = m.to(dtype).to(dev)
    texts, proms = synth.make_inputs(cfg, batch, 1)
    texts, proms = [t.to(dev) for t in texts], [p.to(dev) for p in proms]
    m.generate_audio(texts, proms, seed=1)
    torch.cuda.synchronize()
    reps, iters = 3, cfg.timesteps - 1
    _hip.prof_enable(_hip.K_ALL, reps * (iters // 16 + 2) * (cfg.n_layers * 14 + 4) + 64)
    t0 = time.perf_counter()
    for i in range(reps):
        m.generate_audio(texts, proms, seed=2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    per_class = {name: _hip.prof_read_class(k) for name, k in CLASSES}
    _hip.prof_disable()
    lat = []
    for i in range(5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m.generate_audio(texts[:1], proms[:1], seed=i)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    return {"workload": f"vctk: {cfg_name(cfg)} S_text={cfg.s_text}, {batch} utterances", "seconds_per_batch": dt, "repetitions": reps,
            "codec_tokens_per_s": batch * cfg.n_frames / dt, "p50_utterance_latency_ms": statistics.median(lat[1:]),
            "whole_step_tflops": algorithmic_flops_per_step(cfg, batch) / dt / 1e12, "kernel_classes": class_table(per_class, key)}


def host_cpu_info(cores):
    """What the host is, so that the cpu_baseline figure can be read: CPU model, torch's threading / BLAS back ends, and
    the rate of a plain eager matmul at the denoiser's fc1 shape per dtype (a host whose fp16 GEMM takes a slow reference
    path shows up here as a large f32 / f16 ratio -- the round-2 GPU box ran the fp16 oracle 20x slower per iteration
    than this build's 8-vCPU container)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    par = [ln.strip() for ln in torch.__config__.parallel_info().splitlines() if ln.strip()]
    rates = {}
    a32, b32 = torch.randn(768, 512), torch.randn(2048, 512)
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        a, b = a32.to(dt), b32.to(dt)
        torch.nn.functional.linear(a, b)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 0.3:
            torch.nn.functional.linear(a, b)
            reps += 1
        rates[name] = 2.0 * 768 * 512 * 2048 * reps / (time.perf_counter() - t0) / 1e9
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "threads_granted": cores,
            "torch_parallel_info": par[:12], "eager_linear_768x512x2048_gflops": rates}


def n_q8_extension(dev, dtype, batch, texts, proms):
    """SURVEY.md section 8d config 2, second form (BASELINE.json configs[1] "750 codec frames x 8 quantizers"): the D3PM
    denoising all 8 quantizer levels of a frame jointly (AR(..., n_q=8): summed level embeddings in, 8 x 1025 logits out, every
    (frame, level) sampled like a level-0 token).  An extension of this build -- the reference generates level 0 only -- so
    it is reported beside `value`, never inside it."""
    from vall_e.vall_e import AR, synth
    cfg8 = synth.D3PMConfig.libritts_8q()
    m = AR.from_config(cfg8)
    m.load_state_dict(synth.make_state_dict(cfg8, 0))
    m = m.to(dtype).to(dev)
    m.generate_audio(texts, proms, seed=1)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for i in range(reps):
        out = m.generate_audio(texts, proms, seed=2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert out.shape == (batch, cfg8.canvas, 8)
    return {"model": "D3PM n_q=8 (extension; no reference counterpart)", "seconds_per_batch": dt, "repetitions": reps,
            "codec_tokens_per_s": 8 * batch * cfg8.n_frames / dt,
            "unmasked_fraction": float((out[:, : cfg8.n_frames] != cfg8.mask_id).float().mean())}


def cpu_baseline(cfg, sd32, texts, proms, n_iters):
    """The oracle (op-for-op port of the reference sampler incl. its dense 1025x1025 table matmuls
    and need_weights=True attention), fp16 like the reference, for `n_iters` diffusion iterations of ONE utterance,
    extrapolated to the full utterance -- timed at all granted host cores AND at half of them (an oversubscribed or
    SMT-shared host runs faster at half), the faster of the two is `value`."""
    from oracle import d3pm_oracle as O
    cores = host_cores()
    info = host_cpu_info(cores)
    note(f"cpu_baseline: {info['cpu_model']}, {cores} threads granted; eager linear GFLOP/s {info['eager_linear_768x512x2048_gflops']}")
    torch.set_num_threads(cores)
    note("cpu_baseline: building the reference-style dense tables")
    orc = O.Oracle({k: v.half() for k, v in sd32.items()}, O.Shape.of(cfg), dense=True)
    noise = O.philox_noise(123, cfg.canvas)
    runs = {}
    with torch.no_grad():
        for threads in sorted({cores, max(1, cores // 2)}, reverse=True):
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            x, mask = orc.canvas_init()
            cp, ct = orc.conditions(texts[0], proms[0])
            t_cond = time.perf_counter() - t0
            t0 = time.perf_counter()
            for t in range(cfg.timesteps - 1, cfg.timesteps - 1 - n_iters, -1):
                x = orc.step(x, t, cp, ct, mask, noise(t, 0))
            t_iter = (time.perf_counter() - t0) / n_iters
            runs[threads] = (t_cond, t_iter)
            note(f"cpu_baseline: {threads} threads: condition encoders {t_cond:.2f}s, {t_iter * 1e3:.0f} ms per iteration")
    torch.set_num_threads(cores)
    best = min(runs, key=lambda k: runs[k][0] + runs[k][1] * (cfg.timesteps - 1))
    t_cond, t_iter = runs[best]
    per_utt = t_cond + t_iter * (cfg.timesteps - 1)
    out = {"value": cfg.n_frames / per_utt, "unit": "codec_tokens/s", "cores": best, "kind": "port",
           "sample": f"1 utterance, {n_iters} of {cfg.timesteps - 1} diffusion iterations timed "
                     f"({t_iter * 1e3:.0f} ms each) + condition encoders ({t_cond * 1e3:.0f} ms), extrapolated; "
                     "fp16 eager PyTorch CPU, dense transition tables",
           "by_threads": {str(k): {"ms_per_iteration": v[1] * 1e3, "condition_encoders_ms": v[0] * 1e3,
                                   "codec_tokens_per_s": cfg.n_frames / (v[0] + v[1] * (cfg.timesteps - 1))} for k, v in runs.items()},
           "host": info}
    # the same measurement taken in the build container (python bench.py --cpu-only > profiles/...): what the port costs on a
    # host whose fp16 eager path is not degraded -- a calibration beside the GPU box's figure, never a replacement for it
    try:
        import glob
        ref = sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_baseline_container.json")))
        if ref:
            c = json.load(open(ref[-1]))
            if c.get("workload") == cfg_name(cfg):
                out["container_calibration"] = {"source": os.path.relpath(ref[-1], ROOT), "value": c["value"], "cores": c["cores"],
                                                "sample": c["sample"], "cpu_model": c["host"]["cpu_model"]}
    except (OSError, ValueError, KeyError):
        pass
    return out


def reduce_over_ranks(elapsed: float, world: int, dev):
    """(max over ranks of the timed region, number of ranks that took part): the second is an all-reduce of ones, so a line
    that says n_gpus = N was produced by N live ranks."""
    if world == 1:
        return elapsed, 1
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ones = torch.ones(1, device=dev, dtype=torch.int32)
    torch.distributed.all_reduce(ones, op=torch.distributed.ReduceOp.SUM)
    return t.item(), int(ones.item())


def stand_in_main(args, rank, world):
    """The N-rank path of this script without a GPU: same sharding (dp.generate_audio_dp), same barrier / max-over-ranks timing, same
    JSON contract, a CPU stand-in where the HIP sampler would run.  A rehearsal, never a measurement."""
    import torch.distributed as dist
    from vall_e.vall_e import dp, synth
    if args.backend != "gloo":
        raise SystemExit("--stand-in runs on the CPU: use --backend gloo")
    if world > 1:
        dist.init_process_group("gloo")
    cfg = {"libritts": synth.D3PMConfig.libritts, "native": synth.D3PMConfig.native,
           "vctk": synth.D3PMConfig.vctk_long_prompt}[args.config]()
    batch = args.batch or (32 if args.config != "native" else 1)
    model = StandInModel(cfg)
    texts, proms = synth.make_inputs(cfg, batch * world, 1)

    def fence():
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        dp.generate_audio_dp(model, texts, proms, seed=123 + i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = dp.generate_audio_dp(model, texts, proms, seed=123 + args.warmup + i)
    fence()
    elapsed, ranks_seen = reduce_over_ranks(time.perf_counter() - t0, world, torch.device("cpu"))
    assert out.shape == (batch * world, cfg.canvas)
    single = StandInModel(cfg).generate_audio(texts, proms, seed=123 + args.warmup + args.steps - 1).reshape(batch * world, -1)
    assert torch.equal(out, single.long()), "gathered grid differs from the one-rank grid"
    if rank == 0:
        print(json.dumps({"metric": "EnCodec codec-tokens/sec (whole node), 100-step D3PM", "value": batch * world * cfg.n_frames * args.steps / elapsed,
                          "unit": "codec_tokens/s", "n_gpus": world, "n_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "int64", "data": "synthetic",
                          "config": {"workload": f"{args.config}: CPU stand-in for the sampler", "utterances_per_gpu": batch,
                                     "global_batch": batch * world, "parallelism": f"dp{world}"},
                          "stand_in": True, "invalid_for_headline": "rehearsal of the launcher / sharding / gather path on the CPU: no kernel ran"}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cfg_name(cfg):
    return f"d={cfg.d_model} H={cfg.n_heads} L={cfg.n_layers} T={cfg.n_frames}/{cfg.canvas} S_prompt={cfg.s_prompt} steps={cfg.timesteps - 1}"


def launch_ranks(args) -> int:
    """--gpus N > 1 outside torchrun: this process becomes the launcher.  It makes no GPU call (torch.cuda.device_count() only
    counts devices), starts `python -m torch.distributed.run` with one rank per GPU as a CHILD process (never an exec: a
    process that has touched the GPU must not be replaced) and returns the child's exit code; the ranks inherit stdout, so
    rank 0's JSON line is this command's JSON line."""
    import socket
    import subprocess
    n = args.gpus
    if not args.stand_in:
        have = torch.cuda.device_count()
        if have < n and args.backend == "nccl":
            print(f"[bench] --gpus {n} but only {have} GPU(s) are visible: refusing to start (RCCL needs one device per rank; "
                  "--backend gloo rehearses the path with ranks sharing a device)", file=sys.stderr, flush=True)
            return 2
        if have < 1:
            print("[bench] no GPU visible", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


class StandInModel:
    """--stand-in: what dp.generate_audio_dp needs from a model, on the CPU.  The ids of an utterance are a function of
    (seed, GLOBAL utterance index) only, like the Philox rows of the real sampler, so the gathered grid is the same for
    every rank count."""

    def __init__(self, cfg):
        self.cfg, self.device = cfg, torch.device("cpu")

    def generate_audio(self, text_list, proms_list, *, seed, utt0=0, **_):
        rows = []
        for b in range(len(text_list)):
            g = torch.Generator().manual_seed(int(seed) * 100003 + utt0 + b)
            rows.append(torch.randint(0, self.cfg.n_classes, (self.cfg.canvas,), generator=g))
        return torch.stack(rows) if len(rows) > 1 else rows[0]


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.cpu_only:
        sys.exit(launch_ranks(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if args.cpu_only:
        from vall_e.vall_e import synth
        cfg = {"libritts": synth.D3PMConfig.libritts, "native": synth.D3PMConfig.native,
               "vctk": synth.D3PMConfig.vctk_long_prompt}[args.config]()
        texts, proms = synth.make_inputs(cfg, 1, 1)
        res = cpu_baseline(cfg, synth.make_state_dict(cfg, 0), texts, proms, max(args.cpu_steps, 1))
        res["workload"] = cfg_name(cfg)
        res.pop("container_calibration", None)
        print(json.dumps(res), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stand_in:
        return stand_in_main(args, rank, world)
    local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import __graft_entry__ as g
    g.build()
    from vall_e.vall_e import AR, _hip, dp, synth
    for kv in filter(None, args.tune.split(",")):
        field, value = kv.split("=")
        _hip.set_tuning_field(field.strip(), int(value))

    cfg = {"libritts": synth.D3PMConfig.libritts, "native": synth.D3PMConfig.native,
           "vctk": synth.D3PMConfig.vctk_long_prompt}[args.config]()
    batch = args.batch or (32 if args.config != "native" else 1)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    sd32 = synth.make_state_dict(cfg, 0)
    model = AR.from_config(cfg)
    model.load_state_dict(sd32)
    model = model.to(dtype).to(dev)
    model.loop_streams = args.streams
    texts, proms = synth.make_inputs(cfg, batch * world, 1)
    texts = [t.to(dev) for t in texts]
    proms = [p.to(dev) for p in proms]

    kw = {"steps": args.profile_iters} if args.profile_iters else {}

    def step(i):
        return dp.generate_audio_dp(model, texts, proms, seed=123 + i, **kw)   # shards by rank, all-gathers ids

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        note(f"warmup {i + 1}/{args.warmup} done")
    fence()
    iters = args.profile_iters or (cfg.timesteps - 1)
    if not args.no_kernel_events:     # every kernel class, launches inside the diffusion loop only, each 16th iteration
        _hip.prof_enable(_hip.K_ALL, args.steps * (iters // 16 + 2) * (cfg.n_layers * 14 + 4) + 64)
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    per_class = {name: _hip.prof_read_class(k) for name, k in CLASSES}
    launches, gemm_ms, gemm_flops, gemm_bytes = per_class["gemm"]
    _hip.prof_disable()
    elapsed, ranks_seen = reduce_over_ranks(elapsed, world, dev if args.backend == "nccl" else torch.device("cpu"))
    assert out.shape == (batch * world, cfg.canvas)

    tokens = batch * world * cfg.n_frames * args.steps
    ms_per_step = elapsed / args.steps * 1e3
    result = {
        "metric": "EnCodec codec-tokens/sec (whole node), 100-step D3PM",
        "value": tokens / elapsed, "unit": "codec_tokens/s", "n_gpus": world, "n_ranks_seen": ranks_seen, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.config}: d={cfg.d_model} H={cfg.n_heads} L={cfg.n_layers} "
                               f"T={cfg.n_frames}/{cfg.canvas} S_text={cfg.s_text} S_prompt={cfg.s_prompt} "
                               f"{iters} diffusion iterations, n_q=1",
                   "utterances_per_gpu": batch, "global_batch": batch * world, "parallelism": f"dp{world}",
                   "streams_per_gpu": args.streams},
    }
    if args.tune:
        result["tuning_overrides"] = args.tune
        result["invalid_for_headline"] = f"A/B run with tuning overrides {args.tune}: not the shipped configuration"
    if args.profile_iters:
        result["invalid_for_headline"] = result.get("invalid_for_headline", "") + f" profiling run: {iters} of {cfg.timesteps - 1} diffusion iterations"
    if rank == 0:
        key = args.dtype
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic("gemm") if args.config == "libritts" and batch == 32 else (None, "PMC passes cover the headline workload only")
        result["roofline"] = {"bound": "mfma", "kernel": "gemm_mfma_* (every projection / MLP / final GEMM launch inside the diffusion loop -- the "
                              "largest class; with the LayerNorms folded into the projections (d3pm_tuning.ln_fold, the default) that is 37 launches "
                              "per iteration and there is no stand-alone LayerNorm or row-panel launch left: the `layernorm` row of kernel_classes is "
                              "the token-embedding gather and the per-iteration fc1 FiLM fold)",
                              "achieved": achieved, "peak": MFMA_PEAK_TFLOPS[key], "unit": "TFLOP/s",
                              "frac": achieved / MFMA_PEAK_TFLOPS[key], "traffic": traffic,
                              "traffic_unit": "bytes/launch (L2 fabric-side, rocprofv3 PMC)", "traffic_source": traffic_src,
                              "build_id": build_id(),
                              "algorithmic_bytes_per_launch": gemm_bytes / max(launches, 1),
                              "algorithmic_flops_per_launch": gemm_flops / max(launches, 1),
                              "launches_timed": launches, "avg_launch_us": gemm_ms * 1e3 / max(launches, 1),
                              "timing": "HIP event pairs on the launch stream around every launch of each 16th diffusion "
                                        "iteration of the timed region (bracketing all launches costs ~6 % of the step)"}
        result["kernel_classes"] = class_table(per_class, key)
        result["launches_per_iteration"] = {name: round(v[0] / max(args.steps * len([t for t in range(iters, 0, -1) if t % 16 == 0]), 1), 2)
                                            for name, v in per_class.items() if v[0]}
        whole = algorithmic_flops_per_step(cfg, batch) * iters / (cfg.timesteps - 1) / (ms_per_step * 1e-3) / 1e12
        result["whole_step_tflops"] = whole
        if not args.no_latency:
            note(f"timed region {elapsed:.2f}s; measuring single-utterance latency")
            # third form: one utterance run as a shard of a 32-utterance logical batch (global_batch = 32: it takes the 32 x 32 x 16
            # attention kernels of the big batch, so that its ids are those of the unsplit batch) -- what rank-count independence costs
            for key, use_graph, gb in (("p50_utterance_latency_ms", False, None), ("p50_utterance_latency_graph_replay_ms", True, None),
                                       ("p50_utterance_latency_as_shard_of_32_ms", False, 32)):
                lat = []
                for i in range(6):       # the first two calls of the graph mode warm up and capture
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    model.generate_audio(texts[:1], proms[:1], seed=i, graph=use_graph, global_batch=gb)
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t1) * 1e3)
                result[key] = statistics.median(lat[2:])
        if not args.no_nar and args.config == "libritts":
            result["nar_levels_1to7"] = nar_stage(dev, dtype, batch, cfg, out[:batch])
            d3pm_s = ms_per_step * 1e-3
            result["nar_levels_1to7"]["all_8_levels_codec_tokens_per_s"] = (
                8 * batch * cfg.n_frames / (d3pm_s + result["nar_levels_1to7"]["seconds_per_batch"]))
        if world == 1 and not args.no_nq8 and args.config == "libritts" and dtype != torch.float32:
            note("measuring the n_q = 8 extension")
            result["n_q8_extension"] = n_q8_extension(dev, dtype, batch, texts, proms)
        if world == 1 and not args.no_fp8 and args.config == "libritts" and dtype != torch.float32:
            note("measuring the fp8 / 50-step fast path")
            result["fp8_fast_path"] = fp8_fast_path(dev, dtype, batch, cfg, sd32, texts, proms)
        if world == 1 and not args.no_vctk and args.config == "libritts" and dtype != torch.float32:
            note("measuring the VCTK long-prompt config")
            result["vctk_long_prompt"] = vctk_long_prompt(dev, dtype, batch, args.dtype)
        if world == 1 and args.cpu_steps > 0:
            note("timing the CPU port of the reference sampler")
            cpu_texts, cpu_proms = synth.make_inputs(cfg, 1, 1)
            result["cpu_baseline"] = cpu_baseline(cfg, sd32, cpu_texts, cpu_proms, args.cpu_steps)
            result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be timing its extras: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
