"""CPC train-step benchmark: encoder output frames per second through one full train step
(AudioEncoder -> GRU -> predictor -> InfoNCE -> backward -> [RCCL all-reduce] -> Adam), inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: BASELINE.json configs[1] — AudioPredictiveCodingModel (5-layer strided AudioEncoder 512 ch + AudioGRUModel 256,
12 prediction steps, 100 visible steps), batch 256 per GPU, 20480-sample 16 kHz synthetic clips, bf16 storage / f32
accumulation, softplus score, regularisation 1.0, Adam lr 1e-4.  Weak scaling: every rank runs its own 256 clips with its
own in-batch negatives; gradients are averaged with one RCCL all-reduce per step.

--workload selects the configuration (default cfg1 = BASELINE configs[1], the one the metric is quoted on; the others are BASELINE
configs[2] / configs[3] at their stated sizes and print the same JSON line with their own dominant kernel in `roofline`):
  scalogram   CQT (256 bins, hop 128) + PreprocessingModule + scalogram_resnet_architecture_7 + ar_conv_architecture_3, V = 60, K = 16,
              128 clips of item_length = 97 024 samples per GPU (76 encoder frames per clip); the CQT is part of the step
  conv_ar     AudioEncoder(512) + ConvolutionalArModel(ar_conv_architecture_3), V = 60, K = 12, 256 clips of 20 480 samples
  attention   AudioEncoder(512) + AttentionModel(attention_architecture_1, train-mode dropout 0.1), same shapes

The driver's protocol is what runs: W untimed warm-up steps, then exactly K timed steps (--prewarm adds further untimed steps in front,
default 0: the first dozen steps of a process are 6-8 % slower in every kernel while the clocks ramp; `prewarm_steps` records it).
With --gpus N > 1 and no WORLD_SIZE in the environment the script starts the N ranks itself (`python -m torch.distributed.run` as a
child process, decided before anything touches the GPU) and relays rank 0's JSON line.

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline      dominant kernel (the bf16 MFMA gemm_nt that carries the conv forward + data-gradient GEMMs): algorithmic
                FLOPs of its launches / their HIP-event durations measured inside the timed region, on a sample of its steps
                (every tenth: an event pair around a launch leaves ~12 us of idle queue, see docs/DESIGN_HISTORY_r1-r3.md section 6).  In the step the
                weight-gradient GEMMs run beside these launches on a second stream; `alone` = the same kernel with everything on
                one stream, from 6 extra untimed steps after the timed region.
  hbm_kernel    the HBM-bound piece of the conv stack (encoder layer 1, C_in = 1): algorithmic bytes / HIP-event duration vs 8 TB/s.
  score_gemm    the InfoNCE score contraction alone (both loss branches): algorithmic FLOPs / HIP-event time vs the bf16 MFMA peak.
  trainer_ms_per_step   the same step through ContrastiveEstimationTrainer.train (sampler, logger, loss readback, NaN guard).
  cpu_baseline  the CPU oracle (oracle/cpc_oracle.py, kind "port") timed on this host: CPU model, core count, 3 warm-up + median
                of 10 steps, and the same under autograd anomaly detection (how the reference's train() runs).
  n_ranks_seen  (N > 1) ranks that took part in the run's collectives.
  ms_per_step_median / _max / first_timed_step_ms / slowest_timed_step    the K timed steps one by one: one HIP event per step boundary
                (device-side interval between consecutive boundaries) — a single stall shows up here, the mean hides it.
  secondary     (default run, N = 1) BASELINE configs[2] / configs[3] — scalogram, conv_ar, attention — each with the same W / K protocol in
                this process after the headline region: {ms_per_step, clips_per_s, roofline.frac, loss_last_step, ...}.
  data_parallel (N > 1) what a poor scaling number would need explained: per-rank ms_per_step, the main stream's exposed wait for the
                gradient all-reduce (HIP events around GradAllReduce.finish() on the sampled steps), the bucket plan and its bytes.

Timed region = exactly K steps.  Everything a timed step does has run before it: the optimizer's prepare_ahead callback and the kernel
timer are installed BEFORE the warm-up, the last warm-up step is a sampled (event-bracketed) step, and the per-step events exist before
t0 (round 3's driver run lost 41 ms once, at the head of the timed loop, to first-use work that only the timed region did).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
DOMINANT = "gemm_nt<bf16,bf16,256>"      # = gemm_nt_fast_kernel<bf16, bf16, 2, 4, 8, 4> (256x256 tile)



class StepClock:
    """One host timestamp and one HIP event per step boundary (n steps -> n + 1 marks; the events are created up front)."""

    def __init__(self, n):
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        self.host = [0.0] * (n + 1)
        self.i = 0

    def mark(self):
        self.host[self.i] = time.perf_counter()
        self.ev[self.i].record()
        self.i += 1

    def stats(self):
        """Call after a synchronize.  Device-side interval between consecutive step boundaries + the host's enqueue time per step."""
        n = self.i - 1
        if n <= 0:
            return {}
        dev = [self.ev[i].elapsed_time(self.ev[i + 1]) for i in range(n)]
        host = [(self.host[i + 1] - self.host[i]) * 1e3 for i in range(n)]
        srt = sorted(dev)
        med = 0.5 * (srt[(n - 1) // 2] + srt[n // 2])
        worst = max(range(n), key=lambda i: dev[i])
        return {"ms_per_step_median": round(med, 3), "ms_per_step_max": round(dev[worst], 3), "slowest_timed_step": worst,
                "first_timed_step_ms": round(dev[0], 3), "ms_per_step_min": round(srt[0], 3),
                "host_enqueue_ms_max": round(max(host), 3), "host_enqueue_slowest_step": max(range(n), key=lambda i: host[i]),
                "host_enqueue_ms_median": round(sorted(host)[n // 2], 3), "host_enqueue_ms_first": round(host[0], 3),
                # every timed step, in order (HIP events at the step boundaries): a slow head of the region (the first steps after the
                # warm-up's fence) or a single stall shows here; steps with the kernel timer active carry ~0.1 ms of event brackets
                "step_ms": [round(v, 3) for v in dev]}


def timed_run(step, steps, warmup, timer, every, fence):
    """The driver's protocol: `warmup` untimed steps, then exactly `steps` timed ones between two fences.  The warm-up runs the SAME code as
    the timed loop — step clock marks, and its last step with the kernel timer active — so that no allocation, event kind, library page or
    Python path is used for the first time inside the timed region.  Returns (out, elapsed_s, host_enqueue_s, clock)."""
    # Python's cyclic garbage collector: a full (generation 2) collection walks every tracked object of the process — 38 - 43 ms here, with
    # torch's module trees alive — and lands wherever the allocation count happens to cross its threshold.  It is what round 3's driver run
    # lost at the head of its timed loop, and what runs with other allocation histories lost (or did not) elsewhere.  Everything alive now
    # is long-lived: collect once, BEFORE the warm-up (the GPU is idle anyway), and move it to the permanent generation, so that later
    # collections only look at what the steps allocate.  The trainer's own loop does the same (ContrastiveEstimationTrainer.train).
    import gc
    gc.collect()
    gc.freeze()
    wclock = StepClock(warmup)
    wclock.mark()
    for i in range(warmup):
        timer.active = (i == warmup - 1)
        step(i)
        wclock.mark()
    fence()
    if warmup:
        wclock.stats()
    timer.reset()
    gc.freeze()          # (what the warm-up created — events, timer records — joins the permanent generation: no walk, no pause)
    clock = StepClock(steps)
    sample_at = every // 2
    out = None
    fence()
    t0 = time.perf_counter()
    clock.mark()
    for i in range(steps):
        timer.active = (i % every == sample_at)
        out = step(i)
        clock.mark()
    host_enqueue = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    timer.active = False
    return out, elapsed, host_enqueue, clock


def build_model(dtype, device, seed=0):
    from cpc_audio_amd.audio_model import AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel
    torch.manual_seed(seed)
    model = AudioPredictiveCodingModel(AudioEncoder(), AudioGRUModel(512, 256), enc_size=512, ar_size=256,
                                       visible_steps=100, prediction_steps=12, compute_dtype=dtype)
    return model.to(device)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline():
    """The oracle's train step (torch CPU ops, f32) on BASELINE config 1's shape (B = 8 clips of 20480 samples), SURVEY.md 8(d)
    protocol: all host cores of this process, 3 warm-up steps, median of 10 timed steps; headline with autograd anomaly detection
    off, plus the same with it on — the reference's train() always runs under set_detect_anomaly(True)
    (contrastive_estimation_training.py:96)."""
    from oracle import cpc_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:        # cgroup v2 CPU quota of the GPU box's container, when present
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    threads = torch.get_num_threads()
    B, L, warm, timed = 8, 20480, 3, 10
    data = torch.randn(4 * B, L, generator=torch.Generator().manual_seed(0))

    def run(anomaly):
        tr = O.OracleTrainer(O.init_params(seed=0), 100, 12, score="softplus", regularization=1.0, lr=1e-4)
        times = []
        with torch.autograd.set_detect_anomaly(anomaly):
            for i in range(warm + timed):
                t0 = time.perf_counter()
                tr.step(data[(i % 4) * B:(i % 4 + 1) * B])
                if i >= warm:
                    times.append(time.perf_counter() - t0)
        times.sort()
        return 0.5 * (times[(timed - 1) // 2] + times[timed // 2])

    med, med_anomaly = run(False), run(True)
    return {"value": round(B * 126 / med, 1), "unit": "frames/s", "cores": threads, "kind": "port", "cpu": _cpu_model(),
            "anomaly_on": {"value": round(B * 126 / med_anomaly, 1), "ms_per_step": round(med_anomaly * 1e3, 1)},
            "sample": f"median of {timed} train steps after {warm} warm-up steps, B=8 x 20480 samples (config 1 shape), f32, "
                      f"{threads} torch threads, {med * 1e3:.0f} ms/step; anomaly_on = the same under torch.autograd."
                      f"set_detect_anomaly(True), as the reference's train() runs"}


def _hip_call(*a, **k):
    from cpc_audio_amd import _hip
    return _hip.call(*a, **k)


def _hip_ptr(t):
    from cpc_audio_amd import _hip
    return _hip.ptr(t)


def score_gemm_figures(eng, launches=50):
    """HIP-event time of the InfoNCE score contraction (contrastive_estimation_training.py:12-22) on the engine's own buffers:
    the equal-step form the default branch runs (K batched B x E x B products) and the full (B K) x E x (B K) form of
    score_over_all_timesteps=True; algorithmic FLOPs / time against the dense bf16 MFMA peak."""
    out = {}
    for name, fn, shape in (("default", eng.score_gemm, f"{eng.K} x ({eng.B} x {eng.E} x {eng.B})"),
                            ("all_timesteps", eng.score_gemm_all, f"{eng.B * eng.K} x {eng.E} x {eng.B * eng.K}")):
        for _ in range(5):
            flops = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
        tf = flops / (us * 1e-6) / 1e12
        out[name] = {"shape": shape, "gflop": round(flops / 1e9, 3), "us_per_launch": round(us, 2), "achieved": round(tf, 1),
                     "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS, "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4)}
    if eng.fused_scores_ok():
        # the all-timesteps contraction as the train step runs it since round 4: column log-sum-exp pairs + diagonal + bf16 scores from one launch
        import ctypes as C
        R, E = eng.B * eng.K, eng.E
        fs = lambda n, dt=torch.float32: torch.empty(n, device=eng.device, dtype=dt)
        pm, ps, valid, sb = fs(R // 256 * R), fs(R // 256 * R), fs(R), fs(R * R)
        tg = fs(R * E, eng.dt).normal_()
        run = lambda: _hip_call("cpc_score_lse", _hip_ptr(eng.pred), _hip_ptr(tg), _hip_ptr(sb), _hip_ptr(pm), _hip_ptr(ps), _hip_ptr(valid), R, R, E,
                                C.c_longlong(E), C.c_longlong(E), C.c_longlong(R), 0)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / launches
        tf = 2.0 * R * R * E / (us * 1e-6) / 1e12
        out["all_timesteps_fused"] = {"shape": f"{R} x {E} x {R} + column log-sum-exp pairs + f32 scores", "gflop": round(2.0 * R * R * E / 1e9, 3),
                                      "us_per_launch": round(us, 2), "achieved": round(tf, 1), "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS,
                                      "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                                      "note": "replaces two f32-output score GEMMs and the split column pass of the unfused path"}
    out["note"] = ("back-to-back launches (includes the launch-to-launch gap); one launch per train step in the product path.  The north "
                   "star's >= 50 % MFMA target is DEFINED on global_8gpu: at B = 256 one launch is 0.8 / 9.7 GFLOP on 256 CUs, i.e. bound by "
                   "its prologue / epilogue latency, not by the matrix pipe")
    out["global_8gpu"] = global_score_gemm(eng)
    return out


def global_score_gemm(eng, launches=10):
    """The score contraction at the size BASELINE configs[4] gives it under the reference's DataParallel semantics (global negatives,
    score_over_all_timesteps=True: 8 ranks x 256 clips x 12 steps = 24 576 predictions against 24 576 targets, E = 512):
    cpc_gemm_nt on 24 576 x 512 x 24 576, bf16 operands, f32 scores (2.4 GB) -- the launch engine.GlobalNegatives issues on every rank --
    and the same product with storage-dtype output (what the gradient-side contractions read)."""
    from cpc_audio_amd import _hip
    R, E = 8 * eng.B * eng.K, eng.E
    dev = eng.device
    g = torch.Generator(device="cpu").manual_seed(5)
    a = (torch.randn(R, E, generator=g) * 0.5).to(dev).to(eng.dt)
    b = (torch.randn(R, E, generator=g) * 0.5).to(dev).to(eng.dt)
    res = {"shape": f"{R} x {E} x {R}", "gflop": round(2.0 * R * R * E / 1e9, 1)}
    for name, flags, odt in (("f32_scores", _hip.GEMM_OUT_F32, torch.float32), ("bf16_out", 0, eng.dt)):
        c = torch.empty(R * R, device=dev, dtype=odt)
        run = lambda: _hip.gemm_nt(_hip.ptr(a), _hip.ptr(b), _hip.ptr(c), R, R, E, E, E, R, eng.code, flags=flags)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        tf = 2.0 * R * R * E / (ms * 1e-3) / 1e12
        out_gb = R * R * c.element_size() / 1e9
        res[name] = {"ms_per_launch": round(ms, 4), "achieved": round(tf, 1), "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS,
                     "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "output_gb": round(out_gb, 2),
                     "output_tb_per_s": round(out_gb / ms, 2)}
        del c
    # the same contraction with the column log-sum-exp pairs taken in the GEMM's epilogue (cpc_score_lse, what the all-timesteps loss runs
    # on since round 4): ONE f32 score matrix instead of two (or none, 'no_scores': the rate of GEMM + column pairs alone).  And what ONE RANK of an
    # 8-GPU global-negatives step computes since the strips (engine.GlobalNegatives._all_timesteps_strips): 1 / 8 of the columns.
    import ctypes as C
    P, L = _hip.ptr, C.c_longlong
    pm = torch.empty(R // 256, R, device=dev)
    ps = torch.empty(R // 256, R, device=dev)
    valid = torch.zeros(R, device=dev)

    def timed(run, n=launches):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for name, keep in (("fused_lse_f32_scores", True), ("fused_lse_no_scores", False)):
        sb = torch.empty(R * R, device=dev, dtype=torch.float32) if keep else None
        ms = timed(lambda: _hip.call("cpc_score_lse", P(a), P(b), P(sb), P(pm), P(ps), P(valid), R, R, E, L(E), L(E), L(R), 0))
        tf = 2.0 * R * R * E / (ms * 1e-3) / 1e12
        res[name] = {"ms_per_launch": round(ms, 4), "achieved": round(tf, 1), "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS,
                     "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4), "output_gb": round((R * R * 4 if keep else 0) / 1e9 + 2 * pm.numel() * 4 / 1e9, 3)}
        del sb
    Rl = R // 8
    sb = torch.empty(R * Rl, device=dev, dtype=torch.float32)
    ms = timed(lambda: _hip.call("cpc_score_lse", P(a), P(b), P(sb), P(pm), P(ps), P(valid), R, Rl, E, L(E), L(E), L(Rl), 0))
    tf = 2.0 * R * Rl * E / (ms * 1e-3) / 1e12
    res["one_rank_column_strip"] = {"shape": f"{R} x {E} x {Rl}", "gflop": round(2.0 * R * Rl * E / 1e9, 1), "ms_per_launch": round(ms, 4),
                                    "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TFLOPS, 4),
                                    "note": "what a rank of an 8-GPU global-negatives step runs: all predictions x its own targets (and a "
                                            "second strip of the same size for its own predictions), not the whole matrix"}
    del sb, pm, ps
    torch.cuda.empty_cache()
    return res


def trainer_loop_ms(model, B, L, device, steps=60, warmup=10):
    """ms per step through the drop-in surface itself — ContrastiveEstimationTrainer.train with a logger attached, FileBatchSampler,
    device-resident synthetic dataset, loss readback and NaN guard — as opposed to the engine calls the headline loop issues."""
    from cpc_audio_amd.audio_dataset import SyntheticAudioDataset
    from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer, softplus_score_function

    class Meter:
        def __init__(self):
            self.n = 0

        def update(self, v):
            self.n += 1

    class Logger:
        def __init__(self):
            self.loss_meter, self.score_meter, self.marks = Meter(), Meter(), []

        def log(self, step):
            self.marks.append(time.perf_counter())

    import contextlib
    ds = SyntheticAudioDataset(B * 8, L, seed=3, device=device)
    logger = Logger()
    with contextlib.redirect_stdout(sys.stderr):          # stdout carries the one JSON line only
        tr = ContrastiveEstimationTrainer(model=model, dataset=ds, logger=logger, device=device, regularization=1.0,
                                          score_function=softplus_score_function, prediction_steps=12, ar_size=256)
        tr.verbose = False
        torch.cuda.synchronize()
        tr.train(batch_size=B, epochs=1000, lr=1e-4, num_workers=0, max_steps=warmup + steps)
        torch.cuda.synchronize()
    marks = logger.marks
    # the logger sees step i one step late (host_sync_lag): steady-state spacing of the log calls = time per step
    span = marks[-1] - marks[warmup]
    return round(span / (len(marks) - 1 - warmup) * 1e3, 3), len(marks)


# ------------------------------------------------------------------------------------------------ BASELINE configs[2] / configs[3]
def secondary_workload(name, dtype, device, B):
    """Model, input geometry and CPU-oracle arguments of --workload scalogram / conv_ar / attention (SURVEY.md 8(d) cfg 3 / cfg 4)."""
    from types import SimpleNamespace
    from cpc_audio_amd import configs
    from cpc_audio_amd.audio_model import AudioEncoder, AudioPredictiveCodingModel, ConvolutionalArModel
    torch.manual_seed(0)
    if name == "scalogram":
        from cpc_audio_amd.scalogram_model import PreprocessingModule, ScalogramResidualEncoder, cqt_default_dict
        V, K = 60, 16
        pre = PreprocessingModule(cqt_dict=cqt_default_dict, phase=True)
        enc = ScalogramResidualEncoder(args_dict=configs.fresh(configs.scalogram_resnet_architecture_7), preprocessing_module=pre)
        ar_cfg = configs.fresh(configs.ar_conv_architecture_3)
        model = AudioPredictiveCodingModel(enc, ConvolutionalArModel(dict(ar_cfg)), enc_size=512, ar_size=256, visible_steps=V,
                                           prediction_steps=K, compute_dtype=dtype)
        pre = pre.to(device)
        pre.cqt.precision = "bf16x3" if dtype == "bf16" else "fp32"
        L = int(model.item_length)
        desc = ("BASELINE configs[2]: CQT(256 bins, hop 128, bf16x3) + PreprocessingModule(phase) + scalogram_resnet_architecture_7 + "
                f"ConvolutionalArModel(ar_conv_architecture_3), V=60, K=16, softplus score, reg 1.0, Adam; per-GPU batch {B} x {L} samples")
        return SimpleNamespace(model=model.to(device), pre=pre, L=L, V=V, K=K, desc=desc, amp=0.1, frames=None,
                               oracle=dict(scalogram=[dict(b.cfg) for b in enc.blocks], conv_ar=dict(ar_cfg)), oracle_B=2)
    V, K, L = 60, 12, 20480
    if name == "conv_ar":
        ar_cfg = configs.fresh(configs.ar_conv_architecture_3)
        ar, oracle = ConvolutionalArModel(dict(ar_cfg)), dict(conv_ar=dict(ar_cfg))
        what = "ConvolutionalArModel(ar_conv_architecture_3: six k=5 blocks, BatchNorm1d + residual)"
    else:
        from cpc_audio_amd.attention_model import AttentionModel
        att = configs.fresh(configs.attention_architecture_1)
        ar, oracle = AttentionModel(dict(att)), dict(attention=(att["num_layers"], att["num_heads"]))
        what = "AttentionModel(attention_architecture_1: 3 layers, 8 heads, FF 512, dropout 0.1 in train mode)"
    model = AudioPredictiveCodingModel(AudioEncoder(), ar, enc_size=512, ar_size=256, visible_steps=V, prediction_steps=K, compute_dtype=dtype)
    desc = (f"BASELINE configs[3]: AudioEncoder(5x512) + {what}, V=60, K=12, softplus score, reg 1.0, Adam; per-GPU batch "
            f"{B} x {L} samples (126 frames/clip)")
    return SimpleNamespace(model=model.to(device), pre=None, L=L, V=V, K=K, desc=desc, amp=1.0, frames=126, oracle=oracle, oracle_B=8)


def cpu_baseline_secondary(name, wl):
    """The oracle's train step of the same model family on a bounded sample: 2 warm-up + median of 5 steps at a small batch."""
    from oracle import cpc_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    torch.set_num_threads(cores)
    threads = torch.get_num_threads()
    B, warm, timed = wl.oracle_B, 2, 5
    params = {k: v.detach().float().cpu().clone() for k, v in wl.model.state_dict().items()}
    tr = O.OracleTrainer(params, wl.V, wl.K, score="softplus", regularization=1.0, lr=1e-4, **wl.oracle)
    data = torch.randn(2 * B, wl.L, generator=torch.Generator().manual_seed(0)) * wl.amp
    weights = consts = None
    if wl.pre is not None:
        weights = [m.weight.detach().cpu() for m in wl.pre.cqt.conv_modules]
        consts = (wl.pre.phase_diff.fixed_phase_diff.detach().cpu().reshape(-1).float(), wl.pre.phase_diff.scaling.detach().cpu().reshape(-1).float())
    times = []
    for i in range(warm + timed):
        t0 = time.perf_counter()
        x = data[(i % 2) * B:(i % 2 + 1) * B]
        if wl.pre is not None:          # the scalogram is part of the step (contrastive_estimation_training.py:100-101)
            with torch.no_grad():
                x = O.preprocessing_forward(O.cqt_forward(x.unsqueeze(1), weights, 128), consts)
        tr.step(x)
        if i >= warm:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    frames = wl.frames
    return {"value": round(B * frames / med, 1), "unit": "frames/s", "cores": threads, "kind": "port", "cpu": _cpu_model(),
            "sample": f"median of {timed} oracle train steps after {warm} warm-up steps, B={B} x {wl.L} samples, f32, {threads} torch "
                      f"threads, {med * 1e3:.0f} ms/step"}


def measure_secondary(name, args, world, rank, device, dist, brief=False):
    """One of --workload scalogram / conv_ar / attention under the headline's protocol; returns the JSON line as a dict (rank 0; None on
    the other ranks).  brief: without the per-kernel pass and the CPU baseline — the form the default run carries under `secondary`."""
    from cpc_audio_amd import _hip
    from cpc_audio_amd.engine import FusedAdam, GradAllReduce
    B = args.batch if name == args.workload and args.batch else (128 if name == "scalogram" else 256)
    wl = secondary_workload(name, args.dtype, device, B)
    model = wl.model
    gen = torch.Generator().manual_seed(1000 + rank)
    pool = [(torch.randn(B, wl.L, generator=gen) * wl.amp).to(device) for _ in range(2)]

    def inputs(i):
        w = pool[i % len(pool)]
        return wl.pre(w.unsqueeze(1)) if wl.pre is not None else w

    x0 = inputs(0)
    eng = model.engine_for(x0) if wl.pre is not None else model.engine(B, wl.L)
    if wl.frames is None:
        wl.frames = int(eng.T)
    opt = FusedAdam(model, lr=1e-4)
    sync = GradAllReduce(model, optimizer=None) if world > 1 else None
    opt.skip_flag = eng.nan_flag()
    opt.after_update = eng.prepare_ahead          # next step's operand copies rebuilt off the critical path (as the trainer sets it)

    # the scalogram of batch i + 1 (CQT GEMMs + pointwise kernel) is computed on the side stream while step i runs, as the trainer does
    # (contrastive_estimation_training.InputAhead; CPC_PREPROCESS_AHEAD=0: on the main stream in front of every step, the reference's
    # order).  Every timed step still issues exactly one preprocessing pass and one train step.
    ahead = None
    if wl.pre is not None and os.environ.get("CPC_PREPROCESS_AHEAD", "1") != "0":
        from cpc_audio_amd.contrastive_estimation_training import InputAhead
        ahead = InputAhead(lambda w: wl.pre(w.unsqueeze(1)), device)
    ahead_at = os.environ.get("CPC_PREPROCESS_AHEAD_AT", "block")

    def step(i):
        after = sync.reduce_flag if sync is not None else None
        if ahead is not None:
            if not ahead.pending:
                ahead.submit(pool[i % len(pool)])
            _, x = ahead.take()
            nxt = pool[(i + 1) % len(pool)]
            if ahead_at == "loss":          # A/B: queued behind the loss kernels, i.e. beside the backward pass
                flag = after

                def after(out_):
                    ahead.submit(nxt)
                    if flag is not None:
                        flag(out_)
            elif ahead_at == "start":       # A/B: in front of the step
                ahead.submit(nxt)
            else:                           # default: where the engine's main queue turns latency-bound (ScalogramCPCEngine.side_job)
                eng.side_job = lambda: ahead.submit(nxt)
        else:
            x = inputs(i)
        out = eng.loss_and_grads(x, softplus=True, regularization=1.0, all_timesteps=args.all_timesteps, after_loss=after)
        if sync is not None:
            sync.finish()                       # one RCCL all-reduce (sum) of the flat gradient buffer
        opt.step(grad_scale=1.0 / world)
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # In the timed region only the launches of the dominant GEMM symbol are bracketed with HIP events, in about four sampled steps (an
    # event pair costs ~12 us of idle queue: bracketing all ~290 launches of a configs[2] step made the sampled steps 3 - 5 ms longer and
    # the reported step 0.8 ms).  The per-kernel table (`kernels`, --breakdown) comes from three extra steps AFTER the timed region.
    dom_name = DOMINANT if args.dtype == "bf16" else "gemm_nt<f32,f32,128>"
    timer = _hip.KernelTimer(only=[dom_name])
    timer.active = False
    _hip.set_timer(timer)
    every = max(1, args.steps // 4)
    out, elapsed, host_enqueue, clock = timed_run(step, args.steps, args.prewarm + args.warmup, timer, every, fence)
    sampled = len([i for i in range(args.steps) if i % every == every // 2])
    _hip.set_timer(None)
    loss = float(out[0])
    dom_summary = timer.summary().get(dom_name)
    full_steps = 0 if brief else 3
    full = _hip.KernelTimer(only=None, by_shape=args.breakdown)
    if full_steps:
        _hip.set_timer(full)
        for i in range(full_steps):
            step(args.steps + i)
        fence()
        _hip.set_timer(None)
    n_ranks_seen = 1
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
    if rank != 0:
        return None
    summary = full.summary() if full_steps else {}
    rows = sorted(summary.items(), key=lambda kv: -kv[1][1])
    total_ms = sum(v[1] for _, v in rows)
    gemms = [(k.split(" ")[0], v) for k, v in rows if v[2] > 0 and k.startswith("gemm_")]
    # the dominant kernel SYMBOL: launches of the same kernel template summed (the timer's key = entry point + dtypes + tile)
    dom_key, dom = dom_name, dom_summary
    largest = gemms[0][0] if gemms else None
    line = {
        "metric": "CPC train-step audio frames/sec (enc+AR+InfoNCE)",
        "value": round(B * wl.frames * world * args.steps / elapsed, 1),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_steps": args.prewarm,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        **clock.stats(),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": wl.desc, "global_batch": B * world, "clip_samples": wl.L, "frames_per_clip": wl.frames,
                   "clips_per_s": round(B * world * args.steps / elapsed, 1), "parallelism": f"dp{world}", "loss_last_step": round(loss, 6)},
    }
    if dom is not None:
        cnt, ms, flops = dom
        peak = 157.3 if dom_key.startswith(("gemm_nt<f32", "gemm_tn<f32")) else MFMA_BF16_PEAK_TFLOPS
        ach = flops / (ms * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                            "traffic": None, "kernel": dom_key, "launches": cnt, "avg_launch_ms": round(ms / cnt, 4),
                            "algorithmic_gflop_per_launch": round(flops / cnt / 1e9, 3),
                            "ms_per_step": round(ms / max(sampled, 1), 4),
                            "note": "HIP events around the launches of this symbol in the sampled steps of the timed region (executed FLOPs "
                                    "booked per launch)"
                                    + ("" if (brief or largest == dom_key or args.breakdown) else
                                       f"; the GEMM symbol with the largest summed time in the per-kernel pass is {largest}")}
    line["host_enqueue_ms_per_step"] = round(host_enqueue / args.steps * 1e3, 3)
    if full_steps:
        line["kernels"] = [{"kernel": k, "launches_per_step": round(v[0] / full_steps, 1), "ms_per_step": round(v[1] / full_steps, 4),
                            **({"tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} if v[2] > 0 and v[1] > 0 else {})}
                           for k, v in rows if v[1] >= 0.02 * total_ms]
        line["kernels_note"] = (f"every launch of {full_steps} extra steps AFTER the timed region bracketed with HIP events (not part of ms_per_step; "
                                "the brackets stretch those steps, the kernel durations are the step's)")
        line["event_timed_ms_per_step"] = round(total_ms / full_steps, 3)
    if args.breakdown:
        for k, (cnt, kms, w) in rows:
            tf = f"{w / (kms * 1e-3) / 1e12:8.1f} TF/s" if w > 0 and kms > 0 else ""
            print(f"#   {k:70s} {cnt / full_steps:6.1f}/step {kms / full_steps:9.4f} ms/step {tf}", file=sys.stderr)
    if world > 1:
        line["n_ranks_seen"] = n_ranks_seen
    if not brief and not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline_secondary(name, wl)
    return line


def run_secondary(args, world, rank, device, dist):
    """--workload scalogram / conv_ar / attention: the same protocol and JSON line as the headline workload."""
    line = measure_secondary(args.workload, args, world, rank, device, dist)
    if line is not None:
        print(json.dumps(line), flush=True)


def secondary_block(args, device):
    """`secondary` of the default line: BASELINE configs[2] / configs[3] with the same W / K protocol, one after the other in this process
    (fresh engines; the allocator's cache is emptied in between).  A failure is reported in place, it never takes the headline down."""
    import gc
    out = {}
    for name in ("scalogram", "conv_ar", "attention"):
        gc.collect()
        torch.cuda.empty_cache()
        try:
            t0 = time.perf_counter()
            ln = measure_secondary(name, args, 1, 0, device, None, brief=True)
            cfg = ln["config"]
            out[name] = {"workload": cfg["workload"], "ms_per_step": ln["ms_per_step"], "ms_per_step_median": ln.get("ms_per_step_median"),
                         "ms_per_step_max": ln.get("ms_per_step_max"), "first_timed_step_ms": ln.get("first_timed_step_ms"),
                         "frames_per_s": ln["value"], "clips_per_s": cfg["clips_per_s"], "global_batch": cfg["global_batch"],
                         "clip_samples": cfg["clip_samples"], "loss_last_step": cfg["loss_last_step"],
                         "host_enqueue_ms_per_step": ln["host_enqueue_ms_per_step"],
                         "roofline": {k: ln["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches",
                                                                       "avg_launch_ms", "ms_per_step")} if "roofline" in ln else None,
                         "steps": ln["steps"], "warmup": ln["warmup"], "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:          # noqa: BLE001 -- reported, not raised
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    gc.collect()
    torch.cuda.empty_cache()
    return out


def launch_command(n, argv, port=None):
    """The driver's own launch line for N ranks on one node (rendezvous on 127.0.0.1, a free port unless one is given)."""
    if port is None:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` (this process has not
    touched the GPU and never will), pass its stderr through, relay rank 0's JSON line on stdout, return its exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(launch_command(n, sys.argv[1:]), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "scalogram", "conv_ar", "attention"],
                    help="cfg1 = BASELINE configs[1] (headline); scalogram = configs[2]; conv_ar / attention = configs[3]")
    ap.add_argument("--prewarm", type=int, default=0,
                    help="untimed steps BEFORE the W warm-up steps: the first dozen steps of a process run 6-8 %% slower (every kernel; the "
                         "clocks ramp under sustained load, docs/DESIGN_HISTORY_r1-r3.md section 9.3), which a 5-step warm-up does not cover; reported as "
                         "prewarm_steps.  0 = off")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: 256; 128 for --workload scalogram)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trainer-loop", action="store_true", help="skip the extra run through ContrastiveEstimationTrainer.train")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[2] / configs[3] workloads the default N = 1 run measures after the headline region")
    ap.add_argument("--no-score-gemm", action="store_true",
                    help="skip the stand-alone score-GEMM timings (profiling runs: their launches share the dominant kernel's symbol)")
    ap.add_argument("--breakdown", action="store_true", help="time every kernel (diagnostic run; not the headline number)")
    ap.add_argument("--graph", action="store_true", help="diagnostic: replay the step from a captured hipGraph (single GPU only)")
    ap.add_argument("--all-timesteps", action="store_true",
                    help="diagnostic: score_over_all_timesteps=True (the full (B*K)^2 score matrix); not the headline configuration")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 128 if args.workload == "scalogram" else 256
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))              # before any GPU call: the ranks are child processes, this one only relays

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rehearsal knobs (not used by the driver): several ranks on ONE card over gloo, to exercise the multi-rank code path
        # where only one GPU is available — CPC_BENCH_BACKEND=gloo CPC_BENCH_SHARE_DEVICE=1
        backend = os.environ.get("CPC_BENCH_BACKEND", "nccl")
        if os.environ.get("CPC_BENCH_SHARE_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from cpc_audio_amd import _hip
    from cpc_audio_amd.engine import FusedAdam, GradAllReduce, GraphedStep

    if args.workload != "cfg1":
        run_secondary(args, world, rank, device, dist)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    B, L, T = args.batch, 20480, 126
    model = build_model(args.dtype, device, seed=0)                 # identical parameters on every rank
    eng = model.engine(B, L)
    use_graph = args.graph and world == 1 and not args.breakdown
    opt = FusedAdam(model, lr=1e-4, device_step=use_graph)
    gen = torch.Generator().manual_seed(1000 + rank)                # rank r draws its own clips
    pool = [(torch.randn(B, L, generator=gen)).to(device) for _ in range(4)]

    sync = GradAllReduce(model, optimizer=opt) if world > 1 else None      # Adam follows each reduced piece of the gradient
    opt.skip_flag = eng.nan_flag()          # the trainer's device-side NaN guard is part of the measured step
    graphed = GraphedStep(eng, opt, True, 1.0, args.all_timesteps) if use_graph else None
    if graphed is None and os.environ.get("CPC_PREPARE_AHEAD", "1") != "0":
        opt.after_update = eng.prepare_ahead      # next step's operand copies rebuilt beside the rest of the backward pass; installed
        #                                           BEFORE the warm-up, as the trainer does: its buffers and side-stream launches are warm

    # event-timed kernels: the dominant MFMA kernel (`roofline`: the 256 x 256 bf16 NT GEMM in its three epilogue forms -- register epilogue
    # (conv forward), LDS-staged masked epilogue (data gradients) and the fused layer-2 data gradient / layer-1 weight gradient) and the
    # HBM-bound layer-1 forward (`hbm_kernel`)
    dom = DOMINANT if args.dtype == "bf16" else "gemm_nt<f32,f32,128>"
    dom_keys = [dom, dom.replace("gemm_nt<", "gemm_nt_conv1<")]
    dom_keys += [k + "@target_rows" for k in dom_keys]       # the side stream's row-range launches beside the GRU's backward recurrence
    timer = _hip.KernelTimer(only=None if args.breakdown else dom_keys + ["cpc_conv1_fwd"], by_shape=args.breakdown)
    timer.active = False
    _hip.set_timer(timer)
    # in sampled steps of the timed region (one in `every`, about five) the dominant launches are bracketed with HIP events: an event pair
    # costs ~12 us of idle queue per launch, 0.12 - 0.15 ms per bracketed step.  N > 1: the same steps also bracket GradAllReduce.finish().
    every = 1 if args.breakdown else max(1, args.steps // 5)
    exposed = []

    def step(i):
        if graphed is not None:
            return graphed(pool[i % len(pool)])
        out = eng.loss_and_grads(pool[i % len(pool)], softplus=True, regularization=1.0, all_timesteps=args.all_timesteps,
                                 grad_ready_hook=sync.hook if sync is not None else opt.hook,
                                 after_loss=sync.reduce_flag if sync is not None else None)
        if sync is not None:
            if timer.active:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                sync.finish()                   # RCCL all-reduce (sum) of the flat gradient buffer: overlapped pieces + the head
                e1.record()
                exposed.append((e0, e1))
            else:
                sync.finish()
        opt.step(grad_scale=1.0 / world)
        return out

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out, elapsed, host_enqueue, clock = timed_run(step, args.steps, args.prewarm + args.warmup, timer, every, fence)
    if sync is not None and args.prewarm + args.warmup:
        exposed = exposed[1:]                   # the first pair belongs to the sampled warm-up step
    _hip.set_timer(None)
    loss = float(out[0])
    step_stats = clock.stats()
    # The encoder's weight-gradient GEMMs run on a second stream beside the dominant kernel's launches (engine.py,
    # CPC_WGRAD_STREAM), so a launch's duration in the step includes the share of the chip it gives up.  A short extra pass AFTER
    # the timed region, with everything on one stream, gives the same kernel's figures alone (`roofline.alone`).
    alone, step_gemm_flops = None, 0.0
    if graphed is None and args.dtype == "bf16" and os.environ.get("CPC_WGRAD_STREAM", "1") != "0" and not args.breakdown:
        prev = os.environ.get("CPC_WGRAD_STREAM")
        os.environ["CPC_WGRAD_STREAM"] = "0"
        try:
            for i in range(3):
                step(i)
            t_alone = _hip.KernelTimer(only=dom_keys)
            _hip.set_timer(t_alone)
            t_alone.active = True
            for i in range(6):
                step(i)
            fence()
            _hip.set_timer(None)
            sa = t_alone.summary()
            # one more step with every launch booked: the FLOPs of ALL GEMM launches of a step (NT forward / data gradients, TN weight
            # gradients, GRU and predictor products) -> `whole_step` = those FLOPs over the timed region's ms_per_step
            t_all = _hip.KernelTimer(only=None)
            _hip.set_timer(t_all)
            step(0)
            fence()
            _hip.set_timer(None)
            step_gemm_flops = sum(v[2] for k, v in t_all.summary().items() if k.startswith("gemm_"))
            na = sum(v[0] for k, v in sa.items() if k in dom_keys)
            msa = sum(v[1] for k, v in sa.items() if k in dom_keys)
            fla = sum(v[2] for k, v in sa.items() if k in dom_keys)
            if na and msa > 0:
                alone = {"achieved": round(fla / (msa * 1e-3) / 1e12, 2), "avg_launch_ms": round(msa / na, 4), "launches": na}
        finally:
            if prev is None:
                os.environ.pop("CPC_WGRAD_STREAM", None)
            else:
                os.environ["CPC_WGRAD_STREAM"] = prev
    n_ranks_seen = 1
    dp = None
    if world > 1:
        mine = elapsed
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                     # how many ranks actually took part in the collectives of this run
        n_ranks_seen = int(ones.item())
        torch.cuda.synchronize()
        exp_ms = [a.elapsed_time(b) for a, b in exposed]
        # one row per rank: [ms_per_step (this rank's own clock between its fences), median step (HIP events), slowest step, host enqueue,
        #                    exposed all-reduce wait on the sampled steps]
        row = torch.tensor([mine / args.steps * 1e3, step_stats.get("ms_per_step_median", 0.0), step_stats.get("ms_per_step_max", 0.0),
                            host_enqueue / args.steps * 1e3, sum(exp_ms) / len(exp_ms) if exp_ms else -1.0], device=device, dtype=torch.float64)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        rows = [r.tolist() for r in rows]
        dp = data_parallel_report(sync, rows, len(exp_ms))

    summary = timer.summary()
    if rank == 0:
        frames = B * T * world * args.steps
        per_key = {k: summary[k] for k in dom_keys if k in summary} if not args.breakdown else \
            {k: v for k, v in summary.items() if k.split(" ")[0] in dom_keys}
        n = sum(v[0] for v in per_key.values())
        ms = sum(v[1] for v in per_key.values())
        flops = sum(v[2] for v in per_key.values())
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
        sampled = len([i for i in range(args.steps) if i % every == every // 2])
        # HBM-side bytes per launch of the dominant kernel: NOT a quantity of this run — PMC counters need their own rocprofv3
        # passes (MI355X_MICROARCH.md, HBM section); the committed summary of the latest such pass is quoted and labelled
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("gemm_nt_fast_bf16_256_hbm_bytes_per_launch")
                traffic_source = f"{tj.get('source')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs; not measured in this run)"
            except Exception:
                traffic = None
        line = {
            "metric": "CPC train-step audio frames/sec (enc+AR+InfoNCE)",
            "value": round(frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": args.prewarm,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            **step_stats,
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: AudioEncoder(5x512, strides 5/4/2/2/2) + AudioGRUModel(256) + 12 "
                                   "prediction steps, 100 visible, softplus score, reg 1.0, Adam; per-GPU batch "
                                   f"{B} x 20480 samples (126 frames/clip)",
                       "global_batch": B * world, "clip_samples": L, "parallelism": f"dp{world}",
                       "loss_last_step": round(loss, 6)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": ("gemm_nt_fast_kernel<bf16,bf16,2,4,8,4,...> (all three epilogue forms: conv forward, masked data gradients, "
                                    "fused layer-2 data gradient + layer-1 weight gradient)") if args.dtype == "bf16"
                         else "gemm_nt_fast_kernel<f32,f32,2,2,4,4>",
                         "launches": n, "avg_launch_ms": round(ms / n, 4) if n else None,
                         "algorithmic_gflop_per_launch": round(flops / n / 1e9, 3) if n else None,
                         "ms_per_step": round(ms / max(sampled, 1), 4),
                         "frac_of_step_time": round(ms / max(sampled, 1) / (elapsed / args.steps * 1e3), 3),
                         "symbols": {k: {"launches": v[0], "avg_launch_ms": round(v[1] / v[0], 4), "gflop_per_launch": round(v[2] / v[0] / 1e9, 2),
                                         "achieved": round(v[2] / (v[1] * 1e-3) / 1e12, 1), "frac": round(v[2] / (v[1] * 1e-3) / 1e12 / peak, 4)}
                                     for k, v in per_key.items() if v[0] and v[1] > 0}},
        }
        main_only = [v for k, v in per_key.items() if "@target_rows" not in k and v[0] and v[1] > 0]
        if len(main_only) < len([v for v in per_key.values() if v[0] and v[1] > 0]):
            mf, mm = sum(v[2] for v in main_only), sum(v[1] for v in main_only)
            line["roofline"]["main_stream_launches"] = {
                "achieved": round(mf / (mm * 1e-3) / 1e12, 2), "frac": round(mf / (mm * 1e-3) / 1e12 / peak, 4),
                "launches": sum(v[0] for v in main_only),
                "note": "without the @target_rows launches: the data gradients behind the target frames, a fraction of a round of tiles each, "
                        "issued on the side stream beside the GRU's backward recurrence (engine.CPCEngine._bwd_lane); achieved / frac above "
                        "include them"}
        if alone is not None:
            alone["frac"] = round(alone["achieved"] / peak, 4)
            line["roofline"]["alone"] = alone
            line["roofline"]["note"] = ("achieved / frac: the launches as they run in the timed step, beside the weight-gradient GEMMs on a "
                                        "second stream; alone: the same kernels with everything on one stream (6 untimed steps after the "
                                        "timed region, CPC_WGRAD_STREAM=0)")
        if step_gemm_flops > 0:
            tf = step_gemm_flops / (elapsed / args.steps) / 1e12
            line["whole_step"] = {"gemm_tflop_per_step": round(step_gemm_flops / 1e12, 3), "achieved": round(tf, 1), "unit": "TFLOP/s",
                                  "peak": peak, "frac": round(tf / peak, 4),
                                  "note": "executed FLOPs of every GEMM launch of one step (NT and TN forms) over ms_per_step: the matrix pipe's "
                                          "share of the whole step, GRU recurrence / layer 1 / loss / Adam time included in the denominator"}
        c1 = [v for k, v in summary.items() if k == "cpc_conv1_fwd"]
        if c1 and sum(v[1] for v in c1) > 0:
            # encoder layer 1 (C_in = 1): writes its [B][rows][512] output once (+ one sign bit per element), reads 4 B per input sample
            # (DESIGN.md section 3).  With the target lane (engine.encoder_forward) this is the main stream's launch over the rows the context
            # network needs (91 % of them); the side stream's launch over the target rows runs beside the GRU and is not counted here.
            esz = 2 if args.dtype == "bf16" else 4
            lane = eng._target_lane_rows() if hasattr(eng, "_target_lane_rows") else None
            rows = eng.geo.alloc[0] if lane is None else lane[0]
            nbytes = float(B) * rows * eng.channels[0] * esz + float(B) * min(eng.L_eff, rows * eng.strides[0] + eng.kernels[0]) * 4
            if eng.act_bits[0] is not None:                       # + the sign-bit mask of the output, one bit per element
                nbytes += float(B) * rows * eng.channels[0] / 8
            cn, cms = sum(v[0] for v in c1), sum(v[1] for v in c1)
            gbs = nbytes * cn / (cms * 1e-3) / 1e9
            line["hbm_kernel"] = {"kernel": "conv1_fwd_kernel", "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0,
                                  "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "launches": cn,
                                  "avg_launch_ms": round(cms / cn, 4), "algorithmic_bytes_per_launch": nbytes,
                                  "rows_per_item": int(rows), "rows_allocated": int(eng.geo.alloc[0])}
        if args.breakdown:
            rows = sorted(summary.items(), key=lambda kv: -kv[1][1])
            tot = sum(v[1] for _, v in rows)
            print(f"# kernel breakdown over {args.steps} steps (event-timed; sum {tot / args.steps:.3f} ms/step, "
                  f"wall {elapsed / args.steps * 1e3:.3f} ms/step)", file=sys.stderr)
            for k, (cnt, kms, w) in rows:
                tf = f"{w / (kms * 1e-3) / 1e12:8.1f} TF/s" if w > 0 and kms > 0 else ""
                print(f"#   {k:58s} {cnt / args.steps:6.1f}/step {kms / args.steps:9.4f} ms/step {tf}", file=sys.stderr)
        if world == 1 and args.dtype == "bf16" and not args.breakdown and not args.all_timesteps:
            if not args.no_score_gemm:
                line["score_gemm"] = score_gemm_figures(eng)
            if graphed is None and not args.no_trainer_loop:
                ms_t, n_logged = trainer_loop_ms(model, B, L, device)
                line["trainer_ms_per_step"] = ms_t
                line["trainer_loop"] = (f"ContrastiveEstimationTrainer.train, logger attached, device-resident dataset, "
                                        f"{n_logged} logged steps (first 10 discarded)")
        if world > 1:
            line["n_ranks_seen"] = n_ranks_seen
            line["data_parallel"] = dp
        if world == 1 and args.dtype == "bf16" and not args.no_secondary and not args.breakdown and not args.all_timesteps and graphed is None:
            # BASELINE configs[2] / configs[3] under the same W / K protocol, in this process, after the headline region
            del graphed, sync, opt, eng, model, pool
            line["secondary"] = secondary_block(args, device)
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed at N = 1 only
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def data_parallel_report(sync, rows, n_sampled):
    """`data_parallel` of an N > 1 line: one entry per rank (its own ms per step, median / slowest step by HIP events, host enqueue time,
    exposed all-reduce wait) and the bucket plan of engine.GradAllReduce (flat-gradient ranges in the order their reductions are issued)."""
    plan = sync.describe()
    return {"per_rank_ms_per_step": [round(r[0], 3) for r in rows],
            "per_rank_ms_per_step_median": [round(r[1], 3) for r in rows],
            "per_rank_ms_per_step_max": [round(r[2], 3) for r in rows],
            "per_rank_host_enqueue_ms_per_step": [round(r[3], 3) for r in rows],
            "allreduce_exposed_ms_per_step": [round(r[4], 4) for r in rows],
            "allreduce_exposed_note": (f"main-stream time of GradAllReduce.finish() on {n_sampled} sampled steps (HIP events): the wait for the last "
                                       "overlapped bucket + the reduction of the head of the buffer + Adam on the buckets finished there; "
                                       "everything else travels under the backward pass"),
            **plan}


if __name__ == "__main__":
    main()
