"""Train-step time of BASELINE configs[2]: CQT scalogram (256 bins, hop 128) + scalogram_resnet_architecture_7 +
context network, batch 128 x 97024 samples (the model's item_length), bf16 (first stage f32).

    python tools/scalogram_bench.py [--batch 128] [--steps 5] [--dtype bf16] [--breakdown] [--context gru|conv_ar_3]

Diagnostic; the headline number is bench.py's."""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cpc_audio_amd import _hip  # noqa: E402
from cpc_audio_amd.audio_model import AudioGRUModel, AudioPredictiveCodingModel, ConvolutionalArModel  # noqa: E402
from cpc_audio_amd.engine import FusedAdam  # noqa: E402
from cpc_audio_amd import configs  # noqa: E402
from cpc_audio_amd.scalogram_model import PreprocessingModule, ScalogramResidualEncoder, cqt_default_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--context", default="gru")
    ap.add_argument("--breakdown", action="store_true")
    ap.add_argument("--cqt", default="bf16x3", choices=["fp32", "bf16x3"])
    ap.add_argument("--gp", type=float, default=None, help="Wasserstein gradient penalty factor (linear scores, as the reference's penalty "
                                                              "experiments; needs --context conv_ar_3 with bf16 storage)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, V, K = args.batch, 60, 16
    torch.manual_seed(0)
    pre = PreprocessingModule(cqt_dict=cqt_default_dict, phase=True).to(dev)
    pre.cqt.precision = args.cqt
    enc = ScalogramResidualEncoder(args_dict=configs.fresh(configs.scalogram_resnet_architecture_7), preprocessing_module=pre)
    if args.context == "gru":
        ar = AudioGRUModel(512, 256)
    else:
        ar = ConvolutionalArModel(configs.fresh(configs.ar_conv_architecture_3))
    model = AudioPredictiveCodingModel(enc, ar, enc_size=512, ar_size=256, visible_steps=V, prediction_steps=K,
                                       compute_dtype=args.dtype).to(dev)
    if args.gp is not None:
        model.gradient_penalty_engine = True
    L = model.item_length
    print(f"item_length {L}  receptive_field {enc.receptive_field}  downsampling {enc.downsampling_factor}  "
          f"params {model.parameter_count()}", flush=True)
    g = torch.Generator().manual_seed(1)
    wave = (torch.randn(B, L, generator=g) * 0.1).to(dev)
    opt = None

    def step():
        nonlocal opt
        x = pre(wave.unsqueeze(1))
        eng = model.engine_for(x)
        if opt is None:
            opt = FusedAdam(model, lr=1e-4)
        if args.gp is not None:
            out = eng.loss_and_grads(x, softplus=False, regularization=0.0, all_timesteps=True, gradient_penalty=args.gp)
        else:
            out = eng.loss_and_grads(x, softplus=True, regularization=1.0)
        opt.step()
        return out, x

    for _ in range(2):
        out, x = step()
    torch.cuda.synchronize()
    print("scalogram", tuple(x.shape), "frames", model.engine_for(x).T, f"mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
    timer = None
    if args.breakdown:
        timer = _hip.KernelTimer(only=None, by_shape=True)
        _hip.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, x = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    _hip.set_timer(None)
    T = model.engine_for(x).T
    print(f"{args.context}: {ms:.3f} ms/step  {B * T / ms * 1e3:.0f} encoder frames/s  {B / ms * 1e3:.1f} clips/s  loss {float(out[0]):.5f}")
    if timer is not None:
        rows = sorted(timer.summary().items(), key=lambda kv: -kv[1][1])
        tot = sum(v[1] for _, v in rows)
        print(f"# event-timed kernels: {tot / args.steps:.3f} ms/step")
        for k, (cnt, kms, w) in rows[:40]:
            tf = f"{w / (kms * 1e-3) / 1e12:8.1f} TF/s" if w > 0 and kms > 0 else ""
            print(f"#   {k:70s} {cnt / args.steps:6.1f}/step {kms / args.steps:9.4f} ms/step {tf}")


if __name__ == "__main__":
    main()
