"""Train-step time of BASELINE configs[3] variants (context network swapped for the GRU) at batch 256 x 20480 samples.

    python tools/context_bench.py [--batch 256] [--steps 10] [--dtype bf16]

Prints one line per context network: ms/step, frames/s, loss (diagnostic; the headline number is bench.py's)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cpc_audio_amd.audio_model import AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel, ConvolutionalArModel  # noqa: E402
from cpc_audio_amd.attention_model import AttentionModel  # noqa: E402
from cpc_audio_amd.engine import FusedAdam  # noqa: E402


def contexts():
    from cpc_audio_amd import configs
    conv = dict(configs.fresh(configs.ar_conv_default_dict), channel_count=[512, 256, 256, 256])      # enc_size 512 in front
    att = dict(configs.fresh(configs.attention_architecture_1), dropout=0.0)
    return {"gru_v100": (lambda: AudioGRUModel(512, 256), 100), "gru_v60": (lambda: AudioGRUModel(512, 256), 60),
            "conv_ar_default": (lambda: ConvolutionalArModel(conv), 60), "attention_architecture_1": (lambda: AttentionModel(att), 60)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default=None)
    ap.add_argument("--breakdown", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, K = args.batch, 20480, 12
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, generator=g).to(dev)
    for name, (make, V) in contexts().items():
        if args.only and args.only != name:
            continue
        torch.manual_seed(0)
        model = AudioPredictiveCodingModel(AudioEncoder(), make(), enc_size=512, ar_size=256, visible_steps=V, prediction_steps=K,
                                           compute_dtype=args.dtype).to(dev)
        eng = model.engine(B, L)
        opt = FusedAdam(model, lr=1e-4)
        for _ in range(10):          # the first steps after an engine is built run slower (queues, allocator): keep them out
            out = eng.loss_and_grads(x, softplus=True, regularization=1.0)
            opt.step()
        torch.cuda.synchronize()
        timer = None
        if args.breakdown:
            from cpc_audio_amd import _hip
            timer = _hip.KernelTimer(only=None, by_shape=False)
            _hip.set_timer(timer)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng.loss_and_grads(x, softplus=True, regularization=1.0)
            opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        if timer is not None:
            _hip.set_timer(None)
            for k, (cnt, kms, w) in sorted(timer.summary().items(), key=lambda kv: -kv[1][1])[:14]:
                print(f"#   {k:50s} {cnt / args.steps:6.1f}/step {kms / args.steps:9.4f} ms/step")
        print(f"{name:28s} V={V:3d} {ms:8.3f} ms/step  {B * 126 / ms * 1e3:12.0f} frames/s  loss {float(out[0]):.5f}  "
              f"mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
        del model, eng, opt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
