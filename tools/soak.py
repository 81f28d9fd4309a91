"""Longer run of the real trainer on a small device-resident synthetic set (white noise, so the loss falls only by memorising):
checks that the loss stays finite and decreases, and that device memory does not grow.  python tools/soak.py [--steps 600]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpc_audio_amd.audio_dataset import SyntheticAudioDataset  # noqa: E402
from cpc_audio_amd.audio_model import AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel  # noqa: E402
from cpc_audio_amd.contrastive_estimation_training import ContrastiveEstimationTrainer  # noqa: E402


class Meter:
    def __init__(self):
        self.values = []

    def update(self, v):
        self.values.append(float(v))


class Logger:
    def __init__(self):
        self.loss_meter, self.score_meter, self.marks = Meter(), Meter(), []

    def log(self, step):
        self.marks.append(time.perf_counter())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--scalogram", action="store_true", help="BASELINE configs[2]: CQT + scalogram_resnet_architecture_7 + ar_conv_architecture_3 "
                                                              "through the trainer's preprocessing hook (use --batch 128)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pre = None
    if args.scalogram:
        from cpc_audio_amd import configs
        from cpc_audio_amd.audio_model import ConvolutionalArModel
        from cpc_audio_amd.scalogram_model import PreprocessingModule, ScalogramResidualEncoder, cqt_default_dict
        pre = PreprocessingModule(cqt_dict=cqt_default_dict, phase=True).to(dev)
        pre.cqt.precision = "bf16x3"
        enc = ScalogramResidualEncoder(args_dict=configs.fresh(configs.scalogram_resnet_architecture_7), preprocessing_module=pre)
        model = AudioPredictiveCodingModel(enc, ConvolutionalArModel(configs.fresh(configs.ar_conv_architecture_3)), enc_size=512, ar_size=256,
                                           visible_steps=60, prediction_steps=16, compute_dtype="bf16").to(dev)
        ds = SyntheticAudioDataset(2 * args.batch, model.item_length, seed=3, scale=0.1, device=dev)
    else:
        model = AudioPredictiveCodingModel(AudioEncoder(), AudioGRUModel(512, 256), enc_size=512, ar_size=256, compute_dtype="bf16").to(dev)
        ds = SyntheticAudioDataset(512, model.item_length, seed=3, scale=0.5, device=dev)
    logger = Logger()
    tr = ContrastiveEstimationTrainer(model=model, dataset=ds, logger=logger, device=dev, regularization=1.0, preprocessing=pre,
                                      prediction_steps=16 if args.scalogram else 12)
    tr.verbose = False
    t0 = time.perf_counter()
    tr.train(batch_size=args.batch, epochs=10 ** 6, lr=1e-4, num_workers=0, max_steps=args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    v = logger.loss_meter.values
    m = logger.marks
    if len(m) > 120:          # steady state: spacing of the log calls over the last steps (the logger sees step i one step late)
        tail = m[100:]
        print(f"steady state (steps 100 .. {len(m)}): {(tail[-1] - tail[0]) / (len(tail) - 1) * 1e3:.3f} ms/step")
    print(f"{len(v)} steps in {dt:.2f} s ({dt / len(v) * 1e3:.2f} ms/step incl. host); loss first/last 10 mean "
          f"{sum(v[:10]) / 10:.4f} -> {sum(v[-10:]) / 10:.4f}; finite {all(x == x and abs(x) < 1e30 for x in v)}; "
          f"max memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
