"""Race screen for the multi-stream step: the same batch through loss_and_grads N times on one engine (no optimizer step) must give the
same loss and the same flat gradient, bit for bit, every time — the step has no atomics and a fixed summation order, so any difference is
a missing dependency between the streams (the target lanes, the weight-gradient stream, prepare_ahead).
    python tools/determinism_check.py [--reps 300] [--batch 256]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.audio_model import AudioEncoder, AudioGRUModel, AudioPredictiveCodingModel

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
dev, L = "cuda:0", 20480
torch.manual_seed(0)
model = AudioPredictiveCodingModel(AudioEncoder(), AudioGRUModel(512, 256), enc_size=512, ar_size=256, compute_dtype="bf16").to(dev)
xs = [(torch.randn(a.batch, L, generator=torch.Generator().manual_seed(s)) * 0.5).to(dev) for s in (1, 2)]
eng = model.engine(a.batch, L)
print("lanes:", eng._target_lane_rows() is not None, eng._bwd_lane() is not None)
ref, bad = {}, 0
for i in range(a.reps):
    k = i & 1
    out = eng.loss_and_grads(xs[k], softplus=True, regularization=1.0)
    sig = (float(out[0]), float(model._flat_grad.double().sum()), float(model._flat_grad.double().abs().sum()))
    g = model._flat_grad.clone()
    if k not in ref:
        ref[k] = (sig, g)
    elif not torch.equal(g, ref[k][1]):
        bad += 1
        d = (g - ref[k][1]).abs()
        print(f"rep {i}: gradient differs from the first pass of batch {k}: max abs {float(d.max()):.3e} at {int(d.argmax())}, loss {sig[0]} vs {ref[k][0][0]}")
torch.cuda.synchronize()
print(f"{a.reps} passes over two alternating batches of {a.batch}: {bad} differing passes")
sys.exit(1 if bad else 0)
