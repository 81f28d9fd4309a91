"""A/B of write-through output stores in the NT fast kernels (cpc_debug_set key 6): step time with 0 (plain), 1 (sc1), 2 (sc0 sc1).
Run on the GPU box:  python tools/wt_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cpc_audio_amd import _hip  # noqa: E402
from cpc_audio_amd.engine import FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev, seed=0)
eng = model.engine(256, 20480)
opt = FusedAdam(model, lr=1e-4)
opt.skip_flag = eng.nan_flag()
opt.after_update = eng.prepare_ahead
x = torch.randn(256, 20480, generator=torch.Generator().manual_seed(5)).to(dev)


def run(n):
    for i in range(n):
        out = eng.loss_and_grads(x, softplus=True, regularization=1.0, grad_ready_hook=opt.hook)
        opt.step()
    return out


run(40)
for rep in range(2):
    for mode in (0, 1, 2, 0):
        _hip.lib().cpc_debug_set(6, mode)
        run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(40)
        torch.cuda.synchronize()
        print(f"write-through mode {mode}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step  loss {float(out[0]):.6f}", flush=True)
_hip.lib().cpc_debug_set(6, 0)
