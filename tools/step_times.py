"""Per-step wall time (HIP events) of the first 60 train steps of a fresh process: shows the ramp after a cold start.
    python tools/step_times.py [n_prewarm_forward_passes]"""
import sys, os, time, torch
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from cpc_audio_amd.engine import FusedAdam
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
eng = model.engine(256, 20480)
opt = FusedAdam(model, lr=1e-4)
opt.skip_flag = eng.nan_flag()
opt.after_update = eng.prepare_ahead
pool = [torch.randn(256, 20480).to(dev) for _ in range(4)]
if len(sys.argv) > 1:                       # pre-warm: N launches of the engine's forward (no loss, no update) before step 0
    eng.prepare_weights()
    for _ in range(int(sys.argv[1])):
        eng.encoder_forward(pool[0])
    torch.cuda.synchronize()
evs = []
idle_at = int(os.environ.get("IDLE_AT", "-1"))
for i in range(60):
    if i == idle_at:
        torch.cuda.synchronize(); time.sleep(1.0)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    eng.loss_and_grads(pool[i % 4], softplus=True, regularization=1.0, grad_ready_hook=opt.hook)
    opt.step()
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(60)]
print("per-step ms:", " ".join(f"{t:.2f}" for t in ts))
