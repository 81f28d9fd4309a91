"""Global-negatives score GEMM (24 576 x 512 x 24 576, bf16 operands): 256- vs 128-wide tiles, register vs LDS-staged epilogue, f32 and bf16 output.
    python tools/score_gemm_ab.py      (the first line measured includes the clock ramp)"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cpc_audio_amd import _hip
dev = torch.device("cuda:0")
R, E = 24576, 512
g = torch.Generator().manual_seed(5)
a = (torch.randn(R, E, generator=g) * 0.5).to(dev).bfloat16()
b = (torch.randn(R, E, generator=g) * 0.5).to(dev).bfloat16()
for name, flags, odt in (("f32", _hip.GEMM_OUT_F32, torch.float32), ("bf16", 0, torch.bfloat16)):
    c = torch.empty(R * R, device=dev, dtype=odt)
    for extra_name, extra in (("256", 0), ("128", _hip.GEMM_SMALL_TILE), ("256 lds-epi", _hip.GEMM_NO_PERS), ("128 lds-epi", _hip.GEMM_NO_PERS | _hip.GEMM_SMALL_TILE)):
        run = lambda: _hip.gemm_nt(_hip.ptr(a), _hip.ptr(b), _hip.ptr(c), R, R, E, E, E, R, _hip.BF16, flags=flags | extra)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:5s} {extra_name:12s} {ms:.4f} ms  {2.0*R*R*E/(ms*1e-3)/1e12:7.1f} TF/s", flush=True)
    del c
