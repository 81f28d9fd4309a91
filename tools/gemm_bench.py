"""Micro-benchmark of the conv GEMM shapes of BASELINE config 2 (B=256): conv fwd / dgrad / wgrad of layer 2 (or --layer).
Usage: python tools/gemm_bench.py [--iters 20] [--which fwd,dgrad,wgrad] [--flags 0]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--which", default="fwd,dgrad,wgrad")
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--lout", type=int, default=912)
ap.add_argument("--cin", type=int, default=512)
ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--kw", type=int, default=8)
ap.add_argument("--stride", type=int, default=4)
ap.add_argument("--nsplit", type=int, default=8)
a = ap.parse_args()
dev = "cuda:0"
B, Lo, Ci, Co, kw, s = a.B, a.lout, a.cin, a.cout, a.kw, a.stride
Li = Lo * s
guard = 16 * max(Ci, Co)
bf = torch.bfloat16
x = torch.randn(guard + B * Li * Ci + guard, device=dev).to(bf)
y = torch.zeros(guard + B * Lo * Co + guard, device=dev, dtype=bf)
dy = torch.randn(guard + B * Lo * Co + guard, device=dev).to(bf)
dx = torch.zeros(guard + B * Li * Ci + guard, device=dev, dtype=bf)
w = torch.randn(Co, Ci, kw, device=dev) * 0.05
bias = torch.randn(Co, device=dev)
wf = torch.empty(Co * kw * Ci, device=dev, dtype=bf)
wd = torch.empty(s * Ci * 2 * Co, device=dev, dtype=bf)
_hip.call("cpc_conv_w_prep", _hip.ptr(w), _hip.ptr(wf), _hip.ptr(wd), Co, Ci, kw, s, 1)
slabs = torch.empty(a.nsplit * kw * Ci * Co, device=dev)
P = _hip.ptr
M = B * Lo

def fwd():
    _hip.gemm_nt(P(x, guard), P(wf), P(y, guard), M, Co, kw * Ci, s * Ci, kw * Ci, Co, 1, bias=P(bias), c_rpi=Lo, c_item=Lo * Co,
                 c_valid=Lo - 2, flags=_hip.GEMM_RELU | a.flags)
def dgrad():
    D = 2
    _hip.gemm_nt(P(dy, guard - (D - 1) * Co), P(wd), P(dx, guard), M, s * Ci, D * Co, Co, D * Co, s * Ci, 1, mask=P(x, guard), flags=a.flags)
def wgrad():
    chunk = ((M + a.nsplit - 1) // a.nsplit + 63) // 64 * 64
    _hip.gemm_tn(P(x, guard), P(dy, guard), P(slabs), M, kw * Ci, Co, s * Ci, Co, Co, 1, nsplit=a.nsplit, m_chunk=chunk,
                 slab_stride=kw * Ci * Co, flags=_hip.GEMM_OUT_F32 | a.flags)

fl = 2.0 * M * Co * kw * Ci
for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
    if name not in a.which.split(","):
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:6s} M={M} {ms:8.4f} ms  {fl / ms / 1e9:8.1f} TF/s  flags={a.flags}")
