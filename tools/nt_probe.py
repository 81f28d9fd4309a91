"""Timing probes of the K loop of the 256 x 256 NT kernel (register epilogue): what does a 64-deep stage cost without its LDS-DMA
requests, without its MFMAs, with only the B tile requested?  Plain-row GEMMs,
M = 7 rounds of tiles, N = 512; the slope between two K gives the per-stage cost of each variant.  Results of the probes are garbage
(cpc_debug_set key 4); the default launch is bit-exact.

    python tools/nt_probe.py [--K 2048,8192]"""
import argparse, os, sys
import os
os.environ.setdefault("CPC_ENABLE_PROBES", "1")      # this tool IS a timing probe (see cpc_debug_set in include/cpc_hip.h)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=lambda t: [int(v) for v in t.split(",")], default=[2048, 8192])
ap.add_argument("--N", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
a = ap.parse_args()
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
N = a.N
M = 256 * 128 * 7 * 512 // N          # 7 rounds of 256 tiles
out = torch.zeros(M * N, device=dev, dtype=bf)
NAMES = {0: "full", 1: "no DMA", 2: "no MFMA", 16: "B requests only", 64: "B stage-major"}


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.iters)
    return sorted(ts)[len(ts) // 2]


res = {}
for K in a.K:
    A = torch.randn(M * K, device=dev).to(bf)
    Bt = torch.randn(N * K, device=dev).to(bf)
    line = f"K={K:5d} us per round of tiles:"
    for probe, name in NAMES.items():
        _hip.lib().cpc_debug_set(4, probe)
        t = timed(lambda: _hip.gemm_nt(P(A), P(Bt), P(out), M, N, K, K, K, N, 1)) * 1e3 / 7
        res[(K, probe)] = t
        line += f"  {name}: {t:7.2f}"
    # the same product with the A operand stored stage-major: [K / 64][M][64], a stage's A tile = one dense 32 KiB range
    Ad = torch.randn((K // 64) * M * 64 + 64, device=dev).to(bf)
    _hip.lib().cpc_debug_set(4, 32)
    t = timed(lambda: _hip.gemm_nt(P(Ad), P(Bt), P(out), M, N, K, 64, K, N, 1, a_item=M * 64)) * 1e3 / 7
    res[(K, 32)] = t
    line += f"  A stage-major: {t:7.2f}"
    del Ad
    _hip.lib().cpc_debug_set(4, 0)
    print(line, flush=True)
    del A, Bt
if len(a.K) >= 2:
    k0, k1 = a.K[0], a.K[-1]
    names = dict(NAMES)
    names[32] = "A stage-major"
    print("per 64-deep stage: " + "  ".join(f"{name}: {(res[(k1, p)] - res[(k0, p)]) / ((k1 - k0) / 64):.3f} us" for p, name in names.items()))

# Conv shapes of BASELINE config 2 with the activation operand chunk-major ([C / 64][frames][64]) against the row-major one
# (overlapped rows): forward of layer 3 (kernel 4, stride 2: 4 taps, rows 2 frames apart) and data gradient of layer 3 (2 taps,
# rows 1 frame apart).  Results of the chunk-major launches are not checked (probe).
Cc = 512
for name, Mrows, Nn, taps, rstep in (("layer-3 forward", 256 * 456, 512, 4, 2), ("layer-3 data gradient", 256 * 456, 1024, 2, 1)):
    K = taps * Cc
    lda = rstep * Cc
    frames = Mrows * rstep + taps + 16
    A = torch.relu(torch.randn(frames * Cc + 64, device=dev)).to(bf)
    Bt = (torch.randn(Nn * K, device=dev) * 0.05).to(bf)
    o2 = torch.zeros(Mrows * Nn, device=dev, dtype=bf)
    t_row = timed(lambda: _hip.gemm_nt(P(A), P(Bt), P(o2), Mrows, Nn, K, lda, K, Nn, 1, flags=_hip.GEMM_RELU)) * 1e3
    plane = frames * 64
    Ad = torch.relu(torch.randn((Cc // 64) * plane + 64, device=dev)).to(bf)
    _hip.lib().cpc_debug_set(4, 32)
    _hip.lib().cpc_debug_set(5, taps)
    t_chk = timed(lambda: _hip.gemm_nt(P(Ad), P(Bt), P(o2), Mrows, Nn, K, rstep * 64, K, Nn, 1, a_item=plane, flags=_hip.GEMM_RELU)) * 1e3
    _hip.lib().cpc_debug_set(4, 0)
    _hip.lib().cpc_debug_set(5, 1)
    print(f"{name}: row-major activations {t_row:7.1f} us, chunk-major {t_chk:7.1f} us", flush=True)
    del A, Ad, Bt, o2
