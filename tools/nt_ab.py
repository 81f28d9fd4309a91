"""A/B of gemm_nt variants on the conv GEMM shapes of BASELINE config 2 (B = 256), interleaved rounds in ONE process
(cdna_hip_programming.md rule 24), plus a fit of the per-tile fixed cost (time = rounds * (a * K/64 + f)).

    python tools/nt_ab.py [--rounds 5] [--iters 10] [--fit]

Variants: default (register epilogue, tap-innermost K order for overlapped rows) vs CPC_GEMM_NO_PERS (LDS-staged epilogue)
vs additionally CPC_GEMM_LINEAR_K (storage order).  Operands are post-ReLU
like the real activations (half zeros) for the forward shapes and dense random for the gradients."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--fit", action="store_true")
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--stagger", type=lambda t: [int(v) for v in t.split(",") if v], default=[])
a = ap.parse_args()
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
B, C = a.B, 512
# (name, Lout_alloc, kw, stride) of encoder layers 2..5 at L = 20480 with the unused-frame skip
LAYERS = [("L2", 912, 8, 4), ("L3", 456, 4, 2), ("L4", 228, 4, 2), ("L5", 114, 4, 2)]
guard = 16 * C


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def ab(name, fns, flops):
    """fns: {label: callable}; interleaved rounds, prints median / min per label."""
    for fn in fns.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(a.rounds):
        for k, fn in fns.items():
            res[k].append(timed(fn, a.iters))
    out = []
    for k, v in res.items():
        v = sorted(v)
        med = v[len(v) // 2]
        out.append(f"{k}: med {med * 1e3:8.1f} us {flops / med / 1e9:7.1f} TF/s (min {v[0] * 1e3:8.1f})")
    print(f"{name:14s} " + " | ".join(out), flush=True)


big = max(l[1] * l[3] for l in LAYERS)
x = torch.relu(torch.randn(guard + B * big * C + guard, device=dev)).to(bf)
xo = torch.zeros(guard + B * big * C + guard, device=dev, dtype=bf)
for name, Lo, kw, s in LAYERS:
    M = B * Lo
    w = torch.randn(C, C, kw, device=dev) * 0.05
    bias = torch.randn(C, device=dev)
    wf = torch.empty(C * kw * C, device=dev, dtype=bf)
    D = -(-kw // s)
    wd = torch.empty(s * C * D * C, device=dev, dtype=bf)
    _hip.call("cpc_conv_w_prep", P(w), P(wf), P(wd), C, C, kw, s, 1)
    y = torch.zeros(guard + M * C + guard, device=dev, dtype=bf)
    dy = torch.randn(guard + M * C + guard, device=dev).to(bf)

    def fwd(flags):
        _hip.gemm_nt(P(x, guard), P(wf), P(y, guard), M, C, kw * C, s * C, kw * C, C, 1, bias=P(bias), c_rpi=Lo, c_item=Lo * C,
                     c_valid=Lo - 2, flags=_hip.GEMM_RELU | flags)

    def dgrad(flags):
        _hip.gemm_nt(P(dy, guard - (D - 1) * C), P(wd), P(xo, guard), M, s * C, D * C, C, D * C, s * C, 1, mask=P(x, guard), flags=flags)

    def stag(fn, v):
        def run():
            _hip.lib().cpc_debug_set(1, v)
            fn(0)
            _hip.lib().cpc_debug_set(1, 0)
        return run

    fv = {"default": lambda: fwd(0), "lds": lambda: fwd(_hip.GEMM_NO_PERS), "lds-lin": lambda: fwd(_hip.GEMM_NO_PERS | _hip.GEMM_LINEAR_K)}
    dv = {"default": lambda: dgrad(0), "lds": lambda: dgrad(_hip.GEMM_NO_PERS), "lds-lin": lambda: dgrad(_hip.GEMM_NO_PERS | _hip.GEMM_LINEAR_K)}
    for v in a.stagger:
        fv[f"st{v}"] = stag(fwd, v)
        dv[f"st{v}"] = stag(dgrad, v)
    ab(f"{name} fwd", fv, 2.0 * M * C * kw * C)
    ab(f"{name} dgrad", dv, 2.0 * M * s * C * D * C)
    del y, dy

if a.fit:
    # per-tile fixed cost: 7 full rounds of 256 x 256 tiles at N = 512 (two N tiles share an A panel, as in the conv forward),
    # plain rows (lda = K), K = 512 .. 8192
    M, N = 256 * 128 * 7, 512
    out = torch.zeros(M * N, device=dev, dtype=bf)
    msk = torch.relu(torch.randn(M * N, device=dev)).to(bf)
    pts = []
    for K in (512, 1024, 2048, 4096, 8192):
        A = torch.randn(M * K, device=dev).to(bf)
        Bt = torch.randn(N * K, device=dev).to(bf)
        for label, kw_ in (("plain", {}), ("mask", {"mask": P(msk)}), ("plain-lds", {"flags": _hip.GEMM_NO_PERS}),
                           ("mask-lds", {"mask": P(msk), "flags": _hip.GEMM_NO_PERS})):
            f = lambda: _hip.gemm_nt(P(A), P(Bt), P(out), M, N, K, K, K, N, 1, **kw_)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t = sorted(timed(f, a.iters) for _ in range(a.rounds))[a.rounds // 2]
            pts.append((label, K, t))
            print(f"fit {label:10s} K={K:5d}: {t * 1e3:8.1f} us = {t * 1e3 / 7:7.2f} us per round, {2.0 * M * N * K / t / 1e9:7.1f} TF/s", flush=True)
        del A, Bt
    for label in ("plain", "mask", "plain-lds", "mask-lds"):
        p = [(k, t * 1e3 / 7) for l, k, t in pts if l == label]
        (k0, t0), (k1, t1) = p[0], p[-1]
        slope = (t1 - t0) / ((k1 - k0) / 64)
        print(f"{label}: {slope:.3f} us per 64-deep stage, fixed {t0 - slope * k0 / 64:.2f} us per tile")
