"""What do the sign-bit masks (include/cpc_hip.h, cpc_sign_bits) cost to make and save their consumers?  Layer shapes of
BASELINE config 2 (B = 256), each launch timed alone, interleaved rounds.

    python tools/bits_ab.py [--rounds 5] [--iters 10]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
B, Cc = 256, 512


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


def ab(name, fns):
    for f in fns.values():
        for _ in range(3):
            f()
    res = {k: [] for k in fns}
    for _ in range(a.rounds):
        for k, f in fns.items():
            res[k].append(timed(f))
    print(f"{name:22s} " + " | ".join(f"{k}: {sorted(v)[len(v) // 2]:8.1f} us" for k, v in res.items()), flush=True)


# layer 1: L = 20480, kernel 10, stride 5 -> 4095 valid positions, 3648 allocated (unused-frame skip) as in the engine
L, k1, s1, La1, Lv1 = 20480, 10, 5, 3648, 3648
x = torch.randn(B, L, device=dev)
w1 = torch.randn(Cc, 1, k1, device=dev) * 0.3
b1 = torch.randn(Cc, device=dev) * 0.1
guard = 16 * Cc
y1 = torch.zeros(2 * guard + B * La1 * Cc, device=dev, dtype=bf)
y1b = torch.zeros((2 * guard + B * La1 * Cc) // 8, device=dev, dtype=torch.uint8)
c1 = lambda bits: _hip.call("cpc_conv1_fwd", P(x), P(w1), P(b1), P(y1, guard), B, Cc, s1, k1, L, Lv1, La1, 1, _hip.BF16, P(y1b, guard // 8) if bits else None)
ab("layer-1 forward", {"plain": lambda: c1(False), "with bits": lambda: c1(True)})
mk = lambda t, tb: _hip.call("cpc_sign_bits", P(t), P(tb), C.c_longlong(t.numel()), _hip.BF16)
ab("layer-1 sign bits", {"cpc_sign_bits": lambda: mk(y1, y1b)})

# layers 2..4 forward (producers) and the data gradients that read their masks
for name, Lo, kw, s, xin, xbits in (("layer 2", 912, 8, 4, y1, y1b),):
    pass
LAY = [("layer 2", 912, 8, 4), ("layer 3", 456, 4, 2), ("layer 4", 228, 4, 2), ("layer 5", 114, 4, 2)]
prev, prevb = y1, y1b
for name, Lo, kw, s in LAY:
    w = torch.randn(Cc, Cc, kw, device=dev) * 0.05
    bias = torch.randn(Cc, device=dev) * 0.1
    D = -(-kw // s)
    wf = torch.empty(Cc * kw * Cc, device=dev, dtype=bf)
    wd = torch.empty(s * Cc * D * Cc, device=dev, dtype=bf)
    _hip.call("cpc_conv_w_prep", P(w), P(wf), P(wd), Cc, Cc, kw, s, _hip.BF16)
    y = torch.zeros(2 * guard + B * Lo * Cc, device=dev, dtype=bf)
    yb = torch.zeros((2 * guard + B * Lo * Cc) // 8, device=dev, dtype=torch.uint8)
    dy = torch.zeros(2 * guard + B * Lo * Cc, device=dev, dtype=bf)
    dy[guard:guard + B * Lo * Cc] = torch.randn(B * Lo * Cc, device=dev).to(bf)
    dx = torch.zeros(2 * guard + B * Lo * s * Cc, device=dev, dtype=bf)
    _hip.call("cpc_conv_fwd", P(prev, guard), P(wf), P(bias), P(y, guard), B, Cc, Cc, kw, s, Lo, Lo - 2, 1, C.c_longlong(guard), _hip.BF16)
    ab(f"{name} sign bits", {"cpc_sign_bits": lambda: mk(y, yb)})
    g = lambda bits: _hip.call("cpc_conv_dgrad", P(dy, guard), P(wd), None if bits else P(prev, guard), P(dx, guard), B, Cc, Cc, kw, s, Lo,
                               Lo * s, C.c_longlong(guard), _hip.BF16, P(prevb, guard // 8) if bits else None, None)
    ab(f"{name} data gradient", {"mask = activation": lambda: g(False), "mask = bits": lambda: g(True)})
    prev, prevb = y, yb
    del dy, dx
