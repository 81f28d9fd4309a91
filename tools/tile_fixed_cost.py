"""What a 256 x 256 tile costs beside its K loop, per launch kind of the headline step (B = 256, layer-2 geometry: stride 4, kernel 8,
C_in = 512): the conv forward (register epilogue), the masked data gradient (LDS-staged epilogue, sign-bit mask, per-tile column sums) and
the data gradient fused with layer 1's weight gradient.  C_out sweeps the reduction length; a least-squares line through
(stages per tile, us per round of 256 tiles) gives the per-stage cost (slope) and the fixed cost of a tile (intercept).

    python tools/tile_fixed_cost.py [--iters 10]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--lout", type=int, default=1038)
a = ap.parse_args()
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
B, Lo, Ci, kw, s, kw1, s1 = a.B, a.lout, 512, 8, 4, 10, 5
Li = Lo * s
M = B * Lo
L = (Li - 1) * s1 + kw1 + 8


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
    return sorted(ts)[len(ts) // 2] * 1e3


def fit(pts):
    n = len(pts)
    sx, sy = sum(p[0] for p in pts), sum(p[1] for p in pts)
    sxx, sxy = sum(p[0] * p[0] for p in pts), sum(p[0] * p[1] for p in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return b, (sy - b * sx) / n


xw = torch.randn(B * L, device=dev)
rows = {"fwd": [], "dgrad": [], "dgrad+conv1": []}
for Co in (256, 512, 1024, 2048):
    guard = 16 * max(Ci, Co)
    x = torch.randn(guard + B * Li * Ci + guard, device=dev).to(bf)
    bits = torch.zeros(x.numel() // 8, device=dev, dtype=torch.uint8)
    _hip.call("cpc_sign_bits", P(x), P(bits), C.c_longlong(x.numel()), 1)
    y = torch.zeros(guard + B * Lo * Co + guard, device=dev, dtype=bf)
    dy = torch.randn(guard + B * Lo * Co + guard, device=dev).to(bf)
    dx = torch.zeros(guard + B * Li * Ci + guard, device=dev, dtype=bf)
    w = torch.randn(Co, Ci, kw, device=dev) * 0.05
    bias = torch.randn(Co, device=dev)
    wf = torch.empty(Co * kw * Ci, device=dev, dtype=bf)
    wd = torch.empty(s * Ci * 2 * Co, device=dev, dtype=bf)
    _hip.call("cpc_conv_w_prep", P(w), P(wf), P(wd), Co, Ci, kw, s, 1)
    cs = torch.zeros(int(_hip.lib().cpc_conv_dgrad_colsum_floats(B, Ci, s, Lo)), device=dev)
    sl = torch.empty(int(_hip.lib().cpc_conv_dgrad_conv1_floats(B, Ci, s, Lo, kw1, 0)), device=dev)
    t = timed(lambda: _hip.call("cpc_conv_fwd", P(x, guard), P(wf), P(bias), P(y, guard), B, Ci, Co, kw, s, Lo, Lo - 2, 1, C.c_longlong(guard), 1))
    tiles = -(-M // 256) * (Co // 256)
    rows["fwd"].append((kw * Ci // 64, t / (tiles / 256), t, 2.0 * M * Co * kw * Ci / t / 1e6))
    tiles = -(-M // 256) * (s * Ci // 256)
    fl = 2.0 * M * s * Ci * 2 * Co
    t = timed(lambda: _hip.call("cpc_conv_dgrad", P(dy, guard), P(wd), None, P(dx, guard), B, Ci, Co, kw, s, Lo, Li - 7, C.c_longlong(guard), 1,
                                P(bits, guard // 8), P(cs)))
    rows["dgrad"].append((2 * Co // 64, t / (tiles / 256), t, fl / t / 1e6))
    t = timed(lambda: _hip.call("cpc_conv_dgrad_conv1", P(dy, guard), P(wd), None, P(xw), P(sl), B, Ci, Co, kw, s, Lo, L, kw1, s1, Li - 7,
                                C.c_longlong(guard), 1, P(bits, guard // 8)))
    rows["dgrad+conv1"].append((2 * Co // 64, t / (tiles / 256), t, fl / t / 1e6))
    del x, bits, y, dy, dx, wf, wd
    torch.cuda.empty_cache()
for name, pts in rows.items():
    print(name)
    for st, per_round, t, tf in pts:
        print(f"   {st:4d} stages per tile: {t:8.1f} us per launch, {per_round:7.2f} us per round of tiles, {tf:7.1f} TF/s")
    if name == "fwd":           # (the forward's K is fixed by C_in: its tile count varies instead; no line to fit)
        continue
    b, c = fit([(p[0], p[1]) for p in pts])
    print(f"   per stage {b:.3f} us, fixed cost per tile {c:.2f} us")
