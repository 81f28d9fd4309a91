"""Attention forward and backward (bf16, head size 64): matrix-pipe kernel (default) against the vector kernel (CPC_ATTN_MFMA=0), B x heads = 256 x 8
problems of S = 60.    CPC_ATTN_MFMA=0 python tools/attn_ab.py ; python tools/attn_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip  # noqa: E402

B, S, C, heads = 256, 60, 512, 8
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
qkv = torch.randn(B * S, 3 * C, device=dev).to(bf)
out = torch.zeros(B * S, C, device=dev, dtype=bf)
Pm = torch.zeros(B * heads, S, S, device=dev, dtype=bf)
dout = torch.randn(B * S, C, device=dev).to(bf)
dq = torch.zeros(B * S, 3 * C, device=dev, dtype=bf)
fwd = lambda: _hip.call("cpc_attn_fwd", P(qkv), P(out), P(Pm), B, S, C, heads, 0.0, 0, 0, _hip.BF16)
bwd = lambda: _hip.call("cpc_attn_bwd", P(qkv), P(Pm), P(dout), P(dq), B, S, C, heads, 0.0, 0, 0, _hip.BF16)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return sorted(ts)[2]


print(f"CPC_ATTN_MFMA={os.environ.get('CPC_ATTN_MFMA', '1')}: forward {timed(fwd):.1f} us, backward {timed(bwd):.1f} us per launch")
