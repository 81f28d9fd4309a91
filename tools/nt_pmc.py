"""Workload for SQ counter passes over the NT GEMM variants (plain rows and overlapped rows, K = 2048):

    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/nt_pmc.py
    python tools/nt_pmc.py --summarize <dir> [<dir> ...]      # per kernel symbol: mean of every counter found

Launches: the 256x256 kernel on overlapped rows and its probes (no DMA / no MFMA / B requests only)."""
import argparse, collections, csv, glob, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--summarize", nargs="*")
a = ap.parse_args()
if a.summarize:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.summarize:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        if "gemm_nt" not in k:
            continue
        print(k[:150])
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
    sys.exit(0)
import os
os.environ.setdefault("CPC_ENABLE_PROBES", "1")      # this tool IS a timing probe (see cpc_debug_set in include/cpc_hip.h)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _hip
dev, bf, P = "cuda:0", torch.bfloat16, _hip.ptr
C = 512
M, N, K, lda = 256 * 128 * 4, 512, 2048, 1024
x = torch.relu(torch.randn(M * lda + K + 64, device=dev)).to(bf)
w = (torch.randn(N * K, device=dev) * 0.05).to(bf)
y = torch.zeros(M * N, device=dev, dtype=bf)


def run(flags=0, probe=0):
    _hip.lib().cpc_debug_set(4, probe)
    for _ in range(3):
        _hip.gemm_nt(P(x), P(w), P(y), M, N, K, lda, K, N, 1, flags=flags)
    _hip.lib().cpc_debug_set(4, 0)
    torch.cuda.synchronize()


run(0)
run(0, 1)
run(0, 2)
run(0, 16)
