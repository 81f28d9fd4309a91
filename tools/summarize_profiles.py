"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the tracked summaries under profiles/.

    python tools/summarize_profiles.py r01 [workload]        (workload: cfg1 = default | scalogram | conv_ar | attention)

Writes profiles/<tag>_kernel_stats.csv (copy of the --stats summary), profiles/<tag>_traffic.json (per-kernel HBM bytes
per launch from the FETCH_SIZE / WRITE_SIZE passes) and profiles/traffic_latest.json (read by bench.py).
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports exactly half of the bytes of a 16-byte-
per-lane streaming read (MI355X_MICROARCH.md, HBM section) — calibrated here on conv1_bwd_kernel, whose only large read
is the 0.956 GB layer-1 gradient: FETCH_SIZE reads 0.488 GB for it."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "cfg1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}" + ("" if workload == "cfg1" else f"_{workload}"))
if workload != "cfg1":
    tag = f"{tag}_{workload}"
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """gpurun merges every call's files into the same local directory: take the most recent run's."""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_bench_b256_bf16.csv" if workload == "cfg1" else f"{tag}_kernel_stats.csv"))
try:          # one step of the timed region, kernel by kernel
    shutil.copy(os.path.join(src, "timeline_one_step.txt"), os.path.join(dst, f"{tag}_timeline_one_step.txt"))
except FileNotFoundError:
    pass
try:          # the JSON line bench.py printed under the profiler (HIP-event figures of the same run)
    line = [ln for ln in open(os.path.join(src, "stats.log")) if ln.startswith("{")][-1]
    open(os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"), "w").write(line)
except (IndexError, FileNotFoundError):
    pass


def per_kernel(counter, sub):
    f = newest(os.path.join(src, sub, "**", "*counter_collection.csv"))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel("FETCH_SIZE", "fetch"), per_kernel("WRITE_SIZE", "write")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
    out[k] = {"launches_sampled": len(f), "fetch_size_kb_avg": sum(f) / len(f), "write_size_kb_avg": sum(w) / len(w),
              "hbm_bytes_per_launch": (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024}
json.dump(out, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)

# matrix-pipe occupancy per kernel (the pass may be absent in older runs): SQ_VALU_MFMA_BUSY_CYCLES summed over the chip's 1024
# SIMDs against GRBM_GUI_ACTIVE (summed over the 8 XCDs) -> fraction of SIMD-cycles with the matrix pipe busy
try:
    busy, act = per_kernel("SQ_VALU_MFMA_BUSY_CYCLES", "mfma"), per_kernel("GRBM_GUI_ACTIVE", "mfma")
    mf = {}
    for k in sorted(busy):
        b, a = busy[k], act.get(k, [0.0])
        cyc = sum(a) / len(a) / 8.0            # cycles per dispatch (average over the XCDs)
        mf[k] = {"launches_sampled": len(b), "mfma_busy_cycles_avg": sum(b) / len(b), "gui_active_avg": sum(a) / len(a),
                 "mfma_busy_frac_of_simd_cycles": (sum(b) / len(b)) / (cyc * 1024.0) if cyc > 0 else None}
    json.dump(mf, open(os.path.join(dst, f"{tag}_mfma.json"), "w"), indent=1)
except (ValueError, FileNotFoundError) as e:
    print("no mfma pass:", e)
if workload != "cfg1":
    top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:6]
    for k, v in top:
        print(f"{k[:110]:110s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch x {v['launches_sampled']}")
    sys.exit(0)
# the dominant kernel of bench.py's roofline object = the 256x256 bf16 -> bf16 NT GEMM in both epilogue forms (register epilogue:
# conv forward; LDS-staged: data gradients), without the fused layer-1 variant (template flag C1, "...Lb1ELb1E...")
dom = [k for k in out if "gemm_nt_fast_kernel" in k and "Li2ELi4ELi8ELi4E" in k and "DF16bDF16b" in k and "Li8ELi4ELb1ELb1E" not in k]
n_dom = sum(out[k]["launches_sampled"] for k in dom)
latest = {"source": f"profiles/{tag}_traffic.json", "kernels": dom,
          "gemm_nt_fast_bf16_256_hbm_bytes_per_launch": (sum(out[k]["hbm_bytes_per_launch"] * out[k]["launches_sampled"] for k in dom) / n_dom)
          if n_dom else None}
json.dump(latest, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(latest, indent=1))
