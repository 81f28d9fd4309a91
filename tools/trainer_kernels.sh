cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4bn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4bn/st -- python3 tools/soak.py --scalogram --batch 128 --steps 150 > gpurun_out/r4bn/log.txt 2>&1
F=$(find gpurun_out/r4bn/st -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (ms):", tot / 150 / 1e6)
for r in rows[:60]:
    n = r["Name"]
    if "at::native" in n or "Memcpy" in n or "copy" in n.lower() or "index" in n.lower() or "foreach" in n.lower() or "elementwise" in n:
        print(f"{n[:110]:110s} calls/step {int(r['Calls'])/150:6.1f} us/step {float(r['TotalDurationNs'])/150/1e3:8.1f}")
PY
find gpurun_out/r4bn -name "*kernel_trace.csv" -delete
tail -2 gpurun_out/r4bn/log.txt
