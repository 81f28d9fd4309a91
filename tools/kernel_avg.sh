#!/bin/bash
# Average duration of the kernels whose symbol matches a regex in one bench.py workload (rocprofv3 --kernel-trace --stats on the GPU box):
#   bash tools/kernel_avg.sh <tag> <workload> <regex> [ENV=val ...]      -> gpurun_out/<tag>/kernel_avg.txt
tag=$1; wl=$2; re=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/st -- python3 bench.py --workload $wl --steps 20 --warmup 10 --no-cpu-baseline > gpurun_out/$tag/log.txt 2>&1 || exit 1
F=$(find gpurun_out/$tag/st -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$re" > gpurun_out/$tag/kernel_avg.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:80]:80s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
cat gpurun_out/$tag/kernel_avg.txt
grep '^{' gpurun_out/$tag/log.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'loss', d['config']['loss_last_step'])"
