#!/bin/bash
# On the GPU box: the write-request counters of the L2's memory side for every kernel of tools/store_probe.hip (fill-like vs the layer-1 mapping):
# TCC_EA0_WRREQ (all write requests) and TCC_EA0_WRREQ_64B (the 64-byte ones), one counter pass.   -> gpurun_out/store_probe_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/store_probe_pmc
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe 2>/dev/null
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc -- /tmp/store_probe > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, v in acc.items():
        a = v.get("TCC_EA0_WRREQ_sum", [0]); b = v.get("TCC_EA0_WRREQ_64B_sum", [0])
        line = f"{k[:80]:80s} launches {len(a):4d}  WRREQ {sum(a)/len(a):14.0f}  WRREQ_64B {sum(b)/len(b):14.0f}  bytes/WRREQ at 956 MB: {956301312.0/(sum(a)/len(a)):6.1f}"
        print(line); fh.write(line + "\n")
PY
rm -rf $OUT/pmc
