#!/bin/bash
# On the GPU box: kernel trace of one bench.py workload and the per-step timeline (tools/timeline.py) of a step of the timed region.
#   bash tools/timeline_workload.sh <workload> <first-kernel-of-a-step>      e.g.  scalogram scalogram_pointwise | attention conv1_fwd
WL=${1:-scalogram}; FIRST=${2:-scalogram_pointwise}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl_$WL
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps 6 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $F --first $FIRST --back 2 > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
tail -4 $OUT/timeline.txt
