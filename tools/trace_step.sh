#!/bin/bash
# One step of the headline workload, kernel by kernel: bash tools/trace_step.sh <tag> [ENV=val ...]  -> gpurun_out/<tag>/timeline.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$tag
for v in "$@"; do export "$v"; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/tr -- python3 bench.py --steps 20 --warmup 25 --no-cpu-baseline --no-secondary --no-trainer-loop --no-score-gemm > gpurun_out/$tag/log.txt 2>&1 || exit 1
TR=$(find gpurun_out/$tag/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $TR --first conv1_fwd --back 12 > gpurun_out/$tag/timeline.txt 2>&1
find gpurun_out/$tag -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/$tag/timeline.txt
