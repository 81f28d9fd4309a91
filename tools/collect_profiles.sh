#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes of bench.py.
# Usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write,mfma}
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-trainer-loop > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-trainer-loop > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-trainer-loop > $OUT/write.log 2>&1
# matrix-pipe occupancy of every kernel: SQ_VALU_MFMA_BUSY_CYCLES against the dispatch's GRBM_GUI_ACTIVE (separate pass, counters only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-trainer-loop > $OUT/mfma.log 2>&1
find $OUT -name "*.csv" | head -20
