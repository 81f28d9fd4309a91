#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of bench.py.
# Usage: bash tools/collect_profiles.sh <tag> [workload]     workload: cfg1 (default) | scalogram | conv_ar | attention
#   -> gpurun_out/prof_<tag>[_<workload>]/{stats,fetch,write,mfma}; condense with tools/summarize_profiles.py <tag> [workload]
set -e
TAG=${1:-r01}
WL=${2:-cfg1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$WL" = "cfg1" ]; then OUT=gpurun_out/prof_$TAG; EXTRA="--no-trainer-loop --no-score-gemm --no-secondary"; else OUT=gpurun_out/prof_${TAG}_$WL; EXTRA="--workload $WL"; fi
mkdir -p $OUT
# (>= 20 warm-up steps: the first dozen steps of a process run 5 - 8 % slower in every kernel while the clocks ramp; round 3's sets were
# taken inside that ramp)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 25 --no-cpu-baseline $EXTRA > $OUT/stats.log 2>&1
# one step of the timed region, kernel by kernel (start, queue, duration, idle gap): tools/timeline.py on the trace before it is trimmed
TR=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
if [ "$WL" = "cfg1" ]; then FIRST=conv1_fwd; elif [ "$WL" = "scalogram" ]; then FIRST=stem_stats; else FIRST=conv1_fwd; fi      # (the scalogram of the NEXT batch now runs mid-step on the side stream)
python3 tools/timeline.py $TR --first $FIRST --back 12 > $OUT/timeline_one_step.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/write.log 2>&1
# matrix-pipe occupancy of every kernel: SQ_VALU_MFMA_BUSY_CYCLES against the dispatch's GRBM_GUI_ACTIVE (separate pass, counters only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline $EXTRA > $OUT/mfma.log 2>&1
# the raw traces are tens of MB: keep the per-kernel summaries and the counter tables, drop the rest
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*.csv" | head -20
tail -n 1 $OUT/stats.log | cut -c1-400
