cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tl_scal
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --workload scalogram --steps 6 --warmup 3 --no-cpu-baseline > $OUT/run.log 2>&1
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $F --first scalogram_pointwise --back 2 > $OUT/timeline.txt 2>&1
rm -rf $OUT/trace
tail -5 $OUT/timeline.txt
