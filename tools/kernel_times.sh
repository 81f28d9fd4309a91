# per-kernel average times of a short bench run: bash tools/kernel_times.sh <tag> [pattern]   (summary -> gpurun_out/<tag>/stats.csv)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-kt}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trainer-loop > $OUT/bench.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
python3 - "$OUT/stats.csv" "${2:-.}" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:90]:90s} n={r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
