# like kernel_times.sh for an arbitrary python command: bash tools/kernel_times_cmd.sh <tag> <regex> <script> [args...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/$1; RE=$2; shift 2
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 "$@" > $OUT/run.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
cp $f $OUT/stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
python3 - "$OUT/stats.csv" "$RE" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:100]:100s} n={r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
