# kernel-trace summary of an arbitrary python script: tools/prof_cmd.sh <tag> <script> [args...]
export TMPDIR=/tmp
T=$1; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 "$@" > $O/run.log 2>&1
S=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - $S <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n)[:70]
    print(f"{n:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
tail -12 $O/run.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
