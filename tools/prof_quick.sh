# kernel-trace summary of the headline bench into gpurun_out/$1 (usage: tools/prof_quick.sh tag [bench args])
export TMPDIR=/tmp
T=$1; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-also --steps 50 "$@" > $O/bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
S=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = 0.0
for r in rows[:24]:
    n = r["Name"]
    n = n.replace("void (anonymous namespace)::", "")[:70]
    print(f'{n:70s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f}%')
PY
tail -1 $O/bench.log | cut -c1-400
