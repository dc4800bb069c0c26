# kernel-trace summary of the hoisted multi-view sequence (config 4, chunk 1) -> gpurun_out/$1
export TMPDIR=/tmp
T=${1:-mv}; O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_multiview.py --chunks ${2:-1} --rounds 3 ${3:+--only $3} > $O/run.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
S=$(find $O -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv
python3 - "$S" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    n = r["Name"].replace("void (anonymous namespace)::", "")[:80]
    print(f'{n:80s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f}%')
PY
tail -4 $O/run.log
