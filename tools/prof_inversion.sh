# kernel-trace summary of the inversion step (BASELINE config 5) + the unprofiled rate
export TMPDIR=/tmp
O=gpurun_out/inv; rm -rf $O; mkdir -p $O
python3 tools/bench_inversion.py --steps 104 > $O/bench.json 2> $O/bench.err; cat $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_inversion.py --steps 44 > $O/kt.log 2>&1
S=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp $S $O/inversion_kernel_stats.csv
python3 - $O/inversion_kernel_stats.csv <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total ms", tot / 1e6, "calls", calls)
for r in rows[:45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n)[:64]
    print(f"{n:64s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}")
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
