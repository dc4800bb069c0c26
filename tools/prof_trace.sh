# per-dispatch durations of one kernel family over a few forwards (usage: tools/prof_trace.sh tag substring [bench args])
export TMPDIR=/tmp
T=$1; SUB=$2; shift; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 --repeats 1 "$@" > $O/bench.log 2>&1
S=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$S" "$SUB" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-40:]
for r in tail:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f'{d:8.2f} us grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} wg {r["Workgroup_Size_X"]} lds {r.get("LDS_Block_Size","?")} {r["Kernel_Name"][:60]}')
PY
rm -rf $O/kt
