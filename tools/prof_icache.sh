export TMPDIR=/tmp
O=gpurun_out/pmc_ic; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/i -o p -- python3 tools/bench_style_phase.py 1024 1 > $O/i.log 2>&1
tail -3 $O/i.log
python3 - $(find $O/i -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-40:]
    if "style_phase" not in k and "linear" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
PY
