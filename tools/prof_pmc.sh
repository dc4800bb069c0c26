# per-kernel counter averages of the headline bench (usage: tools/prof_pmc.sh tag "COUNTER ..." [bench args])
export TMPDIR=/tmp
T=$1; CTRS=$2; shift; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc -o p -- python3 bench.py --no-cpu-baseline --no-also --steps 6 --warmup 2 --repeats 1 "$@" > $O/bench.log 2>&1
S=$(find $O -name "*counter_collection.csv" | head -1)
python3 - "$S" <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:60]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg[k]["_waves"].append(int(r.get("Grid_Size", r.get("Grid_Size_X", 64))) // 64)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["_dur"])):
    if sum(v["_dur"]) / len(v["_dur"]) < 3000: continue
    waves = v["_waves"][0]
    out = [f"{k:62s} us {sum(v['_dur'])/len(v['_dur'])/1e3:7.1f} waves {waves:7d}"]
    for c, vals in v.items():
        if c.startswith("_"): continue
        m = sum(vals) / len(vals)
        out.append(f"{c}={m:.3g} ({m / waves:.1f}/wave)")
    print("  ".join(out))
PY
head -2 "$S" | cut -c1-600 > $O/header.txt; rm -rf $O/pmc
