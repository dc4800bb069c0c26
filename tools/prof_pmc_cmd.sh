# instruction-mix + wave-state counters of an arbitrary python script: tools/prof_pmc_cmd.sh <tag> <kernel substring> <script> [args...]
export TMPDIR=/tmp
T=$1; K=$2; shift; shift
O=gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $O/i -o p -- python3 "$@" > $O/i.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/s -o p -- python3 "$@" > $O/s.log 2>&1
python3 - $K $(find $O/i -name "*counter_collection.csv" | head -1) $(find $O/s -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections, re
sub = sys.argv[1]
for path in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]: continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        agg[k]["_waves"].append(max(1, int(r["Grid_Size"]) // 64))
    for k, v in agg.items():
        w = v["_waves"][0]
        print(k, "us", round(sum(v["_dur"]) / len(v["_dur"]) / 1e3, 1), "waves", w,
              {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if not c.startswith("_")})
PY
rm -rf $O/i $O/s
