# LDS-side counters of the render kernel (stand-alone render loop): how busy the LDS is, conflicts, waves waiting on LDS, the
# per-kind active-instruction cycles.  usage: tools/prof_render_lds.sh [run_kernel args]  -> gpurun_out/render_lds.txt
export TMPDIR=/tmp
O=gpurun_out/rlds; rm -rf $O; mkdir -p $O
P() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$N -o p -- python3 tools/run_kernel.py nerf --iters 20 $ARGS > $O/$N.log 2>&1; }
ARGS="$*"
N=a; P SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS
N=b; P SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES
N=c; P SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
N=d; P SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS
python3 - $O > gpurun_out/render_lds.txt <<'PY'
import csv, sys, glob, collections, re
agg = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if "nerf_render_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg["_dur_" + path.split("/")[-2]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items()):
    print(f"{k:28s} mean {sum(v) / len(v):16.1f}  (n = {len(v)})")
PY
cat gpurun_out/render_lds.txt
