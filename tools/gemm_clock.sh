cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gemm_clock; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d gpurun_out/gemm_clock -- python3 tools/gemm_clock.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
d=glob.glob("gpurun_out/gemm_clock/*/")[0]
tr=[r for r in csv.DictReader(open(glob.glob(d+"*kernel_trace.csv")[0])) if "modconv1x1" in r["Kernel_Name"]]
dur={r["Dispatch_Id"]:(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in tr}
cc=collections.defaultdict(dict)
for r in csv.DictReader(open(glob.glob(d+"*counter_collection.csv")[0])):
    if "modconv1x1" in r["Kernel_Name"]: cc[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in cc.items():
    us=dur[k]; flops=2*512*512*512*512
    clk=v["GRBM_GUI_ACTIVE"]/8/us/1e3
    print(f"dur {us:8.1f} us  {flops/us/1e6:6.1f} TFLOP/s  clock {clk:.2f} GHz  mfma busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*v['GRBM_GUI_ACTIVE']/8)*100:.1f}% of SIMD-cycles")
PY
