cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 0 1 2 3 4 5 6 7 8 9 10; do
  CIPS3D_GEMM_CFG=$c rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gemm_cfg$c -- python3 tools/run_kernel.py gemm64 --iters 30 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/gemm_cfg$c/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "modconv1x1" in r["Name"]:
        print("cfg $c", r["Name"][40:75], "avg us", float(r["AverageNs"])/1e3, "min", float(r["MinNs"])/1e3)
PY
done
