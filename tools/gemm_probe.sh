cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gemm_probe; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gemm_probe -- python3 tools/gemm_probe.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/gemm_probe/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "modconv1x1" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# 12 launches per (K, ep) in order
i=0
for K in (64,128,256,512,1024):
    for ep in (0,1):
        d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows[i:i+12]); i+=12
        print(f"K={K:5d} ep={ep} median {d[6]:7.2f} us min {d[0]:7.2f}")
PY
