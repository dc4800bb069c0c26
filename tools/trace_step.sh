# the launch sequence of ONE inversion step (between two render-kernel launches), from a kernel trace
export TMPDIR=/tmp
O=gpurun_out/trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/bench_inversion.py --steps 24 > $O/run.log 2>&1
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - $T <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nerf_render_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("launches in the step:", b - a, " span us:", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
prev_end = t0
for r in rows[a:b]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f}  {n}")
    prev_end = e
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
