# BASELINE config 5 (one flip-inversion step): unprofiled rate + kernel-trace summary of the one-call decoder route and of the
# per-op route (CIPS3D_ONE_CALL_DECODER=0 CIPS3D_HIP_ADAM=0 CIPS3D_FUSED_ADAM=0: round 2's) on the SAME box, + the launch sequence of one step
export TMPDIR=/tmp
O=gpurun_out/inv; rm -rf $O; mkdir -p $O
summ() {
python3 - $1 $2 <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1]))); steps = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
setup = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("(anonymous namespace)::linear_kernel"))
print(f"# {steps} steps: kernel time {tot/1e6:.1f} ms ({(tot-setup)/steps/1e6:.3f} ms per step without the mean-latent set-up), {calls/steps:.0f} launches per step")
for r in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); n = re.sub(r"^void ", "", n)[:64]
    print(f"{n:64s} {int(r['Calls'])/steps:7.1f}/step {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/steps/1e3:9.1f} us/step")
PY
}
python3 tools/bench_inversion.py --steps 104 > $O/bench_one_call.json 2> $O/bench.err; cat $O/bench_one_call.json
[ -n "$QUICK" ] || { CIPS3D_ONE_CALL_DECODER=0 CIPS3D_HIP_ADAM=0 CIPS3D_FUSED_ADAM=0 python3 tools/bench_inversion.py --steps 104 > $O/bench_per_op.json 2>> $O/bench.err; cat $O/bench_per_op.json; }
# (104 steps: the one-time launches of the run -- mean latents, weight packing, ~260 fills of the Adam state -- are ~3 per step of the average)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/bench_inversion.py --steps 104 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/inversion_kernel_stats.csv
summ $O/inversion_kernel_stats.csv 104 > $O/inversion_summary.txt; head -30 $O/inversion_summary.txt
if [ -z "$QUICK" ]; then
export CIPS3D_ONE_CALL_DECODER=0 CIPS3D_HIP_ADAM=0 CIPS3D_FUSED_ADAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt0 -- python3 tools/bench_inversion.py --steps 44 > $O/kt0.log 2>&1
unset CIPS3D_ONE_CALL_DECODER CIPS3D_FUSED_ADAM CIPS3D_HIP_ADAM
cp $(find $O/kt0 -name "*kernel_stats.csv" | head -1) $O/inversion_per_op_kernel_stats.csv
summ $O/inversion_per_op_kernel_stats.csv 44 > $O/inversion_per_op_summary.txt; head -3 $O/inversion_per_op_summary.txt
fi
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 tools/bench_inversion.py --steps 24 > $O/tr.log 2>&1
python3 - $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/inversion_step_trace.txt <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nerf_render_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("# one inversion step under rocprofv3 --kernel-trace (the profiler stretches the host side): launches", b - a, " span us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
prev_end = t0
for r in rows[a:b]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n)[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} gap {(s - prev_end) / 1e3:6.1f} dur {(e - s) / 1e3:7.1f}  {n}")
    prev_end = e
PY
head -2 $O/inversion_step_trace.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; rm -rf $O/kt $O/kt0 $O/tr
