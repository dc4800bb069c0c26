# final-build profiles of round 6 (r06), all from ONE box: smoke, bench lines, kernel-trace summaries (headline, N = 64, config 3,
# fp32_exact, stand-alone 3x3 and upfirdn2d tools), the PMC passes of the headline + FETCH / WRITE passes of the N = 64 and batch-4
# workloads, each summary stamped with the library's source hash (bench.py replays traffic only from a matching one)
export TMPDIR=/tmp
O=gpurun_out/r06; rm -rf $O; mkdir -p $O
# kernel summaries and counters are taken with ONE view in flight (--lanes 1: every launch alone on the device, the durations
# roofline.avg_launch_ms is measured against); `final_lanes2` is the same command with the default two lanes
B="python3 bench.py --no-cpu-baseline --no-also --lanes 1"
B2="python3 bench.py --no-cpu-baseline --no-also"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
stats() {   # stats <name> <program...>: rocprofv3 --kernel-trace --stats of a command, the kernel_stats.csv kept as <name>_kernel_stats.csv
  N=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$N -- "$@" > $O/$N.log 2>&1
  cp $(find $O/$N -name "*kernel_stats.csv" | head -1) $O/${N}_kernel_stats.csv; rm -rf $O/$N
}
stats final $B --steps 50 --repeats 2
stats final_lanes2 $B2 --steps 50 --repeats 2
stats n64 $B --steps 30 --repeats 2 --n-samples 64
stats config3_bf16 $B --batch 4 --decoder-precision bf16 --steps 30 --repeats 2
stats fp32_exact $B --decoder-precision fp32_exact --steps 30 --repeats 2
stats config2_r256 $B --res 256 --steps 50 --repeats 2
stats multiview python3 tools/bench_multiview.py --chunks 1 --rounds 2
stats conv3x3_tool python3 tools/bench_conv3x3.py
stats upfirdn2d_tool python3 tools/bench_upfirdn2d.py
python3 tools/bench_conv3x3.py > $O/conv3x3_tool.jsonl 2> /dev/null
python3 tools/bench_upfirdn2d.py > $O/upfirdn2d_tool.jsonl 2> /dev/null
P="--steps 12 --warmup 3 --repeats 1"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o p -- $B $P > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $O/insts -o p -- $B $P > $O/insts.log 2>&1
for W in "h:" "n64:--n-samples 64" "b4:--batch 4 --decoder-precision bf16"; do
  T=${W%%:*}; A=${W#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$T -o p -- $B $P $A > $O/fetch_$T.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$T -o p -- $B $P $A > $O/write_$T.log 2>&1
done
python3 tools/pmc_summary.py $O/sq/p_counter_collection.csv $O/fetch_h/p_counter_collection.csv $O/write_h/p_counter_collection.csv > $O/pmc_all_kernels.json 2> $O/pmc_summary.err
python3 tools/pmc_summary.py --workload "bench.py --n-samples 64" $O/fetch_n64/p_counter_collection.csv $O/write_n64/p_counter_collection.csv > $O/pmc_n64_traffic.json 2>> $O/pmc_summary.err
python3 tools/pmc_summary.py --workload "bench.py --batch 4 --decoder-precision bf16" $O/fetch_b4/p_counter_collection.csv $O/write_b4/p_counter_collection.csv > $O/pmc_b4_traffic.json 2>> $O/pmc_summary.err
python3 - $O/insts/p_counter_collection.csv > $O/pmc_instruction_mix.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg[k]["_waves"].append(max(1, int(r["Grid_Size"]) // 64))
print("# per-wave instruction counts (SQ_INSTS_* / waves); SQ_INSTS_VALU includes the MFMAs")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["_dur"])):
    if sum(v["_dur"]) / len(v["_dur"]) < 3000: continue
    w = v["_waves"][0]
    print(k, "us_under_pmc", round(sum(v["_dur"]) / len(v["_dur"]) / 1e3, 1), "waves", w,
          {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if not c.startswith("_")})
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
# the bench lines last: with the summaries of THIS library copied where bench.py looks for them, the line carries measured traffic
mkdir -p profiles; for f in pmc_all_kernels pmc_n64_traffic pmc_b4_traffic; do cp $O/$f.json profiles/r06_$f.json; done
python3 bench.py --detail $O/bench_default_detail.json > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 --detail $O/bench_driver_form_detail.json > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python3 tools/bench_multiview.py --lanes 1,2 > $O/multiview_ab.txt 2> /dev/null
python3 tools/graph_lanes_probe.py 3 > $O/graph_lanes_probe.json 2> /dev/null
du -sh $O
