# final-build profiles of round 3 (r03): smoke, kernel-trace summaries (headline, N = 64, config 3), the three PMC passes of the
# headline + FETCH / WRITE passes of the N = 64 and batch-4 workloads (render-kernel traffic), bench lines
export TMPDIR=/tmp
O=gpurun_out/r03; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final -- $B --steps 50 --repeats 2 > $O/final.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n64 -- $B --steps 30 --repeats 2 --n-samples 64 > $O/n64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- $B --batch 4 --decoder-precision bf16 --steps 30 --repeats 2 > $O/c3.log 2>&1
P="--steps 12 --warmup 3 --repeats 1"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o p -- $B $P > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $O/insts -o p -- $B $P > $O/insts.log 2>&1
for W in "h:" "n64:--n-samples 64" "b4:--batch 4 --decoder-precision bf16"; do
  T=${W%%:*}; A=${W#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$T -o p -- $B $P $A > $O/fetch_$T.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$T -o p -- $B $P $A > $O/write_$T.log 2>&1
done
python3 tools/pmc_summary.py $O/sq/p_counter_collection.csv $O/fetch_h/p_counter_collection.csv $O/write_h/p_counter_collection.csv > $O/pmc_all_kernels.json 2> $O/pmc_summary.err
python3 tools/pmc_summary.py $O/fetch_n64/p_counter_collection.csv $O/write_n64/p_counter_collection.csv > $O/pmc_n64_traffic.json 2>> $O/pmc_summary.err
python3 tools/pmc_summary.py $O/fetch_b4/p_counter_collection.csv $O/write_b4/p_counter_collection.csv > $O/pmc_b4_traffic.json 2>> $O/pmc_summary.err
python3 - $O/insts/p_counter_collection.csv > $O/pmc_insts.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg[k]["_waves"].append(max(1, int(r["Grid_Size"]) // 64))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["_dur"])):
    if sum(v["_dur"]) / len(v["_dur"]) < 3000: continue
    w = v["_waves"][0]
    print(k, "us_under_pmc", round(sum(v["_dur"]) / len(v["_dur"]) / 1e3, 1), "waves", w,
          {c: round(sum(x) / len(x) / w, 1) for c, x in v.items() if not c.startswith("_")})
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
