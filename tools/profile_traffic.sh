# the PMC passes of tools/profile_round.sh only (traffic of the three workloads, instruction mix): into gpurun_out/r03
export TMPDIR=/tmp
O=gpurun_out/r03; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
P="--steps 12 --warmup 3 --repeats 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA --output-format csv -d $O/insts -o p -- $B $P > $O/insts.log 2>&1
for W in "n64:--n-samples 64" "b4:--batch 4 --decoder-precision bf16"; do
  T=${W%%:*}; A=${W#*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$T -o p -- $B $P $A > $O/fetch_$T.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$T -o p -- $B $P $A > $O/write_$T.log 2>&1
done
python3 tools/pmc_summary.py $O/fetch_n64/p_counter_collection.csv $O/write_n64/p_counter_collection.csv > $O/pmc_n64_traffic.json
python3 tools/pmc_summary.py $O/fetch_b4/p_counter_collection.csv $O/write_b4/p_counter_collection.csv > $O/pmc_b4_traffic.json
sed -n '/^python3 - \$O\/insts/,/^PY$/p' tools/profile_round.sh > /tmp/insts_cmd.sh; bash /tmp/insts_cmd.sh
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
