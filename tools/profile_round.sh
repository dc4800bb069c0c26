export TMPDIR=/tmp
O=gpurun_out/r02h; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/n64 -- $B --n-samples 64 --steps 30 > $O/n64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d8 -- $B --depth 8 --steps 30 > $O/d8.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o p -- $B --steps 12 --warmup 3 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B --steps 12 --warmup 3 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B --steps 12 --warmup 3 > $O/write.log 2>&1
for prec in bf16 bf16_storage; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c3_${prec}_fetch -o p -- $B --batch 4 --decoder-precision $prec --steps 12 --warmup 3 > $O/c3_${prec}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_${prec}_write -o p -- $B --batch 4 --decoder-precision $prec --steps 12 --warmup 3 > $O/c3_${prec}_w.log 2>&1
done
find $O -name "*.csv" | head -40
# keep merged output small: drop kernel traces of the pmc passes except counter_collection
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O
