export TMPDIR=/tmp
O=gpurun_out/r02i; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 50 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B --steps 12 --warmup 3 > $O/fetch.log 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
python -m pytest tests/test_gpu_parity.py -q -k "fused or chained or stage or golden" 2>&1 | tail -2
