# final-build profiles of round 2 (m): headline kernel stats, config-3 kernel stats (bf16 planes16 run), driver-form bench line
export TMPDIR=/tmp
O=gpurun_out/r02m; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final -- $B --steps 50 > $O/final.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- $B --batch 4 --decoder-precision bf16 --steps 30 > $O/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3s -- $B --batch 4 --decoder-precision bf16_storage --steps 30 > $O/c3s.log 2>&1
find $O -name "*kernel_trace.csv" -delete
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
du -sh $O
