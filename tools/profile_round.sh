# final-build profiles of round 2 (n): headline kernel stats, config-3 kernel stats (bf16 planes16 run), bench lines
export TMPDIR=/tmp
O=gpurun_out/r02n; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2000 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/final -- $B --steps 50 > $O/final.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- $B --batch 4 --decoder-precision bf16 --steps 30 > $O/c3.log 2>&1
find $O -name "*kernel_trace.csv" -delete
du -sh $O
