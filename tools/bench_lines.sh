# the two bench lines (default and the driver's form) + the multi-view A/B, from one box -> gpurun_out/lines/
O=gpurun_out/lines; rm -rf $O; mkdir -p $O
python3 bench.py --detail $O/bench_default_detail.json > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 --detail $O/bench_driver_form_detail.json > $O/bench_driver_form.json 2> $O/bench_driver_form.err
python3 tools/bench_multiview.py --lanes 1,2 > $O/multiview_ab.txt 2> /dev/null
tail -c 1500 $O/bench_default.json; cat $O/multiview_ab.txt
