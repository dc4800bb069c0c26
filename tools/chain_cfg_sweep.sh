export TMPDIR=/tmp
for c in 0 1 2; do
  rm -rf gpurun_out/cc$c
  CIPS3D_CHAIN_CFG=$c rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cc$c -- python3 bench.py --no-cpu-baseline --no-also --steps 40 > gpurun_out/cc$c.log 2>&1
  find gpurun_out/cc$c -name "*kernel_trace.csv" -delete
done
python -m pytest tests/test_gpu_split_fp16.py tests/test_gpu_parity.py -q -k "planes or chain or golden" 2>&1 | tail -2
