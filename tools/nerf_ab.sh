# A/B of build flags (render kernel, chain GEMM, fused stages) on ONE box: usage  bash tools/nerf_ab.sh "<flags A>" "<flags B>" ...
export TMPDIR=/tmp
i=0
for f in "$@"; do
  i=$((i+1)); rm -rf gpurun_out/ab$i
  CIPS3D_HIPCC_FLAGS="$f" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab$i -- python3 bench.py --no-cpu-baseline --no-also --steps 40 > gpurun_out/ab$i.log 2>&1
  find gpurun_out/ab$i -name "*kernel_trace.csv" -delete
  echo "== flags [$f]"; python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/ab$i/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'nerf_render' in r['Name'] or 'chain_gemm' in r['Name'] or 'fused_up' in r['Name']: print('  ', r['Name'][27:82], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
done
