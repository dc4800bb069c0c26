# Unprofiled same-box A/B of prebuilt library variants (tools/ab_build.sh) on the headline bench: variants interleaved, ROUNDS rounds,
# one bench.py process each (200-step regions, median of 5).  usage: ROUNDS=2 tools/ab_bench.sh default tagA tagB ...   [BENCH_ARGS=...]
R=${ROUNDS:-2}
mkdir -p cips_3dplusplus_amd/_ab
if [ ! -f cips_3dplusplus_amd/_ab/lib_default.so ]; then
  python3 -m cips_3dplusplus_amd.build > /dev/null 2>&1
  cp cips_3dplusplus_amd/libcips3d_hip.so cips_3dplusplus_amd/_ab/lib_default.so
  cp cips_3dplusplus_amd/libcips3d_hip.so.srchash cips_3dplusplus_amd/_ab/hash_default; : > cips_3dplusplus_amd/_ab/flags_default
fi
for i in $(seq $R); do
  for v in "$@"; do
    cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so
    cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash
    echo -n "$v: "
    CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_$v)" python3 bench.py --no-also --no-cpu-baseline --steps 200 --repeats 5 --detail "" ${BENCH_ARGS} 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_repeats'], 'render', d['roofline']['avg_launch_ms'])"
  done
done
cp cips_3dplusplus_amd/_ab/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp cips_3dplusplus_amd/_ab/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
