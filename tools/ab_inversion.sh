# Unprofiled same-box A/B of prebuilt library variants on BASELINE config 5 (tools/bench_inversion.py), variants interleaved.
# usage: ROUNDS=2 tools/ab_inversion.sh default tagA ...
R=${ROUNDS:-2}
for i in $(seq $R); do
  for v in "$@"; do
    cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so
    cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash
    echo -n "$v: "
    CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_$v)" python3 tools/bench_inversion.py --steps 150 2>/dev/null | tail -1 | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms')"
  done
done
cp cips_3dplusplus_amd/_ab/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp cips_3dplusplus_amd/_ab/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
