# Same-box A/B of the stand-alone render loop over prebuilt library variants (tools/ab_build.sh): HIP events, processes interleaved.
# usage: ROUNDS=2 tools/nerf_variant_ab.sh "run_kernel args" default tagA tagB ...
R=${ROUNDS:-2}; ARGS="$1"; shift
cp cips_3dplusplus_amd/libcips3d_hip.so /tmp/lib_default.so; cp cips_3dplusplus_amd/libcips3d_hip.so.srchash /tmp/hash_default
for i in $(seq $R); do
  for v in "$@"; do
    if [ $v = default ]; then cp /tmp/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp /tmp/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash; F="";
    else cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so; cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash; F="$(cat cips_3dplusplus_amd/_ab/flags_$v)"; fi
    echo -n "$v: "; CIPS3D_HIPCC_FLAGS="$F" python3 tools/run_kernel.py nerf --iters 200 $ARGS 2>/dev/null | grep "nerf_render kernel"
  done
done
cp /tmp/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp /tmp/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
