# per-phase cycle sums of one early-epilogue wave (0) and one late-epilogue wave (4) of the render kernel, on -DCIPS3D_STAMPS builds made by
#   tools/ab_build.sh st0 "-DCIPS3D_STAMPS -DCIPS3D_STAMP_WAVE=0" st4 "-DCIPS3D_STAMPS -DCIPS3D_STAMP_WAVE=4"
cp cips_3dplusplus_amd/libcips3d_hip.so /tmp/lib_default.so; cp cips_3dplusplus_amd/libcips3d_hip.so.srchash /tmp/hash_default
for v in st0 st4; do
  cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so; cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash
  echo "=== $v $*"; CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_$v)" python3 tools/run_kernel.py nerf --iters 30 "$@" 2>/dev/null | tail -16
done
cp /tmp/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp /tmp/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
