for v in "$@"; do
  cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so
  cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash
  echo "=== $v"
  CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_$v)" bash tools/prof_trace.sh abt_$v chain_gemm 2>/dev/null | tail -11
done
