# A/B of prebuilt library variants (cips_3dplusplus_amd/_ab/{lib,hash,flags}_<tag>): usage tools/ab_run.sh tag...
for v in "$@"; do
  cp cips_3dplusplus_amd/_ab/lib_$v.so cips_3dplusplus_amd/libcips3d_hip.so
  cp cips_3dplusplus_amd/_ab/hash_$v cips_3dplusplus_amd/libcips3d_hip.so.srchash
  echo "=== $v"
  CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_$v)" bash tools/prof_quick.sh ab_$v 2>/dev/null | head -${AB_LINES:-8}
done
