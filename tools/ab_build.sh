# builds library variants for a same-box A/B (tools/ab_run.sh): usage tools/ab_build.sh tag "flags" [tag "flags" ...]
# each variant lands in cips_3dplusplus_amd/_ab/{lib,hash,flags}_<tag>; the default library is rebuilt at the end
mkdir -p cips_3dplusplus_amd/_ab
while [ $# -ge 2 ]; do
  T=$1; F=$2; shift 2
  CIPS3D_HIPCC_FLAGS="$F" python3 -m cips_3dplusplus_amd.build > /tmp/ab_build_$T.log 2>&1 || { echo "build of $T failed"; tail -5 /tmp/ab_build_$T.log; exit 1; }
  cp cips_3dplusplus_amd/libcips3d_hip.so cips_3dplusplus_amd/_ab/lib_$T.so
  cp cips_3dplusplus_amd/libcips3d_hip.so.srchash cips_3dplusplus_amd/_ab/hash_$T
  echo "$F" > cips_3dplusplus_amd/_ab/flags_$T
  echo "built $T ($F)"
done
python3 -m cips_3dplusplus_amd.build > /tmp/ab_build_default.log 2>&1; echo "default rebuilt"
