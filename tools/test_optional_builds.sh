# Runs the parity tests of the optional instantiations (csrc/experimental/ + the L0M render kernel) on a library built with them:
#   CIPS3D_EXPERIMENTAL=1 CIPS3D_HIPCC_FLAGS=-DCIPS3D_NERF_L0M python -m cips_3dplusplus_amd.build ; cp the .so / .srchash to _ab/{lib,hash}_exp
cp cips_3dplusplus_amd/libcips3d_hip.so /tmp/lib_default.so; cp cips_3dplusplus_amd/libcips3d_hip.so.srchash /tmp/hash_default
cp cips_3dplusplus_amd/_ab/lib_exp.so cips_3dplusplus_amd/libcips3d_hip.so; cp cips_3dplusplus_amd/_ab/hash_exp cips_3dplusplus_amd/libcips3d_hip.so.srchash
CIPS3D_EXPERIMENTAL=1 CIPS3D_HIPCC_FLAGS=-DCIPS3D_NERF_L0M python3 -m pytest tests/test_gpu_split_fp16.py tests/test_gpu_parity.py -m gpu -q -k "pair or weight_stationary or layer0 or other_render_arithmetics" 2>&1 | grep -E "^(FAILED|E  )|passed|failed" | head -30
cp /tmp/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp /tmp/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
