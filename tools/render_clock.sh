# In-kernel shader clock of the shipped render kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6) on a -DCIPS3D_CLOCK build made
# by `tools/ab_build.sh clock "-DCIPS3D_CLOCK"` (the variant travels under cips_3dplusplus_amd/_ab/).  One stamp pair around the
# whole kernel per workgroup, after >= 2 s of back-to-back launches on random data; split-fp16 and exact-fp32 instantiations,
# stand-alone render loop and inside the whole forward.  -> gpurun_out/render_clock.jsonl (copy to profiles/r05_render_clock.json)
set -e
O=gpurun_out/render_clock.jsonl; : > $O
cp cips_3dplusplus_amd/libcips3d_hip.so /tmp/lib_default.so; cp cips_3dplusplus_amd/libcips3d_hip.so.srchash /tmp/hash_default
cp cips_3dplusplus_amd/_ab/lib_clock.so cips_3dplusplus_amd/libcips3d_hip.so
cp cips_3dplusplus_amd/_ab/hash_clock cips_3dplusplus_amd/libcips3d_hip.so.srchash
export CIPS3D_HIPCC_FLAGS="$(cat cips_3dplusplus_amd/_ab/flags_clock)"
for P in fp32 fp32_exact; do
  python3 tools/run_kernel.py nerf --iters 50 --precision $P --clock-json $O
  python3 tools/run_kernel.py nerf --iters 50 --precision $P --n-samples 64 --clock-json $O
done
python3 tools/run_kernel.py forward --iters 50 --clock-json $O
python3 tools/run_kernel.py nerf --iters 50 --depth 8 --res 256 --clock-json $O
unset CIPS3D_HIPCC_FLAGS
cp /tmp/lib_default.so cips_3dplusplus_amd/libcips3d_hip.so; cp /tmp/hash_default cips_3dplusplus_amd/libcips3d_hip.so.srchash
cat $O
