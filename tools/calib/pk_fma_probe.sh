# builds tools/calib/pk_fma_probe.hip twice (SLP vectorisation on / off), links, runs; prints the packed instructions found
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
T=${TMPDIR:-/tmp}/pk_probe; mkdir -p $T; cd $T
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-gpu-rdc"
hipcc $F -DKNAME=probe_slp -save-temps=obj -c $R/tools/calib/pk_fma_probe.hip -o $T/slp.o 2> $T/slp.log
cp $T/pk_fma_probe-hip-amdgcn-amd-amdhsa-gfx950.s $T/slp.s
hipcc $F -DKNAME=probe_ref -fno-slp-vectorize -save-temps=obj -c $R/tools/calib/pk_fma_probe.hip -o $T/ref.o 2> $T/ref.log
cp $T/pk_fma_probe-hip-amdgcn-amd-amdhsa-gfx950.s $T/ref.s
hipcc $F -DKNAME=probe_main_unused -DPROBE_MAIN -c $R/tools/calib/pk_fma_probe.hip -o $T/main.o 2> $T/main.log
hipcc --offload-arch=gfx950 $T/slp.o $T/ref.o $T/main.o -o $T/pk_probe
echo "v_pk_fma_f32 in probe_slp: $(grep -c v_pk_fma_f32 $T/slp.s)   in probe_ref: $(grep -c v_pk_fma_f32 $T/ref.s)"
if [ -e /dev/kfd ]; then $T/pk_probe; else echo "(no GPU here: built only)"; fi
