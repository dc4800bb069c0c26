# Same-box A/B of the render kernel with layer 0 + view-direction columns on the matrix cores (default) against the VALU form
# (CIPS3D_NERF_L0M=0): stand-alone render loop, HIP events, processes interleaved.  usage: tools/nerf_l0m_ab.sh [rounds] [run_kernel args]
R=${1:-3}; shift
for i in $(seq $R); do
  for v in 1 0; do
    echo -n "L0M=$v: "; CIPS3D_NERF_L0M=$v python3 tools/run_kernel.py nerf --iters 200 "$@" 2>/dev/null | grep "nerf_render kernel"
  done
done
