for v in "" -DCIPS3D_BWD_NO_SUMS -DCIPS3D_BWD_NO_COS -DCIPS3D_BWD_NO_STASH -DCIPS3D_BWD_NO_MFMA "-DCIPS3D_BWD_NO_SUMS -DCIPS3D_BWD_NO_COS -DCIPS3D_BWD_NO_STASH"; do
  echo "== $v"; CIPS3D_HIPCC_FLAGS="$v" timeout 600 python tools/nerf_bwd_compare.py --iters 5 2>&1 | grep "^H="
done
