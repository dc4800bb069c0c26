export CIPS3D_NERF_PAIR=1
python -m cips_3dplusplus_amd.build > /dev/null 2>&1
python tools/nerf_pair_ab.py 2>&1 | tail -8
python tools/nerf_pair_ab.py --n-samples 23 --batch 2 2>&1 | tail -8
python tools/nerf_pair_ab.py --batch 4 --no-perturb 2>&1 | tail -8
