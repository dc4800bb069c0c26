# Reproducer of the packed-fp32 ToRGB-fold nondeterminism of csrc/chain.hip (bf16 planes kernel, two workgroups per CU): builds
# the library with the fold's FMAs in several forms and counts repeats whose partial sums differ (tools/fold_repeat.py, 60 runs
# each).  CIPS3D_FOLD_PK=1: the C form (hipcc's SLP vectoriser packs channels 1 / 2 into v_pk_fma_f32 chains when CIPS3D_CHAIN_SLP=1);
# CIPS3D_FOLD_PK=2: that instruction pattern written by hand (v_mov_b32 x 2 into a register pair -> v_pk_fma_f32 op_sel_hi) with
# CIPS3D_FOLD_NOP wait states between the moves and the packed instruction.  Output -> profiles/r04_pk_fold_probe.txt
O=gpurun_out/pk_probe.txt; : > $O
run() {  # name, CIPS3D_CHAIN_SLP, flags
  export CIPS3D_CHAIN_SLP=$2 CIPS3D_HIPCC_FLAGS="$3"
  python -m cips_3dplusplus_amd.build > /tmp/pk_probe_build.log 2>&1 || { echo "== $1: BUILD FAILED (no result)" >> $O; tail -3 /tmp/pk_probe_build.log >> $O; return; }
  echo "== $1  (CIPS3D_CHAIN_SLP=$2 $3)" >> $O
  python tools/fold_repeat.py 60 2>&1 | grep "repeats differ" >> $O
}
run "shipped: asm v_fmac_f32, no SLP"                                  0 ""
run "C fmaf, SLP on (compiler-made v_pk_fma_f32 chains)"                1 "-DCIPS3D_FOLD_PK=1"
run "C fmaf, SLP off"                                                   0 "-DCIPS3D_FOLD_PK=1"
run "compiler-made chains + 16 wait states behind each"                 1 "-DCIPS3D_FOLD_PK=1 -DCIPS3D_FOLD_NOP=1"
run "compiler-made chains + 16 wait states in front of each"            1 "-DCIPS3D_FOLD_PK=1 -DCIPS3D_FOLD_NOP=2"
run "hand-made v_mov pair -> v_pk_fma_f32, 0 wait states between"       0 "-DCIPS3D_FOLD_PK=2 -DCIPS3D_FOLD_NOP=0"
run "hand-made v_mov pair -> v_pk_fma_f32, 1 wait state between"        0 "-DCIPS3D_FOLD_PK=2 -DCIPS3D_FOLD_NOP=1"
run "hand-made v_mov pair -> v_pk_fma_f32, 3 wait states between"       0 "-DCIPS3D_FOLD_PK=2 -DCIPS3D_FOLD_NOP=3"
# round 5: the two compiler-made forms round 4's hand-written probe did not cover (csrc/chain.hip, CIPS3D_FOLD_PK=3)
run "hand-made, all four compiler forms (constant-0 start, op_sel high-register broadcast)"   0 "-DCIPS3D_FOLD_PK=3 -DCIPS3D_FOLD_FORMS=3"
run "hand-made, constant-0 start form only"                              0 "-DCIPS3D_FOLD_PK=3 -DCIPS3D_FOLD_FORMS=1"
run "hand-made, op_sel:[0,1,0] high-register broadcast only"             0 "-DCIPS3D_FOLD_PK=3 -DCIPS3D_FOLD_FORMS=2"
run "hand-made, neither (four op_sel_hi:[1,0,1] forms)"                  0 "-DCIPS3D_FOLD_PK=3 -DCIPS3D_FOLD_FORMS=0"
run "compiler-made chains, every load of the wave drained before the epilogue"  1 "-DCIPS3D_FOLD_PK=1 -DCIPS3D_FOLD_NOP=4"
run "compiler-made chains, drained + workgroup barrier before the fold"  1 "-DCIPS3D_FOLD_PK=1 -DCIPS3D_FOLD_NOP=8"
# leave the tree as it was found: default flags, default library (as tools/ab_build.sh does)
unset CIPS3D_CHAIN_SLP CIPS3D_HIPCC_FLAGS
python -m cips_3dplusplus_amd.build > /tmp/pk_probe_build.log 2>&1 && echo "default rebuilt" || { echo "default rebuild FAILED"; tail -3 /tmp/pk_probe_build.log; }
cat $O
