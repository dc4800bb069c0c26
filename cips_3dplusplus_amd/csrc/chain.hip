// The run of 1x1 StyledConvs at the NeRF resolution (conv1 + convs.0-7 of the release decoder: nine 512-channel GEMMs at
// 64^2, models/model_v3.py:602-632) on SPLIT-fp16 activations that STAY split between layers.
//
// Split-fp16 arithmetic (decoder.hip, nerf.hip): x = hi + lo with hi = fp16(x), lo = fp16(x - hi) -- 22 significant bits in
// the 4 bytes of an fp32 -- and a product is three exact fp16 products accumulated in fp32 on v_mfma_f32_16x16x32_f16.
// cips3d_modconv1x1 in CIPS3D_GEMM_SPLIT mode splits the fp32 activations in registers after every LDS read (3 VALU per
// value, repeated by every wave row): VALU-bound at ~15.5 us per layer.  Here the PRODUCER's epilogue splits each output
// once and stores the two halves in the consumer's MFMA fragment order ("planes"), so the main loop of the next layer is
// ds_read_b128 + MFMA only.
//
//   planes layout   P[b][C/8][plane hi|lo][HW][8] (fp16): 16 bytes = the 8 channels 8 cb .. 8 cb + 7 of one pixel, i.e. exactly
//                   one lane's B fragment of a 32-channel MFMA block (lane quarter q <-> channel block 4 kb + q); same bytes
//                   as the fp32 tensor it replaces.  Rows (cb, plane) are pixel-contiguous: LDS-DMA pieces of 64 pixels.
//   weights         cips3d_modulate_weights(CIPS3D_MOD_PACKED | CIPS3D_MOD_SPLIT): natural k order (lane quarter q, element j
//                   <-> channel 32 kb + 8 q + j), 2^8-scaled, hi / lo halves.
//   epilogue        x 2^-8 (exact), NoiseInjection + bias + leaky ReLU, optional folded ToRGB partial sums (as
//                   cips3d_modconv1x1_torgb), then either planes again (the next layer of the run) or fp32 / bf16 NCHW (the
//                   low-resolution GEMM that feeds the first fused up-sampling stage).
//   range           planes hold x * 2^-e with one exponent per (sample, block of 128 pixels) -- GEMM columns are independent, so
//                   a pixel block may carry a scale of its own (cips3d_range, common.h).  The input block's exponent comes from
//                   its producer (x_exp) and is undone on the accumulators together with the weights' 2^-8.  The output's is
//                   chosen HERE from the rigorous bound |out| <= c1 max|in| + c0 (the layer's constants) with max|in| taken
//                   over this workgroup's own pixel block only: the producing launch left the maximum of every
//                   (16 channels x 64 pixels) patch it stored in a small side array (x_pmax: one plain store per wave, no
//                   atomics, nothing to zero), a wave reads the 2 x Cin/16 entries of its pixel block with ONE load in front
//                   of the main loop and reduces them in registers; the eight workgroups of a pixel block read the same
//                   entries and so agree on the exponent without talking to each other.  It rides on the sqrt(2) of the
//                   activation -- no instruction per value -- and row block 0 writes it to out_exp for the consumer; every
//                   wave leaves its own patch maximum in out_pmax.
//                   (Forms measured on the way: per-sample maxima through atomicMax slots and scalar loads, 14.6 us per layer
//                   against 11.1 -- most of it was NOT the atomics but the 8 bytes of __shared__ for the workgroup reduction,
//                   which moved the ring off LDS offset 0 and the allocation past 96 KB: 2.3 us; max|in| from the hi halves
//                   the main loop reads anyway: +0.9 us per layer, its instructions sit between a wave's MFMAs and the stage
//                   barrier in front of the next DMA issue.)  The fp32 / bf16 exit of the run raises out_amax (per-sample
//                   slots) for the fused up-sampling stage that reads it.
//
// Bound: L2 -> LDS bytes (64 x 128 tiles: 384 KB per workgroup and layer) + the launch skeleton; the matrix time is ~1/5 of
// the fp32 MFMA's.
//
// bf16 decoder mode (BASELINE config 3; NP = 1 below): the same kernel on ONE plane of bf16 -- "planes16",
// P16[b][C/8][HW][8] (bf16), weights cips3d_modulate_weights(CIPS3D_MOD_PACKED | CIPS3D_MOD_BF16) -- and one
// v_mfma_f32_16x16x32_bf16 per k-block and tile: half the operand bytes of a launch that is bound by them.  The mode's
// definition (both GEMM operands rounded to bf16, fp32 accumulation; oracle/path.py:modulated_conv2d bf16_gemm) makes the
// stored bf16 activation exactly the operand the next layer's GEMM would have rounded to, so nothing changes numerically
// against the fp32-stored form of the mode; the folded ToRGB still reads the unrounded fp32 registers.
//
// Build note (round 3).  With SLP vectorisation hipcc pairs the ToRGB fold's channel-1 / channel-2 accumulations into
// v_pk_fma_f32 chains (operands picked with op_sel / op_sel_hi broadcasts out of v_mov-assembled register pairs).  In the bf16
// kernel -- two workgroups per CU -- the partial sums of that form then differ from run to run: tools/fold_repeat.py, batch 4,
// 59 of 59 repeats differ from the first (up to 1840 of 393k sums; always channel 2, the low half of the pair, for isolated
// 16-pixel column tiles); never in the split-fp16 kernel (one workgroup per CU), never with one FMA per product.  The ISA of
// the failing build shows nothing illegal around the chains (operands complete behind `s_waitcnt vmcnt(0) lgkmcnt(0)`, plain
// VALU -> VALU dependencies), the LDS exchange and the operand loads were ruled out one by one in round 2: it behaves like a
// packed-fp32 problem of the part under that occupancy, not like a race in this code.  The fold therefore spells its FMAs
// out (asm v_fmac_f32 in the epilogue: 0 of 59 repeats differ, with or without SLP), so correctness does not hang on a
// compiler flag; build.py still compiles this file with -fno-slp-vectorize (no other packed-fp32 arithmetic in the planes
// kernels either; no measurable cost), and tests/test_gpu_bf16_storage.py repeats the folds bit for bit.
#include <stdlib.h>
#include "common.h"

#ifndef CIPS3D_FOLD_PK
#define CIPS3D_FOLD_PK 0
#endif
#ifndef CIPS3D_FOLD_NOP
#define CIPS3D_FOLD_NOP 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

constexpr float kSplitInv = 1.f / 256.f;       // the modulate kernel scaled the weights by 2^8

__device__ __forceinline__ constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

#ifdef CIPS3D_CHAIN_STAMPS
// Diagnostic build only: per-phase cycle sums over all workgroups (wave 0), one batch of atomics per workgroup at its end.
__device__ unsigned long long g_chain_stamps[8];
extern "C" int cips3d_debug_read_chain_stamps(unsigned long long* out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_chain_stamps), 64);
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stamps), z, 64);
  return 0;
}
#define CSTAMP(i) ts_[i] = __builtin_amdgcn_s_memtime()
#else
#define CSTAMP(i)
#endif

struct ChainArgs {
  const _Float16* x;        // planes [B][Cin/8][2][HW][8] (fp16 hi / lo), or planes16 [B][Cin/8][HW][8] (bf16; NP = 1)
  const float* wmp;         // split-packed (NP = 2) or bf16-packed (NP = 1) weights
  void* out;                // planes (out_fmt 1), fp32 NCHW (0), bf16 NCHW (2) or planes16 (3)
  int out_fmt;
  int B, Cin, Cout; int HW;
  int epilogue; const float* noise; int64_t noise_bstride; const float* noise_w; const float* bias;
  const float* rgb_w; float* rgb_part;
  // range tracking (all optional): cips3d_range.  x_exp / out_exp: [B][ceil(HW / 128)] per pixel block; x_exp NULL: x_exp_const.
  // x_pmax / out_pmax: [B][ceil(HW / 64)][C / 16] patch maxima; x_pmax NULL: max|in| = x_max_const
  const int* x_exp; int x_exp_const; const float* x_pmax; float x_max_const; const float* lconst; float* out_amax; int* out_exp;
  float* out_pmax;
  cips3d_reduce_job ride;   // part == NULL: none.  A ToRGB fold this launch carries (cips3d_range::ride)
};



// NP = planes per operand: 2 = split-fp16 (hi, lo; three fp16 MFMAs per tile and k-block), 1 = bf16 (one bf16 MFMA)
template <int WM, int WGM, int WGN, int BK, int NS, int NP = 2>
__global__ void __launch_bounds__(64 * WGM * WGN) chain_gemm_kernel(ChainArgs a) {
#ifdef CIPS3D_CHAIN_STAMPS
  unsigned long long ts_[6];
  CSTAMP(0);
#endif
  constexpr int NW = WGM * WGN;
  constexpr int BM = 16 * WM * WGM, BN = 64 * WGN;
  constexpr int A_STAGE = BM * BK * NP / 2, B_STAGE = BK * BN * NP / 2, STAGE = A_STAGE + B_STAGE;   // floats (NP 2-byte planes)
  constexpr int A_PIECES = A_STAGE / 256, B_PIECES = B_STAGE / 256, PIECES = A_PIECES + B_PIECES;
  constexpr int PW = PIECES / NW;
  constexpr int KB = BK / 32, KQ = KB * NP;      // KQ = 1 KB A pieces per o-tile and stage
  static_assert(PIECES % NW == 0 && BK % 32 == 0, "tile shape");
  static_assert(NP != 2 || BN == CIPS3D_PLANES_EXP_BLOCK, "one planes exponent per workgroup pixel block");
  __shared__ __attribute__((aligned(16))) float lds[NS * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm_i = wave / WGN, wn_i = wave % WGN;
  const int q = lane >> 4, col = lane & 15;
  const int b = blockIdx.z;

  // The riding ToRGB fold (cips3d_reduce_job): one extra row of workgroups (blockIdx.y == Cout / BM) that the host adds to the
  // grid of a launch that leaves CUs free (the 512 -> 256 exit of the 64^2 run: 128 tiles).  As in torgb_reduce_kernel a float4
  // position is shared by four slot groups (group g adds slots g, g + 4, ... in that order, absent ones of a batch of four as
  // zeros; then the groups 0 .. 3, the biases, the skip): here the groups are the lane quarters of a wave, 16 positions per wave.
  if constexpr (NP == 2) {
    if (a.ride.part && (int)blockIdx.y == a.Cout / BM) {
      constexpr int RIDE_K = 12;                 // loads per lane: n_slots <= 4 * RIDE_K
      const int64_t wv = ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * NW + wave;
      const int64_t ride_i = wv * 16 + col;
      const bool ride_on = ride_i < a.ride.n4;
      f32x4 rsl[RIDE_K], rsk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < RIDE_K; ++k)
        rsl[k] = (ride_on && q + 4 * k < a.ride.n_slots)
                     ? *reinterpret_cast<const f32x4*>(a.ride.part + (q + 4 * k) * a.ride.slot_stride + ride_i * 4)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      if (ride_on && a.ride.skip && q == 0) rsk = *reinterpret_cast<const f32x4*>(a.ride.skip + ride_i * 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < RIDE_K / 4; ++it)
        if (q + 16 * it < a.ride.n_slots) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += rsl[4 * it + u][c];
        }
#pragma unroll
      for (int g = 1; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float o = __shfl(v[c], col + 16 * g, 64);
          if (q == 0) v[c] += o;
        }
      if (ride_on && q == 0) {
        const int ch = (int)((ride_i / a.ride.HW4) % 3);
        float bs = 0.f;
        for (int k = 0; k < a.ride.n_bias; ++k) bs += a.ride.bias[k][ch];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += bs;
        if (a.ride.skip) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += rsk[c];
        }
        *reinterpret_cast<f32x4*>(a.ride.out + ride_i * 4) = v;
      }
      return;
    }
  }
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int HW = a.HW, K = a.Cin;
  const int nstage = K / BK;
  // 4 * NW LDS words for the workgroup reductions after the main loop (max|in| of the pixel block; max|out| of an fp32 exit).
  // They live in the ring slot the LAST K stage does not use -- dead for every wave once it is past that stage's barrier.  (A
  // separate __shared__ array moved the ring off LDS offset 0 and rounded the allocation up past 96 KB: every layer of the
  // run was 2.3 us, 20 %, slower for it.)
  float* s_part = lds + (nstage % NS) * STAGE;
  const _Float16* xb = a.x + (int64_t)b * K * HW * NP;           // (Cin/8) * NP planes * HW * 8 two-byte elements
  const float* ab = a.wmp + (int64_t)b * a.Cout * K * NP / 2;   // 2-byte elements, NP planes

  auto stage_load = [&](int st) {
    float* dstA = lds + (st % NS) * STAGE;
    float* dstB = dstA + A_STAGE;
    const int k0 = st * BK;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int piece = j * NW + wave;
      // (saddr form of the LDS-DMA: a uniform base advanced on the scalar unit + a 32-bit lane offset, both made opaque so that
      // instruction selection sees base + zext(offset) -- with per-lane 64-bit pointers every piece first needed a
      // v_lshl_add_u64 into the register pair the previous piece was still issuing from)
      if (piece < A_PIECES) {
        const int ot_l = piece / KQ, kq_l = piece % KQ;
        const char* ub = reinterpret_cast<const char*>(ab + ((((m0 >> 4) + ot_l) * ((K >> 5) * NP) + (k0 >> 5) * NP + kq_l) * 256));
        unsigned vo = lane * 16;
        asm volatile("" : "+s"(ub), "+v"(vo));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + vo),
                                         (__attribute__((address_space(3))) void*)(dstA + piece * 256), 16, 0, 0);
      } else {
        const int pb = piece - A_PIECES;
        const int row = pb / (BN / 64), chunk = pb % (BN / 64);     // row = (channel block of the stage) * NP + plane
        int n = n0 + chunk * 64 + lane;
        if (n > HW - 1) n = HW - 1;                                  // clamp: those columns are never stored
        const char* ub = reinterpret_cast<const char*>(xb + ((int64_t)((k0 >> 3) * NP + row) * HW) * 8);
        unsigned vo = (unsigned)n * 16u;
        asm volatile("" : "+s"(ub), "+v"(vo));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + vo),
                                         (__attribute__((address_space(3))) void*)(dstB + pb * 256), 16, 0, 0);
      }
    }
  };

  // this lane's four pixels: 16 c + col of the wave's 64-pixel strip (c = MFMA column tile)
  const int nl = wn_i * 64 + col;
  int npx[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) npx[c] = n0 + nl + 16 * c;

  float nz[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4[WM];
  f32x4 wrgb[WM][3];
  float nw = 0.f;
  float kin = NP == 2 ? kSplitInv : 1.f;     // accumulator -> true value: the weights' 2^-8 and the input planes' 2^e
  float kout = 1.f, kback = 1.f;             // true value -> what this launch stores (2^-e' for a planes output) and back
  float lc0 = 0.f, lc1 = 0.f;                // the layer's bound constants (planes output)
  float pin = 0.f;                           // this lane's entries of the input's patch maxima (planes output)
  // a planes output under range tracking: the exponent follows from max|in| of this pixel block and the layer's constants
#ifndef CIPS3D_CHAIN_AB
#define CIPS3D_CHAIN_AB 0        // timing-only ablations: 1 no range work at all, 2 no patch-maxima store, 3 no patch-maxima load
#endif
  const bool track = CIPS3D_CHAIN_AB != 1 && NP == 2 && a.out_fmt == 1 && a.lconst;
  // The epilogue's operands are REQUESTED here (pure loads into registers, right behind the first stage's LDS-DMA) and only
  // turned into what the epilogue needs after the main loop (finish_ops).  Round 5: in the old form -- loads and their arithmetic
  // interleaved -- `a.x_exp ? a.x_exp[..] : a.x_exp_const` became ONE flat load of a selected address (array or kernarg word), a
  // flat load's result needs vmcnt(0), and that wait sat in front of every other operand load: the whole first stage had to land
  // before the remaining ~5 dependent round trips (bound constants, patch maxima, noise, bias, ToRGB rows) even started --
  // 1-2 us per launch in front of the main loop.
#ifndef CIPS3D_CHAIN_LATE_OPS
#define CIPS3D_CHAIN_LATE_OPS 1     // 0: finish_ops right behind issue_ops, as before round 5 (A/B)
#endif
  int e_raw = 0;
  f32x4 lc_raw = {0.f, 0.f, 0.f, 0.f};
  auto issue_ops = [&]() {
    if constexpr (NP == 2) {
      const int nblk = (HW + BN - 1) / BN;
      if (a.x_exp) {        // (an agent-scope relaxed load: a plain global load the compiler cannot fold into a select of addresses)
        typedef const __attribute__((address_space(1))) int gi32_t;
        e_raw = __hip_atomic_load(reinterpret_cast<gi32_t*>(reinterpret_cast<uintptr_t>(a.x_exp + b * nblk + blockIdx.x)),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (track) {
        const float* lc = a.lconst + b * 4;
        lc_raw = f32x4{lc[0], lc[1], lc[2], 0.f};
        if (a.x_pmax && CIPS3D_CHAIN_AB != 3) {      // the 2 x Cin/16 patch maxima of this pixel block are contiguous: one load per wave (Cin <= 512)
          const int n_half = (HW + 63) / 64, per = K >> 4;
          const int n_ent = (2 * (int)blockIdx.x + 1 < n_half ? 2 : 1) * per;
          const float* pp = a.x_pmax + ((int64_t)b * n_half + 2 * blockIdx.x) * per;
          if (lane < n_ent) pin = pp[lane];        // (entries past the first 64 -- Cin > 512 only -- follow in finish_ops)
        }
      }
    }
    if (a.epilogue == 1) {
      if (a.noise && a.noise_w) {
        nw = a.noise_w[0];
        const float* nzp = a.noise + (int64_t)b * a.noise_bstride;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (npx[c] < HW) nz[c] = nzp[npx[c]];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
        bias4[i] = *reinterpret_cast<const f32x4*>(a.bias + m0 + (wm_i * WM + i) * 16 + 4 * q);
    }
    if (a.rgb_part) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          wrgb[i][ch] = *reinterpret_cast<const f32x4*>(a.rgb_w + (int64_t)b * 3 * a.Cout + ch * a.Cout + m0 + (wm_i * WM + i) * 16 + 4 * q);
    }
  };
  auto finish_ops = [&]() {
    if constexpr (NP == 2) {
      const int e_in = a.x_exp ? __builtin_amdgcn_readfirstlane(e_raw) : a.x_exp_const;
      kin = cips3d_uniform(kin * cips3d_pow2(e_in));
      if (track) {
        lc0 = lc_raw[0];
        lc1 = fmaxf(lc_raw[1], 1.41421356237309515f * lc_raw[2]);
        if (a.x_pmax && CIPS3D_CHAIN_AB != 3) {
          const int n_half = (HW + 63) / 64, per = K >> 4;
          const int n_ent = (2 * (int)blockIdx.x + 1 < n_half ? 2 : 1) * per;
          const float* pp = a.x_pmax + ((int64_t)b * n_half + 2 * blockIdx.x) * per;
          for (int i = lane + 64; i < n_ent; i += 64) pin = fmaxf(pin, pp[i]);
        } else {
          // no patch maxima: the caller's constant, or what the input's own exponent says -- its producer put a bound of ITS
          // output below 2^15 2^e_in.  That is a bound of a bound (another ~2^5 of slack: the stored values then top out near
          // 2^5 instead of 2^10, still > 27 bits above the pair's floor), so a run alternates: every other layer leaves patch
          // maxima for its consumer (forward.hip), which halves what the tracking costs (0.45 us per writing layer)
          pin = a.x_max_const > 0.f ? a.x_max_const : cips3d_pow2(e_in + 15);
        }
      }
    }
  };
  auto load_ops = [&]() {
    issue_ops();
    if (!CIPS3D_CHAIN_LATE_OPS) finish_ops();
  };
  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Ring of NS stages, loads NS - 1 stages ahead; the epilogue's operand loads go out first (older than every stage, so the
  // first counted wait retires them).  What bounds this kernel (in-kernel stamps, tools/chain_stamps.py, 512 -> 512 at 64^2):
  // 3.1k cycles to the first landed stage, 12.6k for the 8 K stages, 2.5k epilogue -- and inside a stage ~1000 cycles of
  // waiting for the stage's data against ~450 of DMA issue and ~350 of fragment reads + MFMA issue.  A third ring slot,
  // 32-deep stages with 4 or 6 slots, and issuing the DMA behind the MFMAs (all waves, or only the SIMD partners) change
  // nothing or lose: the loop moves 48 KB per stage and CU in ~1600 cycles = 30 B/clk per CU = 18 TB/s chip-wide, which is
  // the L2 -> LDS gather rate of this part (MI355X_MICROARCH.md, "Indexed rows: gather into LDS": 66-73 GB/s per CU).  The
  // launch is bound by L2 -> CU bandwidth for its 384 KB of operand tiles per CU, not by latency and not by the matrix pipe.
  // (Round 6: the run as ONE persistent launch -- this body as the per-layer step of a multi-layer kernel, one counter per pixel-block
  // column between layers, bit-identical results -- was built and measured: a hop cost 2.2-2.4 us where a launch boundary between two
  // of these kernels costs 0.7 us (nine layers 118.2 us in one launch, 104.6 us as nine).  Commit 47e2af1 has the kernel,
  // profiles/r06_chain_seq_hop.jsonl the numbers; it is not in the tree.)
  // (It is the CU's whole L2 port, not the LDS-DMA path: with the A fragments loaded global -> registers directly, one stage
  // ahead, and only the B tile through the ring -- half the DMA bytes, parity-green -- a layer took 11.48 us against 11.34.)
  static_assert(NS >= 2 && (NS - 2) * PW <= 63, "counted vmcnt");
  // (NS = 2: the operand loads ride BEHIND the first stage's DMA -- the loop's first wait is vmcnt(0) anyway, and in front
  // of it they delayed the first stage: 0.367 -> 0.384 ms per forward; deeper rings: in front, older than every counted stage)
  if (NS > 2) load_ops();
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nstage) stage_load(s0);
  if (NS == 2) load_ops();

#ifdef CIPS3D_CHAIN_STAMPS
  unsigned long long ph_[4] = {0, 0, 0, 0}, tp_ = __builtin_amdgcn_s_memtime();
#define PSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[i] += t_ - tp_; tp_ = t_; } while (0)
#else
#define PSTAMP(i)
#endif
  for (int st = 0; st < nstage; ++st) {
    int younger = nstage - 1 - st;
    if (younger > NS - 2) younger = NS - 2;
    // stage st has landed when at most the `younger` stages issued after it are still in flight
    switch (younger) {
      case 0: __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); break;
      case 1: __builtin_amdgcn_s_waitcnt(vmcnt_imm(PW)); break;
      case 2: __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * PW)); break;
      case 3: __builtin_amdgcn_s_waitcnt(vmcnt_imm(3 * PW)); break;
      case 4: __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 * PW)); break;
      default: __builtin_amdgcn_s_waitcnt(vmcnt_imm(5 * PW)); break;
    }
    __builtin_amdgcn_s_barrier();
#ifdef CIPS3D_CHAIN_STAMPS
    if (st == 0) CSTAMP(1);          // first stage landed
#endif
    if (st + NS - 1 < nstage) stage_load(st + NS - 1);
    PSTAMP(1);                         // DMA issue
    const float* sA = lds + (st % NS) * STAGE;
    const float* sB = sA + A_STAGE;
    if constexpr (NP == 2) {
      h8 ah[KB][WM], al[KB][WM], bh[KB][4], bl[KB][4];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          ah[kb][i] = *reinterpret_cast<const h8*>(sA + ((wm_i * WM + i) * KQ + 2 * kb) * 256 + lane * 4);
          al[kb][i] = *reinterpret_cast<const h8*>(sA + ((wm_i * WM + i) * KQ + 2 * kb + 1) * 256 + lane * 4);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {      // row (4 kb + q) * 2 + plane, pixel nl + 16 c: [row][BN px][4 floats]
          bh[kb][c] = *reinterpret_cast<const h8*>(sB + (((kb * 4 + q) * 2 + 0) * BN + nl + 16 * c) * 4);
          bl[kb][c] = *reinterpret_cast<const h8*>(sB + (((kb * 4 + q) * 2 + 1) * BN + nl + 16 * c) * 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kb][i], bh[kb][c], acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb][i], bl[kb][c], acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb][i], bh[kb][c], acc[i][c], 0, 0, 0);
          }
    } else {
      bf8 af[KB][WM], bf[KB][4];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
          af[kb][i] = *reinterpret_cast<const bf8*>(sA + ((wm_i * WM + i) * KQ + kb) * 256 + lane * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c)        // row 4 kb + q (channel block), pixel nl + 16 c
          bf[kb][c] = *reinterpret_cast<const bf8*>(sB + ((kb * 4 + q) * BN + nl + 16 * c) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kb][i], bf[kb][c], acc[i][c], 0, 0, 0);
    }
    PSTAMP(2);                         // fragment reads + MFMA issue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  CSTAMP(2);                         // main loop
  if (CIPS3D_CHAIN_LATE_OPS) finish_ops();
  if (track) {
    // max|in| of the pixel block from the lanes' entries (wave-uniform, no LDS), the bound, the exponent, its powers of two
    const float m_in = cips3d_wave_max_uniform(pin);
    const int e = cips3d_split_exp(fmaf(lc1, m_in * 1.000001f, lc0));
    kout = cips3d_uniform(cips3d_pow2(-e));
    kback = cips3d_uniform(cips3d_pow2(e));
    if (blockIdx.y == 0 && tid == 0) a.out_exp[b * ((HW + BN - 1) / BN) + blockIdx.x] = e;
  }
#if CIPS3D_FOLD_PK && (CIPS3D_FOLD_NOP & 4)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // probe: every load of this wave -- the epilogue's operands -- has returned
#endif
  // ---- epilogue.  D layout: acc[i][c][r] = out[o = obase + 4 q + r][pixel npx[c]]
  // (one accumulator set per 16-row tile: the ToRGB slots are sums of FOUR tiles in a fixed order whatever the workgroup's height --
  // a 128-row workgroup writes two 64-row slots that hold the same bits two 64-row workgroups would have written)
  float prgb[WM][3][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int c = 0; c < 4; ++c) prgb[i][ch][c] = 0.f;
  // v below is the STORED value, out * kout (kout = 1 unless the output is planes): the power of two rides on the constants
  // the epilogue multiplies by anyway, the ToRGB partial sums and the recorded maximum are taken from v and scaled back once
  const float kact = 1.41421356237309515f * kout;
  const float kpre = a.epilogue == 1 ? kin : kin * kout;      // (one uniform multiplier: no branch per value)
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int obase = m0 + (wm_i * WM + i) * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][c][r] * kpre;
        if (a.epilogue == 1) v[r] = lrelu02(fmaf(nw, nz[c], v[r]) + bias4[i][r]) * kact;
      }
      // (columns past HW repeat the last pixel without its noise: a value the patch maximum may include -- it only has to bound)
      mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), mx));
#if CIPS3D_FOLD_PK == 3
      // probe (tools/pk_fold_probe.sh, round 5): EXACTLY the four instruction forms hipcc's SLP build emits for one column (read
      // off its assembly: profiles/r05_slp_fold_isa.md) -- r = 0: v_pk_fma_f32 D, W, V01, 0 op_sel_hi:[1,0,0] (fresh accumulator,
      // inline constant 0, broadcast of the pair's low register); r = 1: ... V01 ... op_sel:[0,1,0] (broadcast of the pair's HIGH
      // register); r = 2, 3: op_sel_hi:[1,0,1] -- round 4's hand-written probe covered only the last form.  CIPS3D_FOLD_FORMS
      // masks which of the first two are used (bit 0: r = 0, bit 1: r = 1); a cleared bit falls back to the last form.
#ifndef CIPS3D_FOLD_FORMS
#define CIPS3D_FOLD_FORMS 3
#endif
      if (a.rgb_part) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t p12 = {prgb[i][1][c], prgb[i][2][c]};
        const f32x2_t v01 = {v[0], v[1]}, v23 = {v[2], v[3]}, v1x = {v[1], 0.f}, v3x = {v[3], 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(prgb[i][0][c]) : "v"(wrgb[i][0][r]), "v"(v[r]));
        const f32x2_t w0 = {wrgb[i][1][0], wrgb[i][2][0]}, w1 = {wrgb[i][1][1], wrgb[i][2][1]}, w2 = {wrgb[i][1][2], wrgb[i][2][2]},
                      w3 = {wrgb[i][1][3], wrgb[i][2][3]};
        if ((CIPS3D_FOLD_FORMS & 1) && i == 0) {
          // (the compiler's first instruction starts the accumulator: prgb is zero before the first row tile)
          asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(p12) : "v"(w0), "v"(v01));
        } else {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p12) : "v"(w0), "v"(v01));
        }
        if (CIPS3D_FOLD_FORMS & 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(p12) : "v"(w1), "v"(v01));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p12) : "v"(w1), "v"(v1x));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p12) : "v"(w2), "v"(v23));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p12) : "v"(w3), "v"(v3x));
        prgb[i][1][c] = p12[0];
        prgb[i][2][c] = p12[1];
      }
#elif CIPS3D_FOLD_PK == 2
      // probe (tools/pk_fold_probe.sh): the SLP build's instruction pattern written by hand -- channel 0 as v_fmac_f32, channels
      // 1 / 2 as ONE register pair per product assembled by two v_mov_b32 into fixed registers and consumed by v_pk_fma_f32 with
      // an op_sel_hi broadcast of v[r] -- with CIPS3D_FOLD_NOP wait states between the v_movs and the packed instruction
      if (a.rgb_part) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t p12 = {prgb[i][1][c], prgb[i][2][c]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(prgb[i][0][c]) : "v"(wrgb[i][0][r]), "v"(v[r]));
          const f32x2_t vb = {v[r], 0.f};
          asm volatile("v_mov_b32 v126, %1\n\tv_mov_b32 v127, %2\n\t"
                       ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
                       "v_pk_fma_f32 %0, v[126:127], %3, %0 op_sel_hi:[1,0,1]"
                       : "+v"(p12) : "v"(wrgb[i][1][r]), "v"(wrgb[i][2][r]), "v"(vb), "n"(CIPS3D_FOLD_NOP) : "v126", "v127");
        }
        prgb[i][1][c] = p12[0];
        prgb[i][2][c] = p12[1];
      }
#else
      if (a.rgb_part) {
#if CIPS3D_FOLD_PK && (CIPS3D_FOLD_NOP & 2)
        asm volatile("s_nop 7\n\ts_nop 7");          // probe: distance between the producers of v[] and the packed chain
#endif
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            // one v_fmac_f32 per product, written out: nothing can pair the channel-1 / channel-2 accumulations into
            // v_pk_fma_f32 (see the build note in the header of this file; VALU results feed VALU here: no hazard state to keep)
#if CIPS3D_FOLD_PK        /* reproducer builds only (tools/pk_fold_probe.sh): the C form SLP packs */
            prgb[i][ch][c] = fmaf(wrgb[i][ch][r], v[r], prgb[i][ch][c]);
#else
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(prgb[i][ch][c]) : "v"(wrgb[i][ch][r]), "v"(v[r]));
#endif
#if CIPS3D_FOLD_PK && (CIPS3D_FOLD_NOP & 1)
        asm volatile("s_nop 7\n\ts_nop 7");          // probe: distance between the packed chain and whatever follows it
#endif
      }
#endif
      if (npx[c] < HW) {
        if (a.out_fmt == 1) {
          // planes: channels obase + 4 q + r live in channel block (obase >> 3) + (q >> 1), elements 4 (q & 1) + r
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          unsigned h0, l0, h1, l1;
          cips3d_split_pair(v[0], v[1], h0, l0);
          cips3d_split_pair(v[2], v[3], h1, l1);
          const h4 hi = __builtin_bit_cast(h4, u32x2_t{h0, h1}), lo = __builtin_bit_cast(h4, u32x2_t{l0, l1});
          _Float16* dst = reinterpret_cast<_Float16*>(a.out) +
                          ((((int64_t)b * (a.Cout >> 3) + (obase >> 3) + (q >> 1)) * 2) * HW + npx[c]) * 8 + 4 * (q & 1);
          cips3d_store_wt8(dst, hi);                      // (write-through: common.h)
          cips3d_store_wt8(dst + (int64_t)HW * 8, lo);
        } else if (a.out_fmt == 3) {
          // planes16: the same channel block / element positions, one bf16 plane (round to nearest even: the operand rounding
          // of the bf16 mode)
          unsigned short* dst = reinterpret_cast<unsigned short*>(a.out) +
                                (((int64_t)b * (a.Cout >> 3) + (obase >> 3) + (q >> 1)) * HW + npx[c]) * 8 + 4 * (q & 1);
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const bf16x2_t p0 = __builtin_convertvector(f32x2_t{v[0], v[1]}, bf16x2_t);
          const bf16x2_t p1 = __builtin_convertvector(f32x2_t{v[2], v[3]}, bf16x2_t);
          cips3d_store_wt8(dst, u32x2_t{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)});
        } else if (a.out_fmt == 2) {
          unsigned short* dst = reinterpret_cast<unsigned short*>(a.out) + ((int64_t)b * a.Cout + obase + 4 * q) * HW + npx[c];
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const bf16x2_t p = __builtin_convertvector(f32x2_t{v[r], v[r + 1]}, bf16x2_t);
            const unsigned bits = __builtin_bit_cast(unsigned, p);
            dst[(int64_t)r * HW] = (unsigned short)(bits & 0xffffu);
            dst[(int64_t)(r + 1) * HW] = (unsigned short)(bits >> 16);
          }
        } else {
          float* dst = reinterpret_cast<float*>(a.out) + ((int64_t)b * a.Cout + obase + 4 * q) * HW + npx[c];
#pragma unroll
          for (int r = 0; r < 4; ++r) cips3d_store_wt(dst + (int64_t)r * HW, v[r]);
        }
      }
    }
  }
#ifdef CIPS3D_CHAIN_STAMPS
  CSTAMP(3);                         // epilogue arithmetic + stores issued
  __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
  CSTAMP(4);                         // stores acknowledged
  if (tid == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&g_chain_stamps[i], ts_[i + 1] - ts_[i]);
    atomicAdd(&g_chain_stamps[7], 1ull);
    atomicAdd(&g_chain_stamps[4], ph_[0]);
    atomicAdd(&g_chain_stamps[5], ph_[1]);
    atomicAdd(&g_chain_stamps[6], ph_[2]);
  }
#endif
  if (CIPS3D_CHAIN_AB != 1 && CIPS3D_CHAIN_AB != 2 && a.out_pmax && n0 + wn_i * 64 < HW) {
    // this wave's patch (16 WM channels x 64 pixels): its largest |out|, one plain store per 16 channels
    static_assert(BN == 64 * WGN, "a wave column = one 64-pixel half block");
    const float m = cips3d_wave_max_uniform(mx * kback);
    if (lane < WM)
      a.out_pmax[((int64_t)b * ((HW + 63) / 64) + (n0 >> 6) + wn_i) * (a.Cout >> 4) + (m0 >> 4) + wm_i * WM + lane] = m;
  }
  if (a.out_amax) {        // fp32 / bf16 exit: the workgroup's largest |out| raises one slot of the sample's amax array
    const float m = cips3d_workgroup_max(mx * kback, s_part, wave, lane, NW);
    if (tid == 0) cips3d_amax_raise_if(a.out_amax + b * CIPS3D_AMAX_FLOATS, m, blockIdx.y * gridDim.x + blockIdx.x);
  }
  if (!a.rgb_part) return;
#if CIPS3D_FOLD_PK && (CIPS3D_FOLD_NOP & 8)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // probe: nothing of this wave in flight, then everybody here
  __syncthreads();
#endif
  // ---- folded ToRGB partial of this workgroup's BM rows: over the 4 lane quarters by shuffles, over the WGM wave rows
  // through LDS (every wave passed the last stage's lgkmcnt(0) and meets at the barrier below)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = prgb[i][ch][c] * kback;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        prgb[i][ch][c] = v;
      }
  __syncthreads();
  float* s_red = lds;                                   // [WGM * WM tiles][3][BN]
  if (q == 0) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int c = 0; c < 4; ++c) s_red[((wm_i * WM + i) * 3 + ch) * BN + nl + 16 * c] = prgb[i][ch][c];
  }
  __syncthreads();
  constexpr int SLOTS = BM / 64;                        // 64-row slots of this workgroup: four 16-row tiles each, summed in tile order
  static_assert(WGM * WM == 4 * SLOTS, "four tiles per ToRGB slot");
  if (wm_i < SLOTS && q < 3) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (npx[c] >= HW) continue;
      float v = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) v += s_red[((wm_i * 4 + m) * 3 + q) * BN + nl + 16 * c];
      cips3d_store_wt(a.rgb_part + (((int64_t)blockIdx.y * SLOTS + wm_i) * a.B + b) * 3 * HW + ((int64_t)q * HW + npx[c]), v);
    }
  }
}

// fp32 [B][C][HW] -> planes [B][C/8][2][HW][8]; one thread per (channel block, pixel)
// (the sample's maximum, every slot read by the thread itself: the samples of a wave may differ)
__device__ __forceinline__ float amax_of_sample(const float* __restrict__ slots) {
  float m = 0.f;
  for (int s_ = 0; s_ < CIPS3D_AMAX_SLOTS; ++s_) m = fmaxf(m, slots[s_ * CIPS3D_AMAX_STRIDE]);
  return m;
}

__global__ void __launch_bounds__(256) to_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ p, int B, int C,
                                                        int HW, const float* __restrict__ x_amax, int* __restrict__ exp_out,
                                                        float* __restrict__ pmax_out) {
  const int64_t total = (int64_t)B * (C / 8) * HW, per = (int64_t)(C / 8) * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % HW);
    const int64_t bc = i / HW;               // b * (C/8) + cb
    float k = 1.f;
    if (x_amax) {                            // planes of x * 2^-e, the sample's maximum just below 2^15 (cips3d_range); every
      const int b = (int)(i / per);          // pixel block of the sample gets that exponent
      const int e_ = cips3d_split_exp(amax_of_sample(x_amax + (int64_t)b * CIPS3D_AMAX_FLOATS));
      k = cips3d_pow2(-e_);
      const int nblk = (HW + CIPS3D_PLANES_EXP_BLOCK - 1) / CIPS3D_PLANES_EXP_BLOCK;
      if (i - b * per < nblk) exp_out[b * nblk + (int)(i - b * per)] = e_;
      if (pmax_out) {                        // every (16 channels x 64 pixels) patch gets the sample's maximum
        const int n_pm = ((HW + 63) / 64) * (C >> 4);
        const float am = amax_of_sample(x_amax + (int64_t)b * CIPS3D_AMAX_FLOATS);
        for (int64_t j = i - b * per; j < n_pm; j += per) pmax_out[(int64_t)b * n_pm + j] = am;
      }
    }
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = x[(bc * 8 + e) * HW + n] * k;
      _Float16 h_, l_;
      cips3d_split16(v, h_, l_);
      hi[e] = h_;
      lo[e] = l_;
    }
    *reinterpret_cast<h8*>(p + ((bc * 2) * HW + n) * 8) = hi;
    *reinterpret_cast<h8*>(p + ((bc * 2 + 1) * HW + n) * 8) = lo;
  }
}

__global__ void __launch_bounds__(256) from_planes_kernel(const _Float16* __restrict__ p, float* __restrict__ x, int B, int C,
                                                          int HW, const int* __restrict__ exps) {
  const int64_t total = (int64_t)B * (C / 8) * HW, per = (int64_t)(C / 8) * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % HW);
    const int64_t bc = i / HW;
    const float k = exps ? cips3d_pow2(exps[(i / per) * ((HW + CIPS3D_PLANES_EXP_BLOCK - 1) / CIPS3D_PLANES_EXP_BLOCK) +
                                            n / CIPS3D_PLANES_EXP_BLOCK]) : 1.f;
    const h8 hi = *reinterpret_cast<const h8*>(p + ((bc * 2) * HW + n) * 8);
    const h8 lo = *reinterpret_cast<const h8*>(p + ((bc * 2 + 1) * HW + n) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[(bc * 8 + e) * HW + n] = ((float)hi[e] + (float)lo[e]) * k;
  }
}

// fp32 [B][C][HW] -> planes16 [B][C/8][HW][8] (bf16, round to nearest even) and back (exact)
__global__ void __launch_bounds__(256) to_planes16_kernel(const float* __restrict__ x, unsigned short* __restrict__ p, int B,
                                                          int C, int HW) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int64_t total = (int64_t)B * (C / 8) * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % HW);
    const int64_t bc = i / HW;               // b * (C/8) + cb
    u32x4_t w;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const bf16x2_t pr = __builtin_convertvector(f32x2_t{x[(bc * 8 + e) * HW + n], x[(bc * 8 + e + 1) * HW + n]}, bf16x2_t);
      w[e >> 1] = __builtin_bit_cast(unsigned, pr);
    }
    *reinterpret_cast<u32x4_t*>(p + (bc * HW + n) * 8) = w;
  }
}

__global__ void __launch_bounds__(256) from_planes16_kernel(const unsigned short* __restrict__ p, float* __restrict__ x, int B,
                                                            int C, int HW) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const int64_t total = (int64_t)B * (C / 8) * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % HW);
    const int64_t bc = i / HW;
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(p + (bc * HW + n) * 8);
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      x[(bc * 8 + e) * HW + n] = __uint_as_float(w[e >> 1] << 16);
      x[(bc * 8 + e + 1) * HW + n] = __uint_as_float(w[e >> 1] & 0xffff0000u);
    }
  }
}

}  // namespace

extern "C" int cips3d_planes_supported(int Cin, int Cout, int64_t HW) {
  return Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0 && HW >= 16 &&
         (int64_t)(Cin > Cout ? Cin : Cout) * HW * 2 + 1024 < ((int64_t)1 << 31);
}

extern "C" int cips3d_to_planes(const float* x, void* planes, int B, int C, int64_t HW, const float* x_amax, int32_t* exp_out,
                                float* pmax_out, void* stream) {
  if (!x || !planes || B < 0 || C <= 0 || HW <= 0 || ((x_amax == nullptr) != (exp_out == nullptr)) || (pmax_out && !x_amax))
    return CIPS3D_E_BADARG;
  if (pmax_out && C % 16 != 0) return CIPS3D_E_UNSUPP;
  if (C % 8 != 0) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>((int64_t)B * (C / 8) * HW, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(to_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                     reinterpret_cast<_Float16*>(planes), B, C, (int)HW, x_amax, exp_out, pmax_out);
  return cips3d_launch_status();
}

extern "C" int cips3d_from_planes(const void* planes, float* x, int B, int C, int64_t HW, const int32_t* exp, void* stream) {
  if (!x || !planes || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (C % 8 != 0) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>((int64_t)B * (C / 8) * HW, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(from_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const _Float16*>(planes), x, B, C, (int)HW, exp);
  return cips3d_launch_status();
}

extern "C" int cips3d_to_planes16(const float* x, void* planes16, int B, int C, int64_t HW, void* stream) {
  if (!x || !planes16 || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (C % 8 != 0) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>((int64_t)B * (C / 8) * HW, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(to_planes16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                     reinterpret_cast<unsigned short*>(planes16), B, C, (int)HW);
  return cips3d_launch_status();
}

extern "C" int cips3d_from_planes16(const void* planes16, float* x, int B, int C, int64_t HW, void* stream) {
  if (!x || !planes16 || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (C % 8 != 0) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>((int64_t)B * (C / 8) * HW, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(from_planes16_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const unsigned short*>(planes16), x, B, C, (int)HW);
  return cips3d_launch_status();
}

extern "C" int cips3d_modconv1x1_planes16(const void* x_planes16, const float* wm, void* out, int out_format, int B, int Cin,
                                          int Cout, int64_t HW, int epilogue, const float* noise, int64_t noise_bstride,
                                          const float* noise_w, const float* bias, const float* rgb_w, float* rgb_part,
                                          int* n_row_blocks, const cips3d_range* rg, void* stream) {
  if ((rgb_w == nullptr) != (rgb_part == nullptr)) return CIPS3D_E_BADARG;
  static const int cfg = getenv("CIPS3D_CHAIN16_CFG") ? atoi(getenv("CIPS3D_CHAIN16_CFG")) : 0;   // A/B knob
  if (n_row_blocks) *n_row_blocks = Cout > 0 ? Cout / 64 : 0;
  if (!x_planes16 || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if ((out_format != 0 && out_format != 2 && out_format != 3) || (epilogue != 0 && epilogue != 1) || (epilogue == 1 && !bias))
    return CIPS3D_E_BADARG;
  if (!cips3d_planes_supported(Cin, Cout, HW)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  ChainArgs a{reinterpret_cast<const _Float16*>(x_planes16), wm, out, out_format, B, Cin, Cout, (int)HW, epilogue, noise,
              noise_bstride, noise_w, bias, rgb_w, rgb_part, nullptr, 0, nullptr, 0.f, nullptr, rg ? rg->out_amax : nullptr, nullptr,
              nullptr, cips3d_reduce_job{}};
  if (rg && rg->ride) return CIPS3D_E_UNSUPP;       // the riding fold exists in the split-planes kernel only
  // 64 x 128 tiles, 64-deep stages of 24 KB in a 3-slot ring (72 KB: two workgroups per CU).  Same-box sweep (rocprofv3, 512 -> 512
  // at 64^2): batch 1 7.45 us / batch 4 17.4 us; a 2-slot ring 8.0 / 17.5; 128-deep stages 7.4 / 24.2 (one workgroup per CU);
  // 128 x 128 tiles (2/3 of the operand bytes per flop) 9.6 / 19.8; 64 x 64 tiles 7.5 / 19.9.
  dim3 grid((unsigned)ceil_div<int64_t>(HW, 128), (unsigned)(Cout / 64), (unsigned)B);
  if (cfg == 1) hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 64, 2, 1>), grid, dim3(512), 0, as_stream(stream), a);
  else if (cfg == 2) hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 128, 2, 1>), grid, dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 64, 3, 1>), grid, dim3(512), 0, as_stream(stream), a);
  return cips3d_launch_status();
}

extern "C" int cips3d_modconv1x1_planes(const void* x_planes, const float* wm, void* out, int out_format, int B, int Cin,
                                        int Cout, int64_t HW, int epilogue, const float* noise, int64_t noise_bstride,
                                        const float* noise_w, const float* bias, const float* rgb_w, float* rgb_part,
                                        int* n_row_blocks, const cips3d_range* rg, void* stream) {
  if ((rgb_w == nullptr) != (rgb_part == nullptr)) return CIPS3D_E_BADARG;
  if (n_row_blocks) *n_row_blocks = Cout > 0 ? Cout / 64 : 0;
  if (!x_planes || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (out_format < 0 || out_format > 2 || (epilogue != 0 && epilogue != 1) || (epilogue == 1 && !bias)) return CIPS3D_E_BADARG;
  if (!cips3d_planes_supported(Cin, Cout, HW)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  // a planes output needs its bound: the layer's constants and somewhere to leave the exponents
  if (rg && out_format == 1 && (!rg->lconst || !rg->out_exp)) return CIPS3D_E_BADARG;
  ChainArgs a{reinterpret_cast<const _Float16*>(x_planes), wm, out, out_format, B, Cin, Cout, (int)HW, epilogue, noise,
              noise_bstride, noise_w, bias, rgb_w, rgb_part, rg ? rg->x_exp : nullptr, rg ? rg->x_exp_const : 0,
              rg ? rg->x_pmax : nullptr, rg ? rg->x_max_const : 0.f, (rg && out_format == 1) ? rg->lconst : nullptr,
              (rg && out_format != 1) ? rg->out_amax : nullptr, rg ? rg->out_exp : nullptr,
              (rg && out_format == 1) ? rg->out_pmax : nullptr, cips3d_reduce_job{}};
  if (rg && out_format == 1 && rg->x_pmax && Cin > 512) return CIPS3D_E_UNSUPP;
  if (rg && rg->ride) {            // a ToRGB fold riding on this launch: one float4 position per thread of the grid
    const cips3d_reduce_job& j = *rg->ride;
    if (!j.part || !j.out || j.n_slots < 1 || j.n_bias < 0 || j.n_bias > CIPS3D_TORGB_FOLD_MAX || j.n4 <= 0 || j.HW4 <= 0 ||
        j.slot_stride <= 0)
      return CIPS3D_E_BADARG;
    // one extra row of workgroups, 16 positions per wave, at most 48 slots (12 loads per lane)
    if (j.n_slots > 48 || j.n4 > (int64_t)ceil_div<int64_t>(HW, 128) * B * 8 * 16) return CIPS3D_E_UNSUPP;
    a.ride = j;
  }
  // 64 x 128 tiles, eight waves, 64-deep stages, 2-slot ring (96 KB): one workgroup per CU at 512 x 4096
  // rg->half_chip (another view's launches are in flight on another stream: cips3d_forward_io.views_in_flight): 128 x 128 tiles --
  // the launch takes half the CUs and two views' layers run side by side instead of interleaving on every CU (two lanes: 0.298-0.302
  // against 0.303-0.308 ms per view, same box x3; alone on the device such a launch is SLOWER, 16.3 against 11.1 us: only on the hint).
  // The ToRGB slots stay 64-row slots (same bits).
  const int64_t tiles64 = (int64_t)B * (Cout / 64) * ceil_div<int64_t>(HW, 128);
  if (rg && rg->half_chip && Cout % 128 == 0 && tiles64 > 192 && tiles64 <= 256) {
    dim3 grid2((unsigned)ceil_div<int64_t>(HW, 128), (unsigned)(Cout / 128) + (a.ride.part ? 1u : 0u), (unsigned)B);
    hipLaunchKernelGGL((chain_gemm_kernel<2, 4, 2, 64, 2>), grid2, dim3(512), 0, as_stream(stream), a);
    return cips3d_launch_status();
  }
  dim3 grid((unsigned)ceil_div<int64_t>(HW, 128), (unsigned)(Cout / 64) + (a.ride.part ? 1u : 0u), (unsigned)B);
  static const int cfg = getenv("CIPS3D_CHAIN_CFG") ? atoi(getenv("CIPS3D_CHAIN_CFG")) : 0;     // A/B knob (ring depth / stage size)
  // (Round 3, measured and not kept: 128 x 128 tiles -- 2/3 of the operand bytes per flop through the L2 -> LDS path, 256
  // workgroups -- 16.3 us per 512 -> 512 layer against 12.2: twice the MFMAs and fragment reads per wave at the same one
  // workgroup per CU; 32 x 128 tiles / four waves for the 512 -> 256 exit, whose 64 x 128 grid covers half the CUs: 10.4 against
  // 10.0 us.)
  if (cfg == 1) hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 64, 3>), grid, dim3(512), 0, as_stream(stream), a);
  else if (cfg == 2) hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 32, 4>), grid, dim3(512), 0, as_stream(stream), a);
  else hipLaunchKernelGGL((chain_gemm_kernel<1, 4, 2, 64, 2>), grid, dim3(512), 0, as_stream(stream), a);
  return cips3d_launch_status();
}
