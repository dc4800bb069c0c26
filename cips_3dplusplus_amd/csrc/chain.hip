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
  __shared__ __attribute__((aligned(16))) float lds[NS * (16 * WM * WGM * BK + BK * 64 * WGN) * NP / 2];
#define CH_BX blockIdx.x
#define CH_BY blockIdx.y
#define CH_BZ blockIdx.z
#define CH_GX gridDim.x
#define CH_SEQ false
#define CH_A0 false
#include "chain_tile_body.h"
#undef CH_BX
#undef CH_BY
#undef CH_BZ
#undef CH_GX
#undef CH_SEQ
#undef CH_A0
}

// One output tile (BM output channels x BN pixels of sample blk_z) of one layer as a step of chain_seq_kernel's walk through several
// layers: the same body at an explicit grid position; its side outputs for the next layer (exponent, patch maxima) leave as
// write-through stores like the planes themselves (a consumer on another CU reads them behind the hop's counter).
template <int WM, int WGM, int WGN, int BK, int NS, int NP = 2>
__device__ __forceinline__ void chain_tile(const ChainArgs& a, float* lds, const int blk_x, const int blk_y, const int blk_z,
                                           const int grid_x, const bool a0_resident) {
#define CH_BX blk_x
#define CH_BY blk_y
#define CH_BZ blk_z
#define CH_GX grid_x
#define CH_SEQ true
#define CH_A0 a0_resident
#include "chain_tile_body.h"
#undef CH_BX
#undef CH_BY
#undef CH_BZ
#undef CH_GX
#undef CH_SEQ
#undef CH_A0
}

// the weight pieces of K stage 0 of a layer into ring slot 0 (what chain_tile's stage_load(0) would request for them)
template <int WM, int WGM, int WGN, int BK, int NP = 2>
__device__ __forceinline__ void chain_prefetch_a0(const ChainArgs& a, float* lds, const int blk_y, const int blk_z) {
  constexpr int NW = WGM * WGN, BM = 16 * WM * WGM, BN = 64 * WGN;
  constexpr int A_STAGE = BM * BK * NP / 2, B_STAGE = BK * BN * NP / 2;
  constexpr int A_PIECES = A_STAGE / 256, PIECES = A_PIECES + B_STAGE / 256, PW = PIECES / NW, KQ = (BK / 32) * NP;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int m0 = blk_y * BM, K = a.Cin;
  const float* ab = a.wmp + (int64_t)blk_z * a.Cout * K * NP / 2;
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int piece = j * NW + wave;
    if (piece < A_PIECES) {
      const int ot_l = piece / KQ, kq_l = piece % KQ;
      const char* ub = reinterpret_cast<const char*>(ab + ((((m0 >> 4) + ot_l) * ((K >> 5) * NP) + kq_l) * 256));
      unsigned vo = lane * 16;
      asm volatile("" : "+s"(ub), "+v"(vo));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + vo),
                                       (__attribute__((address_space(3))) void*)(lds + piece * 256), 16, 0, 0);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Several consecutive layers of the run in ONE launch ("measure one hop").  A workgroup keeps its tile
// position (pixel block x, 64-channel row block y, sample) through every layer.  Layer l + 1's tile reads ALL channels of its
// own pixel block -- the tiles (x, 0 .. Cout/64 - 1) of layer l -- and nothing else of the activations, so the dependency is per
// pixel-block COLUMN: one counter per (sample, x), no grid barrier.  Hop (MI355X_MICROARCH.md, inter-workgroup visibility, the form
// with an acquire): every wave drains its write-through stores (s_waitcnt vmcnt(0)), workgroup barrier, one lane adds 1 to the
// column's counter (agent scope); one lane polls it (relaxed, L1-bypassing) until all row blocks of the column have arrived,
// agent-scope acquire fence (this CU's L1 invalidated), wait, barrier, next layer.  The next layer's first weight stage does not
// depend on the hop and is requested in front of the poll.
// Placement: the columns' workgroups are taken from the linear block id so that the Cout/64 workgroups of a column are 8 apart --
// dealt to ONE XCD by the round-robin dispatch (their hand-off stays in that L2) -- and within 64 consecutive ids (progress needs
// 64 co-resident workgroups, not the whole grid).
// `sync`: [B][columns][2] int32, zero before the first launch; the last workgroup out of a column clears its two words again.
// ------------------------------------------------------------------------------------------------
constexpr int CHAIN_SEQ_MAX = 10;
struct ChainSeqArgs {
  ChainArgs layer[CHAIN_SEQ_MAX];
  int n_layers;
  int cols, rows;            // the common grid of the layers: pixel blocks, 64-channel row blocks
  int* sync;
  int* fault;                // set to 1 when a poll gave up (a co-residency assumption failed): the outputs are then garbage
  int prefetch_a;            // != 0: the next layer's first weight stage is requested in front of the poll (A/B)
};

template <int WM, int WGM, int WGN, int BK, int NS>
__global__ void __launch_bounds__(64 * WGM * WGN) chain_seq_kernel(ChainSeqArgs s) {
  constexpr int STAGE_FLOATS = (16 * WM * WGM * BK + BK * 64 * WGN) * 2 / 2;
  __shared__ __attribute__((aligned(16))) float lds[NS * STAGE_FLOATS];
  const int tid = threadIdx.x;
  const int bz = blockIdx.z;
  // linear id -> (column, row): id = 64 g + 8 y + c8  ->  column 8 g + c8, row y   (rows == 8, cols % 8 == 0: checked by the host)
  const int id = blockIdx.x;
  const int bx = 8 * (id >> 6) + (id & 7), by = (id >> 3) & 7;
  int* cnt = s.sync + ((int64_t)bz * s.cols + bx) * 2;
  bool a0 = false;
  for (int l = 0; l < s.n_layers; ++l) {
    chain_tile<WM, WGM, WGN, BK, NS, 2>(s.layer[l], lds, bx, by, bz, s.cols, a0);
    if (l + 1 == s.n_layers) break;
    // ---- hop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's write-through stores have been acknowledged
    __syncthreads();                                     // ... and every wave's; nobody reads the ring any more
    a0 = s.prefetch_a != 0 && !s.layer[l].rgb_part;      // (a ToRGB fold exchanges its partial sums through ring slot 0)
    if (a0) chain_prefetch_a0<WM, WGM, WGN, BK, 2>(s.layer[l + 1], lds, by, bz);
    if (tid == 0) {
      typedef __attribute__((address_space(1))) int gi32_t;
      gi32_t* c = reinterpret_cast<gi32_t*>(reinterpret_cast<uintptr_t>(cnt));
      __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int want = s.rows * (l + 1);
      int spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *s.fault = 1; break; }      // (~1 s: a row block of this column never became resident)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  // ---- leave the column's words zeroed for the next launch: the last of its workgroups out clears them
  if (s.n_layers > 1) {
    __syncthreads();
    if (tid == 0) {
      typedef __attribute__((address_space(1))) int gi32_t;
      gi32_t* c = reinterpret_cast<gi32_t*>(reinterpret_cast<uintptr_t>(cnt));
      if (__hip_atomic_fetch_add(c + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == s.rows - 1) {
        __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
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

// Several consecutive planes layers in one launch (chain_seq_kernel): layers[l + 1].x_planes == layers[l].out etc. is the caller's
// business; every layer is a planes -> planes GEMM of the same grid (Cout == 512 -> 8 row blocks, ceil(HW / 128) % 8 == 0 pixel
// blocks) with range tracking as cips3d_modconv1x1_planes takes it (rg.ride unsupported).
extern "C" int cips3d_modconv1x1_planes_seq(const cips3d_planes_layer* layers, int n_layers, int B, int64_t HW, int32_t* sync,
                                            int32_t* fault, int flags, void* stream) {
  if (!layers || n_layers < 1 || n_layers > CHAIN_SEQ_MAX || B < 0 || HW <= 0 || !sync || !fault) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  ChainSeqArgs s{};
  s.n_layers = n_layers;
  s.cols = (int)ceil_div<int64_t>(HW, 128);
  s.rows = 8;
  s.sync = sync;
  s.fault = fault;
  s.prefetch_a = (flags & 1) ? 1 : 0;
  if (s.cols % 8 != 0) return CIPS3D_E_UNSUPP;
  for (int l = 0; l < n_layers; ++l) {
    const cips3d_planes_layer& L = layers[l];
    if (!L.x_planes || !L.wm || !L.out || L.Cin <= 0 || L.Cout <= 0 || (L.rgb_w == nullptr) != (L.rgb_part == nullptr) ||
        (L.epilogue != 0 && L.epilogue != 1) || (L.epilogue == 1 && !L.bias))
      return CIPS3D_E_BADARG;
    if (L.out_format != 1 || L.Cout != 64 * s.rows || !cips3d_planes_supported(L.Cin, L.Cout, HW) || L.rg.ride) return CIPS3D_E_UNSUPP;
    if (!L.rg.lconst || !L.rg.out_exp) return CIPS3D_E_BADARG;
    if (L.rg.x_pmax && L.Cin > 512) return CIPS3D_E_UNSUPP;
    s.layer[l] = ChainArgs{reinterpret_cast<const _Float16*>(L.x_planes), L.wm, L.out, 1, B, L.Cin, L.Cout, (int)HW, L.epilogue, L.noise,
                           L.noise_bstride, L.noise_w, L.bias, L.rgb_w, L.rgb_part, L.rg.x_exp, L.rg.x_exp_const, L.rg.x_pmax,
                           L.rg.x_max_const, L.rg.lconst, nullptr, L.rg.out_exp, L.rg.out_pmax, cips3d_reduce_job{}};
  }
  dim3 grid((unsigned)(s.cols * s.rows), 1u, (unsigned)B);
  hipLaunchKernelGGL((chain_seq_kernel<1, 4, 2, 64, 2>), grid, dim3(512), 0, as_stream(stream), s);
  return cips3d_launch_status();
}
