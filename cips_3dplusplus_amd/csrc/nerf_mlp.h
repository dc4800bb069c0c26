// Device helpers shared by the NeRF point-MLP kernels (nerf.hip: forward render; nerf_bwd_fused.hip: recompute + backward):
// operand fragment types, the task shape, the LDS weight ring and the split-fp16 conversion.  See nerf.hip's header for the
// work decomposition and the arithmetic.
#pragma once
#include "common.h"

// CIPS3D_FILM_REVOLUTIONS: the staged FiLM table carries gamma' / 2 pi and c / 2 pi, so that the epilogue's FMA yields the
// sine argument in revolutions and v_sin_f32(fract(.)) takes it as it is -- one multiplication less per activation.  The extra
// rounding (of gamma' / 2 pi, once per table entry) perturbs the argument by |x| 2^-24 relative, the size of an ulp of gamma.
#ifndef CIPS3D_FILM_REVOLUTIONS
#define CIPS3D_FILM_REVOLUTIONS 1
#endif
#if CIPS3D_FILM_REVOLUTIONS && !defined(CIPS3D_EXACT_SINE) && !defined(CIPS3D_REDUCED_SINE)
#define FILM_UNIT 0.159154943091895336f
#define FILM_SIN sin_revolutions
#else
#define FILM_UNIT 1.f
#define FILM_SIN cips3d_sin
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));    // one MFMA 16x16x32 operand fragment (8 fp16 = 4 VGPRs)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int RAYS = 16;    // rays per wave task
constexpr int WAVES = 8;    // waves (tasks) per workgroup

struct NerfArgs {
  cips3d_nerf_params p;
  int groups;          // ray groups of 16 per view
  int tasks_per_view;  // groups * n_chunks rounded up to a multiple of WAVES
  int chunk;           // samples per chunk (uniform trip count)
  int fuse_finish;     // the workgroup's eight chunk waves combine their partials in LDS and write the final maps
  int pad_;             // (keeps the 8-byte alignment of what follows explicit)
  float t_end, t_step; // torch.linspace(0, 1 - 1/N, N): last value and step, computed on the host (kernel arguments are
                       // re-readable scalars; computed in the kernel they ended up as spilled VGPR copies)
};

// LDS floats of the render kernel: slab ring (or the 8 x 16 x H partial exchange of the fused finish, whichever is
// larger) + per-view tables (which double as the 8 x 8 x 16 scalar exchange once the last sample is done)
// pitch of one ray's partial in the exchange: H + 4 floats, so that the 16 rays of a wave start 16 bytes apart in the bank
// pattern (at pitch H every ray of a quarter hit the same banks: 8-way conflicts on the writes, 16-way on the combining reads)
__host__ __device__ constexpr int nerf_xf_pitch(int H) { return H + 4; }
__host__ __device__ constexpr int nerf_ring_floats(int H, int TPS, bool fuse) {
  return (fuse && WAVES * RAYS * nerf_xf_pitch(H) > 2 * 16 * H * TPS) ? WAVES * RAYS * nerf_xf_pitch(H) : 2 * 16 * H * TPS;
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA of one weight slab (SLAB floats, linear copy, 1 KiB per wave-instruction)
// ------------------------------------------------------------------------------------------------
// one 1 KiB piece (the wave's 64 lanes x 16 bytes) of a linear copy global -> LDS
__device__ __forceinline__ void stage_piece(const float* __restrict__ gsrc, float* lds_dst, int piece, int lane) {
  // uniform base (SGPR pair, advanced on the scalar unit) + a 32-bit lane offset: the saddr form of the instruction.  With
  // a per-lane 64-bit pointer every piece needed a v_lshl_add_u64 into the same register pair first, and a wave's eight
  // DMA issues of a step ran one behind the other's address arithmetic
  const char* ub = reinterpret_cast<const char*>(gsrc + piece * 256);
  unsigned vo = lane * 16;
  // (both opaque: the optimiser otherwise re-associates the lane offset into the base, or hoists its zero extension out
  // of the block, where instruction selection no longer sees the base + zext(offset) shape the saddr form needs)
  asm volatile("" : "+s"(ub), "+v"(vo));
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)(ub + vo),
      (__attribute__((address_space(3))) void*)(lds_dst + piece * 256), 16, 0, 0);
}
template <int SLAB, int NW = WAVES>
__device__ __forceinline__ void stage_slab(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane) {
  constexpr int PIECES = SLAB * 4 / 1024;
  constexpr int PER_WAVE = (PIECES + NW - 1) / NW;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = j * NW + wave;
    if (PIECES % NW == 0 || piece < PIECES) stage_piece(gsrc, lds_dst, piece, lane);
  }
}

__device__ __forceinline__ float sigmoidf_acc(float v) { return 1.f / (1.f + expf(-v)); }

// Per-wave streaming state of the weight ring.
struct Ring {
  const float* packed;   // global base of the packed stream (one sample's worth, repeated)
  float* lds;            // 2 slots
  int seq;               // slabs consumed so far
  int seq_end;           // total slabs this workgroup will consume
  int per_sample;        // slabs per sample
};

// x = hi + lo with hi = fp16(x), lo = fp16(x - hi): 22 significant bits in the 4 bytes of an fp32
__device__ __forceinline__ void split2(float x, _Float16& hi, _Float16& lo) {
  cips3d_split16(x, hi, lo);
}
// eight fp32 values (units 4q..4q+3 of tile 2m, then of tile 2m+1) -> the hi / lo B fragments of k-block m
// (cips3d_split_pair: one conversion and two v_fma_mix per two values instead of four instructions and a pack per value)
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) cips3d_split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
  hi = __builtin_bit_cast(h8, u32x4_t{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(h8, u32x4_t{l[0], l[1], l[2], l[3]});
}


}  // namespace
