// Device helpers shared by the NeRF point-MLP kernels (nerf.hip: forward render; nerf_bwd_fused.hip: recompute + backward):
// operand fragment types, the task shape, the LDS weight ring and the split-fp16 conversion.  See nerf.hip's header for the
// work decomposition and the arithmetic.
#pragma once
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));    // one MFMA 16x16x32 operand fragment (8 fp16 = 4 VGPRs)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

constexpr int RAYS = 16;    // rays per wave task
constexpr int WAVES = 8;    // waves (tasks) per workgroup

// ------------------------------------------------------------------------------------------------
// LDS-DMA of one weight slab (SLAB floats, linear copy, 1 KiB per wave-instruction)
// ------------------------------------------------------------------------------------------------
template <int SLAB>
__device__ __forceinline__ void stage_slab(const float* __restrict__ gsrc, float* lds_dst, int wave, int lane) {
  constexpr int PIECES = SLAB * 4 / 1024;
  constexpr int PER_WAVE = (PIECES + WAVES - 1) / WAVES;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = j * WAVES + wave;
    if (PIECES % WAVES == 0 || piece < PIECES) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(gsrc + piece * 256 + lane * 4),
          (__attribute__((address_space(3))) void*)(lds_dst + piece * 256), 16, 0, 0);
    }
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
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split2(v[j], a, b);
    hi[j] = a;
    lo[j] = b;
  }
}


}  // namespace
