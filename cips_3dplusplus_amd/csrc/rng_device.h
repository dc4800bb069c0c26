// Device side of cips3d_rng_fill (csrc/rng.hip): Philox4x32-10 and the per-thread draw, shared with the kernels that host
// slices of a draw beside their own latency-bound work (linear.hip: the mapping chain's launches).
#pragma once
#include "common.h"

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply each: v_mad_u64_u32, a quarter-rate instruction like v_mul_hi / v_mul_lo but one instead of two)
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// thread t of a cips3d_rng_fill call (see rng.hip); callable from any kernel that wants to host a slice of the draw
__device__ __forceinline__ void rng_fill_thread(unsigned seed_lo, unsigned seed_hi, unsigned long long base,
                                                float* __restrict__ normal, long long n_normal,
                                                float* __restrict__ uniform, long long n_uniform, long long t) {
  const long long qn = (n_normal + 3) >> 2, qu = (n_uniform + 3) >> 2;
  if (t >= qn + qu) return;
  const unsigned long long idx = base + (unsigned long long)t;
  unsigned w[4];
  philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), 0x43495053u, 0u, seed_lo, seed_hi, w);
  float v[4];
  float* dst;
  long long first, n;
  if (t < qn) {
    const float k24 = 5.9604644775390625e-08f;      // 2^-24
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float u = ((float)(w[2 * p] >> 8) + 0.5f) * k24;
      const float a = (float)(w[2 * p + 1] >> 8) * k24;
      // -2 ln u = -2 ln2 log2 u  (v_log_f32); u <= 1 so the radicand is >= 0
      const float r = __builtin_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u));
      v[2 * p] = r * __builtin_amdgcn_sinf(a);
      v[2 * p + 1] = r * __builtin_amdgcn_cosf(a);
    }
    dst = normal; first = 4 * t; n = n_normal;
  } else {
    const float k24 = 5.9604644775390625e-08f;
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = (float)(w[p] >> 8) * k24;
    dst = uniform; first = 4 * (t - qn); n = n_uniform;
  }
  if (first + 4 <= n && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    // (plain stores: non-temporal ones were measured -- the noise then reaches the up-sampling stages from HBM instead of the
    // L2 / infinity cache, C = 32 stage 35.8 -> 38.4 us, 0.3477 -> 0.3520 ms per view on one box)
#ifndef CIPS3D_RNG_WT
#define CIPS3D_RNG_WT 1      // A/B (round 5): the draw leaves as write-through stores (common.h) -- it rides on launches whose end-of-
#endif                       // kernel write-back the next mapping layer waits for
#if CIPS3D_RNG_WT
    cips3d_store_wt16(dst + first, make_float4(v[0], v[1], v[2], v[3]));
#else
    *reinterpret_cast<float4*>(dst + first) = make_float4(v[0], v[1], v[2], v[3]);
#endif
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (first + p < n) dst[first + p] = v[p];
  }
}

