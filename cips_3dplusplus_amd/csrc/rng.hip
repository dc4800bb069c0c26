// Fresh noise of one generator forward in ONE launch: the N(0,1) maps of every NoiseInjection (reference
// models/model_v3.py:334-336: image.new_empty(batch, 1, height, width).normal_()) and the per-ray jitter of the stratified
// sampler (nerf_utils.py:110: torch.rand(B, h, w, 1)).  The reference draws both from torch's global CUDA generator; which
// stream that is, is not part of its contract -- the distributions are.  Through torch the two draws cost 10.0 + 4.6 us of
// a 358 us view (a normal kernel at 1.1 TB/s and a 4096-value launch); here: Philox4x32-10 (Salmon et al., SC'11; the
// Random123 known-answer vectors pin the restatement in oracle/rng.py, which this kernel matches bit for bit), Box-Muller on
// the hardware log2 / sine-in-revolutions, 16-byte stores.  HBM-bound: 4 B written per value.
//
//   thread t < ceil(n_normal / 4):  words w0..w3 = philox(counter = (lo, hi of base + t, 'CIPS', 0), key = seed)
//       normal[4t .. 4t+3] = r0 sin(2 pi a0), r0 cos(2 pi a0), r1 sin(2 pi a1), r1 cos(2 pi a1)
//       r = sqrt(-2 ln u), u = ((w >> 8) + 0.5) 2^-24 in (0, 1],  a = (w' >> 8) 2^-24 revolutions
//   the next ceil(n_uniform / 4) threads: uniform[4j .. 4j+3] = (w >> 8) 2^-24 in [0, 1)
#include "common.h"

namespace {

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

__global__ void __launch_bounds__(256) rng_fill_kernel(unsigned seed_lo, unsigned seed_hi, unsigned long long base,
                                                       float* __restrict__ normal, long long n_normal,
                                                       float* __restrict__ uniform, long long n_uniform) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
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
    *reinterpret_cast<float4*>(dst + first) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (first + p < n) dst[first + p] = v[p];
  }
}

// diagnostic: the raw words of threads 0 .. n-1 (tests pin the integer stream against oracle/rng.py)
__global__ void __launch_bounds__(256) rng_words_kernel(unsigned seed_lo, unsigned seed_hi, unsigned long long base,
                                                        unsigned* __restrict__ out, long long n) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const unsigned long long idx = base + (unsigned long long)t;
  unsigned w[4];
  philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), 0x43495053u, 0u, seed_lo, seed_hi, w);
#pragma unroll
  for (int p = 0; p < 4; ++p) out[4 * t + p] = w[p];
}

}  // namespace

extern "C" int cips3d_rng_fill(uint64_t seed, uint64_t base, float* normal, int64_t n_normal, float* uniform, int64_t n_uniform,
                               void* stream) {
  if (n_normal < 0 || n_uniform < 0 || (n_normal > 0 && !normal) || (n_uniform > 0 && !uniform)) return CIPS3D_E_BADARG;
  const int64_t threads = ((n_normal + 3) >> 2) + ((n_uniform + 3) >> 2);
  if (threads == 0) return 0;
  if (threads > ((int64_t)1 << 38)) return CIPS3D_E_UNSUPP;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)ceil_div<int64_t>(threads, 256)), dim3(256), 0, as_stream(stream),
                     (unsigned)seed, (unsigned)(seed >> 32), (unsigned long long)base, normal, (long long)n_normal, uniform,
                     (long long)n_uniform);
  return cips3d_launch_status();
}

extern "C" int64_t cips3d_rng_fill_threads(int64_t n_normal, int64_t n_uniform) {
  if (n_normal < 0 || n_uniform < 0) return CIPS3D_E_BADARG;
  return ((n_normal + 3) >> 2) + ((n_uniform + 3) >> 2);
}

extern "C" int cips3d_rng_words(uint64_t seed, uint64_t base, uint32_t* out, int64_t n_threads, void* stream) {
  if (n_threads < 0 || (n_threads > 0 && !out)) return CIPS3D_E_BADARG;
  if (n_threads == 0) return 0;
  hipLaunchKernelGGL(rng_words_kernel, dim3((unsigned)ceil_div<int64_t>(n_threads, 256)), dim3(256), 0, as_stream(stream),
                     (unsigned)seed, (unsigned)(seed >> 32), (unsigned long long)base, out, (long long)n_threads);
  return cips3d_launch_status();
}
