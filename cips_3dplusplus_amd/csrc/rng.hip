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
#include "rng_device.h"

namespace {

__global__ void __launch_bounds__(256) rng_fill_kernel(unsigned seed_lo, unsigned seed_hi, unsigned long long base,
                                                       float* __restrict__ normal, long long n_normal,
                                                       float* __restrict__ uniform, long long n_uniform) {
  rng_fill_thread(seed_lo, seed_hi, base, normal, n_normal, uniform, n_uniform, (long long)blockIdx.x * 256 + threadIdx.x);
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
