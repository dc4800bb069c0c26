// Shared device/host helpers for the gfx950 kernels of libcips3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cips3d_hip.h"

#define CIPS3D_WAVE 64

static inline int cips3d_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ static inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// arguments of one cips3d_linear call (library-internal; linear.hip / forward.hip)
struct cips3d_linear_args {
  const float* x; int64_t x_stride; const float* W; const float* bias; float* out; int64_t out_stride;
  int B, in_dim, out_dim; float w_scale, b_scale; int pixelnorm, lrelu; float act_gain, out_scale, out_shift;
  const float* trunc_mean; float trunc_psi; int out_repeat; int64_t out_repeat_stride;
};
// a slice [t0, t1) of the threads of one cips3d_rng_fill call, hosted by another launch (rng_device.h)
struct cips3d_rng_job {
  unsigned seed_lo, seed_hi; unsigned long long base;
  float* normal; long long n_normal; float* uniform; long long n_uniform;
  long long t0, t1;
};
int cips3d_linear_pair(const cips3d_linear_args& a, const cips3d_linear_args& b, void* stream,
                       const cips3d_rng_job* job = nullptr);
int cips3d_linear_and_table(const cips3d_linear_args& a, const cips3d_linear_desc* table_dev, int n_desc, int total_rows,
                            void* stream);
// cips3d_linear_table whose launch also zeroes `zero_n` 32-bit words at `zero_ptr` (the forward's range workspace: the amax
// slots every later kernel of the call raises with atomicMax)
int cips3d_linear_table_zero(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int B, float* zero_ptr,
                             int zero_n, void* stream);

// "This pointer is global memory."  A pointer that was itself loaded from memory (a descriptor table's fields) is generic
// to the compiler: its loads and stores become flat_ instructions, which count on lgkmcnt as well as vmcnt and may return
// out of order with respect to both -- every wait behind one is a wait for everything.  (Kernel-argument pointers are
// known to be global already.)
#ifdef CIPS3D_NO_GLOBAL_CAST      /* A/B knob: leave them generic */
template <typename T> __device__ static inline const T* cips3d_g(const T* p) { return p; }
template <typename T> __device__ static inline T* cips3d_g(T* p) { return p; }
#else
template <typename T>
__device__ static inline const __attribute__((address_space(1))) T* cips3d_g(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
template <typename T>
__device__ static inline __attribute__((address_space(1))) T* cips3d_g(T* p) {
  return (__attribute__((address_space(1))) T*)p;
}
#endif

// floor division for possibly negative numerators (b > 0)
__host__ __device__ static inline int floor_div_i(int a, int b) {
  int q = a / b;
  return (q * b > a) ? q - 1 : q;
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// leaky ReLU, slope 0.2: max(v, 0.2 v) is the select (v > 0 ? v : 0.2 v) for every finite v and both zeros, in two
// instructions (v_mul, v_max) instead of three (v_mul, v_cmp, v_cndmask)
__device__ static inline float lrelu02(float v) { return fmaxf(v, v * 0.2f); }

// Index of the table entry owning global row `grow` (row_begin = exclusive prefix sums, ascending): every lane fetches one
// entry's row_begin and a ballot counts the entries that start at or before the row -- one load round trip instead of the
// log2(n) dependent ones of a binary search.  n <= 64 (wave-uniform `grow`).
template <typename Desc>
__device__ static inline int owner_desc(const Desc* __restrict__ table, int n_desc, int grow, int lane) {
  const int rb = lane < n_desc ? table[lane].row_begin : 0x7fffffff;
  const unsigned long long m = __ballot(rb <= grow);
  return __popcll(m) - 1;
}

// sin(x) for the FiLM-SIREN activations.  Arguments reach tens of radians (gamma ~ 30), so the
// reduction must be exact: two-constant Cody-Waite with FMA (k*PI_HI is absorbed by the fused
// multiply-add, PI_LO restores the bits PI_HI lacks), then an odd degree-11 polynomial on
// [-pi/2, pi/2].  Max error ~1 ulp of the result for |x| < ~1e5; about 17 VALU ops.
__device__ static inline float sin_accurate(float x) {
  const float INV_PI = 0.318309886183790672f;
  const float PI_HI = 3.14159274101257324f;        // float(pi)
  const float PI_LO = -8.74227765734758577e-8f;    // pi - PI_HI
  float k = rintf(x * INV_PI);
  float r = fmaf(k, -PI_HI, x);
  r = fmaf(k, -PI_LO, r);
  float s = r * r;
  float p = fmaf(s, -2.3889859e-08f, 2.7525562e-06f);
  p = fmaf(p, s, -1.9840874e-04f);
  p = fmaf(p, s, 8.3333310e-03f);
  p = fmaf(p, s, -1.6666667e-01f);
  float rs = r * s;
  float y = fmaf(rs, p, r);
  int ki = (int)k;
  return __int_as_float(__float_as_int(y) ^ (ki << 31));
}

// sin(x) on the hardware sine: exact two-constant Cody-Waite reduction by 2 pi (k * 2PI_HI is absorbed by the FMA, 2PI_LO
// restores the bits it lacks), then v_sin_f32 on r / (2 pi) in [-0.5, 0.5].  6 VALU slots instead of ~15; max error 3.9e-7,
// mean 5.5e-8 over |x| <= 1e3 (tools/sin_probe.py; sin_accurate: 1.2e-7 / 1.6e-8).  In the FiLM-SIREN network the error of a
// layer's output is dominated by gamma (~30) times the fp32 rounding of its pre-activation (~1e-6), so the renderer's
// distance to an fp64 evaluation does not move (tests/test_gpu_split_fp16.py).
__device__ static inline float sin_hw(float x) {
  const float INV_2PI = 0.159154943091895336f;
  const float TWO_PI_HI = 6.28318548202514648f;        // float(2 pi)
  const float TWO_PI_LO = -1.74845553146951715e-7f;    // 2 pi - TWO_PI_HI
  const float k = rintf(x * INV_2PI);
  float r = fmaf(k, -TWO_PI_HI, x);
  r = fmaf(k, -TWO_PI_LO, r);
  return __builtin_amdgcn_sinf(r * INV_2PI);
}
// cos(x), same reduction, v_cos_f32 (the FiLM-sine derivative of the fused NeRF backward)
__device__ static inline float cos_hw(float x) {
  const float INV_2PI = 0.159154943091895336f;
  const float TWO_PI_HI = 6.28318548202514648f;
  const float TWO_PI_LO = -1.74845553146951715e-7f;
  const float k = rintf(x * INV_2PI);
  float r = fmaf(k, -TWO_PI_HI, x);
  r = fmaf(k, -TWO_PI_LO, r);
  return __builtin_amdgcn_cosf(r * INV_2PI);
}
// sin(x) / cos(x) without a reduction of our own (the default since round 2): t = x / (2 pi) in revolutions, v_fract_f32
// (exact) takes it into [0, 1), v_sin_f32 / v_cos_f32 evaluate there -- 3 VALU slots + the quarter-rate instruction instead
// of 6 + it (render kernel 87.1 -> 81.9 us with sin_hw_direct's two, same box).  What it adds to the argument is the
// rounding of the product t (half an ulp of t: 3e-6 rad for |x| in [50, 100], 1.5e-6 in [25, 50]) -- the size of the
// rounding the reference's own fp32 evaluation of gamma * pre + beta makes twice.  Measured on the D = 8 renderer against
// fp64: features 5.93e-5 with it, 5.80e-5 with the exact reduction, 5.84e-5 for the fp32 oracle itself
// (tests/test_gpu_split_fp16.py::test_split_nerf_is_as_accurate_as_fp32).  The fract keeps every magnitude inside the
// instruction's domain (|t| < 256 without it).
__device__ static inline float sin_hw_direct(float x) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.159154943091895336f));
}
__device__ static inline float cos_hw_direct(float x) {
  return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * 0.159154943091895336f));
}
// sin(2 pi t): the argument already in revolutions (render kernel with CIPS3D_FILM_REVOLUTIONS: 1 / 2 pi folded into the
// staged FiLM table)
__device__ static inline float sin_revolutions(float t) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t)); }
#ifdef CIPS3D_EXACT_SINE
#define cips3d_sin sin_accurate
#define cips3d_cos cosf
#elif defined(CIPS3D_REDUCED_SINE)
#define cips3d_sin sin_hw
#define cips3d_cos cos_hw
#else
#define cips3d_sin sin_hw_direct
#define cips3d_cos cos_hw_direct
#endif

// Split-fp16: x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant bits; see nerf.hip / decoder.hip / chain.hip).
// `x` is made opaque first: hipcc otherwise folds a multiplication that produced x into the conversion (v_fma_mixlo_f16:
// fp16 of the exactly-rounded-once product) while the residual is taken against the fp32-rounded x -- when the two roundings
// of hi disagree (rare) the halves miss x by a whole fp16 ulp (2^-11 relative; measured 2.5e-4 on the fused stage's output).
__device__ static inline void cips3d_split16(float x, _Float16& hi, _Float16& lo) {
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// The same split in two instructions per value.  v_fma_mix{lo,hi}_f16 evaluate fma(a, b, c) on a free mix of fp32 and fp16
// sources and round ONCE to fp16 into the low / high half of the destination:
//   cips3d_split_word(t, k)   {hi | lo << 16} with hi = fp16(t k), lo = fp16(t k - hi) of the EXACT product t k -- the
//                             multiplication by a scale (a power of two of the range tracking, the sqrt(2) of an activation)
//                             rides along, and the pair represents t k itself, not its fp32 rounding, to 22 bits.  This is
//                             the word format of the fused up-sampling stages' LDS hand-off.
//   cips3d_split_pair(a, b)   [hi_a | hi_b << 16] and [lo_a | lo_b << 16]: one v_cvt_pk_f16_f32 + two mixes for two values,
//                             the register format of the MFMA fragments (chain.hip, nerf.hip).
// (cips3d_split16's form costs cvt, cvt back, sub, cvt and a pack per value.  Inline asm: VALU results feeding VALU, no
// hazard state to keep; the consumers -- LDS writes, MFMA fragments -- see ordinary compiler-tracked registers.)
__device__ static inline unsigned cips3d_split_word(float t, float k) {      // k wave-uniform
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "=&v"(d) : "v"(t), "s"(k));
  return d;
}
__device__ static inline unsigned cips3d_split_word(float t) {
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, 0\n\tv_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "=&v"(d) : "v"(t));
  return d;
}
__device__ static inline void cips3d_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b));
}

// eight values -> the hi / lo fragments of v_mfma_f32_16x16x32_f16 (element j = value j)
typedef _Float16 cips3d_h8 __attribute__((ext_vector_type(8)));
__device__ static inline void cips3d_split8(const float (&v)[8], cips3d_h8& hi, cips3d_h8& lo) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) cips3d_split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
  hi = __builtin_bit_cast(cips3d_h8, u32x4_t{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(cips3d_h8, u32x4_t{l[0], l[1], l[2], l[3]});
}

// ---- Range tracking of the decoder's split-fp16 operands (include/cips3d_hip.h: cips3d_range).
// fp16 has 5 exponent bits: an unscaled pair (hi, lo) overflows at |x| >= 65520 and loses fp32's relative accuracy below
// |x| ~ 2^-3 (lo turns subnormal; the pair's absolute floor is 2^-25).  The reference's fp32 convolution
// (models/model_v3.py:296-312) has neither limit, so every decoder activation is split as x * 2^-e with a power of two per
// (tensor, sample) that puts a RIGOROUS bound of max|x| just below 2^15; the consumer multiplies its accumulators by 2^e
// (exact).  The bound of a tensor a kernel produces itself comes from the measured maximum of the kernel's input (amax
// slots, raised with atomicMax by the producing epilogue) and the layer's constants (lconst): |out| <= c1 * max|in| + c0.
// With max|x| within ~2^6 of the bound the pair's floor sits >= 34 bits below max|x| -- under fp32's own accumulation error
// whatever the magnitude of the data is.
__device__ static inline float cips3d_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }   // e in [-126, 127]
// e with bound * 2^-e in [2^14, 2^15): bound = m 2^k (m in [1, 2)) -> e = k - 14; zero / subnormal / inf / nan bounds clamp
__device__ static inline int cips3d_split_exp(float bound) {
  int e = (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127 - 14;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
__device__ static inline float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// max over the CIPS3D_AMAX_SLOTS slots of one sample, as a wave-uniform value.  `slots` must be a wave-uniform address: the
// sixteen loads are then scalar loads (no vector register, no vmcnt slot -- the counted waits of the GEMM rings do not see
// them) and the scale factors derived from the result live in SGPRs through the kernel.
__device__ static inline float cips3d_amax_load(const float* __restrict__ slots) {
#ifdef CIPS3D_RANGE_NO_LOAD      // timing-only ablation
  return 1.f;
#endif
  float m = slots[0];
#pragma unroll
  for (int s_ = 1; s_ < CIPS3D_AMAX_SLOTS; ++s_) m = fmaxf(m, slots[s_ * CIPS3D_AMAX_STRIDE]);
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m)));
}
// ---- Write-through stores (round 5).  A kernel's dirty lines leave its XCD's L2 in the end-of-kernel write-back, in front of the
// next launch; a tensor that the NEXT launch reads (on all eight XCDs) can go to memory while the kernel still runs instead: an
// agent-scope relaxed atomic store is a plain store with the sc1 bit (LLVM's gfx942+ memory model), the line stays valid in this
// L2.  Measured on the 64^2 chain (9 planes outputs of 8.4 MB per view): -5.6 us per view, same-box, three interleaved rounds
// (DESIGN.md 5.2); non-temporal stores: no change.  CIPS3D_WT_STORES=0 compiles every helper to the plain store (A/B).
#ifndef CIPS3D_WT_STORES
#define CIPS3D_WT_STORES 1
#endif
__device__ static inline void cips3d_store_wt(float* p, float v) {
#if CIPS3D_WT_STORES
  typedef __attribute__((address_space(1))) float gf32_t;
  __hip_atomic_store(reinterpret_cast<gf32_t*>(reinterpret_cast<uintptr_t>(p)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}
template <class T8>
__device__ static inline void cips3d_store_wt8(void* p, T8 v) {        // any 8-byte value (four halfs, two words)
  static_assert(sizeof(T8) == 8, "8-byte payload");
#if CIPS3D_WT_STORES
  typedef __attribute__((address_space(1))) unsigned long long gu64_t;
  __hip_atomic_store(reinterpret_cast<gu64_t*>(reinterpret_cast<uintptr_t>(p)), __builtin_bit_cast(unsigned long long, v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *reinterpret_cast<T8*>(p) = v;
#endif
}
template <class T16>
__device__ static inline void cips3d_store_wt16(void* p, T16 v) {      // any 16-byte value; p 16-byte aligned
  static_assert(sizeof(T16) == 16, "16-byte payload");
#if CIPS3D_WT_STORES
  typedef float f32x4_t_ __attribute__((ext_vector_type(4)));
  // (no 16-byte atomic store exists: the instruction written out.  The trailing s_nop 1 covers the store-data hazard the compiler
  // pads for its own stores -- a VALU write of the data registers right behind a > 64-bit store)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(__builtin_bit_cast(f32x4_t_, v)) : "memory");
#else
  *reinterpret_cast<T16*>(p) = v;
#endif
}

__device__ static inline float cips3d_uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
// raise slot `slot` of a sample to v (v >= 0; non-negative floats order like their bit patterns).  No return value.
__device__ static inline void cips3d_amax_raise(float* __restrict__ slots, float v, int slot) {
  atomicMax(reinterpret_cast<unsigned*>(slots) + (slot & (CIPS3D_AMAX_SLOTS - 1)) * CIPS3D_AMAX_STRIDE, __float_as_uint(v));
}
// max over the 16 lanes of a DPP row, in every lane of the row (xor 1, xor 2, mirror within 8, mirror within 16), for v >= 0:
// non-negative floats order like their bit patterns, so the maxima are integer ones -- four v_max_u32_dpp, where fmaxf would
// add a canonicalising v_max per operand -- and no LDS round trips
__device__ static inline float cips3d_row16_max(float v) {
  unsigned u = __float_as_uint(v);
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, true));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x141, 0xf, 0xf, true));
  u = max(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x140, 0xf, 0xf, true));
  return __uint_as_float(u);
}
// sum over the 16 lanes of a DPP row, in every lane of the row (the butterfly of cips3d_row16_max)
__device__ static inline float cips3d_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}
__device__ static inline float cips3d_readlanes4_max(float v) {          // max of lanes 0, 16, 32, 48 (wave-uniform), v >= 0
  const unsigned a = max((unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0), (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const unsigned b = max((unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32), (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return __uint_as_float(max(a, b));
}
// Workgroup maximum of v (>= 0) as a wave-uniform value in every wave: DPP row maxima, one LDS word per (wave, row) -- no
// atomics, nothing to initialise -- one barrier, then lane i < 4 n_waves reads word i.  s_part: 4 * n_waves floats of LDS that
// nobody else touches between the call and the next barrier of the caller; n_waves <= 16.  Every wave of the workgroup calls.
__device__ static inline float cips3d_workgroup_max(float v, float* s_part, int wave, int lane, int n_waves) {
  v = cips3d_row16_max(v);
  if ((lane & 15) == 0) s_part[wave * 4 + (lane >> 4)] = v;
  __syncthreads();
  float m = lane < 4 * n_waves ? s_part[lane] : 0.f;
  return cips3d_readlanes4_max(cips3d_row16_max(m));
}
// wave maximum of v (>= 0) as a wave-uniform value: four DPP steps and four readlanes, no LDS
__device__ static inline float cips3d_wave_max_uniform(float v) { return cips3d_readlanes4_max(cips3d_row16_max(v)); }
// `seen`: what cips3d_amax_peek returned for this slot some time ago (0: always raise).  The slot only ever grows, so a
// workgroup whose maximum does not exceed a value the slot already held has nothing to tell it.
__device__ static inline void cips3d_amax_raise_if(float* __restrict__ slots, float m, int slot, float seen = 0.f) {
#ifdef CIPS3D_RANGE_NO_RECORD    // timing-only ablation
  return;
#endif
  if (!(m <= seen)) cips3d_amax_raise(slots, m, slot);          // (a NaN maximum is recorded too)
}
// the slot's value as the device sees it now (an agent-scope load: served past this XCD's L2 copy); a stale answer only costs
// an atomic that was not needed
__device__ static inline float cips3d_amax_peek(const float* __restrict__ slots, int slot) {
#ifdef CIPS3D_RANGE_NO_PEEK
  return 0.f;
#endif
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(slots) + (slot & (CIPS3D_AMAX_SLOTS - 1)) * CIPS3D_AMAX_STRIDE,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
