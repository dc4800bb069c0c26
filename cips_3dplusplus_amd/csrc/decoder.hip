// StyleGAN2-style decoder of CIPS-3D++ on gfx950 (reference models/model_v3.py:218-341,418-482,592-637).
//
//   cips3d_modulate_weights  per-sample weight modulation + demodulation (model_v3.py:267-278); optionally
//                            emits the matrix in MFMA A-fragment order for the GEMM below
//   cips3d_modconv1x1        the 1x1 ModulatedConv2d as a per-sample GEMM  out[o][n] = sum_i wm[o][i] x[i][n]
//                            on v_mfma_f32_32x32x2_f32, NoiseInjection + FusedLeakyReLU fused in the epilogue
//   cips3d_up2_fir_act       k=1 up-sampling conv = low-res GEMM followed by upfirdn2d(up=2, pad=(2,1)) of
//                            its result (pinned by tests/test_oracle_golden.py), fused with
//                            noise + bias + leaky-ReLU
//   cips3d_torgb             ToRGB (C->3 modulated conv, bias, skip / FIR-upsampled skip add)
//   cips3d_modconv_kxk       direct k x k modulated convolution / stride-2 transpose (generality path)
//
// Rooflines: 64^2 / 128^2 stages are MFMA-bound (2*Cin*Cout flop per pixel against (Cin+Cout)*4 B);
// >= 256^2 stages, the FIR up-sampler and ToRGB are HBM-bound: every kernel reads its input once and
// writes its output once with >= 128-byte row segments.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// ---- split-fp16 inside the fused up-sampling stage (16-channel k-groups, v_mfma_f32_16x16x16_f16): an activation travels
// through LDS as ONE 32-bit word holding its two halves {fp16 hi (bits 0-15), fp16 lo (bits 16-31)} -- the producer splits
// it once -- and a lane assembles its 4-element fragments from four such words with two v_perm_b32 each.  Weights come
// pre-split from the modulate kernel (CIPS3D_MOD_SPLIT16: per o-tile and k-group [lane][hi x4 | lo x4], the 16 bytes of
// the fp32 fragment they replace), scaled by 2^8.
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pack_split(float x) { return __uint_as_float(cips3d_split_word(x)); }
// ... of the exact product x * k (k wave-uniform): the activation's sqrt(2) 2^-e rides on the split, two instructions in all
__device__ __forceinline__ float pack_split(float x, float k) { return __uint_as_float(cips3d_split_word(x, k)); }
// four packed words (fragment elements 0..3) -> the hi fragment and the lo fragment
__device__ __forceinline__ void unpack_frag(float p0, float p1, float p2, float p3, h4& hi, h4& lo) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const unsigned a0 = __float_as_uint(p0), a1 = __float_as_uint(p1), a2 = __float_as_uint(p2), a3 = __float_as_uint(p3);
  hi = __builtin_bit_cast(h4, u32x2_t{__builtin_amdgcn_perm(a1, a0, 0x05040100u), __builtin_amdgcn_perm(a3, a2, 0x05040100u)});
  lo = __builtin_bit_cast(h4, u32x2_t{__builtin_amdgcn_perm(a1, a0, 0x07060302u), __builtin_amdgcn_perm(a3, a2, 0x07060302u)});
}
// split four fp32 values (fragment elements 0..3) in registers
__device__ __forceinline__ void split_frag(float v0, float v1, float v2, float v3, h4& hi, h4& lo) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  unsigned h0, l0, h1, l1;
  cips3d_split_pair(v0, v1, h0, l0);
  cips3d_split_pair(v2, v3, h1, l1);
  hi = __builtin_bit_cast(h4, u32x2_t{h0, h1});
  lo = __builtin_bit_cast(h4, u32x2_t{l0, l1});
}
// the three products of one 16x16x16 block: A fragment word = {hi x4 | lo x4}
__device__ __forceinline__ f32x4 split_mfma16(f32x4 afrag, h4 bh, h4 bl, f32x4 c) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const h4 ah = __builtin_bit_cast(h4, f32x2_t{afrag[0], afrag[1]});
  const h4 al = __builtin_bit_cast(h4, f32x2_t{afrag[2], afrag[3]});
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
}

#ifdef CIPS3D_FUSED_NO_MFMA     // timing-only ablation (tools/): what the fused stages would cost with free matrix work
__device__ __forceinline__ f32x4 fused_mfma(float a, float b, f32x4 c) { c[0] += a * b; return c; }
#else
__device__ __forceinline__ f32x4 fused_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif

// Split-fp16 GEMM mode (CIPS3D_GEMM_SPLIT): fp32-equivalent products at the fp16 matrix rate.  Every operand is the
// unevaluated sum of two fp16 numbers, x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant bits in the 4 bytes
// of an fp32); a product is accumulated in fp32 as the three exact fp16 x fp16 products w_lo x_hi + w_hi x_lo + w_hi x_hi
// on v_mfma_f32_16x16x32_f16 (3 x 16 cycles per 16x16x32 block against 8 x 32 cycles of v_mfma_f32_16x16x4_f32).  The
// dropped term and the representation error are ~2^-22 relative per product, below the rounding fp32 accumulation makes
// on the sum (tools/split_probe.py).  Modulated weights (|wm| <= 1 after demodulation) are pre-scaled by
// kSplitScale = 2^8 into fp16's normal range and split by the modulate kernel; activations are split in registers after
// the LDS read; the epilogue undoes the scale exactly.
constexpr float kSplitScale = 256.f, kSplitInv = 1.f / 256.f;
__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) cips3d_split_pair(v[2 * p], v[2 * p + 1], h[p], l[p]);
  hi = __builtin_bit_cast(h8, u32x4_t{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(h8, u32x4_t{l[0], l[1], l[2], l[3]});
}

// bf16 compute mode of the GEMMs (BASELINE config 3: decoder in bf16 with fp32 accumulate).  Storage and data movement
// stay fp32 and identical; only the fragments are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) in registers and fed to
// v_mfma_f32_16x16x16_bf16.  The packed A layout gives lane (o, q) the channels 16kq + 4e + q (e = 0..3) of row o and the
// B-fragment reads give lane (n, q) the same channels of pixel n, so element e of both operands is MFMA k = 4q + e:
// one bf16 MFMA replaces the four fp32 MFMAs of a k-group.
// Two-wide vector conversions compile to one v_cvt_pk_bf16_f32 per pair AND keep the compiler's hazard bookkeeping
// (an inline-asm cvt feeding an MFMA loses the required wait state between the VALU write and the MFMA read: measured
// wrong results).
__device__ __forceinline__ s16x4 pack_bf16(float a0, float a1, float a2, float a3) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t lo = __builtin_convertvector(f32x2_t{a0, a1}, bf16x2_t);
  const bf16x2_t hi = __builtin_convertvector(f32x2_t{a2, a3}, bf16x2_t);
  return __builtin_bit_cast(s16x4, u32x2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)});
}

// ------------------------------------------------------------------------------------------------
// weight modulation.  One wave per (b, o).
//   plain : wm[b][o][i*ksq + t]
//   packed (ksq == 1, Cout % 16 == 0, Cin % 16 == 0): A-fragment order of v_mfma_f32_16x16x4_f32
//           wmp[b][ot][kq][lane][j] = wm[b][ot*16 + (lane&15)][16*kq + 4*j + (lane>>4)]
//   packed + split (ksq == 1, Cin % 32 == 0; CIPS3D_MOD_SPLIT): A fragments of v_mfma_f32_16x16x32_f16, hi and lo fp16 halves of
//           2^8 wm:  wms[b][ot][kb][plane][lane][j] (fp16) = plane(2^8 wm[b][ot*16 + (lane&15)][32 kb + 8 (lane>>4) + j]) -- one
//           o-tile x 32-channel block = 2 x 1 KiB = the bytes of the two fp32 k-groups it replaces (same LDS-DMA pieces)
//   packed + bf16 (ksq == 1, Cin % 32 == 0; CIPS3D_MOD_BF16): A fragments of v_mfma_f32_16x16x32_bf16 in the same positions,
//           one plane of bf16(wm) (round to nearest even), unscaled: wmb[b][ot][kb][lane][j] -- for cips3d_modconv1x1_planes16
//   packed, ksq == 9 (the 3x3 implicit GEMM, conv3x3.hip): the same fragment order per tap, tap-major:
//           wmp[b][t'][ot][kq][lane][j], t' = t, or 8 - t with CIPS3D_MOD_FLIP (the transposed conv of the up-sampling branch)
// ------------------------------------------------------------------------------------------------
// Range constants of a StyledConv (cips3d_range): lconst = {c0, c1, l1, 0} with |out| <= max(c1, sqrt(2) l1) max|in| + c0.
//   c0 = sqrt(2) (|noise_w| noise_bound + max |bias|)       (leaky ReLU * sqrt(2) is 1-Lipschitz * sqrt(2) through 0)
//   c1 = sqrt(2) w_gain; a demodulated row has unit L2 norm, so its L1 norm is <= sqrt(Cin ksq) = w_gain; with a FIR
//        (the conv's GEMM result is up-sampled before the activation; the result's own maximum is measured) the gain is
//        the FIR's largest polyphase L1 norm (1 for [1,3,3,1])
//   l1 : raised by the rows of a non-demodulated conv to their L1 norm (modulate_row); left 0 otherwise
// One wave.
__device__ __forceinline__ void layer_consts(const float* __restrict__ bias, int n_bias, const float* __restrict__ noise_w,
                                             float noise_bound, float w_gain, const float* __restrict__ fir,
                                             float* __restrict__ lconst, int lane) {
  float bm = 0.f;
  if (bias)
    for (int i = lane; i < n_bias; i += 64) bm = fmaxf(bm, fabsf(bias[i]));
  bm = wave_max(bm);
  float gain = w_gain;
  if (fir) {
    gain = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
      float g = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) g += fabsf(fir[(py + 2 * (t >> 1)) * 4 + px + 2 * (t & 1)]);
      gain = fmaxf(gain, g);
    }
  }
  if (lane == 0) {
    const float nwa = noise_w ? fabsf(noise_w[0]) : 0.f;
    lconst[0] = 1.41421356237309515f * (nwa * noise_bound + bm);
    lconst[1] = 1.41421356237309515f * gain;
  }
}

__device__ __forceinline__ void modulate_row(const float* __restrict__ W, const float* __restrict__ sb_,
                                             float* __restrict__ wm_, int b, int o, int Cout, int Cin, int ksq,
                                             float scale, int demod, int packed, int lane, float* __restrict__ l1_out = nullptr) {
  const int len = Cin * ksq;
  const auto* w = cips3d_g(W + (int64_t)o * len);       // (global, whatever the pointers' provenance: see cips3d_g)
  const auto* sb = cips3d_g(sb_);
  auto* wm = cips3d_g(wm_);
  // (A 16-byte-store form of the split / bf16 layouts -- 8 consecutive channels per lane, bit-identical output -- was built
  // and measured: modulate_table_kernel 10.9 us against 10.3 us.  The launch is a chain of dependent latencies (descriptor,
  // row, reduction, store), not store-bound; not kept.)
  constexpr int MAXV = 16;                 // rows up to 1024 values stay in registers between the two passes
  const bool cached = len <= 64 * MAXV;
  float vreg[MAXV];
  float ss = 0.f;
  if (cached) {
#pragma unroll
    for (int m = 0; m < MAXV; ++m) {
      const int e = lane + 64 * m;
      vreg[m] = e < len ? (scale * w[e]) * sb[e / ksq] : 0.f;
      ss = fmaf(vreg[m], vreg[m], ss);
    }
  } else if (demod) {
    for (int e = lane; e < len; e += 64) {
      const float v = (scale * w[e]) * sb[e / ksq];
      ss = fmaf(v, v, ss);
    }
  }
  if (demod) ss = wave_sum(ss);
  const float d = demod ? rsqrtf(ss + 1e-8f) : 1.f;
  if (!demod && l1_out) {     // no unit norm to lean on: the row's L1 norm bounds |sum_i wm_oi x_i| / max|x| (cips3d_range)
    float l1 = 0.f;
    if (cached) {
#pragma unroll
      for (int m = 0; m < MAXV; ++m) l1 += fabsf(vreg[m]);
    } else {
      for (int e = lane; e < len; e += 64) l1 += fabsf((scale * w[e]) * sb[e / ksq]);
    }
    l1 = wave_sum(l1);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(l1_out), __float_as_uint(l1));
  }
  auto put = [&](int e, float v) {
    if (demod) v *= d;
    if (packed & 32) {            // split-fp16 fragments of the fused stages: 16-channel k-groups, [lane][hi x4 | lo x4]
      const int i = e;
      const int ot = o >> 4, kq = i >> 4;
      const bool chained = (packed & 7) == 2;      // element <-> channel as in the fp32 layouts below
      const int el = chained ? (i & 3) : ((i >> 2) & 3), q = chained ? ((i >> 2) & 3) : (i & 3);
      const float sv = v * kSplitScale;
      _Float16 hi, lo;
      cips3d_split16(sv, hi, lo);
      auto* blk = cips3d_g(reinterpret_cast<_Float16*>(wm_)) + ((((int64_t)b * (Cout >> 4) + ot) * (Cin >> 4) + kq) * 512) +
                      ((q << 4) | (o & 15)) * 8;
      blk[el] = hi;
      blk[4 + el] = lo;
    } else if ((packed & 16) && (packed & 64)) {     // split-fp16 fragments of wm^T (ksq == 1): row = input channel, k = output unit
      const int i = e;
      const int j = o & 7, q = (o >> 3) & 3;
      const float sv = v * kSplitScale;
      _Float16 hi, lo;
      cips3d_split16(sv, hi, lo);
      auto* blk = cips3d_g(reinterpret_cast<_Float16*>(wm_)) + ((((int64_t)b * (Cin >> 4) + (i >> 4)) * (Cout >> 5) + (o >> 5)) * 1024);
      blk[((q << 4) | (i & 15)) * 8 + j] = hi;
      blk[512 + ((q << 4) | (i & 15)) * 8 + j] = lo;
    } else if ((packed & 16) && ksq > 1) {     // split-fp16 tap-PAIR fragments of the 3x3 kernel (conv3x3.hip: modconv3x3_split_kernel)
      // [b][pair 5][o-tile][16-channel stage][hi | lo][lane][8]: lane quarter q holds channels 8 (q & 1) .. + 7 of tap 2 pair + (q >> 1)
      const int i = e / ksq;
      const int tap = (packed & 8) ? ksq - 1 - e % ksq : e % ksq;
      const int pr = tap >> 1, hh = tap & 1;
      const int ot = o >> 4, kq = i >> 4, cg = (i >> 3) & 1, j = i & 7;
      const float sv = v * kSplitScale;
      _Float16 hi, lo;
      cips3d_split16(sv, hi, lo);
      auto* blk = cips3d_g(reinterpret_cast<_Float16*>(wm_)) + ((((int64_t)b * 5 + pr) * (Cout >> 4) + ot) * (Cin >> 4) + kq) * 1024;
      const int ln = ((((hh << 1) | cg) << 4) | (o & 15)) * 8 + j;
      blk[ln] = hi;
      blk[512 + ln] = lo;
      if (tap == ksq - 1) {         // the tenth tap of the last pair: zero weights
        const int lz = ((((2 | cg)) << 4) | (o & 15)) * 8 + j;
        blk[lz] = (_Float16)0.f;
        blk[512 + lz] = (_Float16)0.f;
      }
    } else if (packed & 16) {     // split-fp16 fragments (ksq == 1)
      const int i = e;
      const int ot = o >> 4, kb = i >> 5, j = i & 7, q = (i >> 3) & 3;      // natural k order: k = 8 q + j
      const float sv = v * kSplitScale;
      _Float16 hi, lo;
      cips3d_split16(sv, hi, lo);
      auto* blk = cips3d_g(reinterpret_cast<_Float16*>(wm_)) + ((((int64_t)b * (Cout >> 4) + ot) * (Cin >> 5) + kb) * 1024);
      blk[((q << 4) | (o & 15)) * 8 + j] = hi;
      blk[512 + ((q << 4) | (o & 15)) * 8 + j] = lo;
    } else if (packed & 128) {    // bf16 fragments of v_mfma_f32_16x16x32_bf16 (ksq == 1): the split layout's positions, one plane
      const int i = e;
      const int ot = o >> 4, kb = i >> 5, j = i & 7, q = (i >> 3) & 3;
      const __bf16 r = (__bf16)v;                                            // round to nearest even (the bf16 mode's operand rounding)
      auto* blk = cips3d_g(reinterpret_cast<unsigned short*>(wm_)) + ((((int64_t)b * (Cout >> 4) + ot) * (Cin >> 5) + kb) * 512);
      blk[((q << 4) | (o & 15)) * 8 + j] = __builtin_bit_cast(unsigned short, r);
    } else if (packed & 64) {     // fragments of wm^T (ksq == 1): row = input channel, k = output unit (backward.hip:pack_kernel)
      const int i = e;
      wm[(((int64_t)b * (Cin >> 4) + (i >> 4)) * (Cout >> 4) + (o >> 4)) * 256 + (((o & 3) << 4) | (i & 15)) * 4 + ((o >> 2) & 3)] = v;
    } else if (packed) {
      const int i = ksq == 1 ? e : e / ksq;
      const int tap = ksq == 1 ? 0 : ((packed & 8) ? ksq - 1 - e % ksq : e % ksq);
      const int ot = o >> 4, kq = i >> 4;
      // standard: k-step j of the 16-byte piece, lane quarter q = i & 3.  chained (packed == 2): the roles swap, lane quarter
      // (i >> 2) & 3 and k-step i & 3 -- the order of the MFMA D layout (4 consecutive channels per lane)
      const bool chained = (packed & 7) == 2;
      const int j = chained ? (i & 3) : ((i >> 2) & 3), q = chained ? ((i >> 2) & 3) : (i & 3);
      wm[((((int64_t)b * ksq + tap) * (Cout >> 4) + ot) * (Cin >> 4) + kq) * 256 + ((q << 4) | (o & 15)) * 4 + j] = v;
    } else {
      wm[((int64_t)b * Cout + o) * len + e] = v;
    }
  };
  if (cached) {
#pragma unroll
    for (int m = 0; m < MAXV; ++m) {
      const int e = lane + 64 * m;
      if (e < len) put(e, vreg[m]);
    }
  } else {
    for (int e = lane; e < len; e += 64) put(e, (scale * w[e]) * sb[e / ksq]);
  }
}

__global__ void __launch_bounds__(256) modulate_kernel(const float* __restrict__ W, const float* __restrict__ s,
                                                       int64_t s_stride, float* __restrict__ wm, int B, int Cout,
                                                       int Cin, int ksq, float scale, int demod, int packed) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Cout) return;
  const int b = (int)(row / Cout), o = (int)(row % Cout);
  modulate_row(W, s + (int64_t)b * s_stride, wm, b, o, Cout, Cin, ksq, scale, demod, packed, lane);
}

// every conv of the decoder in one launch: grid.x covers the table's rows, grid.y the samples
__global__ void __launch_bounds__(256) modulate_table_kernel(const cips3d_modulate_desc* __restrict__ table, int n_desc,
                                                             int total_rows, float noise_bound) {
  const int lane = threadIdx.x & 63;
  const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grow >= total_rows) return;
  const int b = blockIdx.y;
  int lo;
  if (n_desc <= 64) {
    lo = owner_desc(table, n_desc, grow, lane);
  } else {
    lo = 0;
    int hi = n_desc - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid].row_begin <= grow) lo = mid; else hi = mid - 1;
    }
  }
  const cips3d_modulate_desc d = table[lo];
  if (d.lconst && grow == d.row_begin)
    layer_consts(d.bias, d.n_bias, d.noise_w, noise_bound, sqrtf((float)(d.Cin * d.ksq)), d.fir, d.lconst + b * 4, lane);
  modulate_row(d.W, d.s + (int64_t)b * d.s_stride, d.out, b, grow - d.row_begin, d.Cout, d.Cin, d.ksq, d.scale,
               d.flags & 1, (d.flags & 2) ? (((d.flags & 4) ? 2 : 1) | (d.flags & 248)) : 0, lane,
               d.lconst ? d.lconst + b * 4 + 2 : nullptr);
}

// cips3d_range_consts: the same constants for one layer from the per-op path; cips3d_absmax: amax slots of an existing tensor
__global__ void __launch_bounds__(64) range_consts_kernel(const float* __restrict__ bias, int n_bias, const float* __restrict__ noise_w,
                                                          float noise_bound, const float* __restrict__ noise_amax, float w_gain,
                                                          const float* __restrict__ fir, float* __restrict__ lconst) {
  float* lc = lconst + blockIdx.x * 4;
  if (threadIdx.x == 0) { lc[2] = 0.f; lc[3] = 0.f; }
  if (noise_amax) noise_bound = fmaxf(noise_bound, cips3d_amax_load(noise_amax));
  layer_consts(bias, n_bias, noise_w, noise_bound, w_gain, fir, lc, threadIdx.x);
}

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ amax) {
  const int b = blockIdx.y;
  const float* xb = x + (int64_t)b * n;
  float m = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(xb) & 15) == 0 ? n / 4 : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(xb)[i];
    m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(xb[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) cips3d_amax_raise(amax + (int64_t)b * CIPS3D_AMAX_FLOATS, m, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------------
// 1x1 modulated conv as GEMM on v_mfma_f32_16x16x4_f32.  Workgroup tile BM x BN, BK-deep stages through a
// 2-slot LDS ring filled by LDS-DMA (A: packed fragments, linear copy; B: BK rows of BN pixels).
// A wave owns WM o-tiles (16 rows each) x 64 pixels.  Its four 16-pixel column tiles are INTERLEAVED over
// the pixels (lane j of column tile c holds pixel 4*j + c), so ONE ds_read_b128 per k-step delivers the
// B fragments of all four column tiles (a per-MFMA ds_read_b32 made the kernel LDS-issue bound at one
// wave per SIMD), and the epilogue stores 16 bytes per lane.
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* x; const float* wmp; float* out;
  int B, Cin, Cout; int64_t HW;
  int epilogue; const float* noise; int64_t noise_bstride; const float* noise_w; const float* bias;
  int bf16;                 // GEMM mode: 0 exact fp32, 1 bf16 operands, 2 split-fp16 (pre-split weights)
  int out_bf16;             // the output is stored as bf16 (CIPS3D_Y_BF16: the low-resolution GEMM of an up-sampling stage)
  // optional: the ToRGB that follows this conv, folded into the epilogue.  Every workgroup writes the partial sums of its
  // BM output rows, rgb_part[blockIdx.y][b][3][HW]; cips3d_torgb_reduce adds the row blocks (and layers) in a fixed order.
  const float* rgb_w;       // plain [B][3][Cout] modulated ToRGB weights (no demodulation)
  float* rgb_part;          // [Cout/BM][B][3][HW]
  // range tracking (cips3d_range): split mode splits x * 2^-e, e from the measured maximum of x; any mode records max |out|
  const float* x_amax;      // [B][CIPS3D_AMAX_FLOATS] or NULL (e = 0)
  float* out_amax;          // [B][CIPS3D_AMAX_FLOATS] or NULL
  // activation-backward epilogue (ACTBWD instantiations, cips3d_modconv1x1_actbwd): see cips3d_actbwd in the public header
  cips3d_actbwd ab;
};

// s_waitcnt immediate that waits until at most n vector-memory operations of this wave are outstanding
// (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4; expcnt / lgkmcnt fields left at "no wait")
__device__ __forceinline__ constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

// MODE: 0 exact fp32 MFMA, 1 bf16 operands (CIPS3D_GEMM_BF16), 2 split-fp16 (CIPS3D_GEMM_SPLIT; A pre-split by the modulate kernel)
// ACTBWD: the data-gradient GEMM of the one-call decoder backward.  Its result is the gradient w.r.t. the OUTPUT of the
// previous StyledConv; the epilogue turns it into the gradient w.r.t. that layer's pre-activation -- adds the contribution of
// the ToRGB that read the same tensor (a rank-3 update), applies the leaky-ReLU derivative from the stored output's sign --
// and reduces the bias / noise-weight / ToRGB-weight gradients of its rows (backward.hip: act_bwd_kernel + torgb_bwd_kernel,
// which read and wrote the whole tensor once more each).
template <int WM, int WGM, int WGN, int BK, int NS, int MODE = 0, bool ACTBWD = false>
// (the ACTBWD forms with a 2-slot ring -- short contractions, HBM-bound -- are compiled for two 8-wave workgroups per CU:
// 128 VGPRs, a handful of spills outside the K loop)
__global__ void __launch_bounds__(64 * WGM * WGN, (ACTBWD && NS == 2 && BK == 32 && WGM * WGN == 8) ? 4 : 1) modconv1x1_kernel(GemmArgs a) {
  constexpr bool BF16 = MODE == 1;
  constexpr bool SPLIT = MODE == 2;
  constexpr int NW = WGM * WGN;               // waves per workgroup: 8 = two per SIMD, so one wave's DMA issue,
                                              // waits and fragment reads run under the partner's MFMAs
  constexpr int BM = 16 * WM * WGM, BN = 64 * WGN;
  constexpr int A_STAGE = BM * BK;            // floats
  constexpr int B_STAGE = BK * BN;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int A_PIECES = A_STAGE / 256;     // 1-KiB pieces
  constexpr int B_PIECES = B_STAGE / 256;
  constexpr int PIECES = A_PIECES + B_PIECES;
  constexpr int PW = PIECES / NW;             // LDS-DMA instructions per wave per stage
  constexpr int KQ = BK / 16;                 // A pieces per o-tile per stage
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  static_assert(PIECES % NW == 0, "every wave must issue the same number of DMA pieces (counted vmcnt)");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  // NS-slot ring, NS-1 stages in flight: the LDS-DMA lands ~1.1 us after issue, so ONE 24 KB stage in
  // flight caps a CU at ~22 GB/s (Little's law) -- below what the MFMA pipe consumes (measured: 63 % of
  // peak at any tile shape).  Waits are counted (vmcnt(N) leaves the younger stages in flight) and the
  // barrier is the raw s_barrier: __syncthreads() would drain every outstanding DMA.
  __shared__ __attribute__((aligned(16))) float lds[NS * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm_i = wave / WGN, wn_i = wave % WGN;
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * BM;
  // split mode: the activations are split as x * kx, kx = 2^-e with max|x| 2^-e in [2^14, 2^15) (the measured maximum of x;
  // common.h, cips3d_range); the accumulators come back with kin = 2^-8 2^e
  float kx = 1.f, kin = kSplitInv;
  if constexpr (SPLIT) {
    if (a.x_amax) {
      const int e = cips3d_split_exp(cips3d_amax_load(a.x_amax + b * CIPS3D_AMAX_FLOATS));
      kx = cips3d_uniform(cips3d_pow2(-e));
      kin = cips3d_uniform(kSplitInv * cips3d_pow2(e));
    }
  }
  // per-lane index arithmetic in 32 bits (the host refuses max(Cin, Cout) * HW >= 2^31); 64-bit products only in the
  // workgroup-uniform sample bases
  const int n0 = blockIdx.x * BN;
  const int HW = (int)a.HW;
  const int K = a.Cin;
  const int nstage = K / BK;
  const float* xb = a.x + (int64_t)b * K * HW;
  const float* ab = a.wmp + (int64_t)b * a.Cout * K;   // packed: [ot][kq][256]

  auto stage_load = [&](int st) {
    float* dstA = lds + (st % NS) * STAGE;
    float* dstB = dstA + A_STAGE;
    const int k0 = st * BK;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int piece = j * NW + wave;
      if (piece < A_PIECES) {
        const int ot_l = piece / KQ, kq_l = piece % KQ;
        const float* src = ab + ((((m0 >> 4) + ot_l) * (K >> 4) + (k0 >> 4) + kq_l) * 256 + lane * 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dstA + piece * 256), 16, 0, 0);
      } else {
        const int pb = piece - A_PIECES;
        const int f = pb * 256 + lane * 4;
        const int row = f / BN, col = f % BN;
        int n = n0 + col;
        if (n > HW - 4) n = HW - 4;                       // clamp: those columns are never stored
        const float* src = xb + ((k0 + row) * HW + n);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dstB + pb * 256), 16, 0, 0);
      }
    }
  };

  const int q = lane >> 4, jn = lane & 15;
  // epilogue operands first (registers, ordinary loads, consumed only after the main loop)
  const int ncol = n0 + wn_i * 64 + jn * 4;                // this lane's 4 consecutive pixels
  f32x4 nz4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bias4[WM];
  f32x4 wrgb[WM][3];
  float nw = 0.f;
  f32x4 aby[ACTBWD ? WM : 1][4], abg[3];                     // ACTBWD: the stored output's rows, the ToRGB gradient's pixels
  // With a 2-slot ring the first wait of the loop is vmcnt(0) anyway, so the operand loads go out right AFTER the first
  // stage's DMA (one exposed round trip less per launch).  Deeper rings count their waits in DMA pieces only: there the
  // operands are retired before the prologue.
  constexpr bool OPS_AFTER_PROLOGUE = NS == 2;
  auto load_ops = [&]() {
    if (a.epilogue == 1) {
      if (a.noise && a.noise_w && ncol < HW) {
        nz4 = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)b * a.noise_bstride + ncol);
        nw = a.noise_w[0];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
        bias4[i] = *reinterpret_cast<const f32x4*>(a.bias + m0 + (wm_i * WM + i) * 16 + 4 * q);
    }
    if constexpr (ACTBWD) {
      const bool ok = ncol < HW;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      nz4 = (a.ab.d_noise_w && ok) ? *reinterpret_cast<const f32x4*>(a.noise + (int64_t)b * a.noise_bstride + ncol) : z4;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        abg[ch] = (a.ab.drgb && ok) ? *reinterpret_cast<const f32x4*>(a.ab.drgb + ((int64_t)b * 3 + ch) * HW + ncol) : z4;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          aby[i][r] = ok ? *reinterpret_cast<const f32x4*>(a.ab.y + ((int64_t)b * a.Cout + m0 + (wm_i * WM + i) * 16 + 4 * q + r) * HW + ncol) : z4;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          wrgb[i][ch] = a.ab.drgb ? *reinterpret_cast<const f32x4*>(a.ab.rgb_w + ((int64_t)b * 3 + ch) * a.Cout + m0 + (wm_i * WM + i) * 16 + 4 * q) : z4;
      }
    }
    if (a.rgb_part) {        // the folded ToRGB's weights of this lane's rows (an epilogue-time load would be an exposed round trip)
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          wrgb[i][ch] = *reinterpret_cast<const f32x4*>(a.rgb_w + (int64_t)b * 3 * a.Cout + ch * a.Cout + m0 + (wm_i * WM + i) * 16 + 4 * q);
    }
  };
  if (!OPS_AFTER_PROLOGUE) {
    load_ops();
    // retire them now so that the counted waits below see DMA pieces only
    if (a.epilogue == 1 || ACTBWD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: NS-1 stages in flight
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nstage) stage_load(s0);
  if (OPS_AFTER_PROLOGUE) load_ops();

  for (int st = 0; st < nstage; ++st) {
    // stage st must have landed; stages st+1 .. st+NS-2 (as far as they exist) stay in flight
    int younger = nstage - 1 - st;
    if (younger > NS - 2) younger = NS - 2;
    if (younger >= 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * PW));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(PW));
    else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    __builtin_amdgcn_s_barrier();      // every wave's pieces of stage st landed; everyone is done with stage st-1
    if (st + NS - 1 < nstage) stage_load(st + NS - 1);    // refills the slot stage st-1 was read from
    const float* sA = lds + (st % NS) * STAGE;
    const float* sB = sA + A_STAGE + wn_i * 64 + jn * 4;
    // Issue every fragment read of the stage first (12 x ds_read_b128 for BK = 32, WM = 2), then run the MFMAs
    // back to back behind counted lgkmcnt waits: with one wave per SIMD a read-then-wait per MFMA group
    // leaves the matrix pipe idle for an LDS round trip every 16 MFMAs (measured 64 % busy).
    if constexpr (SPLIT) {
      // 32-channel blocks: hi / lo A fragments (pre-split, two 1-KiB pieces per o-tile and block) and eight B rows
      // (channel 32 kb + 8 q + j = fragment element j of lane quarter q), split into hi / lo fragments in registers.
      constexpr int KB = BK / 32;
      static_assert(BK % 32 == 0, "split mode pairs the 16-channel groups");
      h8 ah[KB][WM], al[KB][WM];
      f32x4 b8[KB][8];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          ah[kb][i] = *reinterpret_cast<const h8*>(sA + ((wm_i * WM + i) * KQ + 2 * kb) * 256 + lane * 4);
          al[kb][i] = *reinterpret_cast<const h8*>(sA + ((wm_i * WM + i) * KQ + 2 * kb + 1) * 256 + lane * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) b8[kb][j] = *reinterpret_cast<const f32x4*>(sB + (kb * 32 + 8 * q + j) * BN);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = b8[kb][j][c] * kx;
          h8 bh, bl;
          split8(v8, bh, bl);
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[kb][i], bh, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb][i], bl, acc[i][c], 0, 0, 0);
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[kb][i], bh, acc[i][c], 0, 0, 0);
          }
        }
    } else {
    f32x4 afr[KQ][WM], bfr[KQ][4];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
        afr[kq][i] = *reinterpret_cast<const f32x4*>(sA + ((wm_i * WM + i) * KQ + kq) * 256 + lane * 4);
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)   // k row 16*kq + 4*j4 + q, pixels 4*jn..4*jn+3 = B fragments of the 4 column tiles
        bfr[kq][j4] = *reinterpret_cast<const f32x4*>(sB + (kq * 16 + j4 * 4 + q) * BN);
    }
    __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks each read back in front of its first MFMA
    if (BF16) {
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) {
        s16x4 ah[WM], bh[4];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = pack_bf16(afr[kq][i][0], afr[kq][i][1], afr[kq][i][2], afr[kq][i][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[c] = pack_bf16(bfr[kq][0][c], bfr[kq][1][c], bfr[kq][2][c], bfr[kq][3][c]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bh[c], acc[i][c], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[kq][i][j4], bfr[kq][j4][c], acc[i][c], 0, 0, 0);
    }
    }
    // all LDS reads of this stage retired before the slot can be refilled after the next barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- epilogue.  D layout: acc[i][c][r] = out[o = obase + 4*q + r][pixel ncol + c]
  const bool col_ok = ncol < HW;
  if (!ACTBWD && !col_ok && !a.rgb_part && !a.out_amax) return;
  float* ob = a.out + (int64_t)b * a.Cout * HW + ncol;
  float mx = 0.f;
  float rs[ACTBWD ? WM : 1][4][5];             // ACTBWD row sums of this lane's 4 pixels: bias, noise weight, ToRGB weight x 3
  float prgb[3][4];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int c = 0; c < 4; ++c) prgb[ch][c] = 0.f;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int obase = m0 + (wm_i * WM + i) * 16 + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      if constexpr (SPLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= kin;                 // exact: the weights carried 2^8, the activations 2^-e
      }
      if (a.epilogue == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = lrelu02(fmaf(nw, nz4[c], v[c]) + bias4[i][r]) * 1.41421356237309515f;
      }
      if constexpr (ACTBWD) {
        const f32x4 yv = aby[i][r];
        float sb = 0.f, sn = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float g = v[c];
          g = fmaf(wrgb[i][0][r], abg[0][c], g);
          g = fmaf(wrgb[i][1][r], abg[1][c], g);
          g = fmaf(wrgb[i][2][r], abg[2][c], g);
          g *= yv[c] > 0.f ? 1.41421356237309515f : 0.2f * 1.41421356237309515f;
          if (!col_ok) g = 0.f;                      // (clamped columns past the image)
          v[c] = g;
          sb += g;
          sn = fmaf(g, nz4[c], sn);
          s0 = fmaf(abg[0][c], yv[c], s0); s1 = fmaf(abg[1][c], yv[c], s1); s2 = fmaf(abg[2][c], yv[c], s2);
        }
        rs[i][r][0] = sb; rs[i][r][1] = sn; rs[i][r][2] = s0; rs[i][r][3] = s1; rs[i][r][4] = s2;
      }
      if (col_ok) {
        mx = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), mx);
        if (a.out_bf16)
          cips3d_store_wt8(reinterpret_cast<unsigned short*>(a.out) + (int64_t)b * a.Cout * HW + ncol + (obase + r) * HW,
                           pack_bf16(v[0], v[1], v[2], v[3]));
        else
          cips3d_store_wt16(ob + (obase + r) * HW, v);                // (write-through: common.h)
      }
      if (a.rgb_part) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float wc = wrgb[i][ch][r];
#pragma unroll
          for (int c = 0; c < 4; ++c) prgb[ch][c] = fmaf(wc, v[c], prgb[ch][c]);
        }
      }
    }
  }
  if (a.out_amax) {        // the workgroup's largest |out| raises one slot of the sample's amax array (cips3d_range); the LDS
    // words live in the ring slot the last K stage did not use (a __shared__ array of their own would move the ring)
    const float m = cips3d_workgroup_max(mx, lds + (nstage % NS) * STAGE, wave, lane, NW);
    if (tid == 0) cips3d_amax_raise_if(a.out_amax + b * CIPS3D_AMAX_FLOATS, m, blockIdx.y * gridDim.x + blockIdx.x);
  }
  if constexpr (ACTBWD) {
    // row sums: over the 16 lanes of a DPP row (the 64 pixels of this wave), over the WGN wave columns through LDS, then
    // ONE atomic per (row, quantity) and workgroup, issued as contiguous wave-wide instructions
    __syncthreads();                                  // (every wave is past its last fragment read: the ring is free)
    float* s_rs = lds;                                // [5][WGN][BM]
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const float t = cips3d_row16_sum(rs[i][r][k]);
          if (jn == 0) s_rs[(k * WGN + wn_i) * BM + (wm_i * WM + i) * 16 + 4 * q + r] = t;
        }
    __syncthreads();
    for (int idx = tid; idx < 5 * BM; idx += 64 * NW) {
      const int k = idx / BM, row = idx % BM;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WGN; ++w) t += s_rs[(k * WGN + w) * BM + row];
      const int slot = a.ab.slots > 1 ? (int)(blockIdx.x & (a.ab.slots - 1)) : 0;
      float* dst = k == 0 ? a.ab.d_bias : (k == 1 ? a.ab.d_noise_w : (a.ab.d_rgb_w ? a.ab.d_rgb_w + ((int64_t)b * 3 + (k - 2)) * a.Cout : nullptr));
      if (dst) unsafeAtomicAdd(dst + (int64_t)slot * (k < 2 ? a.ab.slot_stride : a.ab.rgb_slot_stride) + m0 + row, t);
    }
    return;
  }
  if (!a.rgb_part) return;
  // ---- ToRGB partial of this workgroup's BM rows: over the 4 lane quarters by shuffles, over the WGM wave rows through
  // LDS (the ring is free: every wave passed the last stage's lgkmcnt(0) and meets at the barrier below)
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = prgb[ch][c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      prgb[ch][c] = v;
    }
  __syncthreads();
  float* s_red = lds;                                   // [WGM][3][BN]
  const int nloc = wn_i * 64 + jn * 4;
  if (q == 0) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      *reinterpret_cast<f32x4*>(s_red + (wm_i * 3 + ch) * BN + nloc) = f32x4{prgb[ch][0], prgb[ch][1], prgb[ch][2], prgb[ch][3]};
  }
  __syncthreads();
  if (wm_i == 0 && q < 3 && col_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < WGM; ++m) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(s_red + (m * 3 + q) * BN + nloc);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += t[c];
    }
    *reinterpret_cast<f32x4*>(a.rgb_part + ((int64_t)blockIdx.y * a.B + b) * 3 * HW + (q * HW + ncol)) = v;
  }
}

template <int WM, int WGM, int WGN, int BK, int NS>
int launch_gemm_actbwd(const GemmArgs& a, hipStream_t st) {
  constexpr int BM = 16 * WM * WGM, BN = 64 * WGN;
  dim3 grid((unsigned)ceil_div<int64_t>(a.HW, BN), (unsigned)(a.Cout / BM), (unsigned)a.B);
  if (a.bf16 == 2) hipLaunchKernelGGL((modconv1x1_kernel<WM, WGM, WGN, BK, NS, 2, true>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else hipLaunchKernelGGL((modconv1x1_kernel<WM, WGM, WGN, BK, NS, 0, true>), grid, dim3(64 * WGM * WGN), 0, st, a);
  return cips3d_launch_status();
}

template <int WM, int WGM, int WGN, int BK, int NS>
int launch_gemm(const GemmArgs& a, hipStream_t st) {
  constexpr int BM = 16 * WM * WGM, BN = 64 * WGN;
  dim3 grid((unsigned)ceil_div<int64_t>(a.HW, BN), (unsigned)(a.Cout / BM), (unsigned)a.B);
  if (a.bf16 == 2) hipLaunchKernelGGL((modconv1x1_kernel<WM, WGM, WGN, BK, NS, 2>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else if (a.bf16) hipLaunchKernelGGL((modconv1x1_kernel<WM, WGM, WGN, BK, NS, 1>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else hipLaunchKernelGGL((modconv1x1_kernel<WM, WGM, WGN, BK, NS, 0>), grid, dim3(64 * WGM * WGN), 0, st, a);
  return cips3d_launch_status();
}

// ------------------------------------------------------------------------------------------------
// 2x polyphase FIR up-sampler (upfirdn2d up=2, pad=(2,1), 4x4 taps) + noise + bias + leaky-ReLU.
// Each thread produces 4 consecutive output pixels of one row (16-byte store).
// ------------------------------------------------------------------------------------------------
// single tap set for one output pixel; taps come from memory (runtime parity => no register array)
__device__ __forceinline__ float up2_tap(const float* __restrict__ src, int H, int W, int oy, int ox,
                                         const float* __restrict__ fir) {
  // u[y][x] = in[(y-2)/2][(x-2)/2] on even (y-2),(x-2); out = sum_{ky,kx} u[oy+ky][ox+kx] * fir[3-ky][3-kx]
  const int ky0 = oy & 1, kx0 = ox & 1;        // first tap with (oy+ky-2) even
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ky = ky0 + 2 * a;
    const int iy = (oy + ky - 2) >> 1;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int kx = kx0 + 2 * c;
      const int ix = (ox + kx - 2) >> 1;
      if (ix < 0 || ix >= W) continue;
      acc = fmaf(src[(int64_t)iy * W + ix], fir[15 - (ky * 4 + kx)], acc);
    }
  }
  return acc;
}

// 3 x 4 input patch (rows iy-1..iy+1, columns 2*qx-1..2*qx+2, zero outside the image) of the 2 x 4 output block
// at rows 2*iy, 2*iy+1, columns 4*qx .. 4*qx+3
// (interior variant: the whole patch is inside the image, no predication)
__device__ __forceinline__ void up2_load_interior(const float* __restrict__ src, int W, int iy, int qx, float (&v)[3][4]) {
  const float* row = src + ((iy - 1) * W + 2 * qx);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float2 mid = *reinterpret_cast<const float2*>(row);
    v[r][0] = row[-1]; v[r][1] = mid.x; v[r][2] = mid.y; v[r][3] = row[2];
    row += W;
  }
}

// the same patches from a bf16 array (CIPS3D_Y_BF16: the low-resolution GEMM result stored as bf16): a bf16 is the upper
// half of the fp32 with the same value, the middle pair is one aligned 32-bit load
typedef unsigned short bf16_t;
__device__ __forceinline__ float bf16_f32(unsigned bits) { return __uint_as_float(bits << 16); }
__device__ __forceinline__ void up2_load_interior(const bf16_t* __restrict__ src, int W, int iy, int qx, float (&v)[3][4]) {
  const bf16_t* row = src + ((iy - 1) * W + 2 * qx);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const unsigned mid = *reinterpret_cast<const unsigned*>(row);
    v[r][0] = bf16_f32(row[-1]); v[r][1] = bf16_f32(mid & 0xffffu); v[r][2] = __uint_as_float(mid & 0xffff0000u);
    v[r][3] = bf16_f32(row[2]);
    row += W;
  }
}
__device__ __forceinline__ void up2_load(const bf16_t* __restrict__ src, int H, int W, int iy, int qx, float (&v)[3][4]) {
  const int c = 2 * qx;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int y = iy - 1 + r;
    const bool yok = (y >= 0) && (y < H);
    const bf16_t* row = src + (yok ? y : 0) * W;
    unsigned mid = 0u;
    if (yok) mid = *reinterpret_cast<const unsigned*>(row + c);
    v[r][1] = bf16_f32(mid & 0xffffu); v[r][2] = __uint_as_float(mid & 0xffff0000u);
    v[r][0] = (yok && c - 1 >= 0) ? bf16_f32(row[c - 1]) : 0.f;
    v[r][3] = (yok && c + 2 < W) ? bf16_f32(row[c + 2]) : 0.f;
  }
}

__device__ __forceinline__ void up2_load(const float* __restrict__ src, int H, int W, int iy, int qx, float (&v)[3][4]) {
  const int c = 2 * qx;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int y = iy - 1 + r;
    const bool yok = (y >= 0) && (y < H);
    const float* row = src + (yok ? y : 0) * W;
    float2 mid = make_float2(0.f, 0.f);           // middle pair is 8-byte aligned (c even, W even)
    if (yok) mid = *reinterpret_cast<const float2*>(row + c);
    v[r][1] = mid.x; v[r][2] = mid.y;
    v[r][0] = (yok && c - 1 >= 0) ? row[c - 1] : 0.f;
    v[r][3] = (yok && c + 2 < W) ? row[c + 2] : 0.f;
  }
}

// FLAT stages (no up-sampling): the 2 x 4 block itself, rows y, y + 1, columns x .. x + 3 (x % 4 == 0), into rows 0 / 1 of the set
__device__ __forceinline__ void flat_load(const float* __restrict__ src, int W, int y, int x, float (&v)[3][4]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(src + ((y + r) * W + x));
#pragma unroll
    for (int c = 0; c < 4; ++c) v[r][c] = t[c];
  }
}
__device__ __forceinline__ void flat_load(const bf16_t* __restrict__ src, int W, int y, int x, float (&v)[3][4]) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint2 t = *reinterpret_cast<const uint2*>(src + ((y + r) * W + x));
    v[r][0] = bf16_f32(t.x & 0xffffu); v[r][1] = __uint_as_float(t.x & 0xffff0000u);
    v[r][2] = bf16_f32(t.y & 0xffffu); v[r][3] = __uint_as_float(t.y & 0xffff0000u);
  }
}

// polyphase FIR of the patch; all tap indices are compile-time constants
__device__ __forceinline__ void up2_fir(const float (&v)[3][4], const float (&kf)[16], float (&o)[2][4]) {
#pragma unroll
  for (int py = 0; py < 2; ++py) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int px = j & 1;
      const int col0 = (j + 1) >> 1;                 // j=0 -> v0,v1 ; j=1,2 -> v1,v2 ; j=3 -> v2,v3
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int ky = py + 2 * a;
        const int r = py + a;                        // py=0: rows iy-1, iy ; py=1: rows iy, iy+1
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) acc = fmaf(v[r][col0 + bb], kf[ky * 4 + px + 2 * bb], acc);
      }
      o[py][j] = acc;
    }
  }
}

__device__ __forceinline__ void up2_block(const float* __restrict__ src, int H, int W, int iy, int qx,
                                          const float (&kf)[16], float (&o)[2][4]) {
  float v[3][4];
  up2_load(src, H, W, iy, qx, v);
  up2_fir(v, kf, o);
}

__global__ void __launch_bounds__(256) up2_fir_act_kernel(const float* __restrict__ y_lo, const float* __restrict__ fir,
                                                          float* __restrict__ out, int B, int C, int H, int W,
                                                          const float* __restrict__ noise, int64_t noise_bstride,
                                                          const float* __restrict__ nwp, const float* __restrict__ bias,
                                                          float* __restrict__ out_amax) {
  const float noise_w = (noise && nwp) ? nwp[0] : 0.f;
  float mx = 0.f;      // out_amax: the largest |out| of the sample this thread is in (a grid-stride loop may cross samples)
  int mx_b = -1;
  float kf[16];   // flipped taps: kf[ky][kx] = fir[3-ky][3-kx]
#pragma unroll
  for (int i = 0; i < 16; ++i) kf[i] = fir[15 - i];
  const int OW = 2 * W;
  const int qw = W / 2;                                    // 4-pixel output quads per row
  const int64_t total = (int64_t)B * C * H * qw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int qx = (int)(t % qw); t /= qw;
    const int iy = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const float* src = y_lo + ((int64_t)b * C + c) * H * W;
    const float bs = bias[c];
    if (out_amax && b != mx_b) {        // (rare: at most B - 1 times per thread)
      if (mx_b >= 0) cips3d_amax_raise(out_amax + mx_b * CIPS3D_AMAX_FLOATS, mx, blockIdx.x);
      mx = 0.f;
      mx_b = b;
    }
    float o[2][4];
    up2_block(src, H, W, iy, qx, kf, o);
    float* dst = out + (((int64_t)b * C + c) * 2 * H + 2 * iy) * OW + 4 * qx;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      float4 nz = make_float4(0.f, 0.f, 0.f, 0.f);
      if (noise) nz = *reinterpret_cast<const float4*>(noise + (int64_t)b * noise_bstride + (int64_t)(2 * iy + py) * OW + 4 * qx);
      float4 r;
      r.x = lrelu02(fmaf(noise_w, nz.x, o[py][0]) + bs) * 1.41421356237309515f;
      r.y = lrelu02(fmaf(noise_w, nz.y, o[py][1]) + bs) * 1.41421356237309515f;
      r.z = lrelu02(fmaf(noise_w, nz.z, o[py][2]) + bs) * 1.41421356237309515f;
      r.w = lrelu02(fmaf(noise_w, nz.w, o[py][3]) + bs) * 1.41421356237309515f;
      cips3d_store_wt16(dst + (int64_t)py * OW, r);
      mx = fmaxf(fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))), mx);
    }
  }
  if (out_amax) {
    // One atomic per WORKGROUP, and only when it would raise the slot (the slot's current value is peeked first: 16 384
    // workgroups raising 8 lines unconditionally cost more than the absmax pass this replaces).  Lanes that ended in another
    // sample than the workgroup's first (a straddling workgroup: at most B - 1 of them) raise theirs directly.
    __shared__ float s_mx[16];
    __shared__ int s_b;
    if (threadIdx.x == 0) s_b = mx_b;
    __syncthreads();
    const int b0 = s_b;
    if (mx_b >= 0 && mx_b != b0) cips3d_amax_raise(out_amax + mx_b * CIPS3D_AMAX_FLOATS, mx, blockIdx.x);
    const float m = cips3d_workgroup_max(mx_b == b0 ? mx : 0.f, s_mx, threadIdx.x >> 6, threadIdx.x & 63, 4);
    if (threadIdx.x == 0 && b0 >= 0) {
      float* slots = out_amax + b0 * CIPS3D_AMAX_FLOATS;
      cips3d_amax_raise_if(slots, m, blockIdx.x, cips3d_amax_peek(slots, blockIdx.x));
    }
  }
}

// StyledConv epilogue on its own (generality path): out = lrelu(x + noise_w*noise + bias[c], 0.2)*sqrt(2)
// (NoiseInjection as ONE fused multiply-add in every kernel of the library: `image + weight * noise` of model_v3.py:317-341 with
// one rounding instead of two -- the same value in the fused stages, the chain, the per-layer GEMM and the differentiable forward)
__global__ void __launch_bounds__(256) noise_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                             int64_t noise_bstride, const float* __restrict__ nwp,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int B, int C, int64_t HW) {
  const float noise_w = (noise && nwp) ? nwp[0] : 0.f;
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i % HW;
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / (HW * C));
    const float nz = noise ? noise[(int64_t)b * noise_bstride + n] : 0.f;
    out[i] = lrelu02(fmaf(noise_w, nz, x[i]) + bias[c]) * 1.41421356237309515f;
  }
}

// ------------------------------------------------------------------------------------------------
// ToRGB: 3 x Cin GEMV per pixel on the VALU (HBM-bound: Cin*4 B read per pixel), + bias + skip.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) torgb_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                    const float* __restrict__ bias, const float* __restrict__ skip,
                                                    int skip_up, const float* __restrict__ fir, float* __restrict__ out,
                                                    int B, int Cin, int H, int W) {
  extern __shared__ float s_w[];   // [3][Cin] of this sample
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * Cin; i += 256) s_w[i] = wm[(int64_t)b * 3 * Cin + i];
  __syncthreads();
  const int64_t HW = (int64_t)H * W;
  const int64_t quads = HW / 4;
  const float* xb = x + (int64_t)b * Cin * HW;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (int64_t)gridDim.x * blockDim.x) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), g = r, bl = r;
#pragma unroll 4
    for (int i = 0; i < Cin; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)i * HW + q * 4);
      const float w0 = s_w[i], w1 = s_w[Cin + i], w2 = s_w[2 * Cin + i];
      r.x = fmaf(w0, v.x, r.x); r.y = fmaf(w0, v.y, r.y); r.z = fmaf(w0, v.z, r.z); r.w = fmaf(w0, v.w, r.w);
      g.x = fmaf(w1, v.x, g.x); g.y = fmaf(w1, v.y, g.y); g.z = fmaf(w1, v.z, g.z); g.w = fmaf(w1, v.w, g.w);
      bl.x = fmaf(w2, v.x, bl.x); bl.y = fmaf(w2, v.y, bl.y); bl.z = fmaf(w2, v.z, bl.z); bl.w = fmaf(w2, v.w, bl.w);
    }
    float4 res[3] = {r, g, bl};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float4 o = res[c];
      const float bs = bias[c];
      o.x += bs; o.y += bs; o.z += bs; o.w += bs;
      if (skip) {
        if (skip_up) {
          const int oy = (int)((q * 4) / W), ox0 = (int)((q * 4) % W);
          const float* sp = skip + ((int64_t)b * 3 + c) * (H / 2) * (W / 2);
          o.x += up2_tap(sp, H / 2, W / 2, oy, ox0 + 0, fir);
          o.y += up2_tap(sp, H / 2, W / 2, oy, ox0 + 1, fir);
          o.z += up2_tap(sp, H / 2, W / 2, oy, ox0 + 2, fir);
          o.w += up2_tap(sp, H / 2, W / 2, oy, ox0 + 3, fir);
        } else {
          const float4 sv = *reinterpret_cast<const float4*>(skip + ((int64_t)b * 3 + c) * HW + q * 4);
          o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w;
        }
      }
      *reinterpret_cast<float4*>(out + ((int64_t)b * 3 + c) * HW + q * 4) = o;
    }
  }
}

// Channel-split variant: a workgroup owns 256/S quads, its S thread slices each reduce Cin/S channels and
// the partial sums meet in LDS.  Keeps every CU busy at low resolution (64^2 x 512 channels is only 1024
// quads) and multiplies the loads in flight per CU at high resolution.
template <int S>
__global__ void __launch_bounds__(256) torgb_split_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                          const float* __restrict__ bias, const float* __restrict__ skip,
                                                          int skip_up, const float* __restrict__ fir,
                                                          float* __restrict__ out, int B, int Cin, int H, int W) {
  constexpr int QB = 256 / S;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];   // [256][12] partials, then [3][Cin] weights
  float* s_part = s_mem;
  float* s_w = s_mem + 256 * 12;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * Cin; i += 256) s_w[i] = wm[(int64_t)b * 3 * Cin + i];
  __syncthreads();
  const int64_t HW = (int64_t)H * W;
  const int64_t quads = HW / 4;
  const int ql = tid % QB, sl = tid / QB;
  const int64_t q = (int64_t)blockIdx.x * QB + ql;
  const bool qok = q < quads;
  const float* xb = x + (int64_t)b * Cin * HW + (qok ? q : 0) * 4;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f), g = r, bl = r;
#pragma unroll 4
  for (int i = sl; i < Cin; i += S) {
    const float4 v = *reinterpret_cast<const float4*>(xb + (int64_t)i * HW);
    const float w0 = s_w[i], w1 = s_w[Cin + i], w2 = s_w[2 * Cin + i];
    r.x = fmaf(w0, v.x, r.x); r.y = fmaf(w0, v.y, r.y); r.z = fmaf(w0, v.z, r.z); r.w = fmaf(w0, v.w, r.w);
    g.x = fmaf(w1, v.x, g.x); g.y = fmaf(w1, v.y, g.y); g.z = fmaf(w1, v.z, g.z); g.w = fmaf(w1, v.w, g.w);
    bl.x = fmaf(w2, v.x, bl.x); bl.y = fmaf(w2, v.y, bl.y); bl.z = fmaf(w2, v.z, bl.z); bl.w = fmaf(w2, v.w, bl.w);
  }
  float4* pp = reinterpret_cast<float4*>(s_part) + (sl * QB + ql) * 3;
  pp[0] = r; pp[1] = g; pp[2] = bl;
  __syncthreads();
  if (tid < QB * 3) {
    const int c = tid / QB, q2 = tid % QB;
    const int64_t qq = (int64_t)blockIdx.x * QB + q2;
    if (qq < quads) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const float4 v = reinterpret_cast<const float4*>(s_part)[(k * QB + q2) * 3 + c];
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      }
      const float bs = bias[c];
      o.x += bs; o.y += bs; o.z += bs; o.w += bs;
      if (skip) {
        if (skip_up) {
          const int oy = (int)((qq * 4) / W), ox0 = (int)((qq * 4) % W);
          const float* sp = skip + ((int64_t)b * 3 + c) * (H / 2) * (W / 2);
          o.x += up2_tap(sp, H / 2, W / 2, oy, ox0 + 0, fir);
          o.y += up2_tap(sp, H / 2, W / 2, oy, ox0 + 1, fir);
          o.z += up2_tap(sp, H / 2, W / 2, oy, ox0 + 2, fir);
          o.w += up2_tap(sp, H / 2, W / 2, oy, ox0 + 3, fir);
        } else {
          const float4 sv = *reinterpret_cast<const float4*>(skip + ((int64_t)b * 3 + c) * HW + qq * 4);
          o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w;
        }
      }
      *reinterpret_cast<float4*>(out + ((int64_t)b * 3 + c) * HW + qq * 4) = o;
    }
  }
}

// any H, W (no 16-byte alignment): one pixel per thread
__global__ void __launch_bounds__(256) torgb_scalar_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                           const float* __restrict__ bias, const float* __restrict__ skip,
                                                           int skip_up, const float* __restrict__ fir,
                                                           float* __restrict__ out, int B, int Cin, int H, int W) {
  const int64_t HW = (int64_t)H * W;
  const int64_t total = (int64_t)B * HW;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const int64_t n = idx % HW;
    const float* xb = x + (int64_t)b * Cin * HW + n;
    const float* w = wm + (int64_t)b * 3 * Cin;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < Cin; ++i) {
      const float v = xb[(int64_t)i * HW];
      acc[0] = fmaf(w[i], v, acc[0]); acc[1] = fmaf(w[Cin + i], v, acc[1]); acc[2] = fmaf(w[2 * Cin + i], v, acc[2]);
    }
    const int oy = (int)(n / W), ox = (int)(n % W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float o = acc[c] + bias[c];
      if (skip) {
        if (skip_up) o += up2_tap(skip + ((int64_t)b * 3 + c) * (H / 2) * (W / 2), H / 2, W / 2, oy, ox, fir);
        else o += skip[((int64_t)b * 3 + c) * HW + n];
      }
      out[((int64_t)b * 3 + c) * HW + n] = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// direct k x k (generality path: k = 3 configs, channel counts the MFMA kernel does not tile)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) modconv_kxk_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                          float* __restrict__ out, int B, int Cin, int Cout, int H,
                                                          int W, int k, int transpose2, int OH, int OW) {
  const int64_t total = (int64_t)B * Cout * OH * OW;
  const int pad = k / 2;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = idx;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH); t /= OH;
    const int o = (int)(t % Cout);
    const int b = (int)(t / Cout);
    const float* wrow = wm + ((int64_t)b * Cout + o) * Cin * k * k;
    const float* xb = x + (int64_t)b * Cin * H * W;
    float acc = 0.f;
    for (int i = 0; i < Cin; ++i) {
      const float* xi = xb + (int64_t)i * H * W;
      const float* wi = wrow + i * k * k;
      for (int ky = 0; ky < k; ++ky) {
        int iy;
        if (transpose2) { const int ty = oy - ky; if (ty < 0 || (ty & 1)) continue; iy = ty >> 1; }
        else iy = oy + ky - pad;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
          int ix;
          if (transpose2) { const int tx = ox - kx; if (tx < 0 || (tx & 1)) continue; ix = tx >> 1; }
          else ix = ox + kx - pad;
          if (ix < 0 || ix >= W) continue;
          acc = fmaf(xi[(int64_t)iy * W + ix], wi[ky * k + kx], acc);
        }
      }
    }
    out[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Fused up-sampling stage:  y_lo --(2x FIR + noise1 + bias1 + lrelu)--> act1 --(1x1 modconv, noise2 + bias2
// + lrelu)--> out2 --(ToRGB + bias + FIR-upsampled skip)--> rgb, one kernel.  act1 (the largest tensor of
// the stage) only ever exists as LDS B-operand stages; out2 is stored only when a later stage needs it
// (never for the last stage, which then writes 12 B per pixel instead of moving ~540 B).
//   workgroup = 8 waves: WGM wave rows cover ALL C output channels (so ToRGB reduces inside the
//   workgroup, deterministically), WGN waves cover WGN image rows of 64 pixels.
//   per 32-channel K stage: every thread builds 2x4-pixel blocks of act1 from a 3x4 patch of y_lo (VALU),
//   writes them to the LDS stage the MFMAs of the next step read; A fragments come straight from L2.
// ------------------------------------------------------------------------------------------------
struct FusedArgs {
  const float* y_lo; const float* fir; const float* noise1; int64_t nbs1; const float* nw1; const float* bias1;
  const float* wm2; const float* noise2; int64_t nbs2; const float* nw2; const float* bias2; float* out2;
  const float* wm_rgb; const float* bias_rgb; const float* skip; int skip_up; float* rgb;
  int B, H, W;   // low-resolution size; the stage outputs 2H x 2W (FLAT: the size of y_lo AND of the outputs)
  int bf16;      // 0: exact fp32; 1: bf16 GEMM operands (CIPS3D_GEMM_BF16); 2: additionally y_lo / y_next are bf16 arrays (CIPS3D_Y_BF16);
                 // 3: fp32-equivalent split-fp16 products (CIPS3D_GEMM_SPLIT; wm2 / wm_next CIPS3D_MOD_SPLIT16-packed)
  // optional (NEXT instantiation): the next stage's low-resolution GEMM y_next = wm_next (C/2 x C, chained pack) out2
  const float* wm_next; float* y_next;
  // range tracking (split mode; cips3d_range): amax of y_lo, the constants of conv1 / conv2; max |y_next| recorded (out2 is
  // not tracked: 64 registers is all the C = 32 stage has -- a split GEMM that reads a stored out2 measures it, cips3d_absmax)
  const float* x_amax; const float* lconst1; const float* lconst2; float* next_amax;
  // next_gain > 0: next_amax does not receive the measured max|y_next| (a workgroup reduction + one atomic per workgroup:
  // 4 us at the 4096 workgroups of the C = 64 stage) but its BOUND next_gain * U2 (next_gain: sqrt(C) for a demodulated next
  // up-conv), written once -- good enough for a consumer that is the last stage of the decoder (cips3d_range)
  float next_gain;
  // FLAT stage (CIPS3D_STAGE_FLAT): a decoder block that does not up-sample -- [StyledConv, StyledConv, ToRGB] at ONE resolution.
  // y_lo is conv1's GEMM result at that resolution, `fir` is unused, the skip image (if any) has the output's size and is added
  // as it is.  Everything behind the first activation is the up-sampling stage's code.
  int flat;
};

// MINW = waves per SIMD the register allocation must leave room for (the per-workgroup chain load -> FIR -> LDS -> MFMA ->
// store is serial, so throughput comes from co-resident workgroups).
#ifndef CIPS3D_C128_LATE
#define CIPS3D_C128_LATE true
#endif
#ifndef CIPS3D_C64_MINW
#define CIPS3D_C64_MINW 4
#endif
#ifndef CIPS3D_C32_MINW
#define CIPS3D_C32_MINW 8      // A/B knob: waves per SIMD the C = 32 stage is compiled for
#endif
#ifndef CIPS3D_FUSED_AB
// timing-only ablations of the fused stages (results garbage; tools/README.md): 1 no scale loads, 2 no y_next record, 4 no ToRGB
// sums (at C = 32, where nothing else reads conv2's output, the compiler then drops conv2 altogether: read that one as "no
// conv2"), 8 no FIR arithmetic, 16 no conv2 epilogue arithmetic, 32 no activation / split of conv1, 64 no patch loads (register
// values instead), 128 no noise / skip loads of the epilogues
#define CIPS3D_FUSED_AB 0
#endif
#ifdef CIPS3D_FUSED_STAMPS
// Diagnostic build only: per-phase cycle sums of wave 0 of every workgroup of the fused up-sampling stages, slot = log2(C) - 5
// (C = 32, 64, 128, 256), accumulated in registers, flushed at the end of the workgroup.
__device__ unsigned long long g_fused_stamps[4][8];
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fst_[i] += t_ - ftp_; ftp_ = t_; } while (0)
#define FSTAMP_FLUSH() do { if (tid == 0 && (blockIdx.x & 31) == 0) {   /* every 32nd workgroup: the atomics must not become the load */ constexpr int sl_ = C == 32 ? 0 : C == 64 ? 1 : C == 128 ? 2 : 3; \
    for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_fused_stamps[sl_][i_], fst_[i_]); atomicAdd(&g_fused_stamps[sl_][7], 1ull); } } while (0)
#else
#define FSTAMP(i)
#define FSTAMP_FLUSH()
#endif

template <int C, int WM, int WGM, int WGN, int RW, int BK, int MINW, int PREC, bool NEXT = false, bool XPREF = false,
          bool LATE_OPS = false, bool FLAT = false>
__global__ void __launch_bounds__(64 * WGM * WGN, MINW) fused_up_conv_kernel(FusedArgs a) {
  constexpr bool BF16 = PREC == 1 || PREC == 2;  // bf16 MFMA operands, fp32 accumulate
  constexpr bool YB = PREC == 2;                 // y_lo (in) and y_next (out) stored as bf16
  constexpr bool SPLIT = PREC == 3;              // fp32-equivalent split-fp16 products (weights CIPS3D_MOD_SPLIT16-packed)
  typedef typename std::conditional<YB, bf16_t, float>::type ylo_t;
  // A wave covers RW image rows x CW columns (64 pixels, 4 consecutive x per lane); the WGN waves of a
  // workgroup are stacked vertically: pixel tile TH x TW.
  constexpr int NT = 64 * WGM * WGN;
  constexpr int CW = 64 / RW, TH = RW * WGN, TW = CW, BN = TH * TW;
  constexpr int NSTAGE = C / BK, NBUF = NSTAGE > 1 ? 2 : 1, KQ = BK / 16;
  constexpr int LPR = CW / 4;                    // lanes (pixel quads) per image row of the wave tile
  static_assert((WGM * WGN == 8 || WGM * WGN == 4) && 16 * WM * WGM == C && TH % 2 == 0 && C % BK == 0, "tile shape");
  constexpr int NBLK = (TH / 2) * (TW / 4);      // 2x4 blocks per channel in the tile
  constexpr int BPT = BK * NBLK / NT;            // blocks per thread per stage
  static_assert((BK * NBLK) % NT == 0 && BPT >= 1, "block split");
  __shared__ __attribute__((aligned(16))) float sB[NBUF * BK * BN];
  __shared__ __attribute__((aligned(16))) float s_nz1[BN + 8];     // + the stage's scale factors (wave 0 -> everyone)
  // (the last stage has no chained GEMM: its B tile is dead when the ToRGB partials are exchanged, so they take its place --
  // 3 KB of LDS less per workgroup, which with <= 64 VGPRs admits an eighth workgroup per CU)
  __shared__ __attribute__((aligned(16))) float s_red_own[NEXT ? WGM * 3 * BN : 4];
  float* s_red = NEXT ? s_red_own : sB;
  __shared__ __attribute__((aligned(16))) float s_wrgb[3 * C];

  const int tid = threadIdx.x;
#ifdef CIPS3D_FUSED_STAMPS
  unsigned long long fst_[7] = {0, 0, 0, 0, 0, 0, 0}, ftp_ = __builtin_amdgcn_s_memtime();
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int wm_i = wave / WGN, wn_i = wave % WGN;
  const int q = lane >> 4, jn = lane & 15;
  const int lrow = jn / LPR, lx4 = jn % LPR;
  const int nloc = (wn_i * RW + lrow) * TW + lx4 * 4;     // this lane's first pixel inside the tile
  const int b = blockIdx.z;
  const int H = a.H, W = a.W, OH = FLAT ? H : 2 * H, OW = FLAT ? W : 2 * W;
  const int tiles_x = OW / TW;
  // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2: in row-major
  // tile order the horizontal neighbours of a tile -- whose FIR halo columns sit in the SAME 128-byte lines as its own
  // first / last columns -- always ran on other XCDs, so every (channel, row) segment pulled three lines through its L2 for
  // one line of unique data (FETCH_SIZE 2.5x the algorithmic bytes at C = 32; tools/calib/fetch_calibrate.hip shows the
  // counter itself is exact for these load widths).  Here XCD k walks the k-th contiguous eighth of the tiles instead:
  // neighbours in both directions share an L2 and are dispatched 8 (resp. 8 * tiles_x) workgroups apart.
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int ox0 = (bid % tiles_x) * TW, oy0 = (bid / tiles_x) * TH;
  // Per-lane index arithmetic stays in 32 bits (a 64 x 64-bit multiply is three quarter-rate VALU instructions): 64-bit
  // products only in the workgroup-uniform per-sample bases (the host refuses C * 4HW >= 2^31)
  const int HWlo = H * W, HWo = OH * OW;
  const int oy = oy0 + wn_i * RW + lrow, ox = ox0 + lx4 * 4;

  // scale factors of the split operands (cips3d_range); set right before the first patch_store below
  float kact1 = 1.41421356237309515f, k2in = kSplitInv, kact2 = 1.41421356237309515f, kback2 = 1.f, kyn = kSplitInv;

#ifndef CIPS3D_FUSED_TILE_DMA
#define CIPS3D_FUSED_TILE_DMA 1      // 0: the noise tile and the ToRGB rows go through registers behind the range block, as before round 5 (A/B)
#endif
  constexpr bool TILE_DMA = CIPS3D_FUSED_TILE_DMA != 0;
  float nw1_u = 0.f;                        // NoiseInjection.weight of conv1 (TILE_DMA: applied where the tile is read)
  // the layer's range constants (uniform addresses: scalar loads -- as long as they are read in FRONT of the LDS-DMA below)
  float c10_u = 0.f, c11_u = 0.f, c20_u = 0.f, c21a_u = 0.f, c21b_u = 0.f;
  if constexpr (SPLIT && TILE_DMA) {
    if (a.x_amax && !(CIPS3D_FUSED_AB & 1)) {
      const float* l1 = a.lconst1 + b * 4;
      c10_u = l1[0]; c11_u = l1[1];
      if (NEXT) {
        const float* l2 = a.lconst2 + b * 4;
        c20_u = l2[0]; c21a_u = l2[1]; c21b_u = l2[2];
      }
    }
  }
  // Everything that does not depend on a barrier is requested first: conv2's first A fragments, its noise / bias.
  const float* ab = a.wm2 + (int64_t)b * C * C;   // packed [ot][kq][256]
  f32x4 afr_next[KQ][WM];
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
    for (int i = 0; i < WM; ++i)
      afr_next[kq][i] = *reinterpret_cast<const f32x4*>(ab + (((wm_i * WM + i) * (C / 16) + kq) * 256 + lane * 4));
  // LATE (single-stage kernels, C = BK): the epilogue's operands are requested after the MFMAs instead of up front -- 24
  // registers that buy a sixth resident wave per SIMD
  constexpr bool LATE = LATE_OPS;
  f32x4 nz2 = {0.f, 0.f, 0.f, 0.f};
  float nw2_u = 0.f;                        // NoiseInjection.weight of conv2 (applied by the FMA that adds the noise)
  f32x4 bias4[WM];
  auto load_epilogue_ops = [&]() {
    if (a.noise2 && a.nw2 && !(CIPS3D_FUSED_AB & 128)) {
      nz2 = *reinterpret_cast<const f32x4*>(a.noise2 + (int64_t)b * a.nbs2 + (oy * OW + ox));
      nw2_u = a.nw2[0];
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) bias4[i] = *reinterpret_cast<const f32x4*>(a.bias2 + (wm_i * WM + i) * 16 + 4 * q);
  };
  if (!LATE) load_epilogue_ops();

  float kf[16];   // flipped taps
#pragma unroll
  for (int i = 0; i < 16; ++i) kf[i] = FLAT ? 0.f : a.fir[15 - i];
  // the lanes that will finish a colour channel (wave row 0, quarter = channel) fetch their skip operands now
  const bool rgb_lane = a.wm_rgb && wm_i == 0 && q < 3;
  float skp[3][4];
  f32x4 skv = {0.f, 0.f, 0.f, 0.f};
  float rgb_bias = 0.f;
  auto load_skip_ops = [&]() {
    if (rgb_lane) {
      rgb_bias = a.bias_rgb[q];
      if (a.skip && !(CIPS3D_FUSED_AB & 128)) {
        if (!FLAT && (a.skip_up & 1)) up2_load(a.skip + (int64_t)b * 3 * HWlo + q * HWlo, H, W, oy >> 1, ox >> 2, skp);
        else skv = *reinterpret_cast<const f32x4*>(a.skip + (int64_t)b * 3 * HWo + (q * HWo + oy * OW + ox));
      }
    }
  };
  if (!LATE) load_skip_ops();

  // FIR patches of one K stage: loaded early (before the MFMAs that precede their use), filtered late.  Tiles whose
  // low-resolution window (rows oy0/2-1 .. oy0/2+TH/2, columns ox0/2-1 .. ox0/2+TW/2) lies inside the image take loads
  // without edge predication (workgroup-uniform branch; 87 % of the tiles at 1024^2).
  const bool interior = FLAT || (oy0 / 2 >= 1 && oy0 / 2 + TH / 2 < H && ox0 / 2 >= 1 && ox0 / 2 + TW / 2 < W);
  // Two patch register sets when the K loop has >= 3 stages and it pays (C = 256): the patches of stage
  // st + 2 are requested at the START of stage st and filtered at the END of stage st + 1 -- two stages of lead.  With one
  // stage of lead (the form C = 64 keeps) a stage lasted as long as a patch round trip: in-kernel stamps (tools/
  // fused_stamps.py) showed 5.6k cycles per stage at C = 256 for 1.5k cycles of matrix work.
#ifndef CIPS3D_FUSED_DEEP
#define CIPS3D_FUSED_DEEP 1      // A/B knob
#endif
#ifndef CIPS3D_DEEP_MIN_C
#define CIPS3D_DEEP_MIN_C 256
#endif
  constexpr bool DEEP = CIPS3D_FUSED_DEEP && NSTAGE >= 3 && C >= CIPS3D_DEEP_MIN_C && (C != 128);   // same-box A/B: C = 256 24.4 -> 22.9 us; C = 128 (252 VGPRs with it) 25.2 -> 25.5: off there
  float pv[BPT][3][4], pw[BPT][3][4];
  auto patch_load_into = [&](int st, float (&pset)[BPT][3][4]) {
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      const int g = tid + NT * u;
      const int ch = g / NBLK, rem = g % NBLK;
      const int by = rem / (TW / 4), qx = rem % (TW / 4);
      const ylo_t* src = reinterpret_cast<const ylo_t*>(a.y_lo) + (int64_t)b * C * HWlo + (st * BK + ch) * HWlo;
      if constexpr (CIPS3D_FUSED_AB & 64) {        // no patch loads: what the latency chain behind them costs
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) pset[u][r][c] = __int_as_float(0x3c000000 + ((tid + st + r * 4 + c) << 8));
      } else
      if constexpr (FLAT) flat_load(src, W, oy0 + 2 * by, ox0 + 4 * qx, pset[u]);
      else if (interior) up2_load_interior(src, W, oy0 / 2 + by, ox0 / 4 + qx, pset[u]);
      else up2_load(src, H, W, oy0 / 2 + by, ox0 / 4 + qx, pset[u]);
    }
  };
  auto patch_load = [&](int st) { patch_load_into(st, pv); };
  auto patch_store_from = [&](int st, float* dst, float (&pset)[BPT][3][4]) {
#pragma unroll
    for (int u = 0; u < BPT; ++u) {
      const int g = tid + NT * u;
      const int ch = g / NBLK, rem = g % NBLK;
      const int by = rem / (TW / 4), qx = rem % (TW / 4);
      float o[2][4];
      if constexpr (FLAT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[0][c] = pset[u][0][c]; o[1][c] = pset[u][1][c]; }
      } else if constexpr (CIPS3D_FUSED_AB & 8) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { o[0][c] = pset[u][0][c] + pset[u][1][c]; o[1][c] = pset[u][1][c] + pset[u][2][c]; }
      } else {
        up2_fir(pset[u], kf, o);
      }
      const float bs = a.bias1[st * BK + ch];
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const f32x4 nz = *reinterpret_cast<const f32x4*>(s_nz1 + (2 * by + py) * TW + qx * 4);
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // (the tile holds the raw noise, its weight rides in the FMA: see noise_bias_act_kernel)
          const float on = fmaf(nw1_u, nz[c], o[py][c]);
          v[c] = (CIPS3D_FUSED_AB & 32) ? on : lrelu02(on + bs);
        }
        if constexpr (SPLIT && !(CIPS3D_FUSED_AB & 32)) {          // split once here (with the activation's gain and range scale: cips3d_split_word);
#pragma unroll                          // every wave row reads the packed halves
          for (int c = 0; c < 4; ++c) v[c] = pack_split(v[c], kact1);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] *= kact1;
        }
        *reinterpret_cast<f32x4*>(dst + ch * BN + (2 * by + py) * TW + qx * 4) = v;
      }
    }
  };
  auto patch_store = [&](int st, float* dst) { patch_store_from(st, dst, pv); };
  // Round 5: the first conv's noise tile and the ToRGB rows have to be in LDS at the first barrier.  Loaded where they were stored
  // -- behind the range block, `load -> wait -> LDS store` each -- they were two more dependent round trips of every workgroup's
  // prologue (three in all, with the patches + amax slots).  Now they travel by LDS-DMA, requested with the first patches: no
  // registers (the C = 32 / 64 stages have none to spare: five more live values spilled 34 / 20), no wait of their own -- the one
  // vmcnt(0) in front of the first barrier covers them, the patches and the range slots.  The tile holds the RAW noise; its weight
  // rides in the FMA that adds it (patch_store).  The request sits BEHIND the uniform loads of the prologue (FIR taps, biases): an
  // LDS-DMA counts as a memory write for the compiler, and uniform loads behind one are no longer scalar loads -- issued at the very
  // top, the 16 taps alone came back in vector registers and the C = 32 / 64 stages spilled 43 / 34.
  // The first patches go out in FRONT of the LDS-DMA requests: behind them, the compiler waited for vmcnt(0) -- the A fragments, the
  // noise tile, the ToRGB rows, all from a cold L2 -- before it reused the DMAs' address registers for the patch addresses.  Same
  // box (rocprofv3 / unprofiled bench, variants interleaved): C = 32 stage 38.5 -> 37.3 us, the view 0.3481 -> 0.3474 ms.
  // (Measured with it and NOT kept: the patch rows as register quads from the load to the FIR, all of a stage's patches under ONE
  // interior / edge branch -- the interior path then issues its six 16-byte loads back to back with no wait between them, where
  // today every row's load is waited for before its middle pair is copied to an even register pair -- 39.1 / 32.9 / 25.1 / 23.1 us
  // for C = 32 / 64 / 128 / 256 against 37.3 / 32.3 / 23.3 / 21.5: the co-resident workgroups cover those waits, the quads cost
  // registers and copies in the K loop.)
#ifndef CIPS3D_FUSED_PATCH_FIRST
#define CIPS3D_FUSED_PATCH_FIRST 1
#endif
  constexpr bool PATCH_FIRST = CIPS3D_FUSED_PATCH_FIRST && TILE_DMA;
  if constexpr (PATCH_FIRST) {
    patch_load(0);
    if (DEEP) patch_load_into(1, pw);          // stage s's patches live in set s & 1 (pv: even, pw: odd)
  }
  if constexpr (TILE_DMA) {
    const bool have_nz = a.noise1 && a.nw1 && !(CIPS3D_FUSED_AB & 128);
    if (have_nz) nw1_u = a.nw1[0];
    if (wave == 0) {                        // BN / 4 <= 64 lanes (BN = 128 in every instantiation): one wave's worth of 16-byte pieces
      static_assert(BN / 4 <= 64 && (3 * C) % 4 == 0, "one LDS-DMA instruction per tile");
      if (lane < BN / 4) {
        if (have_nz) {
          const int r = lane / (TW / 4), x4 = lane % (TW / 4);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.noise1 + (int64_t)b * a.nbs1 + ((oy0 + r) * OW + ox0 + x4 * 4)),
                                           (__attribute__((address_space(3))) void*)s_nz1, 16, 0, 0);
        } else {
          *reinterpret_cast<f32x4*>(s_nz1 + lane * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (a.wm_rgb) {                         // 3 C floats = 3 C / 4 pieces of 16 bytes, 64 per wave-instruction
#pragma unroll
      for (int p0 = 0; p0 < 3 * C / 4; p0 += 64 * WGM * WGN) {
        const int pbase = p0 + wave * 64;   // (wave-uniform LDS base: the hardware adds lane * 16)
        if (pbase < 3 * C / 4 && pbase + lane < 3 * C / 4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.wm_rgb + (int64_t)b * 3 * C + (pbase + lane) * 4),
                                           (__attribute__((address_space(3))) void*)(s_wrgb + pbase * 4), 16, 0, 0);
      }
    }
  }
  if constexpr (!PATCH_FIRST) {
    patch_load(0);
    if (DEEP) patch_load_into(1, pw);          // stage s's patches live in set s & 1 (pv: even, pw: odd)
  }

  // Range of the split operands (cips3d_range, common.h).  act1 is split as act1 2^-e1, e1 from the bound
  // U1 = c1 max|y_lo| + c0 (conv1's constants: FIR gain, noise, bias); conv2's accumulators come back with 2^-8 2^e1.  In the
  // chained form out2 is split as out2 2^-e2, U2 = c1' U1 + c0' (conv2's constants), and y_next comes back with 2^-8 2^e2.
  // Both powers of two ride on the sqrt(2) of the activations: no instruction per value.  Wave 0 makes them -- ONE vector load
  // for the amax slots (lane = slot), issued here with the tile's first loads and waited for with them at the barrier below;
  // as scalar loads in front of the first patch_store their latency was exposed: +2.9 us at C = 32 -- and leaves them behind
  // the noise tile for everyone.

  if constexpr (SPLIT) {
    if (a.x_amax && wave == 0 && !(CIPS3D_FUSED_AB & 1)) {
      const float t = lane < CIPS3D_AMAX_SLOTS ? a.x_amax[b * CIPS3D_AMAX_FLOATS + lane * CIPS3D_AMAX_STRIDE] : 0.f;
      float c10, c11, c20 = 0.f, c21 = 0.f;
      if constexpr (TILE_DMA) {
        c10 = c10_u; c11 = c11_u;
        if (NEXT) { c20 = c20_u; c21 = fmaxf(c21a_u, 1.41421356237309515f * c21b_u); }
      } else {
        const float* l1 = a.lconst1 + b * 4;            // (uniform addresses: scalar loads)
        c10 = l1[0]; c11 = l1[1];
        if (NEXT) {
          const float* l2 = a.lconst2 + b * 4;
          c20 = l2[0];
          c21 = fmaxf(l2[1], 1.41421356237309515f * l2[2]);
        }
      }
      // (FLAT: x_amax is the maximum of conv1's GEMM RESULT -- what the FIR's gain multiplies in the up-sampling form; the row
      // gain of a plain StyledConv's constants is relative to that conv's INPUT)
      if (FLAT) c11 = 1.41421356237309515f;
      const float m_in = cips3d_wave_max_uniform(t);
      const float u1 = fmaf(c11, m_in, c10);
      const int e1 = cips3d_split_exp(u1);
      f32x4 sc = {1.41421356237309515f * cips3d_pow2(-e1), kSplitInv * cips3d_pow2(e1), 1.41421356237309515f, 1.f};
      if (NEXT) {
        const float u2 = fmaf(c21, u1, c20);
        const int e2 = cips3d_split_exp(u2);
        sc[2] = 1.41421356237309515f * cips3d_pow2(-e2);
        sc[3] = cips3d_pow2(e2);
        // the bound of |y_next| instead of its measured maximum (see FusedArgs::next_gain): slot 0, once per sample
        if (a.next_amax && a.next_gain > 0.f && blockIdx.x == 0 && lane == 0) a.next_amax[b * CIPS3D_AMAX_FLOATS] = a.next_gain * u2 * 1.000001f;
      }
      if (lane == 0) *reinterpret_cast<f32x4*>(s_nz1 + BN) = sc;
    }
  }
  // noise of the first conv for this tile, ToRGB weights (TILE_DMA: on their way since the top of the kernel)
  if constexpr (TILE_DMA) {
    __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): this wave's LDS-DMA pieces (and, needed right behind the barrier anyway, its patches)
  } else {
  if (a.noise1 && a.nw1 && !(CIPS3D_FUSED_AB & 128)) nw1_u = a.nw1[0];
  if (tid < BN / 4) {
    const int r = tid / (TW / 4), x4 = tid % (TW / 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a.noise1 && a.nw1 && !(CIPS3D_FUSED_AB & 128)) {
      v = *reinterpret_cast<const f32x4*>(a.noise1 + (int64_t)b * a.nbs1 + ((oy0 + r) * OW + ox0 + x4 * 4));
    }
    *reinterpret_cast<f32x4*>(s_nz1 + r * TW + x4 * 4) = v;
  }
  if (a.wm_rgb)
    for (int i = tid; i < 3 * C; i += NT) s_wrgb[i] = a.wm_rgb[(int64_t)b * 3 * C + i];
  }
  __syncthreads();

  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  FSTAMP(0);        // operand requests, noise staged, first barrier
  if constexpr (SPLIT) {      // the scale factors wave 0 left behind the noise tile (below), now wave-uniform registers
    if (a.x_amax && !(CIPS3D_FUSED_AB & 1)) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(s_nz1 + BN);
      kact1 = cips3d_uniform(sc[0]);
      k2in = cips3d_uniform(sc[1]);
      if (NEXT) {
        kact2 = cips3d_uniform(sc[2]);
        kback2 = cips3d_uniform(sc[3]);
        kyn = cips3d_uniform(kSplitInv * kback2);
      }
    }
  }

  patch_store(0, sB);
  __syncthreads();
  FSTAMP(1);        // FIR + activation + split of stage 0 into LDS (waits for its patches), barrier
#pragma unroll (DEEP ? NSTAGE : 1)
  for (int st = 0; st < NSTAGE; ++st) {
    const float* cur = sB + (NBUF > 1 ? (st & 1) : 0) * BK * BN + nloc;
    f32x4 afr[KQ][WM];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
      for (int i = 0; i < WM; ++i) afr[kq][i] = afr_next[kq][i];
    if (st + 1 < NSTAGE) {   // next stage's operands travel under this stage's MFMAs
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int i = 0; i < WM; ++i)
          afr_next[kq][i] = *reinterpret_cast<const f32x4*>(
              ab + (((wm_i * WM + i) * (C / 16) + (st + 1) * KQ + kq) * 256 + lane * 4));
      if (!DEEP) patch_load(st + 1);
    }
    if (DEEP && st + 2 < NSTAGE) {           // set st & 1 was filtered into LDS at the end of stage st - 1: free
      if (st & 1) patch_load_into(st + 2, pw);
      else patch_load_into(st + 2, pv);
    }
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      if (BF16) {
        f32x4 b4[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) b4[j4] = *reinterpret_cast<const f32x4*>(cur + (kq * 16 + j4 * 4 + q) * BN);
        s16x4 ah[WM], bh[4];
#pragma unroll
        for (int i = 0; i < WM; ++i) ah[i] = pack_bf16(afr[kq][i][0], afr[kq][i][1], afr[kq][i][2], afr[kq][i][3]);
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[c] = pack_bf16(b4[0][c], b4[1][c], b4[2][c], b4[3][c]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[i], bh[c], acc[i][c], 0, 0, 0);
      } else if constexpr (SPLIT) {
        f32x4 b4[4];
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) b4[j4] = *reinterpret_cast<const f32x4*>(cur + (kq * 16 + j4 * 4 + q) * BN);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h4 bh, bl;
          unpack_frag(b4[0][c], b4[1][c], b4[2][c], b4[3][c], bh, bl);
#pragma unroll
          for (int i = 0; i < WM; ++i) acc[i][c] = split_mfma16(afr[kq][i], bh, bl, acc[i][c]);
        }
      } else {
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(cur + (kq * 16 + j4 * 4 + q) * BN);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[i][c] = fused_mfma(afr[kq][i][j4], b4[c], acc[i][c]);
      }
      }
    }
    if (NBUF > 1 && st + 1 < NSTAGE) {
      if (DEEP && ((st + 1) & 1)) patch_store_from(st + 1, sB + ((st + 1) & 1) * BK * BN, pw);
      else patch_store(st + 1, sB + ((st + 1) & 1) * BK * BN);
    }
    __syncthreads();
  }

  FSTAMP(2);        // K loop (conv2 MFMAs, next stage's patches)
  // ---- epilogue of conv2: this lane holds channels o = (wm_i*WM+i)*16 + 4q + r at pixels (oy, ox .. ox+3)
  if (LATE) { load_epilogue_ops(); load_skip_ops(); }
  float prgb[3][4];
  float mxn = 0.f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int c = 0; c < 4; ++c) prgb[ch][c] = 0.f;
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int obase = (wm_i * WM + i) * 16 + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      if constexpr (SPLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] *= k2in;                // exact: conv2's weights carried 2^8, act1 2^-e1
      }
      // (chained split form: v = out2 2^-e2 from here on -- the B operand of the next GEMM; the ToRGB sums are scaled back once)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = (CIPS3D_FUSED_AB & 16) ? v[c] : lrelu02(fmaf(nw2_u, nz2[c], v[c]) + bias4[i][r]) * kact2;
      if (NEXT) {   // keep the activated value where the accumulator was: it is the next GEMM's B operand
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c][r] = v[c];
      }
      if (a.out2) {
        if constexpr (NEXT && SPLIT) {
          *reinterpret_cast<f32x4*>(a.out2 + (int64_t)b * C * HWo + ((obase + r) * HWo + oy * OW + ox)) =
              f32x4{v[0] * kback2, v[1] * kback2, v[2] * kback2, v[3] * kback2};
        } else {
          *reinterpret_cast<f32x4*>(a.out2 + (int64_t)b * C * HWo + ((obase + r) * HWo + oy * OW + ox)) = v;
        }
      }
      if (a.wm_rgb && !(CIPS3D_FUSED_AB & 4)) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float w = s_wrgb[ch * C + obase + r];
#pragma unroll
          for (int c = 0; c < 4; ++c) prgb[ch][c] = fmaf(w, v[c], prgb[ch][c]);
        }
      }
    }
  }
  FSTAMP(3);        // conv2 epilogue (noise, bias, leaky ReLU, ToRGB partial sums, out2 store)
  // ---- NEXT: the next stage's low-resolution GEMM y_next = Wn out2 from the registers.  acc[i][c][r] is channel
  // 16 (wm_i WM + i) + 4 q + r of pixel c: with Wn in the chained pack that IS the B operand of k-step r of k-group
  // (wm_i WM + i).  Each wave row covers its own WM k-groups (split K); rows 1.. park their partial in the (now free) B
  // stages and row 0 adds them in order and stores.
  constexpr int OTN = NEXT ? C / 32 : 1;           // o-tiles of the C/2 output channels
  // Wide stage (C = 256: eight wave rows over the same 64 pixels, eight output tiles): split K would need a 224 KB
  // exchange of partials.  Instead the ACTIVATIONS are exchanged: in two rounds half of the wave rows drop their 32
  // channels into the dead B stages (32 KB, lane-linear: the lane that produced a k-step's fragment is the lane that
  // needs it), and every wave accumulates ITS output tile over the full K.
  constexpr bool XCHG = NEXT && XPREF && OTN % WGM == 0 && WGM % 2 == 0;
  if (XCHG) {
    constexpr int GR = (WGM / 2) * WM;             // k-groups (16 channels) per round
    constexpr int TPW = XCHG ? OTN / WGM : 1;      // output tiles per wave
    static_assert(!XCHG || WGN * GR * 4 * 256 <= NBUF * BK * BN, "half of the activations must fit the B stages");
    const float* an = a.wm_next + (int64_t)b * (C / 2) * C + lane * 4;
    float* sx = sB + wn_i * GR * 4 * 256 + lane * 4;            // this pixel set's exchange area
    f32x4 accx[TPW][4];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int c = 0; c < 4; ++c) accx[tp][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      f32x4 af[TPW][GR];
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int gl = 0; gl < GR; ++gl)
          af[tp][gl] = *reinterpret_cast<const f32x4*>(an + ((wm_i * TPW + tp) * (C / 16) + rd * GR + gl) * 256);
      if (wm_i / (WGM / 2) == rd) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(sx + (((wm_i % (WGM / 2)) * WM + i) * 4 + r) * 256) =
                SPLIT ? f32x4{pack_split(acc[i][0][r]), pack_split(acc[i][1][r]), pack_split(acc[i][2][r]), pack_split(acc[i][3][r])}
                      : f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      }
      __syncthreads();
#pragma unroll
      for (int gl = 0; gl < GR; ++gl) {
        f32x4 bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = *reinterpret_cast<const f32x4*>(sx + (gl * 4 + r) * 256);
        if (BF16) {
          s16x4 bh[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) bh[c] = pack_bf16(bv[0][c], bv[1][c], bv[2][c], bv[3][c]);
#pragma unroll
          for (int tp = 0; tp < TPW; ++tp) {
            const s16x4 ah = pack_bf16(af[tp][gl][0], af[tp][gl][1], af[tp][gl][2], af[tp][gl][3]);
#pragma unroll
            for (int c = 0; c < 4; ++c) accx[tp][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh[c], accx[tp][c], 0, 0, 0);
          }
        } else if constexpr (SPLIT) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            h4 bh, bl;
            unpack_frag(bv[0][c], bv[1][c], bv[2][c], bv[3][c], bh, bl);
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp) accx[tp][c] = split_mfma16(af[tp][gl], bh, bl, accx[tp][c]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
              for (int c = 0; c < 4; ++c)
                accx[tp][c] = fused_mfma(af[tp][gl][r], bv[r][c], accx[tp][c]);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      ylo_t* yn = reinterpret_cast<ylo_t*>(a.y_next) + (int64_t)b * (C / 2) * HWo + (((wm_i * TPW + tp) * 16 + 4 * q) * HWo + oy * OW + ox);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ys = SPLIT ? kyn : 1.f;                  // the chained weights carried 2^8 as well, out2 2^-e2
        const f32x4 yv = {accx[tp][0][r] * ys, accx[tp][1][r] * ys, accx[tp][2][r] * ys, accx[tp][3][r] * ys};
        mxn = fmaxf(fmaxf(fmaxf(fabsf(yv[0]), fabsf(yv[1])), fmaxf(fabsf(yv[2]), fabsf(yv[3]))), mxn);
        if constexpr (YB) cips3d_store_wt8(yn + r * HWo, pack_bf16(yv[0], yv[1], yv[2], yv[3]));
        else cips3d_store_wt16(yn + r * HWo, yv);
      }
    }
    if (a.next_amax && !(a.next_gain > 0.f) && !(CIPS3D_FUSED_AB & 2)) {     // the workgroup's largest |y_next| raises one slot of the sample's amax array; s_nz1 is dead (K loop over)
      static_assert(!NEXT || 4 * WGM * WGN <= BN, "the reduction words fit the noise tile");
      const float m = cips3d_workgroup_max(mxn, s_nz1, wave, lane, WGM * WGN);
      if (tid == 0) cips3d_amax_raise_if(a.next_amax + b * CIPS3D_AMAX_FLOATS, m, blockIdx.x);
    }
  }
  f32x4 accn[OTN][4];
  if (NEXT && !XCHG) {
    static_assert(!NEXT || XCHG || (WGM - 1) * WGN * OTN * 4 * 64 * 4 <= NBUF * BK * BN * 4, "split-K partials must fit the B stages");
    const float* an = a.wm_next + (int64_t)b * (C / 2) * C;      // packed [ot'][kq][256]
    f32x4 afn[OTN][WM];
#pragma unroll
    for (int t = 0; t < OTN; ++t)
#pragma unroll
      for (int i = 0; i < WM; ++i)
        afn[t][i] = *reinterpret_cast<const f32x4*>(an + ((t * (C / 16) + wm_i * WM + i) * 256 + lane * 4));
#pragma unroll
    for (int t = 0; t < OTN; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) accn[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      if (BF16) {
        s16x4 bh[4], ah[OTN];
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[c] = pack_bf16(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3]);
#pragma unroll
        for (int t = 0; t < OTN; ++t) ah[t] = pack_bf16(afn[t][i][0], afn[t][i][1], afn[t][i][2], afn[t][i][3]);
#pragma unroll
        for (int t = 0; t < OTN; ++t)
#pragma unroll
          for (int c = 0; c < 4; ++c) accn[t][c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[t], bh[c], accn[t][c], 0, 0, 0);
      } else if constexpr (SPLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h4 bh, bl;
          split_frag(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3], bh, bl);
#pragma unroll
          for (int t = 0; t < OTN; ++t) accn[t][c] = split_mfma16(afn[t][i], bh, bl, accn[t][c]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int t = 0; t < OTN; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              accn[t][c] = fused_mfma(afn[t][i][r], acc[i][c][r], accn[t][c]);
      }
    }
    if (wm_i > 0) {
      float* sp = sB + ((((wm_i - 1) * WGN + wn_i) * OTN * 4) * 256 + lane * 4);
#pragma unroll
      for (int t = 0; t < OTN; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(sp + (t * 4 + c) * 256) = accn[t][c];
    }
  }
  FSTAMP(4);        // chained next-stage GEMM (MFMAs)
  if (!a.wm_rgb && !NEXT) { FSTAMP_FLUSH(); return; }
  // ---- ToRGB: reduce over the 4 lane quarters, then over the WGM wave rows through LDS
  if (a.wm_rgb) {
    // Quarter q ends up with colour channel q of the wave's four pixels, in the order the shuffles added them ((q0 + q1) + (q2 +
    // q3): same bits).  v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second,
    // v_permlane32_swap the upper half of the first with the lower half of the second: two colours share a register after
    // the first step, all three after the second -- 3 swaps + 3 adds per pixel where the ds_bpermute form spent 6 + 6.
    // (not at C = 32: the swaps work in place on both operands, the copies do not fit that instantiation's 64 registers)
    if constexpr (MINW >= 8) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = prgb[ch][c];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          prgb[ch][c] = (NEXT && SPLIT) ? v * kback2 : v;
        }
      if (q == 0) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          *reinterpret_cast<f32x4*>(s_red + (wm_i * 3 + ch) * BN + nloc) =
              f32x4{prgb[ch][0], prgb[ch][1], prgb[ch][2], prgb[ch][3]};
      }
    } else {
    f32x4 tot;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const auto rg = __builtin_amdgcn_permlane16_swap(__float_as_uint(prgb[0][c]), __float_as_uint(prgb[1][c]), false, false);
      const auto bb = __builtin_amdgcn_permlane16_swap(__float_as_uint(prgb[2][c]), __float_as_uint(prgb[2][c]), false, false);
      const float p = __uint_as_float(rg[0]) + __uint_as_float(rg[1]);        // [R01, G01, R23, G23] by quarter
      const float u = __uint_as_float(bb[0]) + __uint_as_float(bb[1]);        // [B01, B01, B23, B23]
      const auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(u), false, false);
      const float v = __uint_as_float(h[0]) + __uint_as_float(h[1]);          // [R, G, B, B]
      tot[c] = (NEXT && SPLIT) ? v * kback2 : v;
    }
    if (q < 3) *reinterpret_cast<f32x4*>(s_red + (wm_i * 3 + q) * BN + nloc) = tot;
    }
  }
  __syncthreads();
  if (NEXT && !XCHG && wm_i == 0) {
#pragma unroll
    for (int m = 1; m < WGM; ++m) {
      const float* sp = sB + ((((m - 1) * WGN + wn_i) * OTN * 4) * 256 + lane * 4);
#pragma unroll
      for (int t = 0; t < OTN; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 pv4 = *reinterpret_cast<const f32x4*>(sp + (t * 4 + c) * 256);
#pragma unroll
          for (int r = 0; r < 4; ++r) accn[t][c][r] += pv4[r];
        }
    }
    ylo_t* yn = reinterpret_cast<ylo_t*>(a.y_next) + (int64_t)b * (C / 2) * HWo + (oy * OW + ox);
#pragma unroll
    for (int t = 0; t < OTN; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ys = SPLIT ? kyn : 1.f;
        const f32x4 yv = {accn[t][0][r] * ys, accn[t][1][r] * ys, accn[t][2][r] * ys, accn[t][3][r] * ys};
        mxn = fmaxf(fmaxf(fmaxf(fabsf(yv[0]), fabsf(yv[1])), fmaxf(fabsf(yv[2]), fabsf(yv[3]))), mxn);
        if constexpr (YB) cips3d_store_wt8(yn + (t * 16 + 4 * q + r) * HWo, pack_bf16(yv[0], yv[1], yv[2], yv[3]));
        else cips3d_store_wt16(yn + (t * 16 + 4 * q + r) * HWo, yv);
      }
  }
  if (NEXT && !XCHG && a.next_amax && !(a.next_gain > 0.f) && !(CIPS3D_FUSED_AB & 2)) {     // (only wave row 0 stored: the other rows bring 0)
    const float m = cips3d_workgroup_max(mxn, s_nz1, wave, lane, WGM * WGN);
    if (tid == 0) cips3d_amax_raise_if(a.next_amax + b * CIPS3D_AMAX_FLOATS, m, blockIdx.x);
  }
  FSTAMP(5);        // partial exchange of the chained GEMM / ToRGB through LDS, y_next store
  if (!a.wm_rgb) { FSTAMP_FLUSH(); return; }
  if (rgb_lane) {                     // quarter q finishes colour channel q
    const int ch = q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < WGM; ++m) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(s_red + (m * 3 + ch) * BN + nloc);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += t[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += rgb_bias;
    if (a.skip) {
      if (!FLAT && (a.skip_up & 1)) {
        float so[2][4];
        up2_fir(skp, kf, so);
        const int py = oy & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += py ? so[1][c] : so[0][c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += skv[c];
      }
    }
    if (a.skip_up & 2) {
      // CIPS3D_RGB_U8: the image leaves as uint8 -- cips3d_rgb_to_uint8's arithmetic (clamp to [-1, 1], (c + 1) 127.5, round to
      // nearest even) on the value that would have been stored: the multi-view loop's img_tensor_to_pil step
      // (render_video_web_v10.py:1825-1826) without the fp32 image's round trip through memory
      unsigned pk = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float cl = fminf(fmaxf(v[c], -1.f), 1.f);
        pk |= (unsigned)__float2int_rn((cl + 1.f) * 127.5f) << (8 * c);
      }
      *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(a.rgb) + (int64_t)b * 3 * HWo + (ch * HWo + oy * OW + ox)) = pk;
    } else {
      cips3d_store_wt16(a.rgb + (int64_t)b * 3 * HWo + (ch * HWo + oy * OW + ox), v);
    }
  }
  FSTAMP(6);        // rgb: bias, FIR of the skip, store
  FSTAMP_FLUSH();
}

template <int C, int WM, int WGM, int WGN, int RW, int BK, int MINW, bool NEXT, bool XPREF, bool LATE, bool FLAT>
int launch_fused_as(const FusedArgs& a, hipStream_t st) {
  constexpr int TH = RW * WGN, TW = 64 / RW;
  const int up = FLAT ? 1 : 2;
  dim3 grid((unsigned)((up * a.W / TW) * (up * a.H / TH)), 1, (unsigned)a.B);
  if (a.bf16 == 3) hipLaunchKernelGGL((fused_up_conv_kernel<C, WM, WGM, WGN, RW, BK, MINW, 3, NEXT, XPREF, LATE, FLAT>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else if (a.bf16 == 2) hipLaunchKernelGGL((fused_up_conv_kernel<C, WM, WGM, WGN, RW, BK, MINW, 2, NEXT, XPREF, LATE, FLAT>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else if (a.bf16) hipLaunchKernelGGL((fused_up_conv_kernel<C, WM, WGM, WGN, RW, BK, MINW, 1, NEXT, XPREF, LATE, FLAT>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else hipLaunchKernelGGL((fused_up_conv_kernel<C, WM, WGM, WGN, RW, BK, MINW, 0, NEXT, XPREF, LATE, FLAT>), grid, dim3(64 * WGM * WGN), 0, st, a);
  return cips3d_launch_status();
}
template <int C, int WM, int WGM, int WGN, int RW, int BK, int MINW, bool NEXT = false, bool XPREF = false, bool LATE = false>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  return a.flat ? launch_fused_as<C, WM, WGM, WGN, RW, BK, MINW, NEXT, XPREF, LATE, true>(a, st)
                : launch_fused_as<C, WM, WGM, WGN, RW, BK, MINW, NEXT, XPREF, LATE, false>(a, st);
}

}  // namespace

#ifdef CIPS3D_FUSED_STAMPS
extern "C" int cips3d_debug_read_fused_stamps(unsigned long long* out32) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_fused_stamps), 256);
  unsigned long long z[32] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), z, 256);
  return 0;
}
#endif

extern "C" int cips3d_modulate_weights(const float* W, const float* s, int64_t s_stride, float* wm, int B, int Cout,
                                       int Cin, int ksq, float scale, int demodulate, void* stream) {
  if (!W || !s || !wm || B < 0 || Cout <= 0 || Cin <= 0 || ksq <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  // negative ksq is not used; the packed layout is selected with the high bit of `demodulate`
  const int packed = (demodulate & 2) ? (((demodulate & 4) ? 2 : 1) | (demodulate & 248)) : 0;   // bit 3: flipped taps, bit 4: split-fp16, bit 5: split16, bit 6: transposed, bit 7: bf16
  if ((packed & 32) && (ksq != 1 || (packed & 24))) return CIPS3D_E_UNSUPP;
  if ((packed & 64) && (ksq != 1 || (packed & 63) != 1)) return CIPS3D_E_UNSUPP;
  if (packed && ksq == 1 && (Cout % 32 != 0 || Cin % 8 != 0)) return CIPS3D_E_UNSUPP;
  if ((packed & 16) && ksq == 1 && ((packed & 7) != 1 || Cin % 32 != 0)) return CIPS3D_E_UNSUPP;
  if ((packed & 16) && ksq != 1 && (ksq != 9 || (packed & ~8) != 17)) return CIPS3D_E_UNSUPP;      // 3x3: PACKED | SPLIT [| FLIP] only
  if ((packed & 128) && (ksq != 1 || (packed & 127) != 1 || Cin % 32 != 0)) return CIPS3D_E_UNSUPP;
  if (packed && ksq != 1 && (ksq != 9 || (packed & 7) != 1 || Cout % 16 != 0 || Cin % 16 != 0)) return CIPS3D_E_UNSUPP;
  const int64_t rows = (int64_t)B * Cout;
  hipLaunchKernelGGL(modulate_kernel, dim3((unsigned)ceil_div<int64_t>(rows, 4)), dim3(256), 0, as_stream(stream), W,
                     s, s_stride, wm, B, Cout, Cin, ksq, scale, demodulate & 1, packed);
  return cips3d_launch_status();
}

extern "C" int cips3d_modulate_table(const cips3d_modulate_desc* table_dev, int n_desc, int total_rows, int B, float noise_bound,
                                     void* stream) {
  if (!table_dev || n_desc <= 0 || total_rows <= 0 || B < 0 || !(noise_bound >= 0.f)) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(modulate_table_kernel, dim3((unsigned)ceil_div(total_rows, 4), (unsigned)B), dim3(256), 0,
                     as_stream(stream), table_dev, n_desc, total_rows, noise_bound);
  return cips3d_launch_status();
}

__global__ void __launch_bounds__(256) split_words_kernel(const float* __restrict__ x, float k, unsigned* __restrict__ words,
                                                          unsigned* __restrict__ pairs, int64_t n) {
  const float ku = cips3d_uniform(k);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += (int64_t)gridDim.x * 256) {
    words[2 * i] = cips3d_split_word(x[2 * i], ku);
    words[2 * i + 1] = cips3d_split_word(x[2 * i + 1] * k);
    cips3d_split_pair(x[2 * i], x[2 * i + 1], pairs[2 * i], pairs[2 * i + 1]);
  }
}

extern "C" int cips3d_split_words(const float* x, float k, uint32_t* words, uint32_t* pairs, int64_t n, void* stream) {
  if (!x || !words || !pairs || n <= 0 || (n & 1)) return CIPS3D_E_BADARG;
  int64_t blocks = ceil_div<int64_t>(n / 2, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_words_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, k, words, pairs, n);
  return cips3d_launch_status();
}

extern "C" int cips3d_amax_layout(int* slots, int* stride) {
  if (slots) *slots = CIPS3D_AMAX_SLOTS;
  if (stride) *stride = CIPS3D_AMAX_STRIDE;
  return 0;
}

extern "C" int cips3d_absmax(const float* x, int B, int64_t n, float* amax, void* stream) {
  if (!x || !amax || B < 0 || n <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipError_t e = hipMemsetAsync(amax, 0, sizeof(float) * CIPS3D_AMAX_FLOATS * (size_t)B, as_stream(stream));
  if (e != hipSuccess) return (int)e;
  int64_t blocks = ceil_div<int64_t>(n, 256 * 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, as_stream(stream), x, n, amax);
  return cips3d_launch_status();
}

extern "C" int cips3d_absmax_raise(const float* x, int B, int64_t n, float* amax, void* stream) {
  if (!x || !amax || B < 0 || n <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>(n, 256 * 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, as_stream(stream), x, n, amax);
  return cips3d_launch_status();
}

extern "C" int cips3d_range_consts(const float* bias, int n_bias, const float* noise_w, float noise_bound, const float* noise_amax,
                                   float w_gain, const float* fir, float* lconst, int B, void* stream) {
  if (!lconst || B < 0 || n_bias < 0 || (n_bias > 0 && !bias) || !(noise_bound >= 0.f) || !(w_gain >= 0.f)) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(range_consts_kernel, dim3((unsigned)B), dim3(64), 0, as_stream(stream), bias, n_bias, noise_w, noise_bound,
                     noise_amax, w_gain, fir, lconst);
  return cips3d_launch_status();
}

extern "C" int cips3d_modconv1x1_supported(int Cin, int Cout, int64_t HW) {
  // (the kernel keeps intra-sample offsets in 32 bits)
  return (Cin % 32 == 0) && (Cout % 32 == 0) && (HW % 4 == 0) && HW >= 4 &&
         (int64_t)(Cin > Cout ? Cin : Cout) * HW + 256 < ((int64_t)1 << 31);
}

extern "C" int cips3d_modconv1x1(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int64_t HW,
                                 int epilogue, const float* noise, int64_t noise_bstride, const float* noise_w,
                                 const float* bias, const cips3d_range* rg, void* stream) {
  return cips3d_modconv1x1_torgb(x, wm, out, B, Cin, Cout, HW, epilogue, noise, noise_bstride, noise_w, bias, nullptr,
                                 nullptr, nullptr, rg, stream);
}

// rows per workgroup of the tile configuration cips3d_modconv1x1 picks for this Cout (= row blocks of the ToRGB partials)
static int gemm_block_rows(int Cout) {
  if (Cout >= 256 && Cout % 64 == 0) return 64;
  if (Cout == 128) return 128;
  if (Cout == 64) return 64;
  return 32;
}

extern "C" int cips3d_modconv1x1_torgb(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int64_t HW,
                                       int epilogue, const float* noise, int64_t noise_bstride, const float* noise_w,
                                       const float* bias, const float* rgb_w, float* rgb_part, int* n_row_blocks,
                                       const cips3d_range* rg, void* stream) {
  if ((rgb_w == nullptr) != (rgb_part == nullptr)) return CIPS3D_E_BADARG;
  // 128-row tiles for a grid of more than one 64 x 128 tile per CU (see below); the ToRGB partials then come in 128-row blocks
  static const int big_tiles = getenv("CIPS3D_GEMM_BIG") ? atoi(getenv("CIPS3D_GEMM_BIG")) : 1;
  const bool big = big_tiles && Cout >= 256 && Cout % 128 == 0 && Cin % 64 == 0 && HW > 0 && B > 0 &&
                   (int64_t)B * (Cout / 64) * ceil_div<int64_t>(HW, 128) > 256;
  if (n_row_blocks) *n_row_blocks = Cout > 0 ? Cout / (big ? 128 : gemm_block_rows(Cout)) : 0;
  if (!x || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if ((epilogue & CIPS3D_GEMM_BF16) && (epilogue & CIPS3D_GEMM_SPLIT)) return CIPS3D_E_BADARG;
  const int bf16 = (epilogue & CIPS3D_GEMM_BF16) ? 1 : (epilogue & CIPS3D_GEMM_SPLIT) ? 2 : 0;
  const int out_bf16 = (epilogue & CIPS3D_Y_BF16) ? 1 : 0;
  epilogue &= ~(CIPS3D_GEMM_BF16 | CIPS3D_Y_BF16 | CIPS3D_GEMM_SPLIT);
  if (epilogue != 0 && epilogue != 1) return CIPS3D_E_BADARG;
  if (out_bf16 && (epilogue != 0 || rgb_part)) return CIPS3D_E_BADARG;     // bf16 storage is for the pre-FIR GEMM result only
  if (epilogue == 1 && !bias) return CIPS3D_E_BADARG;
  if (!cips3d_modconv1x1_supported(Cin, Cout, HW)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  GemmArgs a{x, wm, out, B, Cin, Cout, HW, epilogue, noise, noise_bstride, noise_w, bias, bf16, out_bf16, rgb_w, rgb_part,
             rg ? rg->x_amax : nullptr, rg ? rg->out_amax : nullptr, cips3d_actbwd{}};
  hipStream_t st = as_stream(stream);
  static const int dbg_cfg = getenv("CIPS3D_GEMM_CFG") ? atoi(getenv("CIPS3D_GEMM_CFG")) : 0;   // tuning knob (tools/)
  if (dbg_cfg && !rgb_part && Cout % 128 == 0) {      // (the ToRGB fold needs the default tiling: gemm_block_rows)
    switch (dbg_cfg) {
      case 1: return launch_gemm<2, 2, 2, 32, 4>(a, st);    // 64 x 128, 4 waves
      case 2: return launch_gemm<1, 4, 2, 32, 4>(a, st);    // 64 x 128, 8 waves
      case 3: return launch_gemm<2, 4, 2, 32, 3>(a, st);    // 128 x 128, 8 waves
      case 4: return launch_gemm<1, 4, 2, 32, 2>(a, st);    // 64 x 128, 8 waves, 1 stage in flight
      case 5: return launch_gemm<2, 2, 4, 32, 3>(a, st);    // 64 x 256, 8 waves
      case 6: return launch_gemm<1, 8, 1, 32, 4>(a, st);    // 128 x 64, 8 waves
      case 7: return launch_gemm<1, 2, 2, 32, 4>(a, st);    // 32 x 128, 4 waves
      case 8: return launch_gemm<2, 4, 1, 32, 2>(a, st);    // 128 x 64, 8 waves, 8 accumulator chains per wave
      case 9: return launch_gemm<2, 4, 1, 32, 3>(a, st);
      case 10: return launch_gemm<2, 2, 2, 32, 2>(a, st);   // 64 x 128, 4 waves, 8 chains
    }
  }
  if (Cout >= 256 && Cout % 64 == 0) {
    // 64 x 128 tiles.  A grid that covers at most half of the 256 CUs (the 512 -> 256 low-res GEMM at 64^2: 128 tiles)
    // runs on smaller tiles instead; not for the ToRGB fold, whose slot count follows gemm_block_rows()
    static const int small_cfg = getenv("CIPS3D_GEMM_SMALL") ? atoi(getenv("CIPS3D_GEMM_SMALL")) : 1;
    if (!rgb_part && small_cfg && (int64_t)B * (Cout / 64) * ceil_div<int64_t>(HW, 128) <= 128) {
      if (small_cfg == 1) return launch_gemm<1, 2, 2, 32, 4>(a, st);    // 32 x 128, 4 waves
      if (small_cfg == 2) return launch_gemm<1, 4, 1, 32, 4>(a, st);    // 64 x 64, 4 waves
    }
    // 8 waves; 64-deep stages (half as many stage barriers: 705.9 -> 703.4 us per 1024^2 view, A/B on one box) when K allows
    // ... for a grid of at most ONE workgroup per CU (the 64^2 layers of a single view: 256 tiles).  A larger grid (batch 2 and
    // up: the inversion step) runs 32-deep stages in a 48 KB ring instead: two workgroups share a CU and one's prologue /
    // epilogue -- a third of a launch, with the L2 port idle -- runs under the other's K loop (inversion step +3.4 %, same box)
    static const int bk = getenv("CIPS3D_GEMM_BK") ? atoi(getenv("CIPS3D_GEMM_BK")) : 0;       // tuning knob (tools/): 32 | 64
    const bool one_round = (int64_t)B * (Cout / 64) * ceil_div<int64_t>(HW, 128) <= 256;
    if (!big && (bk ? bk == 64 : one_round) && Cin % 64 == 0) return launch_gemm<1, 4, 2, 64, 2>(a, st);
    // a grid of more than one 64 x 128 tile per CU (batch 2 and up: the inversion step): 128 x 128 tiles, 64-deep stages -- half
    // the operand bytes per flop through the L2 port and one workgroup per CU again.  Inversion step, same box x3: 341.6-343.2 ->
    // 348.8-349.4 steps/s with this form in the forward and the data-gradient GEMM (32-deep in 3 / 4 slots: 347.1-348.7 / 346.9-347.7)
    if (big) return launch_gemm<2, 4, 2, 64, 2>(a, st);
    return launch_gemm<1, 4, 2, 32, 2>(a, st);
  }
  static const int ns2 = getenv("CIPS3D_GEMM_NS2") ? atoi(getenv("CIPS3D_GEMM_NS2")) : 1;     // (see cips3d_modconv1x1_actbwd)
  if (ns2 && Cin <= 128) {      // (same rows per workgroup as the 4-slot forms: the ToRGB fold's slot count does not change)
    if (Cout == 128) return launch_gemm<1, 8, 1, 32, 2>(a, st);
    if (Cout == 64) return launch_gemm<1, 4, 2, 32, 2>(a, st);
    if (Cout < 256) return launch_gemm<1, 2, 2, 32, 2>(a, st);
  }
  if (Cout == 128) return launch_gemm<1, 8, 1, 32, 4>(a, st);                     // all 128 rows: x read once
  if (Cout == 64) return launch_gemm<1, 4, 2, 32, 4>(a, st);                      // 64 x 128
  return launch_gemm<1, 2, 2, 32, 4>(a, st);                                      // 32 x 128 (any Cout % 32 == 0)
}

extern "C" int cips3d_modconv1x1_actbwd(const float* g, const float* wm_t, float* dpre, int B, int Cin, int Cout, int64_t HW,
                                        int flags, const cips3d_actbwd* ab, const float* noise, int64_t noise_bstride,
                                        const cips3d_range* rg, void* stream) {
  if (!g || !wm_t || !dpre || !ab || !ab->y || B < 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (flags & ~CIPS3D_GEMM_SPLIT) return CIPS3D_E_BADARG;
  if ((ab->drgb == nullptr) != (ab->rgb_w == nullptr) || (ab->d_rgb_w && !ab->drgb)) return CIPS3D_E_BADARG;
  if (ab->d_noise_w && !noise) return CIPS3D_E_BADARG;
  if (ab->slots > 1 && ((ab->slots & (ab->slots - 1)) || ab->slot_stride < Cout || (ab->d_rgb_w && ab->rgb_slot_stride < B * 3 * Cout)))
    return CIPS3D_E_BADARG;
  if (!cips3d_modconv1x1_supported(Cin, Cout, HW)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  GemmArgs a{g, wm_t, dpre, B, Cin, Cout, HW, 0, noise, noise_bstride, nullptr, nullptr, (flags & CIPS3D_GEMM_SPLIT) ? 2 : 0, 0,
             nullptr, nullptr, rg ? rg->x_amax : nullptr, rg ? rg->out_amax : nullptr, *ab};
  hipStream_t st = as_stream(stream);
  if (Cout >= 256 && Cout % 64 == 0) {
    // (as cips3d_modconv1x1: 64-deep stages for a grid of at most one workgroup per CU, else the 48 KB ring and two per CU)
    static const int bk = getenv("CIPS3D_GEMM_BK") ? atoi(getenv("CIPS3D_GEMM_BK")) : 0;       // tuning knob (tools/): 32 | 64
    const bool one_round = (int64_t)B * (Cout / 64) * ceil_div<int64_t>(HW, 128) <= 256;
    if ((bk ? bk == 64 : one_round) && Cin % 64 == 0) return launch_gemm_actbwd<1, 4, 2, 64, 2>(a, st);
    static const int big_tiles = getenv("CIPS3D_GEMM_BIG") ? atoi(getenv("CIPS3D_GEMM_BIG")) : 1;      // (see cips3d_modconv1x1_torgb)
    if (big_tiles && Cout % 128 == 0 && Cin % 64 == 0) return launch_gemm_actbwd<2, 4, 2, 64, 2>(a, st);
    return launch_gemm_actbwd<1, 4, 2, 32, 2>(a, st);
  }
  // Short contractions (K <= 128: the decoder's 128 / 64 / 32-channel layers at 256^2 and above) are HBM-bound streams with
  // two to four K stages: a 2-slot ring (half the LDS) lets a second workgroup share the CU and cover the first one's
  // epilogue loads and stores
  static const int ns2 = getenv("CIPS3D_GEMM_NS2") ? atoi(getenv("CIPS3D_GEMM_NS2")) : 1;
  if (ns2 && Cin <= 128) {
    if (Cout == 128) return launch_gemm_actbwd<1, 8, 1, 32, 2>(a, st);
    if (Cout == 64) return launch_gemm_actbwd<1, 4, 2, 32, 2>(a, st);
    if (Cout < 256) return launch_gemm_actbwd<1, 2, 2, 32, 2>(a, st);
  }
  if (Cout == 128) return launch_gemm_actbwd<1, 8, 1, 32, 4>(a, st);
  if (Cout == 64) return launch_gemm_actbwd<1, 4, 2, 32, 4>(a, st);
  return launch_gemm_actbwd<1, 2, 2, 32, 4>(a, st);
}

extern "C" int cips3d_up2_fir_act(const float* y_lo, const float* fir, float* out, int B, int C, int H, int W,
                                  const float* noise, int64_t noise_bstride, const float* noise_w, const float* bias,
                                  float* out_amax, void* stream) {
  if (!y_lo || !fir || !out || !bias || B < 0 || C <= 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (W % 2 != 0) return CIPS3D_E_UNSUPP;     // 16-byte output quads
  if (B == 0) return 0;
  const int64_t total = (int64_t)B * C * H * (W / 2);
  int64_t blocks = ceil_div<int64_t>(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(up2_fir_act_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), y_lo, fir, out, B,
                     C, H, W, noise, noise_bstride, noise_w, bias, out_amax);
  return cips3d_launch_status();
}

extern "C" int cips3d_noise_bias_act(const float* x, const float* noise, int64_t noise_bstride, const float* noise_w,
                                     const float* bias, float* out, int B, int C, int64_t HW, void* stream) {
  if (!x || !bias || !out || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  int64_t blocks = ceil_div<int64_t>((int64_t)B * C * HW, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(noise_bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, noise,
                     noise_bstride, noise_w, bias, out, B, C, HW);
  return cips3d_launch_status();
}

extern "C" int cips3d_fused_up_conv_supported(int C, int H, int W) {
  // (the kernel keeps intra-sample offsets in 32 bits: C * 2H * 2W must stay below 2^31)
  return (C == 32 || C == 64 || C == 128 || C == 256) && W % 32 == 0 && H % 2 == 0 && H >= 2 &&
         (int64_t)C * 4 * H * W < ((int64_t)1 << 31);
}

extern "C" int cips3d_fused_flat_conv_supported(int C, int H, int W) {
  // the same tiles over an H x W output: 64 pixels per wave row, up to 4 image rows per workgroup
  return (C == 32 || C == 64 || C == 128 || C == 256) && W % 64 == 0 && H % 4 == 0 && H >= 4 &&
         (int64_t)C * H * W < ((int64_t)1 << 31);
}

extern "C" int cips3d_fused_up_conv(const float* y_lo, const float* fir, const float* noise1, int64_t noise1_bstride,
                                    const float* noise_w1, const float* bias1, const float* wm2, const float* noise2,
                                    int64_t noise2_bstride, const float* noise_w2, const float* bias2, float* out2,
                                    const float* wm_rgb, const float* bias_rgb, const float* skip, int skip_up,
                                    float* rgb, int B, int C, int H, int W, const cips3d_range* rg, void* stream) {
  return cips3d_fused_up_conv_next(y_lo, fir, noise1, noise1_bstride, noise_w1, bias1, wm2, noise2, noise2_bstride, noise_w2,
                                   bias2, out2, wm_rgb, bias_rgb, skip, skip_up, rgb, nullptr, nullptr, B, C, H, W, rg, stream);
}

extern "C" int cips3d_fused_up_conv_chains(int C) { return C == 64 || C == 128 || C == 256; }

extern "C" int cips3d_fused_up_conv_next(const float* y_lo, const float* fir, const float* noise1, int64_t noise1_bstride,
                                         const float* noise_w1, const float* bias1, const float* wm2, const float* noise2,
                                         int64_t noise2_bstride, const float* noise_w2, const float* bias2, float* out2,
                                         const float* wm_rgb, const float* bias_rgb, const float* skip, int skip_up,
                                         float* rgb, const float* wm_next, float* y_next, int B, int C, int H, int W,
                                         const cips3d_range* rg, void* stream) {
  const int flat = (skip_up & CIPS3D_STAGE_FLAT) ? 1 : 0;
  if (!y_lo || (!fir && !flat) || !bias1 || !wm2 || !bias2 || B < 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (flat && (skip_up & 1)) return CIPS3D_E_BADARG;           // a flat stage's skip image has the output's size
  if ((wm_next == nullptr) != (y_next == nullptr)) return CIPS3D_E_BADARG;
  if (wm_next && !cips3d_fused_up_conv_chains(C)) return CIPS3D_E_UNSUPP;
  if (!out2 && !wm_rgb && !wm_next) return CIPS3D_E_BADARG;
  if (wm_rgb && (!bias_rgb || !rgb)) return CIPS3D_E_BADARG;
  if (!(flat ? cips3d_fused_flat_conv_supported(C, H, W) : cips3d_fused_up_conv_supported(C, H, W))) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  FusedArgs a{y_lo, fir, noise1, noise1_bstride, noise_w1, bias1, wm2, noise2, noise2_bstride, noise_w2, bias2, out2,
              wm_rgb, bias_rgb, skip, (skip_up & 1) | ((skip_up & CIPS3D_RGB_U8) ? 2 : 0), rgb, B, H, W,
              (skip_up & CIPS3D_GEMM_SPLIT) ? 3 : (skip_up & CIPS3D_GEMM_BF16) ? ((skip_up & CIPS3D_Y_BF16) ? 2 : 1) : 0, wm_next,
              y_next, rg ? rg->x_amax : nullptr, rg ? rg->lconst : nullptr, rg ? rg->lconst2 : nullptr,
              rg ? rg->next_amax : nullptr, rg ? rg->next_gain : 0.f, flat};
  // split mode with range tracking: the bound of act1 needs conv1's constants, the chained form conv2's as well
  if ((skip_up & CIPS3D_GEMM_SPLIT) && rg && rg->x_amax && (!rg->lconst || (wm_next && !rg->lconst2))) return CIPS3D_E_BADARG;
  if ((skip_up & CIPS3D_Y_BF16) && !(skip_up & CIPS3D_GEMM_BF16)) return CIPS3D_E_BADARG;   // bf16 storage implies bf16 operands
  if ((skip_up & CIPS3D_GEMM_SPLIT) && (skip_up & CIPS3D_GEMM_BF16)) return CIPS3D_E_BADARG;
  hipStream_t st = as_stream(stream);
  if (wm_next)                                                            // cips3d_fused_up_conv_chains(C)
    // C = 128: 2 rows x 64 with four waves (512 workgroups) beats the unchained kernel's 4 x 64 / eight waves by 3 us once
    // the chained GEMM is in (sweep on one box, whole-view time)
    // how the stage's activations reach the chained GEMM: exchanged through LDS, every wave owning output tiles over the
    // full K (C = 256: the only form that fits; C = 64: -2.7 us per view against split K), or split K over the wave rows
    // with the partials added through LDS (C = 128: a tie, kept)
    // (last template flag: epilogue operands requested after the MFMAs -- at C = 64 / 128 that removes the spills of the
    // chained form, -2 us / neutral; at C = 256 it measured +1 us and stays off)
    return C == 64    ? launch_fused<64, 2, 2, 2, 1, 16, CIPS3D_C64_MINW, true, true, true>(a, st)
           : C == 128 ? launch_fused<128, 4, 2, 2, 1, 32, 2, true, false, CIPS3D_C128_LATE>(a, st)
                      : launch_fused<256, 2, 8, 1, 2, 64, 2, true, true>(a, st);
  switch (C) {
    // tile shapes / register budgets picked by sweep on MI355X (profiles/r01_i_*): time per stage in the comment
    // C = 32 (one K stage): with the epilogue's operands (noise2, bias2, skip patch) requested after the MFMAs instead of up
    // front the kernel needs 71 instead of 93 registers: seven instead of five resident waves per SIMD, -6 us at 1024^2
    // (round 5, what bounds it: 503.6 VALU instructions per wave x 32 768 waves = 64.5 k issue cycles per SIMD = 30.7 us of its
    // 37.3 -- the stage is VALU-issue bound.  With no arithmetic (-DCIPS3D_FUSED_AB=60) 201.7 VALU / 26.2 us, with no loads either
    // (252) 136.8 VALU / 16.7 us.  Not the wave launch rate: a wave that owns all 32 channels of its 64 pixels (<32, 2, 1, 4, 1, 32,
    // 4>: half the waves, no cross-wave ToRGB exchange) takes 39.6 against 39.4 us on one box, eight waves per workgroup 41.0.  The
    // conv2 epilogue written as channel-pair v_pk_*_f32 by hand: 37.4 against 37.5 -- hipcc's SLP pass had packed it already.)
    case 32: return launch_fused<32, 1, 2, 2, 1, 32, CIPS3D_C32_MINW, false, false, true>(a, st);   // 2 rows x 64, 4 waves     46 us @1024^2
    case 64: return launch_fused<64, 2, 2, 2, 1, 16, 4>(a, st);      // 2 rows x 64, 4 waves, BK 16     42 us @512^2
    case 128: return launch_fused<128, 4, 2, 4, 1, 32, 2>(a, st);    // 4 rows x 64                     33 us @256^2
    case 256: return launch_fused<256, 2, 8, 1, 2, 64, 2>(a, st);    // 2 rows x 32: 256 workgroups at 128^2
  }
  return CIPS3D_E_UNSUPP;
}

// out = skip + sum_s part[s] + sum_k bias_k : the fixed-order fold of the ToRGB partial sums written by
// cips3d_modconv1x1_torgb (slots = layers x row blocks, each [B][3][HW]).  One float4 per thread.
struct RgbReduceArgs {
  const float* part; int n_slots; int n_bias; const float* bias[CIPS3D_TORGB_FOLD_MAX]; const float* skip; float* out;
  int64_t n4, HW4, slot_stride;
};

// block = 64 float4 positions x 4 slot groups: group g adds slots g, g+4, ... (independent loads in flight), the groups
// meet in LDS in a fixed order.
__global__ void __launch_bounds__(256) torgb_reduce_kernel(RgbReduceArgs a) {
  __shared__ f32x4 s_p[4][64];
  const int pl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + pl;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (i < a.n4) {
    for (int s0 = g; s0 < a.n_slots; s0 += 16) {
      f32x4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        t[u] = s0 + 4 * u < a.n_slots ? *reinterpret_cast<const f32x4*>(a.part + (s0 + 4 * u) * a.slot_stride + i * 4)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += t[u][c];
    }
  }
  s_p[g][pl] = v;
  __syncthreads();
  if (g != 0 || i >= a.n4) return;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const f32x4 t = s_p[k][pl];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += t[c];
  }
  const int ch = (int)((i / a.HW4) % 3);
  float bs = 0.f;
  for (int k = 0; k < a.n_bias; ++k) bs += a.bias[k][ch];
#pragma unroll
  for (int c = 0; c < 4; ++c) v[c] += bs;
  if (a.skip) {
    const f32x4 sk = *reinterpret_cast<const f32x4*>(a.skip + i * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += sk[c];
  }
  *reinterpret_cast<f32x4*>(a.out + i * 4) = v;
}

extern "C" int cips3d_torgb_reduce(const float* part, int n_slots, const float* const* biases, int n_bias, const float* skip,
                                   float* out, int B, int64_t HW, void* stream) {
  if (!part || !out || n_slots < 1 || n_bias < 0 || n_bias > CIPS3D_TORGB_FOLD_MAX || (n_bias && !biases) || B < 0 || HW <= 0)
    return CIPS3D_E_BADARG;
  if (HW % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  RgbReduceArgs a{};
  a.part = part; a.n_slots = n_slots; a.n_bias = n_bias;
  for (int k = 0; k < n_bias; ++k) a.bias[k] = biases[k];
  a.skip = skip; a.out = out;
  a.n4 = (int64_t)B * 3 * HW / 4; a.HW4 = HW / 4; a.slot_stride = (int64_t)B * 3 * HW;
  hipLaunchKernelGGL(torgb_reduce_kernel, dim3((unsigned)ceil_div<int64_t>(a.n4, 64)), dim3(256), 0, as_stream(stream), a);
  return cips3d_launch_status();
}

extern "C" int cips3d_torgb(const float* x, const float* wm, const float* bias, const float* skip, int skip_up,
                            const float* fir, float* out, int B, int Cin, int H, int W, void* stream) {
  if (!x || !wm || !bias || !out || B < 0 || Cin <= 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (skip && skip_up && (!fir || (H % 2) || (W % 2))) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  if (W % 4 != 0) {
    int64_t bs = ceil_div<int64_t>((int64_t)B * H * W, 256);
    if (bs > 16384) bs = 16384;
    hipLaunchKernelGGL(torgb_scalar_kernel, dim3((unsigned)bs), dim3(256), 0, as_stream(stream), x, wm, bias, skip,
                       skip_up, fir, out, B, Cin, H, W);
    return cips3d_launch_status();
  }
  const int64_t quads = (int64_t)H * W / 4;
  {
    const size_t lds = sizeof(float) * (256 * 12 + 3 * (size_t)Cin);
    if (quads <= 32768 && Cin >= 64) {
      hipLaunchKernelGGL(torgb_split_kernel<16>, dim3((unsigned)ceil_div<int64_t>(quads, 16), (unsigned)B), dim3(256), lds,
                         as_stream(stream), x, wm, bias, skip, skip_up, fir, out, B, Cin, H, W);
      return cips3d_launch_status();
    }
    if (Cin >= 16) {
      hipLaunchKernelGGL(torgb_split_kernel<4>, dim3((unsigned)ceil_div<int64_t>(quads, 64), (unsigned)B), dim3(256), lds,
                         as_stream(stream), x, wm, bias, skip, skip_up, fir, out, B, Cin, H, W);
      return cips3d_launch_status();
    }
  }
  int64_t bx = ceil_div<int64_t>(quads, 256);
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(torgb_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), sizeof(float) * 3 * Cin,
                     as_stream(stream), x, wm, bias, skip, skip_up, fir, out, B, Cin, H, W);
  return cips3d_launch_status();
}

extern "C" int cips3d_modconv_kxk(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int H, int W,
                                  int k, int transpose2, void* stream) {
  if (!x || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || k <= 0 || (k % 2) == 0)
    return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  const int OH = transpose2 ? 2 * H - 1 + k - 1 : H, OW = transpose2 ? 2 * W - 1 + k - 1 : W;
  const int64_t total = (int64_t)B * Cout * OH * OW;
  int64_t blocks = ceil_div<int64_t>(total, 256);
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(modconv_kxk_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, wm, out, B, Cin,
                     Cout, H, W, k, transpose2, OH, OW);
  return cips3d_launch_status();
}
