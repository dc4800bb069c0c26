// One-call decoder forward + backward for the inversion loop (BASELINE config 5).
//
// The reference gets the whole backward from one `loss.backward()` over Decoder.forward
// (/root/reference/exp/cips3d/models/projector_v10.py:1203-1209, models/model_v3.py:592-637).  cips_3dplusplus_amd/autograd.py
// chains one torch.autograd.Function per HIP op instead: ~420 launches per step with a memset or a copy in front of every
// other one, and every activation gradient read and written once more by an activation-backward pass and once more by a
// ToRGB-backward pass.  Here the decoder is ONE autograd node:
//   cips3d_decoder_grad_forward   style heads (one table launch) -> modulated weights in the three forms the step needs
//                                 (one table launch) -> per layer the forward GEMM with its epilogue, every output kept
//   cips3d_decoder_grad_backward  one memset of every accumulator, then per StyledConv, last to first:
//                                   [FIR transpose of an up-sampling layer]  -> weight-gradient GEMM (split-fp16,
//                                   backward.hip) -> data-gradient GEMM whose epilogue IS the previous layer's activation
//                                   backward, ToRGB backward and parameter reductions (decoder.hip, ACTBWD);
//                                 then ONE launch for the modulation backward of all layers, the style-table backward, and a
//                                 tail launch for the scalar noise weights.
// Gradient operands of the split-fp16 GEMMs are scaled by their measured maxima (amax slots, cips3d_range).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------
// Activation backward without a GEMM in front of it: the LAST StyledConv's output is read by the last ToRGB only, so the
// gradient w.r.t. it is the rank-3 product rgb_w^T drgb (+ g_in when the caller has another contribution).  Same
// arithmetic and reductions as the ACTBWD epilogue of decoder.hip.  grid (ceil(HW / 1024), C, B).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) act_tail_kernel(const float* __restrict__ g_in, const float* __restrict__ y,
                                                       const float* __restrict__ rgb_w, const float* __restrict__ drgb,
                                                       const float* __restrict__ noise, int64_t noise_bstride,
                                                       float* __restrict__ dpre, float* __restrict__ d_bias,
                                                       float* __restrict__ d_nw, float* __restrict__ d_rgb_w,
                                                       float* __restrict__ out_amax, int C, int64_t HW, int slots,
                                                       int slot_stride, int rgb_slot_stride) {
  __shared__ float sh6[24];
  const int c = blockIdx.y, b = blockIdx.z;
  const int slot = slots > 1 ? (int)(blockIdx.x & (slots - 1)) : 0;     // (cips3d_actbwd: one cache line per slot and row block)
  if (d_bias) d_bias += (int64_t)slot * slot_stride;
  if (d_nw) d_nw += (int64_t)slot * slot_stride;
  if (d_rgb_w) d_rgb_w += (int64_t)slot * rgb_slot_stride;
  const int64_t base = ((int64_t)b * C + c) * HW;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float sb = 0.f, sn = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, mx = 0.f;
  if (p0 < HW) {
    f32x4 g = g_in ? *reinterpret_cast<const f32x4*>(g_in + base + p0) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + base + p0);
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0;
    if (drgb) {
      const float* gb = drgb + (int64_t)b * 3 * HW + p0;
      r0 = *reinterpret_cast<const f32x4*>(gb);
      r1 = *reinterpret_cast<const f32x4*>(gb + HW);
      r2 = *reinterpret_cast<const f32x4*>(gb + 2 * HW);
      const float w0 = rgb_w[((int64_t)b * 3 + 0) * C + c], w1 = rgb_w[((int64_t)b * 3 + 1) * C + c],
                  w2 = rgb_w[((int64_t)b * 3 + 2) * C + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = fmaf(w2, r2[e], fmaf(w1, r1[e], fmaf(w0, r0[e], g[e])));
    }
    const float S = 1.41421356237309515f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g[e] *= v[e] > 0.f ? S : 0.2f * S;
      sb += g[e];
      mx = fmaxf(mx, fabsf(g[e]));
      s0 = fmaf(r0[e], v[e], s0); s1 = fmaf(r1[e], v[e], s1); s2 = fmaf(r2[e], v[e], s2);
    }
    cips3d_store_wt16(dpre + base + p0, g);
    if (d_nw) {
      const f32x4 nz = *reinterpret_cast<const f32x4*>(noise + (int64_t)b * noise_bstride + p0);
#pragma unroll
      for (int e = 0; e < 4; ++e) sn = fmaf(g[e], nz[e], sn);
    }
  }
  // six block reductions through ONE barrier: wave sums / maximum first, one LDS word per (wave, quantity)
  sb = wave_sum(sb); sn = wave_sum(sn); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  mx = wave_max(mx);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    float* w = sh6 + wave * 6;
    w[0] = sb; w[1] = sn; w[2] = s0; w[3] = s1; w[4] = s2; w[5] = mx;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    const float t0 = sh6[k], t1 = sh6[6 + k], t2 = sh6[12 + k], t3 = sh6[18 + k];
    if (k < 5) {
      const float t = (t0 + t1) + (t2 + t3);
      float* dst = k == 0 ? d_bias : (k == 1 ? d_nw : d_rgb_w);
      if (dst) unsafeAtomicAdd(k < 2 ? dst + c : dst + ((int64_t)b * 3 + (k - 2)) * C + c, t);
    } else if (out_amax) {
      cips3d_amax_raise_if(out_amax + b * CIPS3D_AMAX_FLOATS, fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)), blockIdx.y * gridDim.x + blockIdx.x);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Transpose of the 2x FIR up-sampler (upfirdn2d up = 2, pad = (2, 1), 4 x 4 taps; the forward is up2_fir_act_kernel of
// decoder.hip):  out[oy] = sum_iy in[iy] k[oy - 2 iy + 1]  =>  g_in[iy][ix] = sum_{t,u} g_out[2iy - 1 + t][2ix - 1 + u] k[t][u].
// One thread = 4 consecutive ix of one row; grid (blocks, B) so that a workgroup's maximum belongs to one sample.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) up2_fir_bwd_kernel(const float* __restrict__ g_hi, const float* __restrict__ fir,
                                                          float* __restrict__ g_lo, float* __restrict__ out_amax, int C,
                                                          int H, int W) {
  __shared__ float sh[16];
  const int b = blockIdx.y;
  const int wq = W >> 2;                                           // W % 4 == 0
  const int64_t per_sample = (int64_t)C * H * wq;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float k[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i] = fir[i];
  float mx = 0.f;
  if (idx < per_sample) {
    const int xq = (int)(idx % wq);
    const int iy = (int)((idx / wq) % H);
    const int c = (int)(idx / ((int64_t)wq * H));
    const int W2 = 2 * W, H2 = 2 * H;
    const float* src = g_hi + ((int64_t)b * C + c) * H2 * W2;
    const int x0 = 8 * xq;                                          // columns 2 ix - 1 + u of the 4 outputs: x0 - 1 .. x0 + 8
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = 2 * iy - 1 + t;
      if (oy < 0 || oy >= H2) continue;
      const float* row = src + (int64_t)oy * W2 + x0;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(row);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + 4);
      const float lft = x0 > 0 ? row[-1] : 0.f;
      const float rgt = x0 + 8 < W2 ? row[8] : 0.f;
      const float v[10] = {lft, a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], rgt};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[j] = fmaf(v[2 * j + u], k[t * 4 + u], acc[j]);
    }
    cips3d_store_wt16(g_lo + (((int64_t)b * C + c) * H + iy) * W + 4 * xq, acc);
    mx = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
  }
  if (out_amax) {
    const float m = cips3d_workgroup_max(mx, sh, threadIdx.x >> 6, threadIdx.x & 63, 4);
    if (threadIdx.x == 0) cips3d_amax_raise_if(out_amax + b * CIPS3D_AMAX_FLOATS, m, blockIdx.x);
  }
}

// d ToRGB.bias = sum_{b,p} drgb, added to up to 8 destinations (the ToRGBs of one resolution share the gradient image)
struct RgbBiasArgs { const float* drgb; float* dst[8]; int n_dst; int B; int64_t HW; };
__global__ void __launch_bounds__(256) rgb_bias_kernel(RgbBiasArgs a) {
  __shared__ float sh[4];
  const int ch = blockIdx.y;
  float acc = 0.f;
  for (int b = 0; b < a.B; ++b) {
    const float* g = a.drgb + ((int64_t)b * 3 + ch) * a.HW;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.HW; p += (int64_t)gridDim.x * 256) acc += g[p];
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0)
    for (int i = 0; i < a.n_dst; ++i) unsafeAtomicAdd(a.dst[i] + ch, acc);
}

// One level of the skip chain's backward in ONE launch: the FIR transpose of the three-channel gradient image (up2_fir_bwd_kernel's
// arithmetic) together with the ToRGB bias gradients on both sides of it -- d ToRGB.bias = sum_{b,p} drgb at a layer's resolution:
// the sums of the image this kernel WRITES go to the layers of the lower resolution (dst_lo), the sums of the image it READS
// (every element owned by exactly one thread: rows 2 iy, 2 iy + 1, columns 8 xq .. 8 xq + 7) to those of the upper one (dst_hi,
// only the chain's first level has no producer that could have summed it).  grid (ceil(H W / 1024), 3, B).
struct DrgbLevelArgs { const float* g_hi; const float* fir; float* g_lo; int H, W; float* dst_hi[8]; int n_hi; float* dst_lo[8]; int n_lo; };
__global__ void __launch_bounds__(256) drgb_level_kernel(DrgbLevelArgs a) {
  __shared__ float sh[4];
  const int c = blockIdx.y, b = blockIdx.z;
  const int H = a.H, W = a.W, wq = W >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  float k[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) k[i] = a.fir[i];
  float s_hi = 0.f, s_lo = 0.f;
  if (idx < H * wq) {
    const int xq = idx % wq, iy = idx / wq;
    const int W2 = 2 * W, H2 = 2 * H;
    const float* src = a.g_hi + ((int64_t)b * 3 + c) * H2 * W2;
    const int x0 = 8 * xq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = 2 * iy - 1 + t;
      if (oy < 0 || oy >= H2) continue;
      const float* row = src + (int64_t)oy * W2 + x0;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(row);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + 4);
      const float lft = x0 > 0 ? row[-1] : 0.f;
      const float rgt = x0 + 8 < W2 ? row[8] : 0.f;
      const float v[10] = {lft, a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], rgt};
      if (t == 1 || t == 2) s_hi += ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[j] = fmaf(v[2 * j + u], k[t * 4 + u], acc[j]);
    }
    cips3d_store_wt16(a.g_lo + (((int64_t)b * 3 + c) * H + iy) * W + 4 * xq, acc);
    s_lo = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
  if (a.n_hi > 0) {
    s_hi = block_sum_256(s_hi, sh);
    if (threadIdx.x == 0)
      for (int i = 0; i < a.n_hi; ++i) unsafeAtomicAdd(a.dst_hi[i] + c, s_hi);
  }
  if (a.n_lo > 0) {
    s_lo = block_sum_256(s_lo, sh);
    if (threadIdx.x == 0)
      for (int i = 0; i < a.n_lo; ++i) unsafeAtomicAdd(a.dst_lo[i] + c, s_lo);
  }
}

// dst[i] = sum of the slot copies (cips3d_slot_reduce)
__global__ void __launch_bounds__(256) slot_reduce_kernel(const cips3d_slot_job* __restrict__ table, int n_jobs) {
  const int lane = threadIdx.x & 63;
  const int ji = owner_desc(table, n_jobs, (int)blockIdx.x, lane);
  const cips3d_slot_job j = table[ji];
  const int i = ((int)blockIdx.x - j.row_begin) * 256 + threadIdx.x;
  if (i >= j.n) return;
  float acc = 0.f;
  for (int s_ = 0; s_ < j.slots; ++s_) acc += j.src[(int64_t)s_ * j.stride + i];
  j.dst[i] = acc;
}

// d NoiseInjection.weight of every StyledConv = sum over its per-channel partials: block l sums row l of [n][stride]
__global__ void __launch_bounds__(256) row_sums_kernel(const float* __restrict__ parts, int stride, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < stride; i += 256) acc += parts[(int64_t)blockIdx.x * stride + i];
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------
// Modulation backward of every decoder layer in one launch (k = 1; backward.hip has the three per-layer kernels):
//   u = scale W[o] s[b],  d = rsqrt(sum u^2 + 1e-8) (demodulated) or 1,  du = d_wm d - d^3 u <d_wm, u>
//   dW[o][i] = scale sum_b du[b][o][i] s[b][i]         (stored: the row belongs to one wave)
//   ds[b][i] += scale sum_o du[b][o][i] W[o][i]        (atomics: one per workgroup = 32 rows, and element)
// A workgroup = 4 waves x 8 rows of one layer; lane l holds elements l, l + 64, ... (Cin <= 512).
// ------------------------------------------------------------------------------------------------
constexpr int MB_ROWS = 32, MB_E = 8, MB_BMAX = 4;
__global__ void __launch_bounds__(256) modulate_table_bwd_kernel(const cips3d_modbwd_desc* __restrict__ table, int n_desc, int B) {
  __shared__ float s_ds[4][MB_BMAX][64 * MB_E];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int di = owner_desc(table, n_desc, (int)blockIdx.x, lane);
  const cips3d_modbwd_desc d = table[di];
  const int r0 = ((int)blockIdx.x - d.row_begin) * MB_ROWS + wave * (MB_ROWS / 4);
  const int Cin = d.Cin;
  float sv[MB_BMAX][MB_E], dsacc[MB_BMAX][MB_E];
#pragma unroll
  for (int b = 0; b < MB_BMAX; ++b)
#pragma unroll
    for (int j = 0; j < MB_E; ++j) {
      const int e = lane + 64 * j;
      sv[b][j] = (b < B && e < Cin) ? d.s[(int64_t)b * d.s_stride + e] : 0.f;
      dsacc[b][j] = 0.f;
    }
  for (int rr = 0; rr < MB_ROWS / 4; ++rr) {
    const int o = r0 + rr;
    if (o >= d.Cout) break;
    float w[MB_E], dw[MB_E];
#pragma unroll
    for (int j = 0; j < MB_E; ++j) {
      const int e = lane + 64 * j;
      w[j] = e < Cin ? d.W[(int64_t)o * Cin + e] : 0.f;
      dw[j] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < MB_BMAX; ++b) {
      if (b >= B) break;
      float g[MB_E], u[MB_E], ss = 0.f, dot = 0.f;
#pragma unroll
      for (int j = 0; j < MB_E; ++j) {
        const int e = lane + 64 * j;
        g[j] = e < Cin ? d.d_wm[((int64_t)b * d.Cout + o) * Cin + e] : 0.f;
        u[j] = (d.scale * w[j]) * sv[b][j];
        ss = fmaf(u[j], u[j], ss);
        dot = fmaf(g[j], u[j], dot);
      }
      float dd = 1.f, cc = 0.f;
      if (d.demodulate) {
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        dd = rsqrtf(ss + 1e-8f);
        cc = dd * dd * dd * dot;
      }
#pragma unroll
      for (int j = 0; j < MB_E; ++j) {
        const float du = fmaf(g[j], dd, -cc * u[j]);
        dw[j] = fmaf(du, sv[b][j], dw[j]);
        dsacc[b][j] = fmaf(du, w[j], dsacc[b][j]);
      }
    }
    if (d.dW) {
#pragma unroll
      for (int j = 0; j < MB_E; ++j) {
        const int e = lane + 64 * j;
        if (e < Cin) d.dW[(int64_t)o * Cin + e] = dw[j] * d.scale;
      }
    }
  }
  if (!d.ds) return;
#pragma unroll
  for (int b = 0; b < MB_BMAX; ++b)
#pragma unroll
    for (int j = 0; j < MB_E; ++j) s_ds[wave][b][lane + 64 * j] = dsacc[b][j];
  __syncthreads();
  for (int i = threadIdx.x; i < B * Cin; i += 256) {
    const int b = i / Cin, e = i % Cin;
    const float t = (s_ds[0][b][e] + s_ds[1][b][e]) + (s_ds[2][b][e] + s_ds[3][b][e]);
    unsafeAtomicAdd(d.ds + (int64_t)b * d.ds_stride + e, t * d.scale);
  }
}

}  // namespace

extern "C" int cips3d_modulate_table_bwd(const cips3d_modbwd_desc* table_dev, int n_desc, int total_blocks, int B, void* stream) {
  if (!table_dev || n_desc <= 0 || n_desc > 64 || total_blocks <= 0 || B < 0) return CIPS3D_E_BADARG;
  if (B > MB_BMAX) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipLaunchKernelGGL(modulate_table_bwd_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), table_dev, n_desc, B);
  return cips3d_launch_status();
}

extern "C" int cips3d_act_tail_bwd(const float* g_in, const float* y, const float* rgb_w, const float* drgb, const float* noise,
                                   int64_t noise_bstride, float* dpre, float* d_bias, float* d_noise_w, float* d_rgb_w,
                                   float* out_amax, int B, int C, int64_t HW, int slots, int slot_stride, int rgb_slot_stride,
                                   void* stream) {
  if (!y || !dpre || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (slots > 1 && ((slots & (slots - 1)) || slot_stride < C || (d_rgb_w && rgb_slot_stride < B * 3 * C))) return CIPS3D_E_BADARG;
  if ((rgb_w == nullptr) != (drgb == nullptr) || (d_rgb_w && !drgb) || (d_noise_w && !noise)) return CIPS3D_E_BADARG;
  if (HW % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipLaunchKernelGGL(act_tail_kernel, dim3((unsigned)ceil_div<int64_t>(HW, 1024), (unsigned)C, (unsigned)B), dim3(256), 0,
                     as_stream(stream), g_in, y, rgb_w, drgb, noise, noise_bstride, dpre, d_bias, d_noise_w, d_rgb_w, out_amax, C, HW, slots,
                     slot_stride, rgb_slot_stride);
  return cips3d_launch_status();
}

extern "C" int cips3d_slot_reduce(const cips3d_slot_job* table_dev, int n_jobs, int total_blocks, void* stream) {
  if (!table_dev || n_jobs <= 0 || n_jobs > 64 || total_blocks <= 0) return CIPS3D_E_BADARG;
  hipLaunchKernelGGL(slot_reduce_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream), table_dev, n_jobs);
  return cips3d_launch_status();
}

extern "C" int cips3d_up2_fir_bwd(const float* g_hi, const float* fir, float* g_lo, float* out_amax, int B, int C, int H, int W,
                                  void* stream) {
  if (!g_hi || !fir || !g_lo || B < 0 || C <= 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (W % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  const int64_t per_sample = (int64_t)C * H * (W / 4);
  hipLaunchKernelGGL(up2_fir_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(per_sample, 256), (unsigned)B), dim3(256), 0,
                     as_stream(stream), g_hi, fir, g_lo, out_amax, C, H, W);
  return cips3d_launch_status();
}

// ------------------------------------------------------------------------------------------------ the two calls
extern "C" int cips3d_sizeof_grad_plan(void) { return (int)sizeof(cips3d_decoder_grad_plan); }
extern "C" int cips3d_sizeof_grad_io(void) { return (int)sizeof(cips3d_decoder_grad_io); }

#define TRY(x) do { const int rc_ = (x); if (rc_ != 0) return rc_; } while (0)

static bool grad_layer_ok(const cips3d_grad_layer& L) {
  if (L.kind < 0 || L.kind > 3 || L.Cin <= 0 || L.Cout <= 0 || L.H <= 0 || L.W <= 0 || !L.wm || !L.bias) return false;
  if (L.kind <= 1 && (!L.wm_t || !L.y || !L.y_amax || !L.g_amax || !L.d_wm || !L.d_bias || !L.d_nw_part || !L.noise_w)) return false;
  if ((L.kind == 1 || L.kind == 3) && !L.fir) return false;
  if (L.kind == 1 && !L.glo_amax) return false;
  if (L.kind >= 2 && (L.Cout != 3 || !L.d_wm || !L.d_bias)) return false;
  return true;
}

extern "C" int cips3d_decoder_grad_forward(const cips3d_decoder_grad_plan* plan, const cips3d_decoder_grad_io* io, void* stream) {
  if (!plan || !io) return CIPS3D_E_BADARG;
  const cips3d_decoder_grad_plan& P = *plan;
  const cips3d_decoder_grad_io& IO = *io;
  const int B = P.B;
  if (B <= 0 || P.n_layers <= 0 || P.n_layers > CIPS3D_GRAD_MAX_LAYERS || !IO.features || !IO.rgb || !P.style_table ||
      !P.mod_table || !P.amax_base || !P.feat_amax || !P.rgb[0] || !P.rgb[1] || ((IO.style_table == nullptr) != (IO.styles == nullptr)))
    return CIPS3D_E_BADARG;
  for (int li = 0; li < P.n_layers; ++li)
    if (!grad_layer_ok(P.layers[li])) return CIPS3D_E_BADARG;
  // the style heads read the W+ styles from the plan's copy (style_table) or straight from the caller's tensor (io.style_table:
  // the same table with its x pointers on that tensor -- one copy launch less per step)
  const cips3d_linear_desc* style_table = IO.style_table ? IO.style_table : P.style_table;
  // every amax row of the forward starts at zero (the kernels only raise them): cleared by the style heads' launch
  if (P.amax_bytes > 0 && P.amax_bytes / 4 < (1 << 30))
    TRY(cips3d_linear_table_zero(style_table, P.style_n, P.style_rows, B, reinterpret_cast<float*>(P.amax_base), (int)(P.amax_bytes / 4), stream));
  else
    TRY(cips3d_linear_table(style_table, P.style_n, P.style_rows, B, stream));
  TRY(cips3d_modulate_table(P.mod_table, P.mod_n, P.mod_rows, B, 0.f, stream));
  // the features' maximum: a feature is a convex combination of sines (nerf.hip), but a caller may hand in anything
  // (feat_amax lies in the block that was just cleared)
  TRY(cips3d_absmax_raise(IO.features, B, (int64_t)P.layers[0].Cin * P.layers[0].H * P.layers[0].W, P.feat_amax, stream));
  const float* x = IO.features;
  const float* x_amax = P.feat_amax;
  const float* skip = nullptr;
  int rgb_i = 0;
  // ToRGB folding (as in cips3d_generator_forward): a non-up-sampling ToRGB that follows a StyledConv is computed from that
  // conv's registers (partial sums per row block, cips3d_modconv1x1_torgb), and the slots of consecutive such layers are
  // folded by ONE cips3d_torgb_reduce when their sum is first needed -- eight cips3d_torgb launches that re-read the
  // activations become two reductions at CompCars 256^2.  The slots live in g[0] (a gradient buffer: idle in the forward;
  // one layer's slots are < 3/32 of its activation).  CIPS3D_GRAD_TORGB_FOLD=0: one cips3d_torgb per layer (A/B knob).
  static const bool fold_on = [] { const char* e = getenv("CIPS3D_GRAD_TORGB_FOLD"); return !(e && e[0] == '0'); }();
  int64_t g_floats = 0;        // what g[0] certainly holds: the largest StyledConv output
  for (int li = 0; li < P.n_layers; ++li)
    if (P.layers[li].kind == 0) {
      const int64_t n = (int64_t)B * P.layers[li].Cout * P.layers[li].H * P.layers[li].W;
      if (n > g_floats) g_floats = n;
    }
  float* const part = P.g[0];
  int fold_slots = 0, fold_nb = 0, fold_H = 0, fold_W = 0;
  const float* fold_bias[CIPS3D_TORGB_FOLD_MAX];
  auto fold_flush = [&](float* dst) -> int {
    if (fold_slots == 0) return 0;
    const int rc = cips3d_torgb_reduce(part, fold_slots, fold_bias, fold_nb, skip, dst, B, (int64_t)fold_H * fold_W, stream);
    skip = dst;
    fold_slots = fold_nb = 0;
    return rc;
  };
  for (int li = 0; li < P.n_layers; ++li) {
    const cips3d_grad_layer& L = P.layers[li];
    const int64_t hw = (int64_t)L.H * L.W;
    const bool split = (L.flags & 1) != 0;
    if (L.kind == 0 || L.kind == 1) {
      const float* nz = L.noise_index >= 0 ? IO.noise[L.noise_index] : nullptr;
      const int64_t nbs = L.noise_index >= 0 ? IO.noise_bstride[L.noise_index] : 0;
      if (!nz) return CIPS3D_E_BADARG;
      cips3d_range rg{};
      rg.x_amax = x_amax;
      if (L.kind == 1 && fold_slots) {          // resolution changes: the folded sum becomes the skip of the next stage
        TRY(fold_flush(P.rgb[rgb_i]));
        rgb_i ^= 1;
      }
      const cips3d_grad_layer* T = li + 1 < P.n_layers ? &P.layers[li + 1] : nullptr;
      const bool fold = fold_on && L.kind == 0 && T && T->kind == 2 && T->Cin == L.Cout && T->H == L.H && T->W == L.W && hw % 4 == 0 &&
                        fold_nb < CIPS3D_TORGB_FOLD_MAX && (fold_slots == 0 || (fold_H == L.H && fold_W == L.W)) &&
                        ((int64_t)fold_slots + L.Cout / 16) * B * 3 * hw <= g_floats;
      if (fold) {
        int nblk = 0;
        rg.out_amax = L.y_amax;
        TRY(cips3d_modconv1x1_torgb(x, L.wm, L.y, B, L.Cin, L.Cout, hw, 1 | (split ? CIPS3D_GEMM_SPLIT : 0), nz, nbs, L.noise_w, L.bias,
                                    T->wm, part + (int64_t)fold_slots * B * 3 * hw, &nblk, &rg, stream));
        fold_slots += nblk;
        fold_bias[fold_nb++] = T->bias;
        fold_H = L.H; fold_W = L.W;
        ++li;                                     // the ToRGB layer is done (its sum is pending in the slots)
        if (li == P.n_layers - 1) TRY(fold_flush(IO.rgb));
      } else if (L.kind == 0) {
        rg.out_amax = L.y_amax;
        TRY(cips3d_modconv1x1(x, L.wm, L.y, B, L.Cin, L.Cout, hw, 1 | (split ? CIPS3D_GEMM_SPLIT : 0), nz, nbs, L.noise_w, L.bias,
                              &rg, stream));
      } else {
        if (!P.y_lo) return CIPS3D_E_BADARG;
        TRY(cips3d_modconv1x1(x, L.wm, P.y_lo, B, L.Cin, L.Cout, hw, 0 | (split ? CIPS3D_GEMM_SPLIT : 0), nullptr, 0, nullptr,
                              nullptr, &rg, stream));
        TRY(cips3d_up2_fir_act(P.y_lo, L.fir, L.y, B, L.Cout, L.H, L.W, nz, nbs, L.noise_w, L.bias, L.y_amax, stream));
      }
      x = L.y;
      x_amax = L.y_amax;
    } else {
      if (fold_slots) {                           // a ToRGB that could not be folded: settle the pending sum first
        TRY(fold_flush(P.rgb[rgb_i]));
        rgb_i ^= 1;
      }
      const bool last = li == P.n_layers - 1;
      float* out = last ? IO.rgb : P.rgb[rgb_i];
      // (x is the previous StyledConv's output, at this ToRGB's resolution H x W)
      TRY(cips3d_torgb(x, L.wm, L.bias, skip, L.kind == 3 ? 1 : 0, L.fir, out, B, L.Cin, L.H, L.W, stream));
      skip = out;
      rgb_i ^= 1;
    }
  }
  return 0;
}

extern "C" int cips3d_decoder_grad_backward(const cips3d_decoder_grad_plan* plan, const cips3d_decoder_grad_io* io, void* stream) {
  if (!plan || !io) return CIPS3D_E_BADARG;
  const cips3d_decoder_grad_plan& P = *plan;
  const cips3d_decoder_grad_io& IO = *io;
  const int B = P.B;
  if (B <= 0 || P.n_layers <= 0 || P.n_layers > CIPS3D_GRAD_MAX_LAYERS || !IO.features || !IO.d_rgb || !P.zero_base ||
      !P.g[0] || !P.g[1] || !P.modbwd_table || !P.feat_amax)
    return CIPS3D_E_BADARG;
  for (int li = 0; li < P.n_layers; ++li)
    if (!grad_layer_ok(P.layers[li])) return CIPS3D_E_BADARG;
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(P.zero_base, 0, (size_t)P.zero_bytes, st) != hipSuccess) return CIPS3D_E_BADARG;

  // ---- gradient images of the skip chain: one per resolution, from the output down (ToRGB.forward: rgb = conv + up(skip))
  // One launch per resolution change (drgb_level_kernel: FIR transpose + the bias sums of the levels on both sides of it); a
  // decoder without an up-sampling ToRGB, or more than eight ToRGBs at one resolution, takes rgb_bias_kernel for the rest.
  const float* drgb_of[CIPS3D_GRAD_MAX_LAYERS] = {};
  {
    struct Level { const float* g; int H, W; const float* fir; float* dst[8]; int n; };
    Level lv[6] = {};
    int nl = 0, slot = 0;
    lv[0].g = IO.d_rgb;
    RgbBiasArgs ba{};
    ba.B = B;
    for (int li = P.n_layers - 1; li >= 0; --li) {
      const cips3d_grad_layer& L = P.layers[li];
      if (L.kind < 2) continue;
      drgb_of[li] = lv[nl].g;
      lv[nl].H = L.H; lv[nl].W = L.W;
      if (lv[nl].n == 8) {        // (overflow of one level's list: settle these eight now)
        ba.drgb = lv[nl].g; ba.HW = (int64_t)L.H * L.W; ba.n_dst = 8;
        for (int i = 0; i < 8; ++i) ba.dst[i] = lv[nl].dst[i];
        hipLaunchKernelGGL(rgb_bias_kernel, dim3(64, 3), dim3(256), 0, st, ba);
        TRY(cips3d_launch_status());
        lv[nl].n = 0;
      }
      lv[nl].dst[lv[nl].n++] = L.d_bias;
      if (L.kind == 3) {          // the skip came through the FIR up-sampler: its gradient lives at half the resolution
        if (slot >= 4 || !P.drgb_lo[slot] || nl >= 5) return CIPS3D_E_BADARG;
        if ((L.W / 2) % 4) return CIPS3D_E_UNSUPP;
        lv[nl].fir = L.fir;
        lv[++nl].g = P.drgb_lo[slot++];
      }
    }
    for (int t = 0; t < nl; ++t) {
      DrgbLevelArgs a{};
      a.g_hi = lv[t].g; a.fir = lv[t].fir; a.g_lo = const_cast<float*>(lv[t + 1].g);
      a.H = lv[t].H / 2; a.W = lv[t].W / 2;
      if (t == 0) { a.n_hi = lv[0].n; for (int i = 0; i < lv[0].n; ++i) a.dst_hi[i] = lv[0].dst[i]; }
      a.n_lo = lv[t + 1].n;
      for (int i = 0; i < lv[t + 1].n; ++i) a.dst_lo[i] = lv[t + 1].dst[i];
      hipLaunchKernelGGL(drgb_level_kernel, dim3((unsigned)ceil_div(a.H * (a.W / 4), 256), 3, (unsigned)B), dim3(256), 0, st, a);
      TRY(cips3d_launch_status());
    }
    if (nl == 0 && lv[0].n > 0) {
      ba.drgb = lv[0].g; ba.HW = (int64_t)lv[0].H * lv[0].W; ba.n_dst = lv[0].n;
      for (int i = 0; i < lv[0].n; ++i) ba.dst[i] = lv[0].dst[i];
      hipLaunchKernelGGL(rgb_bias_kernel, dim3(64, 3), dim3(256), 0, st, ba);
      TRY(cips3d_launch_status());
    }
  }

  // ---- StyledConvs, last to first
  int conv_idx[CIPS3D_GRAD_MAX_LAYERS], n_conv = 0;
  for (int li = 0; li < P.n_layers; ++li)
    if (P.layers[li].kind < 2) conv_idx[n_conv++] = li;
  if (n_conv == 0) return CIPS3D_E_BADARG;
  auto rgb_after = [&](int li) -> int { return (li + 1 < P.n_layers && P.layers[li + 1].kind >= 2) ? li + 1 : -1; };
  auto noise_of = [&](const cips3d_grad_layer& L, const float*& nz, int64_t& nbs) {
    nz = L.noise_index >= 0 ? IO.noise[L.noise_index] : nullptr;
    nbs = L.noise_index >= 0 ? IO.noise_bstride[L.noise_index] : 0;
  };
  int gi = 0;                                     // P.g[gi] holds the current layer's pre-activation gradient (amax: L.g_amax)
  {
    const int li = conv_idx[n_conv - 1];
    const cips3d_grad_layer& L = P.layers[li];
    const int ti = rgb_after(li);
    if (ti < 0) return CIPS3D_E_BADARG;           // (a decoder ends with a ToRGB)
    const cips3d_grad_layer& T = P.layers[ti];
    const int up = L.kind == 1 ? 2 : 1;
    const float* nz; int64_t nbs;
    noise_of(L, nz, nbs);
    TRY(cips3d_act_tail_bwd(nullptr, L.y, T.wm, drgb_of[ti], nz, nbs, P.g[gi], L.d_bias, L.d_nw_part, T.d_wm, L.g_amax, B,
                            L.Cout, (int64_t)L.H * L.W * up * up, L.slots, L.slot_stride, T.rgb_slot_stride, stream));
  }
  for (int ci = n_conv - 1; ci >= 0; --ci) {
    const int li = conv_idx[ci];
    const cips3d_grad_layer& L = P.layers[li];
    const int64_t hw = (int64_t)L.H * L.W;
    const bool split = (L.flags & 1) != 0;
    const float* g = P.g[gi];
    const float* g_amax = L.g_amax;
    if (L.kind == 1) {
      if (!P.g_lo) return CIPS3D_E_BADARG;
      TRY(cips3d_up2_fir_bwd(g, L.fir, P.g_lo, L.glo_amax, B, L.Cout, L.H, L.W, stream));
      g = P.g_lo;
      g_amax = L.glo_amax;
    }
    const float* x = ci > 0 ? P.layers[conv_idx[ci - 1]].y : IO.features;
    const float* x_amax = ci > 0 ? P.layers[conv_idx[ci - 1]].y_amax : P.feat_amax;
    // weight gradient (accumulated into the zeroed d_wm)
    if (split && L.Cout % 32 == 0 && L.Cin % 32 == 0 && hw % 32 == 0)
      TRY(cips3d_gemm_wgrad_split(g, x, L.d_wm, B, L.Cout, L.Cin, hw, g_amax, x_amax, 1, stream));
    else
      TRY(cips3d_gemm_wgrad(g, x, L.d_wm, B, L.Cout, L.Cin, hw, stream));
    // data gradient
    cips3d_range rg{};
    rg.x_amax = g_amax;
    if (ci == 0) {
      if (IO.d_features)
        TRY(cips3d_modconv1x1(g, L.wm_t, IO.d_features, B, L.Cout, L.Cin, hw, split ? CIPS3D_GEMM_SPLIT : 0, nullptr, 0, nullptr,
                              nullptr, &rg, stream));
    } else {
      const int pi = conv_idx[ci - 1];
      const cips3d_grad_layer& Lp = P.layers[pi];
      const int ti = rgb_after(pi);
      cips3d_actbwd ab{};
      ab.y = Lp.y;
      if (ti >= 0) { ab.rgb_w = P.layers[ti].wm; ab.drgb = drgb_of[ti]; ab.d_rgb_w = P.layers[ti].d_wm; }
      ab.d_bias = Lp.d_bias;
      ab.d_noise_w = Lp.d_nw_part;
      ab.slots = Lp.slots; ab.slot_stride = Lp.slot_stride;
      ab.rgb_slot_stride = ti >= 0 ? P.layers[ti].rgb_slot_stride : 0;
      const float* nz; int64_t nbs;
      noise_of(Lp, nz, nbs);
      const int go = gi ^ 1;
      rg.out_amax = Lp.g_amax;              // (every gradient has an amax row of its own, in the block this call zeroed up front)
      TRY(cips3d_modconv1x1_actbwd(g, L.wm_t, P.g[go], B, L.Cout, L.Cin, hw, split ? CIPS3D_GEMM_SPLIT : 0, &ab, nz, nbs, &rg, stream));
      gi = go;
    }
  }

  // ---- parameters: modulation backward of every layer, the style heads, the scalar noise weights
  if (P.slot_table) TRY(cips3d_slot_reduce(P.slot_table, P.slot_n, P.slot_blocks, stream));
  TRY(cips3d_modulate_table_bwd(P.modbwd_table, P.modbwd_n, P.modbwd_blocks, B, stream));
  if (P.d_styles || P.d_style_W)
    TRY(cips3d_linear_table_bwd(IO.style_table ? IO.style_table : P.style_table, P.style_n, P.style_rows, P.style_dim, B, P.s_all, P.ds_all,
                                IO.style_table ? IO.styles : P.styles, P.d_styles, P.style_w_offsets, P.d_style_W, P.d_style_b, stream));
  if (P.d_noise_w && P.nw_parts && P.nw_stride > 0) {
    hipLaunchKernelGGL(row_sums_kernel, dim3((unsigned)n_conv), dim3(256), 0, st, P.nw_parts, P.nw_stride, P.d_noise_w);
    TRY(cips3d_launch_status());
  }
  return 0;
}
