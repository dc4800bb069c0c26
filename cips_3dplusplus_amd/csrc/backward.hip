// Backward building blocks of the generator path (flip inversion, reference
// models/projector_v10.py:211-277,915-1216 drives `loss.backward()` through Generator.forward).
//
// The reference gets every one of these from PyTorch autograd over cuBLAS/cuDNN; here each forward op has an explicit
// HIP backward, chained on the host by torch.autograd.Function objects (cips_3dplusplus_amd/autograd.py):
//   cips3d_linear_bwd            EqualLinear / MappingLinear / LinearLayer   (model_v3.py:40-65,183-210; volume_renderer.py:15-35)
//   cips3d_modulate_bwd          weight modulation + demodulation            (model_v3.py:267-278)
//   cips3d_pack_weights          plain [B,M,K] -> MFMA A-fragment order, optionally of the transpose (data-gradient GEMM)
//   cips3d_gemm_wgrad            dW[b] = dY[b] X[b]^T over the pixels, fp32 MFMA, split over pixel chunks
//   cips3d_noise_bias_act_bwd    NoiseInjection + FusedLeakyReLU             (model_v3.py:327-341; fused_act.py:20-84)
//   cips3d_torgb_bwd             ToRGB (C -> 3 modulated conv + bias)        (model_v3.py:469-482)
// The data gradient of the 1x1 convolution is cips3d_modconv1x1 itself on the packed transpose; the gradient of the FIR
// up-sampler is cips3d_upfirdn2d with swapped factors and flipped taps (upfirdn2d.py:20-143).
//
// Rooflines: the element-wise kernels and the reductions are HBM-bound (each activation read once, each gradient
// written once); cips3d_gemm_wgrad is MFMA work (2*M*K flop per pixel) fed straight from global memory -- both
// operands are pixel-contiguous, so a float4 per lane IS four k-slices of v_mfma_f32_16x16x4_f32 and no LDS
// transpose is needed.
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

// ------------------------------------------------------------------------------------------------ linear
struct LinBwd {
  const float* x; int64_t x_stride; const float* W; const float* out; int64_t out_stride; const float* dout;
  int64_t dout_stride; int B, in_dim, out_dim; float w_scale, b_scale; int lrelu; float act_gain, out_scale;
  float* dx; int64_t dx_stride; float* dW; float* dbias;
};

// d(pre-activation) of y = (lrelu?(pre) * gain) * out_scale + out_shift, pre = x W^T w_scale + b b_scale
__device__ __forceinline__ float lin_dpre(const LinBwd& a, int b, int o) {
  float g = a.dout[(int64_t)b * a.dout_stride + o] * a.out_scale;
  if (a.lrelu) g *= a.act_gain * (a.out[(int64_t)b * a.out_stride + o] > 0.f ? 1.f : 0.2f);
  return g;
}

__global__ void __launch_bounds__(256) linear_bwd_dw_kernel(LinBwd a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.out_dim * a.in_dim) return;
  const int o = (int)(e / a.in_dim), i = (int)(e % a.in_dim);
  float acc = 0.f;
  for (int b = 0; b < a.B; ++b) acc = fmaf(lin_dpre(a, b, o), a.x[(int64_t)b * a.x_stride + i], acc);
  a.dW[e] = acc * a.w_scale;
}

// grid (ceil(in/64), B), 1024 threads = 64 input columns x 16 groups of output rows (W rows read with 256-byte segments)
__global__ void __launch_bounds__(1024) linear_bwd_dx_kernel(LinBwd a) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (i < a.in_dim)
    for (int o = grp; o < a.out_dim; o += 16) acc = fmaf(lin_dpre(a, b, o), a.W[(int64_t)o * a.in_dim + i], acc);
  part[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && i < a.in_dim) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][lane];
    a.dx[(int64_t)b * a.dx_stride + i] = t * a.w_scale;
  }
}

__global__ void __launch_bounds__(256) linear_bwd_db_kernel(LinBwd a) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= a.out_dim) return;
  float acc = 0.f;
  for (int b = 0; b < a.B; ++b) acc += lin_dpre(a, b, o);
  a.dbias[o] = acc * a.b_scale;
}

// ------------------------------------------------------------------------------------------------ modulation
// wm[b][o][e] = u * d,  u = scale W[o][e] s[b][e/ksq],  d = rsqrt(sum_e u^2 + 1e-8) (demodulate) or 1.
// One wave per (b, o): du = dwm d - d^3 u <dwm, u>, written over dwm.
__global__ void __launch_bounds__(256) modulate_bwd_du_kernel(float* __restrict__ dwm, const float* __restrict__ W,
                                                              const float* __restrict__ s, int64_t s_stride, int B,
                                                              int Cout, int len, int ksq, float scale, int demod) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Cout || !demod) return;
  const int b = (int)(row / Cout), o = (int)(row % Cout);
  const float* w = W + (int64_t)o * len;
  const float* sb = s + (int64_t)b * s_stride;
  float* g = dwm + row * len;
  float ss = 0.f, dot = 0.f;
  for (int e = lane; e < len; e += 64) {
    const float u = (scale * w[e]) * sb[e / ksq];
    ss = fmaf(u, u, ss);
    dot = fmaf(g[e], u, dot);
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float d = rsqrtf(ss + 1e-8f);
  const float c = d * d * d * dot;
  for (int e = lane; e < len; e += 64) {
    const float u = (scale * w[e]) * sb[e / ksq];
    g[e] = fmaf(g[e], d, -c * u);
  }
}

// dW[o][e] = scale sum_b du[b][o][e] s[b][e/ksq]
__global__ void __launch_bounds__(256) modulate_bwd_dw_kernel(const float* __restrict__ du, const float* __restrict__ s,
                                                              int64_t s_stride, int B, int Cout, int len, int ksq,
                                                              float scale, float* __restrict__ dW) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)Cout * len) return;
  const int i = (int)(e % len) / ksq;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc = fmaf(du[(int64_t)b * Cout * len + e], s[(int64_t)b * s_stride + i], acc);
  dW[e] = acc * scale;
}

// ds[b][i] = scale sum_o sum_t du[b][o][i*ksq+t] W[o][i*ksq+t]
// grid (ceil(Cin/64), B), 1024 threads = 64 input channels x 16 groups of output rows
__global__ void __launch_bounds__(1024) modulate_bwd_ds_kernel(const float* __restrict__ du, const float* __restrict__ W,
                                                               int B, int Cout, int Cin, int ksq, float scale,
                                                               float* __restrict__ ds, int64_t ds_stride) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int b = blockIdx.y, i = blockIdx.x * 64 + lane;
  const int len = Cin * ksq;
  float acc = 0.f;
  if (i < Cin)
    for (int o = grp; o < Cout; o += 16)
      for (int t = 0; t < ksq; ++t)
        acc = fmaf(du[((int64_t)b * Cout + o) * len + i * ksq + t], W[(int64_t)o * len + i * ksq + t], acc);
  part[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && i < Cin) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][lane];
    ds[(int64_t)b * ds_stride + i] = t * scale;
  }
}

// ------------------------------------------------------------------------------------------------ packing
// out = A-fragment order (see decoder.hip: modulate_row) of wm[b] (M x K) or of its transpose (K x M).
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ wm, float* __restrict__ out, int B, int M,
                                                   int K, int transpose) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * M * K) return;
  const int b = (int)(e / ((int64_t)M * K));
  const int r = (int)((e / K) % M), c = (int)(e % K);       // element wm[b][r][c]
  const bool tr = transpose & 1;
  const int o = tr ? c : r, i = tr ? r : c;   // (row, column) of the packed matrix
  const int OM = tr ? K : M, OK = tr ? M : K;
  if (transpose & 2) {
    // split-fp16 fragments for CIPS3D_GEMM_SPLIT (decoder.hip): 2^8 w = fp16 hi + fp16 lo, natural k order
    const int ot = o >> 4, kb = i >> 5, j = i & 7, q = (i >> 3) & 3;
    const float sv = wm[e] * 256.f;
    _Float16 hi, lo;
    cips3d_split16(sv, hi, lo);
    _Float16* blk = reinterpret_cast<_Float16*>(out) + (((int64_t)b * (OM >> 4) + ot) * (OK >> 5) + kb) * 1024;
    blk[((q << 4) | (o & 15)) * 8 + j] = hi;
    blk[512 + ((q << 4) | (o & 15)) * 8 + j] = lo;
    return;
  }
  const int ot = o >> 4, kq = i >> 4, j = (i >> 2) & 3, q = i & 3;
  out[(((int64_t)b * (OM >> 4) + ot) * (OK >> 4) + kq) * 256 + ((q << 4) | (o & 15)) * 4 + j] = wm[e];
}

// ------------------------------------------------------------------------------------------------ weight-gradient GEMM
// dwm[b][m][k] = sum_p dy[b][m][p] x[b][k][p].  A workgroup = 4 waves = 2 x 2 wave tiles of 32 x 32 outputs; every
// wave walks the same pixel chunk.  Lane (r = lane & 15, q = lane >> 4) loads float4 = pixels p0 + 4q .. 4q+3 of row r:
// component j of all lanes forms one k-slice {p0 + 4q' + j} -- the same permutation of the pixels on both operands,
// which a contraction does not see.  Partial sums of the chunks meet in dwm through fp32 atomics.
__global__ void __launch_bounds__(256) gemm_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ dwm, int M, int K, int64_t P,
                                                         int64_t chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kblocks = ceil_div(K, 64);
  const int mb = blockIdx.x / kblocks, kb = blockIdx.x % kblocks;
  const int m0 = mb * 64 + (wave >> 1) * 32, k0 = kb * 64 + (wave & 1) * 32;
  if (m0 >= M || k0 >= K) return;
  const int b = blockIdx.z;
  const int64_t p_begin = (int64_t)blockIdx.y * chunk;
  const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;
  const float* a0 = dy + ((int64_t)b * M + m0 + r) * P + 4 * q;
  const float* a1 = a0 + 16 * P;
  const float* b0 = x + ((int64_t)b * K + k0 + r) * P + 4 * q;
  const float* b1 = b0 + 16 * P;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t p = p_begin; p < p_end; p += 16) {
    const bool ok = p + 4 * q < p_end;          // P % 4 == 0, chunk % 16 == 0: a float4 is all-in or all-out
    const f32x4 va0 = ok ? *reinterpret_cast<const f32x4*>(a0 + p) : zero;
    const f32x4 va1 = ok ? *reinterpret_cast<const f32x4*>(a1 + p) : zero;
    const f32x4 vb0 = ok ? *reinterpret_cast<const f32x4*>(b0 + p) : zero;
    const f32x4 vb1 = ok ? *reinterpret_cast<const f32x4*>(b1 + p) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[j], vb0[j], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[j], vb1[j], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(va1[j], vb0[j], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(va1[j], vb1[j], acc[1][1], 0, 0, 0);
    }
  }
  // D layout: register t of lane (r, q) = C[4q + t][r]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        unsafeAtomicAdd(dwm + ((int64_t)b * M + m0 + 16 * i + 4 * q + t) * K + k0 + 16 * j + r, acc[i][j][t]);
}

// The same contraction on split-fp16 products (v_mfma_f32_16x16x32_f16, three per fp32 product; decoder.hip explains the
// arithmetic).  Gradients have no natural scale, so both operands are split as v * 2^-e with the power of two that puts the
// tensor's MEASURED maximum (amax slots of the producing kernel, cips3d_range) into [2^14, 2^15); the accumulators go back
// through 2^(e_dy + e_x), exactly.
// Dataflow (the direct-from-global form of gemm_wgrad_kernel moves every operand row once per WAVE: 536 MB per 512 x 512 x
// 4096-pixel layer through the CU's L2 port, 66 us whatever the instruction): a workgroup of 2 x 2 waves owns a
// (32 TM) x (32 TN) output tile and walks its pixel chunk in steps of 32.  Per step every thread fetches (TM + TN) float4 of
// the NEXT step's rows into registers, the waves run the 3 TM TN MFMAs of the CURRENT step on fragments read from LDS, then
// the fetched values are split ONCE (two instructions per value) and written as fp16 hi / lo planes into the other LDS
// buffer: each value crosses the L2 port once per workgroup and is converted once.  LDS rows are 32 pixels = 64 bytes of
// fp16, padded to 80: the 16 rows a fragment read touches then fall into 16 different bank groups.
template <int TM, int TN, int WGM, int WGN>
__global__ void __launch_bounds__(64 * WGM * WGN) wgrad_split_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ dwm, int M, int K, int64_t P, int64_t chunk,
                                                          const float* __restrict__ dy_amax, const float* __restrict__ x_amax,
                                                          int blocks, int n_chunks, int xcd_groups) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 16 * TM * WGM, BN = 16 * TN * WGN, ROWS = BM + BN;
  constexpr int NV = ROWS * 8 / NT;             // float4 per thread and step
  static_assert(ROWS * 8 % NT == 0, "whole float4s per thread");
  constexpr int RS = 40;                        // LDS row stride in halfs (80 bytes)
  __shared__ __attribute__((aligned(16))) _Float16 lds[2][2][ROWS * RS];      // [buffer][hi | lo][row][pixel]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  // Workgroup -> (output tile, pixel chunk, sample).  The tiles of one (chunk, sample) group read the same operand rows:
  // with the dispatcher's round-robin of consecutive workgroups over the 8 XCDs, group g = id % 8 (+ 8 per pass) puts a whole
  // group on ONE XCD, whose L2 then serves every re-read of the group's rows (the operands together do not fit any L2: left
  // to the linear order, the re-reads come from the Infinity Cache at half the rate -- measured 33 vs 2x us).  Placement
  // only: any mapping computes the same sums.
  int blk, grp;
  if (xcd_groups) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    grp = xcd + 8 * (i / blocks);
    blk = i % blocks;
  } else {
    blk = blockIdx.x % blocks;
    grp = blockIdx.x / blocks;
  }
  const int kblocks = ceil_div(K, BN);
  const int mb = blk / kblocks, kb = blk % kblocks;
  const int m0 = mb * BM, k0 = kb * BN;
  const int wm = wave / WGN, wn = wave % WGN;
  const int b = grp / n_chunks;
  int ea = 0, eb = 0;
  if (dy_amax) ea = cips3d_split_exp(cips3d_amax_load(dy_amax + b * CIPS3D_AMAX_FLOATS));
  if (x_amax) eb = cips3d_split_exp(cips3d_amax_load(x_amax + b * CIPS3D_AMAX_FLOATS));
  const float ka = cips3d_uniform(cips3d_pow2(-ea)), kbs = cips3d_uniform(cips3d_pow2(-eb));
  const int64_t p_begin = (int64_t)(grp % n_chunks) * chunk;
  const int64_t p_end = p_begin + chunk < P ? p_begin + chunk : P;            // P % 32 == 0, chunk % 32 == 0
  // staging: float4 f = tid + NT v covers row f / 8 (A rows first, then B rows), pixels 4 (f % 8) .. +3; rows past the
  // matrix are clamped (loaded twice, never stored)
  const float* src[NV];
  float ksc[NV];
  int dst[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int f = tid + NT * v, row = f >> 3, c4 = f & 7;
    const bool isa = row < BM;
    const int gr = isa ? min(m0 + row, M - 1) : min(k0 + row - BM, K - 1);
    src[v] = (isa ? dy + ((int64_t)b * M + gr) * P : x + ((int64_t)b * K + gr) * P) + 4 * c4;
    ksc[v] = isa ? ka : kbs;
    dst[v] = row * RS + 4 * c4;
  }
  // register ring of fetched steps: the fetch of step s + D is issued when step s starts.  Measured at 512 x 512 x 4096 x 2
  // (timing ablations): fetch + fragment reads + barriers alone 15.9 us (201 MB at 12.6 TB/s), + conversion 5.7, + MFMA 3.5,
  // + atomics 5: the phases of a step do not overlap (every wave of the CU is in the same phase between two barriers; starting
  // the second wave of each SIMD with the conversion instead changed nothing, more chunks cost more in atomics than they hide)
  constexpr int D = 4;       // (2, 6, 8 measured on config 5, same box x3: inside its 10 % run-to-run spread)
  f32x4 raw[D][NV];
  // The loop below has NO branch around its loads and LDS stores: with `if (p < p_end)` around them the compiler could not count
  // the loads in flight across the joins and put s_waitcnt vmcnt(0) in front of every use AND every issue -- each step then
  // waited for the fetch it had just requested (a full memory round trip per 32 pixels, ~1 us: the ring was no ring).  A fetch
  // past the chunk re-reads the chunk's last step instead (valid memory, never used), a step past the chunk is stashed as zeros.
  const int64_t p_last = p_end - 32;
  auto fetch = [&](int slot, int64_t p) {
    const int64_t pc = p < p_end ? p : p_last;
#pragma unroll
    for (int v = 0; v < NV; ++v) raw[slot][v] = *reinterpret_cast<const f32x4*>(src[v] + pc);
  };
  auto stash = [&](int slot, int buf, int64_t p) {
    const float live = p < p_end ? 1.f : 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float kv = ksc[v] * live;
      unsigned h0, l0, h1, l1;
      cips3d_split_pair(raw[slot][v][0] * kv, raw[slot][v][1] * kv, h0, l0);
      cips3d_split_pair(raw[slot][v][2] * kv, raw[slot][v][3] * kv, h1, l1);
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      *reinterpret_cast<u32x2_t*>(&lds[buf][0][dst[v]]) = u32x2_t{h0, h1};
      *reinterpret_cast<u32x2_t*>(&lds[buf][1][dst[v]]) = u32x2_t{l0, l1};
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(d, p_begin + 32 * d);
  stash(0, 0, p_begin);
  __syncthreads();
  const int arow = (wm * 16 * TM + r) * RS + 8 * q, brow = (BM + wn * 16 * TN + r) * RS + 8 * q;
  for (int64_t p0 = p_begin; p0 < p_end; p0 += 32 * D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int64_t p = p0 + 32 * d;                // (steps past the chunk multiply zeros)
      const int buf = d & 1;                        // D is even: the buffer parity of a step is the parity of its slot
      cips3d_h8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const cips3d_h8*>(&lds[buf][0][arow + 16 * i * RS]);
        al[i] = *reinterpret_cast<const cips3d_h8*>(&lds[buf][1][arow + 16 * i * RS]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const cips3d_h8*>(&lds[buf][0][brow + 16 * j * RS]);
        bl[j] = *reinterpret_cast<const cips3d_h8*>(&lds[buf][1][brow + 16 * j * RS]);
      }
      fetch(d, p + 32 * D);                         // slot d was stashed during the previous step
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
      stash((d + 1) % D, buf ^ 1, p + 32);
      __syncthreads();
    }
  }
  // D layout: register t of lane (r, q) = C[4q + t][r]; two exact multiplications (2^(ea + eb) alone may not be a float)
  const float oa = cips3d_pow2(ea), ob = cips3d_pow2(eb);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = m0 + wm * 16 * TM + 16 * i + 4 * q + t, k = k0 + wn * 16 * TN + 16 * j + r;
        if (m < M && k < K) unsafeAtomicAdd(dwm + ((int64_t)b * M + m) * K + k, (acc[i][j][t] * oa) * ob);
      }
}

// ------------------------------------------------------------------------------------------------ noise + bias + leaky-ReLU
// y = lrelu(x + nw noise + bias_c) sqrt2  =>  dx = dy sqrt2 (y > 0 ? 1 : 0.2);  dbias_c = sum_{b,p} dx.
// grid (pixel blocks, C, B): one block reduces its 1024 pixels of one channel.
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, float* __restrict__ dbias,
                                                      const float* __restrict__ noise, int64_t noise_bstride,
                                                      float* __restrict__ dnw, int C, int64_t HW) {
  __shared__ float sh[4];
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t base = ((int64_t)b * C + c) * HW;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float sum = 0.f, nsum = 0.f;
  if (p0 < HW) {    // HW % 4 == 0
    const float4 g = *reinterpret_cast<const float4*>(dy + base + p0);
    const float4 v = *reinterpret_cast<const float4*>(y + base + p0);
    const float S = 1.41421356237309515f;
    float4 o;
    o.x = g.x * (v.x > 0.f ? S : 0.2f * S);
    o.y = g.y * (v.y > 0.f ? S : 0.2f * S);
    o.z = g.z * (v.z > 0.f ? S : 0.2f * S);
    o.w = g.w * (v.w > 0.f ? S : 0.2f * S);
    *reinterpret_cast<float4*>(dx + base + p0) = o;
    sum = (o.x + o.y) + (o.z + o.w);
    if (dnw) {
      const float4 nz = *reinterpret_cast<const float4*>(noise + (int64_t)b * noise_bstride + p0);
      nsum = fmaf(o.w, nz.w, fmaf(o.z, nz.z, fmaf(o.y, nz.y, o.x * nz.x)));
    }
  }
  sum = block_sum_256(sum, sh);
  if (threadIdx.x == 0 && dbias) unsafeAtomicAdd(dbias + c, sum);
  if (dnw) {       // d noise_weight = sum_{b,c,p} dx * noise: per-channel partials (one hot address would serialise)
    nsum = block_sum_256(nsum, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(dnw + c, nsum);
  }
}

__global__ void __launch_bounds__(256) sum_to_scalar_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = acc;
}

// dnoise[bn][p] += nw sum_{c in chunk} dx[b][c][p]   (bn = b for per-sample noise, 0 for a shared buffer; dnoise zeroed by
// the host call).  grid (ceil(HW/256), ceil(C/32), B)
__global__ void __launch_bounds__(256) noise_bwd_kernel(const float* __restrict__ dx, int64_t noise_bstride,
                                                        const float* __restrict__ noise_w, float* __restrict__ dnoise,
                                                        int C, int64_t HW) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int b = blockIdx.z, c0 = blockIdx.y * 32, c1 = min(C, c0 + 32);
  float acc = 0.f;
  for (int c = c0; c < c1; ++c) acc += dx[((int64_t)b * C + c) * HW + p];
  unsafeAtomicAdd(dnoise + (int64_t)b * noise_bstride + p, acc * noise_w[0]);
}

// ------------------------------------------------------------------------------------------------ ToRGB
// rgb[r][p] = sum_c wm[b][r][c] x[c][p] + bias_r (+ skip)
//   dx[c][p] = sum_r wm[r][c] drgb[r][p];  dwm[b][r][c] = sum_p drgb[r][p] x[c][p];  dbias_r = sum_{b,p} drgb[r][p]
__global__ void __launch_bounds__(256) torgb_bwd_kernel(const float* __restrict__ drgb, const float* __restrict__ x,
                                                        const float* __restrict__ wm, float* __restrict__ dx,
                                                        float* __restrict__ dwm, float* __restrict__ dbias, int C,
                                                        int64_t HW) {
  __shared__ float sh[4];
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float* g = drgb + (int64_t)b * 3 * HW;
  const float w0 = wm[((int64_t)b * 3 + 0) * C + c], w1 = wm[((int64_t)b * 3 + 1) * C + c],
              w2 = wm[((int64_t)b * 3 + 2) * C + c];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
  if (p0 < HW) {
    const float4 g0 = *reinterpret_cast<const float4*>(g + p0);
    const float4 g1 = *reinterpret_cast<const float4*>(g + HW + p0);
    const float4 g2 = *reinterpret_cast<const float4*>(g + 2 * HW + p0);
    const int64_t xi = ((int64_t)b * C + c) * HW + p0;
    const float4 xv = *reinterpret_cast<const float4*>(x + xi);
    float4 o;
    o.x = fmaf(w2, g2.x, fmaf(w1, g1.x, w0 * g0.x));
    o.y = fmaf(w2, g2.y, fmaf(w1, g1.y, w0 * g0.y));
    o.z = fmaf(w2, g2.z, fmaf(w1, g1.z, w0 * g0.z));
    o.w = fmaf(w2, g2.w, fmaf(w1, g1.w, w0 * g0.w));
    *reinterpret_cast<float4*>(dx + xi) = o;
    s0 = fmaf(g0.w, xv.w, fmaf(g0.z, xv.z, fmaf(g0.y, xv.y, g0.x * xv.x)));
    s1 = fmaf(g1.w, xv.w, fmaf(g1.z, xv.z, fmaf(g1.y, xv.y, g1.x * xv.x)));
    s2 = fmaf(g2.w, xv.w, fmaf(g2.z, xv.z, fmaf(g2.y, xv.y, g2.x * xv.x)));
    if (c == 0) {
      t0 = (g0.x + g0.y) + (g0.z + g0.w);
      t1 = (g1.x + g1.y) + (g1.z + g1.w);
      t2 = (g2.x + g2.y) + (g2.z + g2.w);
    }
  }
  s0 = block_sum_256(s0, sh); s1 = block_sum_256(s1, sh); s2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(dwm + ((int64_t)b * 3 + 0) * C + c, s0);
    unsafeAtomicAdd(dwm + ((int64_t)b * 3 + 1) * C + c, s1);
    unsafeAtomicAdd(dwm + ((int64_t)b * 3 + 2) * C + c, s2);
  }
  if (c == 0 && dbias) {
    t0 = block_sum_256(t0, sh); t1 = block_sum_256(t1, sh); t2 = block_sum_256(t2, sh);
    if (threadIdx.x == 0) { unsafeAtomicAdd(dbias + 0, t0); unsafeAtomicAdd(dbias + 1, t1); unsafeAtomicAdd(dbias + 2, t2); }
  }
}


// ------------------------------------------------------------------------------------------------
// backward of a table of linear heads (cips3d_linear_table_bwd)
// rows kernel: one wave per output row of the table: dW[row][:] = w_scale out_scale sum_b dy[b][row] x[b][:], db[row]
// ------------------------------------------------------------------------------------------------
struct TableBwd {
  const cips3d_linear_desc* table; int n_desc, total_rows, in_dim, B;
  const float* out_base; const float* dy_base; const float* x_base; float* dx_base;
  const int64_t* w_off; float* dW; float* db;
};

__global__ void __launch_bounds__(256) table_bwd_rows_kernel(TableBwd a) {
  const int lane = threadIdx.x & 63;
  const int grow = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (grow >= a.total_rows) return;
  const int di = owner_desc(a.table, a.n_desc, grow, lane);
  const cips3d_linear_desc d = a.table[di];
  const int row = grow - d.row_begin;
  const float* dy = a.dy_base + (d.out - a.out_base) + row;
  const float* x = a.x_base + (d.x - a.x_base);
  float bsum = 0.f;
  for (int b = 0; b < a.B; ++b) bsum += dy[(int64_t)b * d.out_stride];
  if (a.db && lane == 0) a.db[grow] = bsum * (d.b_scale * d.out_scale);
  if (!a.dW) return;
  float* dw = a.dW + a.w_off[di] + (int64_t)row * d.in_dim;
  const float sc = d.w_scale * d.out_scale;
  for (int k = lane * 4; k < d.in_dim; k += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < a.B; ++b) {
      const float g = dy[(int64_t)b * d.out_stride];
      const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)b * d.x_stride + k);
      acc.x = fmaf(g, xv.x, acc.x); acc.y = fmaf(g, xv.y, acc.y); acc.z = fmaf(g, xv.z, acc.z); acc.w = fmaf(g, xv.w, acc.w);
    }
    *reinterpret_cast<float4*>(dw + k) = make_float4(acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc);
  }
}

// columns kernel: block = (head, 64-column block, row group): dx[b][col] += w_scale out_scale sum_{rows of the group} dy[b][row] W[row][col]
constexpr int TB_RG = 4;     // row groups per (head, column block)
__global__ void __launch_bounds__(256) table_bwd_cols_kernel(TableBwd a) {
  __shared__ float s_acc[4][64];
  const int cblocks = (a.in_dim + 63) / 64;
  const int di = blockIdx.x / (cblocks * TB_RG);
  const int rem = blockIdx.x % (cblocks * TB_RG);
  const int cb = rem / TB_RG, rg = rem % TB_RG;
  const cips3d_linear_desc d = a.table[di];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = cb * 64 + tx;
  const bool col_ok = col < d.in_dim;
  const float* dy = a.dy_base + (d.out - a.out_base);
  float* dx = a.dx_base + (d.x - a.x_base);
  const float sc = d.w_scale * d.out_scale;
  for (int b = 0; b < a.B; ++b) {
    float acc = 0.f;
    if (col_ok)
      for (int r = rg * 4 + ty; r < d.out_dim; r += 4 * TB_RG)
        acc = fmaf(dy[(int64_t)b * d.out_stride + r], d.W[(int64_t)r * d.in_dim + col], acc);
    s_acc[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && col_ok) atomicAdd(dx + (int64_t)b * d.x_stride + col, ((s_acc[0][tx] + s_acc[1][tx]) + (s_acc[2][tx] + s_acc[3][tx])) * sc);
    __syncthreads();
  }
}

}  // namespace

extern "C" int cips3d_linear_bwd(const float* x, int64_t x_stride, const float* W, const float* out, int64_t out_stride,
                                 const float* dout, int64_t dout_stride, int B, int in_dim, int out_dim, float w_scale,
                                 float b_scale, int lrelu, float act_gain, float out_scale, float* dx, int64_t dx_stride,
                                 float* dW, float* dbias, void* stream) {
  if (!x || !W || !dout || B < 0 || in_dim <= 0 || out_dim <= 0) return CIPS3D_E_BADARG;
  if (lrelu && !out) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  LinBwd a{x, x_stride, W, out, out_stride, dout, dout_stride, B, in_dim, out_dim, w_scale, b_scale, lrelu, act_gain,
           out_scale, dx, dx_stride, dW, dbias};
  hipStream_t st = as_stream(stream);
  if (dW)
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)out_dim * in_dim, 256)), dim3(256), 0,
                       st, a);
  if (dx)
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((unsigned)ceil_div(in_dim, 64), (unsigned)B), dim3(1024), 0, st, a);
  if (dbias) hipLaunchKernelGGL(linear_bwd_db_kernel, dim3((unsigned)ceil_div(out_dim, 256)), dim3(256), 0, st, a);
  return cips3d_launch_status();
}

extern "C" int cips3d_modulate_bwd(float* dwm, const float* W, const float* s, int64_t s_stride, int B, int Cout, int Cin,
                                   int ksq, float scale, int demodulate, float* dW, float* ds, int64_t ds_stride,
                                   void* stream) {
  if (!dwm || !W || !s || B < 0 || Cout <= 0 || Cin <= 0 || ksq <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipStream_t st = as_stream(stream);
  const int len = Cin * ksq;
  if (demodulate)
    hipLaunchKernelGGL(modulate_bwd_du_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)B * Cout, 4)), dim3(256), 0, st, dwm,
                       W, s, s_stride, B, Cout, len, ksq, scale, demodulate);
  if (dW)
    hipLaunchKernelGGL(modulate_bwd_dw_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)Cout * len, 256)), dim3(256), 0, st,
                       dwm, s, s_stride, B, Cout, len, ksq, scale, dW);
  if (ds)
    hipLaunchKernelGGL(modulate_bwd_ds_kernel, dim3((unsigned)ceil_div(Cin, 64), (unsigned)B), dim3(1024), 0, st, dwm, W, B, Cout,
                       Cin, ksq, scale, ds, ds_stride);
  return cips3d_launch_status();
}

extern "C" int cips3d_pack_weights(const float* wm, float* packed, int B, int M, int K, int transpose, void* stream) {
  if (!wm || !packed || B < 0 || M <= 0 || K <= 0) return CIPS3D_E_BADARG;
  if (M % 16 || K % 16) return CIPS3D_E_UNSUPP;
  if ((transpose & 2) && (((transpose & 1) ? M : K) % 32)) return CIPS3D_E_UNSUPP;     // split fragments: 32-wide k blocks
  if (B == 0) return 0;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)B * M * K, 256)), dim3(256), 0, as_stream(stream),
                     wm, packed, B, M, K, transpose);
  return cips3d_launch_status();
}

extern "C" int cips3d_gemm_wgrad(const float* dy, const float* x, float* dwm, int B, int M, int K, int64_t P, void* stream) {
  if (!dy || !x || !dwm || B < 0 || M <= 0 || K <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (M % 32 || K % 32 || P % 4) return CIPS3D_E_UNSUPP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(dwm, 0, sizeof(float) * (size_t)B * M * K, st);
  if (e != hipSuccess) return (int)e;
  const int blocks = ceil_div(M, 64) * ceil_div(K, 64);
  // enough pixel chunks for ~4 workgroups per CU, at least 256 pixels each
  int64_t n_chunks = ceil_div<int64_t>(1024, (int64_t)blocks * B);
  const int64_t max_chunks = ceil_div<int64_t>(P, 256);
  if (n_chunks > max_chunks) n_chunks = max_chunks;
  if (n_chunks < 1) n_chunks = 1;
  int64_t chunk = ceil_div<int64_t>(ceil_div<int64_t>(P, n_chunks), 16) * 16;
  n_chunks = ceil_div<int64_t>(P, chunk);
  hipLaunchKernelGGL(gemm_wgrad_kernel, dim3((unsigned)blocks, (unsigned)n_chunks, (unsigned)B), dim3(256), 0, st, dy, x, dwm,
                     M, K, P, chunk);
  return cips3d_launch_status();
}

extern "C" int cips3d_gemm_wgrad_split(const float* dy, const float* x, float* dwm, int B, int M, int K, int64_t P,
                                       const float* dy_amax, const float* x_amax, int accumulate, void* stream) {
  if (!dy || !x || !dwm || B < 0 || M <= 0 || K <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (M % 32 || K % 32 || P % 32) return CIPS3D_E_UNSUPP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipStream_t st = as_stream(stream);
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(dwm, 0, sizeof(float) * (size_t)B * M * K, st);
    if (e != hipSuccess) return (int)e;
  }
  // workgroup tile: 128 x 64 (eight waves, two per SIMD: one converts while the other multiplies) where the matrix is that
  // large -- the L2-port bytes go with 1 / BM + 1 / BN -- else 64 x 64 / 32 x 32 (four waves)
  const int big = (M >= 128 && K >= 64) ? 2 : ((M >= 64 && K >= 64) ? 1 : 0);
  // (round 6, same box, 512 x 512 x 4096 x 2: 128 x 128 tiles in three wave layouts 33.1-38.0 us, 128 x 256 45.4 against 31.8 for
  // 128 x 64; twice / four times the pixel chunks 34.1 / 41.8: the tile and the chunk count below stay)
  const int BM = big == 2 ? 128 : (big == 1 ? 64 : 32), BN = big ? 64 : 32;
  const int blocks = ceil_div(M, BM) * ceil_div(K, BN);
  // pixel chunks: one to two workgroups per CU -- every chunk costs M K atomic adds per sample (1.3 TB/s chip-wide,
  // MI355X_MICROARCH.md) -- and >= 256 pixels each
  const int wgs = 256;
  int64_t n_chunks = ceil_div<int64_t>(wgs, (int64_t)blocks * B);
  const int64_t max_chunks = ceil_div<int64_t>(P, 256);
  if (n_chunks > max_chunks) n_chunks = max_chunks;
  if (n_chunks < 1) n_chunks = 1;
  int64_t chunk = ceil_div<int64_t>(ceil_div<int64_t>(P, n_chunks), 32) * 32;
  n_chunks = ceil_div<int64_t>(P, chunk);
  const int groups = (int)n_chunks * B;
  const int xg = (groups % 8 == 0) ? 1 : 0;
  const dim3 grid((unsigned)(blocks * groups));
  const int nc = (int)n_chunks;
  if (big == 2) hipLaunchKernelGGL((wgrad_split_kernel<2, 2, 4, 2>), grid, dim3(512), 0, st, dy, x, dwm, M, K, P, chunk, dy_amax, x_amax, blocks, nc, xg);
  else if (big == 1) hipLaunchKernelGGL((wgrad_split_kernel<2, 2, 2, 2>), grid, dim3(256), 0, st, dy, x, dwm, M, K, P, chunk, dy_amax, x_amax, blocks, nc, xg);
  else hipLaunchKernelGGL((wgrad_split_kernel<1, 1, 2, 2>), grid, dim3(256), 0, st, dy, x, dwm, M, K, P, chunk, dy_amax, x_amax, blocks, nc, xg);
  return cips3d_launch_status();
}

extern "C" int cips3d_noise_bias_act_bwd(const float* dy, const float* y, const float* noise, int64_t noise_bstride,
                                         const float* noise_w, float* dx, float* dnoise, float* dnoise_w, float* dbias,
                                         float* scratch_c, int B, int C, int64_t HW, void* stream) {
  if (!dy || !y || !dx || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (dnoise_w && !scratch_c) return CIPS3D_E_BADARG;
  if ((dnoise || dnoise_w) && (!noise || !noise_w)) return CIPS3D_E_BADARG;
  if (HW % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipStream_t st = as_stream(stream);
  // the per-channel accumulators are zeroed with ONE memset when the caller laid them out back to back (dbias, scratch_c):
  // every memset is a launch of its own, and the inversion step is as much launch-bound as it is GPU-bound
  if (dbias && dnoise_w && scratch_c == dbias + C) {
    hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * 2 * C, st);
    if (e != hipSuccess) return (int)e;
  } else {
    if (dbias) { hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * C, st); if (e != hipSuccess) return (int)e; }
    if (dnoise_w) { hipError_t e = hipMemsetAsync(scratch_c, 0, sizeof(float) * C, st); if (e != hipSuccess) return (int)e; }
  }
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(HW, 1024), (unsigned)C, (unsigned)B), dim3(256), 0, st, dy,
                     y, dx, dbias, noise, noise_bstride, dnoise_w ? scratch_c : nullptr, C, HW);
  if (dnoise_w) hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, st, scratch_c, C, dnoise_w);
  if (dnoise) {
    hipError_t e = hipMemsetAsync(dnoise, 0, sizeof(float) * (size_t)(noise_bstride ? B : 1) * HW, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(noise_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(HW, 256), (unsigned)ceil_div(C, 32), (unsigned)B),
                       dim3(256), 0, st, dx, noise_bstride, noise_w, dnoise, C, HW);
  }
  return cips3d_launch_status();
}

extern "C" int cips3d_torgb_bwd(const float* drgb, const float* x, const float* wm, float* dx, float* dwm, float* dbias,
                                int B, int C, int64_t HW, void* stream) {
  if (!drgb || !x || !wm || !dx || !dwm || B < 0 || C <= 0 || HW <= 0) return CIPS3D_E_BADARG;
  if (HW % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipStream_t st = as_stream(stream);
  const bool adjacent = dbias && dbias == dwm + (size_t)B * 3 * C;       // one memset for both (see cips3d_noise_bias_act_bwd)
  hipError_t e = hipMemsetAsync(dwm, 0, sizeof(float) * ((size_t)B * 3 * C + (adjacent ? 3 : 0)), st);
  if (e != hipSuccess) return (int)e;
  if (dbias && !adjacent) { e = hipMemsetAsync(dbias, 0, sizeof(float) * 3, st); if (e != hipSuccess) return (int)e; }
  hipLaunchKernelGGL(torgb_bwd_kernel, dim3((unsigned)ceil_div<int64_t>(HW, 1024), (unsigned)C, (unsigned)B), dim3(256), 0, st,
                     drgb, x, wm, dx, dwm, dbias, C, HW);
  return cips3d_launch_status();
}

extern "C" int cips3d_linear_table_bwd(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int in_dim, int B,
                                       const float* out_base, const float* dy_base, const float* x_base, float* dx_base,
                                       const int64_t* w_offsets_dev, float* dW, float* db, void* stream) {
  if (!table_dev || n_desc <= 0 || n_desc > 64 || total_rows <= 0 || in_dim <= 0 || B < 0 || !out_base || !dy_base || !x_base ||
      (dW && !w_offsets_dev))
    return CIPS3D_E_BADARG;
  if (in_dim % 4 != 0) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  TableBwd a{table_dev, n_desc, total_rows, in_dim, B, out_base, dy_base, x_base, dx_base, w_offsets_dev, dW, db};
  hipStream_t st = as_stream(stream);
  if (dW || db) hipLaunchKernelGGL(table_bwd_rows_kernel, dim3((unsigned)ceil_div(total_rows, 4)), dim3(256), 0, st, a);
  if (dx_base) hipLaunchKernelGGL(table_bwd_cols_kernel, dim3((unsigned)(n_desc * ((in_dim + 63) / 64) * TB_RG)), dim3(256), 0, st, a);
  return cips3d_launch_status();
}
