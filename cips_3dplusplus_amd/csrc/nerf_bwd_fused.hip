// Fused backward of the NeRF half on gfx950: d loss / d(FiLM table) and d loss / d(cam_poses) from the gradients of the
// feature map and the thumbnail, with the point MLP held in the register file in BOTH directions (the reference obtains
// these from autograd through SirenGenerator.forward and Render.volume_integration, cips3d/volume_renderer.py:39-160,
// cips3d/nerf_utils.py:264-338; flip inversion drives it, models/projector_v10.py:211-277).
//
// The materialised sequence (nerf_bwd.hip + the decoder GEMM) rebuilds every layer as act[b][c][p] in HBM and walks it with
// element-wise kernels: ~10 array passes of B*H*P floats per layer.  Here the work keeps the task shape of the forward
// render kernel (nerf.hip: a wave = 16 rays x a chunk of samples, lane = (quarter, ray), activations in MFMA D layout =
// next layer's B operand, weights streamed L2 -> LDS ring by LDS-DMA, split-fp16 products) and runs as two kernels around
// the per-ray compositing backward:
//
//   nerf_stash_kernel   forward recompute.  Per MFMA layer the fp32 accumulators (W' x, before FiLM) are written to `stash`
//                       in the wave's own register order (1 KiB per wave store, fully coalesced); per point it emits sdf,
//                       the rgb logits and g = <d_features[:, ray], feature> -- all the compositing backward needs.
//   composite_kernel    (nerf_bwd.hip) volume integration forward + backward per ray -> w, d sdf, d rgb logits, d |rays_d|
//   nerf_bwd_kernel     per 16-point tile, layers in reverse.  Epilogue of layer l (VALU):
//                           u    = upstream (accumulator of the previous MFMA layer, or the compositing gradient)
//                           uc   = u * cos(gamma' a + c)          a = stashed accumulator of layer l
//                           S1_c += sum_p uc a,  S2_c += sum_p uc  -> d gamma = 2^-s S1 + bias S2, d beta = S2
//                           y    = uc * gamma'                    (= d pre * 2^-s)
//                       then the data gradient  u_{l-1} = W_l^T d pre_l  on the matrix cores: A = transposed packed stream
//                       (same power-of-two scale 2^s as the forward matrix -> it cancels the 2^-s in y), B = y scaled PER
//                       POINT by a power of two S_p that puts its largest component at 2^12 (gradients are ~1e-6 and fp16's
//                       subnormal floor is 6e-8; the columns of a GEMM are independent, so the scale commutes and is undone
//                       exactly on the accumulator).  d viewdir and d point (3 components each) are VALU dots.
//                       The sums over the 16 points of a tile are a cross-lane reduce-scatter (DPP quad permutes + two
//                       bpermute stages: 49 instructions per 16 values instead of 64 + 15 selects), accumulated in LDS and
//                       flushed with one global atomic per workgroup and table entry.
//   camera_chain_kernel (nerf_bwd.hip) d points, d viewdirs, d |rays_d| -> d cam_poses
//
// HBM traffic: the stash is written once and read once (B * P * depth * H * 4 bytes each way); everything else is per-point
// scalars.  Renderer weights are constants of this path (`optim_render_params: false`, train_cips3d_compcars_v10.yaml:585).
#include <stdlib.h>

#include <atomic>
#include "common.h"
#include "nerf_mlp.h"

#ifdef CIPS3D_BWD_NO_COS         // timing-only ablation
#define cips3d_cos(x) ((x) * 0.001f)
#endif

// the forward recompute evaluates FiLM + sine like the render kernel (nerf.hip: CIPS3D_FILM_REVOLUTIONS)
#ifndef CIPS3D_FILM_REVOLUTIONS
#define CIPS3D_FILM_REVOLUTIONS 1
#endif
#if CIPS3D_FILM_REVOLUTIONS && !defined(CIPS3D_EXACT_SINE) && !defined(CIPS3D_REDUCED_SINE)
#define STASH_FILM_UNIT 0.159154943091895336f
#define STASH_FILM_SIN sin_revolutions
#else
#define STASH_FILM_UNIT 1.f
#define STASH_FILM_SIN cips3d_sin
#endif

namespace {

struct FusedArgs {
  cips3d_nerf_bwd_fused_params p;
  int groups;          // ray groups of 16 per view
  int tasks_per_view;  // groups * n_chunks rounded up to a multiple of WAVES
  int chunk;           // samples per chunk
  float t_end, t_step;
  // partition of p.scratch
  float* tables;       // [10][H]: w_first^T [3][H], 2^s * view-direction columns [3][H], w_sigma [H], w_rgb [3][H]
  float* dFt;          // [B][R][H]
  float* sdf;          // [B][P]
  float* crgb;         // [B][3][P]
  float* g;            // [B][P]
  float* wts;          // [B][P]
  float* Tb;           // [B][P]
  float* dsdf;         // [B][P]
  float* dcrgb;        // [B][3][P]
  float* dptsn;        // [B][3][P]
  float* dvd;          // [B][3][P]
  float* ddnorm;       // [B][R]
  float* sums;         // [B][L][2][H]: S1, S2
};

__host__ __device__ constexpr int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// ------------------------------------------------------------------------------------------------ small preparation kernels
// The three preparation steps of cips3d_nerf_bwd_fused as ONE launch (they were three dependent ~5 us launches in front of the
// backward's first kernel): blocks [0, n_t) transpose d_features, [n_t, n_t + n_p) stage the small tables, the rest zero the
// FiLM sums and the camera gradient (which the kernels behind accumulate into with atomics).
__global__ void __launch_bounds__(256) nerf_bwd_prep_kernel(const float* __restrict__ d_features, float* __restrict__ dFt, int H, int R,
                                                            int tiles_r, int tiles_h, int n_t, const float* __restrict__ w_first,
                                                            const float* __restrict__ w_view, const float* __restrict__ w_sigma,
                                                            const float* __restrict__ w_rgb, const float* __restrict__ packed, int D,
                                                            float* __restrict__ tables, int n_p, float* __restrict__ z0, int nz0,
                                                            float* __restrict__ z1, int nz1) {
  __shared__ float tile[32][33];
  const int bid = blockIdx.x;
  if (bid < n_t) {
    const int b = bid / (tiles_r * tiles_h), rem = bid % (tiles_r * tiles_h);
    const int r0 = (rem % tiles_r) * 32, h0 = (rem / tiles_r) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
      if (h0 + j < H && r0 + tx < R) tile[j][tx] = d_features[((int64_t)b * H + h0 + j) * R + r0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
      if (r0 + j < R && h0 + tx < H) dFt[((int64_t)b * R + r0 + j) * H + h0 + tx] = tile[tx][j];
  } else if (bid < n_t + n_p) {
    const float view_scale = packed[(int64_t)D * H * H + 2 * (D - 1)];
    for (int i = (bid - n_t) * 256 + threadIdx.x; i < 3 * H; i += n_p * 256) {
      const int k = i / H, o = i - k * H;
      tables[i] = w_first[o * 3 + k];
      tables[3 * H + i] = w_view[(int64_t)o * (H + 3) + H + k] * view_scale;
      tables[7 * H + i] = w_rgb[i];
      if (i < H) tables[6 * H + i] = w_sigma[i];
    }
  } else {
    const int nb = (int)gridDim.x - n_t - n_p;
    for (int i = (bid - n_t - n_p) * 256 + threadIdx.x; i < nz0 + nz1; i += nb * 256) {
      if (i < nz0) z0[i] = 0.f;
      else z1[i - nz0] = 0.f;
    }
  }
}

// transposed packed stream, consumption order of the backward: j = 0 view layer, j >= 1 hidden layer l = D - j.
//   packed_t[j][t][m][plane][lane][e] = plane of 2^s W_l[32 m + 16 (e>>2) + 4 (lane>>4) + (e&3)][16 t + (lane&15)]
// (row of the transposed matrix = input unit of W_l; contraction index = output unit of W_l, in D-layout order).
// scales: packed_t[D*H*H + 2 j], [.. + 1] = those of W_l in the forward stream.
__global__ void __launch_bounds__(256) pack_t_kernel(const float* __restrict__ w_hidden, const float* __restrict__ w_view,
                                                     const float* __restrict__ packed, float* __restrict__ packed_t, int H,
                                                     int D) {
  const int64_t per_layer = (int64_t)H * H;
  const int64_t total = per_layer * D;
  _Float16* out = reinterpret_cast<_Float16*>(packed_t);
  const float* fscales = packed + total;
  if (blockIdx.x == 0 && threadIdx.x < D) {
    const int j = threadIdx.x, fl = D - 1 - j;           // forward stream index of the same matrix
    packed_t[total + 2 * j] = fscales[2 * fl];
    packed_t[total + 2 * j + 1] = fscales[2 * fl + 1];
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i / per_layer);
    const int fl = D - 1 - j;
    int64_t rem = i - j * per_layer;
    const int tile_w = 16 * H;
    const int t = (int)(rem / tile_w);
    rem -= (int64_t)t * tile_w;
    const int m = (int)(rem / 512);
    const int lane = (int)((rem % 512) / 8);
    const int e = (int)(rem % 8);
    const int k_in = t * 16 + (lane & 15);                                   // row of W^T = input unit of W
    const int c_out = 32 * m + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);    // contraction index = output unit of W
    float v;
    if (fl < D - 1) v = w_hidden[(int64_t)fl * per_layer + (int64_t)c_out * H + k_in];
    else            v = w_view[(int64_t)c_out * (H + 3) + k_in];
    v *= fscales[2 * fl];
    _Float16 hi, lo;
    cips3d_split16(v, hi, lo);
    _Float16* blk = out + 2 * ((int64_t)j * per_layer + (int64_t)t * tile_w) + (int64_t)m * 1024;
    blk[lane * 8 + e] = hi;
    blk[512 + lane * 8 + e] = lo;
  }
}

// dfilm[b][l][0][c] = sinv_l S1 + bias[l][c] S2,  dfilm[b][l][1][c] = S2   (sinv_0 = 1, sinv_l = 2^-s of forward matrix l-1)
__global__ void __launch_bounds__(256) finalize_film_kernel(const float* __restrict__ sums, const float* __restrict__ bias,
                                                            const float* __restrict__ packed, int B, int L, int H,
                                                            float* __restrict__ dfilm) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * L * H) return;
  const int c = i % H, l = (i / H) % L, b = i / (H * L);
  const float* fscales = packed + (int64_t)(L - 1) * H * H;
  const float sinv = l >= 1 ? fscales[2 * (l - 1) + 1] : 1.f;
  const int64_t o = ((int64_t)(b * L + l) * 2) * H + c;
  const float s1 = sums[o], s2 = sums[o + H];
  dfilm[o] = fmaf(bias[l * H + c], s2, sinv * s1);
  dfilm[o + H] = s2;
}

// ------------------------------------------------------------------------------------------------ ray set-up (both kernels)
// identical arithmetic to nerf_render_kernel (nerf.hip; nerf_utils.py:38-121)
struct Ray {
  float ox, oy, oz, dx, dy, dz, vx, vy, vz, dnorm, nearv, farv, u, t_end, t_step;
  int N, has_u;
  __device__ __forceinline__ float zbase(int k) const {
    if (k >= N) return farv;
    const float t = (k < N / 2) ? t_step * (float)k : t_end - t_step * (float)(N - 1 - k);
    return nearv * (1.f - t) + farv * t;
  }
  __device__ __forceinline__ float z(int k) const {
    const float z0 = zbase(k);
    return has_u ? z0 + (zbase(k + 1) - z0) * u : z0;
  }
};

__device__ __forceinline__ Ray make_ray(const cips3d_nerf_bwd_geom& G, int b, int ray, float t_end, float t_step) {
  Ray r;
  const int S = G.img_size;
  const float focal = G.focals[b];
  r.nearv = G.near_[b]; r.farv = G.far_[b];
  const float* cw = G.cam_poses + 12 * b;
  const int pi = ray / S, pj = ray - pi * S;
  const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
  const float dcx = (px - (float)S * 0.5f) / focal;
  const float dcy = -(py - (float)S * 0.5f) / focal;
  const float dcz = -1.f;
  r.dx = (dcx * cw[0] + dcy * cw[1]) + dcz * cw[2];
  r.dy = (dcx * cw[4] + dcy * cw[5]) + dcz * cw[6];
  r.dz = (dcx * cw[8] + dcy * cw[9]) + dcz * cw[10];
  r.ox = cw[3]; r.oy = cw[7]; r.oz = cw[11];
  float vx = G.static_viewdirs ? dcx : r.dx, vy = G.static_viewdirs ? dcy : r.dy, vz = G.static_viewdirs ? dcz : r.dz;
  const float n = fmaxf(sqrtf((vx * vx + vy * vy) + vz * vz), 1e-12f);
  r.vx = vx / n; r.vy = vy / n; r.vz = vz / n;
  r.dnorm = sqrtf((r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz);
  r.has_u = G.perturb_u != nullptr;
  r.u = r.has_u ? G.perturb_u[(int64_t)b * S * S + ray] : 0.f;
  r.N = G.n_samples;
  r.t_end = t_end; r.t_step = t_step;
  return r;
}

// stage the FiLM table of view b: s_film[l][0][o] = gamma' (gamma 2^-s for the MFMA layers), s_film[l][1][o] = gamma bias + beta
// `unit`: 1 for radians (the backward kernel: its y = uc gamma' needs gamma' itself), 1 / 2 pi for the forward recompute, which
// evaluates the sines exactly as the render kernel does (nerf.hip, CIPS3D_FILM_REVOLUTIONS) so that both fill the same stash
__device__ __forceinline__ void stage_film(const cips3d_nerf_bwd_fused_params& P, int b, int H, int D, float* s_film, int tid,
                                           float unit = 1.f) {
  const int L = D + 1;
  const float* film_b = P.film + (int64_t)b * L * 2 * H;
  const float* scales = P.packed + (int64_t)D * H * H;
  for (int i = tid; i < L * H; i += WAVES * 64) {
    const int l = i / H, o = i - l * H;
    const float gm = film_b[(l * 2) * H + o];
    s_film[(l * 2) * H + o] = (l >= 1 ? gm * scales[2 * (l - 1) + 1] : gm) * unit;
    s_film[(l * 2 + 1) * H + o] = fmaf(gm, P.layer_bias[i], film_b[(l * 2 + 1) * H + o]) * unit;
  }
}

// the MFMA block of one slab step: acc[tt] += A(slab, tile tt) * (Xh + Xl), three fp16 products per fp32 product
template <int NT, int TPS>
__device__ __forceinline__ void mfma_step(const float* slab, const h8 (&Xh)[NT / 2], const h8 (&Xl)[NT / 2], f32x4 (&acc)[TPS],
                                          int lane) {
#ifdef CIPS3D_BWD_NO_MFMA       // timing-only ablation
  acc[0][0] += (float)Xh[0][0] + (float)Xl[NT / 2 - 1][1] + slab[lane];
  return;
#endif
  constexpr int TILE = 16 * NT * 16;
  // software-pipelined over half k-blocks (see nerf.hip:mfma_layer): the next half's fragments are requested before the current
  // half's MFMAs, into the other of two fragment buffers
  constexpr int HT = TPS / 2 > 0 ? TPS / 2 : 1;
  constexpr int NH = TPS / HT;                    // halves per k-block
  constexpr int NG = (NT / 2) * NH;
  h8 fh[2][HT], fl[2][HT];
  auto load_half = [&](int buf, int g) {
    const int m = g / NH, half = g % NH;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const int tt = half * HT + t;
      fh[buf][t] = *reinterpret_cast<const h8*>(slab + tt * TILE + ((2 * m) * 64 + lane) * 4);
      fl[buf][t] = *reinterpret_cast<const h8*>(slab + tt * TILE + ((2 * m + 1) * 64 + lane) * 4);
    }
  };
  load_half(0, 0);
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int m = g / NH, half = g % NH, cur = g & 1;
#pragma unroll
    for (int t = 0; t < HT; ++t) asm volatile("" : "+v"(fh[cur][t]), "+v"(fl[cur][t]));   // the wait sits here (nerf.hip)
    if (g + 1 < NG) load_half(cur ^ 1, g + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < HT; ++t)
      acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cur][t], Xh[m], acc[half * HT + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < HT; ++t)
      acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xl[m], acc[half * HT + t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < HT; ++t)
      acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xh[m], acc[half * HT + t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void step_wait_barrier() {
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's piece of the next slab has landed
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ forward recompute + stash
// One MFMA layer (forward), see nerf.hip:mfma_layer.  Differences: the accumulators go to `stash_l` (after the step barrier,
// so that no store acknowledgement is waited for at it), the view layer accumulates gdot = <dF, f> instead of compositing.
template <int NT, int TPS, bool VIEW>
__device__ __forceinline__ void stash_layer(const h8 (&Xh)[NT / 2], const h8 (&Xl)[NT / 2], h8 (&Yh)[NT / 2], h8 (&Yl)[NT / 2],
                                            float& gdot, float (&chead)[3], float& sdf_acc, bool last, Ring& ring,
                                            const float* film_l, const float* s_wd, const float* s_wc, const float* s_ws,
                                            const float* dF_ray, float* stash_l, float vx, float vy, float vz, int wave,
                                            int lane, int q4o) {
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  constexpr int STEPS = NT / TPS;
  constexpr int R = TPS * 4;
  constexpr int BPS = TPS / 2;
  const bool late = __builtin_amdgcn_readfirstlane(wave) >= WAVES / 2;
#pragma unroll
  for (int sl = 0; sl < STEPS; ++sl) {
    if (ring.seq + 1 < ring.seq_end) {
      const int nxt = (ring.seq + 1) % ring.per_sample;
      stage_slab<SLAB>(ring.packed + (int64_t)nxt * SLAB, ring.lds + ((ring.seq + 1) & 1) * SLAB, wave, lane);
    }
    const float* slab = ring.lds + (ring.seq & 1) * SLAB;
    const int o_base = sl * (TPS * 16) + q4o;
    f32x4 acc[TPS];
    float vxo = vx, vyo = vy, vzo = vz;
    if (VIEW) asm volatile("" : "+v"(vxo), "+v"(vyo), "+v"(vzo));
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int o4 = o_base + tt * 16;
      if (VIEW) {
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_wd + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_wd + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_wd + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[tt][i] = fmaf(wz[i], vzo, fmaf(wy[i], vyo, wx[i] * vxo));
      } else {
        acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    mfma_step<NT, TPS>(slab, Xh, Xl, acc, lane);
    if (late) step_wait_barrier();
    float res[R];
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int o4 = o_base + tt * 16;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
      if (VIEW) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_wc + o4);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_wc + H + o4);
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(s_wc + 2 * H + o4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dF_ray + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = STASH_FILM_SIN(fmaf(g4[i], acc[tt][i], c4[i]));
          gdot = fmaf(d4[i], f, gdot);
          chead[0] = fmaf(w0[i], f, chead[0]);
          chead[1] = fmaf(w1[i], f, chead[1]);
          chead[2] = fmaf(w2[i], f, chead[2]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) res[tt * 4 + i] = STASH_FILM_SIN(fmaf(g4[i], acc[tt][i], c4[i]));
        if (last) {
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) sdf_acc = fmaf(ws4[i], res[tt * 4 + i], sdf_acc);
        }
      }
    }
    if (VIEW) {
      asm volatile("" : "+v"(chead[0]), "+v"(chead[1]), "+v"(chead[2]), "+v"(gdot));
    } else {
#pragma unroll
      for (int k = 0; k < R; ++k) asm volatile("" : "+v"(res[k]));
#pragma unroll
      for (int bb = 0; bb < BPS; ++bb) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = res[(2 * bb) * 4 + j];
        split8(v8, Yh[sl * BPS + bb], Yl[sl * BPS + bb]);
      }
    }
    if (!late) step_wait_barrier();
    // the step's accumulators, in register order: [sl * TPS + tt][lane][4]
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) *reinterpret_cast<f32x4*>(stash_l + ((sl * TPS + tt) * 64 + lane) * 4) = acc[tt];
    ++ring.seq;
  }
}

template <int NT, int TPS>
__global__ void __launch_bounds__(WAVES * 64, 2) nerf_stash_kernel(FusedArgs a) {
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cips3d_nerf_bwd_fused_params& P = a.p;
  const cips3d_nerf_bwd_geom& G = P.geom;
  const int D = P.depth;
  const int L = D + 1;
  float* ringmem = lds;                      // 2 * SLAB
  float* s_film = ringmem + 2 * SLAB;        // L * 2 * H
  float* s_tab = s_film + L * 2 * H;         // 10 * H (the prepared tables)
  const float* s_w0 = s_tab;
  const float* s_wd = s_tab + 3 * H;
  const float* s_ws = s_tab + 6 * H;
  const float* s_wc = s_tab + 7 * H;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int qd = lane >> 4;
  const int pl = lane & 15;

  const int64_t task0 = (int64_t)blockIdx.x * WAVES;
  const int b = (int)(task0 / a.tasks_per_view);
  const int tv = (int)(task0 % a.tasks_per_view) + wave;
  const bool task_ok = tv < a.groups * P.n_chunks;
  const int g = task_ok ? tv / P.n_chunks : 0;
  const int c = task_ok ? tv % P.n_chunks : 0;
  const int S = G.img_size;
  const int R = S * S;
  const int N = G.n_samples;
  const int64_t Pn = (int64_t)R * N;
  const int ray = g * RAYS + pl;                       // R % 16 == 0: always a valid ray
  stage_film(P, b, H, D, s_film, tid, STASH_FILM_UNIT);
  for (int i = tid; i < 10 * H; i += WAVES * 64) s_tab[i] = a.tables[i];

  const float b_sigma = P.b_sigma[0], b_rgb0 = P.b_rgb[0], b_rgb1 = P.b_rgb[1], b_rgb2 = P.b_rgb[2];
  const Ray ry = make_ray(G, b, ray, a.t_end, a.t_step);
  const float span = ry.farv - ry.nearv;
  const float* dF_ray = a.dFt + ((int64_t)b * R + ray) * H;

  Ring ring;
  ring.packed = P.packed;
  ring.lds = ringmem;
  ring.seq = 0;
  ring.per_sample = D * (NT / TPS);
  ring.seq_end = a.chunk * ring.per_sample;
  stage_slab<SLAB>(P.packed, ringmem, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const int64_t tg = task0 + wave;                     // global task index: stash rows of this wave
  const int s_begin = c * a.chunk;
  for (int si = 0; si < a.chunk; ++si) {
    const int sg = s_begin + si;
    const bool live = task_ok && sg < N;
    const int sk = sg < N ? sg : N - 1;
    const float z = ry.z(sk);
    const float ptx = ry.ox + ry.dx * z, pty = ry.oy + ry.dy * z, ptz = ry.oz + ry.dz * z;
    const float nx = ptx * 2.f / span, ny = pty * 2.f / span, nz = ptz * 2.f / span;
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int q4o = 4 * qd + opq;
    float* stash_s = P.stash + ((tg * a.chunk + si) * D) * (int64_t)(16 * H);

    h8 Xh[NT / 2], Xl[NT / 2], Yh[NT / 2], Yl[NT / 2];
    float sdf = 0.f;
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
      float v8[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int o4 = (2 * m + hf) * 16 + q4o;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_w0 + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_w0 + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_w0 + 2 * H + o4);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_film + o4);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_film + H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pre = fmaf(wz[i], nz, fmaf(wy[i], ny, wx[i] * nx));
          v8[hf * 4 + i] = STASH_FILM_SIN(fmaf(g4[i], pre, c4[i]));
        }
        if (D == 1) {
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) sdf = fmaf(ws4[i], v8[hf * 4 + i], sdf);
        }
      }
      split8(v8, Xh[m], Xl[m]);
    }
    float chead[3] = {0.f, 0.f, 0.f};
    float gdot = 0.f;
    for (int l = 1; l < D; ++l) {
      stash_layer<NT, TPS, false>(Xh, Xl, Yh, Yl, gdot, chead, sdf, l == D - 1, ring, s_film + l * 2 * H, s_wd, s_wc, s_ws,
                                  dF_ray, stash_s + (int64_t)(l - 1) * 16 * H, ry.vx, ry.vy, ry.vz, wave, lane, q4o);
#pragma unroll
      for (int i = 0; i < NT / 2; ++i) { Xh[i] = Yh[i]; Xl[i] = Yl[i]; }
    }
    sdf += __shfl_xor(sdf, 16, 64);
    sdf += __shfl_xor(sdf, 32, 64);
    sdf += b_sigma;
    float sdf_unused = 0.f;
    stash_layer<NT, TPS, true>(Xh, Xl, Yh, Yl, gdot, chead, sdf_unused, false, ring, s_film + D * 2 * H, s_wd, s_wc, s_ws,
                               dF_ray, stash_s + (int64_t)(D - 1) * 16 * H, ry.vx, ry.vy, ry.vz, wave, lane, q4o);
    float c0 = chead[0], c1 = chead[1], c2 = chead[2];
    c0 += __shfl_xor(c0, 16, 64); c1 += __shfl_xor(c1, 16, 64); c2 += __shfl_xor(c2, 16, 64); gdot += __shfl_xor(gdot, 16, 64);
    c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64); gdot += __shfl_xor(gdot, 32, 64);
    if (live && qd == 0) {
      const int64_t p = (int64_t)sg * R + ray;
      a.sdf[(int64_t)b * Pn + p] = sdf;
      a.crgb[((int64_t)b * 3 + 0) * Pn + p] = c0 + b_rgb0;
      a.crgb[((int64_t)b * 3 + 1) * Pn + p] = c1 + b_rgb1;
      a.crgb[((int64_t)b * 3 + 2) * Pn + p] = c2 + b_rgb2;
      a.g[(int64_t)b * Pn + p] = gdot;
    }
  }
}

// ------------------------------------------------------------------------------------------------ g from the forward's stash
// g[b][p] = <d_features[:, ray], f[:, p]>,  f = sin(gamma' a + c) of the view layer's stashed accumulators: all that is left to
// rebuild when the differentiable forward (cips3d_nerf_render with cips3d_nerf_params.stash) filled the stash itself.
// Same tasks as the render kernel; one pass over 1/depth of the stash.
template <int NT>
__global__ void __launch_bounds__(WAVES * 64) nerf_g_kernel(FusedArgs a) {
  constexpr int H = NT * 16;
  __shared__ __attribute__((aligned(16))) float s_gc[2 * H];
  const cips3d_nerf_bwd_fused_params& P = a.p;
  const cips3d_nerf_bwd_geom& G = P.geom;
  const int D = P.depth, L = D + 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, qd = lane >> 4, pl = lane & 15;
  const int64_t task0 = (int64_t)blockIdx.x * WAVES;
  const int b = (int)(task0 / a.tasks_per_view);
  const int tv = (int)(task0 % a.tasks_per_view) + wave;
  const bool task_ok = tv < a.groups * P.n_chunks;
  const int g = task_ok ? tv / P.n_chunks : 0;
  const int c = task_ok ? tv % P.n_chunks : 0;
  const int R = G.img_size * G.img_size, N = G.n_samples;
  const int64_t Pn = (int64_t)R * N;
  const int ray = g * RAYS + pl;
  {
    const float* film_b = P.film + ((int64_t)b * L + D) * 2 * H;
    const float sinv = P.packed[(int64_t)D * H * H + 2 * (D - 1) + 1];
    for (int i = tid; i < H; i += WAVES * 64) {
      const float gm = film_b[i];
      s_gc[i] = gm * sinv;
      s_gc[H + i] = fmaf(gm, P.layer_bias[D * H + i], film_b[H + i]);
    }
  }
  __syncthreads();
  const float* dF_ray = a.dFt + ((int64_t)b * R + ray) * H;
  const int64_t tg = task0 + wave;
  for (int si = 0; si < a.chunk; ++si) {
    const int sg = c * a.chunk + si;
    if (!(task_ok && sg < N)) continue;
    const float* st = P.stash + (((tg * a.chunk + si) * D) + (D - 1)) * (int64_t)(16 * H);
    float acc = 0.f;
#pragma unroll 4
    for (int t = 0; t < NT; ++t) {
      const int o4 = t * 16 + 4 * qd;
      const f32x4 av = *reinterpret_cast<const f32x4*>(st + (t * 64 + lane) * 4);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(dF_ray + o4);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_gc + o4);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_gc + H + o4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = fmaf(d4[i], cips3d_sin(fmaf(g4[i], av[i], c4[i])), acc);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (qd == 0) a.g[(int64_t)b * Pn + (int64_t)sg * R + ray] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ backward kernel
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Sums over the 16 lanes of a row (the 16 points of the tile) as a reduce-scatter: every stage halves the number of values a
// lane carries (it keeps the half selected by one bit of its index and hands the other half to its partner), so that lane p
// of the row ends with the row total of value p: unit i = p & 3 of o-tile tt = p >> 2 of the step.
// Stages 1, 2 (partners 1 and 2 lanes away; DPP quad permutes fused into the adds) run per o-tile as soon as its four values
// exist, stages 3, 4 (partners 4 and 8 lanes away; ds_bpermute) once per step over the tiles.
__device__ __forceinline__ float quad_reduce_scatter(float v0, float v1, float v2, float v3, bool b0, bool b1) {
#ifdef CIPS3D_BWD_NO_SUMS
  return v0 + v1 + v2 + v3;
#endif
  const float k0 = b0 ? v1 : v0, s0 = b0 ? v0 : v1;
  const float k1 = b0 ? v3 : v2, s1 = b0 ? v2 : v3;
  const float r0 = k0 + dpp_perm<0xB1>(s0);        // quad_perm [1,0,3,2]
  const float r1 = k1 + dpp_perm<0xB1>(s1);
  const float k = b1 ? r1 : r0, sd = b1 ? r0 : r1;
  return k + dpp_perm<0x4E>(sd);                   // quad_perm [2,3,0,1]
}
template <int TPS>
__device__ __forceinline__ float tile_reduce_scatter(const float (&q)[TPS], bool b2, bool b3) {
  static_assert(TPS == 2 || TPS == 4, "two or four o-tiles per step");
  const float r0 = (b2 ? q[1] : q[0]) + __shfl_xor(b2 ? q[0] : q[1], 4, 64);
  if constexpr (TPS == 4) {
    const float r1 = (b2 ? q[3] : q[2]) + __shfl_xor(b2 ? q[2] : q[3], 4, 64);
    return (b3 ? r1 : r0) + __shfl_xor(b3 ? r0 : r1, 8, 64);
  } else {
    return r0 + __shfl_xor(r0, 8, 64);             // both halves of the row hold the total of value p & 7
  }
}
// adds the step's row totals into the LDS sums of the layer (S1 at sum_l, S2 at sum_l + H)
template <int TPS>
__device__ __forceinline__ void accumulate_sums(const float (&q1)[TPS], const float (&q2)[TPS], float* sum_l, int H, int o_step,
                                                int lane) {
#ifdef CIPS3D_BWD_NO_SUMS       // timing-only ablation
  if (q1[0] == 1.2345f) atomicAdd(sum_l, q2[0]);
  return;
#endif
  const bool b2 = lane & 4, b3 = lane & 8;
  const float t1 = tile_reduce_scatter<TPS>(q1, b2, b3);
  const float t2 = tile_reduce_scatter<TPS>(q2, b2, b3);
  const int p = lane & 15;
  if (p < TPS * 4) {
    const int o = o_step + (p >> 2) * 16 + 4 * (lane >> 4) + (p & 3);
    atomicAdd(sum_l + o, t1);
    atomicAdd(sum_l + H + o, t2);
  }
}

// per-point power-of-two scale: S = 2^(12 - exponent(m)) (1 for m == 0), clamped to 2^+-100; returns S, writes 1/S
__device__ __forceinline__ float point_scale(float m, float& inv) {
  const int e = (__float_as_int(m) >> 23) & 0xff;          // biased exponent; m >= 0
  int k = 139 - e;                                          // 12 - (e - 127)
  k = m > 0.f ? k : 0;
  k = k > 100 ? 100 : (k < -100 ? -100 : k);
  inv = __int_as_float((127 - k) << 23);
  return __int_as_float((127 + k) << 23);
}

// Y (fp32, D layout, the whole layer) -> the split B fragments of the next MFMA layer, scaled per point; returns 1/S
template <int NT>
__device__ __forceinline__ float scale_split(const float (&Yf)[NT * 4], h8 (&Xh)[NT / 2], h8 (&Xl)[NT / 2]) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) m = fmaxf(m, fabsf(Yf[i]));
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float inv;
  const float S = point_scale(m, inv);
#pragma unroll
  for (int mb = 0; mb < NT / 2; ++mb) {
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = Yf[mb * 8 + j] * S;
    split8(v8, Xh[mb], Xl[mb]);
  }
  return inv;
}

// One transposed MFMA layer of the backward + the epilogue of layer l (the layer whose input gradient it produces):
//   u = acc * uscale (+ ws * dsdf when l is the last hidden layer: `sigma`);  a = stash (l >= 1) or W0 . n (l == 0: `first`)
//   uc = u cos(gamma' a + c);  sums;  y = uc gamma'
// l >= 1: y -> Yf (fp32, whole layer).  l == 0: y = d pre_0; dp[j] += W0[c][j] y.
// The slab steps are a ROLLED loop and `first` / `sigma` are run-time (wave-uniform) switches: with every layer kind and step
// unrolled the kernel was ~100 KB of code, more than the instruction cache holds across a sample (1.08 ms at D=6, B=2; the
// forward kernel has 40 KB).  The step's results enter Yf at the top and the array is rotated by R per step, so every index
// is a constant (48 moves per step next to 96 MFMAs).
template <int NT, int TPS>
__device__ __forceinline__ void bwd_layer(const h8 (&Xh)[NT / 2], const h8 (&Xl)[NT / 2], float (&Yf)[NT * 4], float (&dp)[3],
                                          Ring& ring, const float* film_l, float* sum_l, const float* stash_l,
                                          const float* tables, float uscale, bool first, bool sigma, float dsdf, float nx,
                                          float ny, float nz, int wave, int lane, int q4o) {
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  constexpr int STEPS = NT / TPS;
  constexpr int R = TPS * 4;
  const bool late = __builtin_amdgcn_readfirstlane(wave) >= WAVES / 2;
  const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll 1
  for (int sl = 0; sl < STEPS; ++sl) {
    if (ring.seq + 1 < ring.seq_end) {
      const int nxt = (ring.seq + 1) % ring.per_sample;
      stage_slab<SLAB>(ring.packed + (int64_t)nxt * SLAB, ring.lds + ((ring.seq + 1) & 1) * SLAB, wave, lane);
    }
    const float* slab = ring.lds + (ring.seq & 1) * SLAB;
    const int o_base = sl * (TPS * 16) + q4o;
    // operands of the epilogue, requested before the matrix block so that they arrive under it
    // (register diet: the sigma-head weights and the first layer's three weight rows are needed by ONE of the D layers each --
    // they are fetched where they are used, one o-tile at a time, instead of living through the matrix block: the 64 registers
    // they took were what the kernel spilled)
    f32x4 st[TPS];
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      st[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef CIPS3D_BWD_NO_STASH     // (timing-only ablation)
      if (!first) st[tt] = *reinterpret_cast<const f32x4*>(stash_l + ((sl * TPS + tt) * 64 + lane) * 4);
#endif
    }
    f32x4 acc[TPS];
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    mfma_step<NT, TPS>(slab, Xh, Xl, acc, lane);
    if (late) step_wait_barrier();
    if (first) {
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int o4 = o_base + tt * 16;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(tables + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(tables + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(tables + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) st[tt][i] = fmaf(wz[i], nz, fmaf(wy[i], ny, wx[i] * nx));
      }
    }
    float q1[TPS], q2[TPS], yv[R];
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int o4 = o_base + tt * 16;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
      f32x4 wsg = {0.f, 0.f, 0.f, 0.f};
      if (sigma) wsg = *reinterpret_cast<const f32x4*>(tables + 6 * H + o4);
      float e1[4], e2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float av = st[tt][i];
        const float u = fmaf(wsg[i], dsdf, acc[tt][i] * uscale);
        const float uc = u * cips3d_cos(fmaf(g4[i], av, c4[i]));
        e1[i] = uc * av;
        e2[i] = uc;
        yv[tt * 4 + i] = uc * g4[i];
      }
      q1[tt] = quad_reduce_scatter(e1[0], e1[1], e1[2], e1[3], b0, b1);
      q2[tt] = quad_reduce_scatter(e2[0], e2[1], e2[2], e2[3], b0, b1);
    }
    if (first) {
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int o4 = o_base + tt * 16;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(tables + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(tables + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(tables + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dp[0] = fmaf(wx[i], yv[tt * 4 + i], dp[0]);
          dp[1] = fmaf(wy[i], yv[tt * 4 + i], dp[1]);
          dp[2] = fmaf(wz[i], yv[tt * 4 + i], dp[2]);
        }
      }
    }
    accumulate_sums<TPS>(q1, q2, sum_l, H, sl * (TPS * 16), lane);
    // (the sink pass would otherwise move the epilogue below the step barrier, undoing the stagger)
#pragma unroll
    for (int k = 0; k < R; ++k) asm volatile("" : "+v"(yv[k]));
    asm volatile("" : "+v"(dp[0]), "+v"(dp[1]), "+v"(dp[2]));
#pragma unroll
    for (int k = 0; k < NT * 4 - R; ++k) Yf[k] = Yf[k + R];
#pragma unroll
    for (int k = 0; k < R; ++k) Yf[NT * 4 - R + k] = yv[k];
    if (!late) step_wait_barrier();
    ++ring.seq;
  }
}

template <int NT, int TPS>
__global__ void __launch_bounds__(WAVES * 64, 2) nerf_bwd_kernel(FusedArgs a) {
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  constexpr int STEPS = NT / TPS;
  constexpr int R = TPS * 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cips3d_nerf_bwd_fused_params& P = a.p;
  const cips3d_nerf_bwd_geom& G = P.geom;
  const int D = P.depth;
  const int L = D + 1;
  float* ringmem = lds;                      // 2 * SLAB
  float* s_film = ringmem + 2 * SLAB;        // L * 2 * H
  float* s_sum = s_film + L * 2 * H;         // L * 2 * H

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int qd = lane >> 4;
  const int pl = lane & 15;

  const int64_t task0 = (int64_t)blockIdx.x * WAVES;
  const int b = (int)(task0 / a.tasks_per_view);
  const int tv = (int)(task0 % a.tasks_per_view) + wave;
  const bool task_ok = tv < a.groups * P.n_chunks;
  const int g = task_ok ? tv / P.n_chunks : 0;
  const int c = task_ok ? tv % P.n_chunks : 0;
  const int S = G.img_size;
  const int Rn = S * S;
  const int N = G.n_samples;
  const int64_t Pn = (int64_t)Rn * N;
  const int ray = g * RAYS + pl;
  stage_film(P, b, H, D, s_film, tid);
  for (int i = tid; i < L * 2 * H; i += WAVES * 64) s_sum[i] = 0.f;

  const Ray ry = make_ray(G, b, ray, a.t_end, a.t_step);
  const float span = ry.farv - ry.nearv;
  const float* dF_ray = a.dFt + ((int64_t)b * Rn + ray) * H;
  const float* tscales = P.packed_t + (int64_t)D * H * H;     // [j][2]
  const float* fscales = P.packed + (int64_t)D * H * H;

  Ring ring;
  ring.packed = P.packed_t;
  ring.lds = ringmem;
  ring.seq = 0;
  ring.per_sample = D * (NT / TPS);
  ring.seq_end = a.chunk * ring.per_sample;
  stage_slab<SLAB>(P.packed_t, ringmem, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const int64_t tg = task0 + wave;
  const int s_begin = c * a.chunk;
  for (int si = 0; si < a.chunk; ++si) {
    const int sg = s_begin + si;
    const bool live = task_ok && sg < N;
    const int sk = sg < N ? sg : N - 1;
    const float z = ry.z(sk);
    const float nx = (ry.ox + ry.dx * z) * 2.f / span, ny = (ry.oy + ry.dy * z) * 2.f / span,
                nz = (ry.oz + ry.dz * z) * 2.f / span;
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int q4o = 4 * qd + opq;
    const float* stash_s = P.stash + ((tg * a.chunk + si) * D) * (int64_t)(16 * H);
    const int64_t p = (int64_t)sk * Rn + ray;
    const float w = live ? a.wts[(int64_t)b * Pn + p] : 0.f;
    const float dsdf = live ? a.dsdf[(int64_t)b * Pn + p] : 0.f;
    const float dc0 = live ? a.dcrgb[((int64_t)b * 3 + 0) * Pn + p] : 0.f;
    const float dc1 = live ? a.dcrgb[((int64_t)b * 3 + 1) * Pn + p] : 0.f;
    const float dc2 = live ? a.dcrgb[((int64_t)b * 3 + 2) * Pn + p] : 0.f;

    h8 Xh[NT / 2], Xl[NT / 2];
    float Yf[NT * 4];
    float dv[3] = {0.f, 0.f, 0.f}, dp[3] = {0.f, 0.f, 0.f};
    // ---- view layer epilogue (no matrix work): upstream = w dF + Wc^T d(rgb logits); rolled like the slab steps
    {
      const float* film_l = s_film + D * 2 * H;
      const float* stash_l = stash_s + (int64_t)(D - 1) * 16 * H;
      const bool b0 = lane & 1, b1 = lane & 2;
#pragma unroll 1
      for (int sl = 0; sl < STEPS; ++sl) {
        const int o_base = sl * (TPS * 16) + q4o;
        float q1[TPS], q2[TPS], yv[R];
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
          const int o4 = o_base + tt * 16;
          const f32x4 av = *reinterpret_cast<const f32x4*>(stash_l + ((sl * TPS + tt) * 64 + lane) * 4);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(dF_ray + o4);
          const f32x4 wd0 = *reinterpret_cast<const f32x4*>(a.tables + 3 * H + o4);
          const f32x4 wd1 = *reinterpret_cast<const f32x4*>(a.tables + 4 * H + o4);
          const f32x4 wd2 = *reinterpret_cast<const f32x4*>(a.tables + 5 * H + o4);
          const f32x4 wc0 = *reinterpret_cast<const f32x4*>(a.tables + 7 * H + o4);
          const f32x4 wc1 = *reinterpret_cast<const f32x4*>(a.tables + 8 * H + o4);
          const f32x4 wc2 = *reinterpret_cast<const f32x4*>(a.tables + 9 * H + o4);
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
          float e1[4], e2[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float u = fmaf(w, d4[i], fmaf(wc2[i], dc2, fmaf(wc1[i], dc1, wc0[i] * dc0)));
            const float uc = u * cips3d_cos(fmaf(g4[i], av[i], c4[i]));
            e1[i] = uc * av[i];
            e2[i] = uc;
            const float y = uc * g4[i];
            yv[tt * 4 + i] = y;
            dv[0] = fmaf(wd0[i], y, dv[0]);
            dv[1] = fmaf(wd1[i], y, dv[1]);
            dv[2] = fmaf(wd2[i], y, dv[2]);
          }
          q1[tt] = quad_reduce_scatter(e1[0], e1[1], e1[2], e1[3], b0, b1);
          q2[tt] = quad_reduce_scatter(e2[0], e2[1], e2[2], e2[3], b0, b1);
        }
        accumulate_sums<TPS>(q1, q2, s_sum + D * 2 * H, H, sl * (TPS * 16), lane);
#pragma unroll
        for (int k = 0; k < NT * 4 - R; ++k) Yf[k] = Yf[k + R];
#pragma unroll
        for (int k = 0; k < R; ++k) Yf[NT * 4 - R + k] = yv[k];
      }
    }
    float inv_s = scale_split<NT>(Yf, Xh, Xl);
    // ---- transposed MFMA layers j = 0 .. D-1: the input gradient of layer l = D - j, then the epilogue of layer l - 1
    for (int j = 0; j < D; ++j) {
      const int l = D - 1 - j;                                    // the layer whose epilogue follows
      // the transposed stream carries the forward matrix's 2^s, y carries 2^-s: the product of the two stored scales is 1
      const float uscale = inv_s * (tscales[2 * j + 1] * fscales[2 * (D - 1 - j)]);
      const float* st_l = l >= 1 ? stash_s + (int64_t)(l - 1) * 16 * H : stash_s;
      bwd_layer<NT, TPS>(Xh, Xl, Yf, dp, ring, s_film + l * 2 * H, s_sum + l * 2 * H, st_l, a.tables, uscale, l == 0, j == 0,
                         j == 0 ? dsdf : 0.f, nx, ny, nz, wave, lane, q4o);
      if (l >= 1) inv_s = scale_split<NT>(Yf, Xh, Xl);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dv[k] += __shfl_xor(dv[k], 16, 64); dv[k] += __shfl_xor(dv[k], 32, 64);
      dp[k] += __shfl_xor(dp[k], 16, 64); dp[k] += __shfl_xor(dp[k], 32, 64);
    }
    if (live && qd == 0) {
      const int64_t po = (int64_t)sg * Rn + ray;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a.dptsn[((int64_t)b * 3 + k) * Pn + po] = dp[k];
        a.dvd[((int64_t)b * 3 + k) * Pn + po] = dv[k];
      }
    }
  }
  // ---- flush the workgroup's sums
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  float* gs = a.sums + (int64_t)b * L * 2 * H;
  for (int i = tid; i < L * 2 * H; i += WAVES * 64) unsafeAtomicAdd(gs + i, s_sum[i]);
}

template <int NT, int TPS>
int launch_fused(const FusedArgs& a, hipStream_t st) {
  const cips3d_nerf_bwd_fused_params& P = a.p;
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  const int L = P.depth + 1;
  const size_t lds_a = sizeof(float) * ((size_t)2 * SLAB + (size_t)L * 2 * H + 10 * H);
  const size_t lds_b = sizeof(float) * ((size_t)2 * SLAB + (size_t)L * 4 * H);
  if (lds_a > 160 * 1024 || lds_b > 160 * 1024) return CIPS3D_E_UNSUPP;
  static std::atomic<unsigned long long> attr_set{0};
  int dev_id = 0;
  if (hipError_t e = hipGetDevice(&dev_id); e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev_id & 63);
  if (dev_id >= 64 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_stash_kernel<NT, TPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_bwd_kernel<NT, TPS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const cips3d_nerf_bwd_geom& G = P.geom;
  const unsigned wgs = (unsigned)((int64_t)G.B * a.tasks_per_view / WAVES);
  if (P.fwd_sdf) hipLaunchKernelGGL((nerf_g_kernel<NT>), dim3(wgs), dim3(WAVES * 64), 0, st, a);
  else hipLaunchKernelGGL((nerf_stash_kernel<NT, TPS>), dim3(wgs), dim3(WAVES * 64), lds_a, st, a);
  if (int rc = cips3d_launch_status()) return rc;
  if (int rc = cips3d_nerf_bwd_composite(&G, a.sdf, a.crgb, a.g, P.d_thumb, P.sigmoid_beta, a.wts, a.Tb, a.dsdf, a.dcrgb,
                                         a.ddnorm, nullptr, st))
    return rc;
  hipLaunchKernelGGL((nerf_bwd_kernel<NT, TPS>), dim3(wgs), dim3(WAVES * 64), lds_b, st, a);
  return cips3d_launch_status();
}

bool fused_shape_ok(int H, int D, int img_size, int n_samples) {
  if (H != 32 && H != 64 && H != 128 && H != 256) return false;
  if (D < 1 || D > 64 || img_size <= 0 || n_samples <= 0) return false;
  if ((img_size * img_size) % 16 != 0) return false;
  // the narrowest slab (two o-tiles per step) + both tables of the backward kernel within the LDS
  return sizeof(float) * ((size_t)2 * 16 * H * 2 + (size_t)(D + 1) * 4 * H) <= 160 * 1024 &&
         sizeof(float) * ((size_t)2 * 16 * H * 2 + (size_t)(D + 1) * 2 * H + 10 * H) <= 160 * 1024;
}

}  // namespace

extern "C" int cips3d_nerf_bwd_fused_supported(int hidden, int depth, int img_size, int n_samples) {
  return fused_shape_ok(hidden, depth, img_size, n_samples) ? 1 : 0;
}

extern "C" int64_t cips3d_nerf_bwd_fused_stash_floats(int B, int img_size, int n_samples, int hidden, int depth, int n_chunks) {
  if (B <= 0 || img_size <= 0 || n_samples <= 0 || hidden <= 0 || depth <= 0 || n_chunks <= 0) return 0;
  const int64_t groups = ceil_div<int64_t>((int64_t)img_size * img_size, RAYS);
  const int64_t tasks_per_view = ceil_div<int64_t>(groups * n_chunks, WAVES) * WAVES;
  const int64_t chunk = ceil_div(n_samples, n_chunks);
  return (int64_t)B * tasks_per_view * chunk * depth * 16 * hidden;
}

extern "C" int64_t cips3d_nerf_bwd_fused_scratch_floats(int B, int img_size, int n_samples, int hidden, int depth) {
  if (B <= 0 || img_size <= 0 || n_samples <= 0 || hidden <= 0 || depth <= 0) return 0;
  const int64_t R = (int64_t)img_size * img_size, P = R * n_samples, H = hidden, L = depth + 1;
  return align4(10 * H) + align4(B * R * H) + 17 * align4(B * P) + align4(B * R) + align4(B * L * 2 * H);
}

extern "C" int cips3d_nerf_pack_weights_t(const float* w_hidden, const float* w_view, const float* packed, float* packed_t,
                                          int hidden, int depth, void* stream) {
  if (!w_view || !packed || !packed_t || hidden <= 0 || depth < 1 || (depth > 1 && !w_hidden)) return CIPS3D_E_BADARG;
  if (hidden != 32 && hidden != 64 && hidden != 128 && hidden != 256) return CIPS3D_E_UNSUPP;
  if (depth > 64) return CIPS3D_E_UNSUPP;
  const int64_t total = (int64_t)hidden * hidden * depth;
  hipLaunchKernelGGL(pack_t_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0, as_stream(stream), w_hidden,
                     w_view, packed, packed_t, hidden, depth);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_fused(const cips3d_nerf_bwd_fused_params* pp, void* stream) {
  if (!pp) return CIPS3D_E_BADARG;
  const cips3d_nerf_bwd_fused_params& P = *pp;
  const cips3d_nerf_bwd_geom& G = P.geom;
  if (!G.cam_poses || !G.focals || !G.near_ || !G.far_ || G.B < 0 || G.img_size <= 0 || G.n_samples <= 0) return CIPS3D_E_BADARG;
  if (!P.w_first || !P.packed || !P.packed_t || !P.w_view || !P.film || !P.layer_bias || !P.w_sigma || !P.b_sigma || !P.w_rgb ||
      !P.b_rgb || !P.d_features || !P.d_thumb || !P.stash || !P.scratch || !P.dfilm || !P.dcam)
    return CIPS3D_E_BADARG;
  if (P.n_chunks < 1 || P.n_chunks > G.n_samples) return CIPS3D_E_BADARG;
  if (!fused_shape_ok(P.hidden, P.depth, G.img_size, G.n_samples)) return CIPS3D_E_UNSUPP;
  if (G.B == 0) return 0;
  hipStream_t st = as_stream(stream);
  const int H = P.hidden, D = P.depth, L = D + 1, B = G.B;
  const int64_t R = (int64_t)G.img_size * G.img_size, Pn = R * G.n_samples;
  FusedArgs a;
  a.p = P;
  a.groups = (int)(R / RAYS);
  a.tasks_per_view = ceil_div(a.groups * P.n_chunks, WAVES) * WAVES;
  a.chunk = ceil_div(G.n_samples, P.n_chunks);
  a.t_end = (float)(1.0 - 1.0 / (double)G.n_samples);
  a.t_step = G.n_samples > 1 ? a.t_end / (float)(G.n_samples - 1) : 0.f;
  float* s = P.scratch;
  auto take = [&](int64_t n) { float* r = s; s += align4(n); return r; };
  a.tables = take(10 * H);
  a.dFt = take(B * R * H);
  a.sdf = take(B * Pn);
  a.crgb = take(3 * B * Pn);
  if ((P.fwd_sdf != nullptr) != (P.fwd_crgb != nullptr)) return CIPS3D_E_BADARG;
  if (P.fwd_sdf) {               // the differentiable forward filled the stash and these two
    a.sdf = const_cast<float*>(P.fwd_sdf);
    a.crgb = const_cast<float*>(P.fwd_crgb);
  }
  a.g = take(B * Pn);
  a.wts = take(B * Pn);
  a.Tb = take(B * Pn);
  a.dsdf = take(B * Pn);
  a.dcrgb = take(3 * B * Pn);
  a.dptsn = take(3 * B * Pn);
  a.dvd = take(3 * B * Pn);
  a.ddnorm = take(B * R);
  a.sums = take((int64_t)B * L * 2 * H);

  {
    const int tiles_r = (int)ceil_div<int64_t>(R, 32), tiles_h = ceil_div(H, 32), n_t = tiles_r * tiles_h * B, n_p = ceil_div(3 * H, 256);
    const int nz0 = B * L * 2 * H, nz1 = 12 * B, n_z = ceil_div(nz0 + nz1, 1024);
    hipLaunchKernelGGL(nerf_bwd_prep_kernel, dim3((unsigned)(n_t + n_p + n_z)), dim3(256), 0, st, P.d_features, a.dFt, H, (int)R, tiles_r,
                       tiles_h, n_t, P.w_first, P.w_view, P.w_sigma, P.w_rgb, P.packed, D, a.tables, n_p, a.sums, nz0, P.dcam, nz1);
    if (int rc = cips3d_launch_status()) return rc;
  }
  int rc;
  switch (H) {
    case 32: rc = launch_fused<2, 2>(a, st); break;
    case 64: rc = launch_fused<4, 2>(a, st); break;
    case 128: rc = launch_fused<8, 2>(a, st); break;
    default: {
      rc = launch_fused<16, 4>(a, st);
      if (rc == CIPS3D_E_UNSUPP) rc = launch_fused<16, 2>(a, st);
    }
  }
  if (rc) return rc;
  hipLaunchKernelGGL(finalize_film_kernel, dim3(ceil_div(B * L * H, 256)), dim3(256), 0, st, a.sums, P.layer_bias, P.packed, B,
                     L, H, P.dfilm);
  if ((rc = cips3d_launch_status())) return rc;
  return cips3d_nerf_bwd_camera_acc(&G, a.dptsn, a.dvd, a.ddnorm, P.dcam, st);      // (dcam: zeroed by the preparation launch)
}
