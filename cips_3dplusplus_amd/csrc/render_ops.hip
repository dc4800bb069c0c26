// Stand-alone forms of the small renderer steps that the fused kernel (nerf.hip) does in registers, so that every
// function of the reference's `Render` class (cips3d/nerf_utils.py:11-338) has a callable HIP counterpart:
//   cips3d_rays_in_world        Render.get_rays_in_world      (:18-66)
//   cips3d_z_vals               Render.get_z_vals             (:69-121, offset-sampling branch)
//   cips3d_ray_points           Render.get_points + Render.normalize_points (:124-170)
//   cips3d_volume_integration   Render.volume_integration     (:231-338, with_sdf branch)
// All HBM-bound; the first three move a few MB per view, the last one reads the (B,R,N,C) feature tensor once
// (101 MB per view at C=256, N=24) with one wave per ray: lanes cover 4 channels each (16-byte loads, 1 KB contiguous
// per sample) and the N compositing weights are computed once per ray and kept in LDS.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoidf_exact(float x) { return 1.f / (1.f + expf(-x)); }

// thread per ray; outputs [B,S,S,3] (reference layout, ray-major, xyz fastest)
__global__ void __launch_bounds__(256) rays_kernel(const float* __restrict__ cam_poses, const float* __restrict__ focals,
                                                   int S, int static_viewdirs, int B, float* __restrict__ rays_o,
                                                   float* __restrict__ rays_d, float* __restrict__ viewdirs) {
  const int R = S * S;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R) return;
  const int b = (int)(i / R), ray = (int)(i % R);
  const float focal = focals[b];
  const float* cw = cam_poses + 12 * b;
  const int pi = ray / S, pj = ray - pi * S;
  const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
  const float dcx = (px - (float)S * 0.5f) / focal, dcy = -(py - (float)S * 0.5f) / focal, dcz = -1.f;
  const float dx = (dcx * cw[0] + dcy * cw[1]) + dcz * cw[2];
  const float dy = (dcx * cw[4] + dcy * cw[5]) + dcz * cw[6];
  const float dz = (dcx * cw[8] + dcy * cw[9]) + dcz * cw[10];
  float vx = static_viewdirs ? dcx : dx, vy = static_viewdirs ? dcy : dy, vz = static_viewdirs ? dcz : dz;
  const float n = fmaxf(sqrtf((vx * vx + vy * vy) + vz * vz), 1e-12f);
  vx /= n; vy /= n; vz /= n;
  float* o = rays_o + i * 3; o[0] = cw[3]; o[1] = cw[7]; o[2] = cw[11];
  float* d = rays_d + i * 3; d[0] = dx; d[1] = dy; d[2] = dz;
  float* v = viewdirs + i * 3; v[0] = vx; v[1] = vy; v[2] = vz;
}

// thread per (ray, sample); z [B,R,N]
__global__ void __launch_bounds__(256) z_vals_kernel(const float* __restrict__ near_, const float* __restrict__ far_,
                                                     const float* __restrict__ u, int B, int R, int N,
                                                     float* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R * N) return;
  const int k = (int)(i % N);
  const int64_t br = i / N;
  const int b = (int)(br / R);
  const float nearv = near_[b], farv = far_[b];
  // torch.linspace(0, 1 - 1/N, N): symmetric evaluation around the midpoint (same as nerf.hip)
  const float t_end = (float)(1.0 - 1.0 / (double)N);
  const float t_step = N > 1 ? t_end / (float)(N - 1) : 0.f;
  auto zbase = [&](int kk) -> float {
    if (kk >= N) return farv;
    const float t = (kk < N / 2) ? t_step * (float)kk : t_end - t_step * (float)(N - 1 - kk);
    return nearv * (1.f - t) + farv * t;
  };
  const float z0 = zbase(k);
  z[i] = u ? z0 + (zbase(k + 1) - z0) * u[br] : z0;
}

// the classic NeRF stratified branch (offset_sampling = False, nerf_utils.py:98-117): t = linspace(0, 1, N) (symmetric
// evaluation around the midpoint, as torch), one uniform PER SAMPLE u [B,R,N] inside its own interval between the midpoints
__global__ void __launch_bounds__(256) z_vals_stratified_kernel(const float* __restrict__ near_, const float* __restrict__ far_,
                                                                const float* __restrict__ u, int B, int R, int N,
                                                                float* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R * N) return;
  const int k = (int)(i % N);
  const int b = (int)(i / N / R);
  const float nearv = near_[b], farv = far_[b];
  const float t_step = N > 1 ? 1.f / (float)(N - 1) : 0.f;
  auto zbase = [&](int kk) -> float {
    const float t = (kk < N / 2 || N == 1) ? t_step * (float)kk : 1.f - t_step * (float)(N - 1 - kk);   // linspace(0, 1, 1) = [0]
    return nearv * (1.f - t) + farv * t;
  };
  const float z0 = zbase(k);
  if (!u) { z[i] = z0; return; }
  const float lower = k > 0 ? 0.5f * (z0 + zbase(k - 1)) : z0;
  const float upper = k < N - 1 ? 0.5f * (zbase(k + 1) + z0) : z0;
  z[i] = lower + (upper - lower) * u[i];
}

// thread per point; pts / pts_n [B,R,N,3]
__global__ void __launch_bounds__(256) points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ z, const float* __restrict__ near_,
                                                     const float* __restrict__ far_, int B, int R, int N,
                                                     float* __restrict__ pts, float* __restrict__ pts_n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R * N) return;
  const int64_t br = i / N;
  const int b = (int)(br / R);
  const float zz = z[i];
  const float x = rays_o[br * 3] + rays_d[br * 3] * zz, y = rays_o[br * 3 + 1] + rays_d[br * 3 + 1] * zz,
              w = rays_o[br * 3 + 2] + rays_d[br * 3 + 2] * zz;
  if (pts) { pts[i * 3] = x; pts[i * 3 + 1] = y; pts[i * 3 + 2] = w; }
  if (pts_n) {
    const float span = far_[b] - near_[b];
    pts_n[i * 3] = x * 2.f / span; pts_n[i * 3 + 1] = y * 2.f / span; pts_n[i * 3 + 2] = w * 2.f / span;
  }
}

// one wave per ray (4 rays per block); N <= 256
__global__ void __launch_bounds__(256) integrate_kernel(const float* __restrict__ rgb, const float* __restrict__ sdf,
                                                        const float* __restrict__ feat, const float* __restrict__ z,
                                                        const float* __restrict__ rays_d, const float* __restrict__ pts,
                                                        const float* __restrict__ sigmoid_beta, int64_t BR, int N, int C,
                                                        int flags, float* __restrict__ rgb_map,
                                                        float* __restrict__ feature_map, float* __restrict__ xyz,
                                                        float* __restrict__ mask) {
  __shared__ float s_w[4][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wv;
  if (r >= BR) return;
  const bool raw_density = flags & CIPS3D_VI_RAW_DENSITY;
  const float beta = raw_density ? 1.f : sigmoid_beta[0];
  const float dnorm = sqrtf((rays_d[r * 3] * rays_d[r * 3] + rays_d[r * 3 + 1] * rays_d[r * 3 + 1]) +
                            rays_d[r * 3 + 2] * rays_d[r * 3 + 2]);
  // alpha_k for this lane's samples, then the transmittance scan by lane 0 (N is small: 24 .. 128)
  for (int k = lane; k < N; k += 64) {
    const float delta = (k < N - 1 ? z[r * N + k + 1] - z[r * N + k] : 1e10f) * dnorm;
    const float v = sdf[r * N + k];
    // with_sdf: density = sigmoid(-sdf / beta) / beta (nerf_utils.py:276-286); else F.softplus of the raw output (:288-297;
    // torch's threshold 20: identity above it)
    const float sigma = raw_density ? (v > 20.f ? v : log1pf(expf(v))) : sigmoidf_exact(-v / beta) / beta;
    s_w[wv][k] = 1.f - expf(-sigma * delta);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS writes are done
  if (lane == 0) {
    float T = 1.f, wsum = 0.f;
    for (int k = 0; k < N; ++k) {
      const float alpha = s_w[wv][k];
      const float w = alpha * T;
      s_w[wv][k] = w;
      if (k < N - 1) wsum += w;
      T *= (1.f - alpha) + 1e-10f;
    }
    if (flags & CIPS3D_VI_FORCE_BACKGROUND) s_w[wv][N - 1] = 1.f - wsum;   // nerf_utils.py:309-310
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  // rgb / xyz / mask: lanes 0..2 take one colour and one coordinate each
  {
    float cr = 0.f, cx = 0.f;
    if (lane < 3) {
      for (int k = 0; k < N; ++k) {
        const float w = s_w[wv][k];
        cr = fmaf(w, sigmoidf_exact(rgb[(r * N + k) * 3 + lane]), cr);
        cx = fmaf(w, pts[(r * N + k) * 3 + lane], cx);
      }
      rgb_map[r * 3 + lane] = -1.f + 2.f * cr;
      xyz[r * 3 + lane] = cx;
    }
    const float x0 = __shfl(cx, 0, 64), x1 = __shfl(cx, 1, 64), x2 = __shfl(cx, 2, 64);
    if (lane == 0) {
      mask[r * 2] = s_w[wv][N - 1];                         // background probability = last weight
      mask[r * 2 + 1] = -sqrtf((x0 * x0 + x1 * x1) + x2 * x2);   // "depth" = -|xyz|
    }
  }
  if (feat) {
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
      if (c0 + 3 < C) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < N; ++k) {
          const float w = s_w[wv][k];
          const float4 f = *reinterpret_cast<const float4*>(feat + (r * N + k) * C + c0);
          acc.x = fmaf(w, f.x, acc.x); acc.y = fmaf(w, f.y, acc.y); acc.z = fmaf(w, f.z, acc.z); acc.w = fmaf(w, f.w, acc.w);
        }
        *reinterpret_cast<float4*>(feature_map + r * C + c0) = acc;
      } else {
        for (int c = c0; c < C; ++c) {
          float acc = 0.f;
          for (int k = 0; k < N; ++k) acc = fmaf(s_w[wv][k], feat[(r * N + k) * C + c], acc);
          feature_map[r * C + c] = acc;
        }
      }
    }
  }
}

}  // namespace

extern "C" int cips3d_rays_in_world(const float* cam_poses, const float* focals, int img_size, int static_viewdirs, int B,
                                    float* rays_o, float* rays_d, float* viewdirs, void* stream) {
  if (!cam_poses || !focals || !rays_o || !rays_d || !viewdirs || B < 0 || img_size <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  const int64_t n = (int64_t)B * img_size * img_size;
  hipLaunchKernelGGL(rays_kernel, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0, as_stream(stream), cam_poses, focals,
                     img_size, static_viewdirs, B, rays_o, rays_d, viewdirs);
  return cips3d_launch_status();
}

extern "C" int cips3d_z_vals(const float* near_, const float* far_, const float* perturb_u, int B, int R, int N, float* z,
                             void* stream) {
  if (!near_ || !far_ || !z || B < 0 || R <= 0 || N <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(z_vals_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)B * R * N, 256)), dim3(256), 0, as_stream(stream),
                     near_, far_, perturb_u, B, R, N, z);
  return cips3d_launch_status();
}

extern "C" int cips3d_z_vals_stratified(const float* near_, const float* far_, const float* perturb_t, int B, int R, int N,
                                        float* z, void* stream) {
  if (!near_ || !far_ || !z || B < 0 || R <= 0 || N <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(z_vals_stratified_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)B * R * N, 256)), dim3(256), 0,
                     as_stream(stream), near_, far_, perturb_t, B, R, N, z);
  return cips3d_launch_status();
}

extern "C" int cips3d_ray_points(const float* rays_o, const float* rays_d, const float* z, const float* near_,
                                 const float* far_, int B, int R, int N, float* pts, float* pts_normalized, void* stream) {
  if (!rays_o || !rays_d || !z || B < 0 || R <= 0 || N <= 0 || (!pts && !pts_normalized)) return CIPS3D_E_BADARG;
  if (pts_normalized && (!near_ || !far_)) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(points_kernel, dim3((unsigned)ceil_div<int64_t>((int64_t)B * R * N, 256)), dim3(256), 0, as_stream(stream),
                     rays_o, rays_d, z, near_, far_, B, R, N, pts, pts_normalized);
  return cips3d_launch_status();
}

extern "C" int cips3d_volume_integration(const float* rgb, const float* sdf, const float* features, const float* z_vals,
                                         const float* rays_d, const float* pts, const float* sigmoid_beta, int64_t n_rays,
                                         int N, int C, int flags, float* rgb_map, float* feature_map, float* xyz,
                                         float* mask, void* stream) {
  if (!rgb || !sdf || !z_vals || !rays_d || !pts || !rgb_map || !xyz || !mask || n_rays < 0 || N <= 0) return CIPS3D_E_BADARG;
  if (!(flags & CIPS3D_VI_RAW_DENSITY) && !sigmoid_beta) return CIPS3D_E_BADARG;
  if (flags & ~(CIPS3D_VI_RAW_DENSITY | CIPS3D_VI_FORCE_BACKGROUND)) return CIPS3D_E_BADARG;
  if (features && (!feature_map || C <= 0)) return CIPS3D_E_BADARG;
  if (N > 256) return CIPS3D_E_UNSUPP;
  if (features && ((C % 4) || (reinterpret_cast<uintptr_t>(features) & 15) || (reinterpret_cast<uintptr_t>(feature_map) & 15)))
    return CIPS3D_E_UNSUPP;
  if (n_rays == 0) return 0;
  hipLaunchKernelGGL(integrate_kernel, dim3((unsigned)ceil_div<int64_t>(n_rays, 4)), dim3(256), 0, as_stream(stream), rgb, sdf,
                     features, z_vals, rays_d, pts, sigmoid_beta, n_rays, N, C, flags, rgb_map, feature_map, xyz, mask);
  return cips3d_launch_status();
}

// ------------------------------------------------------------------------------------------------
// cips3d_points_linear: a dense layer over a point-major tensor, y[p][o] = epilogue(sum_i W[o][i] x[p][i] + bias[o]),
// for the per-point module forwards of the reference (LinearLayer / FiLMSiren, cips3d/volume_renderer.py:15-85) when
// they are called directly on (b, ..., n, c) tensors.  Both operands are k-contiguous, so a float4 per lane is four
// k-slices of v_mfma_f32_16x16x4_f32 straight from global memory (same permutation of k on both sides); a wave owns
// 16 points x 64 outputs.  Not on the generator's hot path (that is the fused kernel), so untuned: W is re-read from
// L2 by every wave.
//   mode 0: y = out_scale * (acc + bias) + out_shift                       (LinearLayer)
//   mode 1: y = sin(gamma[b][o] * (acc + bias) + beta[b][o])               (FiLMSiren; film = [B][2][out])
// ------------------------------------------------------------------------------------------------
namespace {

typedef float pl_f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) points_linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ film,
                                                            int64_t P_total, int64_t P_per_batch, int in_dim, int out_dim,
                                                            int mode, float out_scale, float out_shift,
                                                            float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
  const int o0 = blockIdx.y * 64;
  if (p0 >= P_total) return;
  const int64_t prow = p0 + r < P_total ? p0 + r : P_total - 1;
  const float* xr = x + prow * in_dim;
  const bool vec = (in_dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  pl_f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = pl_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < in_dim; k0 += 16) {
    const int kk = k0 + 4 * q;
    pl_f32x4 a = {0.f, 0.f, 0.f, 0.f}, bw[4];
    if (vec) {
      if (kk < in_dim) a = *reinterpret_cast<const pl_f32x4*>(xr + kk);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (kk + j < in_dim) a[j] = xr[kk + j];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bw[t] = pl_f32x4{0.f, 0.f, 0.f, 0.f};
      const int o = o0 + 16 * t + r;
      if (o < out_dim) {
        const float* wr = W + (int64_t)o * in_dim;
        if (vec) {
          if (kk < in_dim) bw[t] = *reinterpret_cast<const pl_f32x4*>(wr + kk);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (kk + j < in_dim) bw[t][j] = wr[kk + j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bw[t][j], acc[t], 0, 0, 0);
  }
  // D layout: register e of lane (r, q) = C[point 4q + e][output r]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int o = o0 + 16 * t + r;
    if (o >= out_dim) continue;
    const float bo = bias ? bias[o] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t p = p0 + 4 * q + e;
      if (p >= P_total) continue;
      float v = acc[t][e] + bo;
      if (mode == 1) {
        const float* fb = film + (p / P_per_batch) * 2 * out_dim;
        v = sin_accurate(fmaf(fb[o], v, fb[out_dim + o]));
      } else {
        v = fmaf(v, out_scale, out_shift);
      }
      y[p * out_dim + o] = v;
    }
  }
}

}  // namespace

extern "C" int cips3d_points_linear(const float* x, const float* W, const float* bias, const float* film, int64_t n_points,
                                    int64_t points_per_batch, int in_dim, int out_dim, int mode, float out_scale,
                                    float out_shift, float* y, void* stream) {
  if (!x || !W || !y || n_points < 0 || in_dim <= 0 || out_dim <= 0 || points_per_batch <= 0) return CIPS3D_E_BADARG;
  if (mode != 0 && mode != 1) return CIPS3D_E_BADARG;
  if (mode == 1 && !film) return CIPS3D_E_BADARG;
  if (n_points == 0) return 0;
  hipLaunchKernelGGL(points_linear_kernel, dim3((unsigned)ceil_div<int64_t>(n_points, 64), (unsigned)ceil_div(out_dim, 64)),
                     dim3(256), 0, as_stream(stream), x, W, bias, film, n_points, points_per_batch, in_dim, out_dim, mode,
                     out_scale, out_shift, y);
  return cips3d_launch_status();
}
