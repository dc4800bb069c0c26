// Backward of the NeRF half of the path: d(loss)/d(FiLM gamma, beta) and d(loss)/d(camera pose) from the gradients of the
// feature map and the thumbnail (flip inversion: reference models/projector_v10.py:211-277 optimises the camera angles and
// the NeRF W+ styles through Render.prepare_nerf_inputs -> SirenGenerator -> volume_integration,
// cips3d/nerf_utils.py:18-338, cips3d/volume_renderer.py:39-160, with PyTorch autograd).
//
// First version, materialised: the forward fused kernel (nerf.hip) keeps no activations, so the backward recomputes the
// point MLP layer by layer into HBM in a channel-major layout  act[b][c][p],  p = sample * R + ray  -- the layout of the
// decoder activations, so the hidden GEMMs and their data gradients are cips3d_modconv1x1 (on the packed weights / packed
// transposes).  The kernels here are the element-wise and reduction steps around those GEMMs; all of them are HBM-bound
// (one read + one write of a [B,H,P] array each; ~100 MB per view at H=256, N=24).
//   cips3d_nerf_bwd_points     rays -> normalised points, layer-0 pre-activation and sine
//   cips3d_nerf_bwd_film       pre = acc + bias (+ Wd . viewdir);  h = sin(gamma pre + beta)
//   cips3d_nerf_bwd_heads      rows of a [nr,H] matrix applied over the channels (sigma / rgb heads; W0^T, Wd^T in backward)
//   cips3d_nerf_bwd_dot        g[p] = <dF[:, ray], f[:, p]>
//   cips3d_nerf_bwd_composite  volume integration forward + backward per ray -> w, d(sdf), d(rgb logits)
//   cips3d_nerf_bwd_film_grad  d(pre) = upstream * cos(gamma pre + beta) * gamma, sums for d(gamma), d(beta)
//   cips3d_nerf_bwd_camera     d(points), d(viewdirs) -> d(cam_poses)
//   cips3d_camera_params_bwd   d(cam_poses) -> d(azim, elev)     (nerf_utils.py:344-436, forward-mode duals)
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

struct RayGeom {
  float ox, oy, oz, dx, dy, dz, dcx, dcy, dcz, vx, vy, vz, vnorm, dnorm, nearv, farv, u, t_end, t_step;
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

// identical arithmetic to the ray setup of nerf_render_kernel (nerf.hip; nerf_utils.py:38-121)
__device__ __forceinline__ RayGeom ray_geom(const cips3d_nerf_bwd_geom& G, int b, int ray) {
  RayGeom r;
  const int S = G.img_size;
  const float focal = G.focals[b];
  r.nearv = G.near_[b]; r.farv = G.far_[b];
  const float* cw = G.cam_poses + 12 * b;
  const int pi = ray / S, pj = ray - pi * S;
  const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
  r.dcx = (px - (float)S * 0.5f) / focal;
  r.dcy = -(py - (float)S * 0.5f) / focal;
  r.dcz = -1.f;
  r.dx = (r.dcx * cw[0] + r.dcy * cw[1]) + r.dcz * cw[2];
  r.dy = (r.dcx * cw[4] + r.dcy * cw[5]) + r.dcz * cw[6];
  r.dz = (r.dcx * cw[8] + r.dcy * cw[9]) + r.dcz * cw[10];
  r.ox = cw[3]; r.oy = cw[7]; r.oz = cw[11];
  const float rx = G.static_viewdirs ? r.dcx : r.dx, ry = G.static_viewdirs ? r.dcy : r.dy,
              rz = G.static_viewdirs ? r.dcz : r.dz;
  r.vnorm = fmaxf(sqrtf((rx * rx + ry * ry) + rz * rz), 1e-12f);
  r.vx = rx / r.vnorm; r.vy = ry / r.vnorm; r.vz = rz / r.vnorm;
  r.dnorm = sqrtf((r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz);
  r.has_u = G.perturb_u != nullptr;
  r.u = r.has_u ? G.perturb_u[(int64_t)b * S * S + ray] : 0.f;
  r.N = G.n_samples;
  r.t_end = (float)(1.0 - 1.0 / (double)r.N);
  r.t_step = r.N > 1 ? r.t_end / (float)(r.N - 1) : 0.f;
  return r;
}

__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------- points + layer 0
// grid (ceil(P/256), B).  ptsn [B,3,P]; pre0, h0 [B,H,P]; viewdirs [B,3,R].
__global__ void __launch_bounds__(256) points_kernel(cips3d_nerf_bwd_geom G, const float* __restrict__ w_first,
                                                     const float* __restrict__ bias0, const float* __restrict__ film,
                                                     int film_bstride, int H, float* __restrict__ ptsn,
                                                     float* __restrict__ pre0, float* __restrict__ h0,
                                                     float* __restrict__ viewdirs) {
  const int b = blockIdx.y;
  const int R = G.img_size * G.img_size;
  const int64_t P = (int64_t)R * G.n_samples;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int k = (int)(p / R), ray = (int)(p - (int64_t)k * R);
  const RayGeom r = ray_geom(G, b, ray);
  const float z = r.z(k);
  const float span = r.farv - r.nearv;
  const float nx = (r.ox + r.dx * z) * 2.f / span, ny = (r.oy + r.dy * z) * 2.f / span, nz = (r.oz + r.dz * z) * 2.f / span;
  float* pn = ptsn + (int64_t)b * 3 * P + p;
  pn[0] = nx; pn[P] = ny; pn[2 * P] = nz;
  if (k == 0) {
    float* vd = viewdirs + (int64_t)b * 3 * R + ray;
    vd[0] = r.vx; vd[R] = r.vy; vd[2 * R] = r.vz;
  }
  const float* gm = film + (int64_t)b * film_bstride;       // layer 0: gamma [H], beta [H]
  for (int c = 0; c < H; ++c) {
    const float pre = fmaf(w_first[c * 3 + 2], nz, fmaf(w_first[c * 3 + 1], ny, w_first[c * 3] * nx)) + bias0[c];
    const int64_t o = ((int64_t)b * H + c) * P + p;
    pre0[o] = pre;
    h0[o] = sin_accurate(fmaf(gm[c], pre, gm[H + c]));
  }
}

// ---------------------------------------------------------------------------------------------- FiLM sine forward
// grid (ceil(P/1024), H, B): pre = acc + bias_c (+ wd_c . viewdir(ray)) written over acc; h = sin(gamma_c pre + beta_c).
__global__ void __launch_bounds__(256) film_kernel(float* __restrict__ acc, float* __restrict__ h,
                                                   const float* __restrict__ bias, const float* __restrict__ film,
                                                   int film_bstride, const float* __restrict__ wd, int wd_stride,
                                                   const float* __restrict__ viewdirs, int H, int R, int64_t P) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P) return;
  const float gm = film[(int64_t)b * film_bstride + c], bt = film[(int64_t)b * film_bstride + H + c];
  const int64_t o = ((int64_t)b * H + c) * P + p0;
  float4 a = *reinterpret_cast<const float4*>(acc + o);
  const float bc = bias[c];
  a.x += bc; a.y += bc; a.z += bc; a.w += bc;
  if (wd) {
    const int ray = (int)(p0 % R);
    const float* vd = viewdirs + (int64_t)b * 3 * R + ray;
    const float4 vx = *reinterpret_cast<const float4*>(vd), vy = *reinterpret_cast<const float4*>(vd + R),
                 vz = *reinterpret_cast<const float4*>(vd + 2 * R);
    const float w0 = wd[c * wd_stride], w1 = wd[c * wd_stride + 1], w2 = wd[c * wd_stride + 2];
    a.x += fmaf(w2, vz.x, fmaf(w1, vy.x, w0 * vx.x));
    a.y += fmaf(w2, vz.y, fmaf(w1, vy.y, w0 * vx.y));
    a.z += fmaf(w2, vz.z, fmaf(w1, vy.z, w0 * vx.z));
    a.w += fmaf(w2, vz.w, fmaf(w1, vy.w, w0 * vx.w));
  }
  *reinterpret_cast<float4*>(acc + o) = a;
  float4 s;
  s.x = sin_accurate(fmaf(gm, a.x, bt)); s.y = sin_accurate(fmaf(gm, a.y, bt));
  s.z = sin_accurate(fmaf(gm, a.z, bt)); s.w = sin_accurate(fmaf(gm, a.w, bt));
  *reinterpret_cast<float4*>(h + o) = s;
}

// ---------------------------------------------------------------------------------------------- channel heads
// out[b][r][p] = sum_c Wm[r*rs + c*cs] x[b][c][p] + bias[r]   (nr <= 4).  grid (ceil(P/1024), B); a thread owns 4 consecutive
// points (16-byte loads, 8 channels in flight)
__global__ void __launch_bounds__(256) heads_kernel(const float* __restrict__ x, const float* __restrict__ Wm, int rs, int cs,
                                                    const float* __restrict__ bias, int nr, int H, int64_t P,
                                                    float* __restrict__ out) {
  const int b = blockIdx.y;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= P) return;                                   // P % 4 == 0
  float4 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xp = x + (int64_t)b * H * P + p;
  for (int c0 = 0; c0 < H; c0 += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = c0 + u < H ? *reinterpret_cast<const float4*>(xp + (int64_t)(c0 + u) * P) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (c0 + u >= H) break;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (r < nr) {
          const float w = Wm[r * rs + (c0 + u) * cs];
          acc[r].x = fmaf(w, v[u].x, acc[r].x); acc[r].y = fmaf(w, v[u].y, acc[r].y);
          acc[r].z = fmaf(w, v[u].z, acc[r].z); acc[r].w = fmaf(w, v[u].w, acc[r].w);
        }
    }
  }
  for (int r = 0; r < nr; ++r) {
    const float bs = bias ? bias[r] : 0.f;
    *reinterpret_cast<float4*>(out + ((int64_t)b * nr + r) * P + p) =
        make_float4(acc[r].x + bs, acc[r].y + bs, acc[r].z + bs, acc[r].w + bs);
  }
}

// g[b][p] = sum_c dF[b][c][ray] f[b][c][p];  4 consecutive points (= 4 consecutive rays of one sample) per thread
__global__ void __launch_bounds__(256) dot_kernel(const float* __restrict__ dF, const float* __restrict__ f, int H, int R,
                                                  int64_t P, float* __restrict__ g) {
  const int b = blockIdx.y;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= P) return;
  const int ray = (int)(p % R);                       // R % 4 == 0: the four points share the sample
  const float* fp = f + (int64_t)b * H * P + p;
  const float* dp = dF + (int64_t)b * H * R + ray;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c0 = 0; c0 < H; c0 += 4) {
    float4 a[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = c0 + u < H;
      a[u] = ok ? *reinterpret_cast<const float4*>(fp + (int64_t)(c0 + u) * P) : make_float4(0.f, 0.f, 0.f, 0.f);
      d[u] = ok ? *reinterpret_cast<const float4*>(dp + (int64_t)(c0 + u) * R) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc.x = fmaf(d[u].x, a[u].x, acc.x); acc.y = fmaf(d[u].y, a[u].y, acc.y);
      acc.z = fmaf(d[u].z, a[u].z, acc.z); acc.w = fmaf(d[u].w, a[u].w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(g + (int64_t)b * P + p) = acc;
}

// ---------------------------------------------------------------------------------------------- compositing fwd + bwd
// One thread per ray (nerf_utils.py:264-307).  Writes w, T (scratch), then d(sdf) [B,P] and d(rgb logits) [B,3,P].
__global__ void __launch_bounds__(256) composite_kernel(cips3d_nerf_bwd_geom G, const float* __restrict__ sdf,
                                                        const float* __restrict__ crgb, const float* __restrict__ g,
                                                        const float* __restrict__ dthumb,
                                                        const float* __restrict__ sigmoid_beta, float* __restrict__ w,
                                                        float* __restrict__ Tbuf, float* __restrict__ dsdf,
                                                        float* __restrict__ dcrgb, float* __restrict__ ddnorm,
                                                        float* __restrict__ dbeta_ray) {
  const int b = blockIdx.y;
  const int R = G.img_size * G.img_size;
  const int ray = blockIdx.x * 256 + threadIdx.x;
  if (ray >= R) return;
  const int N = G.n_samples;
  const int64_t P = (int64_t)R * N;
  const RayGeom r = ray_geom(G, b, ray);
  // sigmoid_beta == NULL: with_sdf = False -- `sdf` holds the raw density v, sigma = softplus(v), d sigma / d v = sigmoid(v)
  // (nerf_utils.py:288-297)
  const bool raw = sigmoid_beta == nullptr;
  const float beta = raw ? 1.f : sigmoid_beta[0];
  const float* sp = sdf + (int64_t)b * P + ray;
  float* wp = w + (int64_t)b * P + ray;
  float* tp = Tbuf + (int64_t)b * P + ray;
  float T = 1.f;
  for (int k = 0; k < N; ++k) {
    const float delta = (k < N - 1 ? r.z(k + 1) - r.z(k) : 1e10f) * r.dnorm;
    const float v = sp[(int64_t)k * R];
    const float sigma = raw ? (v > 20.f ? v : log1pf(expf(v))) : sigmoid_acc(-v / beta) / beta;
    const float alpha = 1.f - expf(-sigma * delta);
    tp[(int64_t)k * R] = T;
    wp[(int64_t)k * R] = alpha * T;
    T *= (1.f - alpha) + 1e-10f;
  }
  const float d0 = dthumb[((int64_t)b * 3 + 0) * R + ray], d1 = dthumb[((int64_t)b * 3 + 1) * R + ray],
              d2 = dthumb[((int64_t)b * 3 + 2) * R + ray];
  const float* cp = crgb + (int64_t)b * 3 * P + ray;
  float* dcp = dcrgb + (int64_t)b * 3 * P + ray;
  float S = 0.f;      // sum_{j>k} w_j G_j
  float dn = 0.f;     // d loss / d |rays_d|  (delta_k = dz_k * |rays_d|)
  float dbt = 0.f;    // d loss / d sigmoid_beta of this ray: sigma = s / beta, s = sigmoid(-sdf / beta)
  for (int k = N - 1; k >= 0; --k) {
    const int64_t o = (int64_t)k * R;
    const float dz = (k < N - 1 ? r.z(k + 1) - r.z(k) : 1e10f);
    const float delta = dz * r.dnorm;
    const float sg = raw ? sigmoid_acc(sp[o]) : sigmoid_acc(-sp[o] / beta);
    const float sigma = raw ? (sp[o] > 20.f ? sp[o] : log1pf(expf(sp[o]))) : sg / beta;
    const float e = expf(-sigma * delta);
    const float alpha = 1.f - e;
    const float Tk = tp[o], wk = wp[o];
    const float s0 = sigmoid_acc(cp[o]), s1 = sigmoid_acc(cp[P + o]), s2 = sigmoid_acc(cp[2 * P + o]);
    const float Gk = g[(int64_t)b * P + ray + o] + 2.f * (d0 * s0 + d1 * s1 + d2 * s2);
    const float dalpha = Tk * Gk - S / ((1.f - alpha) + 1e-10f);
    S = fmaf(wk, Gk, S);
    const float dsigma = dalpha * delta * e;
    dn = fmaf(dalpha * sigma * e, dz, dn);
    dsdf[(int64_t)b * P + ray + o] = raw ? dsigma * sg : dsigma * (-sg * (1.f - sg) / (beta * beta));
    if (!raw) dbt = fmaf(dsigma, (sg * (1.f - sg) * sp[o] / beta - sg) / (beta * beta), dbt);
    dcp[o] = 2.f * wk * d0 * s0 * (1.f - s0);
    dcp[P + o] = 2.f * wk * d1 * s1 * (1.f - s1);
    dcp[2 * P + o] = 2.f * wk * d2 * s2 * (1.f - s2);
  }
  ddnorm[(int64_t)b * R + ray] = dn;
  if (dbeta_ray) dbeta_ray[(int64_t)b * R + ray] = dbt;
}

// The same integration with one SAMPLE per thread: 32 rays x N samples per workgroup (N <= 32).  A thread per ray leaves the chip
// to R*B / 64 waves (128 at 64^2, B = 2: 87 us for 0.4 M points); here the transcendental work and the loads of all samples run
// side by side and only the products along the ray -- T_k, the suffix sum S_k, the per-ray sums -- are chained, each thread walking
// the shared per-sample factors in the per-ray loop's own order (the results are the ones composite_kernel produces).
// dynamic LDS: 7 * N * 32 floats.
__global__ void __launch_bounds__(1024) composite_par_kernel(cips3d_nerf_bwd_geom G, const float* __restrict__ sdf,
                                                             const float* __restrict__ crgb, const float* __restrict__ g,
                                                             const float* __restrict__ dthumb,
                                                             const float* __restrict__ sigmoid_beta, float* __restrict__ w,
                                                             float* __restrict__ Tbuf, float* __restrict__ dsdf,
                                                             float* __restrict__ dcrgb, float* __restrict__ ddnorm,
                                                             float* __restrict__ dbeta_ray) {
  extern __shared__ float comp_lds[];
  const int b = blockIdx.y;
  const int R = G.img_size * G.img_size;
  const int N = G.n_samples;
  const int lr = threadIdx.x & 31, k = threadIdx.x >> 5;
  const int ray_u = blockIdx.x * 32 + lr;
  const bool live = ray_u < R;
  const int ray = live ? ray_u : R - 1;
  const int64_t P = (int64_t)R * N;
  float* F = comp_lds;                 // (1 - alpha_k) + 1e-10
  float* Wl = F + N * 32;              // w_k
  float* Gl = Wl + N * 32;             // G_k
  float* Al = Gl + N * 32;             // d alpha_k * sigma_k * e_k
  float* Zl = Al + N * 32;             // dz_k
  float* Dl = Zl + N * 32;             // d sigma_k
  float* Cl = Dl + N * 32;             // d sigma_k / d beta factor
  const int li = k * 32 + lr;
  const RayGeom r = ray_geom(G, b, ray);
  const bool raw = sigmoid_beta == nullptr;
  const float beta = raw ? 1.f : sigmoid_beta[0];
  const int64_t o = (int64_t)k * R + ray;
  const float dz = (k < N - 1 ? r.z(k + 1) - r.z(k) : 1e10f);
  const float delta = dz * r.dnorm;
  const float v = sdf[(int64_t)b * P + o];
  const float sg = raw ? sigmoid_acc(v) : sigmoid_acc(-v / beta);
  const float sigma = raw ? (v > 20.f ? v : log1pf(expf(v))) : sg / beta;
  const float e = expf(-sigma * delta);
  const float alpha = 1.f - e;
  const float f = (1.f - alpha) + 1e-10f;
  F[li] = f;
  const float d0 = dthumb[((int64_t)b * 3 + 0) * R + ray], d1 = dthumb[((int64_t)b * 3 + 1) * R + ray],
              d2 = dthumb[((int64_t)b * 3 + 2) * R + ray];
  const float* cp = crgb + (int64_t)b * 3 * P + o;
  const float s0 = sigmoid_acc(cp[0]), s1 = sigmoid_acc(cp[P]), s2 = sigmoid_acc(cp[2 * P]);
  const float Gk = g[(int64_t)b * P + o] + 2.f * (d0 * s0 + d1 * s1 + d2 * s2);
  Gl[li] = Gk;
  __syncthreads();
  float T = 1.f;
  for (int j = 0; j < k; ++j) T *= F[j * 32 + lr];
  const float wk = alpha * T;
  Wl[li] = wk;
  if (live) {
    Tbuf[(int64_t)b * P + o] = T;
    w[(int64_t)b * P + o] = wk;
  }
  __syncthreads();
  float S = 0.f;      // sum_{j>k} w_j G_j
  for (int j = N - 1; j > k; --j) S = fmaf(Wl[j * 32 + lr], Gl[j * 32 + lr], S);
  const float dalpha = T * Gk - S / f;
  const float dsigma = dalpha * delta * e;
  Al[li] = dalpha * sigma * e;
  Zl[li] = dz;
  if (!raw) {
    Dl[li] = dsigma;
    Cl[li] = (sg * (1.f - sg) * v / beta - sg) / (beta * beta);
  }
  if (live) {
    dsdf[(int64_t)b * P + o] = raw ? dsigma * sg : dsigma * (-sg * (1.f - sg) / (beta * beta));
    float* dcp = dcrgb + (int64_t)b * 3 * P + o;
    dcp[0] = 2.f * wk * d0 * s0 * (1.f - s0);
    dcp[P] = 2.f * wk * d1 * s1 * (1.f - s1);
    dcp[2 * P] = 2.f * wk * d2 * s2 * (1.f - s2);
  }
  __syncthreads();
  if (!live) return;
  if (k == 0) {
    float dn = 0.f;     // d loss / d |rays_d|
    for (int j = N - 1; j >= 0; --j) dn = fmaf(Al[j * 32 + lr], Zl[j * 32 + lr], dn);
    ddnorm[(int64_t)b * R + ray] = dn;
  } else if (k == 1 && dbeta_ray) {
    float dbt = 0.f;
    if (!raw)
      for (int j = N - 1; j >= 0; --j) dbt = fmaf(Dl[j * 32 + lr], Cl[j * 32 + lr], dbt);
    dbeta_ray[(int64_t)b * R + ray] = dbt;
  }
}

// ---------------------------------------------------------------------------------------------- row dots (renderer-weight gradients)
// out[row][c] += sum_{b,p} a[b][row][p] * x[b][c][p mod Px]  (c < nx <= 3),   out[row][3] += sum_{b,p} a[b][row][p]
// The narrow weight gradients of the point MLP: the first layer (x = normalised points), the view-direction columns of the
// view layer (x = per-ray directions, Px = R), the rgb / sigma heads (rows = features, x = d logits) and every bias.
// grid (ceil(P / 1024), rows, B); one atomic per (row, column) and workgroup.
__global__ void __launch_bounds__(256) row_dots_kernel(const float* __restrict__ a, const float* __restrict__ x, int nx,
                                                       int64_t Px, float* __restrict__ out, int rows, int64_t P) {
  __shared__ float sh[4];
  const int row = blockIdx.y, b = blockIdx.z;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (p0 < P) {
    const float4 v = *reinterpret_cast<const float4*>(a + ((int64_t)b * rows + row) * P + p0);
    acc[3] = (v.x + v.y) + (v.z + v.w);
    for (int c = 0; c < nx; ++c) {
      const float4 t = *reinterpret_cast<const float4*>(x + ((int64_t)b * nx + c) * Px + p0 % Px);   // Px % 4 == 0
      acc[c] = fmaf(v.w, t.w, fmaf(v.z, t.z, fmaf(v.y, t.y, v.x * t.x)));
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float t = wave_sum(acc[c]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0 && (c == 3 || c < nx)) unsafeAtomicAdd(out + row * 4 + c, (sh[0] + sh[1]) + (sh[2] + sh[3]));
  }
}

// ---------------------------------------------------------------------------------------------- FiLM backward
// upstream u[c][p] =  mode 0: dh[c][p] (+ ws_c * dsdf[p])                          (hidden layers; dh in `buf`)
//                     mode 1: w[p] * dF[c][ray] + sum_r Wc[r][c] * dcrgb[r][p]      (view layer; `buf` holds f, overwritten)
// dpre = u * cos(gamma pre + beta) * gamma  -> buf;   dgamma_c += sum_p u cos pre,  dbeta_c += sum_p u cos.
// grid (ceil(P/1024), H, B)
__global__ void __launch_bounds__(256) film_grad_kernel(float* __restrict__ buf, const float* __restrict__ pre,
                                                        const float* __restrict__ film, int film_bstride, int mode,
                                                        const float* __restrict__ ws, const float* __restrict__ dsdf,
                                                        const float* __restrict__ wts, const float* __restrict__ dF,
                                                        const float* __restrict__ Wc, const float* __restrict__ dcrgb,
                                                        float* __restrict__ dfilm, int H, int R, int64_t P) {
  __shared__ float sh[4];
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float gm = film[(int64_t)b * film_bstride + c], bt = film[(int64_t)b * film_bstride + H + c];
  float sg = 0.f, sb = 0.f;
  if (p0 < P) {
    const int64_t o = ((int64_t)b * H + c) * P + p0;
    const float4 pr = *reinterpret_cast<const float4*>(pre + o);
    float u[4];
    if (mode == 0) {
      const float4 d = *reinterpret_cast<const float4*>(buf + o);
      u[0] = d.x; u[1] = d.y; u[2] = d.z; u[3] = d.w;
      if (ws) {
        const float4 ds = *reinterpret_cast<const float4*>(dsdf + (int64_t)b * P + p0);
        const float wsc = ws[c];
        u[0] = fmaf(wsc, ds.x, u[0]); u[1] = fmaf(wsc, ds.y, u[1]); u[2] = fmaf(wsc, ds.z, u[2]); u[3] = fmaf(wsc, ds.w, u[3]);
      }
    } else {
      const int ray = (int)(p0 % R);
      const float4 wv = *reinterpret_cast<const float4*>(wts + (int64_t)b * P + p0);
      const float4 df = *reinterpret_cast<const float4*>(dF + ((int64_t)b * H + c) * R + ray);
      const float* dc = dcrgb + (int64_t)b * 3 * P + p0;
      const float4 c0 = *reinterpret_cast<const float4*>(dc), c1 = *reinterpret_cast<const float4*>(dc + P),
                   c2 = *reinterpret_cast<const float4*>(dc + 2 * P);
      const float w0 = Wc[c], w1 = Wc[H + c], w2 = Wc[2 * H + c];
      u[0] = fmaf(wv.x, df.x, fmaf(w2, c2.x, fmaf(w1, c1.x, w0 * c0.x)));
      u[1] = fmaf(wv.y, df.y, fmaf(w2, c2.y, fmaf(w1, c1.y, w0 * c0.y)));
      u[2] = fmaf(wv.z, df.z, fmaf(w2, c2.z, fmaf(w1, c1.z, w0 * c0.z)));
      u[3] = fmaf(wv.w, df.w, fmaf(w2, c2.w, fmaf(w1, c1.w, w0 * c0.w)));
    }
    const float pv[4] = {pr.x, pr.y, pr.z, pr.w};
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float uc = u[i] * cosf(fmaf(gm, pv[i], bt));
      out[i] = uc * gm;
      sg = fmaf(uc, pv[i], sg);
      sb += uc;
    }
    *reinterpret_cast<float4*>(buf + o) = make_float4(out[0], out[1], out[2], out[3]);
  }
  sg = block_sum_256(sg, sh);
  sb = block_sum_256(sb, sh);
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(dfilm + (int64_t)b * film_bstride + c, sg);
    unsafeAtomicAdd(dfilm + (int64_t)b * film_bstride + H + c, sb);
  }
}

// ---------------------------------------------------------------------------------------------- camera chain
// dptsn [B,3,P], dvd_pt [B,3,P] (d loss / d viewdir, per point) -> dcam [B,3,4] (atomics; zeroed by the host call).
// pts_n = (o + d z) 2/span;  rays_d = Rm d_cam;  o = T;  viewdir = normalize(static ? d_cam : rays_d).
// grid (ceil(R / 64), B), 1024 threads: wave w of a workgroup owns a slice of the samples of the group's 64 rays (every term is
// linear in the per-sample sums) -- a thread per ray alone is R*B/64 waves on the whole chip; 12 atomics per workgroup.
__global__ void __launch_bounds__(1024) camera_chain_kernel(cips3d_nerf_bwd_geom G, const float* __restrict__ dptsn,
                                                            const float* __restrict__ dvd_pt,
                                                            const float* __restrict__ ddnorm, float* __restrict__ dcam) {
  __shared__ float sh[12][16];
  const int b = blockIdx.y;
  const int R = G.img_size * G.img_size;
  const int ray = blockIdx.x * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  const int N = G.n_samples;
  const int64_t P = (int64_t)R * N;
  float v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) v[i] = 0.f;
  const int per = (N + 15) / 16;
  const int k_lo = slice * per, k_hi = min(N, k_lo + per);
  if (ray < R && (k_lo < k_hi || slice == 0)) {
    const RayGeom r = ray_geom(G, b, ray);
    const float sc = 2.f / (r.farv - r.nearv);
    float dO[3] = {0.f, 0.f, 0.f}, dD[3] = {0.f, 0.f, 0.f}, dV[3] = {0.f, 0.f, 0.f};
    for (int k = k_lo; k < k_hi; ++k) {
      const float z = r.z(k);
      const int64_t o = (int64_t)b * 3 * P + (int64_t)k * R + ray;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float gp = dptsn[o + j * P] * sc;
        dO[j] += gp;
        dD[j] = fmaf(gp, z, dD[j]);
        dV[j] += dvd_pt[o + j * P];
      }
    }
    if (!G.static_viewdirs) {
      // v = raw / |raw|:  d raw = (dV - v <v, dV>) / |raw|
      const float dotv = (r.vx * dV[0] + r.vy * dV[1]) + r.vz * dV[2];
      dD[0] += (dV[0] - r.vx * dotv) / r.vnorm;
      dD[1] += (dV[1] - r.vy * dotv) / r.vnorm;
      dD[2] += (dV[2] - r.vz * dotv) / r.vnorm;
    }
    if (slice == 0) {   // |rays_d| enters the sample spacing of the compositing (nerf_utils.py:264-268)
      const float gn = ddnorm[(int64_t)b * R + ray] / r.dnorm;
      dD[0] = fmaf(gn, r.dx, dD[0]); dD[1] = fmaf(gn, r.dy, dD[1]); dD[2] = fmaf(gn, r.dz, dD[2]);
    }
    const float dc[3] = {r.dcx, r.dcy, r.dcz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) v[i * 4 + j] = dD[i] * dc[j];
      v[i * 4 + 3] = dO[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float s = wave_sum(v[i]);
    if ((threadIdx.x & 63) == 0) sh[i][slice] = s;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += sh[threadIdx.x][j];
    unsafeAtomicAdd(dcam + 12 * b + threadIdx.x, s);
  }
}

// ---------------------------------------------------------------------------------------------- camera pose backward
// forward-mode derivative of camera_kernel (camera.hip) w.r.t. (azim, elev), contracted with d(extrinsics)
struct Du { float v, a, e; };   // value, d/d azim, d/d elev
__device__ __forceinline__ Du operator+(Du x, Du y) { return Du{x.v + y.v, x.a + y.a, x.e + y.e}; }
__device__ __forceinline__ Du operator-(Du x, Du y) { return Du{x.v - y.v, x.a - y.a, x.e - y.e}; }
__device__ __forceinline__ Du operator*(Du x, Du y) { return Du{x.v * y.v, x.a * y.v + x.v * y.a, x.e * y.v + x.v * y.e}; }
__device__ __forceinline__ Du cst(float c) { return Du{c, 0.f, 0.f}; }
struct D3 { Du x, y, z; };
__device__ __forceinline__ D3 crossd(D3 a, D3 b) {
  return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ D3 unitd(D3 a, float eps) {
  const Du s = (a.x * a.x + a.y * a.y) + a.z * a.z;
  const float n = sqrtf(s.v);
  Du inv;
  if (n > eps) {
    const float i1 = 1.f / n, i3 = -0.5f / (n * s.v);     // d(1/sqrt(s)) = -1/2 s^(-3/2) ds
    inv = Du{i1, i3 * s.a, i3 * s.e};
  } else {
    inv = cst(1.f / eps);
  }
  return D3{a.x * inv, a.y * inv, a.z * inv};
}

__global__ void camera_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ up, const float* __restrict__ dextr,
                                  int B, float* __restrict__ dloc) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float azim = loc[2 * b], elev = loc[2 * b + 1];
  const float ca = cosf(azim), sa = sinf(azim), ce = cosf(elev), se = sinf(elev);
  const D3 dir{Du{ce * sa, ce * ca, -se * sa}, Du{se, 0.f, ce}, Du{ce * ca, -ce * sa, -se * ca}};
  const D3 upv = up ? D3{cst(up[3 * b]), cst(up[3 * b + 1]), cst(up[3 * b + 2])} : D3{cst(0.f), cst(1.f), cst(0.f)};
  const D3 zax = unitd(dir, 1e-5f);
  D3 xax = unitd(crossd(upv, zax), 1e-5f);
  const D3 yax = unitd(crossd(zax, xax), 1e-5f);
  if (fabsf(xax.x.v) <= 5e-3f && fabsf(xax.y.v) <= 5e-3f && fabsf(xax.z.v) <= 5e-3f) xax = unitd(crossd(yax, zax), 1e-5f);
  const Du e[12] = {xax.x, yax.x, zax.x, dir.x, xax.y, yax.y, zax.y, dir.y, xax.z, yax.z, zax.z, dir.z};
  float ga = 0.f, ge = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float g = dextr[12 * b + i];
    ga = fmaf(g, e[i].a, ga);
    ge = fmaf(g, e[i].e, ge);
  }
  dloc[2 * b] = ga;
  dloc[2 * b + 1] = ge;
}

bool geom_ok(const cips3d_nerf_bwd_geom* G) {
  return G && G->cam_poses && G->focals && G->near_ && G->far_ && G->B >= 0 && G->img_size > 0 && G->n_samples > 0 &&
         (G->img_size * G->img_size) % 4 == 0;
}

}  // namespace

extern "C" int cips3d_nerf_bwd_points(const cips3d_nerf_bwd_geom* G, const float* w_first, const float* bias0,
                                      const float* film, int film_bstride, int H, float* ptsn, float* pre0, float* h0,
                                      float* viewdirs, void* stream) {
  if (!geom_ok(G) || !w_first || !bias0 || !film || !ptsn || !pre0 || !h0 || !viewdirs || H <= 0) return CIPS3D_E_BADARG;
  if (G->B == 0) return 0;
  const int64_t P = (int64_t)G->img_size * G->img_size * G->n_samples;
  hipLaunchKernelGGL(points_kernel, dim3((unsigned)ceil_div<int64_t>(P, 256), (unsigned)G->B), dim3(256), 0, as_stream(stream),
                     *G, w_first, bias0, film, film_bstride, H, ptsn, pre0, h0, viewdirs);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_film(float* acc, float* h, const float* bias, const float* film, int film_bstride,
                                    const float* wd, int wd_stride, const float* viewdirs, int B, int H, int R, int64_t P,
                                    void* stream) {
  if (!acc || !h || !bias || !film || B < 0 || H <= 0 || R <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (wd && !viewdirs) return CIPS3D_E_BADARG;
  if (P % 4 || R % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipLaunchKernelGGL(film_kernel, dim3((unsigned)ceil_div<int64_t>(P, 1024), (unsigned)H, (unsigned)B), dim3(256), 0,
                     as_stream(stream), acc, h, bias, film, film_bstride, wd, wd_stride, viewdirs, H, R, P);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_heads(const float* x, const float* Wm, int row_stride, int col_stride, const float* bias,
                                     int n_rows, int B, int H, int64_t P, float* out, void* stream) {
  if (!x || !Wm || !out || n_rows < 1 || n_rows > 4 || B < 0 || H <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  if (P % 4) return CIPS3D_E_UNSUPP;
  hipLaunchKernelGGL(heads_kernel, dim3((unsigned)ceil_div<int64_t>(P, 1024), (unsigned)B), dim3(256), 0, as_stream(stream), x, Wm,
                     row_stride, col_stride, bias, n_rows, H, P, out);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_dot(const float* dF, const float* f, int B, int H, int R, int64_t P, float* g, void* stream) {
  if (!dF || !f || !g || B < 0 || H <= 0 || R <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  if (P % 4 || R % 4) return CIPS3D_E_UNSUPP;
  hipLaunchKernelGGL(dot_kernel, dim3((unsigned)ceil_div<int64_t>(P, 1024), (unsigned)B), dim3(256), 0, as_stream(stream), dF, f, H,
                     R, P, g);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_composite(const cips3d_nerf_bwd_geom* G, const float* sdf, const float* crgb, const float* g,
                                         const float* dthumb, const float* sigmoid_beta, float* w, float* T_scratch,
                                         float* dsdf, float* dcrgb, float* ddnorm, float* dbeta_ray, void* stream) {
  if (!geom_ok(G) || !sdf || !crgb || !g || !dthumb || !w || !T_scratch || !dsdf || !dcrgb || !ddnorm)
    return CIPS3D_E_BADARG;
  if (G->B == 0) return 0;
  const int R = G->img_size * G->img_size;
  if (G->n_samples >= 2 && G->n_samples <= 32)
    hipLaunchKernelGGL(composite_par_kernel, dim3((unsigned)ceil_div(R, 32), (unsigned)G->B), dim3(32 * G->n_samples),
                       sizeof(float) * 7 * 32 * G->n_samples, as_stream(stream), *G, sdf, crgb, g, dthumb, sigmoid_beta, w,
                       T_scratch, dsdf, dcrgb, ddnorm, dbeta_ray);
  else
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)ceil_div(R, 256), (unsigned)G->B), dim3(256), 0, as_stream(stream), *G, sdf,
                       crgb, g, dthumb, sigmoid_beta, w, T_scratch, dsdf, dcrgb, ddnorm, dbeta_ray);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_row_dots(const float* a, const float* x, int nx, int64_t Px, float* out, int B, int rows,
                                        int64_t P, void* stream) {
  if (!a || !out || B < 0 || rows <= 0 || P <= 0 || nx < 0 || nx > 3 || (nx > 0 && (!x || Px <= 0))) return CIPS3D_E_BADARG;
  if (P % 4 || (nx > 0 && (Px % 4 || P % Px))) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipLaunchKernelGGL(row_dots_kernel, dim3((unsigned)ceil_div<int64_t>(P, 1024), (unsigned)rows, (unsigned)B), dim3(256), 0,
                     as_stream(stream), a, x, nx, nx > 0 ? Px : 4, out, rows, P);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_film_grad(float* buf, const float* pre, const float* film, int film_bstride, int mode,
                                         const float* w_sigma, const float* dsdf, const float* weights, const float* dF,
                                         const float* w_rgb, const float* dcrgb, float* dfilm, int B, int H, int R,
                                         int64_t P, void* stream) {
  if (!buf || !pre || !film || !dfilm || B < 0 || H <= 0 || R <= 0 || P <= 0) return CIPS3D_E_BADARG;
  if (mode != 0 && mode != 1) return CIPS3D_E_BADARG;
  if (mode == 1 && (!weights || !dF || !w_rgb || !dcrgb)) return CIPS3D_E_BADARG;
  if (mode == 0 && w_sigma && !dsdf) return CIPS3D_E_BADARG;
  if (P % 4 || R % 4) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  hipLaunchKernelGGL(film_grad_kernel, dim3((unsigned)ceil_div<int64_t>(P, 1024), (unsigned)H, (unsigned)B), dim3(256), 0,
                     as_stream(stream), buf, pre, film, film_bstride, mode, w_sigma, dsdf, weights, dF, w_rgb, dcrgb, dfilm, H,
                     R, P);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_camera(const cips3d_nerf_bwd_geom* G, const float* dptsn, const float* dvd_pt,
                                      const float* ddnorm, float* dcam, void* stream) {
  if (!geom_ok(G) || !dptsn || !dvd_pt || !ddnorm || !dcam) return CIPS3D_E_BADARG;
  if (G->B == 0) return 0;
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(dcam, 0, sizeof(float) * 12 * G->B, st);
  if (e != hipSuccess) return (int)e;
  const int R = G->img_size * G->img_size;
  hipLaunchKernelGGL(camera_chain_kernel, dim3((unsigned)ceil_div(R, 64), (unsigned)G->B), dim3(1024), 0, st, *G, dptsn, dvd_pt,
                     ddnorm, dcam);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_bwd_camera_acc(const cips3d_nerf_bwd_geom* G, const float* dptsn, const float* dvd_pt,
                                          const float* ddnorm, float* dcam, void* stream) {
  if (!geom_ok(G) || !dptsn || !dvd_pt || !ddnorm || !dcam) return CIPS3D_E_BADARG;
  if (G->B == 0) return 0;
  const int R = G->img_size * G->img_size;
  hipLaunchKernelGGL(camera_chain_kernel, dim3((unsigned)ceil_div(R, 64), (unsigned)G->B), dim3(1024), 0, as_stream(stream), *G, dptsn,
                     dvd_pt, ddnorm, dcam);
  return cips3d_launch_status();
}

extern "C" int cips3d_camera_params_bwd(const float* locations, const float* up, const float* dextrinsics, int B,
                                        float* dlocations, void* stream) {
  if (!locations || !dextrinsics || !dlocations || B < 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(camera_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), locations, up, dextrinsics, B,
                     dlocations);
  return cips3d_launch_status();
}
