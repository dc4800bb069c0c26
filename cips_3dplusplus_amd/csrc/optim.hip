// Adam step of the inversion loop (reference: torch.optim.Adam over {camera angles}, {NeRF W+}, {decoder W+, decoder
// parameters [, noise buffers]}, /root/reference/exp/cips3d/models/projector_v10.py:279-390,1195-1216) as ONE bandwidth-bound
// launch per 48 parameter tensors.  torch's own multi-tensor forms spend ~0.2 ms per step on the decoder's ~100 tensors (7 M
// parameters: 7 launches of ~30 us for 196 MB of traffic); one pass over (p, g, m, v) at HBM rate is ~45 us.
//
// Update rule (torch/optim/adam.py, amsgrad = False, weight_decay = 0, maximize = False):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <math.h>

#include "common.h"

namespace {

constexpr int ADAM_MAX = 48;            // entries per launch (kernel-argument budget: 48 x 40 B + prefix sums)
constexpr int ADAM_BLOCK_ELEMS = 4096;  // 256 threads x 4 float4

constexpr int ADAM_HYPERS = 8;          // distinct (lr, betas, eps, step) sets per launch: parameter groups of several optimisers
struct AdamHyper { float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps; };
struct AdamArgs {
  cips3d_adam_entry e[ADAM_MAX];
  int blk_begin[ADAM_MAX + 1];          // exclusive prefix sums of the workgroups per entry
  int n;
  AdamHyper h[ADAM_HYPERS];
  unsigned char hidx[ADAM_MAX];         // entry -> its set
};

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  int ei = 0;
  while (ei + 1 < a.n && (int)blockIdx.x >= a.blk_begin[ei + 1]) ++ei;          // (uniform: <= 48 scalar compares)
  const cips3d_adam_entry E = a.e[ei];
  const AdamHyper hy = a.h[a.hidx[ei]];
  const int64_t base = (int64_t)((int)blockIdx.x - a.blk_begin[ei]) * ADAM_BLOCK_ELEMS;
  const float omb1 = 1.f - hy.b1, omb2 = 1.f - hy.b2;
  auto upd = [&](float& p, float g, float& m, float& v) {
    m = fmaf(hy.b1, m, omb1 * g);
    v = fmaf(hy.b2, v, (omb2 * g) * g);
    p -= hy.lr_over_bc1 * (m / (sqrtf(v) * hy.inv_sqrt_bc2 + hy.eps));
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(E.p) | reinterpret_cast<uintptr_t>(E.g) | reinterpret_cast<uintptr_t>(E.m) |
                     reinterpret_cast<uintptr_t>(E.v)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t i = base + ((int64_t)it * 256 + threadIdx.x) * 4;
    if (i >= E.n) break;
    if (vec && i + 4 <= E.n) {
      float4 p = *reinterpret_cast<float4*>(E.p + i), m = *reinterpret_cast<float4*>(E.m + i), v = *reinterpret_cast<float4*>(E.v + i);
      const float4 g = *reinterpret_cast<const float4*>(E.g + i);
      upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(E.p + i) = p; *reinterpret_cast<float4*>(E.m + i) = m; *reinterpret_cast<float4*>(E.v + i) = v;
    } else {
      for (int64_t j = i; j < i + 4 && j < E.n; ++j) {
        float p = E.p[j], m = E.m[j], v = E.v[j];
        upd(p, E.g[j], m, v);
        E.p[j] = p; E.m[j] = m; E.v[j] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The inversion loss' two squared-difference terms (projector_v10.py:1173-1174: `(target - synth).square().sum() * rgb_weight
// + (target_thumb - synth_thumb).square().sum() * thumb_weight`; F.mse_loss, :1178, is the mean form) as three launches --
// partial sums, their fixed-order total, the gradient -- where the torch expression and its autograd graph are ~23.
constexpr int SQ_BLOCKS_MAX = 512;

__global__ void __launch_bounds__(256) sqdiff_partial_kernel(const float* __restrict__ a0, const float* __restrict__ b0, int64_t n0,
                                                             const float* __restrict__ a1, const float* __restrict__ b1, int64_t n1,
                                                             float* __restrict__ partial) {
  __shared__ float red[2][4];
  float s[2] = {0.f, 0.f};
  const int64_t stride = (int64_t)gridDim.x * 256;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float* a = k ? a1 : a0;
    const float* b = k ? b1 : b0;
    const int64_t n = k ? n1 : n0;
    const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
      const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
      s[k] = fmaf(d0, d0, s[k]); s[k] = fmaf(d1, d1, s[k]); s[k] = fmaf(d2, d2, s[k]); s[k] = fmaf(d3, d3, s[k]);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
      const float d = a[i] - b[i];
      s[k] = fmaf(d, d, s[k]);
    }
    s[k] = wave_sum(s[k]);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s[0]; red[1][w] = s[1]; }
  __syncthreads();
  if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// one workgroup: the partial sums in a fixed order (thread t takes blocks t, t + 256, ...; then the wave / workgroup tree)
__global__ void __launch_bounds__(256) sqdiff_finish_kernel(const float* __restrict__ partial, int blocks, float c0, float c1,
                                                            float* __restrict__ loss) {
  __shared__ float red[2][4];
  float s[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < blocks; i += 256) { s[0] += partial[i * 2]; s[1] += partial[i * 2 + 1]; }
  s[0] = wave_sum(s[0]); s[1] = wave_sum(s[1]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s[0]; red[1][w] = s[1]; }
  __syncthreads();
  if (threadIdx.x == 0)
    loss[0] = c0 * ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) + c1 * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
}

// d loss / d a_k = g * 2 c_k (a_k - b_k); blocks [0, blocks0) take tensor 0
__global__ void __launch_bounds__(256) sqdiff_bwd_kernel(const float* __restrict__ a0, const float* __restrict__ b0, int64_t n0, float c0,
                                                         float* __restrict__ d0, int blocks0, const float* __restrict__ a1,
                                                         const float* __restrict__ b1, int64_t n1, float c1, float* __restrict__ d1,
                                                         const float* __restrict__ gloss) {
  const bool second = (int)blockIdx.x >= blocks0;
  const float* a = second ? a1 : a0;
  const float* b = second ? b1 : b0;
  float* d = second ? d1 : d0;
  const int64_t n = second ? n1 : n0;
  const float f = 2.f * (second ? c1 : c0) * gloss[0];
  const int64_t base = (int64_t)((int)blockIdx.x - (second ? blocks0 : 0)) * 4096;
  const bool vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t i = base + ((int64_t)it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      const float4 x = *reinterpret_cast<const float4*>(a + i), y = *reinterpret_cast<const float4*>(b + i);
      *reinterpret_cast<float4*>(d + i) = make_float4(f * (x.x - y.x), f * (x.y - y.y), f * (x.z - y.z), f * (x.w - y.w));
    } else {
      for (int64_t j = i; j < i + 4 && j < n; ++j) d[j] = f * (a[j] - b[j]);
    }
  }
}

}  // namespace

extern "C" int cips3d_sqdiff_pair_partials(int64_t n0, int64_t n1) {
  const int64_t b = ceil_div<int64_t>((n0 > n1 ? n0 : n1), 4096);
  return (int)(b < 1 ? 1 : (b > SQ_BLOCKS_MAX ? SQ_BLOCKS_MAX : b));
}

extern "C" int cips3d_sqdiff_pair(const float* a0, const float* b0, int64_t n0, float c0, const float* a1, const float* b1,
                                  int64_t n1, float c1, float* partial, float* loss, void* stream) {
  if (n0 < 0 || n1 < 0 || (n0 > 0 && (!a0 || !b0)) || (n1 > 0 && (!a1 || !b1)) || !partial || !loss) return CIPS3D_E_BADARG;
  const int blocks = cips3d_sqdiff_pair_partials(n0, n1);
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a0, b0, n0, a1, b1, n1, partial);
  hipLaunchKernelGGL(sqdiff_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, blocks, c0, c1, loss);
  return cips3d_launch_status();
}

extern "C" int cips3d_sqdiff_pair_bwd(const float* a0, const float* b0, int64_t n0, float c0, float* d0, const float* a1,
                                      const float* b1, int64_t n1, float c1, float* d1, const float* gloss, void* stream) {
  if (n0 < 0 || n1 < 0 || (n0 > 0 && (!a0 || !b0 || !d0)) || (n1 > 0 && (!a1 || !b1 || !d1)) || !gloss) return CIPS3D_E_BADARG;
  const int blocks0 = (int)ceil_div<int64_t>(n0, 4096), blocks1 = (int)ceil_div<int64_t>(n1, 4096);
  if (blocks0 + blocks1 == 0) return 0;
  hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3(blocks0 + blocks1), dim3(256), 0, as_stream(stream), a0, b0, n0, c0, d0, blocks0, a1, b1,
                     n1, c1, d1, gloss);
  return cips3d_launch_status();
}

static bool adam_hyper_of(const cips3d_adam_hyper& g, AdamHyper* out) {
  if (g.step < 1 || !(g.beta1 >= 0.f && g.beta1 < 1.f) || !(g.beta2 >= 0.f && g.beta2 < 1.f)) return false;
  const double bc1 = 1.0 - pow((double)g.beta1, g.step), bc2 = 1.0 - pow((double)g.beta2, g.step);
  *out = AdamHyper{(float)((double)g.lr / bc1), (float)(1.0 / sqrt(bc2)), g.beta1, g.beta2, g.eps};
  return true;
}

// entries of several parameter groups (and optimisers) share launches: entry i takes the hyper-parameters hypers[group_of[i]]
extern "C" int cips3d_adam_step_groups(const cips3d_adam_entry* entries, const int* group_of, int n_entries,
                                       const cips3d_adam_hyper* hypers, int n_hypers, void* stream) {
  if (!entries || !group_of || !hypers || n_entries < 0 || n_hypers <= 0) return CIPS3D_E_BADARG;
  for (int i = 0; i < n_entries; ++i) {
    if (group_of[i] < 0 || group_of[i] >= n_hypers) return CIPS3D_E_BADARG;
    const cips3d_adam_entry& E = entries[i];
    if (!E.p || !E.g || !E.m || !E.v || E.n < 0) return CIPS3D_E_BADARG;
  }
  int first = 0;
  while (first < n_entries) {
    AdamArgs a;
    int local_of[ADAM_HYPERS], n_local = 0, blocks = 0;
    a.n = 0;
    while (first + a.n < n_entries && a.n < ADAM_MAX) {
      const int gi = group_of[first + a.n];
      int li = 0;
      while (li < n_local && local_of[li] != gi) ++li;
      if (li == n_local) {
        if (n_local == ADAM_HYPERS) break;                 // a ninth set: the next launch
        if (!adam_hyper_of(hypers[gi], &a.h[n_local])) return CIPS3D_E_BADARG;
        local_of[n_local++] = gi;
      }
      a.e[a.n] = entries[first + a.n];
      a.hidx[a.n] = (unsigned char)li;
      a.blk_begin[a.n] = blocks;
      blocks += (int)ceil_div<int64_t>(entries[first + a.n].n, ADAM_BLOCK_ELEMS);
      ++a.n;
    }
    a.blk_begin[a.n] = blocks;
    first += a.n;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    const int rc = cips3d_launch_status();
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int cips3d_adam_step(const cips3d_adam_entry* entries, int n_entries, float lr, float beta1, float beta2, float eps,
                                int step, void* stream) {
  if (!entries || n_entries < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return CIPS3D_E_BADARG;
  const cips3d_adam_hyper h{lr, beta1, beta2, eps, step};
  for (int first = 0; first < n_entries; first += ADAM_MAX) {
    int zeros[ADAM_MAX] = {};
    const int n = n_entries - first < ADAM_MAX ? n_entries - first : ADAM_MAX;
    if (const int rc = cips3d_adam_step_groups(entries + first, zeros, n, &h, 1, stream)) return rc;
  }
  return 0;
}
