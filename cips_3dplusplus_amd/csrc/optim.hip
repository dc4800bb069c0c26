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

struct AdamArgs {
  cips3d_adam_entry e[ADAM_MAX];
  int blk_begin[ADAM_MAX + 1];          // exclusive prefix sums of the workgroups per entry
  int n;
  float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps;
};

__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) {
  int ei = 0;
  while (ei + 1 < a.n && (int)blockIdx.x >= a.blk_begin[ei + 1]) ++ei;          // (uniform: <= 48 scalar compares)
  const cips3d_adam_entry E = a.e[ei];
  const int64_t base = (int64_t)((int)blockIdx.x - a.blk_begin[ei]) * ADAM_BLOCK_ELEMS;
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
  auto upd = [&](float& p, float g, float& m, float& v) {
    m = fmaf(a.b1, m, omb1 * g);
    v = fmaf(a.b2, v, (omb2 * g) * g);
    p -= a.lr_over_bc1 * (m / (sqrtf(v) * a.inv_sqrt_bc2 + a.eps));
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

}  // namespace

extern "C" int cips3d_adam_step(const cips3d_adam_entry* entries, int n_entries, float lr, float beta1, float beta2, float eps,
                                int step, void* stream) {
  if (!entries || n_entries < 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return CIPS3D_E_BADARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  for (int first = 0; first < n_entries; first += ADAM_MAX) {
    AdamArgs a;
    a.n = n_entries - first < ADAM_MAX ? n_entries - first : ADAM_MAX;
    int blocks = 0;
    for (int i = 0; i < a.n; ++i) {
      const cips3d_adam_entry& E = entries[first + i];
      if (!E.p || !E.g || !E.m || !E.v || E.n < 0) return CIPS3D_E_BADARG;
      a.e[i] = E;
      a.blk_begin[i] = blocks;
      blocks += (int)ceil_div<int64_t>(E.n, ADAM_BLOCK_ELEMS);
    }
    a.blk_begin[a.n] = blocks;
    a.lr_over_bc1 = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    const int rc = cips3d_launch_status();
    if (rc != 0) return rc;
  }
  return 0;
}
