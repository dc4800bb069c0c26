// cips3d_fused_bias_act: bias + (leaky-ReLU | linear) + gain, forward / first / second derivative.
// Replaces fused.fused_bias_act (reference exp/op/fused_bias_act.cpp:11-20,
// fused_bias_act_kernel.cu:18-98).  HBM-bound: 4 B read + 4 B written per element
// (+4 B for `ref` in the derivative mode); 16-byte accesses whenever the shape allows.
#include "common.h"

namespace {

template <int ACT, int GRAD>
__device__ __forceinline__ float bias_act_one(float x, float ref, float alpha, float scale) {
  float y;
  if (GRAD == 2) {
    y = 0.f;
  } else if (ACT == 3) {
    y = (GRAD == 0) ? ((x > 0.f) ? x : x * alpha) : ((ref > 0.f) ? x : x * alpha);
  } else {
    y = x;
  }
  return y * scale;
}

// VEC = 4: n % 4 == 0, step_b % 4 == 0 (so the 4 elements of a vector share one bias), 16-B
// aligned pointers.  VEC = 1: anything.
template <int ACT, int GRAD, int VEC>
__global__ void __launch_bounds__(256) bias_act_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ ref,
                                                       float* __restrict__ out, int64_t n_vec,
                                                       int64_t step_b_vec, int64_t size_b,
                                                       float alpha, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    float b = 0.f;
    if (bias) b = bias[(i / step_b_vec) % size_b];
    if (VEC == 4) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (GRAD == 1 && ref) r = reinterpret_cast<const float4*>(ref)[i];
      float4 o;
      o.x = bias_act_one<ACT, GRAD>(v.x + b, r.x, alpha, scale);
      o.y = bias_act_one<ACT, GRAD>(v.y + b, r.y, alpha, scale);
      o.z = bias_act_one<ACT, GRAD>(v.z + b, r.z, alpha, scale);
      o.w = bias_act_one<ACT, GRAD>(v.w + b, r.w, alpha, scale);
      reinterpret_cast<float4*>(out)[i] = o;
    } else {
      float r = (GRAD == 1 && ref) ? ref[i] : 0.f;
      out[i] = bias_act_one<ACT, GRAD>(x[i] + b, r, alpha, scale);
    }
  }
}

template <int ACT, int GRAD>
int launch_bias_act(const float* x, const float* bias, const float* ref, float* out, int64_t n,
                    int64_t step_b, int64_t size_b, float alpha, float scale, hipStream_t st) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                         reinterpret_cast<uintptr_t>(ref)) & 15) == 0;
  const bool vec = aligned && (n % 4 == 0) && (!bias || step_b % 4 == 0);
  const int64_t n_vec = vec ? n / 4 : n;
  const int64_t sb = vec ? (bias ? step_b / 4 : 1) : step_b;
  int64_t blocks = ceil_div<int64_t>(n_vec, 256);
  if (blocks > 8192) blocks = 8192;   // grid-stride beyond 32 resident workgroups per CU
  if (vec)
    hipLaunchKernelGGL((bias_act_kernel<ACT, GRAD, 4>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias,
                       ref, out, n_vec, sb, size_b, alpha, scale);
  else
    hipLaunchKernelGGL((bias_act_kernel<ACT, GRAD, 1>), dim3((unsigned)blocks), dim3(256), 0, st, x, bias,
                       ref, out, n_vec, sb, size_b, alpha, scale);
  return cips3d_launch_status();
}

// image post-step of the multi-view loop (render_video_web_v10.py:1825-1826, tl2 img_tensor_to_pil): clamp to
// [-1, 1], map to [0, 255], round to nearest -> uint8.  16 pixels-bytes per thread (4 x float4 in, one 16-B store).
__global__ void __launch_bounds__(256) rgb_to_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out,
                                                        int64_t n16, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = reinterpret_cast<const float4*>(x)[i * 4 + k];
      const float f[4] = {v.x, v.y, v.z, v.w};
      uint32_t pk = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float c = fminf(fmaxf(f[j], -1.f), 1.f);
        pk |= (uint32_t)__float2int_rn((c + 1.f) * 127.5f) << (8 * j);
      }
      w[k] = pk;
    }
    reinterpret_cast<uint4*>(out)[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // tail (n % 16 elements), one thread
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t e = n16 * 16; e < n; ++e) {
      const float c = fminf(fmaxf(x[e], -1.f), 1.f);
      out[e] = (uint8_t)__float2int_rn((c + 1.f) * 127.5f);
    }
}

}  // namespace

extern "C" int cips3d_rgb_to_uint8(const float* rgb, uint8_t* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  if (!rgb || !out || n < 0) return CIPS3D_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(rgb) | reinterpret_cast<uintptr_t>(out)) & 15) return CIPS3D_E_UNSUPP;
  const int64_t n16 = n / 16;
  int64_t blocks = ceil_div<int64_t>(n16 > 0 ? n16 : 1, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rgb_to_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), rgb, out, n16, n);
  return cips3d_launch_status();
}

extern "C" int cips3d_fused_bias_act(const float* x, const float* bias, const float* ref, float* out,
                                     int64_t n, int64_t step_b, int64_t size_b, int act, int grad,
                                     float alpha, float scale, void* stream) {
  if (n == 0) return 0;
  if (!x || !out || n < 0) return CIPS3D_E_BADARG;
  if (bias && (step_b <= 0 || size_b <= 0)) return CIPS3D_E_BADARG;
  if (!bias) { step_b = 1; size_b = 1; }
  if ((act != 1 && act != 3) || grad < 0 || grad > 2) return CIPS3D_E_UNSUPP;
  hipStream_t st = as_stream(stream);
  const int key = act * 10 + grad;
  switch (key) {
    case 10: case 11: return launch_bias_act<1, 0>(x, bias, ref, out, n, step_b, size_b, alpha, scale, st);
    case 12: return launch_bias_act<1, 2>(x, bias, ref, out, n, step_b, size_b, alpha, scale, st);
    case 30: return launch_bias_act<3, 0>(x, bias, ref, out, n, step_b, size_b, alpha, scale, st);
    case 31: return launch_bias_act<3, 1>(x, bias, ref, out, n, step_b, size_b, alpha, scale, st);
    case 32: return launch_bias_act<3, 2>(x, bias, ref, out, n, step_b, size_b, alpha, scale, st);
  }
  return CIPS3D_E_UNSUPP;
}
