// Small dense layers of the mapping networks and style heads.
//   cips3d_linear        MappingLinear / EqualLinear (+PixelNorm, +truncation) of
//                        reference models/model_v3.py:32-65,183-210,1299-1418
//   cips3d_linear_table  many independent heads in one launch: FiLM gamma/beta LinearLayers
//                        (cips3d/volume_renderer.py:15-35,66-67) and ModulatedConv2d.modulation
//                        (models/model_v3.py:254,268)
// GEMV-shaped (batch 1..few): bound by streaming W once from HBM/L2 -> one wave per output row,
// lanes stride the row in 16-byte pieces, butterfly reduction; the batch loops inside the wave so W
// is read once per launch whatever B is.
#include "common.h"
#include "rng_device.h"

namespace {

constexpr int BT = 4;  // batch rows accumulated per pass over a weight row

__device__ __forceinline__ void dot_rows(const float* __restrict__ w, const float* __restrict__ x,
                                         int64_t x_stride, int nb, int in_dim, int lane, float (&acc)[BT]) {
#pragma unroll
  for (int j = 0; j < BT; ++j) acc[j] = 0.f;
  const bool vec = (in_dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(w) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (x_stride % 4 == 0);
  if (vec) {
    for (int i = lane * 4; i < in_dim; i += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(w + i);
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (j < nb) {
          const float4 xv = *reinterpret_cast<const float4*>(x + j * x_stride + i);
          acc[j] = fmaf(wv.x, xv.x, acc[j]);
          acc[j] = fmaf(wv.y, xv.y, acc[j]);
          acc[j] = fmaf(wv.z, xv.z, acc[j]);
          acc[j] = fmaf(wv.w, xv.w, acc[j]);
        }
      }
    }
  } else {
    for (int i = lane; i < in_dim; i += 64) {
      const float wv = w[i];
#pragma unroll
      for (int j = 0; j < BT; ++j)
        if (j < nb) acc[j] = fmaf(wv, x[j * x_stride + i], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < BT; ++j) acc[j] = wave_sum(acc[j]);
}

__device__ __forceinline__ void linear_rows(const cips3d_linear_args& a, int row);

__global__ void __launch_bounds__(256) linear_kernel(cips3d_linear_args a) {
  linear_rows(a, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// two independent layers (the i-th layers of the two mapping networks) in one launch: blocks [0, blocks_a) run `a`
// Blocks from blocks_ab on host a slice of the forward's fresh-noise draw (cips3d_rng_job): the two layers are ~190 work
// groups waiting on one dependent load each, the chip is otherwise idle for the ~4.6 us of this launch.
__global__ void __launch_bounds__(256) linear_pair_kernel(cips3d_linear_args a, cips3d_linear_args b, int blocks_a, int blocks_ab,
                                                          cips3d_rng_job job) {
  if ((int)blockIdx.x < blocks_a) linear_rows(a, blockIdx.x * 4 + (threadIdx.x >> 6));
  else if ((int)blockIdx.x < blocks_ab) linear_rows(b, (blockIdx.x - blocks_a) * 4 + (threadIdx.x >> 6));
  else {
    const long long t = job.t0 + (long long)(blockIdx.x - blocks_ab) * 256 + threadIdx.x;
    if (t < job.t1) rng_fill_thread(job.seed_lo, job.seed_hi, job.base, job.normal, job.n_normal, job.uniform, job.n_uniform, t);
  }
}

__device__ __forceinline__ void linear_rows(const cips3d_linear_args& a, int row) {
  const int lane = threadIdx.x & 63;
  if (row >= a.out_dim) return;
  const float* w = a.W + (int64_t)row * a.in_dim;
  const float b = a.bias ? a.bias[row] * a.b_scale : 0.f;
  for (int b0 = 0; b0 < a.B; b0 += BT) {
    const int nb = min(BT, a.B - b0);
    const float* x = a.x + (int64_t)b0 * a.x_stride;
    float acc[BT];
    dot_rows(w, x, a.x_stride, nb, a.in_dim, lane, acc);
    float nrm[BT];
    if (a.pixelnorm) {
      // PixelNorm folded in: dot(W, x * r) = r * dot(W, x), r = rsqrt(mean(x^2) + 1e-8)
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        float s = 0.f;
        if (j < nb)
          for (int i = lane; i < a.in_dim; i += 64) { const float v = x[j * a.x_stride + i]; s = fmaf(v, v, s); }
        s = wave_sum(s);
        nrm[j] = rsqrtf(s / (float)a.in_dim + 1e-8f);
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (j >= nb) break;
        float y = acc[j];
        if (a.pixelnorm) y *= nrm[j];
        y = fmaf(y, a.w_scale, b);
        if (a.lrelu) y = lrelu02(y) * a.act_gain;
        y = fmaf(y, a.out_scale, a.out_shift);
        if (a.trunc_mean) { const float m = a.trunc_mean[row]; y = fmaf(a.trunc_psi, y - m, m); }
        for (int r = 0; r < a.out_repeat; ++r)   // broadcast of one latent to every layer's style slot
          a.out[(int64_t)(b0 + j) * a.out_stride + r * a.out_repeat_stride + row] = y;
      }
    }
  }
}

__device__ __forceinline__ void table_rows(const cips3d_linear_desc* __restrict__ table, int n_desc, int total_rows, int B,
                                           int grow);

__global__ void __launch_bounds__(256) linear_table_kernel(const cips3d_linear_desc* __restrict__ table,
                                                           int n_desc, int total_rows, int B) {
  table_rows(table, n_desc, total_rows, B, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// a table of heads + blocks that zero a small array (the forward's range workspace, forward.hip)
__global__ void __launch_bounds__(256) linear_table_zero_kernel(const cips3d_linear_desc* __restrict__ table, int n_desc,
                                                                int total_rows, int B, int blocks_t, float* __restrict__ zero_ptr,
                                                                int zero_n) {
  if ((int)blockIdx.x < blocks_t) { table_rows(table, n_desc, total_rows, B, blockIdx.x * 4 + (threadIdx.x >> 6)); return; }
  for (int i = ((int)blockIdx.x - blocks_t) * 256 + threadIdx.x; i < zero_n; i += ((int)gridDim.x - blocks_t) * 256) zero_ptr[i] = 0.f;
}

// one mapping layer and an independent table of heads in one launch: blocks [0, blocks_a) run the layer
__global__ void __launch_bounds__(256) linear_and_table_kernel(cips3d_linear_args a, int blocks_a,
                                                               const cips3d_linear_desc* __restrict__ table, int n_desc,
                                                               int total_rows, int B) {
  if ((int)blockIdx.x < blocks_a) linear_rows(a, blockIdx.x * 4 + (threadIdx.x >> 6));
  else table_rows(table, n_desc, total_rows, B, (blockIdx.x - blocks_a) * 4 + (threadIdx.x >> 6));
}

__device__ __forceinline__ void table_rows(const cips3d_linear_desc* __restrict__ table, int n_desc, int total_rows, int B,
                                           int grow) {
  const int lane = threadIdx.x & 63;
  if (grow >= total_rows) return;
  // the descriptor owning this row (row_begin is an exclusive prefix sum)
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
  const cips3d_linear_desc d = table[lo];
  const int row = grow - d.row_begin;
  const float* w = d.W + (int64_t)row * d.in_dim;
  const float b = d.bias ? d.bias[row] * d.b_scale : 0.f;
  for (int b0 = 0; b0 < B; b0 += BT) {
    const int nb = min(BT, B - b0);
    float acc[BT];
    dot_rows(w, d.x + (int64_t)b0 * d.x_stride, d.x_stride, nb, d.in_dim, lane, acc);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (j >= nb) break;
        const float y = fmaf(acc[j], d.w_scale, b);
        d.out[(int64_t)(b0 + j) * d.out_stride + row] = fmaf(y, d.out_scale, d.out_shift);
      }
    }
  }
}

// PixelNorm as its own op (models/model_v3.py:32-37): one wave per row
__global__ void __launch_bounds__(256) pixel_norm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* xr = x + (int64_t)row * C;
  float s = 0.f;
  for (int i = lane; i < C; i += 64) s = fmaf(xr[i], xr[i], s);
  s = wave_sum(s);
  const float r = rsqrtf(s / (float)C + 1e-8f);
  for (int i = lane; i < C; i += 64) y[(int64_t)row * C + i] = xr[i] * r;
}

}  // namespace

extern "C" int cips3d_pixel_norm(const float* x, float* y, int B, int C, void* stream) {
  if (!x || !y || B < 0 || C <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(pixel_norm_kernel, dim3(ceil_div(B, 4)), dim3(256), 0, as_stream(stream), x, y, B, C);
  return cips3d_launch_status();
}

extern "C" int cips3d_linear(const float* x, int64_t x_stride, const float* W, const float* bias, float* out,
                             int64_t out_stride, int B, int in_dim, int out_dim, float w_scale, float b_scale,
                             int pixelnorm, int lrelu, float act_gain, float out_scale, float out_shift,
                             const float* trunc_mean, float trunc_psi, int out_repeat, int64_t out_repeat_stride,
                             void* stream) {
  if (!x || !W || !out || B < 0 || in_dim <= 0 || out_dim <= 0 || out_repeat < 1) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  cips3d_linear_args a{x, x_stride, W, bias, out, out_stride, B, in_dim, out_dim, w_scale, b_scale, pixelnorm, lrelu,
                       act_gain, out_scale, out_shift, trunc_mean, trunc_psi, out_repeat, out_repeat_stride};
  hipLaunchKernelGGL(linear_kernel, dim3(ceil_div(out_dim, 4)), dim3(256), 0, as_stream(stream), a);
  return cips3d_launch_status();
}

// library-internal (forward.hip): two layers, one launch
int cips3d_linear_pair(const cips3d_linear_args& a, const cips3d_linear_args& b, void* stream, const cips3d_rng_job* job) {
  const int ba = ceil_div(a.out_dim, 4), bb = ceil_div(b.out_dim, 4);
  cips3d_rng_job j{};
  int br = 0;
  if (job && job->t1 > job->t0) { j = *job; br = (int)ceil_div<long long>(j.t1 - j.t0, 256); }
  hipLaunchKernelGGL(linear_pair_kernel, dim3(ba + bb + br), dim3(256), 0, as_stream(stream), a, b, ba, ba + bb, j);
  return cips3d_launch_status();
}

// library-internal (forward.hip): a layer and a table that do not depend on each other, one launch
int cips3d_linear_and_table(const cips3d_linear_args& a, const cips3d_linear_desc* table_dev, int n_desc, int total_rows,
                            void* stream) {
  const int ba = ceil_div(a.out_dim, 4);
  hipLaunchKernelGGL(linear_and_table_kernel, dim3(ba + ceil_div(total_rows, 4)), dim3(256), 0, as_stream(stream), a, ba,
                     table_dev, n_desc, total_rows, a.B);
  return cips3d_launch_status();
}

int cips3d_linear_table_zero(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int B, float* zero_ptr,
                             int zero_n, void* stream) {
  if (!table_dev || n_desc <= 0 || total_rows <= 0 || B < 0 || !zero_ptr || zero_n <= 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  const int bt = ceil_div(total_rows, 4);
  int bz = ceil_div(zero_n, 1024);
  if (bz > 64) bz = 64;
  hipLaunchKernelGGL(linear_table_zero_kernel, dim3(bt + bz), dim3(256), 0, as_stream(stream), table_dev, n_desc, total_rows, B, bt,
                     zero_ptr, zero_n);
  return cips3d_launch_status();
}

extern "C" int cips3d_linear_table(const cips3d_linear_desc* table_dev, int n_desc, int total_rows, int B,
                                   void* stream) {
  if (!table_dev || n_desc <= 0 || total_rows <= 0 || B < 0) return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  hipLaunchKernelGGL(linear_table_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, as_stream(stream),
                     table_dev, n_desc, total_rows, B);
  return cips3d_launch_status();
}
