// Small dense layers of the mapping networks and style heads.
//   cips3d_linear        MappingLinear / EqualLinear (+PixelNorm, +truncation) of
//                        reference models/model_v3.py:32-65,183-210,1299-1418
//   cips3d_linear_table  many independent heads in one launch: FiLM gamma/beta LinearLayers
//                        (cips3d/volume_renderer.py:15-35,66-67) and ModulatedConv2d.modulation
//                        (models/model_v3.py:254,268)
// GEMV-shaped (batch 1..few): bound by streaming W once from HBM/L2 -> one wave per output row,
// lanes stride the row in 16-byte pieces, butterfly reduction; the batch loops inside the wave so W
// is read once per launch whatever B is.
#include <type_traits>

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
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (int i = lane * 4; i < in_dim; i += 256) {
      const v4 wv = *cips3d_g(reinterpret_cast<const v4*>(w + i));         // (global: see cips3d_g)
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (j < nb) {
          const v4 xv = *cips3d_g(reinterpret_cast<const v4*>(x + j * x_stride + i));
          acc[j] = fmaf(wv.x, xv.x, acc[j]);
          acc[j] = fmaf(wv.y, xv.y, acc[j]);
          acc[j] = fmaf(wv.z, xv.z, acc[j]);
          acc[j] = fmaf(wv.w, xv.w, acc[j]);
        }
      }
    }
  } else {
    for (int i = lane; i < in_dim; i += 64) {
      const float wv = cips3d_g(w)[i];
#pragma unroll
      for (int j = 0; j < BT; ++j)
        if (j < nb) acc[j] = fmaf(wv, cips3d_g(x)[j * x_stride + i], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < BT; ++j) acc[j] = wave_sum(acc[j]);
}

__device__ __forceinline__ void linear_rows(const cips3d_linear_args& a, int row);

// everything of a mapping layer behind its dot product (one value; `b` = the row's scaled bias)
__device__ __forceinline__ float linear_post_m(const cips3d_linear_args& a, float b, float acc, float nrm, float m) {
  float y = acc;
  if (a.pixelnorm) y *= nrm;
  y = fmaf(y, a.w_scale, b);
  if (a.lrelu) y = lrelu02(y) * a.act_gain;
  y = fmaf(y, a.out_scale, a.out_shift);
  if (a.trunc_mean) y = fmaf(a.trunc_psi, y - m, m);
  return y;
}
__device__ __forceinline__ float linear_post(const cips3d_linear_args& a, int row, float b, float acc, float nrm) {
  return linear_post_m(a, b, acc, nrm, a.trunc_mean ? a.trunc_mean[row] : 0.f);
}

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
        const float y = linear_post(a, row, b, acc[j], a.pixelnorm ? nrm[j] : 1.f);
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
  const float b = d.bias ? cips3d_g(d.bias)[row] * d.b_scale : 0.f;
  for (int b0 = 0; b0 < B; b0 += BT) {
    const int nb = min(BT, B - b0);
    float acc[BT];
    dot_rows(w, d.x + (int64_t)b0 * d.x_stride, d.x_stride, nb, d.in_dim, lane, acc);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (j >= nb) break;
        const float y = fmaf(acc[j], d.w_scale, b);
        cips3d_g(d.out)[(int64_t)(b0 + j) * d.out_stride + row] = fmaf(y, d.out_scale, d.out_shift);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The style phase in ONE launch (cips3d_style_phase, forward.hip): both mapping chains, the FiLM heads, the modulation heads,
// the call's noise draw and the zeroing of the range workspace.  As launches it is a chain of seven dependent ~5 us kernels
// (profiles/r03_final_kernel_stats.csv: 3 x linear_pair, linear, linear_and_table, linear_table_zero = 33 us) whose arithmetic
// is a few hundred nanoseconds each: what a layer pays is the kernel boundary plus the round trip of its weights.  Here G
// workgroups (one per CU) stay resident for the whole phase:
//   * every weight row a wave will need -- its row of every chain layer, its head rows -- is requested at the START of the
//     launch into registers, so the weights' round trips overlap each other and the first layers;
//   * a layer's outputs travel between workgroups as 8-byte {value, tag} granules: one agent-scope 64-bit atomic store by
//     the producing lane, agent-scope 64-bit atomic loads by the consumers (a granule is never torn and carries its own
//     "ready": no flag, no fence, no cache-wide write-back or invalidate -- MI355X_MICROARCH.md, inter-workgroup visibility,
//     "8-B agent atomics both sides").  tag = generation + 1 of this launch; the generation word is advanced by workgroup 0
//     after its last wait, which every workgroup's last-layer rows have passed -- so every workgroup has read it by then;
//   * every workgroup stages a layer's input vector(s) in LDS (polling its share of the granules) and computes its rows
//     with the arithmetic of dot_rows / linear_post / table_rows, in their order: results are bit-identical to the launches.
// A poll gives up after SP_TIMEOUT ticks of the 100 MHz wall clock (a workgroup that never became resident, a fault):
// the launch then ends with garbage in its outputs and sync[1] != 0 instead of hanging the queue.
// Diagnostics (-DCIPS3D_SP_STAMPS): workgroup 0 records 10 ns ticks since its start in sync[3..].  Off by default: reading the
// wall clock is a scalar memory round trip, and workgroup 0's rows are on everybody's critical path.
#ifdef CIPS3D_SP_STAMPS
#define SP_STAMP(k) do { SP_DRAIN(); if (g == 0 && t == 0) a.sync[k] = (unsigned)(wall_clock64() - t_begin); } while (0)
#else
#define SP_STAMP(k) ((void)0)
#endif
#ifdef CIPS3D_SP_HARD_STAMPS      /* diagnostic build: every stamp drains the wave's memory operations first */
#define SP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define SP_DRAIN() ((void)0)
#endif
typedef float spv4 __attribute__((ext_vector_type(4)));      // (a native vector: loads through address-space pointers)
constexpr int SP_MAXL = CIPS3D_MAX_MAP_LAYERS;
constexpr int SP_PF_FILM = 2, SP_PF_MOD = 6;      // head rows per wave whose weights are requested at the start
constexpr long long SP_TIMEOUT = 20000000;        // 0.2 s

struct StylePhaseArgs {
  cips3d_linear_args r[SP_MAXL], d[SP_MAXL];      // the layers of the two chains (x of layer 0: the call's z)
  int nr, nd, B, G;
  const cips3d_linear_desc* film; int film_n, film_rows;
  const cips3d_linear_desc* mod; int mod_n, mod_rows;
  const float* styles_r; const float* styles_d;   // what the heads' x pointers point into (every slot holds the same w)
  int sd_r, sd_d;
  unsigned long long* xch;                        // granules [layer][chain][B][dim]
  unsigned* sync;                                 // [0] generation, [1] polls that gave up
  int dim;
  int blocks_rng, blocks_zero, zero_n;
  float* zero_ptr;
  cips3d_rng_job job;
};

// Pointers that come out of LDS copies / lane broadcasts are generic to the compiler: its flat loads and stores count on
// lgkmcnt as well, so every later wait for an LDS read also waits for those stores' acknowledgements (microseconds).  These
// casts say "global".
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T* sp_g(const T* p) {
  return (const __attribute__((address_space(1))) T*)p;
}
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T* sp_g(T* p) {
  return (__attribute__((address_space(1))) T*)p;
}

__device__ __forceinline__ void sp_fetch(const float* __restrict__ w, int in_dim, bool vec, int lane, spv4 (&w2)[2]) {
  w2[0] = spv4{0.f, 0.f, 0.f, 0.f};
  w2[1] = w2[0];
  if (vec) {
    if (lane * 4 < in_dim) w2[0] = *sp_g(reinterpret_cast<const spv4*>(w + lane * 4));
    if (lane * 4 + 256 < in_dim) w2[1] = *sp_g(reinterpret_cast<const spv4*>(w + lane * 4 + 256));
  }
}

#define SP_LDS(T, p) ((const __attribute__((address_space(3))) T*)(p))      /* x (and the chain rows' copies) live in LDS */
__device__ __forceinline__ float sp_fma4(const spv4 wv, const float* x, float acc) {
  const spv4 xv = *SP_LDS(spv4, x);
  acc = fmaf(wv.x, xv.x, acc);
  acc = fmaf(wv.y, xv.y, acc);
  acc = fmaf(wv.z, xv.z, acc);
  acc = fmaf(wv.w, xv.w, acc);
  return acc;
}

// dot_rows for ONE batch row with x in LDS: the same products in the same order.  The row's weights at w: its LDS copy
// (W_LDS) or global memory
template <bool W_LDS>
__device__ __forceinline__ float sp_dot(const float* w, const float* x, int in_dim, bool vec, int lane) {
  float acc = 0.f;
  if (vec) {
    for (int i = lane * 4; i < in_dim; i += 256) {
      spv4 wv;
      if (W_LDS) wv = *SP_LDS(spv4, w + i);
      else wv = *sp_g(reinterpret_cast<const spv4*>(w + i));
      acc = sp_fma4(wv, x + i, acc);
    }
  } else {
    for (int i = lane; i < in_dim; i += 64) acc = fmaf(W_LDS ? *SP_LDS(float, w + i) : sp_g(w)[i], *SP_LDS(float, x + i), acc);
  }
  return wave_sum(acc);
}

__device__ __forceinline__ bool sp_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the granules of one layer output (rows of `out_dim` values per batch row, pitch `dim`) -> LDS, as they arrive
__device__ __forceinline__ void sp_gather(const unsigned long long* src, int B, int out_dim, int dim, float* dst, unsigned tag,
                                          long long deadline, unsigned* sync, int t) {
  for (int b = 0; b < B; ++b)
    for (int i = t; i < out_dim; i += 256) {
      const unsigned long long* p = src + b * dim + i;
      unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned spin = 1; (unsigned)(v >> 32) != tag; ++spin) {
        if ((spin & 255u) == 0 && wall_clock64() > deadline) { atomicAdd(sync + 1, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      dst[b * dim + i] = __uint_as_float((unsigned)v);
    }
}

// what a wave keeps of the head rows it requested at the start: row j's scalars live in lane j
struct SpHead {
  float b, ws, os, oh;
  float* out; long long ostr;
  const float* wrow;
  int xoff, in_dim, vec;
};

template <typename T>
__device__ __forceinline__ T sp_from_lane(T v, int j) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "");
  if constexpr (sizeof(T) == 4) {
    int x;
    __builtin_memcpy(&x, &v, 4);
    x = __builtin_amdgcn_readlane(x, j);
    __builtin_memcpy(&v, &x, 4);
  } else {
    int x[2];
    __builtin_memcpy(x, &v, 8);
    x[0] = __builtin_amdgcn_readlane(x[0], j);
    x[1] = __builtin_amdgcn_readlane(x[1], j);
    __builtin_memcpy(&v, x, 8);
  }
  return v;
}

// the table entry owning row h, from the row_begin every lane holds of "its" entry (owner_desc's ballot)
__device__ __forceinline__ int sp_owner(const cips3d_linear_desc* __restrict__ table, int n_desc, int h, int rb) {
  if (n_desc <= 64) return __popcll(__ballot(rb <= h)) - 1;
  int lo = 0, hi = n_desc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].row_begin <= h) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// what the head row h needs, from its table entry (d may be loaded per lane: every field is a function of d and h)
__device__ __forceinline__ SpHead sp_head_from(const cips3d_linear_desc& d, int h, const float* base, int sd) {
  const int row = h - d.row_begin;
  SpHead s;
  s.wrow = d.W + (int64_t)row * d.in_dim;
  s.b = d.bias ? sp_g(d.bias)[row] * d.b_scale : 0.f;
  s.ws = d.w_scale; s.os = d.out_scale; s.oh = d.out_shift;
  s.out = d.out + row; s.ostr = d.out_stride;
  s.xoff = (int)((d.x - base) % sd);
  s.in_dim = d.in_dim;
  s.vec = (d.in_dim % 4 == 0) && sp_al16(s.wrow) && sp_al16(d.x) && (d.x_stride % 4 == 0);
  return s;
}

__device__ __forceinline__ SpHead sp_head_of(const cips3d_linear_desc* __restrict__ table, int n_desc, int h, const float* base,
                                             int sd, int lane) {
  const int rb = (n_desc <= 64 && lane < n_desc) ? table[lane].row_begin : 0x7fffffff;
  const cips3d_linear_desc d = table[sp_owner(table, n_desc, h, rb)];
  return sp_head_from(d, h, base, sd);
}

__device__ __forceinline__ void sp_head_row(const SpHead& s, const float* xs, int B, int dim, int lane) {
  for (int b = 0; b < B; ++b) {
    const float acc = sp_dot<false>(s.wrow, xs + b * dim + s.xoff, s.in_dim, s.vec != 0, lane);
    if (lane == 0) {
      const float y = fmaf(acc, s.ws, s.b);
      sp_g(s.out)[(int64_t)b * s.ostr] = fmaf(y, s.os, s.oh);
    }
  }
}

// Requests the weights of this wave's first PF head rows.  Three dependent round trips for ALL rows together (a row at a time
// it was three per row, one behind the other: ~20 us in front of the first layer): the table's row_begin column; lane j's
// table entry and bias of row j; the rows' weights.
template <int PF>
__device__ __forceinline__ void sp_heads_fetch(const cips3d_linear_desc* __restrict__ table, int n_desc, int total_rows, int q,
                                               const float* base, int sd, int lane, SpHead& mine, spv4 (&wr)[PF][2]) {
  const int first = q * PF, stride = 1;
#pragma unroll
  for (int j = 0; j < PF; ++j) wr[j][0] = wr[j][1] = spv4{0.f, 0.f, 0.f, 0.f};
  if (total_rows <= 0) return;
  const int rb = (n_desc <= 64 && lane < n_desc) ? table[lane].row_begin : 0x7fffffff;
  int my_lo = -1, my_h = 0;
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int h = first + j * stride;
    if (h < total_rows) {
      const int lo = sp_owner(table, n_desc, h, rb);
      if (lane == j) { my_lo = lo; my_h = h; }
    }
  }
  if (my_lo >= 0) mine = sp_head_from(table[my_lo], my_h, base, sd);
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int h = first + j * stride;
    if (h < total_rows)
      sp_fetch(sp_from_lane(mine.wrow, j), sp_from_lane(mine.in_dim, j), sp_from_lane(mine.vec, j) != 0, lane, wr[j]);
  }
}

template <int PF>
__device__ __forceinline__ void sp_heads(const cips3d_linear_desc* __restrict__ table, int n_desc, int total_rows, int q,
                                         int nwaves, const float* base, int sd, const float* xs, int B, int dim, int lane,
                                         const SpHead& mine, const spv4 (&wr)[PF][2], unsigned* dbg = nullptr,
                                         long long t0 = 0) {
  const int first = q * PF, stride = 1;
#ifdef CIPS3D_SP_STAMPS
  SP_DRAIN();
  if (dbg) dbg[0] = (unsigned)(wall_clock64() - t0);
#endif
  // the rows' partial sums first, then their (independent) reductions level by level: six shuffle latencies for all rows
  // instead of six per row
  for (int b = 0; b < B; ++b) {
    float acc[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      acc[j] = 0.f;
      if (first + j * stride < total_rows) {
        const int in_dim = sp_from_lane(mine.in_dim, j), xoff = sp_from_lane(mine.xoff, j);
        const float* x = xs + b * dim + xoff;
        if (sp_from_lane(mine.vec, j)) {
          int i = lane * 4;
          if (i < in_dim) acc[j] = sp_fma4(wr[j][0], x + i, acc[j]);
          if (i + 256 < in_dim) acc[j] = sp_fma4(wr[j][1], x + i + 256, acc[j]);
          const float* wrow = sp_from_lane(mine.wrow, j);
          for (i += 512; i < in_dim; i += 256) acc[j] = sp_fma4(*sp_g(reinterpret_cast<const spv4*>(wrow + i)), x + i, acc[j]);
        } else {
          const float* wrow = sp_from_lane(mine.wrow, j);
          for (int i = lane; i < in_dim; i += 64) acc[j] = fmaf(sp_g(wrow)[i], *SP_LDS(float, x + i), acc[j]);
        }
      }
    }
#ifdef CIPS3D_SP_STAMPS
    SP_DRAIN();
    if (dbg) dbg[1] = (unsigned)(wall_clock64() - t0);
#endif
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int j = 0; j < PF; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    }
#ifdef CIPS3D_SP_STAMPS
    SP_DRAIN();
    if (dbg) dbg[2] = (unsigned)(wall_clock64() - t0);
#endif
    // lane j finishes row j
    float mine_acc = 0.f;
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (lane == j) mine_acc = acc[j];
    if (lane < PF && first + lane * stride < total_rows) {
      const float y = fmaf(mine_acc, mine.ws, mine.b);
      sp_g(mine.out)[(int64_t)b * mine.ostr] = fmaf(y, mine.os, mine.oh);
    }
  }
  for (int h = nwaves * PF + q; h < total_rows; h += nwaves) {      // (more rows than slots: fetched when they are due)
    const SpHead s = sp_head_of(table, n_desc, h, base, sd, lane);
    sp_head_row(s, xs, B, dim, lane);
  }
}

// one chain row of layer l for this wave (wave-uniform): which chain, which row
constexpr int SP_AW = (int)(sizeof(cips3d_linear_args) / 4);      // words of one layer's arguments
static_assert(sizeof(cips3d_linear_args) % 4 == 0, "");
__device__ __forceinline__ cips3d_linear_args sp_args_of(const unsigned* lds_args, int idx) {
  cips3d_linear_args A;
  unsigned w[SP_AW];
#pragma unroll
  for (int k = 0; k < SP_AW; ++k) w[k] = lds_args[idx * SP_AW + k];
  __builtin_memcpy(&A, w, sizeof(A));
  return A;
}
struct SpRow { cips3d_linear_args A; int row, is_r, vec; const float* wrow; };
__device__ __forceinline__ void sp_row_set(SpRow& r, const cips3d_linear_args& A, int row, int is_r) {
  r.is_r = is_r;
  r.A = A;
  r.row = row;
  r.wrow = A.W + (int64_t)row * A.in_dim;
  r.vec = (A.in_dim % 4 == 0) && sp_al16(r.wrow) && sp_al16(A.x) && (A.x_stride % 4 == 0);
}
__device__ __forceinline__ bool sp_row_of(const unsigned* lds_args, int nr, int nd, int l, int c, SpRow& r) {
  const int rows_r = l < nr ? (int)lds_args[l * SP_AW + offsetof(cips3d_linear_args, out_dim) / 4] : 0;
  const int rows_d = l < nd ? (int)lds_args[(SP_MAXL + l) * SP_AW + offsetof(cips3d_linear_args, out_dim) / 4] : 0;
  if (c >= rows_r + rows_d) return false;
  r.is_r = c < rows_r;
  r.A = sp_args_of(lds_args, r.is_r ? l : SP_MAXL + l);
  r.row = r.is_r ? c : c - rows_r;
  r.wrow = r.A.W + (int64_t)r.row * r.A.in_dim;
  r.vec = (r.A.in_dim % 4 == 0) && sp_al16(r.wrow) && sp_al16(r.A.x) && (r.A.x_stride % 4 == 0);
  return true;
}

__global__ void __launch_bounds__(256) style_phase_kernel(StylePhaseArgs a) {
  // [parity][chain][B][dim] staged inputs | [layer][wave][dim] this wave's row of every chain layer | [layer][wave][2] its
  // scaled bias and truncation mean | the layers' arguments (indexed by the stage counter: a by-value struct indexed at run
  // time would be copied to scratch, and reads through a pointer to the argument block are vector loads of a round trip each)
  extern __shared__ __attribute__((aligned(16))) float sp_x[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int g = blockIdx.x;
  if (g >= a.G) {            // passengers: the call's draw, the range workspace
    const int e = g - a.G;
    if (e < a.blocks_rng) {
      const long long tt = a.job.t0 + (long long)e * 256 + t;
      if (tt < a.job.t1)
        rng_fill_thread(a.job.seed_lo, a.job.seed_hi, a.job.base, a.job.normal, a.job.n_normal, a.job.uniform, a.job.n_uniform, tt);
    } else {
      for (int i = (e - a.blocks_rng) * 256 + t; i < a.zero_n; i += a.blocks_zero * 256) a.zero_ptr[i] = 0.f;
    }
    return;
  }
  const int B = a.B, dim = a.dim, G = a.G, nr = a.nr, nd = a.nd;
  const int nl = nr > nd ? nr : nd;
  float* const sp_w = sp_x + 4 * B * dim;
  float* const sp_b = sp_w + nl * 4 * dim;
  unsigned* const sp_a = reinterpret_cast<unsigned*>(sp_b + nl * 4 * 2);
  const unsigned tag = __hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const long long t_begin = wall_clock64();
#ifdef CIPS3D_SP_STAMPS
  const long long c_begin = clock64();
#endif
  const long long deadline = t_begin + SP_TIMEOUT;
  // the layers' arguments on their way to LDS (written below, behind the requests that must not wait for them)
  static_assert(offsetof(StylePhaseArgs, r) == 0 && offsetof(StylePhaseArgs, d) == SP_MAXL * sizeof(cips3d_linear_args), "");
  static_assert(2 * SP_MAXL * SP_AW <= 2 * 256, "");
  const unsigned* ka = (const unsigned*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned kv[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) kv[k] = (t + k * 256 < 2 * SP_MAXL * SP_AW) ? sp_g(ka)[t + k * 256] : 0u;

  // ---- this wave's row of every chain layer: requested now, straight into LDS (LDS-DMA: no registers, no wait until use);
  // lane l fetches the bias and truncation mean of layer l's row.  (Unrolled: here the layer index is a constant and the
  // arguments come from the kernel's scalar registers.)
  {
    float bv = 0.f, mv = 0.f;
#pragma unroll
    for (int l = 0; l < SP_MAXL; ++l) {
      if (l >= nl) continue;
      const int rows_r = l < nr ? a.r[l].out_dim : 0, rows_d = l < nd ? a.d[l].out_dim : 0;
      const int c = g + G * w;
      if (c >= rows_r + rows_d) continue;
      const bool is_r = c < rows_r;
      const int row = is_r ? c : c - rows_r;
      const float* W = is_r ? a.r[l].W : a.d[l].W;
      const float* x = is_r ? a.r[l].x : a.d[l].x;
      const float* bias = is_r ? a.r[l].bias : a.d[l].bias;
      const float* tm = is_r ? a.r[l].trunc_mean : a.d[l].trunc_mean;
      const int in_dim = is_r ? a.r[l].in_dim : a.d[l].in_dim;
      const int64_t xst = is_r ? a.r[l].x_stride : a.d[l].x_stride;
      const float bsc = is_r ? a.r[l].b_scale : a.d[l].b_scale;
      const float* wrow = W + (int64_t)row * in_dim;
      if ((in_dim % 4 == 0) && sp_al16(wrow) && sp_al16(x) && (xst % 4 == 0)) {
        for (int k = 0; k * 256 < in_dim; ++k)
          if (lane * 4 + k * 256 < in_dim)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow + k * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(sp_w + (l * 4 + w) * dim + k * 256), 16, 0, 0);
      }
      if (lane == l) {
        if (bias) bv = sp_g(bias)[row] * bsc;
        if (tm) mv = sp_g(tm)[row];
      }
    }
    if (lane < nl) { sp_b[(lane * 4 + w) * 2 + 0] = bv; sp_b[(lane * 4 + w) * 2 + 1] = mv; }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (t + k * 256 < 2 * SP_MAXL * SP_AW) sp_a[t + k * 256] = kv[k];
  const int hq = g * 4 + w, hwaves = G * 4;      // a wave's head rows are consecutive: its results leave in one store
  SpHead hf{}, hm{};
  spv4 wf[SP_PF_FILM][2], wm[SP_PF_MOD][2];

  for (int l = 0; l <= nl; ++l) {
    float* xr = sp_x + ((l & 1) * 2 + 0) * B * dim;
    float* xd = sp_x + ((l & 1) * 2 + 1) * B * dim;
    // ---- this stage's inputs: the call's z, or the previous layers' granules
    if (l == 0) {
      for (int c = 0; c < 2; ++c) {
        if ((c ? nd : nr) < 1) continue;
        const cips3d_linear_args& A = c ? a.d[0] : a.r[0];      // (the LDS copy of the arguments is not complete yet)
        float* xs = c ? xd : xr;
        for (int b = 0; b < B; ++b)
          for (int i = t; i < A.in_dim; i += 256) xs[b * dim + i] = sp_g(A.x)[(int64_t)b * A.x_stride + i];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the LDS-DMA'd rows: the compiler does not track them)
    } else {
      if (l == 1) {     // the head rows' weights: requested once layer 0 is on its way, in the shadow of the first hand-over
        sp_heads_fetch<SP_PF_FILM>(a.film, a.film_n, a.film_rows, hq, a.styles_r, a.sd_r, lane, hf, wf);
        sp_heads_fetch<SP_PF_MOD>(a.mod, a.mod_n, a.mod_rows, hq, a.styles_d, a.sd_d, lane, hm, wm);
      }
      if (l - 1 < nr)
        sp_gather(a.xch + (int64_t)((l - 1) * 2 + 0) * B * dim, B, (int)sp_a[(l - 1) * SP_AW + offsetof(cips3d_linear_args, out_dim) / 4],
                  dim, xr, tag, deadline, a.sync, t);
      if (l - 1 < nd)
        sp_gather(a.xch + (int64_t)((l - 1) * 2 + 1) * B * dim, B,
                  (int)sp_a[(SP_MAXL + l - 1) * SP_AW + offsetof(cips3d_linear_args, out_dim) / 4], dim, xd, tag, deadline, a.sync, t);
    }
    __syncthreads();
    SP_STAMP(4 + 2 * l);
    // ---- a chain's W+ is complete in every workgroup's LDS: its plain copy for the later kernels (every style slot) leaves
    // in whole rows from a few workgroups -- one dword per producing wave and slot, 32 writers to a 128-byte line from
    // all eight L2s, took microseconds to be acknowledged.  (The layers in between only exist as granules.)
    for (int c = 0; c < 2; ++c) {
      if (l != (c ? nd : nr) || l < 1) continue;
      const cips3d_linear_args A = sp_args_of(sp_a, (c ? SP_MAXL : 0) + l - 1);
      const float* xs = c ? xd : xr;
      for (int p = g; p < B * A.out_repeat; p += G) {
        const int b = p / A.out_repeat, rr = p % A.out_repeat;
        float* o = A.out + (int64_t)b * A.out_stride + rr * A.out_repeat_stride;
        for (int i = t; i < A.out_dim; i += 256) sp_g(o)[i] = *SP_LDS(float, xs + b * dim + i);
      }
    }
    // ---- rows of layer l (both chains)
    if (l < nl) {
      bool first = true;
      SpRow r;
      for (int c = g + G * w; sp_row_of(sp_a, nr, nd, l, c, r); c += 4 * G, first = false) {
        const cips3d_linear_args& A = r.A;
        const float bterm = first ? sp_b[(l * 4 + w) * 2 + 0] : (A.bias ? sp_g(A.bias)[r.row] * A.b_scale : 0.f);
        const float mterm = first ? sp_b[(l * 4 + w) * 2 + 1] : (A.trunc_mean ? sp_g(A.trunc_mean)[r.row] : 0.f);
        const float* xs = r.is_r ? xr : xd;
        unsigned long long* gr = a.xch + (int64_t)(l * 2 + (r.is_r ? 0 : 1)) * B * dim + r.row;
        for (int b = 0; b < B; ++b) {
          // (the dot's weight operand: the row's LDS copy -- same values, same order of products)
          const float acc = (first && r.vec) ? sp_dot<true>(sp_w + (l * 4 + w) * dim, xs + b * dim, A.in_dim, true, lane)
                                             : sp_dot<false>(r.wrow, xs + b * dim, A.in_dim, r.vec != 0, lane);
          float nrm = 1.f;
          if (A.pixelnorm) {
            float s = 0.f;
            for (int i = lane; i < A.in_dim; i += 64) { const float v = *SP_LDS(float, xs + b * dim + i); s = fmaf(v, v, s); }
            s = wave_sum(s);
            nrm = rsqrtf(s / (float)A.in_dim + 1e-8f);
          }
          if (lane == 0) {
            const float y = linear_post_m(A, bterm, acc, nrm, mterm);
            __hip_atomic_store(sp_g(gr + (int64_t)b * dim), ((unsigned long long)tag << 32) | __float_as_uint(y), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    SP_STAMP(5 + 2 * l);
  }
  // ---- heads: stage n of a chain staged that chain's W+ in the buffer of its parity, and no later stage wrote there
  if (a.mod_rows > 0)
    sp_heads<SP_PF_MOD>(a.mod, a.mod_n, a.mod_rows, hq, hwaves, a.styles_d, a.sd_d, sp_x + ((a.nd & 1) * 2 + 1) * B * dim, B,
                        dim, lane, hm, wm, (g == 0 && t == 0) ? a.sync + 21 : nullptr, t_begin);
  SP_STAMP(20);
  if (a.film_rows > 0)
    sp_heads<SP_PF_FILM>(a.film, a.film_n, a.film_rows, hq, hwaves, a.styles_r, a.sd_r, sp_x + ((a.nr & 1) * 2 + 0) * B * dim, B,
                         dim, lane, hf, wf);
  // every workgroup has read the generation long ago (this workgroup has seen the last layer's rows of all of them)
  SP_STAMP(3);
#ifdef CIPS3D_SP_STAMPS
  if (g == 0 && t == 0) a.sync[24] = (unsigned)(clock64() - c_begin);
#endif
  if (g == 0 && t == 0) __hip_atomic_store(a.sync, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// library-internal (forward.hip): the style phase as one launch; CIPS3D_E_UNSUPP when the shapes do not fit it
int cips3d_style_phase_one_launch(const cips3d_linear_args* ar, int nr, const cips3d_linear_args* ad, int nd,
                                  const cips3d_linear_desc* film, int film_n, int film_rows, const cips3d_linear_desc* mod,
                                  int mod_n, int mod_rows, const float* styles_r, int sd_r, const float* styles_d, int sd_d,
                                  void* xch, void* sync, int dim, const cips3d_rng_job* job, float* zero_ptr, int zero_n,
                                  void* stream) {
  if (nr < 1 || nd < 1 || nr > SP_MAXL || nd > SP_MAXL || !xch || !sync || dim <= 0 || dim % 4 != 0 || sd_r <= 0 || sd_d <= 0)
    return CIPS3D_E_UNSUPP;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      return CIPS3D_E_UNSUPP;
    cus = n;
  }
  StylePhaseArgs a{};
  const int B = ar[0].B;
  if (B < 1 || B > 8) return CIPS3D_E_UNSUPP;
  int G = cus;
  for (int l = 0; l < (nr > nd ? nr : nd); ++l) {
    int rows = 0;
    if (l < nr) { a.r[l] = ar[l]; rows += ar[l].out_dim; if (ar[l].in_dim > dim || ar[l].out_dim > dim || ar[l].B != B) return CIPS3D_E_UNSUPP; }
    if (l < nd) { a.d[l] = ad[l]; rows += ad[l].out_dim; if (ad[l].in_dim > dim || ad[l].out_dim > dim || ad[l].B != B) return CIPS3D_E_UNSUPP; }
    if (rows < G) G = rows;
  }
  if (G < 1 || sd_r > dim || sd_d > dim) return CIPS3D_E_UNSUPP;
  a.nr = nr; a.nd = nd; a.B = B; a.G = G;
  a.film = film; a.film_n = film_n; a.film_rows = film ? film_rows : 0;
  a.mod = mod; a.mod_n = mod_n; a.mod_rows = mod ? mod_rows : 0;
  a.styles_r = styles_r; a.styles_d = styles_d; a.sd_r = sd_r; a.sd_d = sd_d;
  a.xch = reinterpret_cast<unsigned long long*>(xch);
  a.sync = reinterpret_cast<unsigned*>(sync);
  a.dim = dim;
  if (job && job->t1 > job->t0) { a.job = *job; a.blocks_rng = (int)ceil_div<long long>(job->t1 - job->t0, 256); }
  if (zero_ptr && zero_n > 0) {
    a.zero_ptr = zero_ptr; a.zero_n = zero_n;
    a.blocks_zero = ceil_div(zero_n, 1024);
    if (a.blocks_zero > 64) a.blocks_zero = 64;
  }
  const int nl = nr > nd ? nr : nd;
  const size_t lds = ((size_t)4 * B * dim + (size_t)nl * 4 * dim + (size_t)nl * 4 * 2) * sizeof(float) + 2 * SP_MAXL * sizeof(cips3d_linear_args);
  if (lds > 150 * 1024) return CIPS3D_E_UNSUPP;
  static size_t lds_max = 64 * 1024;
  if (lds > lds_max) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(style_phase_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024)) != hipSuccess)
      return CIPS3D_E_UNSUPP;
    lds_max = 150 * 1024;
  }
  hipLaunchKernelGGL(style_phase_kernel, dim3(G + a.blocks_rng + a.blocks_zero), dim3(256), lds, as_stream(stream), a);
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
