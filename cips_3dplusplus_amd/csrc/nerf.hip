// NeRF-style volume renderer of CIPS-3D++ on gfx950:
//   get_rays_in_world -> get_z_vals (offset sampling) -> get_points/normalize_points
//   (reference cips3d/nerf_utils.py:18-170) -> FiLM-SIREN point MLP (cips3d/volume_renderer.py:39-160)
//   -> volume_integration (cips3d/nerf_utils.py:230-338), fused in ONE kernel; point activations never
//   leave the register file.
//
// Work decomposition.  A *task* = (view b, group of 16 rays, chunk c of the ray's samples); one
// wavefront per task, eight tasks per 512-thread workgroup (two waves per SIMD).  Lane l works on ray
// (l & 15); the four 16-lane quarters (qd = l >> 4) hold the four k-slices that
// v_mfma_f32_16x16x4_f32 consumes.  The wave walks its chunk's samples one by one; per sample it
// evaluates the whole MLP for its 16 points:
//
//   layer 0 (3 -> H)          VALU, written straight into MFMA "D layout"
//   layers 1..D-1 (H -> H)    Y^T[o][p] = sum_k W[o][k] X^T[k][p] on the matrix cores: A = W tile (from LDS),
//                             B = X^T, D = Y^T.  D has the point on the lane and 4 consecutive hidden
//                             units in its 4 accumulator registers, which (two tiles at a time) is exactly the
//                             B-operand layout of the next layer -> no LDS round trip, no shuffles for activations.
//                             Arithmetic: fp32-equivalent SPLIT-fp16 products (see "Split-fp16 MFMA" below).
//   view layer (H+3 -> H)     3 view-direction terms pre-loaded into the accumulator, then MFMA; the
//                             finished tile is folded straight into the feature accumulators
//   sigma / rgb heads         per-lane dot over its registers + two cross-quarter adds
//   compositing               per-lane running transmittance over the chunk (no scan needed because a
//                             lane owns a ray), partial sums kept in registers
//
// Register budget per lane (H = 256): X 64 + Y 64 + feature accumulators 64 + scalars => < 256, so two
// waves share a SIMD: one wave's sine epilogue (VALU) runs under the other's MFMAs.
//
// Split-fp16 MFMA.  The fp32 matrix instruction (v_mfma_f32_16x16x4_f32) runs at 1/16 of the fp16 rate.  Every operand
// is therefore held as an unevaluated sum of two fp16 numbers, x = x_hi + x_lo (x_hi = fp16(x), x_lo = fp16(x - x_hi):
// 22 significant bits, the same 4 bytes per value as an fp32), weights pre-scaled per layer by a power of two into
// fp16's normal range, and a product w x is accumulated IN FP32 as the three exact fp16 x fp16 products
// w_hi x_hi + w_hi x_lo + w_lo x_hi on v_mfma_f32_16x16x32_f16: 3 instructions of 16 cycles per 16x16x32 block instead
// of 8 of 32 cycles.  The dropped term and the operand representation error are both ~2^-22 relative per PRODUCT, below
// the rounding error fp32 accumulation itself makes on the SUM: against an fp64 run the outputs sit exactly where the
// plain-fp32 path sits (tools/split_probe.py).  The power-of-two scale is undone for free in the FiLM
// multiplier (gamma * 2^-s).
//
// Weights: every CU streams the same (D * H*H) packed floats from L2 through a 2-slot LDS ring, one
// slab = TPS o-tiles (16*TPS rows x H) per step, fetched with global_load_lds (LDS-DMA) while the
// previous slab is being multiplied.  The pack kernel stores W in the exact order the lanes read
// their A fragments, so the LDS image is lane-linear and every ds_read_b128 is conflict-free.
//
// Chunk partials (T, sum w*feat, ...) are combined in sample order by nerf_finish (compositing is
// associative: S = S_a + T_a * S_b, T = T_a * T_b), which also emits the NCHW feature map.
//
// Roofline: MFMA-bound on paper (flops/point = 2*3*H + (D-1)*2*H^2 + 2*(H+3)*H + 2*H*4, executed as 3 fp16 products per
// fp32 product); with the split form the sine epilogues (VALU) and the LDS fragment reads are of the same order as the
// matrix time.  HBM traffic is the partials only (n_chunks * (H+8) * 4 B per ray).
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include "common.h"
#include "nerf_mlp.h"

#ifdef CIPS3D_STAMPS
#ifndef CIPS3D_STAMP_WAVE
#define CIPS3D_STAMP_WAVE 0      // the wave whose phases are summed (0..3 early-epilogue waves, 4..7 late)
#endif
// Diagnostic build only (never in the shipped library): per-phase cycle sums of one wave of every workgroup, accumulated in
// scalar registers and flushed with one batch of atomics at the end of the kernel (stamps that did an atomic each slowed the
// kernel 3x and distorted the barrier waits).
struct StampState { unsigned long long t_prev; unsigned long long acc[16]; };
#define STAMP_PARAM , StampState& stamps_
#define STAMP_ARG , stamps_
__device__ unsigned long long g_nerf_stamps[16];
#define STAMP(i)                                                                         \
  do {                                                                                   \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                          \
    stamps_.acc[i] += t_ - stamps_.t_prev;                                               \
    stamps_.t_prev = t_;                                                                 \
  } while (0)
#define STAMP_FLUSH()                                                                    \
  do {                                                                                   \
    if (wave == CIPS3D_STAMP_WAVE && lane == 0)                                          \
      for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_nerf_stamps[i_], stamps_.acc[i_]);    \
  } while (0)
extern "C" int cips3d_debug_read_stamps(unsigned long long* out16) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_nerf_stamps), 128);
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_nerf_stamps), z, 128);
  return 0;
}
#else
#define STAMP(i)
#define STAMP_FLUSH()
#define STAMP_PARAM
#define STAMP_ARG
#endif

#ifdef CIPS3D_CLOCK
// Diagnostic build only: candidate sine forms side by side (tools/sin_probe.py compares them with fp64)
__device__ static inline float sin_hw_reduced(float x) {
  // exact Cody-Waite reduction by 2 pi (k * 2PI_HI absorbed by the FMA), then the hardware sine on r / (2 pi) in [-0.5, 0.5]
  const float INV_2PI = 0.159154943091895336f;
  const float TWO_PI_HI = 6.28318548202514648f;        // float(2 pi)
  const float TWO_PI_LO = -1.74845553146951715e-7f;    // 2 pi - TWO_PI_HI
  const float k = rintf(x * INV_2PI);
  float r = fmaf(k, -TWO_PI_HI, x);
  r = fmaf(k, -TWO_PI_LO, r);
  return __builtin_amdgcn_sinf(r * INV_2PI);
}
__global__ void sin_probe_kernel(const float* __restrict__ x, float* __restrict__ ya, float* __restrict__ yh, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { ya[i] = sin_accurate(x[i]); yh[i] = sin_hw_reduced(x[i]); }
}
extern "C" int cips3d_debug_sin(const float* x, float* ya, float* yh, int n, void* stream) {
  hipLaunchKernelGGL(sin_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), x, ya, yh, n);
  return cips3d_launch_status();
}
// Diagnostic build only: the shader clock the render kernel actually runs at = d(s_memtime) / d(s_memrealtime) x 100 MHz,
// one stamp pair around the whole kernel per workgroup (MI355X_MICROARCH.md, DVFS give-back item 6).  The sums go to a
// buffer nothing else reads.
__device__ unsigned long long g_nerf_clock[2];
__device__ unsigned long long g_nerf_clock_wg[2 * 4096];      // the last launch's pair per workgroup (median over workgroups)
extern "C" int cips3d_debug_read_clock(unsigned long long* out2) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_nerf_clock), 16);
  unsigned long long z[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_nerf_clock), z, 16);
  return 0;
}
extern "C" int cips3d_debug_read_clock_wg(unsigned long long* out, int n_wg) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nerf_clock_wg), (size_t)16 * (n_wg < 4096 ? n_wg : 4096));
  return 0;
}
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// weight packing (split-fp16).  o-tile t = 16 output units; k-block m = 32 input units = one v_mfma_f32_16x16x32_f16.
//   packed[l][t][m][plane][lane][j] (fp16) = plane(hi|lo) of 2^s_l * W_l[t*16 + (lane&15)][32 m + 16 (j>>2) + 4 (lane>>4) + (j&3)]
// The k order inside a block follows the D layout of the previous layer's accumulators: lane quarter q holds units
// 4q..4q+3 of each 16-unit tile, so the 8 fragment elements of quarter q are units 4q+r of tile 2m (j = r) and of tile
// 2m+1 (j = 4 + r).  One tile = (H/32) blocks x 2 planes x 1 KiB = 16*H*4 bytes, the size of the fp32 tile it replaces.
// The per-layer scales (2^s_l, 2^-s_l) follow the matrices: packed[D*H*H + 2 l], [.. + 1].
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int64_t nerf_packed_floats(int H, int D) { return (int64_t)D * H * H + 2 * 64; }

// one workgroup per layer: s = power of two with max |2^s W| in [512, 1024) (1 for an all-zero matrix)
__global__ void __launch_bounds__(256) nerf_scale_kernel(const float* __restrict__ w_hidden, const float* __restrict__ w_view,
                                                         float* __restrict__ packed, int H, int D) {
  __shared__ float s_max[4];
  const int l = blockIdx.x;
  float m = 0.f;
  if (l < D - 1) {
    const float* w = w_hidden + (int64_t)l * H * H;
    for (int i = threadIdx.x; i < H * H; i += 256) m = fmaxf(m, fabsf(w[i]));
  } else {
    for (int i = threadIdx.x; i < H * H; i += 256) m = fmaxf(m, fabsf(w_view[(int64_t)(i / H) * (H + 3) + (i % H)]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    int e = 0;
    if (m > 0.f && m < 3.0e38f) {
      frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
      e = 10 - e;                    // 2^e' * m in [512, 1024)
      if (e > 100) e = 100;
      if (e < -100) e = -100;
    }
    float* sc = packed + (int64_t)D * H * H + 2 * l;
    sc[0] = ldexpf(1.f, e);
    sc[1] = ldexpf(1.f, -e);
  }
}

__global__ void __launch_bounds__(256) nerf_pack_kernel(const float* __restrict__ w_hidden,
                                                        const float* __restrict__ w_view,
                                                        float* __restrict__ packed, int H, int D) {
  const int64_t per_layer = (int64_t)H * H;
  const int64_t total = per_layer * D;             // one thread per weight: writes its hi and lo halves
  _Float16* out = reinterpret_cast<_Float16*>(packed);
  const float* scales = packed + total;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i / per_layer);
    int64_t rem = i - l * per_layer;               // index inside the layer in (t, m, lane, j) order
    const int tile_w = 16 * H;                     // weights per o-tile
    const int t = (int)(rem / tile_w);
    rem -= (int64_t)t * tile_w;
    const int m = (int)(rem / 512);                // 512 weights per k-block (64 lanes x 8)
    const int lane = (int)((rem % 512) / 8);
    const int j = (int)(rem % 8);
    const int o = t * 16 + (lane & 15);
    const int k = 32 * m + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    float v;
    if (l < D - 1) v = w_hidden[(int64_t)l * per_layer + (int64_t)o * H + k];
    else           v = w_view[(int64_t)o * (H + 3) + k];
    v *= scales[2 * l];
    _Float16 hi, lo;
    cips3d_split16(v, hi, lo);
    // fp16 units: layer base 2*per_layer, tile 2*tile_w, block m: [plane][lane][8]
    _Float16* blk = out + 2 * ((int64_t)l * per_layer + (int64_t)t * tile_w) + (int64_t)m * 1024;
    blk[lane * 8 + j] = hi;
    blk[512 + lane * 8 + j] = lo;
  }
}

// exact-fp32 weight stream: packed32[l][t][m][half][lane][r] = W_l[16 t + (lane & 15)][32 m + 16 half + 4 (lane >> 4) + r] -- the
// piece positions of the split stream ((o-tile t, k-block m, plane) <-> (t, m, half)), each piece the 16 x 16 fp32 block whose
// lane-linear 16 bytes are the lane's A operands of four consecutive v_mfma_f32_16x16x4_f32 k-steps; scales (1, 1) per layer
// behind the matrices, where the split stream keeps its powers of two
__global__ void __launch_bounds__(256) nerf_pack32_kernel(const float* __restrict__ w_hidden, const float* __restrict__ w_view,
                                                          float* __restrict__ packed, int H, int D) {
  const int64_t per_layer = (int64_t)H * H, total = per_layer * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total + 2 * D; i += (int64_t)gridDim.x * blockDim.x) {
    if (i >= total) { packed[i] = 1.f; continue; }
    const int l = (int)(i / per_layer);
    int64_t rem = i - l * per_layer;                 // (t, m, half, lane, r)
    const int t = (int)(rem / (16 * H));
    rem -= (int64_t)t * 16 * H;
    const int m = (int)(rem / 512), half = (int)((rem % 512) / 256), lane = (int)((rem % 256) / 4), r = (int)(rem % 4);
    const int o = t * 16 + (lane & 15), k = 32 * m + 16 * half + 4 * (lane >> 4) + r;
    packed[i] = l < D - 1 ? w_hidden[(int64_t)l * per_layer + (int64_t)o * H + k] : w_view[(int64_t)o * (H + 3) + k];
  }
}

// One MFMA layer for the wave's 16 points (split-fp16 products, fp32 accumulation; see the file header).
//   VIEW = false: Y = sin(gamma' * (W' X) + c)            gamma' = gamma 2^-s, W' = 2^s W (s_film holds gamma')
//   VIEW = true : f = sin(gamma' * (W' X + Wd' v) + c);  FA += w * f;  rgb head partial sums += Wc f
//   last (hidden layers): Y is h_D, the sigma head's partial sum  sdf_acc += Ws . Y  is taken from the epilogue's fp32 values
// The caller guarantees slab `seq` is resident in slot (seq & 1); every slab step prefetches seq+1
// while multiplying and ends with wait + barrier.
// F32: the exact-fp32 instantiation (Generator.set_precision("fp32_exact"); cips3d_nerf_params.packed32): the same slabs, ring
// and register images hold fp32 -- a 1 KiB piece is a 16 x 16 block of W where the split stream has one plane of a 16 x 32
// block, (Xh[m], Xl[m]) are the bits of the fp32 activations of o-tiles 2m, 2m + 1 (the MFMA D layout IS the B operand of
// v_mfma_f32_16x16x4_f32: k-step r of k-quarter q <-> unit 16 T + 4 q + r), eight 32-cycle MFMAs per (tile, k-block) where the
// split kernel issues three 16-cycle ones.  Two waves per SIMD as in the split kernel: one wave's sine epilogue runs under
// the other's matrix block (a one-wave-per-SIMD form of the same arithmetic measured 252 us; this one: see DESIGN).
__host__ __device__ constexpr int nerf_table_floats(int H, int L) { return L * 2 * H + 10 * H; }

template <bool F32>
__device__ __forceinline__ void put8(const float (&v)[8], h8& hi, h8& lo) {
  if constexpr (F32) {
    hi = __builtin_bit_cast(h8, f32x4{v[0], v[1], v[2], v[3]});
    lo = __builtin_bit_cast(h8, f32x4{v[4], v[5], v[6], v[7]});
  } else {
    split8(v, hi, lo);
  }
}

template <int NT, int TPS, bool VIEW, bool STASH, bool F32 = false>
__device__ __forceinline__ void mfma_layer(const h8 (&Xh)[NT / 2], const h8 (&Xl)[NT / 2], h8 (&Yh)[NT / 2], h8 (&Yl)[NT / 2],
                                           float (&FA)[NT * 4], float wgt, float (&chead)[3], float& sdf_acc, bool last,
                                           Ring& ring, const float* film_l, const float* s_wd, const float* s_wc,
                                           const float* s_ws, float vx, float vy, float vz, float* stash_l, int wave, int lane, int q4o STAMP_PARAM) {
  constexpr int H = NT * 16;
  constexpr int TILE = 16 * H;          // floats (= 4-byte hi/lo pairs) of one o-tile's A fragments
  constexpr int SLAB = TILE * TPS;
  constexpr int STEPS = NT / TPS;
  constexpr int R = TPS * 4;            // output values produced per step
  constexpr int MB = NT / 2;            // 32-unit k-blocks of the layer input
  constexpr int BPS = TPS / 2;          // k-blocks of the NEXT layer's input a step completes
  static_assert(TPS % 2 == 0 && NT % TPS == 0, "a slab step must complete whole 32-unit blocks");
  // The slab steps of a layer are fully unrolled (with 96 fp16 MFMAs per step instead of 256 fp32 ones the whole kernel is
  // ~40 KB of code): the step index is a constant, results go straight to their final registers (the rolled loop of the
  // fp32 kernel had to rotate Yh / Yl / FA by 48 moves per step; unrolling measured 107.6 -> 101.0 us).
  // Epilogue stagger.  Waves w and w + WAVES/2 share a SIMD and meet at every slab barrier; with the barrier after the
  // FiLM/sine epilogue both would run its VALU instructions together while the matrix pipe idles.  The upper half of the
  // waves takes the step barrier BEFORE its epilogue instead, which then overlaps the partner wave's next-step MFMAs.  Either
  // position is after this wave's last read of slot (seq&1) and before its next stage_slab into it.
  // In-kernel stamps of this scheme (tools/run_kernel.py nerf on a -DCIPS3D_STAMPS build): the two matrix blocks of a SIMD
  // run together (~2.9k cycles for 192 MFMAs = the pipe is full), then the lower wave's epilogue (~1.3k) with the pipe idle;
  // a step is ~6.4k cycles.  Tried: a true half-period offset (two barriers per step, the upper half one barrier behind, so
  // that one wave's matrix block always faces the other's epilogue): correct, but 127 us instead of 97 -- a matrix block
  // that has the SIMD to itself is bound by its own A-fragment reads (8 x [8 ds_read_b128 -> wait -> 12 MFMAs]; the second
  // wave is what hides that latency today), and double-buffering the fragments needs 32 registers the kernel does not have.
  const bool late_epilogue = __builtin_amdgcn_readfirstlane(wave) >= WAVES / 2;
#pragma unroll
  for (int sl = 0; sl < STEPS; ++sl) {
    if (ring.seq + 1 < ring.seq_end) {
      const int nxt = (ring.seq + 1) % ring.per_sample;
      stage_slab<SLAB>(ring.packed + (int64_t)nxt * SLAB, ring.lds + ((ring.seq + 1) & 1) * SLAB, wave, lane);
    }
    STAMP(8);    // (in-layer stamps: the phase before the first one of a layer is charged to slot 12 / 11 of the previous step)
    const float* slab = ring.lds + (ring.seq & 1) * SLAB;
    const int o_base = sl * (TPS * 16) + q4o;          // this lane's first output unit of the step
    f32x4 acc[TPS];
    // (opaque per step: as loop invariants the operand copies of vx, vy, vz were hoisted out of the sample loop and spilled)
    float vxo = vx, vyo = vy, vzo = vz;
    if (VIEW) asm volatile("" : "+v"(vxo), "+v"(vyo), "+v"(vzo));
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int o4 = o_base + tt * 16;
      if (VIEW) {       // view-direction columns (pre-scaled by 2^s at staging) straight into the accumulator
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_wd + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_wd + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_wd + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[tt][i] = fmaf(wz[i], vzo, fmaf(wy[i], vyo, wx[i] * vxo));
      } else {
        acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // Matrix block, software-pipelined over half k-blocks.  History, all measured: 8 reads -> wait -> 12 MFMAs per k-block
    // (the form of the fp32 kernel) left each wave bound by 8 LDS round trips per step -- ~3.2k cycles for 1.5k of its own
    // matrix work (in-kernel stamps) -- 92.0 us; the next half's 4 reads requested before the current half's 6 MFMAs, through
    // the compiler (which drains lgkmcnt to 0 at every wait, the fresh prefetch included) 89.2 us; the wait provoked in FRONT
    // of the next reads (below) 87.3 us, matrix block ~2.0k cycles per step.
    {
      constexpr int HT = TPS / 2;
      h8 fh[2][HT], fl[2][HT];
      auto load_half = [&](int buf, int m, int half) {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          const int tt = half * HT + t;
          fh[buf][t] = *reinterpret_cast<const h8*>(slab + tt * TILE + ((2 * m) * 64 + lane) * 4);
          fl[buf][t] = *reinterpret_cast<const h8*>(slab + tt * TILE + ((2 * m + 1) * 64 + lane) * 4);
        }
      };
      load_half(0, 0, 0);
#pragma unroll
      for (int g = 0; g < 2 * MB; ++g) {
        const int m = g >> 1, half = g & 1, cur = g & 1;
#pragma unroll
        for (int t = 0; t < HT; ++t) asm volatile("" : "+v"(fh[cur][t]), "+v"(fl[cur][t]));
        if (g + 1 < 2 * MB) load_half(cur ^ 1, (g + 1) >> 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (F32) {      // fh / fl: the fp32 fragments of the k-block's two 16-unit halves; k ascending, the bit-exact chain
          const f32x4 x0 = __builtin_bit_cast(f32x4, Xh[m]), x1 = __builtin_bit_cast(f32x4, Xl[m]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < HT; ++t)
              acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fh[cur][t])[r], x0[r],
                                                                        acc[half * HT + t], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < HT; ++t)
              acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, fl[cur][t])[r], x1[r],
                                                                        acc[half * HT + t], 0, 0, 0);
        } else {
#pragma unroll
        for (int t = 0; t < HT; ++t)
          acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cur][t], Xh[m], acc[half * HT + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < HT; ++t)
          acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xl[m], acc[half * HT + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < HT; ++t)
          acc[half * HT + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xh[m], acc[half * HT + t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(9);    // matrix block
    if (late_epilogue) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's piece of slab seq+1 has landed
      __syncthreads();
    }
    STAMP(10);   // late waves: step barrier
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
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = FILM_SIN(fmaf(g4[i], acc[tt][i], c4[i]));
          res[tt * 4 + i] = fmaf(wgt, f, FA[sl * R + tt * 4 + i]);
          chead[0] = fmaf(w0[i], f, chead[0]);
          chead[1] = fmaf(w1[i], f, chead[1]);
          chead[2] = fmaf(w2[i], f, chead[2]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) res[tt * 4 + i] = FILM_SIN(fmaf(g4[i], acc[tt][i], c4[i]));
        if (last) {     // h_D: sigma head partial (volume_renderer.py:148) from the fp32 values, before they are split
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) sdf_acc = fmaf(ws4[i], res[tt * 4 + i], sdf_acc);
        }
      }
    }
    // The sink pass would otherwise move the sines below the barrier that follows (undoing the stagger): make
    // the results opaque here.
#pragma unroll
    for (int k = 0; k < R; ++k) asm volatile("" : "+v"(res[k]));
    if (VIEW) asm volatile("" : "+v"(chead[0]), "+v"(chead[1]), "+v"(chead[2]));
    // fully unrolled steps: the step index is a constant, results go to their final places
    if (VIEW) {
#pragma unroll
      for (int k = 0; k < R; ++k) FA[sl * R + k] = res[k];
    } else {
#pragma unroll
      for (int bb = 0; bb < BPS; ++bb) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = res[(2 * bb) * 4 + j];
        put8<F32>(v8, Yh[sl * BPS + bb], Yl[sl * BPS + bb]);
      }
    }
    STAMP(11);   // epilogue
    // slab seq+1 has landed for every wave before anyone reads it / before slot (seq&1) is reused
    if (!late_epilogue) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)  (expcnt/lgkmcnt untouched)
      __syncthreads();
    }
    STAMP(12);   // early waves: step barrier
    if constexpr (STASH) {
      // differentiable forward: the step's accumulators in register order [sl * TPS + tt][lane][4], for cips3d_nerf_bwd_fused.
      // Issued after the step barrier, so that no store acknowledgement is waited for at it.
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) *reinterpret_cast<f32x4*>(stash_l + ((sl * TPS + tt) * 64 + lane) * 4) = acc[tt];
    }
    ++ring.seq;
  }
}

// XG: explicit-geometry instantiation (compile-time so that the camera-driven hot path keeps its register allocation)
template <int NT, int TPS, bool XG, bool STASH, bool F32 = false>
__global__ void __launch_bounds__(WAVES * 64, 2) nerf_render_kernel(NerfArgs a) {
  static_assert(!(F32 && STASH), "the differentiable forward stashes the split kernel's accumulators");
  constexpr int H = NT * 16;
  constexpr int SLAB = 16 * H * TPS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cips3d_nerf_params& P = a.p;
  const int D = P.depth;
  const int L = D + 1;
  float* ringmem = lds;                      // 2 * SLAB
  float* s_film = ringmem + (a.fuse_finish ? nerf_ring_floats(H, TPS, true) : 2 * SLAB);   // L * 2 * H
  float* s_w0 = s_film + L * 2 * H;          // [3][H]  first-layer weights, transposed
  float* s_wd = s_w0 + 3 * H;                // [3][H]  view-direction columns of the view layer
  float* s_ws = s_wd + 3 * H;                // [H]     sigma head
  float* s_wc = s_ws + H;                    // [3][H]  rgb head

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int qd = lane >> 4;
  const int pl = lane & 15;
#ifdef CIPS3D_CLOCK
  const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef CIPS3D_STAMPS
  StampState stamps_;
  for (int i_ = 0; i_ < 16; ++i_) stamps_.acc[i_] = 0;
  stamps_.t_prev = __builtin_amdgcn_s_memtime();
#endif

  // ---- task decode (b is uniform over the workgroup: tasks_per_view is a multiple of WAVES)
  const int64_t task0 = (int64_t)blockIdx.x * WAVES;
  const int b = (int)(task0 / a.tasks_per_view);
  const int tv = (int)(task0 % a.tasks_per_view) + wave;
  const bool task_ok = tv < a.groups * P.n_chunks;
  const int g = task_ok ? tv / P.n_chunks : 0;
  const int c = task_ok ? tv % P.n_chunks : 0;
  const int S = P.img_size;
  const int R = P.n_rays > 0 ? P.n_rays : S * S;
  const int ray = g * RAYS + pl;
  // the lane's ray re-derived where it is needed after the set-up (sdf / per-point stores, the chunk partial): kept in a
  // register across the sample loop it was the kernel's one scratch spill (8 bytes a lane written by every wave)
  auto ray_again = [&]() -> int {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return g * RAYS + (t & 15);
  };
  const bool ray_ok = task_ok && ray < R;
  const int rayc = ray < R ? ray : R - 1;

  // ---- stage the small per-view tables
  {
    // FiLM table: s_film[l][0][o] = gamma, s_film[l][1][o] = gamma * bias_l[o] + beta, so that
    // sin(gamma * (W x + bias) + beta) = sin(gamma * (W x) + c) costs one FMA per unit.
    // MFMA layers (l >= 1) run on weights pre-scaled by 2^s (packed layer l - 1): their gamma carries 2^-s, the
    // view-direction columns that are pre-loaded into the view layer's accumulator carry 2^s (both exact).
    const float* film_b = P.film + (int64_t)b * L * 2 * H;
    const float* scales = (F32 ? P.packed32 : P.packed) + (int64_t)D * H * H;      // (packed32: all ones)
    const float view_scale = scales[2 * (D - 1)];
    // Every global load of the staging is requested before the first one is waited for: the loops below used to be a chain of
    // ~12 dependent memory round trips (each table's loop: load -> wait -> LDS store, per iteration; L2 is cold at a kernel's
    // start) -- ~10 k cycles of a workgroup's 146 k at the published shape (in-kernel stamps).  Two elements per
    // thread and table are in flight at a time (the whole staging at hidden 256, depth 2).
    constexpr int NTH = WAVES * 64;
    for (int base = 0; base < L * H || base < 3 * H; base += 2 * NTH) {
      float gm[2], bt[2], lb[2], sc[2], w0v[2], wdv[2], wcv[2], wsv = 0.f;
      int fi[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = base + e * NTH + tid;
        fi[e] = i;
        gm[e] = bt[e] = lb[e] = 0.f; sc[e] = 1.f; w0v[e] = wdv[e] = wcv[e] = 0.f;
        if (i < L * H) {
          const int l = i / H, o = i - l * H;
          gm[e] = film_b[(l * 2) * H + o];
          bt[e] = film_b[(l * 2 + 1) * H + o];
          lb[e] = P.layer_bias[i];
          if (l >= 1) sc[e] = scales[2 * (l - 1) + 1];
        }
        if (i < 3 * H) {
          const int k = i / H, o = i - k * H;
          w0v[e] = P.w_first[o * 3 + k];
          wdv[e] = P.w_view[o * (H + 3) + H + k];
          wcv[e] = P.w_rgb[i];
        }
      }
      if (base == 0 && tid < H) wsv = P.w_sigma[tid];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = fi[e];
        if (i < L * H) {
          const int l = i / H, o = i - l * H;
          s_film[(l * 2) * H + o] = (l >= 1 ? gm[e] * sc[e] : gm[e]) * FILM_UNIT;
          s_film[(l * 2 + 1) * H + o] = fmaf(gm[e], lb[e], bt[e]) * FILM_UNIT;
        }
        if (i < 3 * H) {
          s_w0[i] = w0v[e];
          s_wd[i] = wdv[e] * view_scale;
          s_wc[i] = wcv[e];
        }
      }
      if (base == 0 && tid < H) s_ws[tid] = wsv;
    }
    for (int i = tid + NTH; i < H; i += NTH) s_ws[i] = P.w_sigma[i];      // (hidden widths above the workgroup size: none today)
  }

  const float b_sigma = P.b_sigma[0], b_rgb0 = P.b_rgb[0], b_rgb1 = P.b_rgb[1], b_rgb2 = P.b_rgb[2];
  const bool raw_density = __builtin_amdgcn_readfirstlane(P.raw_density) != 0;
  const float sig_beta = raw_density ? 1.f : P.sigmoid_beta[0];

  // ---- ray setup (nerf_utils.py:38-66)
  const float nearv = P.near_[b], farv = P.far_[b];
  // explicit-geometry mode (VolumeFeatureRenderer.forward(pts, rays_d, viewdirs, z_vals, ...), volume_renderer.py:192-303):
  // the caller's points / directions / depths are read instead of being generated from the camera
  constexpr bool explicit_geom = XG;
  const int64_t bray = (int64_t)b * R + rayc;
  float dx, dy, dz, ox = 0.f, oy = 0.f, oz = 0.f, vx, vy, vz;
  if (explicit_geom) {
    dx = P.x_rays_d[bray * 3]; dy = P.x_rays_d[bray * 3 + 1]; dz = P.x_rays_d[bray * 3 + 2];
    vx = P.x_viewdirs[bray * 3]; vy = P.x_viewdirs[bray * 3 + 1]; vz = P.x_viewdirs[bray * 3 + 2];
  } else {
    const float focal = P.focals[b];
    const float* cw = P.cam_poses + 12 * b;
    const int pi = rayc / S, pj = rayc - pi * S;
    const float px = (float)pj + 0.5f, py = (float)pi + 0.5f;
    const float dcx = (px - (float)S * 0.5f) / focal;
    const float dcy = -(py - (float)S * 0.5f) / focal;
    const float dcz = -1.f;
    dx = (dcx * cw[0] + dcy * cw[1]) + dcz * cw[2];
    dy = (dcx * cw[4] + dcy * cw[5]) + dcz * cw[6];
    dz = (dcx * cw[8] + dcy * cw[9]) + dcz * cw[10];
    ox = cw[3]; oy = cw[7]; oz = cw[11];
    vx = P.static_viewdirs ? dcx : dx; vy = P.static_viewdirs ? dcy : dy; vz = P.static_viewdirs ? dcz : dz;
    const float n = fmaxf(sqrtf((vx * vx + vy * vy) + vz * vz), 1e-12f);
    vx /= n; vy /= n; vz /= n;
  }
  const float dnorm = sqrtf((dx * dx + dy * dy) + dz * dz);
  const float u = (P.perturb_u && !explicit_geom) ? P.perturb_u[bray] : 0.f;
  // Wave-uniform floats that come out of the VALU live in VGPRs and, under this kernel's register pressure, get spilled
  // to scratch and reloaded one dependent round trip at a time at every sample start: pin them in SGPRs.
  auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float span = uniform(farv - nearv);
  const int N = P.n_samples;
  // torch.linspace(0, 1 - 1/N, N): symmetric evaluation around the midpoint
  const float t_end = a.t_end, t_step = a.t_step;
  auto zbase = [&](int k) -> float {  // un-perturbed depth of sample k; k == N gives `far`
    if (k >= N) return farv;
    const float t = (k < N / 2) ? t_step * (float)k : t_end - t_step * (float)(N - 1 - k);
    return nearv * (1.f - t) + farv * t;
  };
  auto zsample = [&](int k) -> float {
    if (explicit_geom) {         // (opaque row index: the 64-bit row pointer would be a loop-invariant scratch spill)
      int64_t br = bray;
      asm volatile("" : "+v"(br));
      return P.x_z_vals[br * N + (k < N ? k : N - 1)];
    }
    const float z0 = zbase(k);
    return P.perturb_u ? z0 + (zbase(k + 1) - z0) * u : z0;
  };

  // ---- per-lane compositing state
  float FA[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) FA[i] = 0.f;
  float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, ax = 0.f, ay = 0.f, az = 0.f, wlast = 0.f;

  Ring ring;
  ring.packed = F32 ? P.packed32 : P.packed;
  ring.lds = ringmem;
  ring.seq = 0;
  ring.per_sample = D * (NT / TPS);
  ring.seq_end = a.chunk * ring.per_sample;
  stage_slab<SLAB>(ring.packed, ringmem, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const int s_begin = c * a.chunk;
  STAMP(0);   // prologue
  for (int si = 0; si < a.chunk; ++si) {
    const int sg = s_begin + si;
    const bool live = ray_ok && sg < N;
    const int sk = sg < N ? sg : N - 1;
    // the ray's spilled constants that the set-up needs come back in ONE batch of reloads (touching them together here)
    // instead of one dependent round trip each where they are first used
    if (!explicit_geom) asm volatile("" ::"v"(u), "v"(dx), "v"(dy), "v"(dz));
    const float z = zsample(sk);
    float ptx, pty, ptz;
    if (explicit_geom) {
      const float* pp = P.x_pts + (bray * N + sk) * 3;
      ptx = pp[0]; pty = pp[1]; ptz = pp[2];
    } else {
      ptx = ox + dx * z; pty = oy + dy * z; ptz = oz + dz * z;
    }
    const float nx = ptx * 2.f / span, ny = pty * 2.f / span, nz = ptz * 2.f / span;
    // Opaque zero folded into every table offset of this iteration: the tables are loop-invariant and
    // LICM would otherwise hoist ~5*H/4 registers of them out of the sample loop (and spill them).
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int q4o = 4 * qd + opq;

    h8 Xh[NT / 2], Xl[NT / 2], Yh[NT / 2], Yl[NT / 2];
    float* stash_s = nullptr;   // this task's stash rows of the sample (differentiable forward)
    if constexpr (STASH) stash_s = P.stash + (((task0 + wave) * a.chunk + si) * D) * (int64_t)(16 * H);
    float sdf = 0.f;            // sigma head partial of this lane's units (taken where h_D is produced in fp32)
    // ---- layer 0: 3 -> H, in D layout, split into the hi / lo B fragments of the first MFMA layer.
    // The hidden layers alternate between the two activation register sets (X -> Y, Y -> X) instead of copying every layer's
    // output back into its input's registers (128 registers' worth of v_mov per layer and sample: 5 % of the sample loop's VALU
    // instructions at D = 2, 12 % at D = 8).  The view layer reads X: with an odd number of hidden layers layer 0 writes Y and one
    // Y -> X layer runs in front of the pairs (a third copy of the layer's code; entering the pair loop at its second half would
    // be an irreducible loop, and that form spilled 250 registers).
    // (the exact-fp32 instantiation spills 170 registers in this form and keeps the copies)
    constexpr bool PP = !F32;
    const bool odd_hidden = PP && ((D - 1) & 1);
    float chead[3] = {0.f, 0.f, 0.f};
    auto layer0 = [&](auto& Oh, auto& Ol) {
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
      float v8[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int o4 = (2 * m + hf) * 16 + q4o;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_film + o4);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_film + H + o4);
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_w0 + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_w0 + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_w0 + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pre = fmaf(wz[i], nz, fmaf(wy[i], ny, wx[i] * nx));
          v8[hf * 4 + i] = FILM_SIN(fmaf(g4[i], pre, c4[i]));
        }
        if (D == 1) {           // no hidden MFMA layer: this is h_D
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) sdf = fmaf(ws4[i], v8[hf * 4 + i], sdf);
        }
      }
      put8<F32>(v8, Oh[m], Ol[m]);
    }
    };
    // ---- hidden layers 1 .. D-1
    if constexpr (PP) {
      int l = 1;
      if (odd_hidden) {          // layer 0 -> Y, the odd hidden layer Y -> X; then pairs
        layer0(Yh, Yl);
        STAMP(1);   // sample setup + layer 0
        mfma_layer<NT, TPS, false, STASH, F32>(Yh, Yl, Xh, Xl, FA, 0.f, chead, sdf, D == 2, ring, s_film + 2 * H, s_wd, s_wc,
                                          s_ws, vx, vy, vz, STASH ? stash_s : nullptr, wave, lane,
                                          q4o STAMP_ARG);
        l = 2;
      } else {
        layer0(Xh, Xl);
        STAMP(1);   // sample setup + layer 0
      }
      for (; l < D; l += 2) {
        mfma_layer<NT, TPS, false, STASH, F32>(Xh, Xl, Yh, Yl, FA, 0.f, chead, sdf, false, ring, s_film + l * 2 * H, s_wd, s_wc,
                                          s_ws, vx, vy, vz, STASH ? stash_s + (int64_t)(l - 1) * 16 * H : nullptr, wave, lane,
                                          q4o STAMP_ARG);
        mfma_layer<NT, TPS, false, STASH, F32>(Yh, Yl, Xh, Xl, FA, 0.f, chead, sdf, l + 1 == D - 1, ring, s_film + (l + 1) * 2 * H, s_wd, s_wc,
                                          s_ws, vx, vy, vz, STASH ? stash_s + (int64_t)l * 16 * H : nullptr, wave, lane,
                                          q4o STAMP_ARG);
      }
    } else {
    layer0(Xh, Xl);
    STAMP(1);   // sample setup + layer 0
    for (int l = 1; l < D; ++l) {
      mfma_layer<NT, TPS, false, STASH, F32>(Xh, Xl, Yh, Yl, FA, 0.f, chead, sdf, l == D - 1, ring, s_film + l * 2 * H, s_wd, s_wc,
                                        s_ws, vx, vy, vz, STASH ? stash_s + (int64_t)(l - 1) * 16 * H : nullptr, wave, lane,
                                        q4o STAMP_ARG);
#pragma unroll
      for (int i = 0; i < NT / 2; ++i) { Xh[i] = Yh[i]; Xl[i] = Yl[i]; }
    }
    }
    STAMP(2);   // hidden layers
    // ---- sigma head on h_D (volume_renderer.py:148): the per-lane partial was accumulated where h_D was produced
    sdf += __shfl_xor(sdf, 16, 64);
    sdf += __shfl_xor(sdf, 32, 64);
    sdf += b_sigma;

    // ---- compositing weight of this sample (nerf_utils.py:264-307); known before the view layer
    const float delta = (sk < N - 1 ? zsample(sk + 1) - z : 1e10f) * dnorm;
    float sigma;
    if (raw_density) {        // with_sdf = False (nerf_utils.py:288-297): softplus of the raw density.  A real (uniform) branch and
      // the hardware exp / log: the library forms held enough temporaries here to spill four registers of the default path
      sigma = sdf > 20.f ? sdf : __logf(1.f + __expf(sdf));
    } else {
      sigma = sigmoidf_acc(-sdf / sig_beta) / sig_beta;
    }
    const float alpha = 1.f - expf(-sigma * delta);
    const float w = live ? alpha * T : 0.f;
    if (live) T *= (1.f - alpha) + 1e-10f;

    STAMP(3);   // sigma head + weight
    // ---- view layer -> features, folded into FA; rgb head partial sums
    float sdf_unused = 0.f;
    mfma_layer<NT, TPS, true, STASH, F32>(Xh, Xl, Yh, Yl, FA, w, chead, sdf_unused, false, ring, s_film + D * 2 * H, s_wd, s_wc, s_ws,
                                     vx, vy, vz, STASH ? stash_s + (int64_t)(D - 1) * 16 * H : nullptr, wave, lane,
                                     q4o STAMP_ARG);
    float c0 = chead[0], c1 = chead[1], c2 = chead[2];
    c0 += __shfl_xor(c0, 16, 64); c1 += __shfl_xor(c1, 16, 64); c2 += __shfl_xor(c2, 16, 64);
    c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
    c0 += b_rgb0; c1 += b_rgb1; c2 += b_rgb2;

    STAMP(4);   // view layer
    cr = fmaf(w, sigmoidf_acc(c0), cr); cg = fmaf(w, sigmoidf_acc(c1), cg); cb = fmaf(w, sigmoidf_acc(c2), cb);
    ax = fmaf(w, ptx, ax); ay = fmaf(w, pty, ay); az = fmaf(w, ptz, az);
    if (sg == N - 1) wlast = w;
    if (P.sdf && live && qd == 0) {
      P.sdf[((int64_t)b * R + ray_again()) * N + sg] = sdf;
    }
    if constexpr (STASH) {
      if (live && qd == 0) {     // per-point inputs of the compositing backward, p = sample * R + ray
        const int64_t Pn = (int64_t)R * N, p = (int64_t)sg * R + ray_again();
        P.bwd_sdf[(int64_t)b * Pn + p] = sdf;
        P.bwd_crgb[((int64_t)b * 3 + 0) * Pn + p] = c0;
        P.bwd_crgb[((int64_t)b * 3 + 1) * Pn + p] = c1;
        P.bwd_crgb[((int64_t)b * 3 + 2) * Pn + p] = c2;
      }
    }
  }

  STAMP(5);   // last compositing tail
#ifdef CIPS3D_CLOCK
  if (tid == 0) {
    const unsigned long long dt_ = __builtin_amdgcn_s_memtime() - clk_t0, dr_ = __builtin_amdgcn_s_memrealtime() - clk_r0;
    atomicAdd(&g_nerf_clock[0], dt_);
    atomicAdd(&g_nerf_clock[1], dr_);
    if (blockIdx.x < 4096) { g_nerf_clock_wg[2 * blockIdx.x] = dt_; g_nerf_clock_wg[2 * blockIdx.x + 1] = dr_; }
  }
#endif
  if (a.fuse_finish) {
    // ---- the eight waves of this workgroup are the eight chunks of ONE ray group (n_chunks == WAVES): exchange the
    // partials through LDS (the ring and the tables are dead) and combine them in sample order,
    //   S = sum_c (prod_{c' < c} T_c') S_c,
    // with the arithmetic of nerf_finish_kernel.  xf[c][ray][ch] (16-byte lane writes), xs[c][k][ray] for the 8 scalars.
    float* xf = ringmem;
    float* xs = s_film;
    __builtin_amdgcn_s_waitcnt(0x0F70);      // no LDS-DMA of this wave still in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                         // every wave is done with the ring and the tables
    {
      float* d = xf + (wave * RAYS + pl) * nerf_xf_pitch(H) + 4 * qd;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        *reinterpret_cast<f32x4*>(d + t * 16) = f32x4{FA[t * 4], FA[t * 4 + 1], FA[t * 4 + 2], FA[t * 4 + 3]};
      float* e = xs + wave * 8 * RAYS + pl;
      if (qd == 0) { e[0 * RAYS] = cr; e[1 * RAYS] = cg; }
      else if (qd == 1) { e[2 * RAYS] = cb; e[3 * RAYS] = ax; }
      else if (qd == 2) { e[4 * RAYS] = ay; e[5 * RAYS] = az; }
      else { e[6 * RAYS] = wlast; e[7 * RAYS] = T; }
    }
    __syncthreads();
    int rr = tid & 15;                       // ray of the group
    // (opaque: the output addresses derived from it were otherwise computed before the sample loop and kept in spilled
    // 64-bit registers across it -- 2.1 MB of scratch traffic per launch for nothing)
    asm volatile("" : "+v"(rr));
    // With fewer chunks per ray (n_chunks = 4, 2, 1: batches of 2, 4, 8 and more) the workgroup holds 8 / n_chunks ray
    // groups, wave w = chunk w % n_chunks of group w / n_chunks: the same combination per group, over n_chunks partials.
    // The batch-1 case (8 chunks, one group) keeps its fully unrolled form: with the chunk count a run-time value the
    // headline launch lost 1.1 us (same-box A/B).
    auto combine = [&](auto nc_) {
    constexpr int NCT = decltype(nc_)::value;
    const int NC = NCT > 0 ? NCT : P.n_chunks;
    const int g0 = ((int)(task0 % a.tasks_per_view)) / NC;      // ray group of wave 0 (task0 is a multiple of 8)
    for (int gi = 0; gi < WAVES / NC; ++gi) {
    const int wb = gi * NC;                                     // first wave of this group
    const int gray = (g0 + gi) * RAYS + rr;
    if (gray < R) {
      float Tp[WAVES];
      Tp[0] = 1.f;
#pragma unroll
      for (int cc = 1; cc < WAVES; ++cc) Tp[cc] = cc < NC ? Tp[cc - 1] * xs[(wb + cc - 1) * 8 * RAYS + 7 * RAYS + rr] : 0.f;
      // features: thread (ray rr, channel quad cq) for cq = tid >> 4 + 32 j
      for (int cq = tid >> 4; cq < H / 4; cq += WAVES * 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < WAVES; ++cc) {
          if (cc < NC) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xf + ((wb + cc) * RAYS + rr) * nerf_xf_pitch(H) + 4 * cq);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(Tp[cc], v[e], acc[e]);
          }
        }
        if (P.features_planes) {
          // split-fp16 planes [b][H/8][hi|lo][R][8]: channels 4 cq .. 4 cq + 3 = elements 4 (cq & 1) .. of channel block cq >> 1
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          unsigned h0, l0, h1, l1;                   // 2^-CIPS3D_FEATURES_EXP: |feature| <= 1 (cips3d_range)
          cips3d_split_pair(acc[0] * 16384.f, acc[1] * 16384.f, h0, l0);
          cips3d_split_pair(acc[2] * 16384.f, acc[3] * 16384.f, h1, l1);
          const h4 hi = __builtin_bit_cast(h4, u32x2_t{h0, h1}), lo = __builtin_bit_cast(h4, u32x2_t{l0, l1});
          _Float16* o = reinterpret_cast<_Float16*>(P.o_features) +
                        ((((int64_t)b * (H / 8) + (cq >> 1)) * 2) * R + gray) * 8 + 4 * (cq & 1);
          cips3d_store_wt8(o, hi);                     // (write-through: the next launch reads the planes on every XCD)
          cips3d_store_wt8(o + (int64_t)R * 8, lo);
        } else {
          float* o = P.o_features + ((int64_t)b * H + 4 * cq) * R + gray;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[(int64_t)e * R] = acc[e];
        }
      }
      const int k = tid >> 4;                // scalar channel 0..6 for the first 7 x 16 threads
      if (k < 7) {
        float acc = 0.f;
#pragma unroll
        for (int cc = 0; cc < WAVES; ++cc)
          if (cc < NC) acc = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + k * RAYS + rr], acc);
        if (k < 3) {
          P.o_thumb[((int64_t)b * 3 + k) * R + gray] = -1.f + 2.f * acc;
        } else if (k < 6) {
          P.o_xyz[((int64_t)b * 3 + (k - 3)) * R + gray] = acc;
        } else {
          const int64_t m0 = P.mask_planar ? (int64_t)b * R + gray : (int64_t)b * 2 * R + gray;     // [2,B,R] or [B,2,R]
          const int64_t m1 = m0 + (P.mask_planar ? (int64_t)P.B * R : (int64_t)R);
          P.o_mask[m0] = acc;
          float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
          for (int cc = 0; cc < WAVES; ++cc) {
            if (cc < NC) {
              sx = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 3 * RAYS + rr], sx);
              sy = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 4 * RAYS + rr], sy);
              sz = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 5 * RAYS + rr], sz);
            }
          }
          P.o_mask[m1] = -sqrtf((sx * sx + sy * sy) + sz * sz);
        }
      }
    }
    }   // ray groups of the workgroup
    };
    if (P.n_chunks == WAVES) combine(std::integral_constant<int, WAVES>{});
    else combine(std::integral_constant<int, 0>{});
    // cips3d_nerf_params.zero_words: scratch of the caller that its NEXT launches expect zeroed (the decoder's measured range rows
    // in a frame of a sequence: a launch of its own otherwise).  Every workgroup clears a slice on its way out -- at the end,
    // where the registers are free and no load of the kernel sits behind the stores.
    if (P.zero_words) {
      for (int64_t i = (int64_t)blockIdx.x * (WAVES * 64) + threadIdx.x; i < P.n_zero_words; i += (int64_t)gridDim.x * (WAVES * 64))
        P.zero_words[i] = 0.f;
    }
    STAMP(6);   // fused finish
    STAMP_FLUSH();
    return;
  }
  // ---- write the chunk partial: part[c][b][ch][ray]
  if (ray_ok) {
    int tid_o = threadIdx.x;               // (opaque: the lane's ray and quarter are re-derived, not kept across the loop)
    asm volatile("" : "+v"(tid_o));
    const int ray_o = g * RAYS + (tid_o & 15), qe = (tid_o >> 4) & 3;
    float* dst = P.part + ((int64_t)(c * P.B + b) * (H + 8)) * R + ray_o;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(int64_t)(t * 16 + 4 * qe + r) * R] = FA[t * 4 + r];
    if (qe == 0) {
      dst[(int64_t)(H + 0) * R] = cr; dst[(int64_t)(H + 1) * R] = cg;
    } else if (qe == 1) {
      dst[(int64_t)(H + 2) * R] = cb; dst[(int64_t)(H + 3) * R] = ax;
    } else if (qe == 2) {
      dst[(int64_t)(H + 4) * R] = ay; dst[(int64_t)(H + 5) * R] = az;
    } else {
      dst[(int64_t)(H + 6) * R] = wlast; dst[(int64_t)(H + 7) * R] = T;
    }
  }
  STAMP_FLUSH();
}

// features[b][ch][ray] = sum_c (prod_{c'<c} T_c') * part[c][b][ch][ray]; same for rgb/xyz/w_last.
__global__ void __launch_bounds__(256) nerf_finish_kernel(const float* __restrict__ part, int C, int B, int R,
                                                          int H, float* __restrict__ features,
                                                          float* __restrict__ thumb, float* __restrict__ xyz,
                                                          float* __restrict__ mask) {
  const int CH = H + 8;
  const int64_t total = (int64_t)B * (H + 7) * R;   // channel H+7 (T) is consumed, not emitted
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ray = (int)(i % R);
  const int ch = (int)((i / R) % (H + 7));
  const int b = (int)(i / ((int64_t)R * (H + 7)));
  float Tp = 1.f, acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* pc = part + ((int64_t)(c * B + b) * CH) * R + ray;
    acc = fmaf(Tp, pc[(int64_t)ch * R], acc);
    Tp *= pc[(int64_t)(H + 7) * R];
  }
  if (ch < H) {
    features[((int64_t)b * H + ch) * R + ray] = acc;
  } else if (ch < H + 3) {
    thumb[((int64_t)b * 3 + (ch - H)) * R + ray] = -1.f + 2.f * acc;
  } else if (ch < H + 6) {
    xyz[((int64_t)b * 3 + (ch - H - 3)) * R + ray] = acc;
  } else {
    mask[((int64_t)b * 2 + 0) * R + ray] = acc;
    // depth = -|xyz| (second mask channel) needs all three xyz sums: this thread recombines them the same way
    float Tq = 1.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int c = 0; c < C; ++c) {
      const float* pc = part + ((int64_t)(c * B + b) * CH) * R + ray;
      ax = fmaf(Tq, pc[(int64_t)(H + 3) * R], ax);
      ay = fmaf(Tq, pc[(int64_t)(H + 4) * R], ay);
      az = fmaf(Tq, pc[(int64_t)(H + 5) * R], az);
      Tq *= pc[(int64_t)(H + 7) * R];
    }
    mask[((int64_t)b * 2 + 1) * R + ray] = -sqrtf((ax * ax + ay * ay) + az * az);
  }
}

// the same, four consecutive rays per thread (16-byte loads / stores); R % 4 == 0
__global__ void __launch_bounds__(256) nerf_finish4_kernel(const float* __restrict__ part, int C, int B, int R, int H,
                                                           float* __restrict__ features, float* __restrict__ thumb,
                                                           float* __restrict__ xyz, float* __restrict__ mask) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int CH = H + 8, R4 = R / 4;
  const int64_t total = (int64_t)B * (H + 7) * R4;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ray = (int)(i % R4) * 4;
  const int ch = (int)((i / R4) % (H + 7));
  const int b = (int)(i / ((int64_t)R4 * (H + 7)));
  v4 Tp = {1.f, 1.f, 1.f, 1.f}, acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    const float* pc = part + ((int64_t)(c * B + b) * CH) * R + ray;
    const v4 v = *reinterpret_cast<const v4*>(pc + (int64_t)ch * R);
    const v4 t = *reinterpret_cast<const v4*>(pc + (int64_t)(H + 7) * R);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] = fmaf(Tp[e], v[e], acc[e]); Tp[e] *= t[e]; }
  }
  if (ch < H) {
    *reinterpret_cast<v4*>(features + ((int64_t)b * H + ch) * R + ray) = acc;
  } else if (ch < H + 3) {
    v4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = -1.f + 2.f * acc[e];
    *reinterpret_cast<v4*>(thumb + ((int64_t)b * 3 + (ch - H)) * R + ray) = o;
  } else if (ch < H + 6) {
    *reinterpret_cast<v4*>(xyz + ((int64_t)b * 3 + (ch - H - 3)) * R + ray) = acc;
  } else {
    *reinterpret_cast<v4*>(mask + ((int64_t)b * 2 + 0) * R + ray) = acc;
    v4 Tq = {1.f, 1.f, 1.f, 1.f}, ax = {0.f, 0.f, 0.f, 0.f}, ay = ax, az = ax;
    for (int c = 0; c < C; ++c) {
      const float* pc = part + ((int64_t)(c * B + b) * CH) * R + ray;
      const v4 x4 = *reinterpret_cast<const v4*>(pc + (int64_t)(H + 3) * R), y4 = *reinterpret_cast<const v4*>(pc + (int64_t)(H + 4) * R),
               z4 = *reinterpret_cast<const v4*>(pc + (int64_t)(H + 5) * R), t = *reinterpret_cast<const v4*>(pc + (int64_t)(H + 7) * R);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ax[e] = fmaf(Tq[e], x4[e], ax[e]); ay[e] = fmaf(Tq[e], y4[e], ay[e]); az[e] = fmaf(Tq[e], z4[e], az[e]);
        Tq[e] *= t[e];
      }
    }
    v4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = -sqrtf((ax[e] * ax[e] + ay[e] * ay[e]) + az[e] * az[e]);
    *reinterpret_cast<v4*>(mask + ((int64_t)b * 2 + 1) * R + ray) = d;
  }
}

template <int NT, int TPS, bool XG, bool STASH, bool F32 = false>
int launch_render_x(const NerfArgs& a, hipStream_t st) {
  const cips3d_nerf_params& P = a.p;
  constexpr int H = NT * 16;
  const size_t lds_bytes = sizeof(float) * (nerf_ring_floats(H, TPS, a.fuse_finish != 0) + (size_t)nerf_table_floats(H, P.depth + 1));
  if (lds_bytes > 160 * 1024) return CIPS3D_E_UNSUPP;
  // the attribute is per device (a process may render on several GPUs) and the flag is shared by host threads
  static std::atomic<unsigned long long> attr_set{0};
  int dev_id = 0;
  if (hipError_t e = hipGetDevice(&dev_id); e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev_id & 63);
  if (dev_id >= 64 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_render_kernel<NT, TPS, XG, STASH, F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const int64_t wgs = (int64_t)P.B * a.tasks_per_view / WAVES;
  hipLaunchKernelGGL((nerf_render_kernel<NT, TPS, XG, STASH, F32>), dim3((unsigned)wgs), dim3(WAVES * 64), lds_bytes, st, a);
  return cips3d_launch_status();
}

template <int NT, int TPS>
int launch_render(const NerfArgs& a, hipStream_t st) {
  if (a.p.packed32) {       // exact fp32 (hidden 256 only: cips3d_nerf_pack_weights32)
    if constexpr (NT == 16) return a.p.x_pts ? launch_render_x<NT, TPS, true, false, true>(a, st) : launch_render_x<NT, TPS, false, false, true>(a, st);
    else return CIPS3D_E_UNSUPP;
  }
  if (a.p.x_pts) return launch_render_x<NT, TPS, true, false>(a, st);
  return a.p.stash ? launch_render_x<NT, TPS, false, true>(a, st) : launch_render_x<NT, TPS, false, false>(a, st);
}

}  // namespace

extern "C" int cips3d_nerf_pack_weights32(const float* w_hidden, const float* w_view, float* packed32, int hidden, int depth,
                                          void* stream) {
  if (!w_view || !packed32 || hidden <= 0 || depth < 1 || (depth > 1 && !w_hidden)) return CIPS3D_E_BADARG;
  if (hidden != 256 || depth > 64) return CIPS3D_E_UNSUPP;
  const int64_t total = (int64_t)hidden * hidden * depth + 2 * depth;
  hipLaunchKernelGGL(nerf_pack32_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0, as_stream(stream), w_hidden,
                     w_view, packed32, hidden, depth);
  return cips3d_launch_status();
}

extern "C" int cips3d_nerf_pack_weights(const float* w_hidden, const float* w_view, float* packed, int hidden,
                                        int depth, void* stream) {
  if (!w_view || !packed || hidden <= 0 || depth < 1 || (depth > 1 && !w_hidden)) return CIPS3D_E_BADARG;
  if (hidden != 32 && hidden != 64 && hidden != 128 && hidden != 256) return CIPS3D_E_UNSUPP;
  if (depth > 64) return CIPS3D_E_UNSUPP;
  const int64_t total = (int64_t)hidden * hidden * depth;
  hipLaunchKernelGGL(nerf_scale_kernel, dim3((unsigned)depth), dim3(256), 0, as_stream(stream), w_hidden, w_view, packed,
                     hidden, depth);
  hipLaunchKernelGGL(nerf_pack_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0,
                     as_stream(stream), w_hidden, w_view, packed, hidden, depth);
  return cips3d_launch_status();
}

extern "C" int64_t cips3d_nerf_packed_floats(int hidden, int depth) {
  return hidden > 0 && depth > 0 ? nerf_packed_floats(hidden, depth) : 0;
}

extern "C" int cips3d_nerf_suggest_chunks(int B, int img_size, int n_samples) {
  if (B <= 0 || img_size <= 0 || n_samples <= 0) return 1;
  const int64_t groups = ceil_div<int64_t>((int64_t)img_size * img_size, 16);
  // want >= 2048 wave tasks (256 CUs x 4 SIMDs x 2 waves); more chunks = more partial traffic
  int64_t want = ceil_div<int64_t>(2048, (int64_t)B * groups);
  if (want < 1) want = 1;
  if (want > n_samples) want = n_samples;
  // round up to a divisor-friendly count: smallest C >= want whose chunk size ceil(N/C) wastes nothing
  int best = (int)want;
  for (int C = (int)want; C <= n_samples; ++C) {
    if (n_samples % C == 0) { best = C; break; }
  }
  return best;
}

extern "C" int64_t cips3d_nerf_part_floats(int B, int img_size, int hidden, int n_chunks) {
  return (int64_t)n_chunks * B * (hidden + 8) * img_size * img_size;
}

// fused finish: needs the chunk waves of a ray group in one workgroup (n_chunks divides 8), the tables large enough for the scalar exchange
// and the partial exchange + tables within the 160 KB of LDS.  CIPS3D_NERF_FUSE_FINISH=0: the stand-alone nerf_finish launch (A/B knob).
static int nerf_fuse_plan(const cips3d_nerf_params* p) {
  const int H = p->hidden, L = p->depth + 1;
  if (H != 32 && H != 64 && H != 128 && H != 256) return 0;
  const int TPS = H == 256 ? 4 : 2;
  if (!(p->o_features && p->o_thumb && p->o_xyz && p->o_mask)) return 0;
  static const int off = getenv("CIPS3D_NERF_FUSE_FINISH") ? atoi(getenv("CIPS3D_NERF_FUSE_FINISH")) == 0 : 0;   // A/B knob
  if (off || p->n_chunks < 1 || WAVES % p->n_chunks != 0 || nerf_table_floats(H, L) < WAVES * 8 * RAYS) return 0;
  return sizeof(float) * ((size_t)nerf_ring_floats(H, TPS, true) + nerf_table_floats(H, L)) <= 160 * 1024 ? 1 : 0;
}

extern "C" int cips3d_nerf_fuses_finish(const cips3d_nerf_params* p) { return p ? nerf_fuse_plan(p) : 0; }

namespace {
__global__ void __launch_bounds__(256) nerf_zero_kernel(float* __restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
}  // namespace

extern "C" int cips3d_nerf_render(const cips3d_nerf_params* p, void* stream) {
  if (!p) return CIPS3D_E_BADARG;
  const cips3d_nerf_params& P = *p;
  if ((P.o_features || P.o_thumb || P.o_xyz || P.o_mask) && !(P.o_features && P.o_thumb && P.o_xyz && P.o_mask))
    return CIPS3D_E_BADARG;
  const int fuse = cips3d_nerf_fuses_finish(p);
  if (P.packed32 && (P.hidden != 256 || P.stash)) return CIPS3D_E_UNSUPP;     // exact fp32: hidden 256, no differentiable forward
  if (!P.near_ || !P.far_ || !P.w_first || (!P.packed && !P.packed32) || !P.w_view || !P.film ||
      !P.layer_bias || !P.w_sigma || !P.w_rgb || !P.b_sigma || !P.b_rgb || (!P.sigmoid_beta && !P.raw_density) || (!P.part && !fuse))
    return CIPS3D_E_BADARG;
  if (P.x_pts ? (!P.x_rays_d || !P.x_viewdirs || !P.x_z_vals || P.n_rays <= 0) : (!P.cam_poses || !P.focals || P.n_rays != 0))
    return CIPS3D_E_BADARG;
  if (P.B < 0 || P.img_size <= 0 || P.n_samples <= 0 || P.depth < 1 || P.n_chunks < 1 ||
      P.n_chunks > P.n_samples)
    return CIPS3D_E_BADARG;
  if ((P.stash || P.bwd_sdf || P.bwd_crgb) && !(P.stash && P.bwd_sdf && P.bwd_crgb && !P.x_pts)) return CIPS3D_E_BADARG;
  if ((P.zero_words == nullptr) != (P.n_zero_words == 0) || P.n_zero_words < 0) return CIPS3D_E_BADARG;
  if (P.mask_planar && !fuse) return CIPS3D_E_UNSUPP;       // (the stand-alone finish writes [B,2,R])
  if (P.B == 0 && P.zero_words) {
    hipLaunchKernelGGL(nerf_zero_kernel, dim3((unsigned)ceil_div<int64_t>(P.n_zero_words, 256)), dim3(256), 0, as_stream(stream), P.zero_words, P.n_zero_words);
    return cips3d_launch_status();
  }
  if (P.B == 0) return 0;
  const bool zero_in_front = P.zero_words && !fuse;
  if (zero_in_front) {        // only the fused finish clears the words itself
    hipLaunchKernelGGL(nerf_zero_kernel, dim3((unsigned)ceil_div<int64_t>(P.n_zero_words, 256)), dim3(256), 0, as_stream(stream), P.zero_words, P.n_zero_words);
    if (const int rc = cips3d_launch_status()) return rc;
  }
  NerfArgs a;
  a.p = P;
  if (zero_in_front) { a.p.zero_words = nullptr; a.p.n_zero_words = 0; }
  a.groups = ceil_div(P.n_rays > 0 ? P.n_rays : P.img_size * P.img_size, RAYS);
  a.tasks_per_view = ceil_div(a.groups * P.n_chunks, WAVES) * WAVES;
  a.chunk = ceil_div(P.n_samples, P.n_chunks);
  a.fuse_finish = fuse;
  a.t_end = (float)(1.0 - 1.0 / (double)P.n_samples);
  a.t_step = P.n_samples > 1 ? a.t_end / (float)(P.n_samples - 1) : 0.f;
  hipStream_t st = as_stream(stream);
  switch (P.hidden) {
    case 32: return launch_render<2, 2>(a, st);
    case 64: return launch_render<4, 2>(a, st);
    case 128: return launch_render<8, 2>(a, st);
    case 256: {
      // four o-tiles per slab step (64 KB slabs, 4 accumulator chains, half as many step barriers / epilogue blocks):
      // 245.5 -> 237.0 us at D=2 N=24, -4 % at N=64 and at D=8, measured A/B on one box.  Deep networks whose FiLM
      // tables no longer fit beside two 64 KB slabs fall back to two tiles per step.
      const int rc = launch_render<16, 4>(a, st);
      return rc == CIPS3D_E_UNSUPP ? launch_render<16, 2>(a, st) : rc;
    }
    default: return CIPS3D_E_UNSUPP;
  }
}

extern "C" int cips3d_nerf_finish(const float* part, int n_chunks, int B, int img_size, int hidden,
                                  float* features, float* thumb_rgb, float* xyz, float* mask, void* stream) {
  if (img_size <= 0) return CIPS3D_E_BADARG;
  return cips3d_nerf_finish_rays(part, n_chunks, B, img_size * img_size, hidden, features, thumb_rgb, xyz, mask, stream);
}

extern "C" int cips3d_nerf_finish_rays(const float* part, int n_chunks, int B, int n_rays, int hidden, float* features,
                                       float* thumb_rgb, float* xyz, float* mask, void* stream) {
  if (!part || !features || !thumb_rgb || !xyz || !mask || n_chunks < 1 || B < 0 || n_rays <= 0 || hidden <= 0)
    return CIPS3D_E_BADARG;
  if (B == 0) return 0;
  const int R = n_rays;
  const int64_t total = (int64_t)B * (hidden + 7) * R;
  hipStream_t st = as_stream(stream);
  const bool vec = (R % 4 == 0) && (((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(features) |
                                      reinterpret_cast<uintptr_t>(thumb_rgb) | reinterpret_cast<uintptr_t>(xyz) |
                                      reinterpret_cast<uintptr_t>(mask)) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(nerf_finish4_kernel, dim3((unsigned)ceil_div<int64_t>(total / 4, 256)), dim3(256), 0, st, part, n_chunks,
                       B, R, hidden, features, thumb_rgb, xyz, mask);
  else
    hipLaunchKernelGGL(nerf_finish_kernel, dim3((unsigned)ceil_div<int64_t>(total, 256)), dim3(256), 0, st, part,
                       n_chunks, B, R, hidden, features, thumb_rgb, xyz, mask);
  return cips3d_launch_status();
}
