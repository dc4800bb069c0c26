// NeRF-style volume renderer of CIPS-3D++ on gfx950, the 32-points-per-wave form of nerf.hip's render kernel
// (reference: cips3d/nerf_utils.py:18-170, 230-338; cips3d/volume_renderer.py:39-160).  Same arithmetic, same packed
// weights, same tables and the same ordered chunk combination as nerf_render_kernel; what changes is the work shape:
//
//   nerf.hip        a wave = 16 rays x ONE sample at a time, 8 waves per workgroup (2 per SIMD, 256 registers each);
//                   loop order o-tile outer / k inner: a slab step finishes 4 o-tiles, its epilogue follows at once
//   this file       a wave = 16 rays x TWO consecutive samples at a time (two 16-column B tiles), 4 waves per workgroup
//                   (1 per SIMD, 512 registers); loop order k outer / o-tile inner: the accumulators of the WHOLE layer
//                   (2 x 16 tiles = 128 registers) stay in the accumulation registers, the layer input dies k-block by
//                   k-block and the epilogue writes the next layer's input over it
//
// Why.  (1) With 16 points per wave every A fragment (one ds_read_b128 of the weight slab) fed three MFMAs: at full matrix
// rate the 8 waves of a CU would ask the LDS for ~170 B/clk, two thirds of what the array delivers, and every wave re-read the
// whole slab.  With two column tiles a fragment feeds six MFMAs: LDS reads per flop halve.  (2) The two samples of a pair
// belong to the SAME 16 rays, so ray set-up, view-direction terms, FiLM / head table reads and the feature accumulators
// (sum_k w_k f_k runs over both columns into one register set) are shared.  (3) Register files: one wave per SIMD owns 512
// registers per lane, but only the first 256 (arch VGPRs) can be VALU operands; the other 256 (AGPRs) are reachable by
// MFMA C / D, loads, stores and v_accvgpr_*; hipcc puts MFMA results there and wants A / B operands in arch VGPRs.  The
// k-outer order is the one whose big array (the accumulators) is MFMA-only: activations 128 + features 64 + fragments in the
// arch file, 128 accumulators in the other.  (An o-outer form with both layer input and output as B fragments -- 256
// registers the VALU touches once -- came out of the register allocator with 1,300 v_accvgpr copies and 269 scratch spills per
// sample pair, whichever way the placement was hinted.)
//
// Work decomposition: task = (view b, group of 16 rays, chunk c of the samples); nc = 1, 2 or 4 chunks per ray chosen by the
// launcher so that the grid fills the chip (1024 waves); the 4 waves of a workgroup are 4 / nc ray groups x nc chunks and
// combine their partials through LDS in sample order (the arithmetic of nerf_finish_kernel) before the maps are written:
// this kernel only exists in the fused-finish form.  Lane l = (ray l & 15, quarter l >> 4), as in nerf.hip.
//
// Weight ring: a slab = 2 k-blocks (64 input units) x all 16 o-tiles = 64 KB, gathered by LDS-DMA from the o-tile-major
// packed stream of cips3d_nerf_pack_weights (a 1 KiB piece = one (o-tile, k-block, hi|lo plane) fragment set, contiguous
// there), 2 slots, one barrier per slab; every wave issues one piece per MFMA group of the slab before.
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include "common.h"
#include "nerf_mlp.h"

namespace {

constexpr int PW = 4;           // waves per workgroup (one per SIMD)

// One MFMA layer's matrix phase for the wave's 2 x 16 points: acc[c][t] = sum_m A(t, m) X[c][m], split-fp16 products.
// The caller guarantees slab `seq` is resident in slot (seq & 1); every slab step fetches seq + 1 meanwhile and ends with
// wait + barrier.  X is dead afterwards (the epilogue overwrites it).
// ABL: timing-only ablations (tools/nerf_pair_ab.py --abl; bit 0 no LDS-DMA, 1 no FiLM / sine epilogues, 2 no layer 0,
// 3 no MFMAs); only compiled into a -DCIPS3D_PAIR_ABLATIONS build, results are garbage
// F32: the exact-fp32 instantiation (cips3d_nerf_params.packed32): v_mfma_f32_16x16x4_f32 on fp32 weights and fp32 activations --
// bit for bit an fmaf chain in k order -- instead of three fp16 products per fp32 product.  Same slabs, same ring: a 1 KiB piece
// is the 16 x 16 fp32 block (o-tile, 16 input units) where the split stream has an (o-tile, 32 units, hi | lo) plane.
template <int NT, int ABL = 0, bool F32 = false>
struct PairMatrix {
  static constexpr int H = NT * 16;
  static constexpr int TILE = 16 * H;          // floats of one o-tile's A fragments in the packed stream
  static constexpr int MB = NT / 2;            // 32-unit k-blocks
  static constexpr int KPS = 2;                // k-blocks per slab
  static constexpr int SLAB = KPS * NT * 512;  // floats: [kb][o-tile][hi|lo][lane][4]
  static constexpr int STEPS = MB / KPS;
  static constexpr int PIECES = SLAB / 256;    // 1 KiB pieces per slab
  static constexpr int PPW = PIECES / PW;      // per wave
  static constexpr int GPK = NT / 2;           // MFMA groups (o-tile pairs) per k-block
  static constexpr int GPS = KPS * GPK;        // per slab
  static_assert(MB % KPS == 0 && PIECES % PW == 0 && GPS % PPW == 0, "slab geometry");

  // piece p of slab (layer lp, step s): kb = p / (2 NT), o-tile t = (p / 2) % NT, plane = p % 2
  __device__ static __forceinline__ void stage_piece_k(const float* __restrict__ packed, float* slot, int lp, int s, int p, int lane) {
    const int kb = p / (2 * NT), t = (p >> 1) % NT, plane = p & 1;
    const float* src = packed + (int64_t)lp * H * H + t * TILE + ((s * KPS + kb) * 2 + plane) * 256;
    const char* ub = reinterpret_cast<const char*>(src);
    unsigned vo = lane * 16;
    asm volatile("" : "+s"(ub), "+v"(vo));         // (the saddr form: see nerf_mlp.h:stage_piece)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + vo),
                                     (__attribute__((address_space(3))) void*)(slot + p * 256), 16, 0, 0);
  }
  __device__ static __forceinline__ void stage_slab_k(const float* __restrict__ packed, float* slot, int lp, int s, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) stage_piece_k(packed, slot, lp, s, j * PW + wave, lane);
  }

  // exact fp32: X[c][T] = this lane's 4 units (4 q + r) of o-tile T of the layer input (= the D layout of the layer before);
  // k-step (T, r) multiplies the weights' column 16 T + 4 q + r -- the order cips3d_nerf_pack_weights32 stores them in
  __device__ static __forceinline__ void run32(f32x4 (&acc)[2][NT], const f32x4 (&X)[2][NT], Ring& ring, int wave, int lane) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int nxt = (ring.seq + 1) % ring.per_sample;
      const int nlp = nxt / STEPS, ns = nxt - nlp * STEPS;
      float* lnext = ring.lds + ((ring.seq + 1) & 1) * SLAB;
      const float* slab = ring.lds + (ring.seq & 1) * SLAB;
      f32x4 fa[2][2][2];                           // [buffer][o-tile of the pair][half of the k-block]
      auto load_group = [&](int buf, int gi) {
        const int kb = gi / GPK, j = gi % GPK;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
            fa[buf][t][hf] = *reinterpret_cast<const f32x4*>(slab + ((kb * NT + 2 * j + t) * 2 + hf) * 256 + lane * 4);
      };
      load_group(0, 0);
#pragma unroll
      for (int gi = 0; gi < GPS; ++gi) {
        const int kb = gi / GPK, j = gi % GPK, m = s * KPS + kb, cur = gi & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(fa[cur][t][0]), "+v"(fa[cur][t][1]));
        if (gi + 1 < GPS) load_group(cur ^ 1, gi + 1);
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj)
          if (jj * GPS / PPW == gi) stage_piece_k(ring.packed, lnext, nlp, ns, jj * PW + wave, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acc[c][2 * j + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][t][hf][r], X[c][2 * m + hf][r],
                                                                         (m == 0 && hf == 0 && r == 0) ? z : acc[c][2 * j + t], 0, 0, 0);
              }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      ++ring.seq;
    }
  }

  __device__ static __forceinline__ void run(f32x4 (&acc)[2][NT], const h8 (&Xh)[2][MB], const h8 (&Xl)[2][MB], Ring& ring,
                                             int wave, int lane) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // the slab after this one (wrapping to a valid slab nobody reads behind the kernel's last: no branch in the stream)
      const int nxt = (ring.seq + 1) % ring.per_sample;
      const int nlp = nxt / STEPS, ns = nxt - nlp * STEPS;
      float* lnext = ring.lds + ((ring.seq + 1) & 1) * SLAB;
      const float* slab = ring.lds + (ring.seq & 1) * SLAB;
      h8 fh[2][2], fl[2][2];
      auto load_group = [&](int buf, int gi) {     // o-tiles 2 j, 2 j + 1 of k-block kb of the slab
        const int kb = gi / GPK, j = gi % GPK;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fh[buf][t] = *reinterpret_cast<const h8*>(slab + ((kb * NT + 2 * j + t) * 2 + 0) * 256 + lane * 4);
          fl[buf][t] = *reinterpret_cast<const h8*>(slab + ((kb * NT + 2 * j + t) * 2 + 1) * 256 + lane * 4);
        }
      };
      load_group(0, 0);
#pragma unroll
      for (int gi = 0; gi < GPS; ++gi) {
        const int kb = gi / GPK, j = gi % GPK, m = s * KPS + kb, cur = gi & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(fh[cur][t]), "+v"(fl[cur][t]));   // the wait lands here, in front of
        if (gi + 1 < GPS) load_group(cur ^ 1, gi + 1);                                          // the next group's reads
#pragma unroll
        for (int jj = 0; jj < PPW; ++jj)
          if (jj * GPS / PPW == gi && !(ABL & 1)) stage_piece_k(ring.packed, lnext, nlp, ns, jj * PW + wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ABL & 8) {
#pragma unroll
          for (int t = 0; t < 2; ++t) asm volatile("" :: "v"(fh[cur][t]), "v"(fl[cur][t]));
          if (m == 0) acc[0][2 * j] = acc[1][2 * j] = acc[0][2 * j + 1] = acc[1][2 * j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          continue;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            acc[c][2 * j + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cur][t], Xh[c][m], m == 0 ? z : acc[c][2 * j + t], 0, 0, 0);
          }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[c][2 * j + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xl[c][m], acc[c][2 * j + t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[c][2 * j + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur][t], Xh[c][m], acc[c][2 * j + t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces of the next slab have landed
      __syncthreads();                         // ... everybody's, and everybody is done reading the slot they go to next
      ++ring.seq;
    }
  }
};

template <int NT, int ABL, bool F32>
__global__ void __launch_bounds__(PW * 64, 1) nerf_render_pair_kernel(NerfArgs a) {
  typedef PairMatrix<NT, ABL, F32> MX;
  constexpr int H = NT * 16;
  constexpr int SLAB = MX::SLAB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cips3d_nerf_params& P = a.p;
  const int D = P.depth;
  const int L = D + 1;
  const float* packed = F32 ? P.packed32 : P.packed;       // the stream this instantiation multiplies (scales behind it: 1 for fp32)
  float* ringmem = lds;                      // 2 * SLAB (>= the PW x 16 x (H + 4) partial exchange of the finish)
  float* s_film = ringmem + 2 * SLAB;        // L * 2 * H
  float* s_w0 = s_film + L * 2 * H;          // [3][H]  first-layer weights, transposed
  float* s_wd = s_w0 + 3 * H;                // [3][H]  view-direction columns of the view layer
  float* s_ws = s_wd + 3 * H;                // [H]     sigma head
  float* s_wc = s_ws + H;                    // [3][H]  rgb head

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int qd = lane >> 4;
  const int pl = lane & 15;

  // ---- task decode (b is uniform over the workgroup: tasks_per_view is a multiple of PW)
  const int nc = P.n_chunks;                 // (the launcher's own chunking, not the caller's: see cips3d_nerf_render_pair)
  const int64_t task0 = (int64_t)blockIdx.x * PW;
  const int b = (int)(task0 / a.tasks_per_view);
  const int tv = (int)(task0 % a.tasks_per_view) + wave;
  const bool task_ok = tv < a.groups * nc;
  const int g = task_ok ? tv / nc : 0;
  const int c = task_ok ? tv % nc : 0;
  const int S = P.img_size;
  const int R = S * S;
  const int ray = g * RAYS + pl;
  auto ray_again = [&]() -> int {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return g * RAYS + (t & 15);
  };
  const bool ray_ok = task_ok && ray < R;
  const int rayc = ray < R ? ray : R - 1;

  // ---- stage the small per-view tables (as nerf_render_kernel)
  {
    const float* film_b = P.film + (int64_t)b * L * 2 * H;
    const float* scales = packed + (int64_t)D * H * H;
    for (int i = tid; i < L * H; i += PW * 64) {
      const int l = i / H, o = i - l * H;
      const float gm = film_b[(l * 2) * H + o];
      s_film[(l * 2) * H + o] = (l >= 1 ? gm * scales[2 * (l - 1) + 1] : gm) * FILM_UNIT;
      s_film[(l * 2 + 1) * H + o] = fmaf(gm, P.layer_bias[i], film_b[(l * 2 + 1) * H + o]) * FILM_UNIT;
    }
    const float view_scale = scales[2 * (D - 1)];
    for (int i = tid; i < 3 * H; i += PW * 64) {
      const int k = i / H, o = i - k * H;
      s_w0[i] = P.w_first[o * 3 + k];
      s_wd[i] = P.w_view[o * (H + 3) + H + k] * view_scale;
      s_wc[i] = P.w_rgb[i];
    }
    for (int i = tid; i < H; i += PW * 64) s_ws[i] = P.w_sigma[i];
  }

  const float b_sigma = P.b_sigma[0], b_rgb0 = P.b_rgb[0], b_rgb1 = P.b_rgb[1], b_rgb2 = P.b_rgb[2];
  const bool raw_density = __builtin_amdgcn_readfirstlane(P.raw_density) != 0;
  const float sig_beta = raw_density ? 1.f : P.sigmoid_beta[0];

  // ---- ray setup (nerf_utils.py:38-66)
  const float nearv = P.near_[b], farv = P.far_[b];
  const int64_t bray = (int64_t)b * R + rayc;
  float dx, dy, dz, ox, oy, oz, vx, vy, vz;
  {
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
  const float u = P.perturb_u ? P.perturb_u[bray] : 0.f;
  auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float span = uniform(farv - nearv);
  const int N = P.n_samples;
  const float t_end = a.t_end, t_step = a.t_step;
  auto zbase = [&](int k) -> float {  // un-perturbed depth of sample k; k == N gives `far`
    if (k >= N) return farv;
    const float t = (k < N / 2) ? t_step * (float)k : t_end - t_step * (float)(N - 1 - k);
    return nearv * (1.f - t) + farv * t;
  };
  auto zsample = [&](int k) -> float {
    const float z0 = zbase(k);
    return P.perturb_u ? z0 + (zbase(k + 1) - z0) * u : z0;
  };

  // ---- per-lane compositing state
  float FA[NT * 4];
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) FA[i] = 0.f;
  float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, ax = 0.f, ay = 0.f, az = 0.f, wlast = 0.f;

  Ring ring;
  ring.packed = packed;
  ring.lds = ringmem;
  ring.seq = 0;
  ring.per_sample = D * MX::STEPS;            // slabs per sample pair
  ring.seq_end = (a.chunk / 2) * ring.per_sample;
  MX::stage_slab_k(packed, ringmem, 0, 0, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const int s_begin = c * a.chunk;
  for (int pi = 0; pi < a.chunk / 2; ++pi) {
    const int sg0 = s_begin + 2 * pi, sg1 = sg0 + 1;
    const bool live0 = ray_ok && sg0 < N, live1 = ray_ok && sg1 < N;
    const int sk0 = sg0 < N ? sg0 : N - 1, sk1 = sg1 < N ? sg1 : N - 1;
    const float z0 = zsample(sk0), z1 = zsample(sk1);
    const float pt[2][3] = {{ox + dx * z0, oy + dy * z0, oz + dz * z0}, {ox + dx * z1, oy + dy * z1, oz + dz * z1}};
    // Opaque zero folded into every table offset of this iteration: the tables are loop-invariant and LICM would otherwise
    // hoist registers full of them out of the sample loop
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const int q4o = 4 * qd + opq;

    h8 Xh[2][NT / 2], Xl[2][NT / 2];           // split: hi / lo B fragments per k-block
    f32x4 Xf[2][NT];                           // F32: the activations themselves, per o-tile
    auto put_act = [&](int cc, int m, const float (&v8)[8]) {       // units of o-tiles 2 m, 2 m + 1 -> the next layer's input
      if constexpr (F32) {
        Xf[cc][2 * m] = f32x4{v8[0], v8[1], v8[2], v8[3]};
        Xf[cc][2 * m + 1] = f32x4{v8[4], v8[5], v8[6], v8[7]};
      } else {
        split8(v8, Xh[cc][m], Xl[cc][m]);
      }
    };
    float sdf[2] = {0.f, 0.f};
    // ---- layer 0: 3 -> H on the VALU, in D layout, split into the hi / lo B fragments of the first MFMA layer
    {
      float nrm[2][3];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int k = 0; k < 3; ++k) nrm[cc][k] = pt[cc][k] * 2.f / span;
#pragma unroll
      for (int m = 0; m < NT / 2; ++m) {
        float v8[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int o4 = (2 * m + hf) * 16 + q4o;
          const f32x4 wx = *reinterpret_cast<const f32x4*>(s_w0 + o4);
          const f32x4 wy = *reinterpret_cast<const f32x4*>(s_w0 + H + o4);
          const f32x4 wz = *reinterpret_cast<const f32x4*>(s_w0 + 2 * H + o4);
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_film + o4);
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_film + H + o4);
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float pre = fmaf(wz[i], nrm[cc][2], fmaf(wy[i], nrm[cc][1], wx[i] * nrm[cc][0]));
              v8[cc][hf * 4 + i] = (ABL & 4) ? nrm[cc][i & 1] : FILM_SIN(fmaf(g4[i], pre, c4[i]));
            }
          if (D == 1) {           // no hidden MFMA layer: this is h_D
            const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int i = 0; i < 4; ++i) sdf[cc] = fmaf(ws4[i], v8[cc][hf * 4 + i], sdf[cc]);
          }
        }
        put_act(0, m, v8[0]);
        put_act(1, m, v8[1]);
      }
    }
    f32x4 acc[2][NT];
    float w0 = 0.f, w1 = 0.f;
    float chead[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    // ---- MFMA layers 1 .. D-1 (hidden) and D (view): ONE copy of the matrix phase in the code, the epilogue by layer kind
    for (int l = 1; l <= D; ++l) {
      if (l == D) {
        // ---- sigma head on h_D (volume_renderer.py:148) and the compositing weights of the pair (nerf_utils.py:264-307)
        float sg[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          float sv = sdf[cc];
          sv += __shfl_xor(sv, 16, 64);
          sv += __shfl_xor(sv, 32, 64);
          sdf[cc] = sv + b_sigma;
          if (raw_density) sg[cc] = sdf[cc] > 20.f ? sdf[cc] : __logf(1.f + __expf(sdf[cc]));
          else sg[cc] = sigmoidf_acc(-sdf[cc] / sig_beta) / sig_beta;
        }
        const float z2 = zsample(sk1 + 1 < N ? sk1 + 1 : N - 1);
        const float delta0 = (sk0 < N - 1 ? z1 - z0 : 1e10f) * dnorm;
        const float delta1 = (sk1 < N - 1 ? z2 - z1 : 1e10f) * dnorm;
        const float alpha0 = 1.f - expf(-sg[0] * delta0);
        w0 = live0 ? alpha0 * T : 0.f;
        if (live0) T *= (1.f - alpha0) + 1e-10f;
        const float alpha1 = 1.f - expf(-sg[1] * delta1);
        w1 = live1 ? alpha1 * T : 0.f;
        if (live1) T *= (1.f - alpha1) + 1e-10f;
      }
      if constexpr (F32) MX::run32(acc, Xf, ring, wave, lane);
      else MX::run(acc, Xh, Xl, ring, wave, lane);
      const float* film_l = s_film + l * 2 * H;
      if (l < D) {
        // hidden: Y = sin(gamma' (W' X) + c), written over X as the next layer's B fragments
        const bool last = l == D - 1;
#pragma unroll
        for (int m = 0; m < NT / 2; ++m) {
          float v8[2][8];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int o4 = (2 * m + hf) * 16 + q4o;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                v8[cc][hf * 4 + i] = (ABL & 2) ? acc[cc][2 * m + hf][i] : FILM_SIN(fmaf(g4[i], acc[cc][2 * m + hf][i], c4[i]));
            if (last) {     // h_D: sigma head partial (volume_renderer.py:148) from the fp32 values, before they are split
              const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
              for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int i = 0; i < 4; ++i) sdf[cc] = fmaf(ws4[i], v8[cc][hf * 4 + i], sdf[cc]);
            }
          }
          put_act(0, m, v8[0]);
          put_act(1, m, v8[1]);
        }
      } else {
        // view: f = sin(gamma' (W' h_D + Wd' v) + c), features folded into FA, rgb head partial sums.  The view-direction
        // term is the same for both samples of the pair (same ray): computed once per unit
        float vxo = vx, vyo = vy, vzo = vz;
        asm volatile("" : "+v"(vxo), "+v"(vyo), "+v"(vzo));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if constexpr (ABL & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) FA[t * 4 + i] += acc[0][t][i] + acc[1][t][i];
            continue;
          }
          const int o4 = t * 16 + q4o;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
          const f32x4 wx = *reinterpret_cast<const f32x4*>(s_wd + o4);
          const f32x4 wy = *reinterpret_cast<const f32x4*>(s_wd + H + o4);
          const f32x4 wz = *reinterpret_cast<const f32x4*>(s_wd + 2 * H + o4);
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(s_wc + o4);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(s_wc + H + o4);
          const f32x4 k2 = *reinterpret_cast<const f32x4*>(s_wc + 2 * H + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dv = fmaf(wz[i], vzo, fmaf(wy[i], vyo, wx[i] * vxo));
            const float f0 = FILM_SIN(fmaf(g4[i], acc[0][t][i] + dv, c4[i]));
            const float f1 = FILM_SIN(fmaf(g4[i], acc[1][t][i] + dv, c4[i]));
            // sample order: the earlier sample of the pair first, as the one-sample kernel adds them
            FA[t * 4 + i] = fmaf(w1, f1, fmaf(w0, f0, FA[t * 4 + i]));
            chead[0][0] = fmaf(k0[i], f0, chead[0][0]); chead[1][0] = fmaf(k0[i], f1, chead[1][0]);
            chead[0][1] = fmaf(k1[i], f0, chead[0][1]); chead[1][1] = fmaf(k1[i], f1, chead[1][1]);
            chead[0][2] = fmaf(k2[i], f0, chead[0][2]); chead[1][2] = fmaf(k2[i], f1, chead[1][2]);
          }
        }
      }
    }
    // ---- rgb head (volume_renderer.py:157-158) and the per-ray sums (nerf_utils.py:308-336), sample sg0 then sg1
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      float c0 = chead[cc][0], c1 = chead[cc][1], c2 = chead[cc][2];
      c0 += __shfl_xor(c0, 16, 64); c1 += __shfl_xor(c1, 16, 64); c2 += __shfl_xor(c2, 16, 64);
      c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
      c0 += b_rgb0; c1 += b_rgb1; c2 += b_rgb2;
      const float w = cc ? w1 : w0;
      cr = fmaf(w, sigmoidf_acc(c0), cr); cg = fmaf(w, sigmoidf_acc(c1), cg); cb = fmaf(w, sigmoidf_acc(c2), cb);
      ax = fmaf(w, pt[cc][0], ax); ay = fmaf(w, pt[cc][1], ay); az = fmaf(w, pt[cc][2], az);
      const int sg = cc ? sg1 : sg0;
      if (sg == N - 1) wlast = w;
      if (P.sdf && (cc ? live1 : live0) && qd == 0) P.sdf[((int64_t)b * R + ray_again()) * N + sg] = sdf[cc];
    }
  }

  // ---- the waves of this workgroup are the nc chunks of PW / nc ray groups: exchange the partials through LDS (the ring and
  // the tables are dead) and combine them in sample order,  S = sum_c (prod_{c' < c} T_c') S_c,  with the arithmetic of
  // nerf_finish_kernel.  xf[wave][ray][ch] (16-byte lane writes), xs[wave][k][ray] for the 8 scalars.
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
  asm volatile("" : "+v"(rr));             // (opaque: keeps the output addresses out of the sample loop's live ranges)
  const int g0 = ((int)(task0 % a.tasks_per_view)) / nc;      // ray group of wave 0 (task0 is a multiple of PW)
  for (int gi = 0; gi < PW / nc; ++gi) {
    const int wb = gi * nc;                                   // first wave of this group
    const int gray = (g0 + gi) * RAYS + rr;
    if (g0 + gi >= a.groups || gray >= R) continue;
    float Tp[PW];
    Tp[0] = 1.f;
#pragma unroll
    for (int cc = 1; cc < PW; ++cc) Tp[cc] = cc < nc ? Tp[cc - 1] * xs[(wb + cc - 1) * 8 * RAYS + 7 * RAYS + rr] : 0.f;
    // features: thread (ray rr, channel quad cq) for cq = tid >> 4 + 16 j
    for (int cq = tid >> 4; cq < H / 4; cq += PW * 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < PW; ++cc) {
        if (cc < nc) {
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
        *reinterpret_cast<h4*>(o) = hi;
        *reinterpret_cast<h4*>(o + (int64_t)R * 8) = lo;
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
      for (int cc = 0; cc < PW; ++cc)
        if (cc < nc) acc = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + k * RAYS + rr], acc);
      if (k < 3) {
        P.o_thumb[((int64_t)b * 3 + k) * R + gray] = -1.f + 2.f * acc;
      } else if (k < 6) {
        P.o_xyz[((int64_t)b * 3 + (k - 3)) * R + gray] = acc;
      } else {
        P.o_mask[((int64_t)b * 2 + 0) * R + gray] = acc;
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int cc = 0; cc < PW; ++cc) {
          if (cc < nc) {
            sx = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 3 * RAYS + rr], sx);
            sy = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 4 * RAYS + rr], sy);
            sz = fmaf(Tp[cc], xs[(wb + cc) * 8 * RAYS + 5 * RAYS + rr], sz);
          }
        }
        P.o_mask[((int64_t)b * 2 + 1) * R + gray] = -sqrtf((sx * sx + sy * sy) + sz * sz);
      }
    }
  }
}

template <int NT, int ABL = 0, bool F32 = false>
int launch_pair(const NerfArgs& a, hipStream_t st) {
  const cips3d_nerf_params& P = a.p;
  constexpr int H = NT * 16;
  const size_t lds_bytes = sizeof(float) * ((size_t)2 * PairMatrix<NT>::SLAB + (size_t)(P.depth + 1) * 2 * H + 10 * H);
  if (lds_bytes > 160 * 1024) return CIPS3D_E_UNSUPP;
  static std::atomic<unsigned long long> attr_set{0};
  int dev_id = 0;
  if (hipError_t e = hipGetDevice(&dev_id); e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev_id & 63);
  if (dev_id >= 64 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_render_pair_kernel<NT, ABL, F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  const int64_t wgs = (int64_t)P.B * a.tasks_per_view / PW;
  hipLaunchKernelGGL((nerf_render_pair_kernel<NT, ABL, F32>), dim3((unsigned)wgs), dim3(PW * 64), lds_bytes, st, a);
  return cips3d_launch_status();
}

}  // namespace

// 1 when cips3d_nerf_render runs the pair kernel for this call (library-internal; the decision is the kernel's own: the
// caller's n_chunks only matters through the `part` layout, which this form never touches)
int cips3d_nerf_pair_applies(const cips3d_nerf_params* p) {
  if (!p) return 0;
  {       // opt-in in both arithmetics while it is the slower form (tools/nerf_pair_ab.py; exact fp32: 252 us against nerf.hip's F32 instantiation)
    const char* knob = getenv("CIPS3D_NERF_PAIR");    // A/B knob, read per call (tests switch it inside one process)
    if (!knob || atoi(knob) == 0) return 0;
  }
  const cips3d_nerf_params& P = *p;
  if (P.hidden != 256 || P.x_pts || P.n_rays != 0 || P.stash || P.bwd_sdf || P.bwd_crgb) return 0;
  if (!(P.o_features && P.o_thumb && P.o_xyz && P.o_mask)) return 0;
  if (P.n_samples < 2) return 0;
  // tables beside two 64 KB slabs; the scalar exchange of the finish needs PW * 8 * 16 floats of them
  const size_t lds_bytes = sizeof(float) * ((size_t)2 * 16 * 256 * 4 + (size_t)(P.depth + 1) * 2 * 256 + 10 * 256);
  return lds_bytes <= 160 * 1024;
}

int cips3d_nerf_render_pair(const cips3d_nerf_params* p, void* stream) {
  NerfArgs a;
  a.p = *p;
  const cips3d_nerf_params& P = a.p;
  a.groups = ceil_div(P.img_size * P.img_size, RAYS);
  // chunks per ray: enough wave tasks for one wave per SIMD on every CU (1024), at most PW (the chunks of a ray meet in one
  // workgroup), each chunk an even number of samples
  const int64_t G = (int64_t)P.B * a.groups;
  int nc = 1;
  while (nc < PW && G * nc < 1024 && (P.n_samples + 2 * nc - 1) / (2 * nc) >= 2) nc *= 2;
  a.p.n_chunks = nc;
  a.chunk = 2 * ceil_div(P.n_samples, 2 * nc);
  a.tasks_per_view = ceil_div(a.groups * nc, PW) * PW;
  a.fuse_finish = 1;
  a.t_end = (float)(1.0 - 1.0 / (double)P.n_samples);
  a.t_step = P.n_samples > 1 ? a.t_end / (float)(P.n_samples - 1) : 0.f;
#ifdef CIPS3D_PAIR_ABLATIONS
  switch (getenv("CIPS3D_PAIR_ABL") ? atoi(getenv("CIPS3D_PAIR_ABL")) : 0) {
    case 1: return launch_pair<16, 1>(a, as_stream(stream));
    case 2: return launch_pair<16, 2>(a, as_stream(stream));
    case 4: return launch_pair<16, 4>(a, as_stream(stream));
    case 6: return launch_pair<16, 6>(a, as_stream(stream));
    case 7: return launch_pair<16, 7>(a, as_stream(stream));
    case 8: return launch_pair<16, 8>(a, as_stream(stream));
    case 9: return launch_pair<16, 9>(a, as_stream(stream));
    default: break;
  }
#endif
  if (P.packed32) return launch_pair<16, 0, true>(a, as_stream(stream));
  return launch_pair<16>(a, as_stream(stream));
}
