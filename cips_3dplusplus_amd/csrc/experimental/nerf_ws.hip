// Weight-stationary form of the fused NeRF render kernel (hidden width 256, camera-driven inference, final maps written by the
// kernel).  Same arithmetic as csrc/nerf.hip -- FiLM-SIREN point MLP on split-fp16 MFMA products, compositing, feature map --
// in another dataflow:
//
//   nerf.hip      a wave owns 16 points and ALL 256 output units of a layer; the layer's weights stream past every wave through
//                 the LDS ring (64 ds_read_b128 of A fragments + 8 LDS-DMA pieces per wave and slab step of 96 MFMAs); the
//                 activations never leave the wave's registers.
//   this file     a workgroup of 8 waves owns a ray group of 16 rays and walks its samples in batches of 64 points (16 rays x
//                 4 consecutive samples).  Wave w owns output units 32 w .. 32 w + 31 of every layer and holds their weights for
//                 all K = 256 inputs IN REGISTERS for the layer's duration (128 VGPRs of A fragments, loaded straight from the
//                 packed stream in L2 while the previous layer's epilogue runs).  The activations of the batch live in LDS as B
//                 fragments -- X[k-block][column tile][hi|lo][lane][8], 64 KB, double-buffered -- written by the wave that
//                 produced them (its 32 units ARE k-block w: the MFMA D layout is the lane's own slot of the B fragment) and read
//                 by everyone: 2 ds_read_b128 per 6 MFMAs where nerf.hip needs 4, no LDS-DMA, no slab barriers.
//
// Because a workgroup sees all samples of its rays in order, transmittance is carried in registers across batches: there are no
// sample chunks and no chunk combination.  What a wave cannot know alone -- the sigma head and the rgb head sum over all 256
// units -- goes through 8 KB of LDS per batch (per-wave partial sums, added in wave order by every wave: deterministic).
//
// MEASURED, and why this is NOT the default (opt-in: CIPS3D_NERF_WS=1; parity-tested against nerf.hip; tools/nerf_pair_ab.py --ws).
// D = 2, N = 24, batch 1, same box, HIP events: 105 us against nerf.hip's 87-92 (N = 64: 254 against 222; batch 4: 379 against 326).
// Timing ablations of the first form (-DCIPS3D_WS_ABL): 122 us; without the MFMAs 68; without the weight reloads 112; without
// the sines 113; without layer 0's sines 119; without the barriers 118.  The matrix blocks alone take ~54 us here as in every
// other form of this kernel (4608 MFMAs x 16 cycles per SIMD = 35 us at 2.1 GHz: the chip holds ~1.4 GHz under them,
// MI355X_MICROARCH.md "DVFS give-back"), and everything else -- layer 0, the epilogues, compositing, the cross-wave sums --
// ADDS to them: the eight waves run the same phase between two barriers, so the two waves of a SIMD are in their matrix
// blocks together and in their VALU work together.  What brought 122 down to 105: the exponentials of compositing evaluated
// once per lane quarter (one sample each) and gathered by shuffles instead of four per lane in every wave; the rgb head's
// sigmoids in wave 0 only.  What did not help: a layer's four (matrix block, epilogue) pairs column tile by column tile (this
// form; 107 -> 105: symmetric waves do not drift apart); the upper wave of every SIMD one matrix block ahead on a second
// accumulator set (284 spilled registers at the kernel's 256).  nerf.hip's waves are coupled only at slab barriers and its upper
// half takes them before its epilogue: that overlap is worth more than the LDS reads and DMA issue this form saves.
//
// Per batch and wave (D = 2): layer 0 on the VALU for its 32 units x 64 points -> X0; barrier; hidden layer: 192 MFMAs, FiLM + sine
// + split of 32 values per lane -> X1, sigma-head partials; barrier; compositing weights; view layer: 192 MFMAs, FiLM + sine,
// features += w f, rgb-head partials.  Two barriers per batch; the rgb partials of a batch are consumed behind the next batch's
// first barrier.
#include <stdlib.h>
#include <atomic>
#include "nerf_mlp.h"

int cips3d_nerf_ws_applies(const cips3d_nerf_params* p);
int cips3d_nerf_render_ws(const cips3d_nerf_params* p, void* stream);

#ifndef CIPS3D_WS_ABL
#define CIPS3D_WS_ABL 0      // timing-only ablations (results garbage): 1 weights loaded once, 2 no FiLM / sine epilogues, 4 no layer 0, 8 no MFMAs, 16 no barriers
#endif

namespace {

constexpr int WS_H = 256, WS_NT = 16, WS_MB = 8;      // hidden width, o-tiles, 32-unit k-blocks
constexpr int WS_CT = 4;                              // column tiles (= consecutive samples of the 16 rays) per batch
constexpr int WS_XBUF = WS_MB * WS_CT * 2 * 256;      // floats of one activation image: [k-block][column tile][hi|lo][lane x 4]
constexpr int WS_RED = WAVES * WS_CT * 16;            // floats of one per-wave partial-sum array: [column tile][ray][wave]

struct WsArgs {
  cips3d_nerf_params p;
  int groups;
  float t_end, t_step;
};

__host__ __device__ constexpr int ws_lds_floats(int L) { return 2 * WS_XBUF + L * 2 * WS_H + 10 * WS_H + 7 * WS_RED; }

// the wave's A fragments of one MFMA layer: o-tiles 2 w, 2 w + 1, all k-blocks, hi and lo planes -- 32 loads of 16 bytes per lane
__device__ __forceinline__ void ws_load_a(h8 (&Ah)[2][WS_MB], h8 (&Al)[2][WS_MB], const float* __restrict__ packed, int lp, int wave,
                                          int lane) {
  // (a uniform base per piece advanced on the scalar unit + one 32-bit lane offset, both opaque: with per-lane 64-bit pointers the 32
  // addresses of each possible next layer were computed ahead of the loop and kept -- in scratch)
  const float* base = packed + ((int64_t)lp * WS_NT + 2 * wave) * (16 * WS_H);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kb = 0; kb < WS_MB; ++kb)
#pragma unroll
      for (int pln = 0; pln < 2; ++pln) {
        const char* ub = reinterpret_cast<const char*>(base + t * (16 * WS_H) + (2 * kb + pln) * 256);
        unsigned vo = lane * 16;
        asm volatile("" : "+s"(ub), "+v"(vo));
        const h8 v = *reinterpret_cast<const h8*>(ub + vo);
        if (pln == 0) Ah[t][kb] = v; else Al[t][kb] = v;
      }
}

// acc[t] += W(o-tile 2 w + t) X(column tile c): k-blocks ascending, per k-block lo*hi, hi*lo, hi*hi -- the order of nerf.hip's
// accumulators.  One column tile at a time (48 MFMAs), its epilogue right behind it: the two waves of a SIMD drift into
// complementary phases, one wave's FiLM / sine work under the other's matrix work.  (All four column tiles first, then all four
// epilogues -- both waves in the same phase between two barriers -- measured 107 us against this form's time: see DESIGN.md.)
// B fragments are read one k-block ahead; the wait is provoked in front of the next reads.
__device__ __forceinline__ void ws_matrix(f32x4 (&acc)[2], const h8 (&Ah)[2][WS_MB], const h8 (&Al)[2][WS_MB], const float* X, int c,
                                          int lane) {
  h8 bh[2], bl[2];
  auto load_b = [&](int buf, int kb) {
    bh[buf] = *reinterpret_cast<const h8*>(X + ((kb * WS_CT + c) * 2 + 0) * 256 + lane * 4);
    bl[buf] = *reinterpret_cast<const h8*>(X + ((kb * WS_CT + c) * 2 + 1) * 256 + lane * 4);
  };
  load_b(0, 0);
#pragma unroll
  for (int kb = 0; kb < ((CIPS3D_WS_ABL & 8) ? 1 : WS_MB); ++kb) {
    const int cur = kb & 1;
    asm volatile("" : "+v"(bh[cur]), "+v"(bl[cur]));
    if (kb + 1 < WS_MB) load_b(cur ^ 1, kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[t][kb], bh[cur], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t][kb], bl[cur], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[t][kb], bh[cur], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ void __launch_bounds__(WAVES * 64, 2) nerf_render_ws_kernel(WsArgs a) {
  constexpr int H = WS_H;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const cips3d_nerf_params& P = a.p;
  const int D = P.depth, L = D + 1;
  float* s_x = lds;                          // two activation images
  float* s_film = s_x + 2 * WS_XBUF;         // L * 2 * H
  float* s_w0 = s_film + L * 2 * H;          // [3][H]
  float* s_wd = s_w0 + 3 * H;                // [3][H]
  float* s_ws = s_wd + 3 * H;                // [H]
  float* s_wc = s_ws + H;                    // [3][H]
  float* s_rsdf = s_wc + 3 * H;              // [ct][ray][wave]
  float* s_rrgb = s_rsdf + WS_RED;           // [batch parity][channel][ct][ray][wave]

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int qd = lane >> 4, pl = lane & 15;
  const int b = blockIdx.x / a.groups, g = blockIdx.x - b * a.groups;
  const int S = P.img_size, R = S * S, N = P.n_samples;
  const int ray = g * RAYS + pl;
  const bool ray_ok = ray < R;
  const int rayc = ray_ok ? ray : R - 1;

  // ---- per-view tables (nerf.hip: same staging)
  {
    const float* film_b = P.film + (int64_t)b * L * 2 * H;
    const float* scales = P.packed + (int64_t)D * H * H;
    for (int i = tid; i < L * H; i += WAVES * 64) {
      const int l = i / H, o = i - l * H;
      const float gm = film_b[(l * 2) * H + o];
      s_film[(l * 2) * H + o] = (l >= 1 ? gm * scales[2 * (l - 1) + 1] : gm) * FILM_UNIT;
      s_film[(l * 2 + 1) * H + o] = fmaf(gm, P.layer_bias[i], film_b[(l * 2 + 1) * H + o]) * FILM_UNIT;
    }
    const float view_scale = scales[2 * (D - 1)];
    for (int i = tid; i < 3 * H; i += WAVES * 64) {
      const int k = i / H, o = i - k * H;
      s_w0[i] = P.w_first[o * 3 + k];
      s_wd[i] = P.w_view[o * (H + 3) + H + k] * view_scale;
      s_wc[i] = P.w_rgb[i];
    }
    for (int i = tid; i < H; i += WAVES * 64) s_ws[i] = P.w_sigma[i];
  }
  // the first MFMA layer's weights travel while the tables are staged and layer 0 runs
  h8 Ah[2][WS_MB], Al[2][WS_MB];
  ws_load_a(Ah, Al, P.packed, 0, wave, lane);       // packed layer 0 = hidden layer 1, or the view layer when D == 1

  const float b_sigma = P.b_sigma[0], b_rgb0 = P.b_rgb[0], b_rgb1 = P.b_rgb[1], b_rgb2 = P.b_rgb[2];
  const bool raw_density = __builtin_amdgcn_readfirstlane(P.raw_density) != 0;
  const float sig_beta = raw_density ? 1.f : P.sigmoid_beta[0];

  // ---- ray setup (nerf_utils.py:38-66; nerf.hip's camera-driven branch)
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
  const float t_end = a.t_end, t_step = a.t_step;
  auto zbase = [&](int k) -> float {
    if (k >= N) return farv;
    const float t = (k < N / 2) ? t_step * (float)k : t_end - t_step * (float)(N - 1 - k);
    return nearv * (1.f - t) + farv * t;
  };
  auto zsample = [&](int k) -> float {
    const float z0 = zbase(k);
    return P.perturb_u ? z0 + (zbase(k + 1) - z0) * u : z0;
  };

  // ---- per-lane state: features of this wave's units for ray pl, compositing sums (every wave carries the same ones)
  float FA[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) FA[i] = 0.f;
  float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, ax = 0.f, ay = 0.f, az = 0.f, wlast = 0.f;
  float wprev[WS_CT] = {0.f, 0.f, 0.f, 0.f};         // the previous batch's weights, until its rgb sums are consumed
  const int ubase = 32 * wave + 4 * qd;                // this lane's first unit of o-tile 2 w (tile 2 w + 1: + 16)
  const int xw = (wave * WS_CT) * 2 * 256 + lane * 4;  // this lane's slot of k-block w, column tile 0, hi plane
  const int n_batches = (N + WS_CT - 1) / WS_CT;
  const int lp_view = D - 1;
  __syncthreads();                                     // tables staged

  // rgb head of a finished batch: partial sums of the 8 waves, in wave order.  Only wave 0 writes the scalar maps, so only it
  // evaluates the sigmoids; lane quarter q takes sample q of the batch (its own running sums; the quarters meet at the end).
  auto consume_rgb = [&](int parity) {
    if (wave != 0) return;
    const float* rr = s_rrgb + parity * 3 * WS_RED;
    float hd[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(rr + ch * WS_RED + (qd * 16 + pl) * WAVES);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(rr + ch * WS_RED + (qd * 16 + pl) * WAVES + 4);
      hd[ch] = ((((((p0[0] + p0[1]) + p0[2]) + p0[3]) + p1[0]) + p1[1]) + p1[2]) + p1[3];
    }
    const float w = qd == 0 ? wprev[0] : qd == 1 ? wprev[1] : qd == 2 ? wprev[2] : wprev[3];
    cr = fmaf(w, sigmoidf_acc(hd[0] + b_rgb0), cr);
    cg = fmaf(w, sigmoidf_acc(hd[1] + b_rgb1), cg);
    cb = fmaf(w, sigmoidf_acc(hd[2] + b_rgb2), cb);
  };

#pragma unroll 1
  for (int bi = 0; bi < n_batches; ++bi) {
    const int s0 = bi * WS_CT;
    int opq = 0;                                       // (keeps the table reads of this batch inside the loop: see nerf.hip)
    asm volatile("" : "+v"(opq));
    const int u0 = ubase + opq;
    float* x_out = s_x + xw;
    // ---- layer 0: 3 -> H on the VALU, this wave's 32 units for the batch's 64 points -> image 0
    float sdf_p[WS_CT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < WS_CT; ++c) {
      const int sk = s0 + c < N ? s0 + c : N - 1;
      const float z = zsample(sk);
      const float nx = (ox + dx * z) * 2.f / span, ny = (oy + dy * z) * 2.f / span, nz = (oz + dz * z) * 2.f / span;
      float v8[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o4 = u0 + 16 * t;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_w0 + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_w0 + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_w0 + 2 * H + o4);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_film + o4);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_film + H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pre = fmaf(wz[i], nz, fmaf(wy[i], ny, wx[i] * nx));
          v8[t * 4 + i] = (CIPS3D_WS_ABL & 4) ? pre : FILM_SIN(fmaf(g4[i], pre, c4[i]));
        }
        if (D == 1) {
          const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) sdf_p[c] = fmaf(ws4[i], v8[t * 4 + i], sdf_p[c]);
        }
      }
      h8 hi, lo;
      split8(v8, hi, lo);
      *reinterpret_cast<h8*>(x_out + (c * 2 + 0) * 256) = hi;
      *reinterpret_cast<h8*>(x_out + (c * 2 + 1) * 256) = lo;
    }
    int img = 0;                                       // image that holds the input of the next MFMA layer
    // ---- hidden layers 1 .. D - 1: image (l - 1) & 1 -> image l & 1
    for (int l = 1; l < D; ++l) {
      if (!(CIPS3D_WS_ABL & 16)) __syncthreads();      // the input image is complete (and the previous reads of the output image are over)
      if (l == 1 && bi > 0) consume_rgb((bi - 1) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this layer's A fragments have landed
      const float* film_l = s_film + l * 2 * H;
      const bool last = l == D - 1;
      const float* xi = s_x + img * WS_XBUF;
      float* xo = s_x + (img ^ 1) * WS_XBUF + xw;
#pragma unroll
      for (int c = 0; c < WS_CT; ++c) {
        f32x4 acc[2];
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        ws_matrix(acc, Ah, Al, xi, c, lane);
        // (the next MFMA layer's weights are requested behind the LAST column tile's matrix work, under the remaining epilogues)
        if (c == WS_CT - 1 && !(CIPS3D_WS_ABL & 1)) ws_load_a(Ah, Al, P.packed, l < D - 1 ? l : lp_view, wave, lane);
        int oc = 0;                                    // (per column tile: the table reads are the same for all four and would be hoisted -- and A spilled)
        asm volatile("" : "+v"(oc));
        float v8[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int o4 = u0 + 16 * t + oc;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v8[t * 4 + i] = (CIPS3D_WS_ABL & 2) ? acc[t][i] : FILM_SIN(fmaf(g4[i], acc[t][i], c4[i]));
          if (last) {
            const f32x4 ws4 = *reinterpret_cast<const f32x4*>(s_ws + o4);
#pragma unroll
            for (int i = 0; i < 4; ++i) sdf_p[c] = fmaf(ws4[i], v8[t * 4 + i], sdf_p[c]);
          }
        }
        h8 hi, lo;
        split8(v8, hi, lo);
        // (pinned here: left to itself the epilogue's arithmetic floats down to the LDS stores, behind the next column tile's MFMAs)
        asm volatile("" : "+v"(hi), "+v"(lo));
        *reinterpret_cast<h8*>(xo + (c * 2 + 0) * 256) = hi;
        *reinterpret_cast<h8*>(xo + (c * 2 + 1) * 256) = lo;
        asm volatile("" : "+v"(sdf_p[c]));
      }
      img ^= 1;
    }
    // ---- sigma head: this wave's partial of every point, then everyone adds the eight partials in wave order
#pragma unroll
    for (int c = 0; c < WS_CT; ++c) {
      float v = sdf_p[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (qd == 0) s_rsdf[(c * 16 + pl) * WAVES + wave] = v;
    }
    if (!(CIPS3D_WS_ABL & 16)) __syncthreads();        // h_D image and sigma partials complete
    if (D == 1 && bi > 0) consume_rgb((bi - 1) & 1);
    float w[WS_CT];
    {
      // lane quarter q evaluates sample s0 + q of ray pl (the exponentials are what costs here, not the scan over the samples)
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(s_rsdf + (qd * 16 + pl) * WAVES);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(s_rsdf + (qd * 16 + pl) * WAVES + 4);
      const float sdf = (((((((p0[0] + p0[1]) + p0[2]) + p0[3]) + p1[0]) + p1[1]) + p1[2]) + p1[3]) + b_sigma;
      const int sgq = s0 + qd;
      const int skq = sgq < N ? sgq : N - 1;
      const float zq = zsample(skq);
      const float delta = (skq < N - 1 ? zsample(skq + 1) - zq : 1e10f) * dnorm;      // nerf_utils.py:264-307
      float sigma;
      if (raw_density) sigma = sdf > 20.f ? sdf : __logf(1.f + __expf(sdf));
      else sigma = sigmoidf_acc(-sdf / sig_beta) / sig_beta;
      const float alpha_q = 1.f - expf(-sigma * delta);
      if (P.sdf && ray_ok && sgq < N && wave == 0) P.sdf[((int64_t)b * R + ray) * N + sgq] = sdf;
#pragma unroll
      for (int c = 0; c < WS_CT; ++c) {
        const float alpha = __shfl(alpha_q, pl + 16 * c, 64);
        const int sg = s0 + c;
        const bool live = ray_ok && sg < N;
        w[c] = live ? alpha * T : 0.f;
        if (live) T *= (1.f - alpha) + 1e-10f;
        if (sg == N - 1) wlast = w[c];
      }
      if (wave == 0) {                                 // xyz sums: the sample of this lane's quarter, like the rgb sums
        const float wq = qd == 0 ? w[0] : qd == 1 ? w[1] : qd == 2 ? w[2] : w[3];
        ax = fmaf(wq, ox + dx * zq, ax); ay = fmaf(wq, oy + dy * zq, ay); az = fmaf(wq, oz + dz * zq, az);
      }
    }
    // ---- view layer: f = sin(gamma' (W' h_D + Wd' v) + c); features += w f; rgb head partials
    {
      f32x4 dv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o4 = u0 + 16 * t;
        const f32x4 wx = *reinterpret_cast<const f32x4*>(s_wd + o4);
        const f32x4 wy = *reinterpret_cast<const f32x4*>(s_wd + H + o4);
        const f32x4 wz = *reinterpret_cast<const f32x4*>(s_wd + 2 * H + o4);
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[t][i] = fmaf(wz[i], vz, fmaf(wy[i], vy, wx[i] * vx));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float* film_l = s_film + D * 2 * H;
      const float* xi = s_x + img * WS_XBUF;
      float ch0[WS_CT] = {0.f, 0.f, 0.f, 0.f}, ch1[WS_CT] = {0.f, 0.f, 0.f, 0.f}, ch2[WS_CT] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < WS_CT; ++c) {
        f32x4 acc[2];
        acc[0] = dv[0];
        acc[1] = dv[1];
        ws_matrix(acc, Ah, Al, xi, c, lane);
        if (c == WS_CT - 1 && bi + 1 < n_batches && !(CIPS3D_WS_ABL & 1))
          ws_load_a(Ah, Al, P.packed, D > 1 ? 0 : lp_view, wave, lane);      // the next batch's first MFMA layer
        int oc = 0;
        asm volatile("" : "+v"(oc));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int o4 = u0 + 16 * t + oc;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(film_l + o4);
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(film_l + H + o4);
          const f32x4 k0 = *reinterpret_cast<const f32x4*>(s_wc + o4);
          const f32x4 k1 = *reinterpret_cast<const f32x4*>(s_wc + H + o4);
          const f32x4 k2 = *reinterpret_cast<const f32x4*>(s_wc + 2 * H + o4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float f = (CIPS3D_WS_ABL & 2) ? acc[t][i] : FILM_SIN(fmaf(g4[i], acc[t][i], c4[i]));
            FA[t * 4 + i] = fmaf(w[c], f, FA[t * 4 + i]);
            ch0[c] = fmaf(k0[i], f, ch0[c]);
            ch1[c] = fmaf(k1[i], f, ch1[c]);
            ch2[c] = fmaf(k2[i], f, ch2[c]);
          }
        }
        // (pinned per column tile: otherwise the whole view epilogue sinks below the fourth tile's MFMAs, 32 accumulators live)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(FA[k]));
        asm volatile("" : "+v"(ch0[c]), "+v"(ch1[c]), "+v"(ch2[c]));
      }
      float* rr = s_rrgb + (bi & 1) * 3 * WS_RED;
#pragma unroll
      for (int c = 0; c < WS_CT; ++c) {
        float v0 = ch0[c], v1 = ch1[c], v2 = ch2[c];
        v0 += __shfl_xor(v0, 16, 64); v1 += __shfl_xor(v1, 16, 64); v2 += __shfl_xor(v2, 16, 64);
        v0 += __shfl_xor(v0, 32, 64); v1 += __shfl_xor(v1, 32, 64); v2 += __shfl_xor(v2, 32, 64);
        if (qd == 0) {
          rr[0 * WS_RED + (c * 16 + pl) * WAVES + wave] = v0;
          rr[1 * WS_RED + (c * 16 + pl) * WAVES + wave] = v1;
          rr[2 * WS_RED + (c * 16 + pl) * WAVES + wave] = v2;
        }
        wprev[c] = w[c];
      }
    }
    // odd depths: the view layer read image 0, which the next batch's layer 0 overwrites
    if (D & 1) __syncthreads();
  }
  __syncthreads();                                     // the last batch's rgb partials
  consume_rgb((n_batches - 1) & 1);

  // ---- outputs: this wave's 32 feature channels of ray pl; wave 0 the scalar maps (volume_renderer.py:192-303 via nerf.hip's finish)
  if (wave == 0) {                                     // the quarters' running sums of the scalar maps meet (all lanes take part)
    cr += __shfl_xor(cr, 16, 64); cg += __shfl_xor(cg, 16, 64); cb += __shfl_xor(cb, 16, 64);
    ax += __shfl_xor(ax, 16, 64); ay += __shfl_xor(ay, 16, 64); az += __shfl_xor(az, 16, 64);
    cr += __shfl_xor(cr, 32, 64); cg += __shfl_xor(cg, 32, 64); cb += __shfl_xor(cb, 32, 64);
    ax += __shfl_xor(ax, 32, 64); ay += __shfl_xor(ay, 32, 64); az += __shfl_xor(az, 32, 64);
  }
  if (ray_ok) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int u4 = ubase + 16 * t;                   // channels u4 .. u4 + 3
      if (P.features_planes) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        unsigned h0, l0, h1, l1;                       // 2^-CIPS3D_FEATURES_EXP: |feature| <= 1 (cips3d_range)
        cips3d_split_pair(FA[t * 4 + 0] * 16384.f, FA[t * 4 + 1] * 16384.f, h0, l0);
        cips3d_split_pair(FA[t * 4 + 2] * 16384.f, FA[t * 4 + 3] * 16384.f, h1, l1);
        const h4 hi = __builtin_bit_cast(h4, u32x2_t{h0, h1}), lo = __builtin_bit_cast(h4, u32x2_t{l0, l1});
        _Float16* o = reinterpret_cast<_Float16*>(P.o_features) + ((((int64_t)b * (H / 8) + (u4 >> 3)) * 2) * R + ray) * 8 + (u4 & 4);
        *reinterpret_cast<h4*>(o) = hi;
        *reinterpret_cast<h4*>(o + (int64_t)R * 8) = lo;
      } else {
        float* o = P.o_features + ((int64_t)b * H + u4) * R + ray;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[(int64_t)i * R] = FA[t * 4 + i];
      }
    }
    if (wave == 0 && qd == 0) {
      P.o_thumb[((int64_t)b * 3 + 0) * R + ray] = -1.f + 2.f * cr;
      P.o_thumb[((int64_t)b * 3 + 1) * R + ray] = -1.f + 2.f * cg;
      P.o_thumb[((int64_t)b * 3 + 2) * R + ray] = -1.f + 2.f * cb;
      P.o_xyz[((int64_t)b * 3 + 0) * R + ray] = ax;
      P.o_xyz[((int64_t)b * 3 + 1) * R + ray] = ay;
      P.o_xyz[((int64_t)b * 3 + 2) * R + ray] = az;
      P.o_mask[((int64_t)b * 2 + 0) * R + ray] = wlast;
      P.o_mask[((int64_t)b * 2 + 1) * R + ray] = -sqrtf((ax * ax + ay * ay) + az * az);
    }
  }
}

}  // namespace

// 1 when cips3d_nerf_render runs this form (library-internal): hidden 256, camera-driven, final maps from the kernel, no stash,
// split arithmetic, tables + two activation images within the 160 KB of LDS, and enough ray groups to fill the chip
int cips3d_nerf_ws_applies(const cips3d_nerf_params* p) {
  if (!p) return 0;
  const char* knob = getenv("CIPS3D_NERF_WS");       // A/B knob, read per call (tests switch it inside one process)
  if (!knob || atoi(knob) == 0) return 0;
  const cips3d_nerf_params& P = *p;
  if (P.hidden != WS_H || P.x_pts || P.n_rays != 0 || P.stash || P.bwd_sdf || P.bwd_crgb || P.packed32 || !P.packed) return 0;
  if (!(P.o_features && P.o_thumb && P.o_xyz && P.o_mask)) return 0;
  if (P.depth < 1 || sizeof(float) * (size_t)ws_lds_floats(P.depth + 1) > 160 * 1024) return 0;
  return 1;
}

int cips3d_nerf_render_ws(const cips3d_nerf_params* p, void* stream) {
  WsArgs a;
  a.p = *p;
  const cips3d_nerf_params& P = a.p;
  a.groups = ceil_div(P.img_size * P.img_size, RAYS);
  a.t_end = (float)(1.0 - 1.0 / (double)P.n_samples);
  a.t_step = P.n_samples > 1 ? a.t_end / (float)(P.n_samples - 1) : 0.f;
  const size_t lds_bytes = sizeof(float) * (size_t)ws_lds_floats(P.depth + 1);
  static std::atomic<unsigned long long> attr_set{0};
  int dev_id = 0;
  if (hipError_t e = hipGetDevice(&dev_id); e != hipSuccess) return (int)e;
  const unsigned long long bit = 1ull << (dev_id & 63);
  if (dev_id >= 64 || !(attr_set.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nerf_render_ws_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(nerf_render_ws_kernel, dim3((unsigned)((int64_t)P.B * a.groups)), dim3(WAVES * 64), lds_bytes, as_stream(stream), a);
  return cips3d_launch_status();
}
