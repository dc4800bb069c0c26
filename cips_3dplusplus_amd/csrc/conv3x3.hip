// 3x3 ModulatedConv2d of the decoder as an implicit GEMM on v_mfma_f32_16x16x4_f32 with LDS halo tiles
// (reference models/model_v3.py:264-314; the k = 3 decoder recipes, decoder_cfg.kernel_size = 3).
//
//   plain      out[o][y][x] = sum_i sum_{ty,tx} w[o][i][ty][tx] x[i][y+ty-1][x+tx-1]          (conv2d, padding 1; :296-311)
//   up-sampling  conv_transpose2d(stride 2, padding 0) -> (2H+1)^2, then Blur(4x4 taps x 4, pad (1,1)) -> (2H)^2  (:280-291).
//              Both steps are convolutions of the zero-stuffed input, so they commute:  the 4x4 FIR is applied to the
//              zero-stuffed INPUT channels first (Z = upfirdn2d(x, 4 f, up = 2, pad = (3, 2)), (2H+2)^2) and the transposed
//              conv becomes a plain "valid" 3x3 correlation of Z with the flipped taps,
//                   out[o][y][x] = sum_i sum_{ty,tx} w[o][i][2-ty][2-tx] Z[i][y+ty][x+tx]
//              (identity checked against the imported reference in tests/test_oracle_golden.py).  Z only ever exists as
//              the LDS tile the MFMAs read: the polyphase FIR fills the tile, the conv_transpose output and the blurred
//              tensor of the reference are never materialised.
//
// One kernel for both forms: a workgroup (4 waves) owns a 4-row x 64-column output tile and 16*WM output channels; per
// 16-channel K stage it stages a (4+2) x (64+2) halo tile per channel in LDS (double buffered; plain: predicated 16-byte
// loads, zero outside the image; up: polyphase FIR of a 2x2 low-resolution neighbourhood per element) and runs the nine
// taps as nine shifted GEMM steps: lane (pixel quad jn, quarter q) reads ONE 6-float window (ds_read_b32 + ds_read_b128
// + ds_read_b32) per channel and tile row and uses it for the three horizontal taps x four interleaved column tiles =
// 12 MFMAs.  A fragments (tap-major packed weights, cips3d_modulate_weights with ksq = 9) come straight from L2, one tile
// row of taps ahead.  Epilogue as the 1x1 GEMM: optional NoiseInjection + bias + leaky ReLU, 16-byte stores.
//
// Roofline: MFMA-bound (2*9*Cin*Cout flop per output pixel against (Cin + Cout) * 4 B).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Conv3Args {
  const float* x; const float* wmp; float* out;
  int B, Cin, Cout, H, W;          // input size; the output is H x W (plain) or 2H x 2W (up)
  const float* fir;                // up only: 4x4 taps as the reference's Blur holds them (already x 4)
  int epilogue; const float* noise; int64_t noise_bstride; const float* noise_w; const float* bias;
  const float* x_amax;             // split kernel: the measured per-sample maximum of x ([B][CIPS3D_AMAX_FLOATS], cips3d_range) or NULL
};

template <int WM, bool UP>
__global__ void __launch_bounds__(256) modconv3x3_kernel(Conv3Args a) {
  constexpr int TH = 4, TW = 64, BK = 16;
  constexpr int P = TW + 8;        // LDS row pitch: [3] = left halo column, [4..67] = the 64 tile columns, [68] = right halo
  constexpr int ROWS = TH + 2;
  constexpr int CS = 448;          // channel stride >= ROWS * P = 432, a multiple of 64 floats: the four lane quarters of a
                                   // ds_read_b128 (four channels) then hit disjoint bank ranges
  constexpr int STAGE = BK * CS;
  constexpr int CHUNKS = BK * ROWS * (P / 4);          // 16-byte pieces of one stage
  constexpr int CPT = (CHUNKS + 255) / 256;            // pieces per thread
  constexpr int RAW = UP ? 8 : 4;                      // staged floats per piece (up: a 2 x 4 low-resolution neighbourhood)
  __shared__ __attribute__((aligned(16))) float sT[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // = tile row
  const int lane = tid & 63;
  const int q = lane >> 4, jn = lane & 15;
  const int b = blockIdx.z;
  const int H = a.H, W = a.W;
  const int OH = UP ? 2 * H : H, OW = UP ? 2 * W : W;
  const int tiles_x = (OW + TW - 1) / TW;
  // XCD-aware tile order (see fused_up_conv_kernel): XCD k walks the k-th contiguous eighth of the pixel tiles, so that the
  // halo columns / rows of neighbouring tiles are found in its own L2 (workgroups go to the XCDs round-robin by linear id)
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int ox0 = (bid % tiles_x) * TW, oy0 = (bid / tiles_x) * TH;
  const int m0 = blockIdx.y * (16 * WM);
  const int K = a.Cin, nstage = K / BK;
  const int HWi = H * W;
  const float* xb = a.x + (int64_t)b * K * HWi;
  const float* ab = a.wmp + (int64_t)b * 9 * a.Cout * K;        // [tap][ot][kq][256]

  float kf[16];                    // up: flipped taps kf[a][b] = fir[3-a][3-b] (upfirdn2d correlates with the flipped kernel)
  if constexpr (UP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) kf[i] = a.fir[15 - i];
  }

  // ---- halo-tile staging, split in a load half (registers) and a store half (LDS) so that a stage's loads travel under
  // the previous stage's MFMAs; two halves per stage (the second half's loads are issued after the first third of the
  // MFMAs), which halves the staging registers -- the up-sampling form holds 8 floats per piece.
  // Per piece, computed once: its LDS offset, its source offset inside a channel-stage and a validity mask (the loads
  // of a stage then cost one address add each; out-of-image elements read element 0 and are replaced by zero).
  constexpr int HALF = (CPT + 1) / 2;
  int p_lds[CPT], p_src[CPT], p_ok[CPT], p_a0[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int g = tid + 256 * u;
    const int ch = g / (ROWS * (P / 4)), rem = g % (ROWS * (P / 4));
    const int t = rem / (P / 4), m = rem % (P / 4);
    const bool live = g < CHUNKS;
    p_lds[u] = live ? ch * CS + t * P + 4 * m : -1;
    if constexpr (!UP) {
      const int iy = oy0 + t - 1, ix = ox0 + 4 * m - 4;            // W % 4 == 0: a piece is inside or outside as a whole
      const bool ok = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
      p_ok[u] = ok ? 1 : 0;
      p_src[u] = ok ? ch * HWi + iy * W + ix : 0;
      p_a0[u] = 0;
    } else {
      // Z rows / columns of this piece: ny = oy0 + t, nx = ox0 + 4 m - 3 .. + 3 (nx is odd).  Low-resolution rows iy0, iy0 + 1
      // and columns j .. j + 3 cover all four elements (see fill_store); bit r * 4 + c of the mask = element (iy0 + r, j + c)
      // lies inside the image.
      const int ny = oy0 + t;
      const int a0 = (ny + 1) & 1;
      const int iy0 = (ny - 3 + a0) >> 1;
      const int j = (ox0 >> 1) + 2 * m - 3;
      int mask = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (live && iy0 + r >= 0 && iy0 + r < H && j + c >= 0 && j + c < W) mask |= 1 << (r * 4 + c);
      p_ok[u] = mask;
      p_src[u] = ch * HWi + iy0 * W + j;                           // may point outside: only masked-in elements are read
      p_a0[u] = a0;
    }
  }
  float raw[HALF][RAW];
  auto fill_load = [&](int st, int u0, int u1) {
    const float* src = xb + (int64_t)st * BK * HWi;                  // workgroup-uniform
#pragma unroll
    for (int uu = 0; uu < HALF; ++uu) {
      const int u = u0 + uu;
      if (u >= u1) continue;
      // opaque copies: without them the per-element addresses and predicates of every piece (7 x 8 of each in the
      // up-sampling form) are hoisted out of the stage loop and held in registers for the whole kernel
      int ps = p_src[u], pk = p_ok[u];
      asm volatile("" : "+v"(ps), "+v"(pk));
      if constexpr (!UP) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ps);
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[uu][c] = pk ? v[c] : 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool ok = (pk >> (r * 4 + c)) & 1;
            const float v = src[ok ? ps + r * W + c : 0];
            raw[uu][r * 4 + c] = ok ? v : 0.f;
          }
      }
    }
  };
  auto fill_store = [&](float* dst, int u0, int u1) {
#pragma unroll
    for (int uu = 0; uu < HALF; ++uu) {
      const int u = u0 + uu;
      if (u >= u1 || p_lds[u] < 0) continue;
      f32x4 v;
      if constexpr (!UP) {
        v = f32x4{raw[uu][0], raw[uu][1], raw[uu][2], raw[uu][3]};
      } else {
        // Z[ny][nx] = sum_{a,b} kf[a][b] xs[ny-3+a][nx-3+b], xs = zero-stuffed x: only a = a0, a0 + 2 with a0 = (ny+1)&1 hit
        // even rows (low-resolution rows iy0, iy0 + 1), likewise for columns.  For the four columns nx .. nx + 3 (nx odd):
        //   e = 0: b0 = 0, columns j, j+1;  e = 1: b0 = 1, columns j+1, j+2;  e = 2: b0 = 0, j+1, j+2;  e = 3: b0 = 1, j+2, j+3
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int b0 = e & 1, c0 = (e + 1) >> 1;
          float acc = 0.f;
#pragma unroll
          for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const float tap = p_a0[u] ? kf[(1 + 2 * da) * 4 + b0 + 2 * db] : kf[(2 * da) * 4 + b0 + 2 * db];
              acc = fmaf(raw[uu][da * 4 + c0 + db], tap, acc);
            }
          v[e] = acc;
        }
      }
      *reinterpret_cast<f32x4*>(dst + p_lds[u]) = v;
    }
  };

  // ---- A fragments: [tap][ot][kq][256], the three taps of one kernel row at a time
  f32x4 afr[3][WM], afr_next[3][WM];
  auto a_load = [&](int st, int ky, f32x4 (&dst)[3][WM]) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < WM; ++i)
        dst[kx][i] = *reinterpret_cast<const f32x4*>(
            ab + ((((ky * 3 + kx) * (a.Cout >> 4) + (m0 >> 4) + i) * (K >> 4) + st) * 256 + lane * 4));
  };

  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  a_load(0, 0, afr_next);
  fill_load(0, 0, HALF);
  fill_store(sT, 0, HALF);
  fill_load(0, HALF, CPT);
  fill_store(sT, HALF, CPT);
  __syncthreads();

#pragma unroll 1
  for (int st = 0; st < nstage; ++st) {
    const float* cur = sT + (st & 1) * STAGE + wave * P + 4 * jn + 3;
    const bool more = st + 1 < nstage;
    float* nxt = sT + ((st + 1) & 1) * STAGE;          // free since the barrier that ended stage st - 1
    if (more) fill_load(st + 1, 0, HALF);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (ky == 1 && more) {
        fill_store(nxt, 0, HALF);
        fill_load(st + 1, HALF, CPT);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < WM; ++i) afr[kx][i] = afr_next[kx][i];
      if (ky < 2) a_load(st, ky + 1, afr_next);
      else if (st + 1 < nstage) a_load(st + 1, 0, afr_next);
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const float* rowp = cur + (4 * j4 + q) * CS + ky * P;
        const float vm = rowp[0];
        const f32x4 v = *reinterpret_cast<const f32x4*>(rowp + 1);
        const float vp = rowp[5];
        const float win[6] = {vm, v[0], v[1], v[2], v[3], vp};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[kx][i][j4], win[c + kx], acc[i][c], 0, 0, 0);
      }
    }
    if (more) fill_store(nxt, HALF, CPT);
    __syncthreads();
  }

  // ---- epilogue.  D layout: acc[i][c][r] = out[o = m0 + 16 i + 4 q + r][pixel (oy, ox + c)]
  const int oy = oy0 + wave, ox = ox0 + 4 * jn;
  if (oy >= OH || ox >= OW) return;
  const int HWo = OH * OW;
  f32x4 nz = {0.f, 0.f, 0.f, 0.f};
  float nw = 0.f;
  if (a.epilogue == 1 && a.noise && a.noise_w) {
    nz = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)b * a.noise_bstride + (oy * OW + ox));
    nw = a.noise_w[0];
  }
  float* ob = a.out + (int64_t)b * a.Cout * HWo + (oy * OW + ox);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int obase = m0 + 16 * i + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
      if (a.epilogue == 1) {
        const float bs = a.bias[obase + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = lrelu02(fmaf(nw, nz[c], v[c]) + bs) * 1.41421356237309515f;
      }
      *reinterpret_cast<f32x4*>(ob + (obase + r) * HWo) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same convolution on split-fp16 products (the decoder's default arithmetic: three exact fp16 x fp16 products per fp32 product
// on v_mfma_f32_16x16x32_f16, fp32 accumulation -- decoder.hip / chain.hip; 5.3 x the fp32 matrix instruction's rate).
//   contraction index   a 32-deep MFMA step holds the 16 channels of a K stage for TWO taps: k = 16 h + ch, tap = 2 p + h for the
//                       tap pairs p = 0 .. 4 (taps in row-major order, the tenth tap is zero weights).  Lane quarter q supplies
//                       k = 8 q .. 8 q + 7: channels 8 (q & 1) .. + 7 of tap 2 p + (q >> 1).
//   weights             cips3d_modulate_weights(ksq = 9, PACKED | SPLIT [| FLIP]): [b][pair][o-tile][16-channel stage][hi | lo]
//                       [lane][8 fp16] of 2^8 w -- one 16-byte load per fragment, straight from L2, five pairs (one stage) ahead.
//   halo tile in LDS    staged ONCE per stage as B fragments: every thread converts 8 channels x 4 columns of one tile row (plain:
//                       eight 16-byte loads; up: the polyphase FIR first) and writes, per column, the 8 channels' hi halves and lo
//                       halves as one 16-byte element each:  T[channel group 2][hi | lo][row 6][column & 3][column >> 2][8 fp16].
//                       A lane's fragment for (tap, column tile c) is the element of column 4 jn + c + kx: consecutive lanes read
//                       consecutive 16-byte elements (conflict-free), no conversion and no shuffling in the main loop.
//   range               x is split as x 2^-e, e from the measured maximum of the sample (cips3d_range; up: times the FIR's gain
//                       bound sum |taps|); the accumulators come back through 2^-8 2^e, exactly.
// Same tile (4 rows x 64 columns x 16 WM channels, four waves), same epilogue as the fp32 kernel above.
// ------------------------------------------------------------------------------------------------
template <int WM, bool UP>
__global__ void __launch_bounds__(256) modconv3x3_split_kernel(Conv3Args a) {
  typedef cips3d_h8 h8;
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  constexpr int TH = 4, TW = 64, BK = 16;
  constexpr int NG = (TW + 8) / 4;                     // 18 groups of 4 columns: column index 3 = left halo, 4 .. 67 = the tile, 68 = right halo
  constexpr int ROWS = TH + 2;
  constexpr int PLANE = ROWS * 4 * NG;                 // 16-byte elements of one (channel group, plane)
  constexpr int STAGE = 4 * PLANE;                     // ... of one 16-channel stage (27 KB)
  constexpr int PIECES = 2 * ROWS * NG;                // staging pieces of a stage: (channel group, row, column group) -- one per thread
  static_assert(PIECES <= 256, "one staging piece per thread");
  __shared__ __attribute__((aligned(16))) u32x4_t sT[2 * STAGE];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // = tile row
  const int lane = tid & 63;
  const int q = lane >> 4, jn = lane & 15;
  const int b = blockIdx.z;
  const int H = a.H, W = a.W;
  const int OH = UP ? 2 * H : H, OW = UP ? 2 * W : W;
  const int tiles_x = (OW + TW - 1) / TW;
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);      // XCD-aware tile order (as above)
  const int ox0 = (bid % tiles_x) * TW, oy0 = (bid / tiles_x) * TH;
  const int m0 = blockIdx.y * (16 * WM);
  const int K = a.Cin, nstage = K / BK;
  const int HWi = H * W;
  const float* xb = a.x + (int64_t)b * K * HWi;
  // [b][pair 5][o-tile][stage][plane 2][lane 64][8 halfs]: 2 KB per (pair, o-tile, stage)
  const _Float16* ab = reinterpret_cast<const _Float16*>(a.wmp) + (int64_t)b * 5 * a.Cout * K * 4;      // (two taps x two planes per weight)

  float kf[16];
  float fgain = 1.f;
  if constexpr (UP) {
    fgain = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { kf[i] = a.fir[15 - i]; fgain += fabsf(kf[i]); }
  }
  // range: x (up: the FIR's output, |Z| <= sum |taps| max|x|) is split as v 2^-e
  float kx = 1.f, kin = 1.f / 256.f;
  if (a.x_amax) {
    const int e = cips3d_split_exp(cips3d_amax_load(a.x_amax + b * CIPS3D_AMAX_FLOATS) * fgain);
    kx = cips3d_uniform(cips3d_pow2(-e));
    kin = cips3d_uniform((1.f / 256.f) * cips3d_pow2(e));
  }

  // ---- staging piece of this thread: channel group cg, tile row t, column group m (columns 4 m .. 4 m + 3)
  const bool live = tid < PIECES;
  const int s_cg = tid / (ROWS * NG), s_rem = tid % (ROWS * NG);
  const int s_t = s_rem / NG, s_m = s_rem % NG;
  const int s_lds = ((s_cg * 2) * ROWS + s_t) * 4 * NG + s_m;         // element of (plane 0, phase 0); + e NG per column, + PLANE for lo
  int s_src = 0, s_mask = 0, s_a0 = 0;
  if constexpr (!UP) {
    const int iy = oy0 + s_t - 1, ix = ox0 + 4 * s_m - 4;              // W % 4 == 0: a piece is inside or outside as a whole
    const bool ok = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
    s_mask = ok ? 1 : 0;
    s_src = ok ? (8 * s_cg) * HWi + iy * W + ix : 0;
  } else {
    const int ny = oy0 + s_t;
    s_a0 = (ny + 1) & 1;
    const int iy0 = (ny - 3 + s_a0) >> 1;
    const int j = (ox0 >> 1) + 2 * s_m - 3;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (live && iy0 + r >= 0 && iy0 + r < H && j + c >= 0 && j + c < W) s_mask |= 1 << (r * 4 + c);
    s_src = (8 * s_cg) * HWi + iy0 * W + j;                             // may point outside: only masked-in elements are read
  }
  // The piece's 8 channels are staged in NR rounds of CR channels (up: two rounds -- a channel's 2 x 4 low-resolution neighbourhood
  // is 8 registers; 32 staging registers either way): load at one point of the stage, FIR + split + LDS store at a later one.
  constexpr int RAWC = UP ? 8 : 4;
  constexpr int NR = UP ? 2 : 1, CR = 8 / NR;
  float raw[CR][RAWC];
  auto fill_load = [&](int st, int rd) {
    const float* src = xb + ((int64_t)st * BK + rd * CR) * HWi;
    // opaque copies: without them the per-element addresses and predicates (8 of each per channel in the up-sampling form) are
    // hoisted out of the stage loop and held in registers for the whole kernel
    int ps = s_src, pk = s_mask;
    asm volatile("" : "+v"(ps), "+v"(pk));
#pragma unroll
    for (int ch = 0; ch < CR; ++ch) {
      if constexpr (!UP) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ps + ch * HWi);
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[ch][c] = pk ? v[c] : 0.f;
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool ok = (pk >> (r * 4 + c)) & 1;
            const float v = src[ok ? ps + ch * HWi + r * W + c : 0];
            raw[ch][r * 4 + c] = ok ? v : 0.f;
          }
      }
    }
  };
  auto fill_store = [&](u32x4_t* dst, int rd) {
    if (!live) return;
    float z[CR][4];
#pragma unroll
    for (int ch = 0; ch < CR; ++ch) {
      if constexpr (!UP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[ch][e] = raw[ch][e] * kx;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {           // (see the fp32 kernel's fill_store for the polyphase indices)
          const int b0 = e & 1, c0 = (e + 1) >> 1;
          float acc = 0.f;
#pragma unroll
          for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const float tap = s_a0 ? kf[(1 + 2 * da) * 4 + b0 + 2 * db] : kf[(2 * da) * 4 + b0 + 2 * db];
              acc = fmaf(raw[ch][da * 4 + c0 + db], tap, acc);
            }
          z[ch][e] = acc * kx;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hi[CR / 2], lo[CR / 2];
#pragma unroll
      for (int jj = 0; jj < CR / 2; ++jj) cips3d_split_pair(z[2 * jj][e], z[2 * jj + 1][e], hi[jj], lo[jj]);
      if constexpr (NR == 1) {
        dst[s_lds + e * NG] = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
        dst[s_lds + e * NG + PLANE] = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
      } else {          // half an element per round: channels 4 rd .. 4 rd + 3 = bytes 8 rd .. 8 rd + 7
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t* d2 = reinterpret_cast<u32x2_t*>(dst);
        d2[2 * (s_lds + e * NG) + rd] = u32x2_t{hi[0], hi[1]};
        d2[2 * (s_lds + e * NG + PLANE) + rd] = u32x2_t{lo[0], lo[1]};
      }
    }
  };

  // ---- B-fragment elements of this lane: tap 2 p + (q >> 1) -> (ky, kx); column tile c -> column 4 jn + c + kx + 3
  // (the tenth tap has zero weights: its lanes read the ninth tap's elements)
  const int h = q >> 1, cgl = q & 1;
  int b_el[5][4];
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const int tap = (2 * p + h) < 9 ? 2 * p + h : 8;
    const int ky = tap / 3, kxl = tap % 3;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c + kxl + 3;
      b_el[p][c] = (((cgl * 2) * ROWS + wave + ky) * 4 + (col & 3)) * NG + jn + (col >> 2);
    }
  }

  // ---- A fragments: one buffer per tap pair, refilled for the next stage right behind its last use (five pairs of lead)
  h8 ah[5][WM], al[5][WM];
  auto a_load = [&](int st, int p) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const _Float16* src = ab + ((((int64_t)p * (a.Cout >> 4) + (m0 >> 4) + i) * (K >> 4) + st) * 2) * 512 + lane * 8;
      ah[p][i] = *reinterpret_cast<const h8*>(src);
      al[p][i] = *reinterpret_cast<const h8*>(src + 512);
    }
  };

  f32x4 acc[WM][4];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < 5; ++p) a_load(0, p);
#pragma unroll
  for (int rd = 0; rd < NR; ++rd) {
    fill_load(0, rd);
    fill_store(sT, rd);
  }
  __syncthreads();

#pragma unroll 1
  for (int st = 0; st < nstage; ++st) {
    const u32x4_t* cur = sT + (st & 1) * STAGE;
    u32x4_t* nxt = sT + ((st + 1) & 1) * STAGE;          // free since the barrier that ended stage st - 1
    const bool more = st + 1 < nstage;
    if (more) fill_load(st + 1, 0);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      if (NR == 2 && p == 2 && more) {                   // second round of the next stage's piece under the remaining tap pairs
        fill_store(nxt, 0);
        fill_load(st + 1, 1);
      }
      h8 bh[4], bl[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bh[c] = __builtin_bit_cast(h8, cur[b_el[p][c]]);
        bl[c] = __builtin_bit_cast(h8, cur[b_el[p][c] + PLANE]);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[p][i], bh[c], acc[i][c], 0, 0, 0);
          acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p][i], bl[c], acc[i][c], 0, 0, 0);
          acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[p][i], bh[c], acc[i][c], 0, 0, 0);
        }
      if (more) a_load(st + 1, p);
    }
    if (more) fill_store(nxt, NR - 1);
    __syncthreads();
  }

  // ---- epilogue.  D layout: acc[i][c][r] = out[o = m0 + 16 i + 4 q + r][pixel (oy, ox + c)]
  const int oy = oy0 + wave, ox = ox0 + 4 * jn;
  if (oy >= OH || ox >= OW) return;
  const int HWo = OH * OW;
  f32x4 nz = {0.f, 0.f, 0.f, 0.f};
  float nw = 0.f;
  if (a.epilogue == 1 && a.noise && a.noise_w) {
    nz = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)b * a.noise_bstride + (oy * OW + ox));
    nw = a.noise_w[0];
  }
  float* ob = a.out + (int64_t)b * a.Cout * HWo + (oy * OW + ox);
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int obase = m0 + 16 * i + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 v = {acc[i][0][r] * kin, acc[i][1][r] * kin, acc[i][2][r] * kin, acc[i][3][r] * kin};
      if (a.epilogue == 1) {
        const float bs = a.bias[obase + r];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = lrelu02(fmaf(nw, nz[c], v[c]) + bs) * 1.41421356237309515f;
      }
      *reinterpret_cast<f32x4*>(ob + (obase + r) * HWo) = v;
    }
  }
}

template <int WM, bool UP>
int launch_conv3_split(const Conv3Args& a, hipStream_t st) {
  const int OH = UP ? 2 * a.H : a.H, OW = UP ? 2 * a.W : a.W;
  dim3 grid((unsigned)(((OW + 63) / 64) * ((OH + 3) / 4)), (unsigned)(a.Cout / (16 * WM)), (unsigned)a.B);
  hipLaunchKernelGGL((modconv3x3_split_kernel<WM, UP>), grid, dim3(256), 0, st, a);
  return cips3d_launch_status();
}

template <int WM, bool UP>
int launch_conv3(const Conv3Args& a, hipStream_t st) {
  const int OH = UP ? 2 * a.H : a.H, OW = UP ? 2 * a.W : a.W;
  dim3 grid((unsigned)(((OW + 63) / 64) * ((OH + 3) / 4)), (unsigned)(a.Cout / (16 * WM)), (unsigned)a.B);
  hipLaunchKernelGGL((modconv3x3_kernel<WM, UP>), grid, dim3(256), 0, st, a);
  return cips3d_launch_status();
}

}  // namespace

extern "C" int cips3d_modconv3x3_supported(int Cin, int Cout, int H, int W, int up) {
  const int64_t OH = up ? 2 * (int64_t)H : H, OW = up ? 2 * (int64_t)W : W;
  // 16-channel K stages and output tiles, 16-byte pieces (plain: W % 4; up: 2W % 4), 32-bit intra-sample offsets
  return Cin > 0 && Cout > 0 && H > 0 && W > 0 && Cin % 16 == 0 && Cout % 16 == 0 && OW % 4 == 0 && (up || W % 4 == 0) &&
         (int64_t)Cin * H * W < ((int64_t)1 << 31) && (int64_t)Cout * OH * OW < ((int64_t)1 << 31) &&
         (int64_t)9 * Cin * Cout < ((int64_t)1 << 31);
}

extern "C" int cips3d_modconv3x3(const float* x, const float* wm, float* out, int B, int Cin, int Cout, int H, int W, int up,
                                 const float* fir, int epilogue, const float* noise, int64_t noise_bstride,
                                 const float* noise_w, const float* bias, const cips3d_range* rg, void* stream) {
  if (!x || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (up && !fir) return CIPS3D_E_BADARG;
  const bool split = (epilogue & CIPS3D_GEMM_SPLIT) != 0;      // wm: cips3d_modulate_weights(ksq = 9, PACKED | SPLIT [| FLIP])
  epilogue &= ~CIPS3D_GEMM_SPLIT;
  if (epilogue != 0 && epilogue != 1) return CIPS3D_E_BADARG;
  if (epilogue == 1 && !bias) return CIPS3D_E_BADARG;
  if (!cips3d_modconv3x3_supported(Cin, Cout, H, W, up)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  Conv3Args a{x, wm, out, B, Cin, Cout, H, W, fir, epilogue, noise, noise_bstride, noise_w, bias, (split && rg) ? rg->x_amax : nullptr};
  hipStream_t st = as_stream(stream);
  const bool wide = Cout % 32 == 0;        // two output tiles per wave share every window read
  if (split) {
    if (up) return wide ? launch_conv3_split<2, true>(a, st) : launch_conv3_split<1, true>(a, st);
    return wide ? launch_conv3_split<2, false>(a, st) : launch_conv3_split<1, false>(a, st);
  }
  if (up) return wide ? launch_conv3<2, true>(a, st) : launch_conv3<1, true>(a, st);
  return wide ? launch_conv3<2, false>(a, st) : launch_conv3<1, false>(a, st);
}
