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
                                 const float* noise_w, const float* bias, void* stream) {
  if (!x || !wm || !out || B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return CIPS3D_E_BADARG;
  if (up && !fir) return CIPS3D_E_BADARG;
  if (epilogue != 0 && epilogue != 1) return CIPS3D_E_BADARG;
  if (epilogue == 1 && !bias) return CIPS3D_E_BADARG;
  if (!cips3d_modconv3x3_supported(Cin, Cout, H, W, up)) return CIPS3D_E_UNSUPP;
  if (B == 0) return 0;
  Conv3Args a{x, wm, out, B, Cin, Cout, H, W, fir, epilogue, noise, noise_bstride, noise_w, bias};
  hipStream_t st = as_stream(stream);
  const bool wide = Cout % 32 == 0;        // two output tiles per wave share every window read
  if (up) return wide ? launch_conv3<2, true>(a, st) : launch_conv3<1, true>(a, st);
  return wide ? launch_conv3<2, false>(a, st) : launch_conv3<1, false>(a, st);
}
