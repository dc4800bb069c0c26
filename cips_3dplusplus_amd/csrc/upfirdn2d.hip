// cips3d_upfirdn2d: zero-insertion up-sampling -> pad/crop -> FIR (true convolution) -> decimation.
// Replaces upfirdn2d_op.upfirdn2d (reference exp/op/upfirdn2d.cpp:12-23 and the tiled / large
// kernels of upfirdn2d_kernel.cu:49-369).
//
// Definition used here (independent derivation, checked against the oracle + goldens):
//   u[y][x]   = in[(y - pad_y0)/up_y][(x - pad_x0)/up_x]  when both divisions are exact and in range,
//               else 0                                    (coordinates of the padded up-sampled grid)
//   out[oy][ox] = sum_{ky,kx} u[oy*down_y + ky][ox*down_x + kx] * k[kh-1-ky][kw-1-kx]
// Only every up-th tap hits a non-zero sample (polyphase), so the inner loops step by `up`.
//
// HBM-bound: 4 B/elem in (x 1/up^2 .. down^2) + 4 B/elem out.  Three kernels: upfirdn2d_fast (4 x 4 FIR, up / down factors the
// generator and its discriminator use: compile-time polyphase structure, 16-byte LDS reads and stores), upfirdn2d_tiled (any
// factors and FIR up to 64 taps: stages the input footprint of a 32x64 output tile in LDS, the taps too; run-time tap stepping),
// upfirdn2d_generic (anything else, e.g. minor > 1).
#include <cstdlib>
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct UfdParams {
  int64_t major;
  int in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, out_h, out_w;
};

constexpr int TILE_OH = 32;
constexpr int TILE_OW = 64;
constexpr int MAX_TAPS = 64;          // kh*kw
constexpr int LDS_IN_FLOATS = 10240;  // 40 KB input footprint budget

__device__ __forceinline__ int first_phase(int base, int up) {
  // smallest k >= 0 with (base + k) % up == 0
  int m = base % up;
  if (m < 0) m += up;
  return m == 0 ? 0 : up - m;
}

__global__ void __launch_bounds__(256) upfirdn2d_tiled(const float* __restrict__ in,
                                                       const float* __restrict__ kernel,
                                                       float* __restrict__ out, UfdParams p,
                                                       int tiles_x, int tiles_y, int tin_h, int tin_w) {
  __shared__ float s_k[MAX_TAPS];
  __shared__ float s_in[LDS_IN_FLOATS];
  const int tid = threadIdx.x;
  int64_t blk = blockIdx.x;
  const int tx = (int)(blk % tiles_x); blk /= tiles_x;
  const int ty = (int)(blk % tiles_y); blk /= tiles_y;
  const int64_t m = blk;  // major index (minor == 1 on this path)
  const int oy0 = ty * TILE_OH, ox0 = tx * TILE_OW;

  // flipped taps: s_k[ky*kw+kx] = k[kh-1-ky][kw-1-kx]
  if (tid < p.kh * p.kw) {
    const int ky = tid / p.kw, kx = tid % p.kw;
    s_k[tid] = kernel[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
  }
  // input rows/cols that the tile can touch
  const int iy0 = floor_div_i(oy0 * p.down_y - p.pad_y0 + p.up_y - 1, p.up_y);  // ceil((..)/up)
  const int ix0 = floor_div_i(ox0 * p.down_x - p.pad_x0 + p.up_x - 1, p.up_x);
  const float* src = in + m * (int64_t)p.in_h * p.in_w;
  for (int i = tid; i < tin_h * tin_w; i += 256) {
    const int r = i / tin_w, c = i - r * tin_w;
    const int iy = iy0 + r, ix = ix0 + c;
    float v = 0.f;
    if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = src[(int64_t)iy * p.in_w + ix];
    s_in[i] = v;
  }
  __syncthreads();

  const int lx = tid & 63;       // column inside the tile
  const int ly0 = tid >> 6;      // 0..3; rows ly0, ly0+4, ...
  const int ox = ox0 + lx;
  if (ox >= p.out_w) return;
  const int bx = ox * p.down_x - p.pad_x0;   // padded-upsampled x of tap kx=0, minus pad
  const int kx0 = first_phase(bx, p.up_x);
  float* dst = out + m * (int64_t)p.out_h * p.out_w;
#pragma unroll 2
  for (int r = 0; r < TILE_OH / 4; ++r) {
    const int oy = oy0 + ly0 + r * 4;
    if (oy >= p.out_h) break;
    const int by = oy * p.down_y - p.pad_y0;
    float acc = 0.f;
    for (int ky = first_phase(by, p.up_y); ky < p.kh; ky += p.up_y) {
      const int iy = (by + ky) / p.up_y - iy0;   // exact division; may be outside the image -> zeros in LDS
      const float* row = s_in + iy * tin_w;
      const float* krow = s_k + ky * p.kw;
      for (int kx = kx0; kx < p.kw; kx += p.up_x) {
        const int ix = (bx + kx) / p.up_x - ix0;
        acc = fmaf(row[ix], krow[kx], acc);
      }
    }
    dst[(int64_t)oy * p.out_w + ox] = acc;
  }
}

// Fast path for the shapes the generator's Blur / Upsample (and the discriminator-side down-sampling) issue: minor == 1, a
// 4 x 4 FIR, (up, down) in {(1, 1), (2, 1), (1, 2)}.  Everything that depended on a run-time division in the tile kernel above
// is a compile-time constant here.  With the tile origin at ox0 (a multiple of the tile width) write ox0 * D - pad_x0 =
// U * q0 + RX, 0 <= RX < U: RX is the same for every tile (TW * D is a multiple of U), so it is a template parameter, and output
// column ox0 + 4 tx + j reads input columns q0 + 4 tx D / U + ceil((RX + j D) / U) + t with taps kx = (U - (RX + j D) % U) % U
// + U t, t = 0 .. ceil(K / U) - 1 -- offsets and tap numbers known per (j, t).  A thread owns a 4 x BY block of outputs: it
// walks the rows of its input window once (vector LDS reads, a row feeds every output row whose taps touch it), the 16 taps
// sit in scalar registers, the accumulation order per output is the tile kernel's (ky ascending, kx ascending: bit-identical
// results), stores are 16 bytes per lane.  Rows of the footprint are staged by whole waves (no division per element).
template <int U, int D, int K, int RX, int RY>
struct UfdFast {
  static constexpr int BX = 4, BY = (D == 2 ? 2 : 4);       // outputs per thread
  static constexpr int LX = 32, LY = 8;                       // threads per tile row / column
  static constexpr int TW = BX * LX, TH = BY * LY;            // 128 x 32 outputs (x 16 when D == 2)
  static constexpr int NT = (K + U - 1) / U;                  // taps per output and dimension that can hit a sample
  static constexpr int off(int r, int j) { return (r + j * D + U - 1) / U; }
  static constexpr int k0(int r, int j) { return (U - (r + j * D) % U) % U; }
  static constexpr int WX = off(RX, BX - 1) + NT, WY = off(RY, BY - 1) + NT;      // a thread's input window
  static constexpr int AL = ((BX * D / U) % 4 == 0) ? 4 : 2;  // floats the window start is aligned to
  static constexpr int WXP = (WX + AL - 1) / AL * AL;
  static constexpr int FW = TW * D / U + NT, FH = TH * D / U + NT;                // footprint of a tile
  static constexpr int PITCH = (FW + AL + 3) / 4 * 4;
  static_assert((BX * D) % U == 0 && (BY * D) % U == 0 && PITCH * FH * 4 <= 48 * 1024, "tile shape");
};

template <int U, int D, int K, int RX, int RY, bool NT>
__global__ void __launch_bounds__(256) upfirdn2d_fast(const float* __restrict__ in, const float* __restrict__ kernel,
                                                      float* __restrict__ out, UfdParams p, int tiles_x, int tiles_y) {
  typedef UfdFast<U, D, K, RX, RY> T;
  __shared__ __attribute__((aligned(16))) float s_in[T::PITCH * T::FH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t blk = blockIdx.x;
  const int tx_i = (int)(blk % tiles_x); blk /= tiles_x;
  const int ty_i = (int)(blk % tiles_y); blk /= tiles_y;
  const int64_t m = blk;
  const int ox0 = tx_i * T::TW, oy0 = ty_i * T::TH;
  const int q0x = floor_div_i(ox0 * D - p.pad_x0, U), q0y = floor_div_i(oy0 * D - p.pad_y0, U);
  const float* src = in + m * (int64_t)p.in_h * p.in_w;
  // flipped taps, uniform addresses: scalar registers
  float kf[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) kf[i] = kernel[K * K - 1 - i];
  // footprint -> LDS: element e = tid + 256 j is (row e / PITCH, column e % PITCH); 8 or 16 loads are in flight per lane before
  // the first LDS store (as a load -> store loop per element the read-dominated shapes -- blur, down-sampling -- ran at 1.7-2.8 TB/s)
  {
    constexpr int TOTAL = T::PITCH * T::FH, PER = (TOTAL + 255) / 256, UNR = D == 2 ? 8 : 16;   // (same-box sweep of 4 / 8 / 16: blur 4.5 / 4.5 / 5.4 TB/s, down-sampling 2.8 / 3.0 / 2.9)
#pragma unroll 1
    for (int j0 = 0; j0 < PER; j0 += UNR) {
      float v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = tid + 256 * (j0 + u);
        const int r = e / T::PITCH, c = e - r * T::PITCH;
        const int iy = q0y + r, ix = q0x + c;
        const bool ok = e < TOTAL && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        v[u] = ok ? src[(int64_t)iy * p.in_w + ix] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int e = tid + 256 * (j0 + u);
        if (e < TOTAL) s_in[e] = v[u];
      }
    }
  }
  __syncthreads();
  const int tx = tid % T::LX, ty = tid / T::LX;
  const float* win0 = s_in + (T::BY * ty * D / U) * T::PITCH + T::BX * tx * D / U;
  float acc[T::BY][T::BX];
#pragma unroll
  for (int i = 0; i < T::BY; ++i)
#pragma unroll
    for (int j = 0; j < T::BX; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int wy = 0; wy < T::WY; ++wy) {
    float win[T::WXP];
#pragma unroll
    for (int c = 0; c < T::WXP; c += T::AL) {
      if (T::AL == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(win0 + wy * T::PITCH + c);
        win[c] = v[0]; win[c + 1] = v[1]; win[c + 2] = v[2]; win[c + 3] = v[3];
      } else {
        const float2 v = *reinterpret_cast<const float2*>(win0 + wy * T::PITCH + c);
        win[c] = v.x; win[c + 1] = v.y;
      }
    }
#pragma unroll
    for (int i = 0; i < T::BY; ++i) {
      const int t = wy - T::off(RY, i);
      const int ky = T::k0(RY, i) + U * t;
      if (t < 0 || t >= T::NT || ky >= K) continue;
#pragma unroll
      for (int j = 0; j < T::BX; ++j)
#pragma unroll
        for (int tt = 0; tt < T::NT; ++tt) {
          const int kx = T::k0(RX, j) + U * tt;
          if (kx < K) acc[i][j] = fmaf(win[T::off(RX, j) + tt], kf[ky * K + kx], acc[i][j]);
        }
    }
  }
  float* dst = out + m * (int64_t)p.out_h * p.out_w;
  const int ox = ox0 + T::BX * tx;
  const bool vec = (p.out_w & 3) == 0 && ox + 3 < p.out_w;
#pragma unroll
  for (int i = 0; i < T::BY; ++i) {
    const int oy = oy0 + T::BY * ty + i;
    if (oy >= p.out_h) break;
    float* o = dst + (int64_t)oy * p.out_w + ox;
    if (vec) {
      // NT: an output that cannot stay in the 256 MB Infinity Cache anyway leaves with non-temporal stores (same-box A/B on a
      // 134 MB -> 537 MB up-sampling: 3.6 -> 5.6 TB/s; no difference below ~270 MB of traffic, where the default policy keeps
      // the result on-die for its consumer; non-temporal LOADS of the input lost 20-25 % everywhere)
      if (NT) __builtin_nontemporal_store(f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]}, reinterpret_cast<f32x4*>(o));
      else *reinterpret_cast<f32x4*>(o) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    } else {
#pragma unroll
      for (int j = 0; j < T::BX; ++j)
        if (ox + j < p.out_w) o[j] = acc[i][j];
    }
  }
}

template <int U, int D, int RX, int RY>
void launch_fast(const float* in, const float* kernel, float* out, const UfdParams& p, hipStream_t st) {
  typedef UfdFast<U, D, 4, RX, RY> T;
  const int tiles_x = ceil_div(p.out_w, T::TW), tiles_y = ceil_div(p.out_h, T::TH);
  const dim3 grid((unsigned)((int64_t)tiles_x * tiles_y * p.major));
  if ((int64_t)p.major * p.out_h * p.out_w * 4 > (int64_t)192 << 20)
    hipLaunchKernelGGL((upfirdn2d_fast<U, D, 4, RX, RY, true>), grid, dim3(256), 0, st, in, kernel, out, p, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL((upfirdn2d_fast<U, D, 4, RX, RY, false>), grid, dim3(256), 0, st, in, kernel, out, p, tiles_x, tiles_y);
}

// Any size / any minor: one thread per output element, taps read straight from global/L2.
__global__ void __launch_bounds__(256) upfirdn2d_generic(const float* __restrict__ in,
                                                         const float* __restrict__ kernel,
                                                         float* __restrict__ out, UfdParams p,
                                                         int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    int64_t t = idx;
    const int mi = (int)(t % p.minor); t /= p.minor;
    const int ox = (int)(t % p.out_w); t /= p.out_w;
    const int oy = (int)(t % p.out_h); t /= p.out_h;
    const int64_t m = t;
    const int by = oy * p.down_y - p.pad_y0, bx = ox * p.down_x - p.pad_x0;
    float acc = 0.f;
    for (int ky = first_phase(by, p.up_y); ky < p.kh; ky += p.up_y) {
      const int iy = (by + ky) / p.up_y;
      if (by + ky < 0 || iy >= p.in_h) continue;
      for (int kx = first_phase(bx, p.up_x); kx < p.kw; kx += p.up_x) {
        const int ix = (bx + kx) / p.up_x;
        if (bx + kx < 0 || ix >= p.in_w) continue;
        const float v = in[((m * p.in_h + iy) * (int64_t)p.in_w + ix) * p.minor + mi];
        acc = fmaf(v, kernel[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)], acc);
      }
    }
    out[idx] = acc;
  }
}

}  // namespace

extern "C" int cips3d_upfirdn2d(const float* input, const float* kernel, float* out, int64_t major,
                                int in_h, int in_w, int minor, int kernel_h, int kernel_w, int up_x,
                                int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                int pad_y1, void* stream) {
  if (major == 0) return 0;                       // empty batch: nothing to do (the tensors may have no storage)
  if (!input || !kernel || !out) return CIPS3D_E_BADARG;
  if (major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kernel_h <= 0 || kernel_w <= 0 ||
      up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0)
    return CIPS3D_E_BADARG;
  UfdParams p;
  p.major = major; p.in_h = in_h; p.in_w = in_w; p.minor = minor; p.kh = kernel_h; p.kw = kernel_w;
  p.up_x = up_x; p.up_y = up_y; p.down_x = down_x; p.down_y = down_y; p.pad_x0 = pad_x0; p.pad_y0 = pad_y0;
  const int span_h = in_h * up_y + pad_y0 + pad_y1 - kernel_h;
  const int span_w = in_w * up_x + pad_x0 + pad_x1 - kernel_w;
  if (span_h < 0 || span_w < 0) return CIPS3D_E_BADARG;
  p.out_h = span_h / down_y + 1;
  p.out_w = span_w / down_x + 1;
  if (major == 0) return 0;
  hipStream_t st = as_stream(stream);

  // input footprint of one output tile
  const int tin_h = ((TILE_OH - 1) * down_y + kernel_h - 1) / up_y + 2;
  const int tin_w = ((TILE_OW - 1) * down_x + kernel_w - 1) / up_x + 2;
  const bool tiled = minor == 1 && kernel_h * kernel_w <= MAX_TAPS && tin_h * tin_w <= LDS_IN_FLOATS;
  static const bool fast_on = [] { const char* e = getenv("CIPS3D_UPFIRDN_FAST"); return !(e && e[0] == '0'); }();   // A/B and test knob
  const bool fast = fast_on && minor == 1 && kernel_h == 4 && kernel_w == 4 && up_x == up_y && down_x == down_y &&
                    ((up_x == 1 && down_x <= 2) || (up_x == 2 && down_x == 1)) &&
                    (int64_t)ceil_div(p.out_w, 128) * ceil_div(p.out_h, 16) * major <= 0x7fffffffLL &&
                    (int64_t)p.out_w * down_x + 4 + (pad_x0 < 0 ? -(int64_t)pad_x0 : pad_x0) < (1 << 30) &&
                    (int64_t)p.out_h * down_y + 4 + (pad_y0 < 0 ? -(int64_t)pad_y0 : pad_y0) < (1 << 30);
  if (fast) {
    if (up_x == 1 && down_x == 1) launch_fast<1, 1, 0, 0>(input, kernel, out, p, st);
    else if (up_x == 1) launch_fast<1, 2, 0, 0>(input, kernel, out, p, st);
    else {
      const int rx = ((-pad_x0) % 2 + 2) % 2, ry = ((-pad_y0) % 2 + 2) % 2;
      if (rx == 0 && ry == 0) launch_fast<2, 1, 0, 0>(input, kernel, out, p, st);
      else if (rx == 1 && ry == 0) launch_fast<2, 1, 1, 0>(input, kernel, out, p, st);
      else if (rx == 0) launch_fast<2, 1, 0, 1>(input, kernel, out, p, st);
      else launch_fast<2, 1, 1, 1>(input, kernel, out, p, st);
    }
  } else if (tiled) {
    const int tiles_x = ceil_div(p.out_w, TILE_OW), tiles_y = ceil_div(p.out_h, TILE_OH);
    const int64_t blocks = (int64_t)tiles_x * tiles_y * major;
    if (blocks > 0x7fffffffLL) return CIPS3D_E_UNSUPP;
    hipLaunchKernelGGL(upfirdn2d_tiled, dim3((unsigned)blocks), dim3(256), 0, st, input, kernel, out, p,
                       tiles_x, tiles_y, tin_h, tin_w);
  } else {
    const int64_t total = major * p.out_h * (int64_t)p.out_w * minor;
    int64_t blocks = ceil_div<int64_t>(total, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(upfirdn2d_generic, dim3((unsigned)blocks), dim3(256), 0, st, input, kernel, out, p,
                       total);
  }
  return cips3d_launch_status();
}
