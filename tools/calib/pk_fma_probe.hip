// Stand-alone probe of the packed-fp32 pattern hipcc's SLP vectoriser makes of csrc/chain.hip's ToRGB fold (the build whose
// partial sums differed from run to run at two workgroups per CU, DESIGN.md / profiles/r04_pk_fold_probe.txt).  The same
// arithmetic -- MFMA accumulators -> scale, noise, bias, leaky ReLU -> three channels of sum_r w[ch][r] v[r] per pixel ->
// cross-quarter shuffles -> LDS exchange over the wave rows -- is compiled TWICE from this file, once with SLP vectorisation
// (v_pk_fma_f32 chains: kernel probe_slp) and once without (v_fmac_f32: probe_ref); the host compares every sum of every
// launch with the first launch of the same kernel (determinism) and the two kernels with each other.
//   bash tools/calib/pk_fma_probe.sh        (builds both objects, links, runs)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#ifndef KNAME
#define KNAME probe_ref
#endif
constexpr int WM = 2, WGM = 4, WGN = 2, BN = 128;

extern "C" __global__ void __launch_bounds__(512, 4) KNAME(const float* __restrict__ in, const float* __restrict__ wr,
                                                          float* __restrict__ out, int rounds) {
  __shared__ float lds[24 * 1024 / 4];          // two workgroups per CU, as the failing kernel had
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, nl = lane & 15;
  const int wm_i = wave / WGN, wn_i = wave % WGN;
  f32x4 acc[WM][4];
  const h8 a = {1, 2, 3, 4, 5, 6, 7, 8};
  h8 b;
  for (int e = 0; e < 8; ++e) b[e] = (_Float16)in[(blockIdx.x * 512 + tid) * 8 + e];
  for (int i = 0; i < WM; ++i)
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wrgb[WM][3], bias4[WM];
  float nz[4];
  for (int i = 0; i < WM; ++i) {
    for (int ch = 0; ch < 3; ++ch) wrgb[i][ch] = *reinterpret_cast<const f32x4*>(wr + ch * 512 + (wm_i * WM + i) * 16 + 4 * q);
    bias4[i] = *reinterpret_cast<const f32x4*>(wr + 1536 + (wm_i * WM + i) * 16 + 4 * q);
  }
  for (int c = 0; c < 4; ++c) nz[c] = wr[2048 + (wn_i * 64 + nl + 16 * c)];
  for (int rd = 0; rd < rounds; ++rd) {
    _Pragma("unroll")
    for (int k = 0; k < 8; ++k)
      _Pragma("unroll")
      for (int i = 0; i < WM; ++i)
        _Pragma("unroll")
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i][c], 0, 0, 0);
    float prgb[3][4];
    _Pragma("unroll")
    for (int ch = 0; ch < 3; ++ch)
      _Pragma("unroll")
      for (int c = 0; c < 4; ++c) prgb[ch][c] = 0.f;
    const float kpre = 1.f / 256.f, kact = 1.41421356f, nw = 0.2f;
    _Pragma("unroll")
    for (int i = 0; i < WM; ++i)
      _Pragma("unroll")
      for (int c = 0; c < 4; ++c) {
        float v[4];
        _Pragma("unroll")
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][c][r] * kpre;
          v[r] = fmaxf((v[r] + nz[c] * nw) + bias4[i][r], 0.2f * ((v[r] + nz[c] * nw) + bias4[i][r])) * kact;
        }
        _Pragma("unroll")
        for (int ch = 0; ch < 3; ++ch)
          _Pragma("unroll")
          for (int r = 0; r < 4; ++r) prgb[ch][c] = fmaf(wrgb[i][ch][r], v[r], prgb[ch][c]);
      }
    _Pragma("unroll")
    for (int ch = 0; ch < 3; ++ch)
      _Pragma("unroll")
      for (int c = 0; c < 4; ++c) {
        float v = prgb[ch][c];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        prgb[ch][c] = v;
      }
    __syncthreads();
    if (q == 0)
      _Pragma("unroll")
      for (int ch = 0; ch < 3; ++ch)
        _Pragma("unroll")
        for (int c = 0; c < 4; ++c) lds[(wm_i * 3 + ch) * BN + wn_i * 64 + nl + 16 * c] = prgb[ch][c];
    __syncthreads();
    if (wm_i == 0 && q < 3)
      _Pragma("unroll")
      for (int c = 0; c < 4; ++c) {
        float v = 0.f;
        _Pragma("unroll")
        for (int m = 0; m < WGM; ++m) v += lds[(m * 3 + q) * BN + wn_i * 64 + nl + 16 * c];
        out[((size_t)(blockIdx.x * rounds + rd) * 3 + q) * BN + wn_i * 64 + nl + 16 * c] = v;
      }
    __syncthreads();
  }
}

#ifdef PROBE_MAIN
#include <cstdio>
#include <cstring>
#include <vector>
extern "C" __global__ void probe_slp(const float*, const float*, float*, int);
extern "C" __global__ void probe_ref(const float*, const float*, float*, int);
int main() {
  const int wgs = 2048, rounds = 8;
  std::vector<float> hin((size_t)wgs * 512 * 8), hw(4096);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
  for (auto& x : hin) x = rnd();
  for (auto& x : hw) x = rnd();
  const size_t n_out = (size_t)wgs * rounds * 3 * BN;
  float *d_in, *d_w, *d_out;
  hipMalloc(&d_in, hin.size() * 4); hipMalloc(&d_w, hw.size() * 4); hipMalloc(&d_out, n_out * 4);
  hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> first[2], cur(n_out);
  for (int k = 0; k < 2; ++k) {
    int bad_runs = 0; size_t worst = 0;
    for (int rep = 0; rep < 40; ++rep) {
      hipMemset(d_out, 0xff, n_out * 4);
      if (k == 0) hipLaunchKernelGGL(probe_slp, dim3(wgs), dim3(512), 0, 0, d_in, d_w, d_out, rounds);
      else hipLaunchKernelGGL(probe_ref, dim3(wgs), dim3(512), 0, 0, d_in, d_w, d_out, rounds);
      hipMemcpy(cur.data(), d_out, n_out * 4, hipMemcpyDeviceToHost);
      if (rep == 0) { first[k] = cur; continue; }
      size_t nd = 0;
      for (size_t i = 0; i < n_out; ++i) nd += memcmp(&cur[i], &first[k][i], 4) != 0;
      bad_runs += nd > 0; worst = nd > worst ? nd : worst;
    }
    printf("%s: %d of 39 repeats differ from the first launch (worst: %zu of %zu sums)\n", k == 0 ? "probe_slp (v_pk_fma_f32 chains)" : "probe_ref (v_fmac_f32)      ", bad_runs, worst, n_out);
  }
  size_t nd = 0;
  for (size_t i = 0; i < n_out; ++i) nd += memcmp(&first[0][i], &first[1][i], 4) != 0;
  printf("first launches of the two kernels: %zu of %zu sums differ\n", nd, n_out);
  return 0;
}
#endif
