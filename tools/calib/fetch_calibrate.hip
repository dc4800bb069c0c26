// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this package's kernels use (MI355X_MICROARCH.md, HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel streams
// the same 1 GiB buffer (4x the Infinity Cache) exactly once with a different load width / pattern and folds the values into
// a checksum so that nothing is optimised away:
//   read_b32 / read_b64 / read_b128   one coalesced load of 4 / 8 / 16 bytes per lane
//   read_fir_patch                    the FIR patch pattern of the fused up-sampling stages: per lane a float2 at an even
//                                     column plus the two single floats around it (row[-1], row[2]) -> every element is
//                                     requested twice, by neighbouring lanes
//   read_b16_pairs                    bf16 storage mode: one 32-bit load + two 16-bit loads per lane
// build: hipcc -O3 --offload-arch=gfx950 tools/calib/fetch_calibrate.hip -o tools/calib/fetch_calibrate
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- tools/calib/fetch_calibrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void read_b32(const float* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void read_b64(const float2* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; acc += v.x + v.y; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void read_b128(const float4* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
// lane i: float2 at column 2 i, floats at 2 i - 1 and 2 i + 2 (n2 = number of float2 positions)
__global__ void read_fir_patch(const float* __restrict__ p, size_t n2, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const float* row = p + 2 * i;
    const float2 mid = *reinterpret_cast<const float2*>(row);
    acc += mid.x + mid.y + (i > 0 ? row[-1] : 0.f) + (i + 1 < n2 ? row[2] : 0.f);
  }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void read_b16_pairs(const unsigned short* __restrict__ p, size_t n2, float* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned short* row = p + 2 * i;
    acc += *reinterpret_cast<const unsigned*>(row) + (i > 0 ? row[-1] : 0u) + (i + 1 < n2 ? row[2] : 0u);
  }
  if (acc == 0x12345678u) out[0] = (float)acc;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void* buf; float* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(buf, 0, bytes));
  const dim3 grid(256 * 8), block(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_b32, grid, block, 0, 0, (const float*)buf, bytes / 4, out);
    hipLaunchKernelGGL(read_b64, grid, block, 0, 0, (const float2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read_b128, grid, block, 0, 0, (const float4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(read_fir_patch, grid, block, 0, 0, (const float*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read_b16_pairs, grid, block, 0, 0, (const unsigned short*)buf, bytes / 4, out);
  }
  CHECK(hipDeviceSynchronize());
  printf("each kernel streamed %zu bytes once\n", bytes);
  return 0;
}
