// ABI bookkeeping + cips3d_camera_params (reference cips3d/nerf_utils.py:344-436 and the custom
// up-vector variant :466-564, `locations` branch): look-at pose on the unit sphere.
// One thread per view; a few dozen flops each -- launch-latency bound, nothing to tile.
#include "common.h"

extern "C" int cips3d_abi_version(void) { return CIPS3D_ABI_VERSION; }

extern "C" const char* cips3d_strerror(int code) {
  if (code == 0) return "success";
  if (code == CIPS3D_E_BADARG) return "cips3d: bad argument (null pointer or non-positive size)";
  if (code == CIPS3D_E_UNSUPP) return "cips3d: configuration not supported by the gfx950 kernels";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "cips3d: unknown error";
}

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 unit3(V3 a, float eps) {
  const float n = fmaxf(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), eps);
  return V3{a.x / n, a.y / n, a.z / n};
}

__global__ void camera_kernel(const float* __restrict__ loc, const float* __restrict__ fov_deg, float fov_s,
                              const float* __restrict__ up, float dist_radius, int img_size, int B,
                              float* __restrict__ extr, float* __restrict__ focal, float* __restrict__ near_,
                              float* __restrict__ far_) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float azim = loc[2 * b], elev = loc[2 * b + 1];
  const float fov = (fov_deg ? fov_deg[b] : fov_s) * 3.14159265358979323846f / 180.f;
  focal[b] = 0.5f * (float)img_size / tanf(fov);
  near_[b] = 1.f - dist_radius;
  far_[b] = 1.f + dist_radius;
  const V3 dir{cosf(elev) * sinf(azim), sinf(elev), cosf(elev) * cosf(azim)};
  const V3 upv = up ? V3{up[3 * b], up[3 * b + 1], up[3 * b + 2]} : V3{0.f, 1.f, 0.f};
  const V3 zax = unit3(dir, 1e-5f);
  V3 xax = unit3(cross3(upv, zax), 1e-5f);
  const V3 yax = unit3(cross3(zax, xax), 1e-5f);
  // torch.isclose(x, 0, atol=5e-3) with rtol*|0| = 0
  if (fabsf(xax.x) <= 5e-3f && fabsf(xax.y) <= 5e-3f && fabsf(xax.z) <= 5e-3f) xax = unit3(cross3(yax, zax), 1e-5f);
  float* e = extr + 12 * b;   // [R^T | T]: columns x, y, z axes, then the camera location (dist = 1)
  e[0] = xax.x; e[1] = yax.x; e[2] = zax.x;  e[3] = dir.x;
  e[4] = xax.y; e[5] = yax.y; e[6] = zax.y;  e[7] = dir.y;
  e[8] = xax.z; e[9] = yax.z; e[10] = zax.z; e[11] = dir.z;
}

}  // namespace

extern "C" int cips3d_camera_params(const float* locations, const float* fov_deg, float fov_deg_scalar,
                                    const float* up, float dist_radius, int img_size, int B, float* extrinsics,
                                    float* focal, float* near_, float* far_, void* stream) {
  if (B == 0) return 0;                          // empty batch (the tensors may have no storage)
  if (!locations || !extrinsics || !focal || !near_ || !far_ || B < 0 || img_size <= 0) return CIPS3D_E_BADARG;
  hipLaunchKernelGGL(camera_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, as_stream(stream), locations, fov_deg,
                     fov_deg_scalar, up, dist_radius, img_size, B, extrinsics, focal, near_, far_);
  return cips3d_launch_status();
}
