// issue rate of DPP lane shifts on gfx950: wave_shr:1 against row_shr:1 (hipcc --offload-arch=gfx950 dpp_rate.hip -o dpp_rate)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void k(float* p, int n, unsigned long long* t) {
  float v0 = p[threadIdx.x], v1 = v0 * 2.f, v2 = v0 * 3.f, v3 = v0 * 5.f, a = 0.f;
  unsigned long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    v0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v0), CTRL, 0xf, 0xf, true)) + 1.f;
    v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v1), CTRL, 0xf, 0xf, true)) + 1.f;
    v2 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v2), CTRL, 0xf, 0xf, true)) + 1.f;
    v3 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v3), CTRL, 0xf, 0xf, true)) + 1.f;
  }
  unsigned long long t1 = clock64();
  a = v0 + v1 + v2 + v3;
  p[threadIdx.x] = a;
  if (threadIdx.x == 0) *t = t1 - t0;
}
int main() {
  float* p; unsigned long long* t; unsigned long long h;
  hipMalloc(&p, 256); hipMalloc(&t, 8); hipMemset(p, 0, 256);
  const int n = 4096;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0x138>, dim3(1), dim3(64), 0, 0, p, n, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("wave_shr:1  %.2f cycles per (dpp mov + add)\n", (double)h / (4.0 * n));
    hipLaunchKernelGGL(k<0x111>, dim3(1), dim3(64), 0, 0, p, n, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("row_shr:1   %.2f cycles per (dpp mov + add)\n", (double)h / (4.0 * n));
    hipLaunchKernelGGL(k<0x142>, dim3(1), dim3(64), 0, 0, p, n, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("row_bcast15 %.2f cycles per (dpp mov + add)\n", (double)h / (4.0 * n));
  }
  return 0;
}
