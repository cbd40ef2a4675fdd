// What does a chain kernel's row phase cost?  128 rows x 1 KB per workgroup (4 waves, one workgroup per CU as in the chain kernels),
// read into registers and written back, with the lane -> address maps of (a) the kernels (lane = row: a 16-byte piece of 32 different
// rows per wave instruction, 32 cache lines touched) and (b) a tiled layout (a wave instruction = 1 KB contiguous, 8 lines).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rowload.hip -o /tmp/rowload && /tmp/rowload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE, bool STORE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ X, float* __restrict__ Y, int rows) {
  extern __shared__ unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5, w = tid >> 6;
  const size_t tile = (size_t)blockIdx.x * 4 + w;                 // 32 rows
  f32x4 v[32];
  if (MODE == 0) {
    const float* px = X + (tile * 32 + r) * 256 + 4 * h;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const f32x4*>(px + 32 * (i >> 2) + 8 * (i & 3));
  } else {
    const float* px = X + tile * 32 * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = *reinterpret_cast<const f32x4*>(px + i * 256);
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
  if (s.x == 12345.678f) lds[tid] = 1;                             // keep the LDS allocation
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = v[i] * 1.5f + s;
  if (STORE) {
    if (MODE == 0) {
      float* py = Y + (tile * 32 + r) * 256 + 4 * h;
#pragma unroll
      for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4*>(py + 32 * (i >> 2) + 8 * (i & 3)) = v[i];
    } else {
      float* py = Y + tile * 32 * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4*>(py + i * 256) = v[i];
    }
  } else {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) t += v[i];
    if (t.x == 1.2345f) Y[tid] = t.y;
  }
}
template <int MODE, bool STORE>
static float run(const std::vector<float*>& xs, const std::vector<float*>& ys, int rows) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((k<MODE, STORE>), dim3(rows / 128), dim3(256), 131072, 0, xs[i % xs.size()], ys[i % ys.size()], rows);
  hipEventRecord(e0, 0);
  const int reps = 40;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, STORE>), dim3(rows / 128), dim3(256), 131072, 0, xs[i % xs.size()], ys[i % ys.size()], rows);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
int main() {
  const int rows = 131072;
  std::vector<float*> xs(4), ys(4);
  for (auto& p : xs) { hipMalloc(&p, (size_t)rows * 1024); hipMemset(p, 0, (size_t)rows * 1024); }
  for (auto& p : ys) hipMalloc(&p, (size_t)rows * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    printf("lane = row  : read %.1f us, read + write %.1f us   (134 MB each way)\n", run<0, false>(xs, ys, rows), run<0, true>(xs, ys, rows));
    printf("tiled       : read %.1f us, read + write %.1f us\n", run<1, false>(xs, ys, rows), run<1, true>(xs, ys, rows));
  }
  return 0;
}
