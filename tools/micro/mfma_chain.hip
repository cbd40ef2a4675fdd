// v_mfma_f32_32x32x16_f16 on gfx950: cycles per MFMA for dependent chains against independent accumulators, one and two waves per SIMD
// (hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain).  Both s_memtime ticks and wall time are printed: the ratio is the
// tick rate of s_memtime, which is what the in-kernel stamps of the chain kernels count in.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NACC accumulators in rotation, n rounds of 4 * NACC MFMAs; WAVES waves per SIMD (blockDim = 256 * WAVES, one workgroup per CU)
template <int NACC, bool SMALL>
__global__ __launch_bounds__(512) void k(float* out, int n, unsigned long long* t) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
  f32x16 acc[NACC];
  f32x4 acs[NACC];
  for (int j = 0; j < NACC; ++j) { for (int e = 0; e < 16; ++e) acc[j][e] = 0.f; for (int e = 0; e < 4; ++e) acs[j][e] = 0.f; }
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        if constexpr (SMALL) acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acs[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) { for (int e = 0; e < 16; ++e) s += acc[j][e]; for (int e = 0; e < 4; ++e) s += acs[j][e]; }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *t = t1 - t0;
}

template <int NACC, bool SMALL>
void run(const char* name, int waves, int blocks, float* out, unsigned long long* t) {
  const int n = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned long long h = 0; float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(256 * waves), 0, 0, out, n, t);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  }
  const double per_wave = 4.0 * NACC * n;             // MFMAs per wave
  printf("%-44s waves/SIMD %d, %3d CUs: %6.1f ticks per MFMA and wave = %5.1f per MFMA and SIMD; %.1f ns per MFMA and SIMD (ticks / ns = %.2f)\n", name, waves, blocks,
         (double)h / per_wave, (double)h / per_wave / waves, 1e6 * ms / per_wave / waves, (double)h / (1e6 * ms));
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&t, 8);
  for (int blocks : {1, 256}) {
    run<1, false>("32x32x16 f16, ONE accumulator (dependent)", 1, blocks, out, t);
    run<2, false>("32x32x16 f16, two accumulators alternating", 1, blocks, out, t);
    run<4, false>("32x32x16 f16, four accumulators", 1, blocks, out, t);
    run<1, false>("32x32x16 f16, ONE accumulator (dependent)", 2, blocks, out, t);
    run<2, false>("32x32x16 f16, two accumulators alternating", 2, blocks, out, t);
    run<1, true>("16x16x32 f16, ONE accumulator (dependent)", 1, blocks, out, t);
    run<2, true>("16x16x32 f16, two accumulators alternating", 1, blocks, out, t);
    run<4, true>("16x16x32 f16, four accumulators", 1, blocks, out, t);
    run<1, true>("16x16x32 f16, ONE accumulator (dependent)", 2, blocks, out, t);
  }
  return 0;
}
