// What matrix rate does THIS part sustain?  The roofline of the dense f16x3 convolutions prices against the nominal fp16 peak
// (2.5 PFLOP/s = 256 CUs x 4 SIMDs x 1024 FLOP per cycle x 2.4 GHz); with every CU issuing MFMAs back to back the chip holds 1.5 - 1.75 GHz
// (profiles/r05_notes.md: tools/micro/mfma_chain.hip, stage_ring.hip), so a kernel that never idled its matrix pipes would still
// show ~0.63 - 0.7 of that peak.  dcf_calib_mfma_rate measures the rate on the device it runs on: bare v_mfma_f32_32x32x16_f16 (or
// 16x16x32) loops, operands in registers, one wave per SIMD, one workgroup per CU, non-trivial operand values (the clock a part holds
// depends on the switching activity).  bench.py reports it beside the roofline (`roofline.checks.mfma_sustained`); nothing in the
// forward depends on it.
#include <hip/hip_runtime.h>

#include "common.h"
#include "../../include/decafnet_hip.h"

namespace dcf {
namespace {

typedef _Float16 c_f16x8 __attribute__((ext_vector_type(8)));

template <bool SMALL>
__global__ __launch_bounds__(256, 1) void k_mfma_rate(float* __restrict__ out, int rounds) {
  const int lane = threadIdx.x & 63;
  c_f16x8 a[2], b[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[j][e] = (_Float16)(0.37f * (float)((lane * 7 + j * 13 + e * 3) % 17) - 2.5f);
      b[j][e] = (_Float16)(0.29f * (float)((lane * 3 + j * 7 + e * 5) % 23) - 3.1f);
    }
  f32x16 acc[4];
  f32x4 acs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    acs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int i = 0; i < rounds; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (SMALL) acs[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r & 1], b[j & 1], acs[j], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[r & 1], b[j & 1], acc[j], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[j][e];
    s += acs[j].x + acs[j].y + acs[j].z + acs[j].w;
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace
}  // namespace dcf

extern "C" int dcf_calib_mfma_rate(int32_t shape, int32_t mfmas_per_wave, int32_t* n_cus, float* ns_per_mfma) {
  using namespace dcf;
  DCF_CHECK((shape == 0 || shape == 1) && mfmas_per_wave >= 16 && n_cus && ns_per_mfma, "dcf_calib_mfma_rate: shape 0 / 1, >= 16 MFMAs per wave, non-null outputs");
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  DCF_HIP(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  float* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  const int rounds = mfmas_per_wave / 16, reps = 4;
  auto launch = [&]() {
    if (shape == 0) hipLaunchKernelGGL(k_mfma_rate<false>, dim3(cus), dim3(256), 0, 0, out, rounds);
    else hipLaunchKernelGGL(k_mfma_rate<true>, dim3(cus), dim3(256), 0, 0, out, rounds);
  };
  // (every step's status is kept so that the buffer and the events are released on any exit)
  hipError_t err = hipMalloc((void**)&out, (size_t)cus * 256 * sizeof(float));
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  if (err == hipSuccess) {
    // the clock settles over the first launches: time a few back-to-back ones behind them
    for (int i = 0; i < 6; ++i) launch();
    err = hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    if (err == hipSuccess) err = hipEventRecord(e1, 0);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (err == hipSuccess) err = hipGetLastError();
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (out) (void)hipFree(out);
  DCF_HIP(err);
  *n_cus = cus;
  *ns_per_mfma = ms * 1e6f / (float)reps / (float)(rounds * 16);
  return 0;
}
