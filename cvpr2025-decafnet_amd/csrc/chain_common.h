// Device helpers shared by the chain kernels (dec_chain.hip, enc_chain.hip): the transposed-product formulation of ffn_chain.hip /
// head_chain.hip -- a lane owns a ROW, every 32-channel tile in the MFMA D layout (channels 32 ot + (e & 3) + 8 (e >> 2) + 4 h).
#pragma once
#include "common.h"

namespace dcf {
namespace chain {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float SA = 16.f, SW = 256.f, UNSCALE = 1.f / 4096.f;     // the f16x3 scaling of gemm_bf16s.hip

__device__ __forceinline__ void split2_f16(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{x0 * s, x1 * s}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = __builtin_fmaf(x0, s, -(float)h[0]), r1 = __builtin_fmaf(x1, s, -(float)h[1]);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
// eight consecutive accumulator slots -> the B operand (hi, lo planes) of one K step
__device__ __forceinline__ void split8(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
  unsigned h4[4], l4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2_f16(v[2 * i], v[2 * i + 1], s, h4[i], l4[i]);
  hi = __builtin_bit_cast(f16x8, u32x4{h4[0], h4[1], h4[2], h4[3]});
  lo = __builtin_bit_cast(f16x8, u32x4{l4[0], l4[1], l4[2], l4[3]});
}

// one 1 KiB LDS-DMA piece (ffn_chain.hip): lane l copies the 16 bytes at sbase + voff to LDS byte lds_dst + 16 l
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  // M0 is written and NOT restored: nothing else in these kernels reads it (gfx9+ LDS instructions do not; tools/isa_gate.py and
  // tests/test_abi.py keep every other use of m0 out of the shipped objects).  DCF_GLDS_KEEP_M0 = the save / restore form
  // (two more scalar instructions per request, ~2 % of a chain kernel's stage).
#ifdef DCF_GLDS_KEEP_M0
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}

__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }


// lane i <- lane i - 1 / lane i + 1 of the wave; the lane without a source (0 / 63) takes `edge`
__device__ __forceinline__ float shr1(float v, float edge) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float shl1(float v, float edge) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// LayerNorm statistics of the lane's row: the lane holds 128 of its 256 channels, the other lane half the rest (blocks.py:125-131:
// mean, then the mean of squared deviations)
template <int C>
__device__ __forceinline__ void row_stats(const f32x4 (&v)[32], float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  mean = xor32_sum(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const f32x4 d = v[i] - mean;
    q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
  }
  rstd = 1.0f / sqrtf(xor32_sum(q) * (1.0f / C) + 1e-5f);
}


}  // namespace chain
// in-kernel stamps of the diagnostic builds (tools/dc_stamp.sh, tools/ea_stamp.sh): cycles one wave spends in the segments of a kernel
#if defined(DCF_DC_STAMP) || defined(DCF_EA_STAMP)
__device__ __forceinline__ unsigned long long dc_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(i) do { const unsigned long long t_ = dcf::dc_stamp(); acc_[i] += t_ - last_; last_ = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

}  // namespace dcf
