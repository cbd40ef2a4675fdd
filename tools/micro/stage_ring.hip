// The MFMA stages of the head kernel (head_chain.hip) by themselves: a 2 x 48 KB LDS ring of weight fragments filled by LDS-DMA, one
// barrier per stage, every wave reading the whole stage for its own rows, B operand in registers.  Question: do TWO waves per SIMD with
// 16-row tiles (v_mfma_f32_16x16x32_f16, 256 registers, 8 waves share the requests) run the same stage faster than ONE wave per SIMD
// with 32-row tiles (v_mfma_f32_32x32x16_f16, the shipped form) -- i.e. does the second wave hide the request cost and the waits?
//   mode 0: 4 waves x 32 rows, per stage 8 groups of 6 fragment reads + 9 MFMAs (32 cycles each), 12 requests per wave
//   mode 1: 8 waves x 16 rows, per stage 8 groups of 6 fragment reads + 9 MFMAs (16 cycles each), 6 requests per wave
// Same bytes per stage, same MFMA cycles per SIMD and stage (2 304); a tile = 32 stages (two trunk layers at C = 256), one tile per
// workgroup, 1 024 workgroups.  Timing only (the sums are written so that nothing is optimised away).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stage_ring.hip -o /tmp/stage_ring && /tmp/stage_ring
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PIECES = 48, STAGE = PIECES * 1024, NSTAGE = 32;

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int MODE, bool DMA, bool BAR = true, bool LDSR = true>
__global__ __launch_bounds__(MODE == 1 ? 512 : 256, 1) void k(const unsigned short* __restrict__ img, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int NWAVE = MODE == 1 ? 8 : 4, NPW = PIECES / NWAVE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  f16x8 xh[8], xl[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[kk][e] = (_Float16)(0.37f * (float)((lane * 7 + kk * 13 + e * 3) % 17) - 2.5f);
      xl[kk][e] = (_Float16)(0.0004f * (float)((lane * 5 + kk * 11 + e) % 19));
    }
  auto issue_piece = [&](int sl, int parity, int i) __attribute__((always_inline)) {
    const int pc = w + NWAVE * i;
    if (DMA) glds16(img + ((size_t)sl * PIECES + pc) * 512, lane16, (unsigned)parity * STAGE + (unsigned)pc * 1024u);
  };
#pragma unroll
  for (int i = 0; i < NPW; ++i) issue_piece(0, 0, i);
  f16x8 yh[8], yl[8];                                  // mode 2: the second 16-row tile's planes
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      yh[kk][e] = (_Float16)(0.29f * (float)((lane * 3 + kk * 7 + e * 5) % 23) - 3.1f);
      yl[kk][e] = (_Float16)(0.0003f * (float)((lane * 11 + kk * 5 + e) % 13));
    }
  f32x16 Z[3];
  f32x4 Zs[2][3], Zt[2][3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) Z[t][e] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) { Zs[a][t] = f32x4{0.f, 0.f, 0.f, 0.f}; Zt[a][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  for (int sp = 0; sp < NSTAGE / 2; ++sp) {
#pragma unroll
    for (int HF = 0; HF < 2; ++HF) {
      const int s = 2 * sp + HF;
      if (BAR) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      unsigned ab = lane16 + HF * STAGE;
      asm volatile("" : "+v"(ab));
      const unsigned char* buf = lds + ab;
      const int nsl = (s + 1) % NSTAGE;
      const bool has_next = s + 1 < NSTAGE;
      f16x8 fa[2][3][2];
      // group gq: the six fragments of (tap, plane) for K step gq (mode 0) / for (16-channel tile gq >> 2, K step gq & 3) (mode 1)
      auto frags = [&](int gq, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
          if (LDSR) {
            fa[set][tap][0] = *reinterpret_cast<const f16x8*>(buf + (2 * (tap * 8 + gq)) * 1024);
            fa[set][tap][1] = *reinterpret_cast<const f16x8*>(buf + (2 * (tap * 8 + gq) + 1) * 1024);
          } else {
            fa[set][tap][0] = xh[(gq + tap) & 7]; fa[set][tap][1] = xl[(gq + 2 * tap) & 7];
          }
        }
      };
      frags(0, 0);
      constexpr int PPG = (NPW + 6) / 7;               // requests per group (none in the last one)
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) {
        const int set = gq & 1, kk = HF * 4 + (MODE == 0 ? gq >> 1 : gq & 3);
        if (gq + 1 < 8) frags(gq + 1, set ^ 1);
        auto piece = [&](int j) __attribute__((always_inline)) {
          const int i = gq * PPG + j;
          if (j < PPG && gq + 1 < 8 && i < NPW && has_next) issue_piece(nsl, HF ^ 1, i);
        };
        piece(0);
        if constexpr (MODE == 0) {
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) Z[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][tap][1], xh[kk], Z[tap], 0, 0, 0);
          Z[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][0][0], xl[kk], Z[0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          piece(1);
          Z[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][1][0], xl[kk], Z[1], 0, 0, 0);
          Z[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][2][0], xl[kk], Z[2], 0, 0, 0);
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) Z[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][tap][0], xh[kk], Z[tap], 0, 0, 0);
        } else if constexpr (MODE == 2) {
          const int a = gq >> 2;
#define M16(acc, A, B) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc, 0, 0, 0)
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) { M16(Zs[a][tap], fa[set][tap][1], xh[kk]); M16(Zt[a][tap], fa[set][tap][1], yh[kk]); }
          M16(Zs[a][0], fa[set][0][0], xl[kk]); M16(Zt[a][0], fa[set][0][0], yl[kk]);
          __builtin_amdgcn_sched_barrier(0);
          piece(1);
          M16(Zs[a][1], fa[set][1][0], xl[kk]); M16(Zt[a][1], fa[set][1][0], yl[kk]);
          M16(Zs[a][2], fa[set][2][0], xl[kk]); M16(Zt[a][2], fa[set][2][0], yl[kk]);
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) { M16(Zs[a][tap], fa[set][tap][0], xh[kk]); M16(Zt[a][tap], fa[set][tap][0], yh[kk]); }
#undef M16
        } else {
          const int a = gq >> 2;
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) Zs[a][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[set][tap][1], xh[kk], Zs[a][tap], 0, 0, 0);
          Zs[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[set][0][0], xl[kk], Zs[a][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          piece(1);
          Zs[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[set][1][0], xl[kk], Zs[a][1], 0, 0, 0);
          Zs[a][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[set][2][0], xl[kk], Zs[a][2], 0, 0, 0);
#pragma unroll
          for (int tap = 0; tap < 3; ++tap) Zs[a][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[set][tap][0], xh[kk], Zs[a][tap], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += Z[t][e];
#pragma unroll
    for (int a = 0; a < 2; ++a) sum += Zs[a][t].x + Zs[a][t].y + Zs[a][t].z + Zs[a][t].w + Zt[a][t].x + Zt[a][t].y + Zt[a][t].z + Zt[a][t].w;
  }
  out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
}

template <int MODE, bool DMA, bool BAR = true, bool LDSR = true>
static void run(const char* name, const unsigned short* img, float* out, int grid) {
  const int lds = 2 * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, DMA, BAR, LDSR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int threads = MODE == 1 ? 512 : 256;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, DMA, BAR, LDSR>), dim3(grid), dim3(threads), lds, 0, img, out);
  hipEventRecord(e0, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, DMA, BAR, LDSR>), dim3(grid), dim3(threads), lds, 0, img, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  const double us = ms * 1e3 / reps, rounds = grid / 256.0;
  // MFMA work: per SIMD and stage 72 x 32 = 2 304 cycles in both modes
  printf("%-52s %7.1f us per launch, %6.2f us per tile, %5.1f ns per MFMA slot (32 cycles of one SIMD)\n", name, us, us / rounds, 1e3 * us / rounds / (NSTAGE * 72.0));
}

int main() {
  unsigned short* img; float* out;
  const size_t halfs = (size_t)NSTAGE * PIECES * 512;
  hipMalloc(&img, halfs * 2); hipMalloc(&out, 1024 * 512 * 4);
  unsigned short* h = (unsigned short*)malloc(halfs * 2);
  for (size_t i = 0; i < halfs; ++i) h[i] = (unsigned short)(0x2c00 + (i * 2654435761u >> 22) % 0x0c00);   // fp16 values in [0.06, 2)
  hipMemcpy(img, h, halfs * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, true>("4 waves x 32 rows (32x32x16), weight stream", img, out, 1024);
    run<1, true>("8 waves x 16 rows (16x16x32), weight stream", img, out, 1024);
    run<2, true>("4 waves x 2 x 16 rows (16x16x32), weight stream", img, out, 1024);
    run<2, false>("4 waves x 2 x 16 rows (16x16x32), NO weight stream", img, out, 1024);
    run<2, false, false, false>("4 waves x 2 x 16 rows, MFMAs only", img, out, 1024);
    run<0, false>("4 waves x 32 rows, NO weight stream (timing only)", img, out, 1024);
    run<1, false>("8 waves x 16 rows, NO weight stream (timing only)", img, out, 1024);
    run<0, true, false>("4 waves x 32 rows, stream, NO barrier / wait", img, out, 1024);
    run<0, false, false>("4 waves x 32 rows, NO stream, NO barrier", img, out, 1024);
    run<0, false, true, false>("4 waves x 32 rows, NO stream, NO fragment reads", img, out, 1024);
    run<0, false, false, false>("4 waves x 32 rows, MFMAs only", img, out, 1024);
    run<1, false, false, false>("8 waves x 16 rows, MFMAs only", img, out, 1024);
    run<0, true, true, false>("4 waves x 32 rows, stream, NO fragment reads", img, out, 1024);
    printf("\n");
  }
  return 0;
}
