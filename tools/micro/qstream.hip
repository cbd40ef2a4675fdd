// What bounds the cross-attention core's q / ctx stream (BASELINE config 2: 8 x 4096 rows x 1024 channels fp32, 16 heads of 64)?
// A wave owns 16 rows x one head (256 B per row).  Compared, with the real kernel's grid (8 x 16 x 8 workgroups of 4 waves, 64-row
// groups, two groups in flight per wave) and its occupancy (two workgroups per CU, set by the dynamic LDS size):
//   load  0: the kernel's map -- lane (r, g) loads 16 B at 64 c + 16 g of row r, c = 0 .. 3: every instruction touches HALF of 16 lines,
//            the other half by the next instruction (a hit on a pending miss)
//   load  1: LDS-DMA, lane l of instruction i loads chunk (l % 16) of row 4 i + l / 16: an instruction = 4 rows x 256 B = 8 whole lines;
//            the LDS image is swizzled (chunk k of row r at slot (k + 2 (r % 8)) % 16) so that the B-fragment reads are conflict free
//   store 0: lane (r, g) stores 16 B at 64 c + 16 g of row r (the kernel's map); store 1: through LDS, whole rows per instruction;
//   store 2: none
//   hipcc -O3 --offload-arch=gfx950 tools/micro/qstream.hip -o /tmp/qstream && /tmp/qstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

constexpr int C = 1024, D = 64, T = 4096, HEADS = 16, B = 8;

template <int LOAD, int STORE, int QA>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ Q, float* __restrict__ O) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  const int head = blockIdx.y, b = blockIdx.z;
  const int n_groups = T / 64;
  // per wave: QA tiles of 4 KB (q ring) + one 4 KB tile for the stores
  unsigned char* ring = lds + wave * (QA + 1) * 4096;
  const unsigned ring_off = (unsigned)(wave * (QA + 1) * 4096);
  f32x4 qn[QA][4];
  int issued = 0, mark[QA];                      // vector-memory operations issued so far / right behind each slot's requests
  auto tile_base = [&](int grp) { return Q + ((size_t)b * T + (size_t)grp * 64 + wave * 16) * C + (size_t)head * D; };
  auto fetch = [&](int grp, int slot) __attribute__((always_inline)) {
    if (grp >= n_groups) return;
    issued += 4; mark[slot] = issued;
    if (LOAD == 0) {
      const float* qp = tile_base(grp) + (size_t)r * C + 4 * g;
#pragma unroll
      for (int c = 0; c < 4; ++c) qn[slot][c] = *reinterpret_cast<const f32x4*>(qp + 16 * c);
    } else {
      const float* base = tile_base(grp);
      const int rho = lane >> 4, pp = lane & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + rho;
        const int kch = (pp - 2 * (row & 7)) & 15;
        glds16(base, (unsigned)((row * C + kch * 4) * 4), ring_off + slot * 4096 + i * 1024);
      }
    }
  };
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto process = [&](int grp, int slot) __attribute__((always_inline)) {
    f32x4 q[4];
    if (LOAD == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = qn[slot][c];
    } else {
      // everything younger than this tile's requests may stay in flight (results return in order)
      switch (issued - mark[slot]) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
      const unsigned char* tp = ring + slot * 4096 + (r >> 2) * 1024 + (r & 3) * 256;
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = *reinterpret_cast<const f32x4*>(tp + ((4 * c + g + 2 * (r & 7)) & 15) * 16);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { q[c] = q[c] * 1.5f + 0.25f; acc += q[c]; }
    if (LOAD == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    fetch(grp + QA * (int)gridDim.x, slot);
    if (STORE != 2) issued += 4;
    float* op = O + ((size_t)b * T + (size_t)grp * 64 + wave * 16) * C + (size_t)head * D;
    if (STORE == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(op + (size_t)r * C + 4 * g + 16 * c) = q[c];
    } else if (STORE == 1) {
      unsigned char* sp = ring + QA * 4096;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<f32x4*>(sp + (r >> 2) * 1024 + (r & 3) * 256 + ((4 * c + g + 2 * (r & 7)) & 15) * 16) = q[c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int rho = lane >> 4, pp = lane & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + rho;
        const int kch = (pp - 2 * (row & 7)) & 15;
        const f32x4 v = *reinterpret_cast<const f32x4*>(sp + i * 1024 + lane * 16);
        *reinterpret_cast<f32x4*>(op + (size_t)row * C + kch * 4) = v;
      }
    }
  };
#pragma unroll
  for (int a = 0; a < QA; ++a) fetch(blockIdx.x + a * gridDim.x, a);
  for (int grp = blockIdx.x; grp < n_groups; grp += QA * gridDim.x) {
#pragma unroll
    for (int a = 0; a < QA; ++a) {
      const int gq = grp + a * (int)gridDim.x;
      if (gq < n_groups) process(gq, a);
    }
  }
  if (acc.x == 1.2345f) O[tid] = acc.y;
}

template <int LOAD, int STORE, int QA>
static float run(const std::vector<float*>& qs, const std::vector<float*>& os, int gx, int lds_bytes) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<LOAD, STORE, QA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const dim3 grid(gx, HEADS, B);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<LOAD, STORE, QA>), grid, dim3(256), lds_bytes, 0, qs[i % qs.size()], os[i % os.size()]);
  hipEventRecord(e0, 0);
  const int reps = 60;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<LOAD, STORE, QA>), grid, dim3(256), lds_bytes, 0, qs[i % qs.size()], os[i % os.size()]);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps;
}

int main() {
  const size_t bytes = (size_t)B * T * C * 4;
  std::vector<float*> qs(3), os(3);
  for (auto& p : qs) { hipMalloc(&p, bytes); hipMemset(p, 0, bytes); }
  for (auto& p : os) hipMalloc(&p, bytes);
  const double mb = bytes / 1e6;
#define ROW(L, S, QA_, GX, LDS) { const float us = run<L, S, QA_>(qs, os, GX, LDS); \
    printf("load %d store %d  tiles in flight %d  gx %2d  lds %3d KB : %6.1f us  %5.2f TB/s\n", L, S, QA_, GX, LDS / 1024, us, (S == 2 ? 1 : 2) * mb / us); }
  for (int rep = 0; rep < 2; ++rep) {
    ROW(0, 0, 2, 8, 70 * 1024);
    ROW(1, 0, 2, 8, 70 * 1024);
    ROW(0, 1, 2, 8, 70 * 1024);
    ROW(1, 1, 2, 8, 70 * 1024);
    ROW(1, 1, 3, 8, 70 * 1024);
    ROW(1, 1, 2, 8, 50 * 1024);
    ROW(1, 1, 2, 12, 50 * 1024);
    ROW(1, 0, 2, 8, 50 * 1024);
    ROW(1, 0, 3, 8, 70 * 1024);
    ROW(0, 0, 2, 8, 50 * 1024);
    ROW(0, 2, 2, 8, 70 * 1024);
    ROW(1, 2, 2, 8, 70 * 1024);
    ROW(1, 2, 3, 8, 70 * 1024);
    ROW(1, 2, 2, 8, 50 * 1024);
    printf("\n");
  }
  return 0;
}
