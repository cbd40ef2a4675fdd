// How long does ONE workgroup need for the input rows of a chain-kernel tile (128 rows x C fp32) when the memory system is NOT
// saturated -- the situation of a chain kernel's prologue, whose workgroups are out of step with each other?
//   mode 0: the kernels' map: lane (r, h) of wave w loads 16 bytes at channel 16 kk + 4 h (+ 8) of row 32 w + r: every instruction
//           touches 32 rows, every 128-byte line is touched by four instructions
//   mode 1: LDS-DMA, a request = 8 rows x 128 bytes (whole lines, each touched once), all requests of the tile in flight at once, then
//           the lane = row fragment reads from LDS (swizzled image)
// G workgroups (one per CU at most), each walks `tiles` tiles one after the other (addresses far apart: cold).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/tileload.hip -o /tmp/tileload && /tmp/tileload
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int C, int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* __restrict__ X, float* __restrict__ out, int tiles, int tile_stride_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int KS = C / 16, SL = C / 32;            // K steps; 128-byte slabs per row
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; ++t) {
    const float* base = X + ((size_t)blockIdx.x * tiles + t) * (size_t)tile_stride_rows * C;
    f32x4 v[2 * KS];
    if (MODE == 0) {
      const float* px = base + (size_t)(w * 32 + r) * C + 4 * h;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) { v[2 * kk] = *reinterpret_cast<const f32x4*>(px + 16 * kk); v[2 * kk + 1] = *reinterpret_cast<const f32x4*>(px + 16 * kk + 8); }
    } else {
      // image: slab s (128 bytes of every row), row q: 8 chunks of 16 bytes at ((s * 128 + q) * 8 + ((c + q) & 7)) * 16
      // request (s, g): rows 8 g .. 8 g + 7 of slab s: lane l -> row 8 g + (l >> 3), chunk l & 7
      const int total = SL * 16;                       // requests per tile (1 KB each)
      for (int i = w; i < total; i += 4) {
        const int s = i >> 4, g = i & 15;
        const int q = 8 * g + (lane >> 3), slot = lane & 7, c = (slot - q) & 7;
        glds16(base, (unsigned)((q * C + 32 * s + 4 * c) * 4), (unsigned)(i * 1024));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int q = w * 32 + r;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int s = kk >> 1, c0 = 4 * (kk & 1) + h, c1 = c0 + 2;
        v[2 * kk] = *reinterpret_cast<const f32x4*>(lds + ((s * 128 + q) * 8 + ((c0 + q) & 7)) * 16);
        v[2 * kk + 1] = *reinterpret_cast<const f32x4*>(lds + ((s * 128 + q) * 8 + ((c1 + q) & 7)) * 16);
      }
      __syncthreads();                                  // (the image is free again)
    }
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) acc += v[i];
  }
  if (acc.x == 1.2345f) out[tid] = acc.y;
}

template <int C, int MODE>
static float run(const float* X, float* out, int G, int tiles) {
  const int lds = MODE == 0 ? 1024 : 128 * C * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<C, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<C, MODE>), dim3(G), dim3(256), lds, 0, X, out, tiles, 128);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<C, MODE>), dim3(G), dim3(256), lds, 0, X + (size_t)(i + 1) * 4099 * C, out, tiles, 128);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / reps / tiles;
}

int main() {
  const size_t rows = (size_t)256 * 64 * 128 + 65536;
  float* X; float* out;
  hipMalloc(&X, rows * 288 * 4); hipMemset(X, 0, rows * 288 * 4);
  hipMalloc(&out, 4096);
  for (int rep = 0; rep < 2; ++rep) {
    for (int G : {8, 64, 256}) {
      const int tiles = 32;
      printf("C = 288, %3d workgroups x %d tiles: lane = row loads %6.2f us per tile, LDS-DMA whole lines + fragment reads %6.2f us per tile\n", G, tiles,
             run<288, 0>(X, out, G, tiles), run<288, 1>(X, out, G, tiles));
      printf("C = 256, %3d workgroups x %d tiles: lane = row loads %6.2f us per tile, LDS-DMA whole lines + fragment reads %6.2f us per tile\n", G, tiles,
             run<256, 0>(X, out, G, tiles), run<256, 1>(X, out, G, tiles));
    }
  }
  return 0;
}
