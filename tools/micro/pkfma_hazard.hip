// Standalone reproducer for the miscompute that round 3 sidestepped with -fno-slp-vectorize (profiles/r04_pkfma_hazard.md):
// a packed fp32 FMA whose LOW lane takes the HIGH dword of a register pair (op_sel) that an LDS read has just written.
//
//   hipcc -O2 --offload-arch=gfx950 tools/micro/pkfma_hazard.hip -o /tmp/pkfma_hazard && /tmp/pkfma_hazard
//
// Every workgroup keeps a table of (mean, rstd) pairs in LDS.  A wave reads one pair per lane with ds_read2_b32 (what the
// compiler emitted for `mean = wg[2 i], rstd = wg[2 i + 1]` in gemm_common.h), waits with s_waitcnt lgkmcnt(0) and feeds the
// pair to the instruction sequence the SLP vectoriser produced -- the exact four v_pk_fma_f32 of the failing epilogue -- or to one
// of the variants below; the result is compared with scalar v_fma_f32 on registers read again later.  Around it the waves keep
// the LDS busy with the transpose traffic of the real epilogue (16 ds_write_b32 + 4 ds_read_b128 per step).
//   variant 0: ds_read2_b32 -> s_waitcnt lgkmcnt(0) -> 4 x v_pk_fma_f32, low lane reads the HIGH dword (op_sel:[0,1,0])   <- the bug
//   variant 1: the same with s_nop 7 x 2 behind the wait
//   variant 2: the pair copied to other registers by v_mov_b32 first; the packed ops read the LOW dword in both lanes (op_sel_hi)
//   variant 3: scalar v_fma_f32 on the pair (what -fno-slp-vectorize emits)
//   variant 4: as 0 with ds_read_b64 instead of ds_read2_b32
//   variant 5: as 0 with one independent VALU instruction (v_mov of an unrelated register) behind the wait
//   variant 6 .. 9: as 0 with s_nop 0 / 1 / 3 / 7 (1 / 2 / 4 / 8 wait states) behind the wait
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ int g_mfma_partner = 0;        // 1: the waves of the odd workgroups run back-to-back MFMAs on the same SIMDs meanwhile (what the
                                          // other resident waves of the GEMM do while one is in its epilogue)

template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, unsigned* __restrict__ bad, unsigned* __restrict__ bad_by_lane, int iters) {
  __shared__ float tile[4][32 * 36];
  __shared__ float stats[2 * 128 + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (g_mfma_partner && (blockIdx.x & 1)) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (float)(lane + e)); b[e] = (_Float16)(0.02f * (float)(e + 1)); }
    f32x16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    for (int it = 0; it < iters * 24; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[1] == 12345.678f) bad[1] = 2;
    return;
  }
  const int r = lane & 31, h = lane >> 5, rr = lane >> 3, c4 = (lane & 7) * 4;
  unsigned nbad = 0;
  float acc16[16];
  for (int e = 0; e < 16; ++e) acc16[e] = in[(blockIdx.x * 256 + tid) * 16 + e];
  for (int it = 0; it < iters; ++it) {
    // fresh statistics every iteration: the registers that take the pair hold something else before the load
    __syncthreads();
    if (tid < 128) { stats[2 * tid] = 0.25f + 0.001f * (float)((tid + it) & 63); stats[2 * tid + 1] = 1.5f + 0.01f * (float)((tid * 7 + it) & 31); }
    __syncthreads();
    for (int q = 0; q < 4; ++q) {
      // the epilogue's transpose: D layout -> LDS -> 4 consecutive columns of one row per lane
      for (int e = 0; e < 16; ++e) tile[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + r] = acc16[e] + (float)q;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[wave][(rr + 8 * q) * 36 + c4]);
      const f32x4 l = {0.5f, -0.25f, 0.75f, 1.25f}, bb = {0.1f, 0.2f, 0.3f, 0.4f};
      const unsigned addr = (unsigned)(size_t)(&stats[2 * (wave * 32 + rr + 8 * q)]);      // LDS byte address of the lane's pair
      f32x2 xy = {v.x, v.y}, zw = {v.z, v.w}, lxy = {l.x, l.y}, lzw = {l.z, l.w}, bxy = {bb.x, bb.y}, bzw = {bb.z, bb.w};
      f32x2 pr;                                                                                // (mean, rstd)
      f32x2 oxy, ozw;
#define PK_SEQ(LOAD, EXTRA)                                                                                         \
  asm volatile(LOAD "\n\t"                                                                                          \
               "s_waitcnt lgkmcnt(0)\n\t" EXTRA                                                                       \
               "v_pk_fma_f32 %1, %0, %4, %1 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"                     \
               "v_pk_fma_f32 %2, %0, %5, %2 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"                     \
               "v_pk_fma_f32 %2, %2, %0, %7 op_sel:[0,1,0]\n\t"                                                      \
               "v_pk_fma_f32 %1, %1, %0, %6 op_sel:[0,1,0]\n\t"                                                      \
               : "=&v"(pr), "+v"(xy), "+v"(zw), "+v"(junk)                                                           \
               : "v"(lxy), "v"(lzw), "v"(bxy), "v"(bzw), "v"(addr)                                                   \
               : "memory")
      float junk = (float)lane;
      if constexpr (V == 0) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", ""); oxy = xy; ozw = zw; }
      else if constexpr (V == 1) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "s_nop 7\n\ts_nop 7\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 4) { PK_SEQ("ds_read_b64 %0, %8", ""); oxy = xy; ozw = zw; }
      else if constexpr (V == 5) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "v_mov_b32 %3, %3\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 6) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "s_nop 0\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 7) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "s_nop 1\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 8) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "s_nop 3\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 9) { PK_SEQ("ds_read2_b32 %0, %8 offset1:1", "s_nop 7\n\t"); oxy = xy; ozw = zw; }
      else if constexpr (V == 10 || V == 11) {
        // the failing epilogue's registers verbatim: the pair lands in v[18:19], v18 is ALSO the address register of the load, the
        // last packed op writes the pair it reads through op_sel.  V == 11: one independent ds_read_b128 is still in flight when
        // the pair is requested (as the transpose read of the real epilogue is)
        float o0, o1, o2, o3;
        f32x4 extra;
        const unsigned taddr = (unsigned)(size_t)(&tile[wave][(rr + 8 * q) * 36 + c4]);
        if constexpr (V == 10) {
        asm volatile(
            "v_mov_b32 v18, %5\n\t"
            "v_mov_b32 v62, %6\n\tv_mov_b32 v63, %7\n\tv_mov_b32 v64, %8\n\tv_mov_b32 v65, %9\n\t"
            "v_mov_b32 v30, 0.5\n\tv_mov_b32 v31, 0xbe800000\n\tv_mov_b32 v32, 0x3f400000\n\tv_mov_b32 v33, 0x3fa00000\n\t"
            "v_mov_b32 v22, 0x3dcccccd\n\tv_mov_b32 v23, 0x3e4ccccd\n\tv_mov_b32 v24, 0x3e99999a\n\tv_mov_b32 v25, 0x3ecccccd\n\t"
            "v_mov_b32 v19, 0x4b30\n\t"
            "s_nop 4\n\t"
            "ds_read_b128 %4, %10\n\ts_waitcnt lgkmcnt(0)\n\t"
            "ds_read2_b32 v[18:19], v18 offset1:1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_fma_f32 v[62:63], v[18:19], v[30:31], v[62:63] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 v[20:21], v[18:19], v[32:33], v[64:65] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 v[20:21], v[20:21], v[18:19], v[24:25] op_sel:[0,1,0]\n\t"
            "v_pk_fma_f32 v[18:19], v[62:63], v[18:19], v[22:23] op_sel:[0,1,0]\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %0, v18\n\tv_mov_b32 %1, v19\n\tv_mov_b32 %2, v20\n\tv_mov_b32 %3, v21\n\t"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(extra)
            : "v"(addr), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(taddr)
            : "memory", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v30", "v31", "v32", "v33", "v62", "v63", "v64", "v65");
        } else {
        asm volatile(
            "v_mov_b32 v18, %5\n\t"
            "v_mov_b32 v62, %6\n\tv_mov_b32 v63, %7\n\tv_mov_b32 v64, %8\n\tv_mov_b32 v65, %9\n\t"
            "v_mov_b32 v30, 0.5\n\tv_mov_b32 v31, 0xbe800000\n\tv_mov_b32 v32, 0x3f400000\n\tv_mov_b32 v33, 0x3fa00000\n\t"
            "v_mov_b32 v22, 0x3dcccccd\n\tv_mov_b32 v23, 0x3e4ccccd\n\tv_mov_b32 v24, 0x3e99999a\n\tv_mov_b32 v25, 0x3ecccccd\n\t"
            "v_mov_b32 v19, 0x4b30\n\t"
            "s_nop 4\n\t"
            "ds_read_b128 %4, %10\n\t"
            "ds_read2_b32 v[18:19], v18 offset1:1\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_fma_f32 v[62:63], v[18:19], v[30:31], v[62:63] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 v[20:21], v[18:19], v[32:33], v[64:65] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
            "v_pk_fma_f32 v[20:21], v[20:21], v[18:19], v[24:25] op_sel:[0,1,0]\n\t"
            "v_pk_fma_f32 v[18:19], v[62:63], v[18:19], v[22:23] op_sel:[0,1,0]\n\t"
            "s_nop 4\n\t"
            "v_mov_b32 %0, v18\n\tv_mov_b32 %1, v19\n\tv_mov_b32 %2, v20\n\tv_mov_b32 %3, v21\n\t"
            : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(extra)
            : "v"(addr), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(taddr)
            : "memory", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v30", "v31", "v32", "v33", "v62", "v63", "v64", "v65");
        }
        oxy = f32x2{o0, o1}; ozw = f32x2{o2, o3};
        if (extra.x == 123.456f) nbad += 1000000;
      }
      else if constexpr (V == 2) {
        float m2, rs;
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pr) : "v"(addr) : "memory");
        asm volatile("v_mov_b32 %0, %1" : "=v"(m2) : "v"(pr.x));
        asm volatile("v_mov_b32 %0, %1" : "=v"(rs) : "v"(pr.y));
        oxy = f32x2{__builtin_fmaf(__builtin_fmaf(-m2, lxy.x, xy.x), rs, bxy.x), __builtin_fmaf(__builtin_fmaf(-m2, lxy.y, xy.y), rs, bxy.y)};
        ozw = f32x2{__builtin_fmaf(__builtin_fmaf(-m2, lzw.x, zw.x), rs, bzw.x), __builtin_fmaf(__builtin_fmaf(-m2, lzw.y, zw.y), rs, bzw.y)};
      } else if constexpr (V == 3) {
        asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pr) : "v"(addr) : "memory");
        float mean = pr.x, rs = pr.y;
        asm volatile("" : "+v"(mean), "+v"(rs));
        oxy = f32x2{__builtin_fmaf(__builtin_fmaf(-mean, lxy.x, xy.x), rs, bxy.x), __builtin_fmaf(__builtin_fmaf(-mean, lxy.y, xy.y), rs, bxy.y)};
        ozw = f32x2{__builtin_fmaf(__builtin_fmaf(-mean, lzw.x, zw.x), rs, bzw.x), __builtin_fmaf(__builtin_fmaf(-mean, lzw.y, zw.y), rs, bzw.y)};
      }
      // reference: scalar arithmetic on the table's values read the ordinary way, long after the loads above
      const float mean = stats[2 * (wave * 32 + rr + 8 * q)], rs = stats[2 * (wave * 32 + rr + 8 * q) + 1];
      const float e0 = __builtin_fmaf(__builtin_fmaf(-mean, l.x, v.x), rs, bb.x), e1 = __builtin_fmaf(__builtin_fmaf(-mean, l.y, v.y), rs, bb.y);
      const float e2 = __builtin_fmaf(__builtin_fmaf(-mean, l.z, v.z), rs, bb.z), e3 = __builtin_fmaf(__builtin_fmaf(-mean, l.w, v.w), rs, bb.w);
      const unsigned m = (oxy.x != e0 ? 1u : 0u) | (oxy.y != e1 ? 2u : 0u) | (ozw.x != e2 ? 4u : 0u) | (ozw.y != e3 ? 8u : 0u);
      if (m) { ++nbad; atomicAdd(&bad_by_lane[lane * 4 + (m & 1 ? 0 : (m & 4 ? 2 : (m & 2 ? 1 : 3)))], 1u); }
      acc16[q] += oxy.x * 1e-6f;                             // keep the chain alive
    }
  }
  if (nbad) atomicAdd(bad, nbad);
  if (acc16[0] == 12345.678f) bad[1] = 1;
}

template <int V>
static void run(const char* what, const float* in, unsigned* bad, unsigned* by_lane, int blocks, int iters) {
  CHECK(hipMemset(bad, 0, 8));
  CHECK(hipMemset(by_lane, 0, 256 * 4));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, in, bad, by_lane, iters);
  CHECK(hipDeviceSynchronize());
  unsigned hb[2];
  std::vector<unsigned> hl(256);
  CHECK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hl.data(), by_lane, 1024, hipMemcpyDeviceToHost));
  const double total = (double)blocks * 256 * iters * 4;
  unsigned q[4] = {0, 0, 0, 0}, comp[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l)
    for (int c = 0; c < 4; ++c) { q[l / 16] += hl[l * 4 + c]; comp[c] += hl[l * 4 + c]; }
  printf("variant %d  %-58s wrong %9u of %.3g   lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u   first wrong component x/y/z/w: %u %u %u %u\n", V, what,
         hb[0], total, q[0], q[1], q[2], q[3], comp[0], comp[1], comp[2], comp[3]);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 2048, iters = argc > 2 ? atoi(argv[2]) : 200;
  float* in;
  unsigned *bad, *by_lane;
  std::vector<float> h((size_t)blocks * 256 * 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f - 0.5f;
  CHECK(hipMalloc(&in, h.size() * 4));
  CHECK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&bad, 8));
  CHECK(hipMalloc(&by_lane, 1024));
  for (int rep = 0; rep < 2; ++rep) {
    const int partner = rep;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_partner), &partner, sizeof(int)));
    printf("---- MFMA partner waves on the same SIMDs: %s\n", partner ? "yes" : "no");
    run<0>("ds_read2_b32, wait, v_pk_fma_f32 op_sel (low lane <- high dword)", in, bad, by_lane, blocks, iters);
    run<6>("... + s_nop 0 behind the wait", in, bad, by_lane, blocks, iters);
    run<7>("... + s_nop 1", in, bad, by_lane, blocks, iters);
    run<8>("... + s_nop 3", in, bad, by_lane, blocks, iters);
    run<9>("... + s_nop 7", in, bad, by_lane, blocks, iters);
    run<1>("... + s_nop 7, s_nop 7", in, bad, by_lane, blocks, iters);
    run<5>("... + one unrelated v_mov_b32 behind the wait", in, bad, by_lane, blocks, iters);
    run<4>("ds_read_b64 instead of ds_read2_b32", in, bad, by_lane, blocks, iters);
    run<2>("pair copied by v_mov_b32, scalar v_fma_f32", in, bad, by_lane, blocks, iters);
    run<3>("scalar v_fma_f32 on the loaded pair", in, bad, by_lane, blocks, iters);
    run<10>("the failing epilogue's registers verbatim (v[18:19], v18 = address)", in, bad, by_lane, blocks, iters);
    run<11>("... with a ds_read_b128 still in flight at the pair's load", in, bad, by_lane, blocks, iters);
  }
  return 0;
}
