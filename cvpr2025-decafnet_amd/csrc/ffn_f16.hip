// Fused FFN (blocks.py:535-538 inside TransformerEncoder / TransformerDecoder): Y = epilogue(GELU(X W1^T + b1) W2^T + b2)
// in ONE kernel, f16x3 arithmetic (see gemm_bf16s.hip), so that the 4E-wide hidden activations never leave the CU.
//
// Why: with several videos per forward the two FFN GEMMs are bound by the traffic of their fp32 activations -- fc reads
// M x E and WRITES M x 4E floats, proj reads them back (251 MB each way at M = 49152, E = 256; ~3 TB/s sustained, 85-91 us
// per GEMM whatever the tile).  Here a workgroup owns 64 rows: the X tile is split once into two fp16 planes in LDS, the
// hidden dimension is walked in 4 chunks of E columns -- GEMM 1 of a chunk (K = E) -> bias, GELU, split -> LDS as the A
// operand of GEMM 2 of the chunk (K = E of the 4E) -- and the E-wide output accumulators stay in registers across the
// chunks.  HBM traffic: M x E in, M x E out (+ the residual); both weight images (2 x 2 MiB at E = 256 in two planes)
// stream from L2 once per 64-row tile: 43 B/clk/CU at full MFMA rate.
//
// Four waves side by side along the columns, wave tile 64 x (TN * 32), E = TN * 128.  LDS: X planes and H planes
// [2][64][E * 2 + 16 B] (row pitch = 4 dwords mod 64: conflict-free ds_read_b128) + a wave-private transpose tile each:
// 150 KiB at E = 256, one workgroup per CU.
//
// STATUS: correct (tests/test_gpu_ops.py::test_fused_ffn, and end to end with DCF_FFN_FUSE_MIN_ROWS=64) but NOT the
// default: measured at M = 49152, E = 256 the kernel takes 300 us (172 TFLOP/s) against 175 us for the fc + proj pair --
// with one wave per SIMD (150 KiB of LDS) the GELU epilogue (20 k cycles per tile), the A-fragment LDS reads (16 k) and
// the weight stream (32 k) serialise with the 49 k MFMA cycles instead of overlapping them; the 32-row variant (two
// workgroups per CU) doubles the weight bytes per row and is slower still (450 us).  Enabled by
// DCF_FFN_FUSE_MIN_ROWS=<rows>.  The warp-specialised variant below (DCF_FFN_TM=3: producer waves run GEMM 1 + GELU of
// unit u while consumer waves run GEMM 2 of unit u - 1, two waves per SIMD) overlaps the phases and reaches 255 us, but
// its 32-row units fetch every weight fragment twice: 4 MB of weights per 64-row tile = 12.4 TB/s of L2 -> CU traffic over
// the chip, which is where that path saturates in practice.  Variant 4 (column-block units, every fragment fetched once
// per 64-row tile) reaches 223 us and ties the default bench (13.73 vs 13.65 M clips/s): its producer waves -- MFMA + the
// A-fragment LDS reads + GELU + transposes, ~10 k cycles per unit against 3-4 k for the consumers -- are the critical path;
// variants 5 / 6 rebalance the roles (eight producers of one 32x32 fragment; four or eight consumers; 12 / 16 waves) and
// reach 203 us = 253 TFLOP/s -- still behind the pair alone, a tie in the bench.
#include <cstdio>
#include <cstdlib>

#include "gemm_common.h"

namespace dcf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr float FFN_SA = 16.f, FFN_SW = 256.f;          // the f16x3 operand scales of gemm_bf16s.hip
constexpr int FFN_BLK = 2 * 3 * 64 * 8;                 // elements of one weight-image block (gemm_bf16s.hip)

__device__ __forceinline__ void ffn_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{x0 * FFN_SA, x1 * FFN_SA}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = __builtin_fmaf(x0, FFN_SA, -(float)h[0]), r1 = __builtin_fmaf(x1, FFN_SA, -(float)h[1]);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}

struct FfnArgs {
  GemmArgs p;                   // the second GEMM's view: A = X (lda), Ws = W2 image, bias = b2, C, R, ls, rowmask, flags, M, N = E, K = 4E
  const unsigned short* W1s;    // image of fc.weight (4E, E)
  const float* b1;              // (4E)
};

// TM = row tiles of 32 per workgroup (BM = 32 TM).  TM = 1: 75 KiB of LDS at E = 256, two workgroups per CU (the GELU
// epilogue of one overlaps the MFMAs of the other) at twice the weight bytes per row; TM = 2: 150 KiB, one per CU.
// The weight fragments of a chunk form ONE stream of 2 KT tiles (KT of W1, then KT of W2) through a four-deep register
// ring: the tile of step s + 3 -- which may belong to the next product or the next chunk -- is requested at step s.
template <int TN, int TM>
__global__ __launch_bounds__(256, TM == 1 ? 2 : 1) void ffn_f16_kernel(FfnArgs a) {
  constexpr int E = TN * 128, KT = E / 32, NC = 4, KT2 = NC * KT, BM = TM * 32;
  constexpr int ROWX = E * 2 + 16;                      // bytes per (plane, row)
  constexpr int PLANE = BM * ROWX;
  constexpr int TPR = 256 / BM, PPT = (E / 8) / TPR;    // threads per row, 8-float pieces per thread
  constexpr int NS = 4;                                 // depth of the weight-fragment ring
  static_assert((2 * KT) % NS == 0, "the ring index must be static inside a chunk");
  extern __shared__ unsigned char smem_f[];
  unsigned char* Xs = smem_f;                           // [2][BM][ROWX]
  unsigned char* Hs = smem_f + 2 * PLANE;               // [2][BM][ROWX]
  float* tile = reinterpret_cast<float*>(smem_f + 4 * PLANE) + (threadIdx.x >> 6) * EPI_WAVE_FLOATS;

  const GemmArgs& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int M = p.M;
  if (m0 >= M) return;

  const bf16x8* w1 = reinterpret_cast<const bf16x8*>(a.W1s) + lane;
  const bf16x8* w2 = reinterpret_cast<const bf16x8*>(p.Ws) + lane;
  // step s of chunk c: s < KT -> W1 block (n32 = c * E/32 + wn TN + j, kt = s); else W2 block (n32 = wn TN + j, kt = c KT + s - KT)
  bf16x8 ring[NS][2][TN][2];                            // [slot][16-k chunk][tile][plane]
  auto request = [&](int c, int s, bf16x8 (&b)[2][TN][2]) __attribute__((always_inline)) {
    if (s >= 2 * KT) { s -= 2 * KT; c = c + 1 < NC ? c + 1 : c; }        // into the next chunk (clamped at the end: redundant)
    const bool first = s < KT;
    const bf16x8* base = first ? w1 : w2;
    const int64_t nb = first ? (int64_t)c * (E / 32) + wn * TN : (int64_t)wn * TN;
    const int64_t ktot = first ? KT : KT2, kt = first ? s : c * KT + (s - KT);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][j][pl] = base[((nb + j) * ktot + kt) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) request(0, s, ring[s]);

  // ---- X tile -> two fp16 planes in LDS (pieces of 8 floats, TPR consecutive threads read TPR * 32 contiguous bytes)
  {
    const int row = tid / TPR;
    const bool ok = m0 + row < M;
    const float* src = p.A + (int64_t)(ok ? m0 + row : 0) * p.lda;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int pc = j * TPR + (tid % TPR);             // piece of 8 floats
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v0 = *reinterpret_cast<const f32x4*>(src + pc * 8);
        v1 = *reinterpret_cast<const f32x4*>(src + pc * 8 + 4);
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      ffn_split2(v0.x, v0.y, h0, l0); ffn_split2(v0.z, v0.w, h1, l1);
      ffn_split2(v1.x, v1.y, h2, l2); ffn_split2(v1.z, v1.w, h3, l3);
      *reinterpret_cast<u32x4*>(Xs + row * ROWX + pc * 16) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(Xs + PLANE + row * ROWX + pc * 16) = u32x4{l0, l1, l2, l3};
    }
  }
  __syncthreads();

  f32x16 acc2[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;

  const float unscale = 1.f / (FFN_SA * FFN_SW);
  bool bad = false;

  // K tile kt of one product: A fragments from the LDS planes at As, B fragments from ring slot `b`
  auto mfmas = [&](f32x16 (&acc)[TM][TN], const unsigned char* As, int kt, const bf16x8 (&b)[2][TN][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 af[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE + (i * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, b[c][j][1]), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][1]), __builtin_bit_cast(f16x8, b[c][j][0]), acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, b[c][j][0]), acc[i][j], 0, 0, 0);
        }
    }
  };

  const int rr = lane >> 3, c4 = (lane & 7) * 4;        // read-back role of the transpose tile: row rr + 8 q, columns c4 .. c4 + 3
  for (int c = 0; c < NC; ++c) {
    // ---- GEMM 1 of the chunk: hidden columns c * E + [wn * TN * 32, ...)
    f32x16 acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[i][j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < KT; ++s) {
      request(c, s + NS - 1, ring[(s + NS - 1) % NS]);
      mfmas(acc1, Xs, s, ring[s % NS]);
    }
    // ---- bias, GELU, split -> H planes (through the wave-private transpose tile: 4 consecutive columns of a row per lane)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int colb = (wn * TN + j) * 32 + c4;        // column inside the chunk
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + c * E + colb);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc1[i][j][e] * unscale;
          bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
          tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4) + bias;
          v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
          unsigned h0, h1, l0, l1;
          ffn_split2(v.x, v.y, h0, l0); ffn_split2(v.z, v.w, h1, l1);
          const int row = i * 32 + rr + 8 * q;
          *reinterpret_cast<u32x2*>(Hs + row * ROWX + colb * 2) = u32x2{h0, h1};
          *reinterpret_cast<u32x2*>(Hs + PLANE + row * ROWX + colb * 2) = u32x2{l0, l1};
        }
      }
    __syncthreads();                                     // the chunk of H is complete
    // ---- GEMM 2 of the chunk: K tiles c * KT .. of the 4E-wide K
#pragma unroll
    for (int s = KT; s < 2 * KT; ++s) {
      request(c, s + NS - 1, ring[(s + NS - 1) % NS]);
      mfmas(acc2, Hs, s - KT, ring[s % NS]);
    }
    __syncthreads();                                     // H may be overwritten
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc2[i][j][e] * unscale;
        bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
        acc2[i][j][e] = v;
      }
  if (bad && p.status) atomicOr(p.status, 1u);
  if (gemm_wide_ok(p)) gemm_epilogue_wide<1, 4, TM, TN>(p, acc2, m0, 0, 0, wn, lane, tile);
  else gemm_epilogue<1, 4, TM, TN>(p, acc2, m0, 0, 0, wn, r, h);
}

// ---- warp-specialised variant ---------------------------------------------------------------------------------------
// Eight waves, two per SIMD.  Waves 0-3 ("producers") run GEMM 1 + GELU for units u = (chunk, row half) = 0 .. 7 of the
// 64-row tile -- 32 rows x E hidden columns each -- into H buffer u & 1; waves 4-7 ("consumers") run GEMM 2 of unit u - 1
// out of H buffer (u - 1) & 1 into the output accumulators of that row half.  One workgroup barrier per unit; on every
// SIMD the consumer's MFMAs run under the producer's GELU / split / LDS writes.  Producers stream W1, consumers W2, each
// through its own four-deep fragment ring.  LDS: X planes [2][64][ROWX], two H buffers [2][32][ROWX], four transpose tiles.
template <int TN>
__global__ __launch_bounds__(512, 2) void ffn_f16_ws_kernel(FfnArgs a) {
  constexpr int E = TN * 128, KT = E / 32, NC = 4, KT2 = NC * KT, NU = 2 * NC;
  constexpr int ROWX = E * 2 + 16;
  constexpr int XPL = 64 * ROWX, HPL = 32 * ROWX;       // bytes of one X plane / one H-buffer plane
  constexpr int NS = 4;
  static_assert(KT % NS == 0, "ring index static per unit");
  extern __shared__ unsigned char smem_f[];
  unsigned char* Xs = smem_f;                           // [2][64][ROWX]
  unsigned char* Hb = smem_f + 2 * XPL;                 // [2 buffers][2 planes][32][ROWX]
  const GemmArgs& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 4;
  const int wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int M = p.M;
  if (m0 >= M) return;
  float* tile = reinterpret_cast<float*>(smem_f + 2 * XPL + 4 * HPL) + wn * EPI_WAVE_FLOATS;   // producers (and the final epilogue)

  bf16x8 ring[NS][2][TN][2];
  const bf16x8* wimg = reinterpret_cast<const bf16x8*>(producer ? a.W1s : p.Ws) + lane;
  // fragment tile t of this wave's stream, t = 0 .. NU * KT - 1 (unit u = t / KT, k tile kt = t % KT, chunk c = u >> 1):
  // producer: W1 block (n32 = c * E/32 + wn TN + j, kt) of KT;  consumer: W2 block (n32 = wn TN + j, c KT + kt) of KT2
  auto request = [&](int t, bf16x8 (&b)[2][TN][2]) __attribute__((always_inline)) {
    if (t >= NU * KT) t = NU * KT - 1;
    const int u = t / KT, kt = t - u * KT, c = u >> 1;
    const int64_t nb = producer ? (int64_t)c * (E / 32) + wn * TN : (int64_t)wn * TN;
    const int64_t ktot = producer ? KT : KT2, kk = producer ? kt : c * KT + kt;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][j][pl] = wimg[((nb + j) * ktot + kk) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);

  {  // X tile -> planes, all 512 threads: 8 threads per row, E / 64 pieces of 8 floats each
    const int row = tid >> 3;
    const bool ok = m0 + row < M;
    const float* src = p.A + (int64_t)(ok ? m0 + row : 0) * p.lda;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int pc = j * 8 + (tid & 7);
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v0 = *reinterpret_cast<const f32x4*>(src + pc * 8);
        v1 = *reinterpret_cast<const f32x4*>(src + pc * 8 + 4);
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      ffn_split2(v0.x, v0.y, h0, l0); ffn_split2(v0.z, v0.w, h1, l1);
      ffn_split2(v1.x, v1.y, h2, l2); ffn_split2(v1.z, v1.w, h3, l3);
      *reinterpret_cast<u32x4*>(Xs + row * ROWX + pc * 16) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(Xs + XPL + row * ROWX + pc * 16) = u32x4{l0, l1, l2, l3};
    }
  }
  __syncthreads();

  const float unscale = 1.f / (FFN_SA * FFN_SW);

  // 32 rows x (TN * 32) columns, K = E: A fragments at `As` (plane stride `apl`), B fragments = ring tiles t0 .. t0 + KT - 1
  auto product = [&](f32x16 (&acc)[TN], const unsigned char* As, int apl, int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      request(t0 + kt + NS - 1, ring[(kt + NS - 1) % NS]);
      const bf16x8 (&b)[2][TN][2] = ring[kt % NS];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 af[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[pl] = *reinterpret_cast<const bf16x8*>(As + pl * apl + r * ROWX + kt * 64 + c * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, b[c][j][1]), acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, b[c][j][0]), acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, b[c][j][0]), acc[j], 0, 0, 0);
        }
      }
    }
  };

  // step st = 0 .. NU: producers work on unit st (st < NU), consumers on unit st - 1 (st >= 1); one barrier per step.
  // The two roles are separate code paths (separate register budgets) that execute the same number of barriers.
  if (producer) {
    bool bad = false;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll 1
    for (int st = 0; st <= NU; ++st) {
      if (st < NU) {
        const int c = st >> 1, hf = st & 1;
        f32x16 acc1[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
        product(acc1, Xs + hf * 32 * ROWX, XPL, st * KT);
        unsigned char* Hw = Hb + hf * 2 * HPL;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int colb = (wn * TN + j) * 32 + c4;
          const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + c * E + colb);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = acc1[j][e] * unscale;
            bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
            tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = v;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4) + bias;
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
            unsigned h0, h1, l0, l1;
            ffn_split2(v.x, v.y, h0, l0); ffn_split2(v.z, v.w, h1, l1);
            const int row = rr + 8 * q;
            *reinterpret_cast<u32x2*>(Hw + row * ROWX + colb * 2) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(Hw + HPL + row * ROWX + colb * 2) = u32x2{l0, l1};
          }
        }
      }
      __syncthreads();
    }
    if (bad && p.status) atomicOr(p.status, 1u);
  } else {
    bool bad = false;
    f32x16 acc2[2][TN];                                 // [row half][tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
    __syncthreads();                                    // step 0: nothing to consume yet
#pragma unroll 1
    for (int u = 0; u < NU; u += 2) {                   // units u (row half 0, H buffer 0) and u + 1 (half 1, buffer 1)
      product(acc2[0], Hb, HPL, u * KT);
      __syncthreads();
      product(acc2[1], Hb + 2 * HPL, HPL, (u + 1) * KT);
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc2[i][j][e] * unscale;
          bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
          acc2[i][j][e] = v;
        }
    if (bad && p.status) atomicOr(p.status, 1u);
    if (gemm_wide_ok(p)) gemm_epilogue_wide<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, lane, tile);
    else gemm_epilogue<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, r, h);
  }
}

// ---- warp-specialised, column-block units (variant 4) -----------------------------------------------------------------
// Like the kernel above, but a unit is a block of 128 HIDDEN COLUMNS for all 64 rows: producers (4 waves, 64 rows x 32
// columns each) run GEMM 1 (K = E) + GELU of block u into H buffer u & 1 ([2 planes][64][128 * 2 + 16 B]); consumers (4
// waves, 64 rows x TN * 32 output columns each) add block u - 1 (K = 128 of the 4E) to the output accumulators.  Every
// weight fragment is fetched once per 64-row tile: 2 MB per tile at E = 256 instead of 4.
template <int TN>
__global__ __launch_bounds__(512, 2) void ffn_f16_ws2_kernel(FfnArgs a) {
  constexpr int E = TN * 128, KT = E / 32, KT2 = 4 * KT, HB = 128, NU = 4 * E / HB, KTC = HB / 32;
  constexpr int ROWX = E * 2 + 16, ROWH = HB * 2 + 16;
  constexpr int XPL = 64 * ROWX, HPL = 64 * ROWH;
  constexpr int NS = 4;
  static_assert(KT % NS == 0 && KTC % NS == 0, "ring index static per unit");
  extern __shared__ unsigned char smem_f[];
  unsigned char* Xs = smem_f;                           // [2][64][ROWX]
  unsigned char* Hb = smem_f + 2 * XPL;                 // [2 buffers][2 planes][64][ROWH]
  const GemmArgs& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 4;
  const int wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int M = p.M;
  if (m0 >= M) return;
  float* tile = reinterpret_cast<float*>(smem_f + 2 * XPL + 4 * HPL) + wn * EPI_WAVE_FLOATS;

  {  // X tile -> planes, all 512 threads
    const int row = tid >> 3;
    const bool ok = m0 + row < M;
    const float* src = p.A + (int64_t)(ok ? m0 + row : 0) * p.lda;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int pc = j * 8 + (tid & 7);
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v0 = *reinterpret_cast<const f32x4*>(src + pc * 8);
        v1 = *reinterpret_cast<const f32x4*>(src + pc * 8 + 4);
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      ffn_split2(v0.x, v0.y, h0, l0); ffn_split2(v0.z, v0.w, h1, l1);
      ffn_split2(v1.x, v1.y, h2, l2); ffn_split2(v1.z, v1.w, h3, l3);
      *reinterpret_cast<u32x4*>(Xs + row * ROWX + pc * 16) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(Xs + XPL + row * ROWX + pc * 16) = u32x4{l0, l1, l2, l3};
    }
  }
  const float unscale = 1.f / (FFN_SA * FFN_SW);

  if (producer) {
    // stream: W1 blocks (n32 = u * 4 + wn, kt) of KT, t = u * KT + kt
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(a.W1s) + lane;
    bf16x8 ring[NS][2][2];                              // [slot][16-k chunk][plane]
    auto request = [&](int t, bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
      if (t >= NU * KT) t = NU * KT - 1;
      const int u = t / KT, kt = t - u * KT;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][pl] = wimg[((int64_t)(u * 4 + wn) * KT + kt) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    __syncthreads();                                    // X planes complete
    bool bad = false;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll 1
    for (int u = 0; u <= NU; ++u) {
      if (u < NU) {
        f32x16 acc1[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          request(u * KT + kt + NS - 1, ring[(kt + NS - 1) % NS]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Xs + (i * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
              const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Xs + XPL + (i * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
              acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][1]), acc1[i], 0, 0, 0);
              acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1[i], 0, 0, 0);
              acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1[i], 0, 0, 0);
            }
          }
        }
        unsigned char* Hw = Hb + (u & 1) * 2 * HPL;
        const int colb = wn * 32 + c4;                  // column inside the block
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + u * HB + colb);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float v = acc1[i][e] * unscale;
            bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
            tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = v;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4) + bias;
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
            unsigned h0, h1, l0, l1;
            ffn_split2(v.x, v.y, h0, l0); ffn_split2(v.z, v.w, h1, l1);
            const int row = i * 32 + rr + 8 * q;
            *reinterpret_cast<u32x2*>(Hw + row * ROWH + colb * 2) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(Hw + HPL + row * ROWH + colb * 2) = u32x2{l0, l1};
          }
        }
      }
      __syncthreads();
    }
    if (bad && p.status) atomicOr(p.status, 1u);
  } else {
    // stream: W2 blocks (n32 = wn * TN + j, kt = u * KTC + k) of KT2, t = u * KTC + k
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(p.Ws) + lane;
    bf16x8 ring[NS][2][TN][2];
    auto request = [&](int t, bf16x8 (&b)[2][TN][2]) __attribute__((always_inline)) {
      if (t >= KT2) t = KT2 - 1;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) b[cc][j][pl] = wimg[((int64_t)(wn * TN + j) * KT2 + t) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    bool bad = false;
    f32x16 acc2[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
    __syncthreads();                                    // X planes complete (not used here)
    __syncthreads();                                    // producers' unit 0
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
      const unsigned char* Hr = Hb + (u & 1) * 2 * HPL;
#pragma unroll
      for (int k = 0; k < KTC; ++k) {
        request(u * KTC + k + NS - 1, ring[(k + NS - 1) % NS]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bf16x8 af[2][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(Hr + pl * HPL + (i * 32 + r) * ROWH + k * 64 + c * 32 + h * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][1]), acc2[i][j], 0, 0, 0);
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][1]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][0]), acc2[i][j], 0, 0, 0);
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][0]), acc2[i][j], 0, 0, 0);
            }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc2[i][j][e] * unscale;
          bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
          acc2[i][j][e] = v;
        }
    if (bad && p.status) atomicOr(p.status, 1u);
    if (gemm_wide_ok(p)) gemm_epilogue_wide<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, lane, tile);
    else gemm_epilogue<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, r, h);
  }
}

// ---- variant 5: twelve waves (eight producers of one 32x32 fragment each: column block wave & 3, row half wave >> 2; four
// consumers), three waves per SIMD: per unit a SIMD carries 2 x 1.5 k producer + 3 k consumer MFMA cycles and the producers'
// GELU / LDS work is spread over twice the waves.  Transposes go through 16-row half tiles (LDS: 155.6 KiB at E = 256).
template <int TN>
__global__ __launch_bounds__(768, 1) void ffn_f16_ws3_kernel(FfnArgs a) {
  constexpr int E = TN * 128, KT = E / 32, KT2 = 4 * KT, HB = 128, NU = 4 * E / HB, KTC = HB / 32;
  constexpr int ROWX = E * 2 + 16, ROWH = HB * 2 + 16;
  constexpr int XPL = 64 * ROWX, HPL = 64 * ROWH;
  constexpr int NS = 4;
  static_assert(KT % NS == 0 && KTC % NS == 0, "ring index static per unit");
  extern __shared__ unsigned char smem_f[];
  unsigned char* Xs = smem_f;                           // [2][64][ROWX]
  unsigned char* Hb = smem_f + 2 * XPL;                 // [2 buffers][2 planes][64][ROWH]
  const GemmArgs& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 8;                     // waves 0-7: (column block wave & 3, row half wave >> 2); 8-11: consumers
  const int wn = wave & 3, ph = (wave >> 2) & 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int M = p.M;
  if (m0 >= M) return;
  float* tbase = reinterpret_cast<float*>(smem_f + 2 * XPL + 4 * HPL);
  float* tile = tbase + wn * EPI_WAVE_FLOATS;          // consumers' epilogue tiles ...
  float* tile16 = tbase + wave * (16 * EPI_PITCH);     // ... alias the producers' half tiles (16 rows each), which are dead by then

  if (tid < 512) {  // X tile -> planes, the first 512 threads
    const int row = tid >> 3;
    const bool ok = m0 + row < M;
    const float* src = p.A + (int64_t)(ok ? m0 + row : 0) * p.lda;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int pc = j * 8 + (tid & 7);
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v0 = *reinterpret_cast<const f32x4*>(src + pc * 8);
        v1 = *reinterpret_cast<const f32x4*>(src + pc * 8 + 4);
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      ffn_split2(v0.x, v0.y, h0, l0); ffn_split2(v0.z, v0.w, h1, l1);
      ffn_split2(v1.x, v1.y, h2, l2); ffn_split2(v1.z, v1.w, h3, l3);
      *reinterpret_cast<u32x4*>(Xs + row * ROWX + pc * 16) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(Xs + XPL + row * ROWX + pc * 16) = u32x4{l0, l1, l2, l3};
    }
  }
  const float unscale = 1.f / (FFN_SA * FFN_SW);

  if (producer) {
    // stream: W1 blocks (n32 = u * 4 + wn, kt) of KT, t = u * KT + kt
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(a.W1s) + lane;
    bf16x8 ring[NS][2][2];                              // [slot][16-k chunk][plane]
    auto request = [&](int t, bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
      if (t >= NU * KT) t = NU * KT - 1;
      const int u = t / KT, kt = t - u * KT;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][pl] = wimg[((int64_t)(u * 4 + wn) * KT + kt) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    __syncthreads();                                    // X planes complete
    bool bad = false;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll 1
    for (int u = 0; u <= NU; ++u) {
      if (u < NU) {
        f32x16 acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          request(u * KT + kt + NS - 1, ring[(kt + NS - 1) % NS]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Xs + (ph * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Xs + XPL + (ph * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][1]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1, 0, 0, 0);
          }
        }
        unsigned char* Hw = Hb + (u & 1) * 2 * HPL;
        const int colb = wn * 32 + c4;                  // column inside the block
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + u * HB + colb);
        // the 32x32 fragment goes through a 16-row transpose tile in two halves: elements e with (e >> 3) == hh are rows
        // 16 hh .. 16 hh + 15 of the fragment (row = (e & 3) + 8 (e >> 2) + 4 h)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) {
            const int e = hh * 8 + e8;
            const float v = acc1[e] * unscale;
            bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
            tile16[((e & 3) + 8 * ((e >> 2) & 1) + 4 * h) * EPI_PITCH + r] = v;
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile16 + (rr + 8 * q) * EPI_PITCH + c4) + bias;
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
            unsigned h0, h1, l0, l1;
            ffn_split2(v.x, v.y, h0, l0); ffn_split2(v.z, v.w, h1, l1);
            const int row = ph * 32 + hh * 16 + rr + 8 * q;
            *reinterpret_cast<u32x2*>(Hw + row * ROWH + colb * 2) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(Hw + HPL + row * ROWH + colb * 2) = u32x2{l0, l1};
          }
        }
      }
      __syncthreads();
    }
    if (bad && p.status) atomicOr(p.status, 1u);
  } else {
    // stream: W2 blocks (n32 = wn * TN + j, kt = u * KTC + k) of KT2, t = u * KTC + k
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(p.Ws) + lane;
    constexpr int NS = 2;                               // three waves per SIMD: 168 registers, the output accumulators take 64
    bf16x8 ring[NS][2][TN][2];
    auto request = [&](int t, bf16x8 (&b)[2][TN][2]) __attribute__((always_inline)) {
      if (t >= KT2) t = KT2 - 1;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) b[cc][j][pl] = wimg[((int64_t)(wn * TN + j) * KT2 + t) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    bool bad = false;
    f32x16 acc2[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
    __syncthreads();                                    // X planes complete (not used here)
    __syncthreads();                                    // producers' unit 0
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
      const unsigned char* Hr = Hb + (u & 1) * 2 * HPL;
#pragma unroll
      for (int k = 0; k < KTC; ++k) {
        request(u * KTC + k + NS - 1, ring[(k + NS - 1) % NS]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bf16x8 af[2][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(Hr + pl * HPL + (i * 32 + r) * ROWH + k * 64 + c * 32 + h * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][1]), acc2[i][j], 0, 0, 0);
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][1]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][0]), acc2[i][j], 0, 0, 0);
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][0]), __builtin_bit_cast(f16x8, ring[k % NS][c][j][0]), acc2[i][j], 0, 0, 0);
            }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = acc2[i][j][e] * unscale;
          bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
          acc2[i][j][e] = v;
        }
    if (bad && p.status) atomicOr(p.status, 1u);
    if (gemm_wide_ok(p)) gemm_epilogue_wide<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, lane, tile);
    else gemm_epilogue<1, 4, 2, TN>(p, acc2, m0, 0, 0, wn, r, h);
  }
}

// ---- variant 6 (E = 256): sixteen waves -- the eight producers of variant 5 and EIGHT consumers of 64 rows x 32 output columns
// (each W2 fragment still fetched once), four waves per SIMD at <= 128 registers, four-deep rings on both sides.
template <int TN>
__global__ __launch_bounds__(1024, 1) void ffn_f16_ws4_kernel(FfnArgs a) {
  static_assert(TN == 2, "sixteen-wave variant: E = 256 only");
  constexpr int E = TN * 128, KT = E / 32, KT2 = 4 * KT, HB = 128, NU = 4 * E / HB, KTC = HB / 32;
  constexpr int ROWX = E * 2 + 16, ROWH = HB * 2 + 16;
  constexpr int XPL = 64 * ROWX, HPL = 64 * ROWH;
  constexpr int NS = 4;
  static_assert(KT % NS == 0 && KTC % NS == 0, "ring index static per unit");
  extern __shared__ unsigned char smem_f[];
  unsigned char* Xs = smem_f;                           // [2][64][ROWX]
  unsigned char* Hb = smem_f + 2 * XPL;                 // [2 buffers][2 planes][64][ROWH]
  const GemmArgs& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave < 8;                     // waves 0-7: (column block wave & 3, row half wave >> 2); 8-15: consumers, 32 output columns each
  const int cw = wave - 8;
  const int wn = wave & 3, ph = (wave >> 2) & 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 64;
  const int M = p.M;
  if (m0 >= M) return;
  float* tbase = reinterpret_cast<float*>(smem_f + 2 * XPL + 4 * HPL);
  float* tile16 = tbase + (wave & 7) * (16 * EPI_PITCH);   // producers' half tiles (16 rows each)

  if (tid < 512) {  // X tile -> planes, the first 512 threads
    const int row = tid >> 3;
    const bool ok = m0 + row < M;
    const float* src = p.A + (int64_t)(ok ? m0 + row : 0) * p.lda;
#pragma unroll
    for (int j = 0; j < E / 64; ++j) {
      const int pc = j * 8 + (tid & 7);
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        v0 = *reinterpret_cast<const f32x4*>(src + pc * 8);
        v1 = *reinterpret_cast<const f32x4*>(src + pc * 8 + 4);
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      ffn_split2(v0.x, v0.y, h0, l0); ffn_split2(v0.z, v0.w, h1, l1);
      ffn_split2(v1.x, v1.y, h2, l2); ffn_split2(v1.z, v1.w, h3, l3);
      *reinterpret_cast<u32x4*>(Xs + row * ROWX + pc * 16) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(Xs + XPL + row * ROWX + pc * 16) = u32x4{l0, l1, l2, l3};
    }
  }
  const float unscale = 1.f / (FFN_SA * FFN_SW);

  if (producer) {
    // stream: W1 blocks (n32 = u * 4 + wn, kt) of KT, t = u * KT + kt
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(a.W1s) + lane;
    bf16x8 ring[NS][2][2];                              // [slot][16-k chunk][plane]
    auto request = [&](int t, bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
      if (t >= NU * KT) t = NU * KT - 1;
      const int u = t / KT, kt = t - u * KT;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][pl] = wimg[((int64_t)(u * 4 + wn) * KT + kt) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    __syncthreads();                                    // X planes complete
    bool bad = false;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll 1
    for (int u = 0; u <= NU; ++u) {
      if (u < NU) {
        f32x16 acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          request(u * KT + kt + NS - 1, ring[(kt + NS - 1) % NS]);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Xs + (ph * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Xs + XPL + (ph * 32 + r) * ROWX + kt * 64 + c * 32 + h * 16);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][1]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[kt % NS][c][0]), acc1, 0, 0, 0);
          }
        }
        unsigned char* Hw = Hb + (u & 1) * 2 * HPL;
        const int colb = wn * 32 + c4;                  // column inside the block
        const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b1 + u * HB + colb);
        // the 32x32 fragment goes through a 16-row transpose tile in two halves: elements e with (e >> 3) == hh are rows
        // 16 hh .. 16 hh + 15 of the fragment (row = (e & 3) + 8 (e >> 2) + 4 h)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) {
            const int e = hh * 8 + e8;
            const float v = acc1[e] * unscale;
            bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
            tile16[((e & 3) + 8 * ((e >> 2) & 1) + 4 * h) * EPI_PITCH + r] = v;
          }
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile16 + (rr + 8 * q) * EPI_PITCH + c4) + bias;
            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
            unsigned h0, h1, l0, l1;
            ffn_split2(v.x, v.y, h0, l0); ffn_split2(v.z, v.w, h1, l1);
            const int row = ph * 32 + hh * 16 + rr + 8 * q;
            *reinterpret_cast<u32x2*>(Hw + row * ROWH + colb * 2) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(Hw + HPL + row * ROWH + colb * 2) = u32x2{l0, l1};
          }
        }
      }
      __syncthreads();
    }
    if (bad && p.status) atomicOr(p.status, 1u);
  } else {
    // stream: W2 blocks (n32 = cw, kt = u * KTC + k) of KT2, t = u * KTC + k
    const bf16x8* wimg = reinterpret_cast<const bf16x8*>(p.Ws) + lane;
    bf16x8 ring[NS][2][2];                              // [slot][16-k chunk][plane]
    auto request = [&](int t, bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
      if (t >= KT2) t = KT2 - 1;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) b[cc][pl] = wimg[((int64_t)cw * KT2 + t) * (FFN_BLK / 8) + (cc * 3 + pl) * 64];
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) request(s, ring[s]);
    bool bad = false;
    f32x16 acc2[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[i][0][e] = 0.f;
    __syncthreads();                                    // X planes complete (not used here)
    __syncthreads();                                    // producers' unit 0
#pragma unroll 1
    for (int u = 0; u < NU; ++u) {
      const unsigned char* Hr = Hb + (u & 1) * 2 * HPL;
#pragma unroll
      for (int k = 0; k < KTC; ++k) {
        request(u * KTC + k + NS - 1, ring[(k + NS - 1) % NS]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Hr + (i * 32 + r) * ROWH + k * 64 + c * 32 + h * 16);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Hr + HPL + (i * 32 + r) * ROWH + k * 64 + c * 32 + h * 16);
            acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[k % NS][c][1]), acc2[i][0], 0, 0, 0);
            acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, ring[k % NS][c][0]), acc2[i][0], 0, 0, 0);
            acc2[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, ring[k % NS][c][0]), acc2[i][0], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc2[i][0][e] * unscale;
        bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
        acc2[i][0][e] = v;
      }
    if (bad && p.status) atomicOr(p.status, 1u);
    gemm_epilogue<1, 8, 2, 1>(p, acc2, m0, 0, 0, cw, r, h);
  }
}

bool ffn_fused_supported(int E) { return E == 128 || E == 256; }

// p: the proj GEMM as launch_gemm_split would take it (A = the fc input X, K = 4E, N = E, Ws = proj image, bias = proj bias);
// W1s / b1: fc image and bias.  Both images must be f16x3 images.
int launch_ffn_f16(const GemmArgs& p, const unsigned short* W1s, const float* b1, hipStream_t st, int variant) {
  const int E = p.N;
  DCF_CHECK(ffn_fused_supported(E) && p.K == 4 * E, "launch_ffn_f16: E=%d K=%d unsupported", E, p.K);
  DCF_CHECK(p.A && p.Ws && W1s && b1 && p.bias && p.C && p.lda % 4 == 0, "launch_ffn_f16: null / misaligned operand");
  if (p.M <= 0) return 0;
  FfnArgs a{p, W1s, b1};
  static const int tm_env = getenv("DCF_FFN_TM") ? atoi(getenv("DCF_FFN_TM")) : 2;
  const int tm = variant ? variant : tm_env;          // 1: 32-row tiles, 2: 64-row tiles, 3: warp-specialised
  const int bm = tm == 2 ? 64 : 32;
  const int rowx = E * 2 + 16;
  const size_t lds = (size_t)4 * bm * rowx + (size_t)4 * EPI_WAVE_FLOATS * sizeof(float);
  ProfScope prof("ffn_fused_f16x3", st, 2.0 * 2.0 * p.M * (double)E * 4.0 * E, 4.0 * (3.0 * p.M * E + 2.0 * 8.0 * E * E));
  dim3 grid((p.M + bm - 1) / bm);
#define FFN_LAUNCH(TN_, TM_) do { \
    static bool done = false; \
    if (!done) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_kernel<TN_, TM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; } \
    hipLaunchKernelGGL((ffn_f16_kernel<TN_, TM_>), grid, dim3(256), lds, st, a); } while (0)
  if (tm == 6 && E == 256) {                              // sixteen waves: eight producers, eight consumers
    const size_t lds6 = (size_t)2 * 64 * rowx + (size_t)4 * 64 * (128 * 2 + 16) + (size_t)4 * EPI_WAVE_FLOATS * sizeof(float);
    static bool done6 = false;
    if (!done6) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6)); done6 = true; }
    hipLaunchKernelGGL(ffn_f16_ws4_kernel<2>, dim3((p.M + 63) / 64), dim3(1024), lds6, st, a);
    DCF_HIP(hipGetLastError());
    return 0;
  }
  if (tm == 5 || tm == 6) {                               // twelve waves: eight narrow producers, four consumers
    const size_t lds5 = (size_t)2 * 64 * rowx + (size_t)4 * 64 * (128 * 2 + 16) + (size_t)4 * EPI_WAVE_FLOATS * sizeof(float);
    dim3 grid5((p.M + 63) / 64);
    static bool done5[2] = {false, false};
    if (E == 256) {
      if (!done5[1]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5)); done5[1] = true; }
      hipLaunchKernelGGL(ffn_f16_ws3_kernel<2>, grid5, dim3(768), lds5, st, a);
    } else {
      if (!done5[0]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5)); done5[0] = true; }
      hipLaunchKernelGGL(ffn_f16_ws3_kernel<1>, grid5, dim3(768), lds5, st, a);
    }
    DCF_HIP(hipGetLastError());
    return 0;
  }
  if (tm == 4) {                                          // warp-specialised, column-block units
    const size_t lds4 = (size_t)2 * 64 * rowx + (size_t)4 * 64 * (128 * 2 + 16) + (size_t)4 * EPI_WAVE_FLOATS * sizeof(float);
    dim3 grid4((p.M + 63) / 64);
    static bool done4[2] = {false, false};
    if (E == 256) {
      if (!done4[1]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4)); done4[1] = true; }
      hipLaunchKernelGGL(ffn_f16_ws2_kernel<2>, grid4, dim3(512), lds4, st, a);
    } else {
      if (!done4[0]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4)); done4[0] = true; }
      hipLaunchKernelGGL(ffn_f16_ws2_kernel<1>, grid4, dim3(512), lds4, st, a);
    }
    DCF_HIP(hipGetLastError());
    return 0;
  }
  if (tm == 3) {                                          // warp-specialised: 64 rows, eight waves
    const size_t lds_ws = (size_t)2 * 64 * rowx + (size_t)4 * 32 * rowx + (size_t)4 * EPI_WAVE_FLOATS * sizeof(float);
    dim3 grid_ws((p.M + 63) / 64);
    static bool done_ws[2] = {false, false};
    if (E == 256) {
      if (!done_ws[1]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws)); done_ws[1] = true; }
      hipLaunchKernelGGL(ffn_f16_ws_kernel<2>, grid_ws, dim3(512), lds_ws, st, a);
    } else {
      if (!done_ws[0]) { DCF_HIP(hipFuncSetAttribute((const void*)ffn_f16_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ws)); done_ws[0] = true; }
      hipLaunchKernelGGL(ffn_f16_ws_kernel<1>, grid_ws, dim3(512), lds_ws, st, a);
    }
    DCF_HIP(hipGetLastError());
    return 0;
  }
  if (E == 256) { if (tm == 2) FFN_LAUNCH(2, 2); else FFN_LAUNCH(2, 1); }
  else { if (tm == 2) FFN_LAUNCH(1, 2); else FFN_LAUNCH(1, 1); }
#undef FFN_LAUNCH
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
