// Pieces shared by the fp32-MFMA GEMM (gemm.hip) and the bf16-split GEMM (gemm_bf16s.hip).
#pragma once
#include "common.h"
#include "gemm.h"

namespace dcf {

struct GemmBatch {
  GemmArgs g[3];
};

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (private L2 each), so blocks b and
// b + 8 share an L2.  The column tiles of one row tile all read the same A rows: number the tiles so that they
// get ids 8 apart (same XCD, dispatched back to back) instead of a whole grid row apart -- measured with
// rocprofv3 FETCH_SIZE the k3 head convolution (3 column tiles) re-fetched its activations 9x before this.
//   id = (row_tile / 8 * ncol + col) * 8 + row_tile % 8
template <int BM, int BN>
__device__ __forceinline__ bool tile_origin(const GemmArgs& p, int& m0, int& n0) {
  const int ncol = p.N / BN;
  const int id = blockIdx.x;
  const int grp = id / (8 * ncol), w = id - grp * (8 * ncol);
  const int rt = grp * 8 + (w & 7), ct = w >> 3;
  m0 = rt * BM;
  n0 = ct * BN;
  return m0 < p.M;              // padding tiles of the last group exit (uniformly, before any barrier)
}
template <int BM, int BN>
static inline unsigned tile_grid(const GemmArgs& p) {
  const int rows = (p.M + BM - 1) / BM;
  return (unsigned)(((rows + 7) / 8) * 8 * (p.N / BN));
}

// Fused epilogue of one workgroup tile.  acc[i][j] is the 32x32 D fragment of wave tile (i, j):
// col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)  (same map for f32 and bf16 MFMA).
// Written for instruction count: with K = 256 the K loop is only 8 tiles long and a per-element epilogue (row
// bound branch + 64-bit address arithmetic per store, ~25 vector instructions per element) was 4/5 of all vector
// instructions of the kernel (rocprofv3 SQ_INSTS_VALU).  Here full tiles skip the row tests (FULL), addresses are
// 32-bit offsets from a uniform tile base, and the residual / row-mask loads are issued together before the
// first store (stores to C may alias R for the compiler: interleaved they were 16 dependent round trips).
template <bool FULL, int ACT, bool RES, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_body(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0_, int n0_, int wm, int wn,
                                                   int r, int h) {
  const int flags = p.flags;
  const int m0 = __builtin_amdgcn_readfirstlane(m0_), n0 = __builtin_amdgcn_readfirstlane(n0_);   // scalar tile bases
  const int rows_left = p.M - m0;                                   // > 0
  float* __restrict__ Cb = p.C + (int64_t)m0 * p.ldc + n0;
  const float* __restrict__ Rb = RES ? p.R + (int64_t)m0 * p.ldr + n0 : nullptr;
  const uint8_t* __restrict__ Mb = (RES && (flags & (G_RES_MASK | G_OUT_MASK))) ? p.rowmask + m0 : nullptr;
  const unsigned ldc = (unsigned)p.ldc, ldr = (unsigned)p.ldr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned row0 = (unsigned)((wm * TM + i) * 32 + 4 * h);   // + (e & 3) + 8 * (e >> 2)
    float mk[16];
    if constexpr (RES) {
#pragma unroll
      for (int e = 0; e < 16; ++e) mk[e] = 1.f;
      if (Mb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
          if (FULL || (int)row < rows_left) mk[e] = Mb[row] ? 1.f : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned col = (unsigned)((wn * TN + j) * 32 + r);
      const float bias = p.bias ? p.bias[n0 + col] : 0.f;
      float ls = 1.f;
      float res[16];
      if constexpr (RES) {
        if (p.ls) ls = p.ls[n0 + col];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
          res[e] = (FULL || (int)row < rows_left) ? Rb[row * ldr + col] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
        float v = acc[i][j][e] + bias;
        if constexpr (ACT == 1) v = gelu_erf(v);
        if constexpr (ACT == 2) v = fmaxf(v, 0.f);
        if constexpr (RES) {
          if (flags & G_OUT_MASK) v *= mk[e];
          float r_ = res[e];
          if (flags & G_RES_MASK) r_ *= mk[e];
          v = r_ + ls * v;
        }
        if (FULL || (int)row < rows_left) Cb[row * ldc + col] = v;
      }
    }
  }
}

template <bool FULL, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_flags(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                    int r, int h) {
  // one uniform branch on the flag class instead of per-element selects
  const int act = (p.flags & G_GELU) ? 1 : ((p.flags & G_RELU) ? 2 : 0);
  if (p.flags & G_RES) {
    if (act == 0) gemm_epilogue_body<FULL, 0, true, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
    else if (act == 1) gemm_epilogue_body<FULL, 1, true, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
    else gemm_epilogue_body<FULL, 2, true, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
  } else {
    if (act == 0) gemm_epilogue_body<FULL, 0, false, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
    else if (act == 1) gemm_epilogue_body<FULL, 1, false, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
    else gemm_epilogue_body<FULL, 2, false, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
  }
}

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int r,
                                              int h) {
  if (m0 + WM * TM * 32 <= p.M) gemm_epilogue_flags<true, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
  else gemm_epilogue_flags<false, WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
}

}  // namespace dcf
