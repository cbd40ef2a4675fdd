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
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int r,
                                              int h) {
  const int M = p.M;
  const int flags = p.flags;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + r;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float ls = ((flags & G_RES) && p.ls) ? p.ls[col] : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < M) {
          float v = acc[i][j][e] + bias;
          if (flags & G_GELU) v = gelu_erf(v);
          if (flags & G_RELU) v = fmaxf(v, 0.f);
          if (flags & G_RES) {
            float mk = 1.f;
            if (flags & (G_RES_MASK | G_OUT_MASK)) mk = p.rowmask[row] ? 1.f : 0.f;
            if (flags & G_OUT_MASK) v *= mk;
            float res = p.R[(int64_t)row * p.ldr + col];
            if (flags & G_RES_MASK) res *= mk;
            v = res + ls * v;
          }
          p.C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

}  // namespace dcf
