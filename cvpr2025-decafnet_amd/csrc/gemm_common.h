// Pieces shared by the fp32-MFMA GEMM (gemm.hip) and the bf16-split GEMM (gemm_bf16s.hip).
#pragma once
#include "common.h"
#include "gemm.h"

namespace dcf {

// Up to three GEMMs of one shape share a grid (blockIdx.z selects g[z]).  zcount > 0 (channel-major tile kernel without scores only,
// launch_gemm_split_z): blockIdx.z < zcount selects operand set z of g[0] -- the same GEMM over zcount (A, C) pairs, e.g. the expert
// half of vid_map of every video of a forward in ONE launch, each with its own row-tile selection (GemmArgs::tile_skip).
struct GemmBatch {
  GemmArgs g[3];
  int zcount;
  const float* zA[GEMM_ZMAX];
  float* zC[GEMM_ZMAX];
  const uint8_t* zskip[GEMM_ZMAX];
  int zskip_nq[GEMM_ZMAX];
};
template <class T>
__device__ __forceinline__ T gemm_zsel(const T (&a)[GEMM_ZMAX], int z) {      // (a cascade of scalar selects: a kernel-argument array indexed by
  T r = a[0];                                                                //  blockIdx would be copied to scratch)
#pragma unroll
  for (int i = 1; i < GEMM_ZMAX; ++i) r = z == i ? a[i] : r;
  return r;
}

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

// ---- epilogue with 16-byte stores --------------------------------------------------------------------------------
// The MFMA D fragment gives a lane ONE column of 16 rows: 16 dword stores (and 16 dword residual loads) per 32x32
// fragment, 64 store instructions per 64x64 wave tile, and the store tail of a kernel is issue bound, not bandwidth
// bound.  Here every fragment goes through a wave-private LDS tile [32][36] (conflict-free dword writes in D layout,
// 16-byte reads) and comes back as 4 consecutive columns of one row per lane: 4 stores and 4 residual loads of 16
// bytes per fragment.  Needs 16-byte aligned C / R rows (ldc, ldr multiples of 4, checked by the caller).
constexpr int EPI_PITCH = 36;
constexpr int EPI_WAVE_FLOATS = 32 * EPI_PITCH;

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void stats_flush(const GemmArgs& p, const float* wg_out, int m0, int c0, int n_cols, int tid, int bn_out);

// Row statistics between GEMMs (GemmArgs::stats_in / stats_out): `wg` is a workgroup-shared LDS block of
// stats_lds_floats<WM, WN, TM>() floats behind the wave-private transpose tiles: [BM] (mean, rstd) of the consumer's rows,
// then [WN][BM] (sum, sum of squares) of the producer's wave tiles.
template <int WM, int WN, int TM>
constexpr int stats_lds_floats() { return (1 + WN) * (WM * TM * 32) * 2; }

template <bool FULL, int ACT, bool RES, int WM, int WN, int TM, int TN, bool STATS>
__device__ __forceinline__ void gemm_epilogue_wide_body(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0_, int n0_, int wm, int wn,
                                                        int lane, float* tile /* wave private, EPI_WAVE_FLOATS */, float* wg) {
  constexpr int BM = WM * TM * 32;
  // STATS is a kernel template argument: the plain kernels carry none of this (it cost the 128x256 tile 144 bytes of scratch)
  const bool fold = STATS && p.stats_in != nullptr, emit = STATS && p.stats_out != nullptr;       // uniform
  const float* wg_in = wg;
  float* wg_out = wg + 2 * BM;
  const int flags = p.flags;
  const int m0 = __builtin_amdgcn_readfirstlane(m0_), n0 = __builtin_amdgcn_readfirstlane(n0_);
  const int rows_left = p.M - m0;
  float* __restrict__ Cb = p.C + (int64_t)m0 * p.ldc + n0;
  const float* __restrict__ Rb = RES ? p.R + (int64_t)m0 * p.ldr + n0 : nullptr;
  const uint8_t* __restrict__ Mb = (RES && (flags & (G_RES_MASK | G_OUT_MASK))) ? p.rowmask + m0 : nullptr;
  const unsigned ldc = (unsigned)p.ldc, ldr = (unsigned)p.ldr;
  const int r = lane & 31, h = lane >> 5;
  const int rr = lane >> 3, c4 = (lane & 7) * 4;                       // read-back role: row rr + 8 p, columns c4 .. c4+3
  constexpr int NF = TM * TN;                                          // 32x32 fragments of the wave, f = i * TN + j
  // The residual rows of fragment f + 2 are requested while fragment f is transposed and stored: one exposed round trip per
  // wave tile instead of one per fragment (8 x ~2 k cycles of a 128x256 tile's 16 k-cycle epilogue at K = 256).  Everything
  // else the fragments read (row masks, bias, LayerScale) is fetched before the first of them.
  f32x4 res[2][4];
  auto load_res = [&](int f, f32x4 (&dst)[4]) __attribute__((always_inline)) {
    const int i = f / TN, j = f % TN;
    const unsigned colb = (unsigned)((wn * TN + j) * 32 + c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned row = (unsigned)((wm * TM + i) * 32 + rr + 8 * q);
      dst[q] = (FULL || (int)row < rows_left) ? *reinterpret_cast<const f32x4*>(Rb + row * ldr + colb) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if constexpr (RES) {
    load_res(0, res[0]);
    if constexpr (NF > 1) load_res(1, res[1]);
  }
  unsigned mkbits = ~0u;                                               // bit 4 i + q: row mask of read-back row q of fragment row i
  if constexpr (RES) {
    if (Mb) {
      mkbits = 0u;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = (wm * TM + i) * 32 + rr + 8 * q;
          if ((FULL || row < rows_left) ? Mb[row] != 0 : true) mkbits |= 1u << (4 * i + q);
        }
    }
  }
  f32x4 bias[TN], ls[TN], lns[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const unsigned colb = (unsigned)((wn * TN + j) * 32 + c4);
    bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ls[j] = f32x4{1.f, 1.f, 1.f, 1.f};
    lns[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias[j] = *reinterpret_cast<const f32x4*>(p.bias + n0 + colb);
    if constexpr (RES) { if (p.ls) ls[j] = *reinterpret_cast<const f32x4*>(p.ls + n0 + colb); }
    if (fold) lns[j] = *reinterpret_cast<const f32x4*>(p.ln_s + n0 + colb);
  }
  float ps[4] = {0.f, 0.f, 0.f, 0.f}, pss[4] = {0.f, 0.f, 0.f, 0.f};     // emit: this lane's share of rows rr + 8 q of row block i
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int i = f / TN, j = f % TN;
    const unsigned colb = (unsigned)((wn * TN + j) * 32 + c4);
    // D layout -> LDS: element e of lane (r, h) is row (e & 3) + 8 (e >> 2) + 4 h, column r
#pragma unroll
    for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = acc[i][j][e];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned row = (unsigned)((wm * TM + i) * 32 + rr + 8 * q);
      f32x4 v = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4);
      if (fold) {                                                        // LayerNorm of the A row, folded (see GemmArgs)
        float mean = wg_in[2 * ((wm * TM + i) * 32 + rr + 8 * q)], rstd = wg_in[2 * ((wm * TM + i) * 32 + rr + 8 * q) + 1];
#if defined(DCF_FOLD_NOP)        // diagnostic builds of tools/micro/pkfma_repro.py (profiles/r04_pkfma_hazard.md): never the product build
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(mean), "+v"(rstd));
#elif defined(DCF_FOLD_NOP0)
        asm volatile("s_nop 0" : "+v"(mean), "+v"(rstd));
#elif defined(DCF_FOLD_NOP3)
        asm volatile("s_nop 3" : "+v"(mean), "+v"(rstd));
#elif defined(DCF_FOLD_EMPTY)
        asm volatile("" : "+v"(mean), "+v"(rstd));
#elif defined(DCF_FOLD_MOV)
        { float m2, r2; asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(m2), "=&v"(r2) : "v"(mean), "v"(rstd)); mean = m2; rstd = r2; }
#endif
        // explicit fused multiply-adds: the compiler's own contraction differs between tile instantiations, and a row must
        // get the same bits whatever tile shape its batch size selects
        const f32x4 l = lns[j], bb = bias[j];
        v.x = __builtin_fmaf(__builtin_fmaf(-mean, l.x, v.x), rstd, bb.x);
        v.y = __builtin_fmaf(__builtin_fmaf(-mean, l.y, v.y), rstd, bb.y);
        v.z = __builtin_fmaf(__builtin_fmaf(-mean, l.z, v.z), rstd, bb.z);
        v.w = __builtin_fmaf(__builtin_fmaf(-mean, l.w, v.w), rstd, bb.w);
      } else {
        v += bias[j];
      }
      if constexpr (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      if constexpr (ACT == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if constexpr (RES) {
        const float mkq = (mkbits >> (4 * i + q)) & 1u ? 1.f : 0.f;
        if (flags & G_OUT_MASK) v *= mkq;
        f32x4 r_ = res[f & 1][q];
        if (flags & G_RES_MASK) r_ *= mkq;
        v = r_ + ls[j] * v;
      }
      if (FULL || (int)row < rows_left) *reinterpret_cast<f32x4*>(Cb + row * ldc + colb) = v;
      if (emit) {
        if (j == 0) { ps[q] = 0.f; pss[q] = 0.f; }
        ps[q] += (v.x + v.y) + (v.z + v.w);
        pss[q] += __builtin_fmaf(v.x, v.x, v.y * v.y) + __builtin_fmaf(v.z, v.z, v.w * v.w);
      }
    }
    if (emit && j == TN - 1) {                                          // the wave's TN * 32 columns of rows (i, rr + 8 q) are complete
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float s1 = group_sum<8>(ps[q]), s2 = group_sum<8>(pss[q]);
        if ((lane & 7) == 0) {
          float* o = wg_out + 2 * (wn * BM + (wm * TM + i) * 32 + rr + 8 * q);
          o[0] = s1; o[1] = s2;
        }
      }
    }
    if constexpr (RES) { if (f + 2 < NF) load_res(f + 2, res[f & 1]); }
  }
  if (emit) stats_flush<WM, WN, TM, TN>(p, wg_out, m0, n0, p.N, (wm * WN + wn) * 64 + lane, WN * TN * 32);
}

// the consumer's rows: (mean, rstd) from the slots of stats_in -> wg[0 .. 2 BM).  Called at kernel entry, before the K loop:
// the block lies behind the A tile AND the transpose tiles in LDS, the K loop's barriers order it before the epilogue reads
// it, and the latency of the statistics loads (a dependent chain per workgroup: +15 % on the ffn.fc kernels when it sat in
// the epilogue) hides under the first operand loads
template <int BM, int NT>
__device__ __forceinline__ void stats_load(const GemmArgs& p, float* wg, int m0, int tid) {
  for (int t = tid; t < BM; t += NT) {
    const int row = m0 + t;
    float s1 = 0.f, s2 = 0.f;
    if (row < p.M) {
      const float* sp = p.stats_in + (int64_t)row * p.stats_slots * 2;
      for (int k = 0; k < p.stats_slots; ++k) { s1 += sp[2 * k]; s2 += sp[2 * k + 1]; }
    }
    const float inv = 1.0f / (float)p.K;
    const float mean = s1 * inv;
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.f);
    wg[2 * t] = mean;
    wg[2 * t + 1] = 1.0f / sqrtf(var + 1e-5f);
    if (row < p.M && ln_ill(mean, var) && p.status) atomicOr(p.status, 2u);      // (common.h LN_ILL_RATIO)
  }
}

// the producer's rows: add the WN wave tiles of a row in a fixed order and write the slots this workgroup tile covers
// (n_cols output columns per row in all, the tile starts at output column c0 and is BN_out wide)
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void stats_flush(const GemmArgs& p, const float* wg_out, int m0, int c0, int n_cols, int tid,
                                            int bn_out) {
  constexpr int BM = WM * TM * 32, NT = WM * WN * 64;
  __syncthreads();
  const int slots = n_cols / p.stats_w, first = c0 / p.stats_w, cover = bn_out / p.stats_w;
  for (int t = tid; t < BM; t += NT) {
    const int row = m0 + t;
    if (row >= p.M) continue;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) { s1 += wg_out[2 * (w * BM + t)]; s2 += wg_out[2 * (w * BM + t) + 1]; }
    float* o = p.stats_out + ((int64_t)row * slots + first) * 2;
    o[0] = s1; o[1] = s2;
    for (int k = 1; k < cover; ++k) { o[2 * k] = 0.f; o[2 * k + 1] = 0.f; }
  }
}

template <bool FULL, int WM, int WN, int TM, int TN, bool STATS>
__device__ __forceinline__ void gemm_epilogue_wide_flags(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                         int lane, float* tile, float* wg) {
  const int act = (p.flags & G_GELU) ? 1 : ((p.flags & G_RELU) ? 2 : 0);
  if (p.flags & G_RES) {
    if (act == 0) gemm_epilogue_wide_body<FULL, 0, true, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
    else if (act == 1) gemm_epilogue_wide_body<FULL, 1, true, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
    else gemm_epilogue_wide_body<FULL, 2, true, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
  } else {
    if (act == 0) gemm_epilogue_wide_body<FULL, 0, false, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
    else if (act == 1) gemm_epilogue_wide_body<FULL, 1, false, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
    else gemm_epilogue_wide_body<FULL, 2, false, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
  }
}

// true if the 16-byte epilogue applies to this operand set (row pitches and base pointers 16-byte aligned)
__device__ __forceinline__ bool gemm_wide_ok(const GemmArgs& p) {
  bool ok = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
  if (p.flags & G_RES) ok = ok && (p.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(p.R) & 15) == 0;
  if (p.bias) ok = ok && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
  if ((p.flags & G_RES) && p.ls) ok = ok && (reinterpret_cast<uintptr_t>(p.ls) & 15) == 0;
  if (p.stats_in) ok = ok && (reinterpret_cast<uintptr_t>(p.ln_s) & 15) == 0;
  return ok;
}

template <int WM, int WN, int TM, int TN, bool STATS = false>
__device__ __forceinline__ void gemm_epilogue_wide(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                   int lane, float* tile, float* wg) {
  if (m0 + WM * TM * 32 <= p.M) gemm_epilogue_wide_flags<true, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
  else gemm_epilogue_wide_flags<false, WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, tile, wg);
}

// ---- epilogue of the AdaLN projection (G_ADALN) ---------------------------------------------------------------
// The N output columns come in blocks of 64: 32 scale columns, then the 32 shift columns of the same channels (the caller
// orders the weight rows like that), so the two 32x32 fragments of a wave's 64-column tile are (scale, shift) of the same 32
// channels.  After the transposition through the wave's LDS tile a lane holds 4 consecutive channels of one row of each: it
// reads the 4 modulated channels of R and writes 4 channels of C, 16 bytes each.  The (rows, 2E) projection never reaches
// memory and the separate modulation pass (3 reads + 1 write of a row) is gone.
template <int WM, int WN, int TM, int TN, bool STATS = false>
__device__ __forceinline__ void gemm_epilogue_adaln(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0_, int n0_, int wm, int wn,
                                                    int lane, float* tile /* wave private, EPI_WAVE_FLOATS */, float* wg) {
  if constexpr (TN % 2 == 0) {
    constexpr int BM = WM * TM * 32;
    const bool emit = STATS && p.stats_out != nullptr;          // uniform: (sum, sum of squares) of the modulated rows for the next LayerNorm
    float* wg_out = wg + 2 * BM;
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pss[4] = {0.f, 0.f, 0.f, 0.f};
    const int m0 = __builtin_amdgcn_readfirstlane(m0_), n0 = __builtin_amdgcn_readfirstlane(n0_);
    const int rows_left = p.M - m0;
    float* __restrict__ Cb = p.C + (int64_t)m0 * p.ldc + n0 / 2;
    const float* __restrict__ Rb = p.R + (int64_t)m0 * p.ldr + n0 / 2;
    const unsigned ldc = (unsigned)p.ldc, ldr = (unsigned)p.ldr;
    const int r = lane & 31, h = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    constexpr int NP = TM * TN / 2;                      // (scale, shift) fragment pairs of the wave, f = i * (TN / 2) + jp
    f32x4 res[2][4];
    auto load_res = [&](int f, f32x4 (&dst)[4]) __attribute__((always_inline)) {
      const int i = f / (TN / 2), jp = f % (TN / 2);
      const unsigned ch = (unsigned)((wn * (TN / 2) + jp) * 32 + c4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned row = (unsigned)((wm * TM + i) * 32 + rr + 8 * q);
        dst[q] = (int)row < rows_left ? *reinterpret_cast<const f32x4*>(Rb + row * ldr + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    load_res(0, res[0]);
    if constexpr (NP > 1) load_res(1, res[1]);
#pragma unroll
    for (int f = 0; f < NP; ++f) {
      const int i = f / (TN / 2), jp = f % (TN / 2);
      const unsigned ch = (unsigned)((wn * (TN / 2) + jp) * 32 + c4);
      f32x4 bs = {0.f, 0.f, 0.f, 0.f}, bh = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        bs = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * TN + 2 * jp) * 32 + c4);
        bh = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * TN + 2 * jp + 1) * 32 + c4);
      }
      f32x4 sc[4];
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = acc[i][2 * jp][e];
#pragma unroll
      for (int q = 0; q < 4; ++q) sc[q] = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4) + bs;
#pragma unroll
      for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_PITCH + r] = acc[i][2 * jp + 1][e];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned row = (unsigned)((wm * TM + i) * 32 + rr + 8 * q);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(tile + (rr + 8 * q) * EPI_PITCH + c4) + bh;
        const f32x4 o = res[f & 1][q] * sc[q] + sh;
        if ((int)row < rows_left) *reinterpret_cast<f32x4*>(Cb + row * ldc + ch) = o;
        if (emit) {
          if (jp == 0) { ps[q] = 0.f; pss[q] = 0.f; }
          ps[q] += (o.x + o.y) + (o.z + o.w);
          pss[q] += __builtin_fmaf(o.x, o.x, o.y * o.y) + __builtin_fmaf(o.z, o.z, o.w * o.w);
        }
      }
      if (emit && jp == TN / 2 - 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float s1 = group_sum<8>(ps[q]), s2 = group_sum<8>(pss[q]);
          if ((lane & 7) == 0) {
            float* o = wg_out + 2 * (wn * BM + (wm * TM + i) * 32 + rr + 8 * q);
            o[0] = s1; o[1] = s2;
          }
        }
      }
      if (f + 2 < NP) load_res(f + 2, res[f & 1]);
    }
    if (emit) stats_flush<WM, WN, TM, TN>(p, wg_out, m0, n0 / 2, p.N / 2, (wm * WN + wn) * 64 + lane, WN * TN * 16);
  } else {
    __builtin_trap();                                    // the launcher only sends G_ADALN to tiles with an even TN
  }
}

// ---- epilogue with a fused channel LayerNorm (blocks.py:125-131) ---------------------------------------------
// The workgroup tile spans all N = WN * TN * 32 output channels of its BM = TM * 32 rows (WM = 1).  Row statistics
// are taken in two passes like the reference (mean, then the mean of squared deviations): every lane adds its TN
// column tiles, the 32 lanes x WN waves of a row are summed through LDS (`red`: [WN][BM][36] floats, `stat`: [BM]).
constexpr int LN_PITCH = 36;      // floats per (wave, row) line: 16-byte reads of 4 lanes of a quad hit distinct banks

template <int WN, int TM, int TN>
__device__ __forceinline__ void ln_row_reduce(float (&ps)[TM][16], float scale, bool rsqrt_eps, float* red, float* stat, int wn,
                                              int lane) {   // in: per-lane partial sums; out (in place): the row statistic
  constexpr int BM = TM * 32, NT = WN * 64;
  const int r = lane & 31, h = lane >> 5, tid = wn * 64 + lane;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(wn * BM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LN_PITCH + r] = ps[i][e];
  __syncthreads();
  for (int row = tid >> 2; row < BM; row += NT / 4) {            // four lanes per row, 8 of the 32 columns each
    const int part = tid & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WN; ++w) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(red + (w * BM + row) * LN_PITCH + part * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(red + (w * BM + row) * LN_PITCH + part * 8 + 4);
      s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
    }
    s += dpp_zero<DPP_XOR1>(s);
    s += dpp_zero<DPP_XOR2>(s);
    if (part == 0) stat[row] = rsqrt_eps ? 1.0f / sqrtf(s * scale + 1e-5f) : s * scale;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(stat + i * 32 + 8 * e4 + 4 * h);   // rows (e & 3) = 0..3
      ps[i][4 * e4 + 0] = t.x; ps[i][4 * e4 + 1] = t.y; ps[i][4 * e4 + 2] = t.z; ps[i][4 * e4 + 3] = t.w;
    }
}

template <int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_ln(const GemmArgs& p, f32x16 (&acc)[TM][TN], int m0_, int wn, int lane, float* lds) {
  constexpr int BM = TM * 32, N = WN * TN * 32;
  float* red = lds;
  float* stat = lds + WN * BM * LN_PITCH;
  const int r = lane & 31, h = lane >> 5;
  const int flags = p.flags;
  const int m0 = __builtin_amdgcn_readfirstlane(m0_);
  const int rows_left = p.M - m0;
  const bool full = rows_left >= BM;
  const unsigned ldc = (unsigned)p.ldc, ldr = (unsigned)p.ldr, ldy = (unsigned)p.ldy;
  float* __restrict__ Cb = p.C ? p.C + (int64_t)m0 * p.ldc : nullptr;
  const float* __restrict__ Rb = (flags & G_RES) ? p.R + (int64_t)m0 * p.ldr : nullptr;
  const uint8_t* __restrict__ Mb = ((flags & G_RES) && (flags & (G_RES_MASK | G_OUT_MASK))) ? p.rowmask + m0 : nullptr;
  // 1. the ordinary epilogue value, kept in acc (and written to C if asked for)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned row0 = (unsigned)(i * 32 + 4 * h);
    float mk[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) mk[e] = 1.f;
    if (Mb) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
        if (full || (int)row < rows_left) mk[e] = Mb[row] ? 1.f : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned col = (unsigned)((wn * TN + j) * 32 + r);
      const float bias = p.bias ? p.bias[col] : 0.f;
      const float ls = (Rb && p.ls) ? p.ls[col] : 1.f;
      float res[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) res[e] = 0.f;
      if (Rb) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
          if (full || (int)row < rows_left) res[e] = Rb[row * ldr + col];
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
        float v = acc[i][j][e] + bias;
        if (flags & G_GELU) v = gelu_erf(v);
        if (flags & G_RELU) v = fmaxf(v, 0.f);
        if (Rb) {
          if (flags & G_OUT_MASK) v *= mk[e];
          float r_ = res[e];
          if (flags & G_RES_MASK) r_ *= mk[e];
          v = r_ + ls * v;
        }
        acc[i][j][e] = v;
        if (Cb && (full || (int)row < rows_left)) Cb[row * ldc + col] = v;
      }
    }
  }
  // 2. mean (acc becomes the deviation from it), 3. 1 / sqrt(var + eps); one 32-float array is reused throughout
  float ps[TM][16];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) s += acc[i][j][e];
      ps[i][e] = s;
    }
  ln_row_reduce<WN, TM, TN>(ps, 1.0f / N, false, red, stat, wn, lane);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float mean = ps[i][e];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) { const float d = acc[i][j][e] - mean; acc[i][j][e] = d; s += d * d; }
      ps[i][e] = s;
    }
  ln_row_reduce<WN, TM, TN>(ps, 1.0f / N, true, red, stat, wn, lane);
  // 4. affine (+ ReLU, + pe * mask) and store
  float* __restrict__ Yb = p.Y + (int64_t)m0 * p.ldy;
  const int t0 = p.ln_pe ? m0 % p.ln_T : 0;            // BM <= ln_T (checked by the launcher)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned row0 = (unsigned)(i * 32 + 4 * h);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned col = (unsigned)((wn * TN + j) * 32 + r);
      const float w = p.ln_w[col], b = p.ln_b[col];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const unsigned row = row0 + (e & 3) + 8 * (e >> 2);
        if (full || (int)row < rows_left) {
          float y = acc[i][j][e] * ps[i][e] * w + b;
          if (p.ln_relu) y = fmaxf(y, 0.f);
          if (p.ln_pe && p.ln_mask[m0 + row]) {
            int t = t0 + (int)row;                     // (m0 + row) % ln_T with one scalar modulo per tile
            if (t >= p.ln_T) t -= p.ln_T;
            y += p.ln_pe[(int64_t)t * N + col];
          }
          Yb[row * ldy + col] = y;
        }
      }
    }
  }
}

}  // namespace dcf
