// The attention half of a fusion layer (TransformerDecoder, blocks.py:632-646) as one kernel: see dec_chain.h.
//
// Formulation (ffn_chain.hip / head_chain.hip): every product runs TRANSPOSED on v_mfma_f32_32x32x16_f16, Y^T = W X^T, so a lane
// owns a ROW of the sequence -- lane (r = lane & 31, h = lane >> 5) of wave w holds row 32 w + r of the workgroup's 128-row window
// and, of every 32-channel tile ot, the 16 channels 32 ot + (e & 3) + 8 (e >> 2) + 4 h (e = 0 .. 15): the MFMA D layout.  The B
// operand of K step kk = 2 ot + q is then the lane's registers [8 q, 8 q + 8) of tile ot, provided the A fragments enumerate the 16
// channels of a K step in that order ("chain order", launch_split_chain1): the output of one product is the input of the next with
// no data movement.  Per row everything else is per lane: LayerNorm = a sum over the lane's 128 values + one exchange with the
// other lane half (v_permlane32_swap), the depthwise k3 convolution = wave_shr:1 / wave_shl:1 lane shifts (the rows at the wave
// boundaries come through LDS, the two rows next to the workgroup's window are loaded and normalised by waves 0 and 3), the softmax
// over the <= 64 keys of the row = the same reduction over the S^T accumulators.  The chain per head:
//   Q_h^T = Wq_h qc^T            (2 tiles x 16 K steps)      qc planes resident (128 registers)
//   S^T   = K_h Q_h^T            (lk2 key tiles x 4 K steps)  keys on the accumulator rows: softmax in registers
//   O_h^T = V_h^T P^T            (2 tiles x 2 lk2 K steps)    -> fp16 planes of ctx (128 registers once all heads are done)
// then per 32-channel block j:  (scale_j, shift_j)^T = Wp_j ctx^T (2 tiles x 16 K steps), q3 = adaln(q) scale + shift, stored.
// All A fragments (weights AND the text's K / V^T) stream through a two-buffer LDS ring by LDS-DMA, one barrier per stage:
// 4 x (Wq_h 64 KiB, K_h | V_h 16 lk2 KiB), 8 x Wp_j 64 KiB.  One wave per SIMD (512 registers), 128 rows per workgroup.
#include "dec_chain.h"

#include <type_traits>

#include "chain_common.h"
#include "common.h"

namespace dcf {

namespace {

using namespace chain;

constexpr int DE = 256, DHEADS = 4, DHD = 64;
constexpr int STAGE = 65536;                   // bytes per ring buffer (64 pieces of 1 KiB)
constexpr int WGROWS = 128;
// LDS behind the ring (floats)
constexpr int P_LNW = 0, P_LNB = 256, P_QNW = 512, P_QNB = 768, P_DW = 1024, P_BQ = 1792, P_BP = 2048, P_MS = 2560, P_END = 2624;
constexpr int X_LAST = P_END;                  // [5][256]: last[0] = the row before the window, last[w + 1] = row 31 of wave w
constexpr int X_FIRST = X_LAST + 5 * 256;      // [5][256]: first[w] = row 0 of wave w, first[4] = the row behind the window
constexpr int LDS_FLOATS = X_FIRST + 5 * 256;
constexpr int LDS_BYTES = 2 * STAGE + LDS_FLOATS * (int)sizeof(float);
#ifdef DCF_DC_STAMP
__device__ unsigned long long dcf_dc_stamps[16];   // diagnostic build only (tools/dc_stamp.sh): wave 0 of workgroup 1
#endif
}  // namespace

// ---- images ---------------------------------------------------------------------------------------------------------------------
__global__ void k_split_chain1(const float* __restrict__ W, unsigned short* __restrict__ img, int N, int K, unsigned* __restrict__ overflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (n, k pair)
  if (i >= N * (K / 2)) return;
  const int n = i / (K / 2), k = (i - n * (K / 2)) * 2;
  const float w0 = W[(size_t)n * K + k], w1 = W[(size_t)n * K + k + 1];
  unsigned hi, lo;
  split2_f16(w0, w1, SW, hi, lo);
  if (!(__builtin_fabsf(w0) * SW <= 65504.f) || !(__builtin_fabsf(w1) * SW <= 65504.f)) {
    if (overflow) atomicOr(overflow, 1u);
  }
  const int n32 = n >> 5, rr = n & 31, kk = k >> 4, kr = k & 15;
  const int a = kr >> 3, h = (kr >> 2) & 1, ii = kr & 3, j = 4 * a + ii;
  const size_t o = ((size_t)(n32 * (K / 16) + kk) * 2) * 512 + (size_t)(h * 32 + rr) * 8 + j;
  *reinterpret_cast<unsigned*>(img + o) = hi;
  *reinterpret_cast<unsigned*>(img + o + 512) = lo;
}

// per (query b, head hd): 8 lk2 pieces of K ((kt, ks, plane): lane (h, r), half j = K[32 kt + r][64 hd + 16 ks + 8 (j >> 2) + 4 h +
// (j & 3)] d^-1/4), then 8 lk2 pieces of V^T ((ct, kt, q, plane): lane (h, r), half j = V[32 kt + 16 q + 8 (j >> 2) + 4 h + (j & 3)]
// [64 hd + 32 ct + r]); keys beyond the text are zero.  Unscaled fp16 hi / lo planes as in k_xattn_mfma (attn.hip).
__global__ void k_kv_image(const float* __restrict__ K, const float* __restrict__ V, const uint8_t* __restrict__ kvmask, int B, int Lk,
                           int lk2, unsigned short* __restrict__ img, float* __restrict__ kmask) {
  const int per_head = 16 * lk2 * 256;                            // (piece, lane, pair) triples per (b, head)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 64) {
    const int b = i / 64, key = i % 64;
    kmask[i] = (key < Lk && kvmask[(size_t)b * Lk + key]) ? 0.f : -INFINITY;
  }
  if (i >= B * DHEADS * per_head) return;
  const int b = i / (DHEADS * per_head), rem = i - b * (DHEADS * per_head);
  const int hd = rem / per_head, rem2 = rem - hd * per_head;
  const int piece = rem2 >> 8, lane = (rem2 >> 2) & 63, jp = rem2 & 3;
  const int h = lane >> 5, r = lane & 31, plane = piece & 1, j = 2 * jp;
  const float scale = 1.0f / sqrtf(sqrtf((float)DHD));
  float v0 = 0.f, v1 = 0.f;
  if (piece < 8 * lk2) {
    const int ks = (piece >> 1) & 3, kt = piece >> 3;
    const int key = 32 * kt + r, ch = DHD * hd + 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
    if (key < Lk) { v0 = K[((size_t)b * Lk + key) * DE + ch] * scale; v1 = K[((size_t)b * Lk + key) * DE + ch + 1] * scale; }
  } else {
    const int pv = piece - 8 * lk2;
    const int q = (pv >> 1) & 1, kt = (pv >> 2) % lk2, ct = (pv >> 2) / lk2;
    const int ch = DHD * hd + 32 * ct + r, key = 32 * kt + 16 * q + 8 * (j >> 2) + 4 * h + (j & 3);
    if (key < Lk) v0 = V[((size_t)b * Lk + key) * DE + ch];
    if (key + 1 < Lk) v1 = V[((size_t)b * Lk + key + 1) * DE + ch];
  }
  const f16x2 hh = __builtin_convertvector(f32x2{v0, v1}, f16x2);
  const f16x2 ll = __builtin_convertvector(f32x2{v0 - (float)hh[0], v1 - (float)hh[1]}, f16x2);
  const size_t o = (((size_t)b * DHEADS + hd) * 16 * lk2 + piece) * 512 + (size_t)lane * 8 + j;
  *reinterpret_cast<unsigned*>(img + o) = __builtin_bit_cast(unsigned, plane == 0 ? hh : ll);
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------------
template <int LK2, bool SINGLE = false>
__global__ __launch_bounds__(256, 1) void k_dec_chain(DecChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* ldf = reinterpret_cast<float*>(lds + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wins = (p.T + WGROWS - 1) / WGROWS;                 // windows per sequence
  const int b = (int)blockIdx.x / wins, t0 = ((int)blockIdx.x - b * wins) * WGROWS;
  const int t = t0 + w * 32 + r;                                 // this lane's position in sequence b
  const bool inseq = t < p.T;
  const int64_t row = (int64_t)b * p.T + (inseq ? t : p.T - 1);
  constexpr int KVP = 16 * LK2;                                  // pieces of a (K_h | V_h) stage

  // ---- the weight stream: stage s = 2 hd (Wq_hd), 2 hd + 1 (K_hd | V_hd) for hd < 4, then 8 + j (Wp_j); buffer = s & 1
  const unsigned short* kv_b = p.KV + (size_t)b * DHEADS * KVP * 512;
  auto issue_piece = [&](int s, int i) __attribute__((always_inline)) {     // piece w + 4 i of stage s (s < 16)
    const bool kv = s < 8 && (s & 1);
    const int np = kv ? KVP : 64;
    int pc = w + 4 * i;
    pc = pc < np ? pc : np - 1;
    const unsigned short* src = s >= 8 ? p.Wp + (size_t)(s - 8) * 64 * 512 : (kv ? kv_b + (size_t)(s >> 1) * KVP * 512 : p.Wq + (size_t)(s >> 1) * 64 * 512);
    glds16(src + (size_t)pc * 512, lane16, (unsigned)(s & 1) * STAGE + (unsigned)pc * 1024u);
  };
#ifdef DCF_DC_STAMP
  unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = dc_stamp();
#endif

  // ---- the lane's row: 128 channels in D layout, xv[4 ot + g] = channels 32 ot + 8 g + 4 h .. + 3
  const float* px = p.X + row * p.ldx + 4 * h;
  const bool valid = inseq && p.mask[row] != 0;
  f32x4 xv[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) xv[i] = *reinterpret_cast<const f32x4*>(px + 32 * (i >> 2) + 8 * (i & 3));
  // the two rows next to the window (waves 0 and 3): lane l takes channels 4 l .. 4 l + 3 of the whole row
  const int te = w == 0 ? t0 - 1 : t0 + WGROWS;
  const bool edge_wave = w == 0 || w == 3;
  const bool evalid = edge_wave && te >= 0 && te < p.T && p.mask[(int64_t)b * p.T + (te >= 0 && te < p.T ? te : 0)] != 0;
  f32x4 ev = f32x4{0.f, 0.f, 0.f, 0.f};
  if (evalid) ev = *reinterpret_cast<const f32x4*>(p.X + ((int64_t)b * p.T + te) * p.ldx + 4 * lane);
  // (the rows first: their latency is the kernel's start-up; the first stage of the weight stream has the whole front end to land)
#pragma unroll
  for (int i = 0; i < 16; ++i) issue_piece(0, i);

  // ---- per-channel parameters -> LDS
  {
    ldf[P_LNW + tid] = p.ln_q_w[tid]; ldf[P_LNB + tid] = p.ln_q_b[tid];
    ldf[P_QNW + tid] = p.qn_w[tid]; ldf[P_QNB + tid] = p.qn_b[tid];
    ldf[P_DW + tid] = p.dw[tid]; ldf[P_DW + 256 + tid] = p.dw[256 + tid]; ldf[P_DW + 512 + tid] = p.dw[512 + tid];
    ldf[P_BQ + tid] = p.bq[tid];
    ldf[P_BP + tid] = p.bp[tid]; ldf[P_BP + 256 + tid] = p.bp[256 + tid];
    if (tid < 64) ldf[P_MS + tid] = p.kmask[(size_t)b * 64 + tid];
  }


  if (!valid) {
#pragma unroll
    for (int i = 0; i < 32; ++i) xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float mean1, rstd1;
  row_stats<DE>(xv, mean1, rstd1);
  STAMP(0);
  __syncthreads();                                               // parameters are in LDS
  // xq = ln_xattn_q(q) * mask, in place
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int c = 32 * (i >> 2) + 8 * (i & 3) + 4 * h;
    const f32x4 g = *reinterpret_cast<const f32x4*>(ldf + P_LNW + c), bb = *reinterpret_cast<const f32x4*>(ldf + P_LNB + c);
    f32x4 y = (xv[i] - mean1) * rstd1 * g + bb;
    if (!valid) y = f32x4{0.f, 0.f, 0.f, 0.f};
    xv[i] = y;
  }
  // boundary rows of the wave -> LDS; the window's outer neighbours from waves 0 / 3
  if (r == 31) {
#pragma unroll
    for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4*>(ldf + X_LAST + (w + 1) * 256 + 32 * (i >> 2) + 8 * (i & 3) + 4 * h) = xv[i];
  }
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4*>(ldf + X_FIRST + w * 256 + 32 * (i >> 2) + 8 * (i & 3) + 4 * h) = xv[i];
  }
  if (edge_wave) {
    f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
    const float s = wave_sum((ev.x + ev.y) + (ev.z + ev.w));
    const float mean = s * (1.0f / DE);
    const f32x4 d = ev - mean;
    const float var = wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / DE);
    const float rs = 1.0f / sqrtf(var + 1e-5f);
    if (evalid) y = d * rs * *reinterpret_cast<const f32x4*>(ldf + P_LNW + 4 * lane) + *reinterpret_cast<const f32x4*>(ldf + P_LNB + 4 * lane);
    *reinterpret_cast<f32x4*>(ldf + (w == 0 ? X_LAST : X_FIRST + 4 * 256) + 4 * lane) = y;
  }
  STAMP(1);
  __syncthreads();
  STAMP(2);
  // depthwise k3 convolution along the rows (MaskedConv1D: the inputs are already masked), in place
  {
    const bool is32 = lane == 32, is31 = lane == 31;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = 32 * (i >> 2) + 8 * (i & 3) + 4 * h;
      const f32x4 lastv = *reinterpret_cast<const f32x4*>(ldf + X_LAST + w * 256 + c);
      const f32x4 firstv = *reinterpret_cast<const f32x4*>(ldf + X_FIRST + (w + 1) * 256 + c);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(ldf + P_DW + c), w1 = *reinterpret_cast<const f32x4*>(ldf + P_DW + 256 + c),
                  w2 = *reinterpret_cast<const f32x4*>(ldf + P_DW + 512 + c);
      f32x4 pv, nv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {                              // (the lane shifts with every lane active)
        const float sp = shr1(xv[i][e], lastv[e]), sn = shl1(xv[i][e], firstv[e]);
        pv[e] = is32 ? lastv[e] : sp;
        nv[e] = is31 ? firstv[e] : sn;
      }
      xv[i] = w0 * pv + w1 * xv[i] + w2 * nv;
    }
  }
  // qc = q_norm(conv) as the B operand of the query projection: K step kk = 2 ot + q <- xv[4 ot + 2 q], xv[4 ot + 2 q + 1]
  f16x8 qh[16], ql[16];
  {
    float mean2, rstd2;
    row_stats<DE>(xv, mean2, rstd2);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float v8[8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = 2 * kk + u, c = 32 * (i >> 2) + 8 * (i & 3) + 4 * h;
        const f32x4 g = *reinterpret_cast<const f32x4*>(ldf + P_QNW + c), bb = *reinterpret_cast<const f32x4*>(ldf + P_QNB + c);
        const f32x4 y = (xv[i] - mean2) * rstd2 * g + bb;
        v8[4 * u] = y.x; v8[4 * u + 1] = y.y; v8[4 * u + 2] = y.z; v8[4 * u + 3] = y.w;
      }
      split8(v8, SA, qh[kk], ql[kk]);
    }
  }

  STAMP(3);
  // ---- heads: Q_h, softmax(K_h Q_h^T), V_h^T P^T -> ctx planes
  f16x8 cth[16], ctl[16];
  const float qscale = 1.0f / sqrtf(sqrtf((float)DHD));         // d^-1/4 on q (and on k, in the image): blocks.py:179, :379
  auto stage_begin = [&](int s) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of stage s have landed ...
    STAMP(4);
    __syncthreads();                                             // ... everybody's have, and nobody reads the other buffer any more
    STAMP(5);
  };
  // Two 32-row output tiles over 16 K steps: acc[t2] += A(t2, kk) B[kk], three products per step (lo x hi, hi x lo, hi x hi), the
  // two tiles alternating.  The fragments of step kk + 1 are read while step kk computes; `dma(kk)` requests pieces of the next
  // stage at the start of the step; `side(kk, slot)` is vector work of ANOTHER part of the chain (the previous head's context
  // split, the previous block's epilogue) placed between the MFMAs: a wave issues in order, so vector work hides behind MFMAs only
  // where it sits between them in program order (the compiler left to itself puts it behind the last one: ffn_chain.hip).
  auto gemm2 = [&](const unsigned char* buf, const f16x8 (&bh)[16], const f16x8 (&bl)[16], f32x16 (&acc)[2], auto&& dma, auto&& side)
                   __attribute__((always_inline)) {
    f16x8 fr[2][4];
    auto frags = [&](int kk, int set) __attribute__((always_inline)) {
      fr[set][0] = *reinterpret_cast<const f16x8*>(buf + ((0 * 16 + kk) * 2) * 1024);
      fr[set][1] = *reinterpret_cast<const f16x8*>(buf + ((0 * 16 + kk) * 2 + 1) * 1024);
      fr[set][2] = *reinterpret_cast<const f16x8*>(buf + ((1 * 16 + kk) * 2) * 1024);
      fr[set][3] = *reinterpret_cast<const f16x8*>(buf + ((1 * 16 + kk) * 2 + 1) * 1024);
    };
    frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int set = kk & 1;
      if (kk + 1 < 16) frags(kk + 1, set ^ 1);
      acc[0] = mma(fr[set][1], bh[kk], acc[0]);
      acc[1] = mma(fr[set][3], bh[kk], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      side(kk, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mma(fr[set][0], bl[kk], acc[0]);
      acc[1] = mma(fr[set][2], bl[kk], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
      side(kk, 1);
      dma(kk);                                                   // (behind the step's fragment reads; 258 against 264 us in front of them)
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mma(fr[set][0], bh[kk], acc[0]);
      acc[1] = mma(fr[set][2], bh[kk], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the context of a head (O, two 32-channel tiles in D layout) -> planes of K steps 2 (2 hd + ct) + q of the projection, one pair
  // of values per call: piece u = 0 .. 15 <-> (ct, q, i) = (u >> 3, (u >> 2) & 1, u & 3)
  f32x16 O[2];
  u32x4 cx_h, cx_l;
  auto ctx_piece = [&](int hd, int u) __attribute__((always_inline)) {
    const int ct = u >> 3, q = (u >> 2) & 1, i = u & 3;
    unsigned hi, lo;
    split2_f16(O[ct][8 * q + 2 * i], O[ct][8 * q + 2 * i + 1], SA, hi, lo);
    cx_h[i] = hi; cx_l[i] = lo;
    if (i == 3) {
      cth[2 * (2 * hd + ct) + q] = __builtin_bit_cast(f16x8, cx_h);
      ctl[2 * (2 * hd + ct) + q] = __builtin_bit_cast(f16x8, cx_l);
    }
  };
#pragma unroll
  for (int hd = 0; hd < DHEADS; ++hd) {
    // -- stage 2 hd: Q_h^T = Wq_h qc^T; beside its MFMAs the context split of the previous head
    f32x16 QA[2];
    {
      stage_begin(2 * hd);
      const unsigned char* buf = lds + ((2 * hd) & 1) * STAGE + lane16;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int e = 0; e < 16; ++e) QA[t2][e] = 0.f;
      gemm2(buf, qh, ql, QA,
            [&](int kk) __attribute__((always_inline)) {                                  // the next stage (K_h | V_h)
              if (kk < KVP / 4) issue_piece(2 * hd + 1, kk);
            },
            [&](int kk, int slot) __attribute__((always_inline)) { if (hd > 0 && slot == 0) ctx_piece(hd - 1, kk); });
    }
    STAMP(6);
    // q_h * d^-1/4 as unscaled fp16 planes (attn.hip): K step ks = 2 t2 + q <- QA[t2][8 q .. 8 q + 7]
    f16x8 sh_[4], sl_[4];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int g = 2 * q + u;
          const f32x4 bq = *reinterpret_cast<const f32x4*>(ldf + P_BQ + DHD * hd + 32 * t2 + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v8[4 * u + e] = __builtin_fmaf(QA[t2][4 * g + e], UNSCALE, bq[e]) * qscale;
        }
        split8(v8, 1.f, sh_[2 * t2 + q], sl_[2 * t2 + q]);
      }
    STAMP(7);
    // -- stage 2 hd + 1: S^T = K_h Q_h^T, softmax over the keys, O_h^T = V_h^T P^T
    {
      stage_begin(2 * hd + 1);
      const unsigned char* buf = lds + ((2 * hd + 1) & 1) * STAGE + lane16;
      f32x16 S[LK2];
#pragma unroll
      for (int kt = 0; kt < LK2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) S[kt][e] = 0.f;
      f16x8 kf[2][LK2][2];
      auto kfrags = [&](int ks, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < LK2; ++kt) {
          kf[set][kt][0] = *reinterpret_cast<const f16x8*>(buf + ((kt * 4 + ks) * 2) * 1024);
          kf[set][kt][1] = *reinterpret_cast<const f16x8*>(buf + ((kt * 4 + ks) * 2 + 1) * 1024);
        }
      };
      kfrags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int set = ks & 1;
        if (ks + 1 < 4) kfrags(ks + 1, set ^ 1);
        // the next stage (Wq of the next head, or the first projection block): 16 pieces per wave, two per K step here, the rest
        // beside the second product
        issue_piece(2 * hd + 2, 2 * ks);
        issue_piece(2 * hd + 2, 2 * ks + 1);
        if constexpr (!SINGLE) {                       // (dcf_config::attn_mode 1 keeps the hi x hi product alone: its own instantiation)
#pragma unroll
          for (int kt = 0; kt < LK2; ++kt) S[kt] = mma(kf[set][kt][0], sl_[ks], S[kt]);
#pragma unroll
          for (int kt = 0; kt < LK2; ++kt) S[kt] = mma(kf[set][kt][1], sh_[ks], S[kt]);
        }
#pragma unroll
        for (int kt = 0; kt < LK2; ++kt) S[kt] = mma(kf[set][kt][0], sh_[ks], S[kt]);
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP(8);
      // the first fragments of the second product are on their way while the softmax runs
      f16x8 vf[2][2][2];
      auto vfrags = [&](int st_, int set) __attribute__((always_inline)) {          // st_ = kt * 2 + q
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          const int pc = 8 * LK2 + ((ct * LK2 + (st_ >> 1)) * 2 + (st_ & 1)) * 2;
          vf[set][ct][0] = *reinterpret_cast<const f16x8*>(buf + pc * 1024);
          vf[set][ct][1] = *reinterpret_cast<const f16x8*>(buf + (pc + 1) * 1024);
        }
      };
      vfrags(0, 0);
      // softmax over the keys of the row: slot e of lane half h = key 32 kt + (e & 3) + 8 (e >> 2) + 4 h
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < LK2; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 mk = *reinterpret_cast<const f32x4*>(ldf + P_MS + 32 * kt + 8 * g + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) { S[kt][4 * g + e] += mk[e]; mx = fmaxf(mx, S[kt][4 * g + e]); }
        }
      mx = xor32_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < LK2; ++kt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { S[kt][e] = fast_exp(S[kt][e] - mx); sum += S[kt][e]; }
      const float inv = 1.0f / xor32_sum(sum);                   // all keys masked: NaN row, as the reference
      f16x8 ph[LK2][2], pl[LK2][2];
#pragma unroll
      for (int kt = 0; kt < LK2; ++kt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] = S[kt][8 * q + e] * inv;
          split8(v8, 1.f, ph[kt][q], pl[kt][q]);
        }
      STAMP(9);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[ct][e] = 0.f;
#pragma unroll
      for (int st_ = 0; st_ < 2 * LK2; ++st_) {
        const int set = st_ & 1, kt = st_ >> 1, q = st_ & 1;
        if (st_ + 1 < 2 * LK2) vfrags(st_ + 1, set ^ 1);
        // (the rest of the next stage's pieces)
#pragma unroll
        for (int i = 0; i < 8 / (2 * LK2); ++i) issue_piece(2 * hd + 2, 8 + st_ * (8 / (2 * LK2)) + i);
        if constexpr (!SINGLE) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vf[set][ct][0], pl[kt][q], O[ct]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vf[set][ct][1], ph[kt][q], O[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vf[set][ct][0], ph[kt][q], O[ct]);
        __builtin_amdgcn_sched_barrier(0);
      }
      STAMP(10);
    }
  }

  // ---- projection blocks: (scale_j, shift_j) = Wp_j ctx, q3 = Xa * scale + shift.  The context split of the last head rides in
  // the first eight K steps of block 0 (its planes are K steps 12 .. 15); the epilogue of block j - 1 beside the MFMAs of block j.
  float ps = 0.f, pss = 0.f;
  float* qout = p.Q3 + row * p.ldq + 4 * h;
  f32x16 A2[2][2];                                               // [block parity][scale, shift]
  f32x4 xr[2][4];                                                // the rows' own channels of a block again (L2): Xa is not kept in registers
  auto load_xr = [&](int j) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) xr[j & 1][g] = *reinterpret_cast<const f32x4*>(px + 32 * j + 8 * g);
  };
  f32x4 ep_sc, ep_sf;                                            // (scale, shift) of the group whose first half has run
  auto epilogue = [&](int j, int g, int half) __attribute__((always_inline)) {   // half 0: scale / shift; half 1: modulate, store, statistics
    if (half == 0) {
      const f32x4 bs = *reinterpret_cast<const f32x4*>(ldf + P_BP + 64 * j + 8 * g + 4 * h);
      const f32x4 bh = *reinterpret_cast<const f32x4*>(ldf + P_BP + 64 * j + 32 + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ep_sc[e] = __builtin_fmaf(A2[j & 1][0][4 * g + e], UNSCALE, bs[e]);
        ep_sf[e] = __builtin_fmaf(A2[j & 1][1][4 * g + e], UNSCALE, bh[e]);
      }
    } else {
      f32x4 xa = valid ? xr[j & 1][g] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (!p.affine) xa = (xa - mean1) * rstd1;                  // adaln: LayerNorm without affine (blocks.py:620-621)
      const f32x4 o = xa * ep_sc + ep_sf;
      if (inseq) *reinterpret_cast<f32x4*>(qout + 32 * j + 8 * g) = o;
      ps += (o.x + o.y) + (o.z + o.w);
      pss += __builtin_fmaf(o.x, o.x, o.y * o.y) + __builtin_fmaf(o.z, o.z, o.w * o.w);
    }
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    stage_begin(8 + j);
    const unsigned char* buf = lds + (j & 1) * STAGE + lane16;
    // xr of block j - 1 (requested a stage ago) has landed with the vmcnt(0) of stage_begin: tell the compiler, or its own wait at
    // the first use would also wait for the LDS-DMA pieces requested since (it does not count them, the hardware does, in order)
    if (j > 0) asm volatile("" : "+v"(xr[(j - 1) & 1][0]), "+v"(xr[(j - 1) & 1][1]), "+v"(xr[(j - 1) & 1][2]), "+v"(xr[(j - 1) & 1][3]));
    load_xr(j);                                                  // consumed by epilogue(j), beside the MFMAs of block j + 1
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int e = 0; e < 16; ++e) A2[j & 1][t2][e] = 0.f;
    if (j == 0) {
      // K steps 0 .. 11 first is the natural order: the planes of K steps 12 .. 15 are complete after the first eight steps
      gemm2(buf, cth, ctl, A2[0],
            [&](int kk) __attribute__((always_inline)) {
              issue_piece(9, kk);
            },
            [&](int kk, int slot) __attribute__((always_inline)) { if (kk < 8) ctx_piece(DHEADS - 1, 2 * kk + slot); });
    } else {
      gemm2(buf, cth, ctl, A2[j & 1],
            [&](int kk) __attribute__((always_inline)) {
              if (j + 1 < 8) issue_piece(8 + j + 1, kk);
            },
            [&](int kk, int slot) __attribute__((always_inline)) { if ((kk & 3) == 0) epilogue(j - 1, kk >> 2, slot); });
    }
    STAMP(11);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" : "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[1][2]), "+v"(xr[1][3]));
#pragma unroll
  for (int g = 0; g < 4; ++g) { epilogue(7, g, 0); epilogue(7, g, 1); }
  STAMP(12);
  const float s1 = xor32_sum(ps), s2 = xor32_sum(pss);
  if (p.stats_out && inseq && h == 0) {
    const int slots = DE / p.stats_w;
    float* o = p.stats_out + row * slots * 2;
    o[0] = s1; o[1] = s2;
    for (int k = 1; k < slots; ++k) { o[2 * k] = 0.f; o[2 * k + 1] = 0.f; }
  }
  // a non-finite accumulator anywhere in the chain (an operand left the fp16 range) makes the row's sum non-finite; rows whose
  // keys are all masked are NaN by definition (reference behaviour) and do not raise the flag
  const bool bad = !(__builtin_fabsf(s1) <= 3.4028234664e38f) && inseq;
  if (bad && p.status) {
    bool anykey = false;
#pragma unroll
    for (int k = 0; k < 64; ++k) anykey = anykey || ldf[P_MS + k] == 0.f;
    if (anykey) atomicOr(p.status, 1u);
  }
#ifdef DCF_DC_STAMP
  STAMP(13);
  if (blockIdx.x == 1 && tid == 0)
    for (int i = 0; i < 16; ++i) dcf_dc_stamps[i] = acc_[i];
#endif
}

#ifdef DCF_DC_STAMP
}  // namespace dcf
extern "C" int dcf_debug_dc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcf::dcf_dc_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace dcf {
#endif

bool dec_chain_supports(int E, int heads, int Lk) { return E == DE && heads == DHEADS && Lk >= 1 && Lk <= 64; }

size_t chain1_image_halfs(int N, int K) { return (size_t)N * K * 2; }

int launch_split_chain1(const float* W, unsigned short* img, int N, int K, hipStream_t stream, unsigned* overflow) {
  DCF_CHECK(N % 32 == 0 && K % 16 == 0, "launch_split_chain1: N = %d, K = %d", N, K);
  const int n = N * (K / 2);
  hipLaunchKernelGGL(k_split_chain1, dim3((n + 255) / 256), dim3(256), 0, stream, W, img, N, K, overflow);
  DCF_HIP(hipGetLastError());
  return 0;
}

size_t kv_image_halfs(int lk2) { return (size_t)DHEADS * 16 * lk2 * 512; }

int launch_kv_image(const float* K, const float* V, const uint8_t* kvmask, int B, int Lk, int lk2, unsigned short* img, float* kmask,
                    hipStream_t stream) {
  DCF_CHECK(B > 0 && Lk >= 1 && Lk <= 32 * lk2 && (lk2 == 1 || lk2 == 2), "launch_kv_image: Lk = %d, lk2 = %d", Lk, lk2);
  const int n = B * DHEADS * 16 * lk2 * 256;
  hipLaunchKernelGGL(k_kv_image, dim3((n + 255) / 256), dim3(256), 0, stream, K, V, kvmask, B, Lk, lk2, img, kmask);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_dec_chain(const DecChainArgs& a, hipStream_t stream) {
  DCF_CHECK(a.B > 0 && a.T > 0 && a.X && a.mask && a.Wq && a.KV && a.kmask && a.Wp && a.Q3, "launch_dec_chain: null argument");
  DCF_CHECK(a.lk2 == 1 || a.lk2 == 2, "launch_dec_chain: lk2 = %d", a.lk2);
  DCF_CHECK(!a.stats_out || (a.stats_w > 0 && DE % a.stats_w == 0), "launch_dec_chain: stats_out needs a slot width dividing %d", DE);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DCF_CHECK(al16(a.X) && al16(a.Q3) && al16(a.Wq) && al16(a.Wp) && al16(a.KV) && a.ldx % 4 == 0 && a.ldq % 4 == 0,
            "launch_dec_chain: operands must be 16-byte aligned with row pitches that are multiples of 4");
  static bool attr_set[64] = {};                         // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_chain<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dec_chain<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set[dev] = true;
  }
  const unsigned grid = (unsigned)(a.B * ((a.T + WGROWS - 1) / WGROWS));
  if (a.attn_single) {
    if (a.lk2 == 1) hipLaunchKernelGGL((k_dec_chain<1, true>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((k_dec_chain<2, true>), dim3(grid), dim3(256), LDS_BYTES, stream, a);
  } else if (a.lk2 == 1) hipLaunchKernelGGL(k_dec_chain<1>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(k_dec_chain<2>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
