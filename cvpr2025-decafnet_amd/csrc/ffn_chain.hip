// FFN as one kernel: fc (256 -> 1024), erf GELU, proj (1024 -> 256), residual / LayerScale / mask epilogue -- the hidden
// activations (4 KiB per row as fp32, written and read back by the GEMM pair: 48 KB per clip of the forward's HBM traffic)
// never leave the registers.  blocks.py:535-538, 589-590.
//
// Both GEMMs run TRANSPOSED on the matrix cores: H^T = W1 X^T and Y^T = W2 H^T.  With v_mfma_f32_32x32x16_f16 the D fragment
// of lane (r = lane & 31, h = lane >> 5) holds column r = one ROW of X, and the B operand of the next product wants exactly
// that: 8 consecutive k of column r per lane.  So a wave that owns 32 rows of X chains the two products without moving the
// hidden activations across lanes: the 16 accumulator values of a 32-wide hidden chunk become, after GELU and the fp16 split,
// the B operand of two K steps of the second product.  The only thing to arrange is WHICH hidden unit lands in which
// accumulator slot: D slot e of lane half h is output row i = (e & 3) + 8 (e >> 2) + 4 h, the B operand of K step c2 wants
// k = 16 c2 + 8 h + j in half j of lane half h -- feeding W1 row pi(i) as A row i with pi = "swap bits 2 and 3" makes slot
// e hold hidden unit 16 (e >> 3) + 8 h + (e & 7), i.e. k = e of the lane's half.  The A fragments come from the ordinary
// weight images (gemm_bf16s.hip, k_split_planes); pi is applied by the LDS read address (conflict free: a bijection mod 16).
//
// A wave keeps its 32 rows of X as fp16 planes (128 registers) and the 32 x 256 output accumulators (128 registers): one
// wave per SIMD, 512 registers, four waves = 128 rows per workgroup, one workgroup per CU.  The weights stream through LDS:
// per 32-wide hidden chunk 32 KiB of W1 rows and 32 KiB of W2 columns (hi / lo planes), fetched by LDS-DMA
// (global_load_lds_dwordx4: a 1 KiB piece is one fragment of all 64 lanes) one chunk ahead into a two-buffer ring, one
// barrier per chunk.  Per chunk a wave issues 96 MFMAs and 64 ds_read_b128; the GELU (16 values) of chunk t - 1 and the
// second product of chunk t - 2 run beside the first product of chunk t, so no MFMA waits for vector work of its own
// iteration.  Products and accumulation order are those of the GEMM pair (hi x lo, lo x hi, hi x hi per K step, K ascending).
#include "ffn_chain.h"

#include <type_traits>

#include "common.h"

namespace dcf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int FE = 256, FH = 1024, NCHUNK = FH / 32;
constexpr int BLK = 2 * 3 * 64 * 8;            // halfs per (32 rows, 32 k) block of a weight image: [chunk c][plane 0..2][lane][8]
constexpr int PIECE = 64 * 8;                  // halfs per (chunk, plane): one wave-wide fragment, 1 KiB
constexpr float SA = 16.f, UNSCALE = 1.f / 4096.f;   // the f16x3 scaling of gemm_bf16s.hip (activations 2^4, weights 2^8)
constexpr int STAGE = 65536;                   // bytes per ring buffer: 32 pieces of W1 (chunk t), 32 pieces of W2 (chunk t - 2)
constexpr int LDS_BYTES = 2 * STAGE + (2 * FH + 2 * FE) * (int)sizeof(float);

__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{x0 * SA, x1 * SA}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = __builtin_fmaf(x0, SA, -(float)h[0]), r1 = __builtin_fmaf(x1, SA, -(float)h[1]);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}

// one 1 KiB LDS-DMA piece: lane l copies the 16 bytes at sbase + voff (sbase wave-uniform: a scalar register pair, voff = 16 l: ONE
// vector register for every piece of the kernel) to LDS byte lds_dst + 16 l
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

#ifdef DCF_FFN_STAMP
// diagnostic build only (tools/ffn_stamp.sh): cycles wave 0 of workgroup 0 spends in the segments of an iteration
__device__ unsigned long long dcf_ffn_stamps[8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(i) do { const unsigned long long t_ = stamp(); acc_[i] += t_ - last_; last_ = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

}  // namespace

template <bool FOLD>
__global__ __launch_bounds__(256, 1) void k_ffn_chain(FfnChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* lds_s = reinterpret_cast<float*>(lds + 2 * STAGE);        // [1024] ln_s (FOLD), then [1024] b1
  float* lds_c = lds_s + FH;
  float* lds_b2 = lds_c + FH;                                      // [256] proj bias, [256] LayerScale (1 where there is none)
  float* lds_ls = lds_b2 + FE;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128;
  const int row = m0 + w * 32 + r;
  const int row_c = row < p.M ? row : p.M - 1;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned lds0 = 0u;        // the dynamic array is the kernel's only LDS object: it starts at LDS byte 0

  // stage t of the weight stream: W1 rows of chunk t (waves 0, 1) and W2 columns of chunk t - 2 (waves 2, 3), 16 pieces of 1 KiB
  // per wave.  Everything about a piece is wave-uniform scalar arithmetic and there is no branch: stages that do not exist
  // (W1 from t = 32 on, W2 before t = 2) fetch a chunk that does, into LDS nobody reads.
  const bool w1 = w < 2;
  const unsigned short* dma_base = w1 ? p.W1s + (size_t)w * 4 * BLK : p.W2s + (size_t)(w - 2) * 4 * (FH / 32) * BLK;
  const unsigned dma_chunk = w1 ? (unsigned)(FE / 32) * BLK : (unsigned)BLK;          // halfs from chunk to chunk
  const unsigned dma_tile = w1 ? (unsigned)BLK : (unsigned)(FH / 32) * BLK;          // halfs between the four 4-piece groups
  const unsigned dma_dst = lds0 + (w1 ? 0u : 32768u) + (unsigned)(w & 1) * 16384u;
  auto issue_piece = [&](int t, int i) __attribute__((always_inline)) {
    int ch = w1 ? t : t - 2;
    ch = ch < 0 ? 0 : (ch > NCHUNK - 1 ? NCHUNK - 1 : ch);
    const unsigned short* src = dma_base + (size_t)ch * dma_chunk + (size_t)(i >> 2) * dma_tile + (((i >> 1) & 1) * 3 + (i & 1)) * PIECE;
    glds16(src, lane16, dma_dst + (unsigned)(t & 1) * STAGE + (unsigned)i * 1024u);
  };
  auto issue = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) issue_piece(t, i);
  };
#ifdef DCF_FFN_STAMP
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = stamp();
#endif
  issue(0);

  // fc bias (and the folded LayerNorm's row sums) -> LDS
  {
    const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.b1 + tid * 4);
    *reinterpret_cast<f32x4*>(lds_c + tid * 4) = c4;
    if constexpr (FOLD) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + tid * 4);
      *reinterpret_cast<f32x4*>(lds_s + tid * 4) = s4;
    }
    lds_b2[tid] = p.b2[tid];
    lds_ls[tid] = p.ls ? p.ls[tid] : 1.f;
  }
  float mean = 0.f, rstd = 1.f;
  if constexpr (FOLD) {            // as stats_load of gemm_common.h
    const float* sp = p.stats + (int64_t)row_c * p.stats_slots * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < p.stats_slots; ++k) { s1 += sp[2 * k]; s2 += sp[2 * k + 1]; }
    const float inv = 1.0f / (float)FE;
    mean = s1 * inv;
    rstd = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.f) + 1e-5f);
  }

  // the wave's 32 rows of X as B operands: K step s = 16 k, lane (r, h) holds k = 16 s + 8 h .. + 7 of row r, two fp16 planes
  f16x8 xh[16], xl[16];
  {
    const float* px = p.X + (int64_t)row_c * p.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(px + 16 * s), v1 = *reinterpret_cast<const f32x4*>(px + 16 * s + 4);
      unsigned hi[4], lo[4];
      split2_f16(v0.x, v0.y, hi[0], lo[0]);
      split2_f16(v0.z, v0.w, hi[1], lo[1]);
      split2_f16(v1.x, v1.y, hi[2], lo[2]);
      split2_f16(v1.z, v1.w, hi[3], lo[3]);
      xh[s] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
      xl[s] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});

    }
  }

  STAMP(5);
  f32x16 Y[8];
#pragma unroll
  for (int ot = 0; ot < 8; ++ot)
#pragma unroll
    for (int e = 0; e < 16; ++e) Y[ot][e] = 0.f;
  f32x16 H0, H1;                   // first-product accumulators of even / odd chunks: iteration t fills one while the GELU reads the other
  f16x8 bh[2], bl[2];              // GELU(H) of chunk t - 2 as the B operands of its two K steps
#pragma unroll
  for (int e = 0; e < 16; ++e) { H0[e] = 0.f; H1[e] = 0.f; }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) { bh[c][e] = (_Float16)0.f; bl[c][e] = (_Float16)0.f; }

  const bool live = row < p.M;
  const float mk = (p.rowmask && !p.rowmask[row_c]) ? 0.f : 1.f;
  const float* Rr = p.R + (int64_t)row_c * p.ldr + 4 * h;
  float* Cr = p.C + (int64_t)row_c * p.ldc + 4 * h;
  f32x4 res[4][8];
  auto load_res = [&](int op, f32x4 (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const f32x4*>(Rr + 64 * op + 8 * j);
  };

  const unsigned a1_off = (unsigned)(h * 32 + ((r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1))) * 16u;   // slot of W1 row pi(r)
  const unsigned a2_off = 32768u + (unsigned)lane * 16u;

  // iteration t: [A] first product of chunk t -> Hn; [B] GELU + split of chunk t - 1 (Hc) -> bhn / bln; [C] second product of
  // chunk t - 2 (bh / bl) -> Y.  The three are independent of each other inside an iteration.
  auto iter = [&](int t, auto do_a, auto do_b, auto do_c, auto pre, f32x16& Hn, const f32x16& Hc) __attribute__((always_inline)) {
    constexpr bool DA = decltype(do_a)::value, DB = decltype(do_b)::value, DC = decltype(do_c)::value;
    STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage t have landed ...
    STAMP(0);
    __syncthreads();                                       // ... everybody's have, and nobody reads stage t - 1 any more
    STAMP(1);
    if constexpr (decltype(pre)::value) {                  // last iteration: the X planes are dead, their registers take the residual rows
#pragma unroll
      for (int op = 0; op < 4; ++op) load_res(op, res[op]);
    }
    STAMP(2);
    const unsigned char* buf = lds + (t & 1) * STAGE;
    f32x4 sv[4], cv[4];
    if constexpr (DB) {
      const int hb = (t - 1) * 32 + h * 8;
      cv[0] = *reinterpret_cast<const f32x4*>(lds_c + hb);      cv[1] = *reinterpret_cast<const f32x4*>(lds_c + hb + 4);
      cv[2] = *reinterpret_cast<const f32x4*>(lds_c + hb + 16); cv[3] = *reinterpret_cast<const f32x4*>(lds_c + hb + 20);
      if constexpr (FOLD) {
        sv[0] = *reinterpret_cast<const f32x4*>(lds_s + hb);      sv[1] = *reinterpret_cast<const f32x4*>(lds_s + hb + 4);
        sv[2] = *reinterpret_cast<const f32x4*>(lds_s + hb + 16); sv[3] = *reinterpret_cast<const f32x4*>(lds_s + hb + 20);
      }
    }
    if constexpr (DA) {
#pragma unroll
      for (int e = 0; e < 16; ++e) Hn[e] = 0.f;
    }
    u32x4 nh[2], nl[2];
    float g_prev = 0.f;
    // fragments of slot s + 1 are read while slot s computes: fa = W1 (hi, lo), fc = W2 (hi, lo), two register sets
    f16x8 fa[2][2], fc[2][2];
    auto frags = [&](int s, int set) __attribute__((always_inline)) {
      const int q = (s >> 1) * 4 + (s & 1) * 2;
      if constexpr (DA) {
        fa[set][0] = *reinterpret_cast<const f16x8*>(buf + q * 1024 + a1_off);
        fa[set][1] = *reinterpret_cast<const f16x8*>(buf + (q + 1) * 1024 + a1_off);
      }
      if constexpr (DC) {
        fc[set][0] = *reinterpret_cast<const f16x8*>(buf + q * 1024 + a2_off);
        fc[set][1] = *reinterpret_cast<const f16x8*>(buf + (q + 1) * 1024 + a2_off);
      }
    };
    frags(0, 0);
    // A wave issues in order: MFMAs in a row leave the vector unit idle behind them, and a run of vector work leaves the
    // matrix unit idle.  The compiler's schedule is exactly that (all 96 MFMAs of an iteration, then the ~380 vector
    // instructions of the GELU; sched_group_barrier pipelines are not honoured in this kernel), so the order is pinned with
    // real dependences: an empty asm statement after every MFMA takes the MFMA's accumulator, the fragment of the NEXT MFMA and
    // the live values of the GELU stage written beside it -- one MFMA and one stage (4 - 6 vector instructions, at most two of
    // them transcendental: what a 32-cycle MFMA hides) between two such statements.
#define DCF_PIN(acc, fnext, x0, x1) asm volatile("" : "+a"(acc), "+v"(fnext), "+v"(x0), "+v"(x1))
#define DCF_PIN_MEM(acc, x0, x1) asm volatile("" : "+a"(acc), "+v"(x0), "+v"(x1) : : "memory")
#define DCF_PIN0(acc, fnext) asm volatile("" : "+a"(acc), "+v"(fnext))           // (slots whose GELU stage is empty)
#define DCF_PIN0_MEM(acc) asm volatile("" : "+a"(acc) : : "memory")
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int set = s & 1;
      if (s + 1 < 16) frags(s + 1, set ^ 1);
      // the next stage's 16 pieces, two per slot in the first half of the iteration: issuing one costs the wave ~60 cycles (a
      // whole stage in a row was 1030 cycles of 4900 per iteration with the matrix unit idle), beside an MFMA most of it hides;
      // the last of them still has half an iteration to land
      if constexpr (!decltype(pre)::value) {               // (the last iteration has no successor: nothing may be in flight when the wave ends)
        if (s < 8) { issue_piece(t + 1, 2 * s); issue_piece(t + 1, 2 * s + 1); }
      }
      // the GELU of hidden unit 16 (s >> 3) + 8 h + (s & 7) of chunk t - 1 (gelu_erf of common.h, cut into stages)
      float v = 0.f, z = 0.f, tt = 0.f, ex = 0.f, pl = 0.f, gg = 0.f, d0 = 0.f, d1 = 0.f;
      unsigned hi = 0u, lo = 0u;
      auto stage = [&](int k) __attribute__((always_inline)) {
        if constexpr (DB) {
          if (k == 0) {
            v = Hc[s] * UNSCALE;
            const float cb = cv[(s >> 3) * 2 + ((s & 7) >> 2)][s & 3];
            if constexpr (FOLD) v = __builtin_fmaf(__builtin_fmaf(-mean, sv[(s >> 3) * 2 + ((s & 7) >> 2)][s & 3], v), rstd, cb);
            else v += cb;
            z = fabsf(v) * 0.70710678118654752440f;
          } else if (k == 1) {
            tt = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
            ex = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
          } else if (k == 2) {
            pl = tt * (0.254829592f + tt * (-0.284496736f + tt * (1.421413741f + tt * (-1.453152027f + tt * 1.061405429f))));
          } else if (k == 3) {
            const float pe = pl * ex;
            const float one_plus_erf = v >= 0.f ? 2.0f - pe : pe;
            gg = 0.5f * v * one_plus_erf;
            if (!(s & 1)) g_prev = gg;
          } else if (k == 4) {
            if (s & 1) {                                    // split2_f16 of the pair (s - 1, s), first half
              const f16x2 hh = __builtin_convertvector(f32x2{g_prev * SA, gg * SA}, f16x2);
              hi = __builtin_bit_cast(unsigned, hh);
              d0 = __builtin_fmaf(g_prev, SA, -(float)hh[0]);
              d1 = __builtin_fmaf(gg, SA, -(float)hh[1]);
            }
          } else {
            if (s & 1) {
              lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{d0, d1}, f16x2));
              nh[s >> 3][(s & 7) >> 1] = hi;
              nl[s >> 3][(s & 7) >> 1] = lo;
            }
          }
        }
      };
      if constexpr (DA && DC) {
        // K step s of H^T = W1 X^T and (output tile s >> 1, K step s & 1) of Y^T += W2 H^T, alternating: consecutive MFMAs feed
        // different accumulators (a row of MFMAs into ONE accumulator issues at about half rate: head_chain.hip measured 63
        // cycles per MFMA for 27 in a row against 32)
        Hn = mma(fa[set][1], xh[s], Hn);
        stage(0);
        DCF_PIN(Hn, fc[set][1], v, z);
        Y[s >> 1] = mma(fc[set][1], bh[s & 1], Y[s >> 1]);
        stage(1);
        DCF_PIN(Y[s >> 1], fa[set][0], tt, ex);
        Hn = mma(fa[set][0], xl[s], Hn);
        stage(2);
        DCF_PIN(Hn, fc[set][0], pl, tt);
        Y[s >> 1] = mma(fc[set][0], bl[s & 1], Y[s >> 1]);
        stage(3);
        DCF_PIN(Y[s >> 1], fa[set][0], gg, g_prev);
        Hn = mma(fa[set][0], xh[s], Hn);
        stage(4);
        if (s & 1) DCF_PIN(Hn, fc[set][0], d0, d1);
        else DCF_PIN0(Hn, fc[set][0]);
        Y[s >> 1] = mma(fc[set][0], bh[s & 1], Y[s >> 1]);
        stage(5);
        if (s & 1) DCF_PIN_MEM(Y[s >> 1], hi, lo);
        else DCF_PIN0_MEM(Y[s >> 1]);
      } else {
      if constexpr (DA) {                                   // K step s of H^T = W1 X^T
        Hn = mma(fa[set][1], xh[s], Hn);
        stage(0);
        DCF_PIN(Hn, fa[set][0], v, z);
        Hn = mma(fa[set][0], xl[s], Hn);
        stage(1);
        DCF_PIN(Hn, fa[set][0], tt, ex);
        Hn = mma(fa[set][0], xh[s], Hn);
        stage(2);
        if constexpr (DC) DCF_PIN(Hn, fc[set][1], pl, tt);
        else DCF_PIN_MEM(Hn, pl, tt);
      } else {
        stage(0); stage(1); stage(2);
      }
      if constexpr (DC) {                                   // output tile s >> 1, K step s & 1 of Y^T += W2 H^T
        Y[s >> 1] = mma(fc[set][1], bh[s & 1], Y[s >> 1]);
        stage(3);
        DCF_PIN(Y[s >> 1], fc[set][0], gg, g_prev);
        Y[s >> 1] = mma(fc[set][0], bl[s & 1], Y[s >> 1]);
        stage(4);
        DCF_PIN(Y[s >> 1], fc[set][0], d0, d1);
        Y[s >> 1] = mma(fc[set][0], bh[s & 1], Y[s >> 1]);
        stage(5);
        DCF_PIN_MEM(Y[s >> 1], hi, lo);
      } else {
        stage(3); stage(4); stage(5);
      }
      }
    }
#undef DCF_PIN
#undef DCF_PIN_MEM
#undef DCF_PIN0
#undef DCF_PIN0_MEM
    if constexpr (DB) {
      bh[0] = __builtin_bit_cast(f16x8, nh[0]); bh[1] = __builtin_bit_cast(f16x8, nh[1]);
      bl[0] = __builtin_bit_cast(f16x8, nl[0]); bl[1] = __builtin_bit_cast(f16x8, nl[1]);
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  iter(0, T_{}, F_{}, F_{}, F_{}, H0, H1);
  iter(1, T_{}, T_{}, F_{}, F_{}, H1, H0);
  for (int t = 2; t < NCHUNK; t += 2) {                    // (two iterations per trip: the accumulators swap roles without a copy)
    iter(t, T_{}, T_{}, T_{}, F_{}, H0, H1);
    iter(t + 1, T_{}, T_{}, T_{}, F_{}, H1, H0);
  }
  iter(NCHUNK, F_{}, T_{}, T_{}, F_{}, H0, H1);
  iter(NCHUNK + 1, F_{}, F_{}, T_{}, T_{}, H1, H0);

  // epilogue: lane (r, h) holds, of row r, the output columns 32 ot + 8 g + 4 h .. + 3 in Y[ot][4 g .. 4 g + 3].  The residual
  // rows were requested at the start of the last iteration (32 loads of 16 bytes per lane into the registers the X planes no
  // longer need); bias and LayerScale wait in LDS.
  STAMP(3);
  bool bad = false;
  float ps = 0.f, pss = 0.f;
#pragma unroll
  for (int op = 0; op < 4; ++op) {
#pragma unroll
    for (int o2 = 0; o2 < 2; ++o2) {
      const int ot = 2 * op + o2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 32 * ot + 8 * g + 4 * h;
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(lds_b2 + col), lsv = *reinterpret_cast<const f32x4*>(lds_ls + col);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = Y[ot][4 * g + i] * UNSCALE;
          bad |= !(__builtin_fabsf(a) <= 3.4028234664e38f);
          v[i] = a;
        }
        v += b2;
        v *= mk;
        v = res[op][o2 * 4 + g] + lsv * v;
        if (live) *reinterpret_cast<f32x4*>(Cr + 32 * ot + 8 * g) = v;
        ps += (v.x + v.y) + (v.z + v.w);
        pss += __builtin_fmaf(v.x, v.x, v.y * v.y) + __builtin_fmaf(v.z, v.z, v.w * v.w);
      }
    }
  }
  if (p.stats_out) {               // (sum, sum of squares) of the row: slot 0 carries it, the other slots of the row are zero
    const float s1 = xor32_sum(ps), s2 = xor32_sum(pss);
    if (live && h == 0) {
      const int slots = FE / p.stats_w;
      float* o = p.stats_out + (int64_t)row * slots * 2;
      o[0] = s1; o[1] = s2;
      for (int k = 1; k < slots; ++k) { o[2 * k] = 0.f; o[2 * k + 1] = 0.f; }
    }
  }
  if (bad && p.status) atomicOr(p.status, 1u);
#ifdef DCF_FFN_STAMP
  STAMP(4);
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) dcf_ffn_stamps[i] = acc_[i];
#endif
}

// (sum, sum of squares) of every row in the slot layout of GemmArgs::stats_out (slot 0 carries the row, the others are zero):
// what a producer GEMM's epilogue writes, for callers whose rows come from somewhere else.  One wave per row.
__global__ __launch_bounds__(256) void k_row_stats(const float* __restrict__ X, int64_t ldx, float* __restrict__ stats, int rows, int C,
                                                   int slots) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = X[(int64_t)row * ldx + c]; s1 += v; s2 = __builtin_fmaf(v, v, s2); }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane < slots) { stats[((int64_t)row * slots + lane) * 2] = lane == 0 ? s1 : 0.f; stats[((int64_t)row * slots + lane) * 2 + 1] = lane == 0 ? s2 : 0.f; }
}

int launch_row_stats(const float* X, int64_t ldx, float* stats, int rows, int C, int stats_w, hipStream_t stream) {
  DCF_CHECK(stats_w > 0 && C % stats_w == 0 && C / stats_w <= 64, "launch_row_stats: bad slot width");
  hipLaunchKernelGGL(k_row_stats, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, X, ldx, stats, rows, C, C / stats_w);
  DCF_HIP(hipGetLastError());
  return 0;
}

#ifdef DCF_FFN_STAMP
}  // namespace dcf
extern "C" int dcf_debug_ffn_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcf::dcf_ffn_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace dcf {
#endif

int launch_ffn_chain(const FfnChainArgs& a, hipStream_t stream) {
  DCF_CHECK(a.M > 0 && a.X && a.W1s && a.b1 && a.W2s && a.b2 && a.R && a.C, "launch_ffn_chain: null argument");
  DCF_CHECK(!a.stats || (a.ln_s && a.stats_slots >= 1), "launch_ffn_chain: stats need ln_s and stats_slots");
  DCF_CHECK(!a.stats_out || (a.stats_w > 0 && FE % a.stats_w == 0), "launch_ffn_chain: stats_out needs a slot width dividing %d", FE);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DCF_CHECK(al16(a.X) && al16(a.R) && al16(a.C) && al16(a.b1) && al16(a.b2) && al16(a.W1s) && al16(a.W2s) && (!a.ls || al16(a.ls)) &&
                (!a.ln_s || al16(a.ln_s)) && a.ldx % 4 == 0 && a.ldr % 4 == 0 && a.ldc % 4 == 0,
            "launch_ffn_chain: operands must be 16-byte aligned with row pitches that are multiples of 4");
  static bool attr_set[64] = {};                         // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn_chain<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn_chain<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set[dev] = true;
  }
  const unsigned grid = (unsigned)((a.M + 127) / 128);
  if (a.stats) hipLaunchKernelGGL(k_ffn_chain<true>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(k_ffn_chain<false>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
