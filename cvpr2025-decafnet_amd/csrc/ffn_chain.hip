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

#include <stdlib.h>

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
constexpr float U16 = SA * UNSCALE;                  // the GELU runs on SA x (hidden pre-activation): its result is the next product's scaled operand
constexpr float CZ16 = GELU_CZ / SA;
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
  // M0 is written and NOT restored: nothing else in these kernels reads it (gfx9+ LDS instructions do not; tools/isa_gate.py and
  // tests/test_abi.py keep every other use of m0 out of the shipped objects).  DCF_GLDS_KEEP_M0 = the save / restore form
  // (two more scalar instructions per request, ~2 % of a chain kernel's stage).
#ifdef DCF_GLDS_KEEP_M0
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
#else
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
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
__device__ unsigned long long dcf_pair_stamps[16];     // k_ffn_pair: [0..7] producer wave 0, [8..15] consumer wave 4 of workgroup 0
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
    *reinterpret_cast<f32x4*>(lds_c + tid * 4) = c4 * SA;                // (x 16: the GELU's operand scale)
    if constexpr (FOLD) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.ln_s + tid * 4);
      *reinterpret_cast<f32x4*>(lds_s + tid * 4) = s4 * SA;
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
    const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.f);
    rstd = 1.0f / sqrtf(var + 1e-5f);
    if (row < p.M && ln_ill(mean, var) && p.status) atomicOr(p.status, 2u);       // (common.h LN_ILL_RATIO)
  }
  // folded LayerNorm: 16 (rstd (acc U - mean s) + c) = acc r1 + (16 s) r2 + 16 c
  const float r1 = FOLD ? rstd * U16 : U16, r2 = -rstd * mean;
  float one = 1.0f;
  asm volatile("" : "+s"(one));      // (an opaque 1: `g - (float)h` as ONE v_fma_mix_f32 instead of a conversion and a subtraction)

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
          if (k == 0) {                                     // v = 16 x the hidden pre-activation
            const float cb = cv[(s >> 3) * 2 + ((s & 7) >> 2)][s & 3];
            if constexpr (FOLD) v = __builtin_fmaf(Hc[s], r1, __builtin_fmaf(sv[(s >> 3) * 2 + ((s & 7) >> 2)][s & 3], r2, cb));
            else v = __builtin_fmaf(Hc[s], r1, cb);
            z = fabsf(v) * CZ16;
          } else if (k == 1) {
            tt = __builtin_amdgcn_rcpf(__builtin_fmaf(z, GELU_CT, 1.0f));
            ex = __builtin_amdgcn_exp2f(-z * z);
          } else if (k == 2) {
            pl = gelu_half_poly(tt);
          } else if (k == 3) {
            gg = __builtin_fmaf(-fabsf(v), pl * ex, relu_max(v));         // = 16 gelu
            if (!(s & 1)) g_prev = gg;
          } else if (k == 4) {
            if (s & 1) {                                    // split2_f16 of the pair (s - 1, s), first half
              const f16x2 hh = __builtin_convertvector(f32x2{g_prev, gg}, f16x2);
              hi = __builtin_bit_cast(unsigned, hh);
              d0 = __builtin_fmaf(g_prev, one, -(float)hh[0]);
              d1 = __builtin_fmaf(gg, one, -(float)hh[1]);
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

// ---- the same FFN on EIGHT waves: a producer and a consumer wave per 32 rows, two waves per SIMD at 256 registers -------------
// In the four-wave kernel above ONE wave per SIMD issues, in order, the 96 MFMAs of a hidden chunk, the ~300 vector instructions
// of its GELU / operand split and the weight stream's requests.  Here the two products of a row tile live in two waves that
// share a SIMD:
//   producer P_j (waves 0 - 3): X planes of its 32 rows (128 registers), H^T = W1 X^T for chunk t, GELU + split of chunk t - 1,
//                               the result -- the B operand of the second product, 4 KiB -- goes to LDS;
//   consumer C_j (waves 4 - 7): the 32 x 256 output accumulators (128 registers), Y^T += W2 H^T for chunk t - 2, the epilogue.
// The matrix pipe of the SIMD takes MFMAs from both (48 + 48 per chunk, the same 96), the producer's vector work issues beside
// the consumer's MFMAs as well as its own, and each wave issues half of the stage's weight requests (8 instead of 16).
// The kernel is persistent (one workgroup per CU walks its row tiles): a tile costs 33 iterations instead of 34, the consumer's
// epilogue (residual rows in, output rows out) runs beside the producer's first products of the NEXT tile, and the producer
// loads the next tile's rows beside the consumer's last products -- the row phase is no longer exposed.  Arithmetic (products,
// their order per accumulator, GELU, epilogue) is that of k_ffn_chain: the two kernels agree bit for bit (tests/test_gpu_ops.py).
// Measured (profiles/r05_notes.md section 1, tools/ffn_pair_probe.sh): alone on 131 072 rows the two kernels tie (405 - 420 us by
// box); in the eight-video forward this one is 3 % faster per launch (1.66 against 1.71 ms for the six FFN launches: levels 1 and 2
// have 2 and 1 tiles per CU, where the overlapped row phase counts most).  Its stamps: producer loop 2 860 cycles, consumer loop
// 3 800, 96 MFMAs = 3 072; the rest of an iteration's ~4 600 is the exchange handshake, the stage wait and the barrier skew of
// eight waves.  Without ANY weight stream the kernel takes 365 - 390 us, without the GELU 385 - 395: neither is the bound by
// itself any more; what remains is the per-chunk synchronisation of this LDS-ring design.
//
// LDS: the same two 64 KiB stages (W1 chunk t | W2 chunk t - 2), ONE 4 KiB exchange buffer per pair and a word per pair that the
// consumer sets to the iteration number once it has the buffer's content in registers; the producer checks that word before it writes
// the next content -- the first K step's half in the middle of its iteration, the second at the end (it practically never waits:
// the consumer reads the buffer first thing after the barrier, some 1 500 cycles earlier).  Bias vectors behind that: 154 KiB in all.
constexpr int XCH = 4096;                      // bytes of an exchange buffer: bh[0] | bh[1] | bl[0] | bl[1], 64 lanes x 16 bytes each
constexpr int PAIR_FLAGS = 2 * STAGE + 4 * XCH;
constexpr int PAIR_BIAS = PAIR_FLAGS + 64;
constexpr int PAIR_LDS_BYTES = PAIR_BIAS + (2 * FH + 2 * FE) * (int)sizeof(float);
constexpr int TILE_ITERS = NCHUNK + 1;         // 33: producer iterations per row tile (32 first products + one GELU-and-reload)
#ifndef DCF_PAIR_CSHARE
#define DCF_PAIR_CSHARE 4
#endif
constexpr int PAIR_CSHARE = DCF_PAIR_CSHARE;   // W2 pieces of a stage (of a wave's eight) the consumer requests itself while it multiplies

template <bool FOLD>
__global__ __launch_bounds__(512, 2) void k_ffn_pair(FfnChainArgs p, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* xch_all = lds + 2 * STAGE;
  volatile int* flags = reinterpret_cast<volatile int*>(lds + PAIR_FLAGS);
  float* lds_s = reinterpret_cast<float*>(lds + PAIR_BIAS);        // [1024] ln_s (FOLD), then [1024] b1, [256] b2, [256] LayerScale
  float* lds_c = lds_s + FH;
  float* lds_b2 = lds_c + FH;
  float* lds_ls = lds_b2 + FE;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = w & 3;                                             // the pair = the 32-row slice of the tile
  const bool producer = w < 4;                                     // (the consumers as the older waves of their SIMDs: 423 against 392 us)
  const unsigned lane16 = (unsigned)lane * 16u;
  const int my_tiles = (tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total_iters = my_tiles * TILE_ITERS + 2;               // the consumer runs two iterations behind the producer
  unsigned char* xch = xch_all + j * XCH;

  for (int i = tid; i < FH; i += 512) { lds_c[i] = p.b1[i] * SA; lds_s[i] = FOLD ? p.ln_s[i] * SA : 0.f; }   // (x 16: the GELU's operand scale)
  if (tid < FE) { lds_b2[tid] = p.b2[tid]; lds_ls[tid] = p.ls ? p.ls[tid] : 1.f; }

  if (producer) {
    // ---------------------------------------------------------------- producer ----------------------------------------------
    // weight stream: W1 rows of the next chunk, K blocks 2 j and 2 j + 1 (8 pieces of 1 KiB)
    const unsigned short* dma_base = p.W1s + (size_t)(2 * j) * BLK;
    const unsigned dma_dst = (unsigned)(2 * j) * 4096u;
    auto issue_piece = [&](int g_next, int ch, int i) __attribute__((always_inline)) {
      const unsigned short* src = dma_base + (size_t)ch * (FE / 32) * BLK + (size_t)(i >> 2) * BLK + (((i >> 1) & 1) * 3 + (i & 1)) * PIECE;
      glds16(src, lane16, dma_dst + (unsigned)(g_next & 1) * STAGE + (unsigned)i * 1024u);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(0, 0, i);
    // ... and, while its consumer multiplies, the last 8 - PAIR_CSHARE of that wave's eight W2 pieces: a request stalls the wave that
    // issues it for 60 - 100 cycles; the consumer -- the younger wave of the SIMD, it gets the matrix pipe when this one does not want
    // it -- is the one the barrier waits for (3 780 against 3 280 cycles with eight requests each), this wave has the slack
    const unsigned short* w2_base = p.W2s + (size_t)(2 * j) * (FH / 32) * BLK;
    auto issue_piece2 = [&](int g_next, int ch, int i) __attribute__((always_inline)) {
      const unsigned short* src = w2_base + (size_t)ch * BLK + (size_t)(i >> 2) * (FH / 32) * BLK + (((i >> 1) & 1) * 3 + (i & 1)) * PIECE;
      glds16(src, lane16, 32768u + (unsigned)(2 * j) * 4096u + (unsigned)(g_next & 1) * STAGE + (unsigned)i * 1024u);
    };

    f16x8 xh[16], xl[16];
    // folded LayerNorm: 16 (rstd (acc U - mean s) + c) = acc r1 + (16 s) r2 + 16 c, r1 = rstd U16, r2 = - rstd mean (per row = per lane)
    float r1 = U16, r2 = 0.f;
    float one = 1.0f;
    asm volatile("" : "+s"(one));    // (an opaque 1: `g - (float)h` as ONE v_fma_mix_f32 instead of a conversion and a subtraction)
    auto row_of = [&](int tile_k) __attribute__((always_inline)) {
      const int row = ((int)blockIdx.x + tile_k * (int)gridDim.x) * 128 + j * 32 + r;
      return row < p.M ? row : p.M - 1;
    };
    auto row_stats = [&](int row_c, float& r1_o, float& r2_o) __attribute__((always_inline)) {
      if constexpr (FOLD) {                                          // as stats_load of gemm_common.h
        const float* sp = p.stats + (int64_t)row_c * p.stats_slots * 2;
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < p.stats_slots; ++k) { s1 += sp[2 * k]; s2 += sp[2 * k + 1]; }
        const float inv = 1.0f / (float)FE;
        const float mean = s1 * inv;
        const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.f);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (ln_ill(mean, var) && p.status) atomicOr(p.status, 2u);                // (common.h LN_ILL_RATIO; a clamped row repeats row M - 1)
        r1_o = rstd * U16;
        r2_o = -rstd * mean;
      }
    };
    auto split_rows = [&](const f32x4 (&raw)[32]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        unsigned hi[4], lo[4];
        split2_f16(raw[2 * s].x, raw[2 * s].y, hi[0], lo[0]);
        split2_f16(raw[2 * s].z, raw[2 * s].w, hi[1], lo[1]);
        split2_f16(raw[2 * s + 1].x, raw[2 * s + 1].y, hi[2], lo[2]);
        split2_f16(raw[2 * s + 1].z, raw[2 * s + 1].w, hi[3], lo[3]);
        xh[s] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
        xl[s] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
      }
    };
    {
      const int row_c = row_of(0);
      const float* px = p.X + (int64_t)row_c * p.ldx + 8 * h;
      f32x4 raw[32];
#pragma unroll
      for (int s = 0; s < 16; ++s) { raw[2 * s] = *reinterpret_cast<const f32x4*>(px + 16 * s); raw[2 * s + 1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4); }
      row_stats(row_c, r1, r2);
      split_rows(raw);
    }
    f32x16 H0, H1;
    const unsigned a1_off = (unsigned)(h * 32 + ((r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1))) * 16u;   // slot of W1 row pi(r)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);                                // lgkmcnt(0): the bias vectors are in LDS
    __builtin_amdgcn_s_barrier();
#ifdef DCF_FFN_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = stamp();
#endif

    // iteration g (i = its index inside the tile): [A] first product of chunk i -> Hn (i < 32); [B] GELU + split of chunk i - 1
    // (Hc) -> exchange buffer (i > 0); [X] i = 32: the next tile's rows.  ch_next: the W1 chunk of iteration g + 1 (-1: none).
    auto iter = [&](int g, int i, int ch_next, int next_tile_k, auto do_a, auto do_b, auto do_x, f32x16& Hn, const f32x16& Hc)
                    __attribute__((always_inline)) {
      constexpr bool DA = decltype(do_a)::value, DB = decltype(do_b)::value, DX = decltype(do_x)::value;
      const unsigned char* buf = lds + (g & 1) * STAGE;
      // what the consumer does in THIS iteration (a product: it then requests only PAIR_CSHARE of its W2 pieces) and the W2 chunk it
      // needs in the next one (it stands at i + 1 of this tile then, or at 0 of the next tile / of the drain)
      const bool c_prod = i >= 2 || (i == 0 && g >= TILE_ITERS);
      const int ch2_next = (i + 1 >= 2 && i + 1 < TILE_ITERS) ? i - 1 : (i + 1 == TILE_ITERS ? NCHUNK - 1 : -1);
      f32x4 raw[32];
      float r1_n = U16, r2_n = 0.f;
      if constexpr (DX) {
        if (next_tile_k >= 0) {
          const int row_c = row_of(next_tile_k);
          const float* px = p.X + (int64_t)row_c * p.ldx + 8 * h;
#pragma unroll
          for (int s = 0; s < 16; ++s) { raw[2 * s] = *reinterpret_cast<const f32x4*>(px + 16 * s); raw[2 * s + 1] = *reinterpret_cast<const f32x4*>(px + 16 * s + 4); }
          row_stats(row_c, r1_n, r2_n);
        } else {
#pragma unroll
          for (int s = 0; s < 32; ++s) raw[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if constexpr (DA) {
#pragma unroll
        for (int e = 0; e < 16; ++e) Hn[e] = 0.f;
      }
      u32x4 nh[2], nl[2];
      float gp[2];                                                     // 16 gelu of the hidden units of slots s - 1 (even), s (odd)
      f16x8 fa[2][2];
      f32x4 cq[2], sq[2];                                              // bias (and folded row sums) of four hidden units, one group ahead
      const int hb = (i - 1) * 32 + h * 8;
      auto frags = [&](int s, int set) __attribute__((always_inline)) {
        const int q = (s >> 1) * 4 + (s & 1) * 2;
        fa[set][0] = *reinterpret_cast<const f16x8*>(buf + q * 1024 + a1_off);
        fa[set][1] = *reinterpret_cast<const f16x8*>(buf + (q + 1) * 1024 + a1_off);
      };
      auto quad = [&](int qd, int set) __attribute__((always_inline)) {
        const int off = (qd >> 1) * 16 + (qd & 1) * 4;
        cq[set] = *reinterpret_cast<const f32x4*>(lds_c + hb + off);
        if constexpr (FOLD) sq[set] = *reinterpret_cast<const f32x4*>(lds_s + hb + off);
      };
      if constexpr (DA) frags(0, 0);
      if constexpr (DB) quad(0, 0);
      // The order is pinned as in k_ffn_chain: an empty asm statement behind an MFMA takes its accumulator, the next MFMA's fragment
      // and the LIVE results of the vector steps written beside it.  Per hidden unit 16 - 17 vector instructions (common.h
      // gelu_erf in steps, on 16 x the pre-activation), i.e. 5 - 6 in the shadow of each of the slot's three MFMAs.
#define DCF_PINV3(acc, fnext, x0, x1, x2) asm volatile("" : "+v"(acc), "+v"(fnext), "+v"(x0), "+v"(x1), "+v"(x2))
#define DCF_PINV2(acc, fnext, x0, x1) asm volatile("" : "+v"(acc), "+v"(fnext), "+v"(x0), "+v"(x1))
#define DCF_PINV1M(acc, x0) asm volatile("" : "+v"(acc), "+v"(x0) : : "memory")
#define DCF_PINV2M(acc, x0, x1) asm volatile("" : "+v"(acc), "+v"(x0), "+v"(x1) : : "memory")
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int set = s & 1;
        if constexpr (DA) { if (s + 1 < 16) frags(s + 1, set ^ 1); }
        if constexpr (DB) { if ((s & 3) == 0 && s + 4 < 16) quad((s >> 2) + 1, ((s >> 2) + 1) & 1); }
#ifndef DCF_PAIR_NO_DMA
        if (s < 8 && ch_next >= 0) issue_piece(g + 1, ch_next, s);
        if (s >= 8 && s < 16 - PAIR_CSHARE && c_prod && ch2_next >= 0) issue_piece2(g + 1, ch2_next, PAIR_CSHARE + s - 8);
#endif
        if constexpr (DB) {
          if (s == 8) {
            // the first K step's operand (hidden units of slots 0 .. 7) is complete: out it goes, half an iteration before the barrier.
            // The consumer has had this buffer's previous content in registers since the start of the iteration (flags[j] == g).
            while (flags[j] < g) __builtin_amdgcn_s_sleep(1);
            *reinterpret_cast<u32x4*>(xch + 0 * 1024 + lane16) = nh[0];
            *reinterpret_cast<u32x4*>(xch + 2 * 1024 + lane16) = nl[0];
          }
        }
        // the GELU of hidden unit 16 (s >> 3) + 8 h + (s & 7) of chunk i - 1
        float v, tt, ex;
        auto step_a = [&]() __attribute__((always_inline)) {             // 16 x pre-activation, t, e^{-z^2}
          const float cb = cq[(s >> 2) & 1][s & 3];
          if constexpr (FOLD) v = __builtin_fmaf(Hc[s], r1, __builtin_fmaf(sq[(s >> 2) & 1][s & 3], r2, cb));
          else v = __builtin_fmaf(Hc[s], r1, cb);
          const float z = fabsf(v) * CZ16;
          tt = __builtin_amdgcn_rcpf(__builtin_fmaf(z, GELU_CT, 1.0f));
          ex = __builtin_amdgcn_exp2f(-z * z);
        };
        auto step_b = [&]() __attribute__((always_inline)) {             // 16 gelu
          gp[s & 1] = __builtin_fmaf(-fabsf(v), gelu_half_poly(tt) * ex, relu_max(v));
        };
        auto step_c = [&]() __attribute__((always_inline)) {             // split2_f16 of the pair (s - 1, s)
          const f16x2 hh = __builtin_convertvector(f32x2{gp[0], gp[1]}, f16x2);
          const float d0 = __builtin_fmaf(gp[0], one, -(float)hh[0]), d1 = __builtin_fmaf(gp[1], one, -(float)hh[1]);
          nh[s >> 3][(s & 7) >> 1] = __builtin_bit_cast(unsigned, hh);
          nl[s >> 3][(s & 7) >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{d0, d1}, f16x2));
        };
#ifdef DCF_PAIR_NO_GELU
        if constexpr (DB) { gp[s & 1] = Hc[s]; if (s & 1) { nh[s >> 3][(s & 7) >> 1] = __builtin_bit_cast(unsigned, gp[0]); nl[s >> 3][(s & 7) >> 1] = __builtin_bit_cast(unsigned, gp[1]); } }
        if constexpr (DA) {
          Hn = mma(fa[set][1], xh[s], Hn);
          Hn = mma(fa[set][0], xl[s], Hn);
          Hn = mma(fa[set][0], xh[s], Hn);
        }
#else
        if constexpr (DA && DB) {                                       // K step s of H^T = W1 X^T with the vector steps in the MFMAs' shadows
          Hn = mma(fa[set][1], xh[s], Hn);
          step_a();
          DCF_PINV3(Hn, fa[set][0], v, tt, ex);
          Hn = mma(fa[set][0], xl[s], Hn);
          step_b();
          DCF_PINV2(Hn, fa[set][0], gp[0], gp[1]);
          Hn = mma(fa[set][0], xh[s], Hn);
          if (s & 1) { step_c(); DCF_PINV2M(Hn, nh[s >> 3][(s & 7) >> 1], nl[s >> 3][(s & 7) >> 1]); }
          else DCF_PINV1M(Hn, gp[0]);
        } else if constexpr (DA) {
          Hn = mma(fa[set][1], xh[s], Hn);
          Hn = mma(fa[set][0], xl[s], Hn);
          Hn = mma(fa[set][0], xh[s], Hn);
        } else {
          step_a(); step_b();
          if (s & 1) step_c();
        }
#endif
      }
#undef DCF_PINV3
#undef DCF_PINV2
#undef DCF_PINV1M
#undef DCF_PINV2M
      STAMP(DX ? 4 : 0);                                               // the slot loop (4: the iteration that reloads the rows)
      if constexpr (DB) {                                              // (the first K step's half went out behind slot 8)
        *reinterpret_cast<u32x4*>(xch + 1 * 1024 + lane16) = nh[1];
        *reinterpret_cast<u32x4*>(xch + 3 * 1024 + lane16) = nl[1];
      }
      STAMP(1);                                                        // handshake + exchange write
      if constexpr (DX) {
        split_rows(raw);
        r1 = r1_n;
        r2 = r2_n;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      STAMP(2);                                                        // (row split,) wait for the stage's pieces
      __builtin_amdgcn_s_barrier();
      STAMP(3);                                                        // barrier
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int g = 0;
    for (int k = 0; k < my_tiles; ++k) {
      iter(g, 0, 1, -1, T_{}, F_{}, F_{}, H0, H1); ++g;
      for (int i = 1; i < NCHUNK - 1; i += 2) {                       // (two iterations per trip: the accumulators swap roles without a copy)
        iter(g, i, i + 1, -1, T_{}, T_{}, F_{}, H1, H0); ++g;
        iter(g, i + 1, i + 2, -1, T_{}, T_{}, F_{}, H0, H1); ++g;
      }
      iter(g, NCHUNK - 1, -1, -1, T_{}, T_{}, F_{}, H1, H0); ++g;
      const bool more = k + 1 < my_tiles;
      iter(g, NCHUNK, more ? 0 : -1, more ? k + 1 : -1, F_{}, T_{}, T_{}, H0, H1); ++g;
    }
    for (int e = 0; e < 2; ++e) __builtin_amdgcn_s_barrier();          // the consumer's last product and its epilogue
#ifdef DCF_FFN_STAMP
    if (blockIdx.x == 0 && lane == 0 && j == 0)
      for (int e = 0; e < 8; ++e) dcf_pair_stamps[e] = acc_[e];
#endif
  } else {
    // ---------------------------------------------------------------- consumer ----------------------------------------------
    const unsigned short* dma_base = p.W2s + (size_t)(2 * j) * (FH / 32) * BLK;
    const unsigned dma_dst = 32768u + (unsigned)(2 * j) * 4096u;
    auto issue_piece = [&](int g_next, int ch, int i) __attribute__((always_inline)) {
      const unsigned short* src = dma_base + (size_t)ch * BLK + (size_t)(i >> 2) * (FH / 32) * BLK + (((i >> 1) & 1) * 3 + (i & 1)) * PIECE;
      glds16(src, lane16, dma_dst + (unsigned)(g_next & 1) * STAGE + (unsigned)i * 1024u);
    };
    const unsigned a2_off = 32768u + (unsigned)lane * 16u;
    f32x16 Y[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[ot][e] = 0.f;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();

#ifdef DCF_FFN_STAMP
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = stamp();
#endif
    int i = 0, tile_k = 0;                                            // the producer's position: iteration i of its tile tile_k
    for (int g = 0; g < total_iters; ++g) {
      const unsigned char* buf = lds + (g & 1) * STAGE;
      // what this iteration holds: i == 1: the epilogue of the previous tile; i == 0: its last chunk; otherwise chunk i - 2 of this tile
      const bool have_prev = g >= TILE_ITERS;
      const bool prod = (i >= 2 && tile_k < my_tiles) || (i == 0 && have_prev);
      // the W2 chunk of iteration g + 1
      int i1 = i + 1, k1 = tile_k;
      if (i1 == TILE_ITERS) { i1 = 0; ++k1; }
      const int ch_next = (i1 >= 2 && k1 < my_tiles) ? i1 - 2 : ((i1 == 0) ? NCHUNK - 1 : -1);
      if (prod) {
        f16x8 bh[2], bl[2];
        bh[0] = *reinterpret_cast<const f16x8*>(xch + 0 * 1024 + lane16);
        bh[1] = *reinterpret_cast<const f16x8*>(xch + 1 * 1024 + lane16);
        bl[0] = *reinterpret_cast<const f16x8*>(xch + 2 * 1024 + lane16);
        bl[1] = *reinterpret_cast<const f16x8*>(xch + 3 * 1024 + lane16);
        // Eight blocks: K step kk = b >> 2 of the output tiles 2 m, 2 m + 1 (m = b & 3).  The compiler's own schedule of this loop is
        // `ds_read -> s_waitcnt -> MFMA` in series (it minimises registers), and an MFMA that accumulates into the result of the one
        // before it issues 64 cycles after it, not 32 (profiles/r03_notes.md section 9): written out instead, a block requests
        // the NEXT block's four fragments, issues its six MFMAs alternating between the two tiles' accumulators (per accumulator the
        // order of the four-wave kernel: lo x hi, hi x lo, hi x hi) and ends with the wait for the fragments, which have had
        // ~200 cycles by then -- everything a block hands on is valid when it ends.
        f16x8 fHa[2], fLa[2], fHb[2], fLb[2];
        const unsigned char* fbase = buf + a2_off;
        fHa[0] = *reinterpret_cast<const f16x8*>(fbase);
        fLa[0] = *reinterpret_cast<const f16x8*>(fbase + 1024);
        fHb[0] = *reinterpret_cast<const f16x8*>(fbase + 4096);
        fLb[0] = *reinterpret_cast<const f16x8*>(fbase + 5120);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        flags[j] = g;
        if (i == 2) {
#pragma unroll
          for (int ot = 0; ot < 8; ++ot)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[ot][e] = 0.f;
        }
        const unsigned faddr = (unsigned)(g & 1) * STAGE + a2_off;      // LDS byte address of this lane's piece of fragment 0 of the W2 half
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int set = b & 1, m = b & 3, kk = b >> 2;
#ifndef DCF_PAIR_NO_DMA
          if (b < PAIR_CSHARE && ch_next >= 0) issue_piece(g + 1, ch_next, b);       // (the rest: its producer, see there)
#endif
          if (b + 1 < 8) {
            const int mn = (b + 1) & 3, kn = (b + 1) >> 2;
            const int qa = (2 * mn) * 4 + kn * 2, qb = (2 * mn + 1) * 4 + kn * 2;
            asm volatile("ds_read_b128 %[nHa], %[addr] offset:%[oHa]\n\t"
                         "ds_read_b128 %[nLa], %[addr] offset:%[oLa]\n\t"
                         "ds_read_b128 %[nHb], %[addr] offset:%[oHb]\n\t"
                         "ds_read_b128 %[nLb], %[addr] offset:%[oLb]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[ya], %[cLa], %[bh], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cLb], %[bh], %[yb]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[ya], %[cHa], %[bl], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cHb], %[bl], %[yb]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[ya], %[cHa], %[bh], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cHb], %[bh], %[yb]\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : [ya] "+v"(Y[2 * m]), [yb] "+v"(Y[2 * m + 1]), [nHa] "=&v"(fHa[set ^ 1]), [nLa] "=&v"(fLa[set ^ 1]),
                           [nHb] "=&v"(fHb[set ^ 1]), [nLb] "=&v"(fLb[set ^ 1])
                         : [cHa] "v"(fHa[set]), [cLa] "v"(fLa[set]), [cHb] "v"(fHb[set]), [cLb] "v"(fLb[set]), [bh] "v"(bh[kk]), [bl] "v"(bl[kk]),
                           [addr] "v"(faddr), [oHa] "i"(qa * 1024), [oLa] "i"((qa + 1) * 1024), [oHb] "i"(qb * 1024), [oLb] "i"((qb + 1) * 1024)
                         : "memory");
          } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %[ya], %[cLa], %[bh], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cLb], %[bh], %[yb]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[ya], %[cHa], %[bl], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cHb], %[bl], %[yb]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[ya], %[cHa], %[bh], %[ya]\n\t"
                         "v_mfma_f32_32x32x16_f16 %[yb], %[cHb], %[bh], %[yb]"
                         : [ya] "+v"(Y[2 * m]), [yb] "+v"(Y[2 * m + 1])
                         : [cHa] "v"(fHa[set]), [cLa] "v"(fLa[set]), [cHb] "v"(fHb[set]), [cLb] "v"(fLb[set]), [bh] "v"(bh[kk]), [bl] "v"(bl[kk]));
          }
        }
      } else {
        flags[j] = g;
        if (ch_next >= 0) {
#pragma unroll
          for (int s = 0; s < 8; ++s) issue_piece(g + 1, ch_next, s);
        }
        if (i == 1 && have_prev) {
          // epilogue of tile tile_k - 1: lane (r, h) holds, of row r, the output columns 32 ot + 8 gq + 4 h .. + 3 in Y[ot][4 gq .. 4 gq + 3]
          const int row = ((int)blockIdx.x + (tile_k - 1) * (int)gridDim.x) * 128 + j * 32 + r;
          const int row_c = row < p.M ? row : p.M - 1;
          const bool live = row < p.M;
          const float mk = (p.rowmask && !p.rowmask[row_c]) ? 0.f : 1.f;
          const float* Rr = p.R + (int64_t)row_c * p.ldr + 4 * h;
          // output rows through a buffer resource: no branch around the stores (a per-lane `if (live)` around them let the compiler
          // merge the four store groups into one block and keep the whole output row in registers)
          const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((int64_t)p.M * p.ldc * 4), 0x00020000);
          unsigned c_off = live ? (unsigned)(((int64_t)row * p.ldc + 4 * h) * 4) : 0x80000000u;
          asm volatile("" : "+v"(c_off));
          float ps = 0.f, pss = 0.f;
          // (one laundered per-lane LDS address + constants: computed here, every time, so that the 64 addresses of the bias reads
          // are not hoisted out of the iteration loop into registers the accumulators need)
          unsigned ep_off = (unsigned)(PAIR_BIAS + 2 * FH * (int)sizeof(float)) + 16u * (unsigned)h;
          asm volatile("" : "+v"(ep_off));
          const unsigned char* ep_lds = lds + ep_off;
          // four passes of 64 output columns: the residual rows of the next pass are in flight while this one is put together in
          // their registers and stored (the accumulators leave 96 registers to this wave)
          f32x4 res[8];
#pragma unroll
          for (int op = 0; op < 4; ++op) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) res[jj] = *reinterpret_cast<const f32x4*>(Rr + 64 * op + 8 * jj);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o2 = 0; o2 < 2; ++o2) {
              const int ot = 2 * op + o2;
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                const int col = 32 * ot + 8 * gq;
                const f32x4 b2 = *reinterpret_cast<const f32x4*>(ep_lds + col * 4), lsv = *reinterpret_cast<const f32x4*>(ep_lds + FE * 4 + col * 4);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = Y[ot][4 * gq + e] * UNSCALE;
                v += b2;
                v *= mk;
                v = res[o2 * 4 + gq] + lsv * v;
                res[o2 * 4 + gq] = v;
                ps += (v.x + v.y) + (v.z + v.w);
                pss += __builtin_fmaf(v.x, v.x, v.y * v.y) + __builtin_fmaf(v.z, v.z, v.w * v.w);
              }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)                    // (rows beyond M: an offset outside the buffer, the store is dropped)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, res[k]), c_rsrc, c_off + (unsigned)(64 * op + 8 * k) * 4u, 0, 0);
            asm volatile("" : "+v"(ps), "+v"(pss));        // (the sums are taken here, not sunk into the `stats_out` branch with the whole row kept alive for them)
            __builtin_amdgcn_sched_barrier(0);
          }
          // a non-finite accumulator anywhere in the row makes the row's sum non-finite (x * 0 and x + y keep NaN / inf - inf): one test
          // per row instead of one per value
          const bool bad = !(__builtin_fabsf(ps) <= 3.4028234664e38f);
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
        }
      }
      if (prod) STAMP(0); else STAMP(4);                               // product iteration / epilogue (or idle) iteration
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      STAMP(2);
      __builtin_amdgcn_s_barrier();
      STAMP(3);
      if (++i == TILE_ITERS) { i = 0; ++tile_k; }
    }
#ifdef DCF_FFN_STAMP
    if (blockIdx.x == 0 && lane == 0 && j == 0)
      for (int e = 0; e < 8; ++e) dcf_pair_stamps[8 + e] = acc_[e];
#endif
  }
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
extern "C" int dcf_debug_pair_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcf::dcf_pair_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace dcf {
#endif

static bool ffn_pair_default() {
  static const bool off = getenv("DCF_FFN_FOUR_WAVES") != nullptr;     // developer switch: the four-wave kernel
  return !off;
}

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
  static int n_cu[64] = {};
  if (dev >= 0 && dev < 64 && !n_cu[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn_pair<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ffn_pair<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PAIR_LDS_BYTES));
    hipDeviceProp_t prop;
    DCF_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const bool pair = a.variant == 2 || (a.variant == 0 && ffn_pair_default());
  if (pair) {
    // persistent: one workgroup per CU walks its row tiles (tile t of workgroup b: b + t * grid)
    const unsigned cus = (unsigned)(dev >= 0 && dev < 64 ? n_cu[dev] : 256);
    const unsigned pgrid = grid < cus ? grid : cus;
    if (a.stats) hipLaunchKernelGGL(k_ffn_pair<true>, dim3(pgrid), dim3(512), PAIR_LDS_BYTES, stream, a, (int)grid);
    else hipLaunchKernelGGL(k_ffn_pair<false>, dim3(pgrid), dim3(512), PAIR_LDS_BYTES, stream, a, (int)grid);
  } else if (a.stats) hipLaunchKernelGGL(k_ffn_chain<true>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(k_ffn_chain<false>, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
