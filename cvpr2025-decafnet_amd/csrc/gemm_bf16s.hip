// fp32-accurate GEMM on the 16-bit matrix cores by operand splitting: "f16x3" (default, see below) and "bf16x6".
//
// Every fp32 operand x is written exactly as hi + mid + lo with three bf16 values (8 significant
// bits each).  A product a*b is then the sum of 9 bf16 x bf16 products, each EXACT in fp32; the six
// with i + j <= 2 (hh, hm, mh, hl, lh, mm) carry everything down to 2^-24 |ab| -- the same size as one
// fp32 rounding -- and are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Measured against an fp64
// reference the result is slightly MORE accurate than a plain fp32 FMA chain (rms 1.5e-7 vs 3.4e-7 at
// K = 1024, tools/split_numerics.py), i.e. this is not a reduced-precision mode.
//
// Why: the fp32 MFMA (v_mfma_f32_32x32x2_f32) retires 2 k per 64 cycles = 32 cycles per k-step of a
// 32x32 tile; the bf16 MFMA retires 16 k per 32 cycles, six of them 12 cycles per k -- 2.67x the rate
// (fp32-equivalent peak 2.5 PFLOP/s / 6 = 417 TFLOP/s against 157).  (A two-plane bf16 mode -- hh, hm, mh, error
// ~2^-16 per product -- existed until the fp16 two-plane mode below replaced it: same MFMA count, 15x the accuracy.)
//
// Weights are split once at model finalisation ([3][N][K] bf16 planes); activations are split while
// they are staged global -> LDS (v_cvt_pk_bf16_f32, ~6 VALU ops per element, hidden under the MFMAs).
// LDS tiles: [plane][row][32 k] bf16 with an 80-byte row pitch (20 dwords = 4 * 5: every 16-lane
// ds_read_b128 group covers 64 distinct banks).  Lane (r = lane & 31, h = lane >> 5) reads the 8
// consecutive k values 8h..8h+7 of row r of a 16-wide chunk with one ds_read_b128 per plane.
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#include "gemm_common.h"


namespace dcf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int SBK = 32;            // k per LDS tile
constexpr int ROWB = 80;           // bytes per (plane, row): 64 data + 16 pad

// ---- "f16x3": the same idea on the fp16 matrix cores with TWO planes per operand -------------------------------
// fp16 carries 11 significant bits, so x = hi + lo with two fp16 values holds 22 bits and the three products
// hh, hl, lh (each exact in fp32) carry a*b down to ~2^-22 |ab|: measured against fp64 the result has the error of a plain
// fp32 FMA chain (rms 2.9e-7 at K = 256, 3.4e-7 at K = 1024, both for the native fp32 path and for this one; bf16x3 is
// 4.4e-6) at HALF the MFMA work of bf16x6 and 2/3 of its operand bytes.  What fp16 lacks is exponent range, so operands
// are pre-scaled by exact powers of two -- activations by 2^4 (GemmArgs::a_scale overrides: the vid_map GEMMs on the raw
// feature files use 1, i.e. |a| < 65504), weights by 2^8 -- and the accumulators un-scaled by 2^-12
// in the epilogue: |a| < 4094 and |w| < 255.9 convert without overflow, the low planes stay normal fp16 numbers for
// |a| >= 2^-7 / |w| >= 2^-11, and below that the absolute representation error is <= 2^-29 / 2^-33 (an fp32 ulp of 1.0
// is 2^-23).  Out-of-range operands give inf/NaN accumulators: the epilogue raises the sticky status word the caller
// passes in GemmArgs::status instead of letting a later ReLU / max swallow them.  Weights are range-checked once when
// they are split (the model falls back to bf16x6 if they do not fit).
constexpr int T_F16 = 16;          // `nterms` code of this mode (6 and 3 are the bf16 modes)
constexpr float F16_SA = 16.f, F16_SW = 256.f, F16_UNSCALE = 1.f / 4096.f;

// fp16 mode: hi = fp16(s x), "mid" = fp16(s x - hi); lo unused
// split two floats into packed bf16 pairs: hi, mid, lo (round to nearest even at every level)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  bf16x2 m = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
  mid = __builtin_bit_cast(unsigned, m);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  bf16x2 l = __builtin_convertvector(f32x2{s0, s1}, bf16x2);
  lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ void split2_f16(float x0, float x1, float s, unsigned& hi, unsigned& mid) {
  const f16x2 h = __builtin_convertvector(f32x2{x0 * s, x1 * s}, f16x2);          // v_cvt_pk_f16_f32 (RTNE)
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = __builtin_fmaf(x0, s, -(float)h[0]), r1 = __builtin_fmaf(x1, s, -(float)h[1]);   // exact residuals
  mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}
// the A operand split of mode NTERMS
template <int NTERMS>
__device__ __forceinline__ void split_a(float x0, float x1, float sa, unsigned& hi, unsigned& mid, unsigned& lo) {
  if constexpr (NTERMS == T_F16) { split2_f16(x0, x1, sa, hi, mid); lo = 0u; }
  else split2(x0, x1, hi, mid, lo);
}
// activation pre-scale of a launch: 2^4 unless the caller knows better (GemmArgs::a_scale, a power of two)
__device__ __forceinline__ float a_scale_of(const GemmArgs& p) { return p.a_scale > 0.f ? p.a_scale : F16_SA; }
template <int NTERMS>
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (NTERMS == T_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// f16 mode: un-scale a finished accumulator fragment and raise the status word on a non-finite value
template <int NTERMS>
__device__ __forceinline__ void finish_acc(f32x16& acc, float unscale, bool& bad) {
  if constexpr (NTERMS == T_F16) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = acc[e] * unscale;
      bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
      acc[e] = v;
    }
  }
}

// Weight image = MFMA B fragments in the order the kernel consumes them.  One 6 KiB block per (32 output
// rows n32, 32-wide K tile kt), blocks ordered [n32][kt]; inside a block
//   [16-wide K chunk c = 0,1][plane][lane = h * 32 + r][8 bf16]  =  W[n32*32 + r][kt*32 + c*16 + h*8 .. +7]
// so ONE fully coalesced 1 KiB wave load (global_load_dwordx4) delivers the bf16x8 B operand of every lane
// for one (chunk, plane).  W never touches LDS: staging the 60 KiB W tile per K step through ds_write_b128
// (~79 B/clk/CU) was the co-bottleneck of the first version of this kernel (MFMA busy 25 %).
template <bool F16>
__global__ void k_split_planes(const float* __restrict__ W, unsigned short* __restrict__ out, int N, int K, int64_t ldw,
                               unsigned* __restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // pair index
  const int64_t pairs = (int64_t)N * K / 2;
  if (i >= pairs) return;
  const int n = (int)(i / (K / 2)), k = (int)(i % (K / 2)) * 2;
  unsigned hi, mid, lo = 0u;
  const float w0 = W[n * ldw + k], w1 = W[n * ldw + k + 1];
  if constexpr (F16) {
    split2_f16(w0, w1, F16_SW, hi, mid);
    if (!(__builtin_fabsf(w0) * F16_SW <= 65504.f) || !(__builtin_fabsf(w1) * F16_SW <= 65504.f)) {
      if (overflow) atomicOr(overflow, 1u);
    }
  } else {
    split2(w0, w1, hi, mid, lo);
  }
  const int64_t blk = (int64_t)(n >> 5) * (K >> 5) + (k >> 5);
  const int kk = k & 31, c = kk >> 4, h = (kk >> 3) & 1, j = kk & 7, r = n & 31;
  const int64_t o = blk * (2 * 3 * 64 * 8) + ((int64_t)(c * 3) * 64 + h * 32 + r) * 8 + j;
  *reinterpret_cast<unsigned*>(out + o) = hi;
  *reinterpret_cast<unsigned*>(out + o + 64 * 8) = mid;
  *reinterpret_cast<unsigned*>(out + o + 2 * 64 * 8) = lo;
}

int launch_split_planes(const float* W, unsigned short* out, int N, int K, int64_t ldw, hipStream_t st, int nterms,
                        unsigned* overflow) {
  DCF_CHECK(K % 32 == 0 && N % 32 == 0, "split_planes: N and K must be multiples of 32");
  const int64_t pairs = (int64_t)N * K / 2;
  if (nterms == T_F16)
    hipLaunchKernelGGL(k_split_planes<true>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, W, out, N, K, ldw, overflow);
  else
    hipLaunchKernelGGL(k_split_planes<false>, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, W, out, N, K, ldw, overflow);
  DCF_HIP(hipGetLastError());
  return 0;
}

// 3 waves per SIMD (<= 168 registers): measured 3.16 -> 3.00 ms per step over the compiler's default of 2
// LN = true: the variant with the fused LayerNorm epilogue (its own kernel so that the extra scalars and the
// statistics registers do not cost the plain kernel its third wave per SIMD)
// NST = depth of the register pipeline: the global loads of K tile kt + NST - 1 (A raw fp32, B fragments) are issued
// during step kt.  One K step is only 12 (64x32 wave tile, f16x3) .. 48 MFMAs = 400 .. 1500 cycles while an L2 /
// Infinity Cache hit takes ~1000 and HBM ~2000+, and at T = 16384 most grids give one wave per SIMD, so nothing but
// the prefetch distance hides that latency.  Buffers are indexed statically (steps unrolled in groups of NST).
// STATS = true: the instantiation whose epilogue writes / consumes row statistics (GemmArgs::stats_out / stats_in)
// ALN = true (A_ROWS_TAP3): the A rows get LayerNorm + ReLU while they are staged (GemmArgs::a_stats)
// SCORE = true (A_CHANMAJOR): the sidekick scores of the rows on the side (GemmArgs::score_out)
template <int WM, int WN, int TM, int TN, int AMODE, int NTERMS, bool LN = false, int NST = 3, bool STATS = false, bool ALN = false, bool SCORE = false>
__global__ __launch_bounds__(WM * WN * 64, ((TM * TN >= 5 || TM >= 4 || (NST > 2 && TM * TN >= 4)) ? 2 : 3)) void gemm_bf16s_kernel(GemmBatch batch) {
  constexpr int NT = WM * WN * 64;                    // 4 or 8 wavefronts
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int NPL = NTERMS == 6 ? 3 : 2;            // planes used: hi, mid (, lo); T_F16: the two fp16 planes
  constexpr int ACH = (BM * 4 + NT - 1) / NT;         // 8-float chunks of the A tile per thread (the last may be partial)
  constexpr bool APART = (BM * 4) % NT != 0;          // 3-wave workgroups: chunk ids >= BM * 4 do not exist
  constexpr int BLK = 2 * 3 * 64 * 8;                 // bf16 elements of one weight block (6 KiB)
  static_assert(WM * WN == 2 || WM * WN == 3 || WM * WN == 4 || WM * WN == 6 || WM * WN == 8, "2, 3, 4, 6 or 8 wavefronts");
  static_assert(!(APART && AMODE == A_CHANMAJOR), "channel-major A needs a whole number of chunks per thread");
  static_assert(AMODE == A_ROWS || AMODE == A_ROWS_TAP3 || AMODE == A_CHANMAJOR, "unknown A mode");

  extern __shared__ unsigned char smem_b[];
  unsigned char* As = smem_b;                          // [NPL][BM][ROWB]: the only LDS tile (A is shared by the N-waves)
  // row-statistics block of the workgroup: behind the A tile and behind the epilogue's transpose tiles
  constexpr int WG_OFF = (NPL * BM * ROWB > WM * WN * EPI_WAVE_FLOATS * 4 ? NPL * BM * ROWB : WM * WN * EPI_WAVE_FLOATS * 4);
  float* wg_stats = reinterpret_cast<float*>(smem_b + WG_OFF);

  GemmArgs p = (LN || batch.zcount > 0) ? batch.g[0] : (blockIdx.z == 0 ? batch.g[0] : (blockIdx.z == 1 ? batch.g[1] : batch.g[2]));
  if constexpr (AMODE == A_CHANMAJOR && !SCORE && !LN) {
    if (batch.zcount > 0) {                             // uniform: operand set blockIdx.z of the one GEMM (GemmBatch::zcount)
      const int z = blockIdx.z;
      p.A = gemm_zsel(batch.zA, z); p.C = gemm_zsel(batch.zC, z); p.tile_skip = gemm_zsel(batch.zskip, z); p.skip_nq = gemm_zsel(batch.zskip_nq, z);
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  int m0, n0;
  if (!tile_origin<BM, BN>(p, m0, n0)) return;
  const int M = p.M, K = p.K;
  if constexpr (AMODE == A_CHANMAJOR && BM % 64 == 0 && !SCORE) {
    if (p.tile_skip) {                                 // uniform: row tiles no query of this video keeps (GemmArgs::tile_skip)
      const int f0 = m0 >> 6, f1 = min((m0 + BM + 63) >> 6, (M + 63) >> 6);
      unsigned any = 0;
      for (int q = 0; q < p.skip_nq; ++q)
        for (int f = f0; f < f1; ++f) any |= p.tile_skip[(int64_t)q * p.skip_stride + f];
      if (!any) return;
    }
  }
  const int KT = K / SBK;
  const float sa = a_scale_of(p);
  if constexpr (STATS) { if (p.stats_in) stats_load<BM, NT>(p, wg_stats, m0, tid); }     // uniform
  // ALN: per-row (rstd, -mean * rstd) of the A rows m0 - 1 .. m0 + BM (the k3 taps reach one row beyond the tile on both
  // sides) and the LayerNorm parameters of the cin channels, in LDS behind the statistics block; written here, read from
  // the first store_a on, which is behind the first barrier of the K loop
  float* aln_row = wg_stats + stats_lds_floats<WM, WN, TM>();
  float* aln_g = aln_row + 2 * (BM + 2);
  if constexpr (ALN) {
    static_assert(AMODE == A_ROWS_TAP3, "ALN is a k3-convolution mode");
    const float inv = 1.0f / (float)p.cin;
    for (int t = tid; t < BM + 2; t += NT) {
      int row = m0 - 1 + t;
      row = row < 0 ? 0 : (row < M ? row : M - 1);
      const float* sp = p.a_stats + (int64_t)row * p.a_stats_slots * 2;
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < p.a_stats_slots; ++k) { s1 += sp[2 * k]; s2 += sp[2 * k + 1]; }
      const float mean = s1 * inv;
      const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv), 0.f);
      const float rstd = 1.0f / sqrtf(var + 1e-5f);
      if (ln_ill(mean, var) && p.status) atomicOr(p.status, 2u);            // (common.h LN_ILL_RATIO)
      aln_row[2 * t] = rstd;
      aln_row[2 * t + 1] = -mean * rstd;
    }
    for (int c = tid; c < p.cin; c += NT) { aln_g[c] = p.a_ln_g[c]; aln_g[p.cin + c] = p.a_ln_b[c]; }
  }
  // SCORE: the text vectors in LDS behind the statistics block (read from the first store_a on, behind the K loop's first barrier);
  // per thread the partial sums of its 4 rows over its k pairs: sum of squares + one dot product per query
  static_assert(!SCORE || (AMODE == A_CHANMAJOR && !ALN && !LN), "SCORE is a channel-major mode");
  float* sc_tn = wg_stats + stats_lds_floats<WM, WN, TM>();
  const bool do_score = SCORE && p.score_out != nullptr && n0 == 0;         // uniform: the first column tile carries the scores
  f32x4 sc_ss[ACH], sc_dot[ACH][GEMM_SCORE_MAXQ];
  if constexpr (SCORE) {
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      sc_ss[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < GEMM_SCORE_MAXQ; ++q) sc_dot[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (do_score)
      for (int c = tid; c < p.score_nq * K; c += NT) sc_tn[c] = p.score_tn[c];
  }

  // K-invariant addressing (see gemm.hip: nothing but 16-byte loads inside the K loop)
  // A_ROWS / TAP3: thread -> (row, 8-float piece); A_CHANMAJOR (A[m][k] = X[k*lda + m]): thread -> (k pair p, 4 rows)
  const float* a_ptr[ACH];
  unsigned a_flag[ACH];
#pragma unroll
  for (int i = 0; i < ACH; ++i) {
    const int id = i * NT + tid;
    if constexpr (AMODE == A_CHANMAJOR) {
      const int pk = id & 15, m4 = id >> 4;
      const int m = m0 + m4 * 4;
      a_flag[i] = m < M ? 1u : 0u;                     // M % 4 == 0 (checked by the launcher)
      a_ptr[i] = p.A + (int64_t)(2 * pk) * p.lda + (m < M ? m : 0);
    } else {
      const int row = id >> 2, c8 = id & 3;
      const int m = m0 + row;
      unsigned f = 0;
      if (m < M && (!APART || id < BM * 4)) {
        if constexpr (AMODE == A_ROWS) f = (p.flags & G_AMASK) ? (p.rowmask[m] ? 1u : 0u) : 1u;
        else f = p.nbr[m];
      }
      a_flag[i] = f;
      a_ptr[i] = p.A + (int64_t)(m < M ? m : 0) * p.lda + c8 * 8;
    }
  }
  // B fragments of this wave's TN column tiles: block (n32, kt) at ((n32 * KT + kt) * BLK), lane slot lane*8
  const bf16x8* w_ptr[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
    w_ptr[j] = reinterpret_cast<const bf16x8*>(p.Ws + ((int64_t)(n0 / 32 + wn * TN + j) * KT) * BLK) + lane;

  // k3 convolutions walk K channel slab by channel slab with the three taps innermost (physical K tile = tap * cin / 32 + slab):
  // the taps read rows m - 1, m, m + 1 of the SAME 128-byte channel slab back to back, so two of the three reads hit L1 / L2.
  // Tap-major order re-read the whole row tile three times, 9 K tiles apart: the PMC passes showed 3.3x the algorithmic
  // A bytes on the HBM side of L2 for these kernels (profiles/r03_notes.md).
  // Interleaved A/B on one box (8 x 16384 rows, us per launch, tap-major -> slab-major): 128x96 pair launch 878 -> 854,
  // 128x256 tile 261120x256x768 307 -> 303 and 131072x256x768 156.5 -> 153.3.
  const bool slab_major = AMODE == A_ROWS_TAP3 && (p.flags & G_TAPSLAB);      // uniform
  const int slabs = slab_major ? p.cin / SBK : 1;
  auto phys_kt = [&](int kt) __attribute__((always_inline)) -> int {
    if (slab_major) { const int slab = kt / 3, tap = kt - 3 * slab; return tap * slabs + slab; }
    return kt;
  };
  f32x4 araw_[NST][ACH][2];
  auto load_a = [&](int kt_, f32x4 (&araw)[ACH][2]) __attribute__((always_inline)) {
    const int k0 = phys_kt(kt_) * SBK;
    int64_t shift = k0;
    unsigned bit = 1u;
    if constexpr (AMODE == A_ROWS_TAP3) {
      const int tap = slab_major ? kt_ % 3 : k0 / p.cin;
      bit = tap == 0 ? 2u : (tap == 1 ? 1u : 4u);
      shift = (int64_t)(tap - 1) * p.lda + (k0 - tap * p.cin);
    }
    if constexpr (AMODE == A_CHANMAJOR) shift = (int64_t)k0 * p.lda;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (a_flag[i] & bit) {
        v0 = *reinterpret_cast<const f32x4*>(a_ptr[i] + shift);
        if constexpr (AMODE == A_CHANMAJOR) v1 = *reinterpret_cast<const f32x4*>(a_ptr[i] + shift + p.lda);   // k + 1
        else v1 = *reinterpret_cast<const f32x4*>(a_ptr[i] + shift + 4);
      }
      araw[i][0] = v0;
      araw[i][1] = v1;
    }
  };
  auto store_a = [&](const f32x4 (&araw_in)[ACH][2], int kt_) __attribute__((always_inline)) {
    f32x4 araw[ACH][2];
#pragma unroll
    for (int i = 0; i < ACH; ++i) { araw[i][0] = araw_in[i][0]; araw[i][1] = araw_in[i][1]; }
    if constexpr (ALN) {
      // y = relu((x - mean) rstd g + beta) on the rows that exist (a masked / out-of-sequence tap stays 0: the convolution sees
      // relu(LN(x)) * mask, blocks.py:98-99); explicit fmaf: the same bits in every tile instantiation
      const int slab = slab_major ? kt_ / 3 : (kt_ * SBK % p.cin) / SBK;
      const int tap = slab_major ? kt_ - 3 * slab : kt_ * SBK / p.cin;
      const unsigned bit = tap == 0 ? 2u : (tap == 1 ? 1u : 4u);
      const int c0 = slab * SBK + (tid & 3) * 8;             // id & 3 == tid & 3: NT is a multiple of 4
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(aln_g + c0), g1 = *reinterpret_cast<const f32x4*>(aln_g + c0 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(aln_g + p.cin + c0), b1 = *reinterpret_cast<const f32x4*>(aln_g + p.cin + c0 + 4);
#pragma unroll
      for (int i = 0; i < ACH; ++i) {
        const int id = i * NT + tid;
        if (APART && id >= BM * 4) continue;
        if (!(a_flag[i] & bit)) continue;
        const int row = id >> 2;
        const float rs = aln_row[2 * (row + tap)], sh = aln_row[2 * (row + tap) + 1];
        f32x4 x0 = araw[i][0], x1 = araw[i][1];
        x0.x = fmaxf(__builtin_fmaf(__builtin_fmaf(x0.x, rs, sh), g0.x, b0.x), 0.f);
        x0.y = fmaxf(__builtin_fmaf(__builtin_fmaf(x0.y, rs, sh), g0.y, b0.y), 0.f);
        x0.z = fmaxf(__builtin_fmaf(__builtin_fmaf(x0.z, rs, sh), g0.z, b0.z), 0.f);
        x0.w = fmaxf(__builtin_fmaf(__builtin_fmaf(x0.w, rs, sh), g0.w, b0.w), 0.f);
        x1.x = fmaxf(__builtin_fmaf(__builtin_fmaf(x1.x, rs, sh), g1.x, b1.x), 0.f);
        x1.y = fmaxf(__builtin_fmaf(__builtin_fmaf(x1.y, rs, sh), g1.y, b1.y), 0.f);
        x1.z = fmaxf(__builtin_fmaf(__builtin_fmaf(x1.z, rs, sh), g1.z, b1.z), 0.f);
        x1.w = fmaxf(__builtin_fmaf(__builtin_fmaf(x1.w, rs, sh), g1.w, b1.w), 0.f);
        araw[i][0] = x0; araw[i][1] = x1;
      }
    }
    if constexpr (SCORE) {
      if (do_score) {
        const int kk = phys_kt(kt_) * SBK + 2 * (tid & 15);            // pk = id & 15 = tid & 15 (NT is a multiple of 16)
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
          const f32x4 x0 = araw[i][0], x1 = araw[i][1];                // rows m .. m + 3 at k and k + 1 (zeros beyond M)
#pragma unroll
          for (int e = 0; e < 4; ++e) sc_ss[i][e] = __builtin_fmaf(x1[e], x1[e], __builtin_fmaf(x0[e], x0[e], sc_ss[i][e]));
#pragma unroll
          for (int q = 0; q < GEMM_SCORE_MAXQ; ++q) {
            if (q < p.score_nq) {                                      // uniform
              const float2 t2 = *reinterpret_cast<const float2*>(sc_tn + q * K + kk);
#pragma unroll
              for (int e = 0; e < 4; ++e) sc_dot[i][q][e] = __builtin_fmaf(x1[e], t2.y, __builtin_fmaf(x0[e], t2.x, sc_dot[i][q][e]));
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int id = i * NT + tid;
      if constexpr (AMODE == A_CHANMAJOR) {
        // araw[i][0] = rows m..m+3 at k = 2 pk, araw[i][1] the same rows at k + 1: one packed bf16 pair per row and
        // plane.  Lanes run over pk first: the 32 lanes of a ds_write_b32 group hit 32 distinct banks.
        const int pk = id & 15, m4 = id >> 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned hi, mid, lo;
          split_a<NTERMS>(araw[i][0][e], araw[i][1][e], sa, hi, mid, lo);
          const int row = m4 * 4 + e;
          *reinterpret_cast<unsigned*>(As + (0 * BM + row) * ROWB + pk * 4) = hi;
          *reinterpret_cast<unsigned*>(As + (1 * BM + row) * ROWB + pk * 4) = mid;
          if constexpr (NPL == 3) *reinterpret_cast<unsigned*>(As + (2 * BM + row) * ROWB + pk * 4) = lo;
        }
      } else {
        if (APART && id >= BM * 4) continue;
        const int row = id >> 2, c8 = id & 3;
        unsigned h0, h1, h2, h3, m0_, m1, m2, m3, l0, l1, l2, l3;
        split_a<NTERMS>(araw[i][0].x, araw[i][0].y, sa, h0, m0_, l0);
        split_a<NTERMS>(araw[i][0].z, araw[i][0].w, sa, h1, m1, l1);
        split_a<NTERMS>(araw[i][1].x, araw[i][1].y, sa, h2, m2, l2);
        split_a<NTERMS>(araw[i][1].z, araw[i][1].w, sa, h3, m3, l3);
        const u32x4 hi = {h0, h1, h2, h3}, mid = {m0_, m1, m2, m3}, lo = {l0, l1, l2, l3};
        *reinterpret_cast<u32x4*>(As + (0 * BM + row) * ROWB + c8 * 16) = hi;
        *reinterpret_cast<u32x4*>(As + (1 * BM + row) * ROWB + c8 * 16) = mid;
        if constexpr (NPL == 3) *reinterpret_cast<u32x4*>(As + (2 * BM + row) * ROWB + c8 * 16) = lo;
      }
    }
  };
  // B fragments of one K tile: [chunk][tile][plane]
  bf16x8 bfr_[NST][2][TN][NPL];
  auto load_b = [&](int kt_, bf16x8 (&bfr)[2][TN][NPL]) __attribute__((always_inline)) {
    const int kt = phys_kt(kt_);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bfr[c][j][pl] = w_ptr[j][(int64_t)kt * (BLK / 8) + (c * 3 + pl) * 64];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // one K step: tile kt is in stage S of the register buffers; stage (S + NST - 1) % NST (consumed by step kt - 1) takes
  // the loads of tile kt + NST - 1 (clamped to the last tile: a redundant load instead of a branch)
  auto step = [&](int kt, auto stage) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value;
    constexpr int SN = (S + NST - 1) % NST;
    __syncthreads();                    // every wave finished reading As (tile kt-1)
    store_a(araw_[S], kt);
    __syncthreads();
    const int kn = kt + NST - 1 < KT ? kt + NST - 1 : KT - 1;
    load_a(kn, araw_[SN]);
    load_b(kn, bfr_[SN]);
#pragma unroll
    for (int c = 0; c < SBK / 16; ++c) {
      bf16x8 a[TM][NPL];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          a[i][pl] = *reinterpret_cast<const bf16x8*>(As + (pl * BM + (wm * TM + i) * 32 + r) * ROWB + c * 32 + h * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          // smallest terms first
          if constexpr (NTERMS == 6) {
            acc[i][j] = mma<NTERMS>(a[i][1], bfr_[S][c][j][1], acc[i][j]);
            acc[i][j] = mma<NTERMS>(a[i][0], bfr_[S][c][j][2], acc[i][j]);
            acc[i][j] = mma<NTERMS>(a[i][2], bfr_[S][c][j][0], acc[i][j]);
          }
          acc[i][j] = mma<NTERMS>(a[i][0], bfr_[S][c][j][1], acc[i][j]);
          acc[i][j] = mma<NTERMS>(a[i][1], bfr_[S][c][j][0], acc[i][j]);
          acc[i][j] = mma<NTERMS>(a[i][0], bfr_[S][c][j][0], acc[i][j]);
        }
      }
    }
  };
  auto group = [&](int kt, int count) __attribute__((always_inline)) {   // count == NST in the main loop
    if constexpr (NST >= 1) { if (count > 0) step(kt + 0, std::integral_constant<int, 0>{}); }
    if constexpr (NST >= 2) { if (count > 1) step(kt + 1, std::integral_constant<int, 1 % NST>{}); }
    if constexpr (NST >= 3) { if (count > 2) step(kt + 2, std::integral_constant<int, 2 % NST>{}); }
    if constexpr (NST >= 4) { if (count > 3) step(kt + 3, std::integral_constant<int, 3 % NST>{}); }
  };
#pragma unroll
  for (int s_ = 0; s_ < NST - 1; ++s_) {
    const int k_ = s_ < KT ? s_ : KT - 1;
    load_a(k_, araw_[s_]);
    load_b(k_, bfr_[s_]);
  }
  int kt = 0;
  for (; kt + NST <= KT; kt += NST) {
    step(kt + 0, std::integral_constant<int, 0>{});
    if constexpr (NST >= 2) step(kt + 1, std::integral_constant<int, 1 % NST>{});
    if constexpr (NST >= 3) step(kt + 2, std::integral_constant<int, 2 % NST>{});
    if constexpr (NST >= 4) step(kt + 3, std::integral_constant<int, 3 % NST>{});
  }
  group(kt, KT - kt);                   // the last KT % NST tiles
  if constexpr (SCORE) {
    if (do_score) {
      // the 16 lanes of a k-pair group hold the partial sums of the same 4 rows: butterfly over them (fixed order), lane 0 writes
#pragma unroll
      for (int i = 0; i < ACH; ++i) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) sc_ss[i][e] += __shfl_xor(sc_ss[i][e], off);
#pragma unroll
          for (int q = 0; q < GEMM_SCORE_MAXQ; ++q)
            if (q < p.score_nq) {
#pragma unroll
              for (int e = 0; e < 4; ++e) sc_dot[i][q][e] += __shfl_xor(sc_dot[i][q][e], off);
            }
        }
        const int id = i * NT + tid;
        const int m = m0 + (id >> 4) * 4;
        if ((id & 15) == 0 && m < M) {
          f32x4 inv = {1.f, 1.f, 1.f, 1.f};
          if (p.score_norm) {
#pragma unroll
            for (int e = 0; e < 4; ++e) inv[e] = 1.0f / (sqrtf(sc_ss[i][e]) + 1e-4f);
          }
#pragma unroll
          for (int q = 0; q < GEMM_SCORE_MAXQ; ++q)
            if (q < p.score_nq) *reinterpret_cast<f32x4*>(p.score_out + (int64_t)q * M + m) = sc_dot[i][q] * inv;
        }
      }
    }
  }
  if constexpr (NTERMS == T_F16) {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) finish_acc<NTERMS>(acc[i][j], 1.f / (sa * F16_SW), bad);
    if (bad && p.status) atomicOr(p.status, 1u);
  }
  if constexpr (LN) {                                   // tile spans the whole output row (n0 = 0)
    __syncthreads();                                    // the A tile is dead: its LDS becomes the row-statistics scratch
    gemm_epilogue_ln<WN, TM, TN>(p, acc, m0, wn, lane, reinterpret_cast<float*>(smem_b));
  } else {
    if (p.flags & G_ADALN) {                            // uniform
      __syncthreads();
      gemm_epilogue_adaln<WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem_b) + wave * EPI_WAVE_FLOATS, wg_stats);
    } else if (gemm_wide_ok(p)) {                       // uniform
      __syncthreads();                                  // the A tile is dead: every wave takes a private transpose tile in it
      gemm_epilogue_wide<WM, WN, TM, TN, STATS>(p, acc, m0, n0, wm, wn, lane, reinterpret_cast<float*>(smem_b) + wave * EPI_WAVE_FLOATS, wg_stats);
    } else {
      gemm_epilogue<WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, r, h);
    }
  }
}

// ---- k-sliced variant for grids that cannot fill the chip --------------------------------------------------
// The upper pyramid levels (128 .. 4096 rows) give at most a few hundred 64x64 tiles, one workgroup per CU or
// less, and the kernel above then runs one K tile per ~1.2 us: nothing but the latency of the loads issued one
// tile earlier (12 MFMAs of work per wave in between).  Here the four waves of a workgroup split K instead of
// the tile: each wave owns the whole 64x64 tile for a quarter of the K tiles, loads its A fragments straight
// from global memory (8 consecutive k per lane = the MFMA A operand, no LDS, no barrier in the loop) and its B
// fragments from the weight image, so four independent load streams with 48 MFMAs per tile are in flight per
// workgroup and the K loop is 4x shorter.  The partial tiles are summed through LDS in a fixed order (wave 0, 1,
// 2, 3 -- deterministic), wave q finishing quadrant q with the common epilogue.
// TM = 2: 64x64 tile; TM = 1: 32x64 tile (twice the workgroups, half the MFMAs per wave: the very small levels)
// KS = number of waves = K slices (4, or 8 for the 32-row tile with K >= 512)
template <int NTERMS, int TM, int KS = 4>
__global__ __launch_bounds__(KS * 64) void gemm_bf16s_kslice_kernel(GemmBatch batch) {
  constexpr int NPL = NTERMS == 6 ? 3 : 2;
  constexpr int BLK = 2 * 3 * 64 * 8;
  constexpr int NF = TM * 2;                           // 32x32 fragments of the tile = owner waves
  __shared__ f32x4 red[NF][KS - 1][4][64];             // [owner fragment][source slot][quarter][lane], 48 / 24 / 56 KiB

  const GemmArgs p = blockIdx.z == 0 ? batch.g[0] : (blockIdx.z == 1 ? batch.g[1] : batch.g[2]);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int m0, n0;
  if (!tile_origin<TM * 32, 64>(p, m0, n0)) return;
  const int M = p.M;
  const int KT = p.K / SBK;
  const float sa = a_scale_of(p);
  const int kb = KT * wave / KS, ke = KT * (wave + 1) / KS;

  const float* a_ptr[TM];
  bool a_ok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + i * 32 + r;
    a_ok[i] = m < M && (!(p.flags & G_AMASK) || p.rowmask[m]);
    a_ptr[i] = p.A + (int64_t)(m < M ? m : 0) * p.lda + h * 8;
  }
  const bf16x8* w_ptr[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) w_ptr[j] = reinterpret_cast<const bf16x8*>(p.Ws + ((int64_t)(n0 / 32 + j) * KT) * BLK) + lane;

  f32x4 araw[TM][2][2];                                 // [row tile][chunk][half]
  bf16x8 bfr[2][2][NPL];                                // [chunk][col tile][plane]
  auto load = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[i]) {
          v0 = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * SBK + c * 16);
          v1 = *reinterpret_cast<const f32x4*>(a_ptr[i] + kt * SBK + c * 16 + 4);
        }
        araw[i][c][0] = v0;
        araw[i][c][1] = v1;
      }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bfr[c][j][pl] = w_ptr[j][(int64_t)kt * (BLK / 8) + (c * 3 + pl) * 64];
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kb < ke) load(kb);
  for (int kt = kb; kt < ke; ++kt) {
    bf16x8 a[TM][2][NPL];                               // [row tile][chunk][plane]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        unsigned h0, h1, h2, h3, m0_, m1, m2, m3, l0, l1, l2, l3;
        split_a<NTERMS>(araw[i][c][0].x, araw[i][c][0].y, sa, h0, m0_, l0);
        split_a<NTERMS>(araw[i][c][0].z, araw[i][c][0].w, sa, h1, m1, l1);
        split_a<NTERMS>(araw[i][c][1].x, araw[i][c][1].y, sa, h2, m2, l2);
        split_a<NTERMS>(araw[i][c][1].z, araw[i][c][1].w, sa, h3, m3, l3);
        a[i][c][0] = __builtin_bit_cast(bf16x8, (u32x4){h0, h1, h2, h3});
        a[i][c][1] = __builtin_bit_cast(bf16x8, (u32x4){m0_, m1, m2, m3});
        if constexpr (NPL == 3) a[i][c][2] = __builtin_bit_cast(bf16x8, (u32x4){l0, l1, l2, l3});
      }
    bf16x8 bc[2][2][NPL];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) bc[c][j][pl] = bfr[c][j][pl];
    if (kt + 1 < ke) load(kt + 1);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (NTERMS == 6) {
            acc[i][j] = mma<NTERMS>(a[i][c][1], bc[c][j][1], acc[i][j]);
            acc[i][j] = mma<NTERMS>(a[i][c][0], bc[c][j][2], acc[i][j]);
            acc[i][j] = mma<NTERMS>(a[i][c][2], bc[c][j][0], acc[i][j]);
          }
          acc[i][j] = mma<NTERMS>(a[i][c][0], bc[c][j][1], acc[i][j]);
          acc[i][j] = mma<NTERMS>(a[i][c][1], bc[c][j][0], acc[i][j]);
          acc[i][j] = mma<NTERMS>(a[i][c][0], bc[c][j][0], acc[i][j]);
        }
  }

  // hand the fragments this wave does not finish to their owners (waves 0 .. NF-1)
#pragma unroll
  for (int q = 0; q < NF; ++q) {
    if (q == wave) continue;
    const int slot = wave < q ? wave : wave - 1;
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const f32x16& t = acc[q >> 1][q & 1];
      red[q][slot][e4][lane] = f32x4{t[e4 * 4 + 0], t[e4 * 4 + 1], t[e4 * 4 + 2], t[e4 * 4 + 3]};
    }
  }
  __syncthreads();
  if (wave >= NF) return;
  f32x16 own;
  if constexpr (TM == 2) own = wave == 0 ? acc[0][0] : (wave == 1 ? acc[0][1] : (wave == 2 ? acc[1][0] : acc[1][1]));
  else own = wave == 0 ? acc[0][0] : acc[0][1];
  f32x16 fin[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) fin[0][0][e] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {                         // fixed order: wave 0, 1, 2, ...
    if (s == wave) {
#pragma unroll
      for (int e = 0; e < 16; ++e) fin[0][0][e] += own[e];
    } else {
      const int slot = s < wave ? s : s - 1;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 t = red[wave][slot][e4][lane];
        fin[0][0][e4 * 4 + 0] += t.x; fin[0][0][e4 * 4 + 1] += t.y; fin[0][0][e4 * 4 + 2] += t.z; fin[0][0][e4 * 4 + 3] += t.w;
      }
    }
  }
  if constexpr (NTERMS == T_F16) {
    bool bad = false;
    finish_acc<NTERMS>(fin[0][0], 1.f / (sa * F16_SW), bad);
    if (bad && p.status) atomicOr(p.status, 1u);
  }
  gemm_epilogue<TM, 2, 1, 1>(p, fin, m0, n0, wave >> 1, wave & 1, r, h);
}

template <int WM, int WN, int TM, int TN>
static int launch_cfg_s(const GemmBatch& b, int count, GemmAMode mode, int nterms, hipStream_t stream, double work_fraction = 1.0) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const GemmArgs& p = b.g[0];
  DCF_CHECK(!(p.flags & G_ADALN) || TN % 2 == 0, "launch_gemm_split: G_ADALN needs a tile whose waves span 64 columns (got %dx%d)", BM, BN);
  dim3 grid(tile_grid<BM, BN>(p), 1, count);
  char name[96];
  static const bool shapes = getenv("DCF_PROF_SHAPES") != nullptr;   // per-shape labels for tools/ (not used by bench.py)
  const char* fam = nterms == 6 ? "gemm_bf16x6" : "gemm_f16x3";
  if (shapes) snprintf(name, sizeof(name), "%s<%dx%d,%s>[%dx%dx%dx%d]", fam, BM, BN, mode == A_ROWS ? "rows" : mode == A_ROWS_TAP3 ? "tap3" : "chanmajor", count, p.M, p.N, p.K);
  else snprintf(name, sizeof(name), "%s<%dx%d,%s>", fam, BM, BN, mode == A_ROWS ? "rows" : mode == A_ROWS_TAP3 ? "tap3" : "chanmajor");
  const double mnk = work_fraction * (double)count * p.M * (double)p.N * p.K;     // (work_fraction: the share of the row tiles a gated launch runs at least)
  ProfScope prof(name, stream, 2.0 * mnk,
                 4.0 * count * ((double)p.M * p.K / (mode == A_ROWS_TAP3 ? 3 : 1) + 1.5 * (double)p.N * p.K + (double)p.M * p.N * ((p.flags & G_RES) ? 2 : 1)));
  const int npl = nterms == 6 ? 3 : 2;
  size_t lds = (size_t)npl * BM * ROWB;
  {   // epilogue: a transpose tile per wave; behind both that and the A tile, the row-statistics block of the workgroup
    const size_t epi = (size_t)WM * WN * EPI_WAVE_FLOATS * sizeof(float);
    if (lds < epi) lds = epi;
    lds += stats_lds_floats<WM, WN, TM>() * sizeof(float);
  }
  if (p.stats_out || p.stats_in) {
    DCF_CHECK((mode == A_ROWS || (mode == A_ROWS_TAP3 && !p.stats_in)) && !p.ln_w && p.stats_w > 0, "launch_gemm_split: row statistics need A_ROWS (or a k3 producer), no fused LayerNorm and stats_w > 0");
    if (p.stats_out) DCF_CHECK((((p.flags & G_ADALN) ? BN / 2 : BN) % p.stats_w) == 0, "launch_gemm_split: tile width %d is not a multiple of stats_w = %d", BN, p.stats_w);
    if (p.stats_in) DCF_CHECK(p.ln_s && p.stats_slots >= 1, "launch_gemm_split: stats_in needs ln_s and stats_slots");
  }
  if (p.ln_w) {
    DCF_CHECK(WM == 1 && TM == 2 && TN >= 2 && BN == p.N, "launch_gemm_split: fused LayerNorm needs a tile spanning all %d columns", p.N);
    const size_t need = ((size_t)WN * BM * LN_PITCH + BM) * sizeof(float);
    if (need > lds) lds = need;
  }
  // register pipeline depth: the small wave tiles of the f16x3 mode (64x32 / 32x32 per wave) keep a third stage of A / B
  // registers at 3 waves per SIMD; measured per step of one video (tools/nst_sweep.sh): 64x128 tiles 0.220 / 0.210 /
  // 0.220 ms and 64x64 tiles 0.056 / 0.051 / 0.055 ms at depth 2 / 3 / 4; the 64x64-per-wave tiles lose their third
  // wave at depth 3 (227 registers) and run 0.217 -> 0.222 ms, larger tiles spill (tools/kernel_resources.py).
  constexpr bool DEEP = TM * TN <= 2 && WM * TM <= 2;
#define LSK(MODE_, NT_, LN_) do { \
    if constexpr (NT_ == T_F16 && DEEP) \
      hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, MODE_, NT_, LN_, 3>), grid, dim3(WM * WN * 64), lds, stream, b); \
    else \
      hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, MODE_, NT_, LN_, 2>), grid, dim3(WM * WN * 64), lds, stream, b); \
  } while (0)
#define LS(MODE_, NT_) LSK(MODE_, NT_, false)
#define LSN(MODE_, NT_) LSK(MODE_, NT_, true)
  bool want_stats = false, want_aln = false;
  for (int i = 0; i < count && i < 3; ++i) {
    want_stats = want_stats || b.g[i].stats_out || b.g[i].stats_in;
    want_aln = want_aln || b.g[i].a_stats;
  }
  if (mode == A_ROWS_TAP3 && (want_stats || want_aln)) {
    // k3 convolutions of a trunk: the producer writes row statistics (STATS), the consumer normalises its A rows (ALN); the two
    // tiles such a GEMM is dispatched to (gemm_can_norm_a)
    if constexpr (WM == 1 && WN == 4 && TM == 4 && TN == 2) {
      DCF_CHECK(!(want_stats && want_aln), "launch_gemm_split: a k3 GEMM either writes row statistics or normalises its A rows, not both");
      for (int i = 0; i < count; ++i) {
        DCF_CHECK(!b.g[i].stats_in, "launch_gemm_split: stats_in is an A_ROWS feature");
        if (want_aln) DCF_CHECK(b.g[i].a_stats && b.g[i].a_ln_g && b.g[i].a_ln_b && b.g[i].a_stats_slots >= 1 && b.g[i].cin % 8 == 0 &&
                                (reinterpret_cast<uintptr_t>(b.g[i].a_ln_g) & 3) == 0, "launch_gemm_split: incomplete a_stats arguments");
      }
      const size_t lds_aln = lds + (want_aln ? ((size_t)2 * (BM + 2) + 2 * (size_t)p.cin) * sizeof(float) : 0);
      if (want_aln) {
        if (nterms == 6) hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS_TAP3, 6, false, 2, false, true>), grid, dim3(WM * WN * 64), lds_aln, stream, b);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS_TAP3, T_F16, false, 2, false, true>), grid, dim3(WM * WN * 64), lds_aln, stream, b);
      } else {
        if (nterms == 6) hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS_TAP3, 6, false, 2, true, false>), grid, dim3(WM * WN * 64), lds, stream, b);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS_TAP3, T_F16, false, 2, true, false>), grid, dim3(WM * WN * 64), lds, stream, b);
      }
    } else {
      DCF_CHECK(false, "launch_gemm_split: no row-statistics / A-normalising k3 kernel for %dx%d tiles", BM, BN);
    }
  } else
  if (want_stats) {
    // only the tiles an A_ROWS GEMM with N % 64 == 0 is dispatched to have the statistics instantiation
    if constexpr (BN % 64 == 0 && !(WM == 2 && TM == 2)) {
      if (nterms == 6) {
        if constexpr (DEEP) hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS, 6, false, 2, true>), grid, dim3(WM * WN * 64), lds, stream, b);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS, 6, false, 2, true>), grid, dim3(WM * WN * 64), lds, stream, b);
      } else {
        if constexpr (DEEP) hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS, T_F16, false, 3, true>), grid, dim3(WM * WN * 64), lds, stream, b);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_ROWS, T_F16, false, 2, true>), grid, dim3(WM * WN * 64), lds, stream, b);
      }
    } else {
      DCF_CHECK(false, "launch_gemm_split: no row-statistics kernel for %dx%d tiles", BM, BN);
    }
  } else
  if (p.ln_w) {
    if constexpr (WM == 1 && TM == 2 && TN >= 2) {
      DCF_CHECK(mode != A_CHANMAJOR && count == 1, "launch_gemm_split: fused LayerNorm: unsupported mode");
      if (mode == A_ROWS) { if (nterms == 6) LSN(A_ROWS, 6); else LSN(A_ROWS, T_F16); }
      else { if (nterms == 6) LSN(A_ROWS_TAP3, 6); else LSN(A_ROWS_TAP3, T_F16); }
    } else {
      DCF_CHECK(false, "launch_gemm_split: fused LayerNorm is not built for this tile");
    }
  } else
  if (mode == A_ROWS) { if (nterms == 6) LS(A_ROWS, 6); else LS(A_ROWS, T_F16); }
  else if (mode == A_ROWS_TAP3) { if (nterms == 6) LS(A_ROWS_TAP3, 6); else LS(A_ROWS_TAP3, T_F16); }
  else {
    // channel-major A is only instantiated for the 64-row tiles the vid_map shapes use
    if constexpr (WM == 1 && WN == 4 && TM == 2) {
      bool want_score = false;
      int score_q = 0;
      for (int i = 0; i < count && i < 3; ++i) {
        if (!b.g[i].score_out) continue;
        want_score = true;
        DCF_CHECK(b.g[i].score_tn && b.g[i].score_nq >= 1 && b.g[i].score_nq <= GEMM_SCORE_MAXQ && b.g[i].M % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(b.g[i].score_out) & 15) == 0, "launch_gemm_split: bad score arguments (1 .. %d queries)", GEMM_SCORE_MAXQ);
        score_q = b.g[i].score_nq > score_q ? b.g[i].score_nq : score_q;
      }
      if (want_score) {
        const size_t lds_sc = lds + (size_t)score_q * p.K * sizeof(float);
        // (no hipFuncAttributeMaxDynamicSharedMemorySize on these instantiations: the caller keeps the text vectors within the
        // default 64 KiB -- engine.hip scores_on_gemm -- and sends wider features through k_sidekick_*)
        DCF_CHECK(lds_sc <= 65536, "launch_gemm_split: %d text vectors of %d channels beside the tile need %zu bytes of LDS (> 64 KiB)", score_q, p.K, lds_sc);
        constexpr int NST_ = DEEP ? 3 : 2;
        if (nterms == 6) hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_CHANMAJOR, 6, false, 2, false, false, true>), grid, dim3(WM * WN * 64), lds_sc, stream, b);
        else hipLaunchKernelGGL((gemm_bf16s_kernel<WM, WN, TM, TN, A_CHANMAJOR, T_F16, false, NST_, false, false, true>), grid, dim3(WM * WN * 64), lds_sc, stream, b);
      } else {
        if (nterms == 6) LS(A_CHANMAJOR, 6); else LS(A_CHANMAJOR, T_F16);
      }
    }
    else DCF_CHECK(false, "launch_gemm_split: channel-major A needs a 64-row tile");
  }
#undef LS
#undef LSN
#undef LSK
  DCF_HIP(hipGetLastError());
  return 0;
}

static int launch_kslice(const GemmBatch& b, int count, int nterms, hipStream_t stream) {
  const GemmArgs& p = b.g[0];
  // 32-row tiles while 64-row tiles would leave most CUs without a workgroup
  constexpr long small_max = 128;
  const long tiles64 = (long)((p.M + 63) / 64) * (p.N / 64) * count;
  const bool small = tiles64 <= small_max;
  dim3 grid(small ? tile_grid<32, 64>(p) : tile_grid<64, 64>(p), 1, count);
  char name[96];
  static const bool shapes = getenv("DCF_PROF_SHAPES") != nullptr;
  const char* fam = nterms == 6 ? "gemm_bf16x6" : "gemm_f16x3";
  if (shapes) snprintf(name, sizeof(name), "%s<%dx64,kslice>[%dx%dx%dx%d]", fam, small ? 32 : 64, count, p.M, p.N, p.K);
  else snprintf(name, sizeof(name), "%s<64x64,kslice>", fam);
  const double mnk = (double)count * p.M * (double)p.N * p.K;
  ProfScope prof(name, stream, 2.0 * mnk, 4.0 * count * ((double)p.M * p.K + 1.5 * (double)p.N * p.K + (double)p.M * p.N * ((p.flags & G_RES) ? 2 : 1)));
  if (small) {   // (eight K slices for K = 1024 on these tiles measured the same as four: 2.34 vs 2.35 ms per step)
    if (nterms == 6) hipLaunchKernelGGL((gemm_bf16s_kslice_kernel<6, 1>), grid, dim3(256), 0, stream, b);
    else hipLaunchKernelGGL((gemm_bf16s_kslice_kernel<T_F16, 1>), grid, dim3(256), 0, stream, b);
  } else {
    if (nterms == 6) hipLaunchKernelGGL((gemm_bf16s_kslice_kernel<6, 2>), grid, dim3(256), 0, stream, b);
    else hipLaunchKernelGGL((gemm_bf16s_kslice_kernel<T_F16, 2>), grid, dim3(256), 0, stream, b);
  }
  DCF_HIP(hipGetLastError());
  return 0;
}

// mirrors the dispatch below: true when an A_ROWS GEMM of this shape goes to the 128x256 or the 64x256 tile kernel, whose
// waves own 64 columns = one (scale, shift) pair of 32-column fragments
bool gemm_can_fuse_adaln(int M, int N, int K) {
  if (N % 256 != 0 || K % SBK != 0) return false;
  const long tiles64 = (long)((M + 63) / 64) * (N / 64);
  if (K >= 4 * SBK && tiles64 <= 512) return false;                   // k-sliced kernel (small grids)
  return M >= 65536 || (long)((M + 63) / 64) * (N / 256) >= 448;      // 128x256 / 64x256 (WANT workgroups)
}

// mirrors the dispatch below for A_ROWS: k-sliced kernel -> no; tile kernels -> their width must be a multiple of 64 (all
// of the N % 64 == 0 tiles are)
bool gemm_can_carry_stats(int M, int N, int K, int count, int nterms) {
  if (nterms == 0 || N % 64 != 0 || K % SBK != 0 || M <= 0) return false;
  const long tiles64 = (long)((M + 63) / 64) * (N / 64) * count;
  constexpr long kslice_max = 512;
  if (K >= 4 * SBK && tiles64 <= ((K >= 16 * SBK && nterms != T_F16) ? kslice_max : kslice_max / 2)) return false;
  return true;
}

// mirrors the A_ROWS_TAP3 dispatch below (no forced tile, no fused LayerNorm)
bool gemm_can_norm_a(int M, int N, int K, int nterms, int* stats_w) {
  if (nterms == 0 || K % SBK != 0 || M <= 0) return false;
  // the 128x256 tile only.  On the three-wave 128x96 tile (reg_head trunks, N = 288) the normalisation in the A staging made
  // the consumer 1.7x slower (two GEMMs per launch: 0.82 -> 1.42 ms at 261120 rows) for 0.12 ms of LayerNorm launches saved:
  // three waves stage the 128 rows that four stage on the wide tile, and that kernel's staging is already its critical path
  if (N % 256 == 0 && M >= 65536) { if (stats_w) *stats_w = 64; return true; }
  (void)nterms;
  return false;
}

bool gemm_can_fuse_ln(int M, int N, int K, GemmAMode mode) {
  // The fused kernel needs a 64 x N tile, i.e. M / 64 workgroups.  Measured at T = 16384 (rocprofv3): with 256
  // workgroups (M = 16384, one per CU) conv + LN fused 64 us vs 40 + 8 us as two kernels on 64x128 tiles; with 510
  // (M = 32640, the heads) 79 vs 74 + 12.5 us.  So: only where the grid still gives two workgroups per CU.
  constexpr long min_tiles = 448;            // (re-swept in f16x3 mode: 448 / 256 / 128 -> 1.892 / 1.973 / 1.988 ms per one-video step)
  return mode != A_CHANMAJOR && N == 256 && K % SBK == 0 && (M + 63) / 64 >= min_tiles;
}

// same contract as launch_gemm; every g[i].Ws must hold the pre-tiled bf16 planes of g[i].W (launch_split_planes)
int launch_gemm_split(const GemmArgs* g, int count, GemmAMode mode, int nterms, hipStream_t stream) {
  DCF_CHECK(count >= 1 && count <= 3, "launch_gemm_split: count %d out of range", count);
  DCF_CHECK(nterms == T_F16 || nterms == 6, "launch_gemm_split: nterms must be 16 (f16x3) or 6 (bf16x6)");
  GemmBatch b{};
  for (int i = 0; i < 3; ++i) b.g[i] = g[i < count ? i : 0];
  if (mode == A_ROWS_TAP3)                                  // k3 convolutions walk K slab-major (see gemm_bf16s_kernel)
    for (int i = 0; i < 3; ++i) b.g[i].flags |= G_TAPSLAB;
  const GemmArgs& p = g[0];
  for (int i = 0; i < count; ++i) {
    DCF_CHECK(g[i].M == p.M && g[i].N == p.N && g[i].K == p.K, "launch_gemm_split: grouped shapes differ");
    DCF_CHECK(g[i].A && g[i].Ws && (g[i].C || (g[i].ln_w && g[i].Y)), "launch_gemm_split: null operand");
    DCF_CHECK(g[i].lda % 4 == 0, "launch_gemm_split: lda %% 4 != 0");
    if (mode == A_CHANMAJOR) DCF_CHECK(g[i].M % 4 == 0 && !(g[i].flags & G_AMASK), "launch_gemm_split: channel-major A needs M %% 4 == 0, no row mask");
    if (mode == A_ROWS_TAP3) DCF_CHECK(g[i].nbr && g[i].cin % 32 == 0 && g[i].K == 3 * g[i].cin, "launch_gemm_split: bad tap3 args");
    if (g[i].flags & (G_AMASK | G_RES_MASK | G_OUT_MASK)) DCF_CHECK(g[i].rowmask, "launch_gemm_split: rowmask missing");
    if (g[i].flags & G_RES) DCF_CHECK(g[i].R, "launch_gemm_split: residual missing");
    if (g[i].stats_out || g[i].stats_in) {
      // the 16-byte epilogue carries them (gemm_wide_ok), and only the tile kernels have it
      int sw_ = 0;
      DCF_CHECK((mode == A_ROWS && gemm_can_carry_stats(g[i].M, g[i].N, g[i].K, count, nterms)) ||
                (mode == A_ROWS_TAP3 && !g[i].stats_in && gemm_can_norm_a(g[i].M, g[i].N, g[i].K, nterms, &sw_) && sw_ == g[i].stats_w),
                "launch_gemm_split: row statistics: %dx%dx%d (x%d) runs on a kernel without them", g[i].M, g[i].N, g[i].K, count);
      DCF_CHECK(g[i].ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(g[i].C) & 15) == 0 && (!g[i].bias || (reinterpret_cast<uintptr_t>(g[i].bias) & 15) == 0) &&
                (!(g[i].flags & G_RES) || (g[i].ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(g[i].R) & 15) == 0 && (!g[i].ls || (reinterpret_cast<uintptr_t>(g[i].ls) & 15) == 0))) &&
                (!g[i].stats_in || (g[i].ln_s && (reinterpret_cast<uintptr_t>(g[i].ln_s) & 15) == 0 && g[i].stats_slots >= 1)) && g[i].stats_w > 0 && !g[i].ln_w,
                "launch_gemm_split: row statistics need 16-byte aligned C / R / bias / ls / ln_s, stats_w > 0 and no fused LayerNorm");
    }
    if (g[i].flags & G_ADALN)
      DCF_CHECK(g[i].flags == G_ADALN && mode == A_ROWS && g[i].R && !g[i].ln_w && g[i].ldc % 4 == 0 && g[i].ldr % 4 == 0 &&
                (reinterpret_cast<uintptr_t>(g[i].C) & 15) == 0 && (reinterpret_cast<uintptr_t>(g[i].R) & 15) == 0 &&
                (!g[i].bias || (reinterpret_cast<uintptr_t>(g[i].bias) & 15) == 0) && gemm_can_fuse_adaln(g[i].M, g[i].N, g[i].K),
                "launch_gemm_split: G_ADALN needs A_ROWS, a residual, 16-byte aligned C / R, no other epilogue flag and a tile-kernel grid");
  }
  if (p.M <= 0) return 0;
  DCF_CHECK(p.K > 0 && p.K % SBK == 0 && p.N > 0 && p.N % 32 == 0, "launch_gemm_split: bad N=%d / K=%d", p.N, p.K);
  // Tiles put all four waves side by side along N (WM = 1): every B fragment is then fetched by exactly one wave
  // and the 64-row A tile in LDS is shared by all of them.
  const int N = p.N;
  auto wgs = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * (N / bn) * count; };
  constexpr long WANT = 448;            // ~2 workgroups per CU (32640x256x768: 64x256 tiles, 510 workgroups, 70.8 us vs 76.9 with 64x128)
  static const char* forced = getenv("DCF_GEMM_CFG");      // developer switch (tools/gemm_sweep.py): force one tile shape
  if (forced) {
    int bm = 0, bn = 0;
    if (sscanf(forced, "%dx%d", &bm, &bn) == 2 && bn > 0 && N % bn == 0) {
      if (bm == 64 && bn == 256) return launch_cfg_s<1, 4, 2, 2>(b, count, mode, nterms, stream);
      if (bm == 64 && bn == 128) return launch_cfg_s<1, 4, 2, 1>(b, count, mode, nterms, stream);
      if (bm == 64 && bn == 64) return launch_cfg_s<2, 2, 1, 1>(b, count, mode, nterms, stream);
      if (bm == 128 && bn == 128) return launch_cfg_s<2, 2, 2, 2>(b, count, mode, nterms, stream);
      if (bm == 128 && bn == 256) return launch_cfg_s<1, 4, 4, 2>(b, count, mode, nterms, stream);
    }
  }
  if (p.ln_w) {                                          // fused LayerNorm: the tile must span the row
    DCF_CHECK(count == 1 && p.Y && p.ln_b && gemm_can_fuse_ln(p.M, N, p.K, mode), "launch_gemm_split: LayerNorm cannot be fused for %dx%dx%d", p.M, N, p.K);
    if (p.ln_pe) DCF_CHECK(p.ln_mask && p.ln_T >= 64, "launch_gemm_split: fused position encoding needs a mask and T >= 64");
    return launch_cfg_s<1, 4, 2, 2>(b, count, mode, nterms, stream);
  }
  // small grids: split K over the waves instead of the tile (see gemm_bf16s_kslice_kernel)
  constexpr long kslice_max = 512;
  // (8192x256x256, 512 tiles: tile kernel 14.9 us vs 18.0 k-sliced; 8192x256x1024: 40 vs 38 -> short K switches at 256 tiles)
  // (f16x3: 8192x256x1024, 512 tiles: tile kernel 22.7 us vs 30.4 k-sliced -> 256 tiles for every K)
  if (mode == A_ROWS && N % 64 == 0 && p.K >= 4 * SBK && wgs(64, 64) <= ((p.K >= 16 * SBK && nterms != T_F16) ? kslice_max : kslice_max / 2))
    return launch_kslice(b, count, nterms, stream);
  if (mode == A_CHANMAJOR) {
    DCF_CHECK(N % 128 == 0, "launch_gemm_split: channel-major A needs N %% 128 == 0");
    if (N % 256 == 0 && wgs(64, 256) >= WANT) return launch_cfg_s<1, 4, 2, 2>(b, count, mode, nterms, stream);
    return launch_cfg_s<1, 4, 2, 1>(b, count, mode, nterms, stream);
  }
  // batched queries (M >= 64 K rows): 128x256 tiles, 128x64 per wave = half the weight fetches per MFMA at 2 waves/SIMD
  // (131072x256x1024: 189 vs 180 TFLOP/s; at M = 16384 the same tile is 30 % slower)
  if (N % 256 == 0 && p.M >= 65536 && mode != A_CHANMAJOR) return launch_cfg_s<1, 4, 4, 2>(b, count, mode, nterms, stream);
  if (N % 256 == 0 && wgs(64, 256) >= WANT) return launch_cfg_s<1, 4, 2, 2>(b, count, mode, nterms, stream);   // 64x256, 64x64 per wave
  // N = 288 (heads on E + 32 channels).  Also measured for M = 32640, K = 864: 64x288 tiles of three 64x96 waves
  // with the LayerNorm fused (each weight fragment fetched once, but 252 registers = 2 waves/SIMD) 122 us, 64x96
  // tiles of two 32x96 waves 107 us, 128x96 tiles of six 64x32 waves 117 us, against 100 + 16 us (LayerNorm kernel) for the
  // four-wave 128x96 tile below.  f16x3, M = 163200 (five videos): 256x96 tiles of six 128x32 waves 1.45 ms for the two head
  // launches against 1.05 ms for the three-wave 128x96 tile.
  if (N % 96 == 0 && N % 64 != 0) {                                                                             // 128x96
    // f16x3: three waves side by side, 128x32 each (every weight fragment fetched by exactly one wave, 4 row tiles of
    // A per fragment).  The four-wave stack of 32x96 tiles fetches each fragment four times through the 64 B/clk
    // vector-memory path: ~135 B/clk per CU at full MFMA rate once the MFMA work halved (bf16x6: half that).
    if (nterms == T_F16) return launch_cfg_s<1, 3, 4, 1>(b, count, mode, nterms, stream);
    return launch_cfg_s<4, 1, 1, 3>(b, count, mode, nterms, stream);
  }
  if (N % 160 == 0 && N % 64 != 0) return launch_cfg_s<4, 1, 1, 5>(b, count, mode, nterms, stream);             // 128x160
  if (N % 128 == 0 && wgs(64, 128) >= WANT) return launch_cfg_s<1, 4, 2, 1>(b, count, mode, nterms, stream);   // 64x128
  if (N % 64 == 0) return launch_cfg_s<2, 2, 1, 1>(b, count, mode, nterms, stream);                             // 64x64
  return launch_cfg_s<4, 1, 1, 1>(b, count, mode, nterms, stream);
}

// The same channel-major GEMM (f16x3 / bf16x6 operand split) over nz (A, C) pairs in ONE grid of 64 x 256 tiles, each pair with its own
// row-tile selection (GemmArgs::tile_skip; skip[z] may be null): see GemmBatch::zcount.
int launch_gemm_split_z(const GemmArgs& base, int nz, const float* const* A, float* const* C, const uint8_t* const* skip, const int* skip_nq,
                        int nterms, hipStream_t stream, double work_fraction) {
  DCF_CHECK(nz >= 1 && nz <= GEMM_ZMAX, "launch_gemm_split_z: %d operand sets (1 .. %d)", nz, GEMM_ZMAX);
  DCF_CHECK(nterms == T_F16 || nterms == 6, "launch_gemm_split_z: nterms must be 16 (f16x3) or 6 (bf16x6)");
  DCF_CHECK(base.Ws && base.M > 0 && base.M % 4 == 0 && base.N % 256 == 0 && base.K > 0 && base.K % SBK == 0 && base.lda % 4 == 0 && !base.flags &&
                !base.score_out && !base.ln_w && !base.stats_out && !base.stats_in,
            "launch_gemm_split_z: a plain channel-major GEMM with N %% 256 == 0 only");
  GemmBatch b{};
  for (int i = 0; i < 3; ++i) b.g[i] = base;
  b.zcount = nz;
  for (int z = 0; z < nz; ++z) {
    DCF_CHECK(A[z] && C[z], "launch_gemm_split_z: null operand");
    b.zA[z] = A[z]; b.zC[z] = C[z]; b.zskip[z] = skip ? skip[z] : nullptr; b.zskip_nq[z] = skip && skip[z] ? skip_nq[z] : 0;
  }
  b.g[0].A = A[0]; b.g[0].C = C[0];
  return launch_cfg_s<1, 4, 2, 2>(b, nz, A_CHANMAJOR, nterms, stream, work_fraction);
}

}  // namespace dcf
