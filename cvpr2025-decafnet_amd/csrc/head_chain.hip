// A prediction head as one kernel: [MaskedConv1D(k3) -> LayerNorm -> ReLU] x 2 -> MaskedConv1D(k3, 1 or 2 outputs), head.py:53-64,
// :95-103, over all rows of the pyramid.  The trunk activations (two round trips of C floats per row and layer through HBM with
// the GEMM + LayerNorm launches this replaces) stay in registers.
//
// As in ffn_chain.hip the products run transposed, Y^T = W X^T, so that a lane owns a ROW of the sequence: a wave holds a window
// of 32 consecutive pyramid rows, X^T as fp16 hi / lo planes in registers (the B operand of v_mfma_f32_32x32x16_f16), and the
// weights stream through LDS as A fragments (LDS-DMA, two-buffer ring, one barrier per stage).  A k3 convolution is three such
// products with the same B operand, Z_tap = W_tap X^T; the taps are put together on the OUTPUT side, where moving a row is moving
// a lane: y[r] = Z_0[r - 1] + Z_1[r] + Z_2[r + 1] with wave_shr:1 / wave_shl:1 DPP moves, each term under the neighbour flag of
// row r (MaskedConv1D masks its input; rows of different sequences lie back to back in the pyramid).  The 32 x C outputs of a
// layer stay in the accumulator layout -- lane (r, h) holds channels 32 ot + 8 g + 4 h .. + 3 of row r in Y[ot][4 g .. 4 g + 3] --
// LayerNorm is a per-lane sum plus one exchange between the two lane halves, and after ReLU and the fp16 split the registers
// Y[ot][8 q .. 8 q + 7] ARE the B operand of K step 2 ot + q of the next layer, provided the weight fragments enumerate the 16
// channels of a K step in that order (k = 16 kk + 8 (j >> 2) + 4 h + (j & 3) for half j of lane half h): the "chain image" of
// launch_split_chain3.  The first layer reads its input rows from memory in the same channel order.
// A workgroup's four waves hold 128 consecutive rows.  A row needs its two neighbours per layer: inside a wave they are the
// neighbouring lanes, across a wave boundary the two products concerned (tap 0 of the last row, tap 2 of the first) go through
// LDS and are added after the next barrier; the window's outer rows have nobody to ask, so of the 128 rows the inner 122 come out
// valid after two trunk layers and the output convolution (windows of consecutive workgroups overlap by 6 rows).
#include "head_chain.h"

#include <type_traits>

#include "common.h"

namespace dcf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float SA = 16.f, SW = 256.f, UNSCALE = 1.f / 4096.f;     // the f16x3 scaling of gemm_bf16s.hip
constexpr int HALO = 3, WGROWS = 128, VALID = WGROWS - 2 * HALO;   // a workgroup's window of consecutive rows, of which the inner 122 come out valid

template <int C>
struct Geo {
  static constexpr int KS = C / 16;              // K steps (16 channels) per tap
  static constexpr int KH = KS / 2;              // K steps per stage: a stage = (output tile, half of K, three taps)
  static constexpr int NT = C / 32;              // 32-channel output tiles
  static constexpr int PIECES = 3 * KH * 2;      // 1 KiB pieces per stage: (tap, K step, plane)
  static constexpr int STAGE = PIECES * 1024;
  static constexpr int SPL = 2 * NT;             // stages per layer (even: the ring buffer of a stage is its K half)
  static constexpr int NPW = (PIECES + 3) / 4;   // pieces a wave requests per stage
  static_assert(C % 32 == 0 && KS % 2 == 0, "C must be a multiple of 32");
};

__device__ __forceinline__ void split2_f16(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  const f16x2 h = __builtin_convertvector(f32x2{x0 * s, x1 * s}, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  const float r0 = __builtin_fmaf(x0, s, -(float)h[0]), r1 = __builtin_fmaf(x1, s, -(float)h[1]);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, f16x2));
}

// one 1 KiB LDS-DMA piece (ffn_chain.hip): lane l copies the 16 bytes at sbase + voff to LDS byte lds_dst + 16 l
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

#ifdef DCF_HC_STAMP
// diagnostic build only (tools/hc_stamp.sh): cycles wave 0 of workgroup 0 spends in the segments of the kernel
__device__ unsigned long long dcf_hc_stamps[8];
__device__ __forceinline__ unsigned long long hc_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(i) do { const unsigned long long t_ = hc_stamp(); acc_[i] += t_ - last_; last_ = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x16 mma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// lane i <- lane i - 1 / lane i + 1 of the wave (lanes without a source read 0)
__device__ __forceinline__ float from_prev(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float from_next(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }

__device__ __forceinline__ int find_level_hc(const LevelTable* lt, int r) {
  int l = 0;
  while (l + 1 < lt->n_levels && r >= lt->start[l + 1]) ++l;
  return l;
}

}  // namespace

// Wp [C][3][C] fp32 -> chain image: stage (ot, hf), piece (tap, kq, plane), lane (h, r'), half j:
//   W[32 ot + r'][tap][16 (hf KH + kq) + 8 (j >> 2) + 4 h + (j & 3)] * 2^8 as fp16 hi (plane 0) / lo (plane 1)
template <int C>
__global__ void k_split_chain3(const float* __restrict__ Wp, unsigned short* __restrict__ img, unsigned* __restrict__ overflow) {
  using G = Geo<C>;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (n, tap, k pair)
  if (i >= C * 3 * (C / 2)) return;
  const int n = i / (3 * (C / 2)), rem = i - n * (3 * (C / 2)), tap = rem / (C / 2), k = (rem - tap * (C / 2)) * 2;
  const float w0 = Wp[((size_t)n * 3 + tap) * C + k], w1 = Wp[((size_t)n * 3 + tap) * C + k + 1];
  unsigned hi, lo;
  split2_f16(w0, w1, SW, hi, lo);
  if (!(__builtin_fabsf(w0) * SW <= 65504.f) || !(__builtin_fabsf(w1) * SW <= 65504.f)) {
    if (overflow) atomicOr(overflow, 1u);
  }
  const int ot = n >> 5, rr = n & 31, kk = k >> 4, hf = kk / G::KH, kq = kk - hf * G::KH, kr = k & 15;
  const int a = kr >> 3, h = (kr >> 2) & 1, ii = kr & 3, j = 4 * a + ii;
  const size_t piece = (size_t)(ot * 2 + hf) * G::PIECES + (size_t)(tap * G::KH + kq) * 2;
  const size_t o = (piece * 64 + h * 32 + rr) * 8 + j;
  *reinterpret_cast<unsigned*>(img + o) = hi;
  *reinterpret_cast<unsigned*>(img + o + 64 * 8) = lo;
}

// NO = 2 is the compiled width of the output convolution; a head with one output leaves the second one's weights zero in LDS.
// blockIdx.y picks the head: two heads of the same width on the same rows (cls_head2 and reg_head) share a grid, so the partly
// filled last round of workgroups is paid once, not twice.
struct HeadChainBatch { HeadChainArgs a[2]; };
template <int C>
__global__ __launch_bounds__(256, 1) void k_head_chain(HeadChainBatch batch) {
  constexpr int NO = 2;
  const HeadChainArgs& p = batch.a[blockIdx.y];
  using G = Geo<C>;
  constexpr int KS = G::KS, KH = G::KH, NT = G::NT, STAGE = G::STAGE, PIECES = G::PIECES, NPW = G::NPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* lds_ln = reinterpret_cast<float*>(lds + 2 * STAGE);      // ln1_w, ln1_b, ln2_w, ln2_b: 4 x [C]
  float* lds_wo = lds_ln + 4 * C;                                  // [NO][3][C]
  // boundary rows of the waves: xch[parity][wave][0: last row's tap-0 product, 1: first row's tap-2 product][lane half][16]
  float* lds_x = lds_wo + NO * 3 * C;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto xch = [&](int par, int wave, int dir) __attribute__((always_inline)) -> float* { return lds_x + (((par * 4 + wave) * 2 + dir) * 2 + h) * 16; };
  const int wr = w * 32 + r;                                                       // row of the workgroup's window
  const int row = (int)blockIdx.x * VALID - HALO + wr;                           // this lane's pyramid row
  const bool inr = row >= 0 && row < p.rows;
  const int row_c = row < 0 ? 0 : (row < p.rows ? row : p.rows - 1);
  const unsigned fl = inr ? p.nbr[row_c] : 0u;
  // neighbour flags with the accumulator un-scaling folded in.  The left / right neighbour of a row is the previous / next lane,
  // except for rows 0 / 31 of a wave: theirs sits in the neighbouring WAVE and comes through LDS (exchange below); the window's
  // first and last row have none (halo rows)
  const float sfl = (fl & 1u) ? UNSCALE : 0.f;
  const float lfl_in = ((fl & 2u) && r != 0) ? UNSCALE : 0.f, rfl_in = ((fl & 4u) && r != 31) ? UNSCALE : 0.f;
  const float lfl_x = ((fl & 2u) && r == 0 && w > 0) ? UNSCALE : 0.f, rfl_x = ((fl & 4u) && r == 31 && w < 3) ? UNSCALE : 0.f;
  const float lfl1 = (fl & 2u) ? 1.f : 0.f, rfl1 = (fl & 4u) ? 1.f : 0.f, sfl1 = (fl & 1u) ? 1.f : 0.f;   // (output convolution: unscaled partials)

  // weight stream: stage g = layer * SPL + 2 ot + hf; wave w requests pieces w, w + 4, ... (the last one twice where PIECES is
  // not a multiple of 4: the same bytes to the same place)
  auto issue_piece = [&](const unsigned short* img, int sl, int parity, int i) __attribute__((always_inline)) {
    int pc = w + 4 * i;
    pc = pc < PIECES ? pc : PIECES - 1;
#ifndef DCF_HC_NO_DMA       // (ablation builds of tools/hc_stamp.sh: timing only, results are garbage)
    glds16(img + ((size_t)sl * PIECES + pc) * 512, lane16, (unsigned)parity * STAGE + (unsigned)pc * 1024u);
#endif
  };
#ifdef DCF_HC_STAMP
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = hc_stamp();
#endif
#pragma unroll
  for (int i = 0; i < NPW; ++i) issue_piece(p.W1c, 0, 0, i);

  for (int i = tid; i < C; i += 256) {
    lds_ln[i] = p.ln1_w[i]; lds_ln[C + i] = p.ln1_b[i]; lds_ln[2 * C + i] = p.ln2_w[i]; lds_ln[3 * C + i] = p.ln2_b[i];
  }
  for (int i = tid; i < NO * 3 * C; i += 256) lds_wo[i] = i < p.NO * 3 * C ? p.Wout[i] : 0.f;

  // Register budget: X planes 8 KS, Y 16 NT, Z 48, fragments 48.  Y and Z live in accumulation registers (192 of 256 at C = 288);
  // the planes of the first XA K steps join them there (MFMA operands may), which leaves the ordinary registers room for two
  // fragment sets beside the rest of the planes
  constexpr int XA = (256 - 16 * NT - 96) / 8 - 1 > 0 ? (256 - 16 * NT - 96) / 8 - 1 : 0;
  // the window's rows as B operands in chain order: K step kk, lane (r, h): channels 16 kk + 4 h .. + 3 and 16 kk + 8 + 4 h .. + 3
  f16x8 xh[KS], xl[KS];
  {
    const float* px = p.X + (int64_t)row_c * p.ldx + 4 * h;
    const bool use = (fl & 1u) != 0;                       // MaskedConv1D multiplies its input by the mask
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#ifndef DCF_HC_NO_X        // (ablation build of tools/hc_stamp.sh: no input rows)
      f32x4 v0 = *reinterpret_cast<const f32x4*>(px + 16 * kk), v1 = *reinterpret_cast<const f32x4*>(px + 16 * kk + 8);
#else
      f32x4 v0 = f32x4{0.1f, 0.2f, 0.3f, (float)lane}, v1 = f32x4{0.5f, (float)kk, 0.7f, 0.8f};
      (void)px;
#endif
      if (!use) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
      unsigned hi[4], lo[4];
      split2_f16(v0.x, v0.y, SA, hi[0], lo[0]);
      split2_f16(v0.z, v0.w, SA, hi[1], lo[1]);
      split2_f16(v1.x, v1.y, SA, hi[2], lo[2]);
      split2_f16(v1.z, v1.w, SA, hi[3], lo[3]);
      xh[kk] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
      xl[kk] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
      if (kk < XA) { asm volatile("" : "+a"(xh[kk])); asm volatile("" : "+a"(xl[kk])); }
    }
  }

  STAMP(5);
  f32x16 Y[NT], Z2[2][3];          // Z2[ot & 1]: the three taps' products of output tile ot (the previous tile's are put together meanwhile)
  bool bad = false;

  // after the next barrier: rows 0 / 31 of the wave take the neighbouring waves' share of tile OT
  auto fixup = [&](auto ot_) __attribute__((always_inline)) {
    constexpr int OT = decltype(ot_)::value;
    const float* pl_ = xch(OT & 1, w > 0 ? w - 1 : 0, 0);
    const float* pr_ = xch(OT & 1, w < 3 ? w + 1 : 3, 1);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(pl_ + 4 * g4), b = *reinterpret_cast<const f32x4*>(pr_ + 4 * g4);
#pragma unroll
      for (int i = 0; i < 4; ++i) Y[OT][4 * g4 + i] = __builtin_fmaf(a[i], lfl_x, __builtin_fmaf(b[i], rfl_x, Y[OT][4 * g4 + i]));
    }
    asm volatile("" : "+a"(Y[OT]));
  };

  // one stage: the three taps of output tile OT over K half HF; after the second half the taps are put together into Y[OT]
  auto stage = [&](auto ot_, auto hf_, int layer) __attribute__((always_inline)) {
    constexpr int OT = decltype(ot_)::value, HF = decltype(hf_)::value;
    STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the stage have landed ...
    STAMP(0);
    __syncthreads();                                       // ... everybody's have, and nobody reads the other buffer any more
    STAMP(1);
    if constexpr (HF == 1 && OT > 0) fixup(std::integral_constant<int, OT - 1>{});    // (tile OT - 1 was put together during stage (OT, 0))
    // SPL is even: stage parity = K half.  The base is made opaque per stage: hoisted out of the layer loop, every fragment
    // address became a register of its own (54 of them, and spills); this way they are immediate offsets of one register
    unsigned ab = lane16 + HF * STAGE;
    asm volatile("" : "+v"(ab));
    const unsigned char* buf = lds + ab;
    // the next stage: same layer, or the first one of layer 2, or none
    constexpr bool LAST = OT == NT - 1 && HF == 1;
    const unsigned short* nimg = LAST ? p.W2c : (layer ? p.W2c : p.W1c);
    const int nsl = LAST ? 0 : 2 * OT + HF + 1;
    const bool has_next = !LAST || layer == 0;             // wave uniform
    f32x16 (&Z)[3] = Z2[OT & 1];
    if constexpr (HF == 0) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) Z[t][e] = 0.f;
    }
    // A group = one K step of all three taps: six fragments, nine MFMAs issued so that consecutive ones feed different
    // accumulators (27 MFMAs in a row into one accumulator ran at 63 cycles each instead of 32: the chain Z[tap] += ... is a
    // dependence the matrix unit does not forward at full rate).  Per accumulator the order stays lo x hi, hi x lo, hi x hi with
    // K ascending.  Fragments are read one group (nine MFMAs, ~290 cycles) ahead.
    f16x8 fa[2][3][2];
    auto frags = [&](int kq, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
#ifndef DCF_HC_NO_LDS
        fa[set][tap][0] = *reinterpret_cast<const f16x8*>(buf + (2 * (tap * KH + kq)) * 1024);
        fa[set][tap][1] = *reinterpret_cast<const f16x8*>(buf + (2 * (tap * KH + kq) + 1) * 1024);
#else
        fa[set][tap][0] = xh[(kq + tap) % KS]; fa[set][tap][1] = xl[(kq + 2 * tap) % KS];
#endif
      }
    };
    frags(0, 0);
    constexpr int PPG = (NPW + KH - 2) / (KH - 1);          // pieces of the next stage requested per group (none in the last one)
    float t0[4], t1[4], t2[4], p0[4], n2[4];
#pragma unroll
    for (int kq = 0; kq < KH; ++kq) {
      const int set = kq & 1, kk = HF * KH + kq;
      if (kq + 1 < KH) frags(kq + 1, set ^ 1);
      // the next stage's pieces: at most one in front of the group's MFMAs and one in the middle (the four waves request at the
      // same moment and a piece occupies the CU's vector-memory path for ~16 cycles: two in a row per wave queue up behind eight)
      auto piece = [&](int j) __attribute__((always_inline)) {
        const int i = kq * PPG + j;
        if (j < PPG && kq + 1 < KH && i < NPW) {
          if constexpr (LAST) { if (has_next) issue_piece(nimg, nsl, HF ^ 1, i); }
          else issue_piece(nimg, nsl, HF ^ 1, i);
        }
      };
      if (HF == 0 && OT > 0 && kq < 4) {
        // the previous tile's taps, a quarter (4 of a lane's 16 values) per group and one micro-step behind every MFMA
        f32x16 (&Zp)[3] = Z2[(OT + 1) & 1];
        constexpr int PT = OT > 0 ? OT - 1 : 0;
        const int g4 = kq;
        piece(0);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int tap = k % 3, term = k / 3;
          if (k == 4) piece(1);
          Z[tap] = mma(term == 0 ? fa[set][tap][1] : fa[set][tap][0], term == 1 ? xl[kk] : xh[kk], Z[tap]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (k == 0) t0[i] = Zp[0][4 * g4 + i];
            if (k == 1) t2[i] = Zp[2][4 * g4 + i];
            if (k == 2) p0[i] = from_prev(t0[i]);
            if (k == 3) n2[i] = from_next(t2[i]);
            if (k == 4) t1[i] = Zp[1][4 * g4 + i] * sfl;
            if (k == 5) t1[i] = __builtin_fmaf(n2[i], rfl_in, t1[i]);
            if (k == 6) t1[i] = __builtin_fmaf(p0[i], lfl_in, t1[i]);
            if (k == 7) Y[PT][4 * g4 + i] = t1[i];
          }
          if (k == 8) {
            if (r == 31) *reinterpret_cast<f32x4*>(xch(PT & 1, w, 0) + 4 * g4) = f32x4{t0[0], t0[1], t0[2], t0[3]};
            if (r == 0) *reinterpret_cast<f32x4*>(xch(PT & 1, w, 1) + 4 * g4) = f32x4{t2[0], t2[1], t2[2], t2[3]};
          }
          const int tn = (k + 1) % 3, pn = (k + 1) / 3 == 0 ? 1 : 0;
#define DCF_HPIN(x) asm volatile("" : "+v"(fa[set][tap][term == 0 ? 1 : 0]), "+v"(fa[set][tn][pn]), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
          if (k == 0) DCF_HPIN(t0);
          if (k == 1) DCF_HPIN(t2);
          if (k == 2) DCF_HPIN(p0);
          if (k == 3) DCF_HPIN(n2);
          if (k >= 4 && k <= 6) DCF_HPIN(t1);
#undef DCF_HPIN
        }
        if (kq == 3) asm volatile("" : "+a"(Y[PT]));
      } else {
      piece(0);
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) Z[tap] = mma(fa[set][tap][1], xh[kk], Z[tap]);
      Z[0] = mma(fa[set][0][0], xl[kk], Z[0]);
      __builtin_amdgcn_sched_barrier(0);
      piece(1);
      Z[1] = mma(fa[set][1][0], xl[kk], Z[1]);
      Z[2] = mma(fa[set][2][0], xl[kk], Z[2]);
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) Z[tap] = mma(fa[set][tap][0], xh[kk], Z[tap]);
      static_assert(PPG <= 2, "two request slots per group");
      }
      // nothing moves across a group's end: the compiler's own order read every fragment right before its MFMA (no prefetch)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (HF == 1 && OT == NT - 1) {
      STAMP(2);
      // (all lane shifts of a group first, then the arithmetic: a DPP move right behind the instruction that wrote its source
      // register waits, and sixteen of them in a dependent row made this step 2 100 cycles per tile)
      f32x4 bz0[4], bz2[4];                                  // this lane's tap-0 / tap-2 products: what a neighbouring wave may need
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float p0[4], n2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { bz0[g4][i] = Z[0][4 * g4 + i]; bz2[g4][i] = Z[2][4 * g4 + i]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { p0[i] = from_prev(bz0[g4][i]); n2[i] = from_next(bz2[g4][i]); }
#pragma unroll
        for (int i = 0; i < 4; ++i) Y[OT][4 * g4 + i] = __builtin_fmaf(p0[i], lfl_in, __builtin_fmaf(n2[i], rfl_in, Z[1][4 * g4 + i] * sfl));
      }
      if (r == 31) {                                         // the next wave's first row adds these (still scaled by 2^12) ...
        float* o = xch(OT & 1, w, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<f32x4*>(o + 4 * g4) = bz0[g4];
      }
      if (r == 0) {                                          // ... and the previous wave's last row these
        float* o = xch(OT & 1, w, 1);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<f32x4*>(o + 4 * g4) = bz2[g4];
      }
      // park the tile in accumulation registers until the LayerNorm: left to itself the compiler keeps all of Y beside the X planes
      // in the 256 ordinary registers (288 wanted) and spills the shifted taps to scratch, with 64 accumulation registers idle
      asm volatile("" : "+a"(Y[OT]));
      STAMP(3);
    }
  };
  // LayerNorm over the C channels of the lane's row + ReLU, in place on Y (blocks.py:125-131: mean, then the mean of squared
  // deviations); g / b: the layer's parameters in LDS
  auto ln_relu = [&](const float* g, const float* b) __attribute__((always_inline)) {
    float s = 0.f;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += Y[ot][e];
    const float mean = xor32_sum(s) * (1.0f / C);
    bad |= !(__builtin_fabsf(mean) <= 3.4028234664e38f);   // one test per row and layer: a non-finite accumulator anywhere in the row
                                                           // (an operand left the fp16 range) makes its sum non-finite
    float q = 0.f;
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float d = Y[ot][e] - mean; q = __builtin_fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(xor32_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 gw = *reinterpret_cast<const f32x4*>(g + 32 * ot + 8 * g4 + 4 * h), gb = *reinterpret_cast<const f32x4*>(b + 32 * ot + 8 * g4 + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) Y[ot][4 * g4 + i] = fmaxf(__builtin_fmaf((Y[ot][4 * g4 + i] - mean) * rstd, gw[i], gb[i]), 0.f);
      }
  };

  for (int layer = 0; layer < 2; ++layer) {
    // (a compile-time walk over the output tiles: Y[OT] must be a register)
    auto tiles = [&](auto self, auto ot_) __attribute__((always_inline)) -> void {
      constexpr int OT = decltype(ot_)::value;
      if constexpr (OT < NT) {
        stage(ot_, std::integral_constant<int, 0>{}, layer);
        stage(ot_, std::integral_constant<int, 1>{}, layer);
        self(self, std::integral_constant<int, OT + 1>{});
      }
    };
    tiles(tiles, std::integral_constant<int, 0>{});
    STAMP(2);
    __syncthreads();
    fixup(std::integral_constant<int, NT - 1>{});
    ln_relu(lds_ln + 2 * C * layer, lds_ln + 2 * C * layer + C);
    if (layer == 0) {                                      // the next layer's B operand: Y[ot][8 q .. 8 q + 7] = K step 2 ot + q
#pragma unroll
      for (int ot = 0; ot < NT; ++ot)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          unsigned hi[4], lo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) split2_f16(Y[ot][8 * q + 2 * i], Y[ot][8 * q + 2 * i + 1], SA, hi[i], lo[i]);
          xh[2 * ot + q] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
          xl[2 * ot + q] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
          if (2 * ot + q < XA) { asm volatile("" : "+a"(xh[2 * ot + q])); asm volatile("" : "+a"(xl[2 * ot + q])); }
        }
    }
  }

  STAMP(4);
  // output convolution (head.py:60, :99): per-lane tap partials over the lane's channels, the two lane halves added, the taps
  // put together across rows as above
  float pt[3][NO];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int o = 0; o < NO; ++o) pt[t][o] = 0.f;
#pragma unroll
  for (int ot = 0; ot < NT; ++ot)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
      for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(lds_wo + (o * 3 + t) * C + 32 * ot + 8 * g4 + 4 * h);
          pt[t][o] += (Y[ot][4 * g4] * wv.x + Y[ot][4 * g4 + 1] * wv.y) + (Y[ot][4 * g4 + 2] * wv.z + Y[ot][4 * g4 + 3] * wv.w);
        }
  // (the same exchange for the output convolution's tap partials: 2 NO values per wave)
  float d0[NO], d1[NO], d2[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) { d0[o] = xor32_sum(pt[0][o]); d1[o] = xor32_sum(pt[1][o]); d2[o] = xor32_sum(pt[2][o]); }
  __syncthreads();                                         // (everybody is past the last fix-up: the exchange area is free)
  if (h == 0) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      if (r == 31) lds_x[(w * 2 + 0) * 2 + o] = d0[o];
      if (r == 0) lds_x[(w * 2 + 1) * 2 + o] = d2[o];
    }
  }
  __syncthreads();
  float yo[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) {
    const float xl_ = lds_x[((w > 0 ? w - 1 : 0) * 2 + 0) * 2 + o], xr_ = lds_x[((w < 3 ? w + 1 : 3) * 2 + 1) * 2 + o];
    // (the lane shifts first, with every lane active: a DPP move reads 0 from a lane that a branch has switched off)
    const float sp = from_prev(d0[o]), sn = from_next(d2[o]);
    const float pv = r == 0 ? (w > 0 ? xl_ : 0.f) : sp;
    const float nx = r == 31 ? (w < 3 ? xr_ : 0.f) : sn;
    yo[o] = __builtin_fmaf(pv, lfl1, __builtin_fmaf(nx, rfl1, d1[o] * sfl1)) + (o < p.NO ? p.bout[o] : 0.f);
  }
  if (h == 0 && wr >= HALO && wr < HALO + VALID && inr) {
    const LevelTable* lt = p.lt;
    const int l = find_level_hc(lt, row);
    int64_t dst = row;
    if (p.query_major) {
      const int rel = row - lt->start[l];
      const int b = rel / lt->T[l], t = rel - b * lt->T[l];
      dst = (int64_t)b * lt->S + lt->off[l] + t;
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float y = yo[o];
      if (p.mode == 1) y = fmaxf(y * lt->scale[l], 0.f);
      if (o < p.NO) p.out[dst * p.NO + o] = y;
    }
  }
  if (bad && p.status) atomicOr(p.status, 1u);
#ifdef DCF_HC_STAMP
  STAMP(6);
#ifndef DCF_HC_STAMP_WG
#define DCF_HC_STAMP_WG 1              // (a later round: -DDCF_HC_STAMP_WG=1500)
#endif
  if (blockIdx.x == DCF_HC_STAMP_WG && blockIdx.y == 0 && tid == 0)
    for (int i = 0; i < 8; ++i) dcf_hc_stamps[i] = acc_[i];
#endif
}

#ifdef DCF_HC_STAMP
}  // namespace dcf
extern "C" int dcf_debug_hc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcf::dcf_hc_stamps), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace dcf {
#endif

bool head_chain_supports(int C, int NO) { return (C == 256 || C == 288) && (NO == 1 || NO == 2); }

size_t head_chain_image_halfs(int C) { return (size_t)(C / 32) * 2 * (3 * (C / 32) * 2) * 512; }

int launch_split_chain3(const float* Wp, unsigned short* img, int C, hipStream_t stream, unsigned* overflow) {
  DCF_CHECK(C == 256 || C == 288, "launch_split_chain3: C = %d (256 or 288)", C);
  const int n = C * 3 * (C / 2);
  if (C == 256) hipLaunchKernelGGL(k_split_chain3<256>, dim3((n + 255) / 256), dim3(256), 0, stream, Wp, img, overflow);
  else hipLaunchKernelGGL(k_split_chain3<288>, dim3((n + 255) / 256), dim3(256), 0, stream, Wp, img, overflow);
  DCF_HIP(hipGetLastError());
  return 0;
}

template <int C>
static int launch_hc(const HeadChainArgs* a, int count, hipStream_t stream) {
  constexpr int bytes = 2 * Geo<C>::STAGE + (4 * C + 2 * 3 * C + 2 * 4 * 2 * 2 * 16) * (int)sizeof(float);
  static bool attr_set[64] = {};                         // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_chain<C>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set[dev] = true;
  }
  HeadChainBatch b{};
  for (int i = 0; i < count; ++i) b.a[i] = a[i];
  const unsigned grid = (unsigned)((a[0].rows + VALID - 1) / VALID);
  hipLaunchKernelGGL((k_head_chain<C>), dim3(grid, (unsigned)count), dim3(256), bytes, stream, b);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_head_chain(const HeadChainArgs* a, int count, int C, hipStream_t stream) {
  DCF_CHECK(count == 1 || count == 2, "launch_head_chain: one or two heads per launch");
  for (int i = 0; i < count; ++i) {
    DCF_CHECK(head_chain_supports(C, a[i].NO), "launch_head_chain: C = %d, NO = %d (C 256 / 288, NO 1 / 2)", C, a[i].NO);
    DCF_CHECK(a[i].rows > 0 && a[i].rows == a[0].rows && a[i].X && a[i].nbr && a[i].W1c && a[i].W2c && a[i].ln1_w && a[i].ln1_b && a[i].ln2_w &&
                  a[i].ln2_b && a[i].Wout && a[i].bout && a[i].lt && a[i].out, "launch_head_chain: null argument or unequal row counts");
    DCF_CHECK((reinterpret_cast<uintptr_t>(a[i].X) & 15) == 0 && a[i].ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(a[i].W1c) & 15) == 0 &&
                  (reinterpret_cast<uintptr_t>(a[i].W2c) & 15) == 0, "launch_head_chain: X / weight images must be 16-byte aligned");
  }
  return C == 256 ? launch_hc<256>(a, count, stream) : launch_hc<288>(a, count, stream);
}

}  // namespace dcf
