// The attention half of an encoder layer on chip, part 1 (enc_chain.h): ln_attn, the three depthwise k3 convolutions, q / k / v_norm
// and the query / key / value projections of blocks.py:462-470, :348-350 in ONE kernel (k_enc_qkv).
//
// The formulation of dec_chain.hip: a lane owns a row -- lane (r = lane & 31, h = lane >> 5) of wave w holds OUTPUT row 32 w + r of
// the workgroup's 128-row window, 128 of its 256 channels in the MFMA D layout (tile ot, slot e: channel 32 ot + (e & 3) + 8 (e >> 2)
// + 4 h); LayerNorm is a per-lane sum + one exchange between the lane halves; the depthwise convolution's neighbour rows are the
// neighbouring lanes (wave_shr:1 / wave_shl:1, the rows at the wave boundaries through LDS, the rows next to the window loaded by
// waves 0 / 3); the normalised convolution output, split into fp16 planes, IS the B operand of the transposed projection
// Y^T = W qc^T, whose A fragments stream through a two-buffer LDS ring (chain images, launch_split_chain1).  ln_attn(x) stays in
// registers for the three branches; per branch: convolution -> LayerNorm -> planes (128 registers) -> four stages of two 32-channel
// output tiles, the finished tiles stored beside the MFMAs of the next stage.
// Stride 1 only (level 0 and the stem layers): at stride 2 a lane would hold the even AND the odd input row of its output row (256
// registers) beside the planes, which does not fit one wave's 512 registers (628 bytes of scratch per lane in the build that tried):
// levels >= 1 keep k_enc_pre + the grouped GEMM for this half and use k_enc_attn for the other.
#include "enc_chain.h"

#include <type_traits>

#include "chain_common.h"
#include "common.h"

namespace dcf {

namespace {

using namespace chain;

constexpr int EE = 256;
constexpr int STAGE = 65536;                   // bytes per ring buffer (64 pieces of 1 KiB)
constexpr int WGROWS = 128;
// LDS behind the ring (floats)
constexpr int P_LNW = 0, P_LNB = 256, P_DW = 512, P_FS = 2816, P_FC = 3584, P_END = 4352;      // (P_FS / P_FC: the folded LayerNorm's s[n], c[n] of q, k, v)
constexpr int X_LAST = P_END;                  // [5][256] normalised rows: [0] the row before the window, [w + 1] the last row of wave w
constexpr int X_OTHER = X_LAST + 5 * 256;      // [5][256] [w] row 0 of wave w, [4] the row behind the window
constexpr int LDS_FLOATS = X_OTHER + 5 * 256;
constexpr int LDS_BYTES = 2 * STAGE + LDS_FLOATS * (int)sizeof(float);

}  // namespace

__global__ __launch_bounds__(256, 1) void k_enc_qkv(EncQkvArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* ldf = reinterpret_cast<float*>(lds + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  // Every per-lane LDS access below is base + 16 h + CONSTANT: the bases are made opaque so that the constants become the
  // instructions' immediate offsets (left to itself the compiler keeps one address register per constant alive across the three
  // branches -- 100+ registers and 1 KB of scratch per lane)
  // (the OFFSETS are opaque, not the pointers: a laundered pointer loses its LDS address space and every access becomes a flat one)
  unsigned o_lh = 4u * (unsigned)h, o_last = (unsigned)(X_LAST + w * 256 + 4 * h), o_oth = (unsigned)(X_OTHER + w * 256 + 4 * h);
  asm volatile("" : "+v"(o_lh), "+v"(o_last), "+v"(o_oth));
  float* lh = ldf + o_lh;                                        // parameters
  float* lh_last = ldf + o_last;                                 // [0] the row before this wave's first (normalised), [256] this wave's last
  float* lh_oth = ldf + o_oth;                                   // [0] this wave's first row, [256] the row behind its last
  const int To = p.T_in;
  const int wins = (To + WGROWS - 1) / WGROWS;                  // windows per sequence
  const int b = (int)blockIdx.x / wins, t0 = ((int)blockIdx.x - b * wins) * WGROWS;
  const int t = t0 + w * 32 + r;                                 // this lane's OUTPUT position in sequence b
  const bool inseq = t < To;
  const int tc = inseq ? t : To - 1;
  const int64_t ibase = (int64_t)b * p.T_in, orow = (int64_t)b * To + tc;

  // ---- the weight stream: per branch four stages (output tiles 2 pair, 2 pair + 1), buffer = pair & 1
  auto issue_piece = [&](const unsigned short* Wimg, int pair, int i) __attribute__((always_inline)) {     // piece w + 4 i of a stage
    const int pc = w + 4 * i;
    glds16(Wimg + (size_t)pair * 64 * 512 + (size_t)pc * 512, lane16, (unsigned)(pair & 1) * STAGE + (unsigned)pc * 1024u);
  };

  // ---- the lane's input row(s): 128 channels in D layout, v[4 ot + g] = channels 32 ot + 8 g + 4 h .. + 3
  const int te = tc;
  const bool valid_e = inseq && p.mask_in[ibase + te] != 0;
  f32x4 xe[32];
  {
    const float* px = p.X + (ibase + te) * p.ldx + 4 * h;
#pragma unroll
    for (int i = 0; i < 32; ++i) xe[i] = *reinterpret_cast<const f32x4*>(px + 32 * (i >> 2) + 8 * (i & 3));
    
  }
  // the rows next to the window: row t0 - 1 (wave 0) and row t0 + 128 (wave 3); lane l takes channels 4 l .. 4 l + 3
  const bool edge_wave = w == 0 || w == 3;
  const int tedge = w == 0 ? t0 - 1 : t0 + WGROWS;
  const bool evalid = edge_wave && tedge >= 0 && tedge < p.T_in && p.mask_in[ibase + (tedge >= 0 && tedge < p.T_in ? tedge : 0)] != 0;
  f32x4 ev = f32x4{0.f, 0.f, 0.f, 0.f};
  if (evalid) ev = *reinterpret_cast<const f32x4*>(p.X + (ibase + tedge) * p.ldx + 4 * lane);
#pragma unroll
  for (int i = 0; i < 16; ++i) issue_piece(p.W[0], 0, i);

  // ---- per-channel parameters -> LDS
  {
    ldf[P_LNW + tid] = p.ln_w[tid]; ldf[P_LNB + tid] = p.ln_b[tid];
#pragma unroll
    for (int op = 0; op < 3; ++op) {
      ldf[P_DW + op * 768 + tid] = p.dw[op][tid]; ldf[P_DW + op * 768 + 256 + tid] = p.dw[op][256 + tid]; ldf[P_DW + op * 768 + 512 + tid] = p.dw[op][512 + tid];
      ldf[P_FS + op * 256 + tid] = p.fs[op][tid]; ldf[P_FC + op * 256 + tid] = p.fc[op][tid];
    }
  }
  if (!valid_e) {
#pragma unroll
    for (int i = 0; i < 32; ++i) xe[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  
  const bool is32 = lane == 32, is31 = lane == 31;
  __syncthreads();                                               // parameters (and the raw boundary rows) are in LDS
  
  // ---- xn = ln_attn(x) * mask, in place
  auto ln_inplace = [&](f32x4 (&v)[32], bool valid) __attribute__((always_inline)) {
    float mean, rstd;
    row_stats<EE>(v, mean, rstd);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = 32 * (i >> 2) + 8 * (i & 3);
      const f32x4 g = *reinterpret_cast<const f32x4*>(lh + P_LNW + c), bb = *reinterpret_cast<const f32x4*>(lh + P_LNB + c);
      f32x4 y = (v[i] - mean) * rstd * g + bb;
      if (!valid) y = f32x4{0.f, 0.f, 0.f, 0.f};
      v[i] = y;
    }
  };
  ln_inplace(xe, valid_e);
  
  // boundary rows of the wave -> LDS; the window's outer neighbours from waves 0 / 3
  {
    if (r == 31) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        f32x4 v;
        v = xe[i];
        *reinterpret_cast<f32x4*>(lh_last + 256 + 32 * (i >> 2) + 8 * (i & 3)) = v;
      }
    }
          if (r == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) *reinterpret_cast<f32x4*>(lh_oth + 32 * (i >> 2) + 8 * (i & 3)) = xe[i];
      }
    
    if (edge_wave) {
      f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
      const float s = wave_sum((ev.x + ev.y) + (ev.z + ev.w));
      const float mean = s * (1.0f / EE);
      const f32x4 d = ev - mean;
      const float var = wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / EE);
      const float rs = 1.0f / sqrtf(var + 1e-5f);
      if (evalid) y = d * rs * *reinterpret_cast<const f32x4*>(ldf + P_LNW + 4 * lane) + *reinterpret_cast<const f32x4*>(ldf + P_LNB + 4 * lane);
      *reinterpret_cast<f32x4*>(ldf + (w == 0 ? X_LAST : X_OTHER + 4 * 256) + 4 * lane) = y;
    }
  }
  __syncthreads();
  // ---- the three branches
  auto stage_begin = [&](int s) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of stage s have landed ...
    __syncthreads();                                             // ... everybody's have, and nobody reads the other buffer any more
  };
  // (dec_chain.hip: two 32-row output tiles over 16 K steps, fragments one step ahead, vector work between the MFMAs)
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
      dma(kk);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mma(fr[set][0], bh[kk], acc[0]);
      acc[1] = mma(fr[set][2], bh[kk], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x16 A2[2][2];                                               // [stage parity][tile of the pair]
  float chk = 0.f;
  // (a rolled loop over the branches: unrolled, the three copies of everything below cost registers and spilled)
#pragma unroll 1
  for (int op = 0; op < 3; ++op) {
    const unsigned short* Wimg = op == 0 ? p.W[0] : (op == 1 ? p.W[1] : p.W[2]);
    const unsigned short* Wnext = op == 0 ? p.W[1] : p.W[2];
    float* outp = (op == 0 ? p.out[0] : (op == 1 ? p.out[1] : p.out[2])) + orow * EE + 4 * h;
    const int pdw = P_DW + op * 768, pfs = P_FS + op * 256, pfc = P_FC + op * 256;
    // depthwise k3 convolution along the rows (MaskedConv1D: the inputs are masked already), split into the B operand's planes as
    // it is produced; the branch's LayerNorm (q / k / v_norm) is folded into the projection -- W' = W diag(g) in the weight image,
    // (mean, rstd) of the raw convolution output applied in the epilogue -- so the normalised rows never exist and the convolution
    // output needs no registers of its own (one-pass variance, as the row statistics the GEMMs carry: gemm_common.h stats_load)
    f16x8 ph[16], pl[16];
    float fmean, frstd;
    {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = 2 * kk + u, c = 32 * (i >> 2) + 8 * (i & 3);
          const f32x4 lastv = *reinterpret_cast<const f32x4*>(lh_last + c);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(lh + pdw + c), w1 = *reinterpret_cast<const f32x4*>(lh + pdw + 256 + c),
                      w2 = *reinterpret_cast<const f32x4*>(lh + pdw + 512 + c);
          f32x4 pv, nv, y;
                      const f32x4 firstv = *reinterpret_cast<const f32x4*>(lh_oth + 256 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {                          // (the lane shifts with every lane active)
              const float sp = shr1(xe[i][e], lastv[e]), sn = shl1(xe[i][e], firstv[e]);
              pv[e] = is32 ? lastv[e] : sp;
              nv[e] = is31 ? firstv[e] : sn;
            }
            y = w0 * pv + w1 * xe[i] + w2 * nv;
          
          s1 += (y.x + y.y) + (y.z + y.w);
          s2 += __builtin_fmaf(y.x, y.x, y.y * y.y) + __builtin_fmaf(y.z, y.z, y.w * y.w);
          v8[4 * u] = y.x; v8[4 * u + 1] = y.y; v8[4 * u + 2] = y.z; v8[4 * u + 3] = y.w;
        }
        split8(v8, SA, ph[kk], pl[kk]);
      }
      const float mean = xor32_sum(s1) * (1.0f / EE);
      const float var = fmaxf(__builtin_fmaf(-mean, mean, xor32_sum(s2) * (1.0f / EE)), 0.f);
      fmean = mean;
      frstd = 1.0f / sqrtf(var + 1e-5f);
      if (inseq && ln_ill(mean, var) && p.status) atomicOr(p.status, 2u);         // (common.h LN_ILL_RATIO)
    }
    // output tile t2 of pair `pr`, group g: the folded LayerNorm rstd (acc - mean s[n]) + c[n] (GemmArgs::stats_in), store
    auto epilogue = [&](int pr, int t2, int g) __attribute__((always_inline)) {
      const int tile = 2 * pr + t2;
      const f32x4 fs = *reinterpret_cast<const f32x4*>(lh + pfs + 32 * tile + 8 * g);
      const f32x4 fc = *reinterpret_cast<const f32x4*>(lh + pfc + 32 * tile + 8 * g);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(__builtin_fmaf(-fmean, fs[e], A2[pr & 1][t2][4 * g + e] * UNSCALE), frstd, fc[e]);
      if (inseq) *reinterpret_cast<f32x4*>(outp + 32 * tile + 8 * g) = v;
      chk += (v.x + v.y) + (v.z + v.w);
    };
#pragma unroll
    for (int pair = 0; pair < 4; ++pair) {
      stage_begin(0);
      const unsigned char* buf = lds + (pair & 1) * STAGE + lane16;
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int e = 0; e < 16; ++e) A2[pair & 1][t2][e] = 0.f;
      gemm2(buf, ph, pl, A2[pair & 1],
            [&](int kk) __attribute__((always_inline)) {           // the next stage: this branch's next pair, or the next branch's first
              if (pair < 3) issue_piece(Wimg, pair + 1, kk);
              else if (op < 2) issue_piece(Wnext, 0, kk);
            },
            [&](int kk, int slot) __attribute__((always_inline)) { if (pair > 0 && slot == 0 && (kk & 1) == 0) epilogue(pair - 1, kk >> 3, (kk >> 1) & 3); });
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) epilogue(3, u >> 2, u & 3);
  }
  // a non-finite accumulator anywhere (an operand left the fp16 range) makes the row's checksum non-finite
  if (inseq && !(__builtin_fabsf(chk) <= 3.4028234664e38f) && p.status) atomicOr(p.status, 1u);
}

// ---- part 2: sliding-window attention + attn.proj + the residual --------------------------------------------------------------
// A wave owns 32 consecutive rows.  The keys of its queries (|i - j| <= 4) are its own 32 rows plus the 4 rows before and the 4 behind:
// S^T = K_h Q_h^T is a 32 x 32 tile over the wave's own keys, whose A operand is the wave's K rows exactly as they are loaded
// (lane = key row), plus a HALO tile whose A rows 0 .. 3 / 4 .. 7 are the neighbouring rows (loaded by lanes 0 .. 7).  The band
// |key - query| <= win / 2 is a per-slot predicate of the D layout (slot e of lane half h is key (e & 3) + 8 (e >> 2) + 4 h; of the
// halo tile only slots 0 .. 3 exist), the softmax a reduction over the lane's 16 + 4 slots and the other lane half, and P (D layout:
// lane = query row, slots = keys) the B operand of O_h^T = V_h^T P^T.  V^T wants lane = channel: it is read from the ordinary
// (row, channel) rows with one dword per key (a half wave reads 32 consecutive channels of one row).  O^T has lane = row again:
// ctx, split into planes, feeds attn.proj as in dec_chain.hip (chain image through the LDS ring).  q and k carry d^-1/4 each in the
// reference; here the product is scaled by d^-1/2 afterwards (16 + 4 multiplications instead of 128 per head).
namespace {
constexpr int AWG_ROWS = 128;
constexpr int A_BP = 0, A_LS = 256, A_END = 512;
constexpr int A_LDS_BYTES = 2 * STAGE + A_END * (int)sizeof(float);
#ifdef DCF_EA_STAMP
__device__ unsigned long long dcf_ea_stamps[16];   // diagnostic build only (tools/ea_stamp.sh): wave 0 of workgroup 1
#endif
}  // namespace

template <bool SINGLE>
__global__ __launch_bounds__(256, 1) void k_enc_attn(EncAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* ldf = reinterpret_cast<float*>(lds + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wins = (p.T + AWG_ROWS - 1) / AWG_ROWS;
  const int b = (int)blockIdx.x / wins, t0 = ((int)blockIdx.x - b * wins) * AWG_ROWS;
  const int base = t0 + 32 * w;                                  // sequence position of the wave's row 0
  const int t = base + r;
  const bool inseq = t < p.T;
  const int tc = inseq ? t : p.T - 1;
  const int64_t seq = (int64_t)b * p.T, row = seq + tc;
  const bool live = inseq && p.mask[row] != 0;                   // padded query rows are forced to 0 (blocks.py:293)
  const int half = p.win / 2;
  // validity of the wave's 32 rows as keys: bit i of km = row i is a valid key, bit i of ki = it exists at all
  const unsigned km = (unsigned)__ballot(h == 0 && live), ki = (unsigned)__ballot(h == 0 && inseq);
  // the halo tile: slot i < 4 = row base - 4 + i, slot 4 <= i < 8 = row base + 28 + i (lanes 0 .. 7 of the lower half look them up)
  const int th = r < 4 ? base - 4 + r : base + 28 + r;
  const bool hex = r < 8 && th >= 0 && th < p.T;
  const int thc = th < 0 ? 0 : (th < p.T ? th : p.T - 1);
  const unsigned kih = (unsigned)__ballot(h == 0 && hex), kmh = (unsigned)__ballot(h == 0 && hex && p.mask[seq + thc] != 0);
#ifdef DCF_EA_STAMP
  unsigned long long acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = dc_stamp();
#endif

  auto issue_piece = [&](int pair, int i) __attribute__((always_inline)) {
    const int pc = w + 4 * i;
    glds16(p.Wp + (size_t)pair * 64 * 512 + (size_t)pc * 512, lane16, (unsigned)(pair & 1) * STAGE + (unsigned)pc * 1024u);
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) issue_piece(0, i);
  ldf[A_BP + tid] = p.bp[tid];
  ldf[A_LS + tid] = p.ls ? p.ls[tid] : 1.f;

  const float sscale = 1.0f / sqrtf(64.f);                       // d^-1/4 on q AND k (blocks.py:179, :359) = d^-1/2 on the product
  const float* pq = p.Q + row * EE + 4 * h;
  const float* pk = p.K + row * EE + 4 * h;
  const float* pkh = p.K + (seq + thc) * EE + 4 * h;             // the lane's halo row (lanes 0 .. 7)
  // V^T operand: lane (c = r, h) reads channel 64 hd + 32 ct + c of the wave's rows; the row of key slot i, clamped into the sequence
  // (a key outside it has probability exactly 0, its value only has to be finite)
  const float* pv = p.V + seq * EE + r;
  f16x8 cth[16], ctl[16];
  // raw operands of a head, requested one head ahead (two register sets): Q_h, K_h of the lane's row (8 + 8 pieces of 16 bytes) and
  // the 32 + 8 values of V_h^T (2 channel tiles x (16 own + 4 halo) keys of the lane's lane half)
  int voff[16], voffh[4];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int tk = base + 16 * (k >> 3) + 8 * ((k & 7) >> 2) + 4 * h + (k & 3);
    tk = tk < p.T ? tk : p.T - 1;
    voff[k] = tk * EE;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                  // halo K step: half j of lane half h = halo slot 4 h + j
    int tk = h == 0 ? base - 4 + j : base + 32 + j;
    tk = tk < 0 ? 0 : (tk < p.T ? tk : p.T - 1);
    voffh[j] = tk * EE;
  }
  f32x4 rq[2][8], rk[2][8];
  float rv[2][32];
  auto load_head = [&](int hd) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      rq[hd & 1][u] = *reinterpret_cast<const f32x4*>(pq + 64 * hd + 8 * u);
      rk[hd & 1][u] = *reinterpret_cast<const f32x4*>(pk + 64 * hd + 8 * u);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
#ifdef ENC_NO_VGATHER          // (ablation build of tools/ec_ablate.sh: timing only)
        rv[hd & 1][16 * ct + k] = (float)(voff[k] & 7) * 0.1f;
#else
        rv[hd & 1][16 * ct + k] = pv[voff[k] + 64 * hd + 32 * ct];
#endif
      }
  };
  load_head(0);
  STAMP(0);
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) {
    // the halo rows of this head (the neighbouring waves / workgroups load the same rows: L2 hits), then the next head's own rows
    f32x4 rkh[8];
    float rvh[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rkh[u] = r < 8 ? *reinterpret_cast<const f32x4*>(pkh + 64 * hd + 8 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int j = 0; j < 4; ++j) rvh[4 * ct + j] = pv[voffh[j] + 64 * hd + 32 * ct];
    if (hd + 1 < 4) load_head(hd + 1);
    STAMP(1);
    // Q_h, K_h as planes: K step ks = 2 t2 + q <- channels 64 hd + 32 t2 + 16 q + (chain order) = pieces 2 ks, 2 ks + 1
    f16x8 qh[4], ql[4], kh[4], kl[4], hh[4], hl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a8[8], b8[8], c8[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a8[4 * u + e] = rq[hd & 1][2 * ks + u][e]; b8[4 * u + e] = rk[hd & 1][2 * ks + u][e]; c8[4 * u + e] = rkh[2 * ks + u][e]; }
      split8(a8, 1.f, qh[ks], ql[ks]);
      split8(b8, 1.f, kh[ks], kl[ks]);
      split8(c8, 1.f, hh[ks], hl[ks]);
    }
    STAMP(2);
    // V_h^T fragments: (ct, q): half j of lane half h = key 16 q + 8 (j >> 2) + 4 h + (j & 3); the halo K step: halves 0 .. 3 = slot 4 h + j
    f16x8 vh[2][3], vl[2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = rv[hd & 1][16 * ct + 8 * q + j];
        split8(v8, 1.f, vh[ct][q], vl[ct][q]);
      }
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v8[j] = j < 4 ? rvh[4 * ct + j] : 0.f;
      split8(v8, 1.f, vh[ct][2], vl[ct][2]);
    }
    STAMP(3);
    // S^T = K_h Q_h^T: slot e of lane (r, h) = key (e & 3) + 8 (e >> 2) + 4 h against query row r; SH: the halo tile
    f32x16 S, SH;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = 0.f; SH[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if constexpr (!SINGLE) {                         // (dcf_config::attn_mode 1 keeps the hi x hi product alone: its own instantiation)
        S = mma(kl[ks], qh[ks], S);
        SH = mma(hl[ks], qh[ks], SH);
        S = mma(kh[ks], ql[ks], S);
        SH = mma(hh[ks], ql[ks], SH);
      }
      S = mma(kh[ks], qh[ks], S);
      SH = mma(hh[ks], qh[ks], SH);
    }
    STAMP(4);
    // band + key mask + softmax (blocks.py:252-262, :279-294): keys outside the window or the sequence are -inf, padded keys inside
    // it get -1e4, padded queries come out 0
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int d = i - r;
      const bool in = d >= -half && d <= half && ((ki >> i) & 1u);
      const float pen = ((km >> i) & 1u) ? 0.f : -1e4f;
      S[e] = in ? __builtin_fmaf(S[e], sscale, pen) : -INFINITY;
      mx = fmaxf(mx, S[e]);
    }
    float sh4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = e + 4 * h;                                   // halo slot: row base - 4 + i (i < 4) or base + 28 + i
      const int d = (i < 4 ? i - 4 : 28 + i) - r;
      const bool in = d >= -half && d <= half && ((kih >> i) & 1u);
      const float pen = ((kmh >> i) & 1u) ? 0.f : -1e4f;
      sh4[e] = in ? __builtin_fmaf(SH[e], sscale, pen) : -INFINITY;
      mx = fmaxf(mx, sh4[e]);
    }
    mx = xor32_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = fast_exp(S[e] - mx); sum += S[e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh4[e] = fast_exp(sh4[e] - mx); sum += sh4[e]; }
    const float inv = live ? 1.0f / xor32_sum(sum) : 0.f;
    STAMP(5);
    f16x8 ph[3], pl[3];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = live ? S[8 * q + e] * inv : 0.f;
      split8(v8, 1.f, ph[q], pl[q]);
    }
    {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = (e < 4 && live) ? sh4[e & 3] * inv : 0.f;
      split8(v8, 1.f, ph[2], pl[2]);
    }
    STAMP(6);
    // O_h^T = V_h^T P^T -> ctx planes of K steps 2 (2 hd + ct) + q of the projection
    f32x16 O[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[ct][e] = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if constexpr (!SINGLE) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vl[ct][q], ph[q], O[ct]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vh[ct][q], pl[q], O[ct]);
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) O[ct] = mma(vh[ct][q], ph[q], O[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = O[ct][8 * q + e];
        split8(v8, SA, cth[2 * (2 * hd + ct) + q], ctl[2 * (2 * hd + ct) + q]);
      }
    STAMP(7);
  }
  const bool owns = inseq;                                       // every row of the window leaves the kernel here

  // ---- attn.proj + the residual: x' = skip * mask + ls * (proj(ctx) + b), row statistics for the folded ln_ffn
  auto stage_begin = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
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
      dma(kk);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mma(fr[set][0], bh[kk], acc[0]);
      acc[1] = mma(fr[set][2], bh[kk], acc[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  unsigned o_lh = 4u * (unsigned)h;
  asm volatile("" : "+v"(o_lh));
  const float* lh = ldf + o_lh;
  const float* pr = p.R + row * p.ldr + 4 * h;
  float* py = p.Y + row * p.ldy + 4 * h;
  const float mk = live ? 1.f : 0.f;
  f32x16 A2[2][2];
  f32x4 rr[2][8];                                                // the skip rows' channels of a stage (requested a stage ahead)
  float ps = 0.f, pss = 0.f;
  auto load_r = [&](int pair) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[pair & 1][u] = *reinterpret_cast<const f32x4*>(pr + 64 * pair + 32 * (u >> 2) + 8 * (u & 3));
  };
  auto epilogue = [&](int pair, int t2, int g) __attribute__((always_inline)) {
    const int c = 64 * pair + 32 * t2 + 8 * g;
    const f32x4 bb = *reinterpret_cast<const f32x4*>(lh + A_BP + c), lsv = *reinterpret_cast<const f32x4*>(lh + A_LS + c);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(A2[pair & 1][t2][4 * g + e], UNSCALE, bb[e]);
    v = rr[pair & 1][4 * t2 + g] * mk + lsv * v;
    if (owns) *reinterpret_cast<f32x4*>(py + c) = v;
    ps += (v.x + v.y) + (v.z + v.w);
    pss += __builtin_fmaf(v.x, v.x, v.y * v.y) + __builtin_fmaf(v.z, v.z, v.w * v.w);
  };
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) {
    STAMP(8);
    stage_begin();
    STAMP(9);
    if (pair > 0) asm volatile("" : "+v"(rr[(pair - 1) & 1][0]), "+v"(rr[(pair - 1) & 1][1]), "+v"(rr[(pair - 1) & 1][2]), "+v"(rr[(pair - 1) & 1][3]),
                               "+v"(rr[(pair - 1) & 1][4]), "+v"(rr[(pair - 1) & 1][5]), "+v"(rr[(pair - 1) & 1][6]), "+v"(rr[(pair - 1) & 1][7]));
    load_r(pair);
    const unsigned char* buf = lds + (pair & 1) * STAGE + lane16;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int e = 0; e < 16; ++e) A2[pair & 1][t2][e] = 0.f;
    gemm2(buf, cth, ctl, A2[pair & 1],
          [&](int kk) __attribute__((always_inline)) { if (pair < 3) issue_piece(pair + 1, kk); },
          [&](int kk, int slot) __attribute__((always_inline)) { if (pair > 0 && slot == 0 && (kk & 1) == 0) epilogue(pair - 1, kk >> 3, (kk >> 1) & 3); });
  }
  STAMP(8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(10);
  asm volatile("" : "+v"(rr[1][0]), "+v"(rr[1][1]), "+v"(rr[1][2]), "+v"(rr[1][3]), "+v"(rr[1][4]), "+v"(rr[1][5]), "+v"(rr[1][6]), "+v"(rr[1][7]));
#pragma unroll
  for (int u = 0; u < 8; ++u) epilogue(3, u >> 2, u & 3);
  const float s1 = xor32_sum(ps), s2 = xor32_sum(pss);
  if (p.stats_out && owns && h == 0) {
    const int slots = EE / p.stats_w;
    float* o = p.stats_out + row * slots * 2;
    o[0] = s1; o[1] = s2;
    for (int k = 1; k < slots; ++k) { o[2 * k] = 0.f; o[2 * k + 1] = 0.f; }
  }
  if (owns && !(__builtin_fabsf(s1) <= 3.4028234664e38f) && p.status) atomicOr(p.status, 1u);
#ifdef DCF_EA_STAMP
  STAMP(11);
  if (blockIdx.x == 1 && tid == 0)
    for (int i = 0; i < 16; ++i) dcf_ea_stamps[i] = acc_[i];
#endif
}

#ifdef DCF_EA_STAMP
}  // namespace dcf
extern "C" int dcf_debug_ea_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dcf::dcf_ea_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
namespace dcf {
#endif

int launch_enc_attn(const EncAttnArgs& a, hipStream_t stream) {
  DCF_CHECK(a.B > 0 && a.T > 0 && a.Q && a.K && a.V && a.mask && a.Wp && a.bp && a.R && a.Y && a.win >= 1 && a.win <= 9 && (a.win & 1),
            "launch_enc_attn: bad arguments (window odd, <= 9)");
  DCF_CHECK(!a.stats_out || (a.stats_w > 0 && EE % a.stats_w == 0), "launch_enc_attn: stats_out needs a slot width dividing %d", EE);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DCF_CHECK(al16(a.Q) && al16(a.K) && al16(a.V) && al16(a.Wp) && al16(a.R) && al16(a.Y) && a.ldr % 4 == 0 && a.ldy % 4 == 0,
            "launch_enc_attn: operands must be 16-byte aligned with row pitches that are multiples of 4");
  static bool attr_set[64] = {};
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_attn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS_BYTES));
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_attn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS_BYTES));
    attr_set[dev] = true;
  }
  const unsigned grid = (unsigned)(a.B * ((a.T + AWG_ROWS - 1) / AWG_ROWS));
  if (a.attn_single) hipLaunchKernelGGL(k_enc_attn<true>, dim3(grid), dim3(256), A_LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(k_enc_attn<false>, dim3(grid), dim3(256), A_LDS_BYTES, stream, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

bool enc_chain_supports(int E, int heads, int win) { return E == EE && heads == 4 && win >= 1 && win <= 9 && (win & 1); }

int launch_enc_qkv(const EncQkvArgs& a, hipStream_t stream) {
  DCF_CHECK(a.B > 0 && a.T_in > 0 && a.X && a.mask_in && a.ln_w && a.ln_b,
            "launch_enc_qkv: bad arguments");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  for (int op = 0; op < 3; ++op)
    DCF_CHECK(a.dw[op] && a.fs[op] && a.fc[op] && a.W[op] && a.out[op] && al16(a.W[op]) && al16(a.out[op]), "launch_enc_qkv: null or misaligned argument");
  DCF_CHECK(al16(a.X) && a.ldx % 4 == 0, "launch_enc_qkv: X must be 16-byte aligned, its row pitch a multiple of 4");
  static bool attr_set[64] = {};                         // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_qkv), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set[dev] = true;
  }
  const unsigned grid = (unsigned)(a.B * ((a.T_in + WGROWS - 1) / WGROWS));
  hipLaunchKernelGGL(k_enc_qkv, dim3(grid), dim3(256), LDS_BYTES, stream, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
