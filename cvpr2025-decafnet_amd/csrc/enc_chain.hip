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
// A wave owns 32 consecutive rows of which the inner 24 come out (rows 4 .. 27; the windows of consecutive waves overlap by 8 rows), so
// every key a query may attend to (|i - j| <= 4) is one of the wave's own 32 rows: S^T = K_h Q_h^T is ONE 32 x 32 tile per head whose A
// operand is the wave's K rows exactly as they are loaded (lane = key row), the band |key - query| <= win / 2 is a per-slot predicate of
// the D layout (slot e of lane half h is key (e & 3) + 8 (e >> 2) + 4 h), the softmax a reduction over the lane's 16 slots and the
// other lane half, and P (D layout: lane = query row, slots = keys) the B operand of O_h^T = V_h^T P^T.  V^T wants lane = channel:
// it is read from the ordinary (row, channel) rows with one dword per key (a half wave reads 32 consecutive channels of one row).
// O^T has lane = row again: ctx, split into planes, feeds attn.proj as in dec_chain.hip (chain image through the LDS ring).
namespace {
constexpr int AW_VALID = 24, AW_HALO = 4, AWG_ROWS = 4 * AW_VALID;     // rows a wave / a workgroup produces
constexpr int A_BP = 0, A_LS = 256, A_END = 512;
constexpr int A_LDS_BYTES = 2 * STAGE + A_END * (int)sizeof(float);
}  // namespace

__global__ __launch_bounds__(256, 1) void k_enc_attn(EncAttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* ldf = reinterpret_cast<float*>(lds + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int wins = (p.T + AWG_ROWS - 1) / AWG_ROWS;
  const int b = (int)blockIdx.x / wins, t0 = ((int)blockIdx.x - b * wins) * AWG_ROWS;
  const int base = t0 + AW_VALID * w - AW_HALO;                  // sequence position of the wave's row 0
  const int t = base + r;
  const bool inseq = t >= 0 && t < p.T;
  const int tc = t < 0 ? 0 : (t < p.T ? t : p.T - 1);
  const int64_t row = (int64_t)b * p.T + tc;
  const bool live = inseq && p.mask[row] != 0;                   // padded query rows are forced to 0 (blocks.py:293)
  const bool owns = inseq && r >= AW_HALO && r < AW_HALO + AW_VALID && t < t0 + AWG_ROWS;   // this lane's row leaves the kernel here
  const int half = p.win / 2;
  // validity of the wave's 32 rows as keys: bit i of km = row i is a valid key, bit i of ki = it exists at all
  const unsigned km = (unsigned)__ballot(h == 0 && live), ki = (unsigned)__ballot(h == 0 && inseq);

  auto issue_piece = [&](int pair, int i) __attribute__((always_inline)) {
    const int pc = w + 4 * i;
    glds16(p.Wp + (size_t)pair * 64 * 512 + (size_t)pc * 512, lane16, (unsigned)(pair & 1) * STAGE + (unsigned)pc * 1024u);
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) issue_piece(0, i);
  ldf[A_BP + tid] = p.bp[tid];
  ldf[A_LS + tid] = p.ls ? p.ls[tid] : 1.f;

  const float qscale = 1.0f / sqrtf(sqrtf(64.f));               // d^-1/4 on q AND k (blocks.py:179, :359)
  const float* pq = p.Q + row * EE + 4 * h;
  const float* pk = p.K + row * EE + 4 * h;
  // V^T operand: lane (c = r, h) reads channel 64 hd + 32 ct + c of the wave's rows; the row of key slot i, clamped into the sequence
  // (a key outside it has probability exactly 0, its value only has to be finite)
  const float* pv = p.V + (int64_t)b * p.T * EE + r;
  f16x8 cth[16], ctl[16];
  // raw operands of a head, requested one head ahead (two register sets): Q_h, K_h of the lane's row (8 + 8 pieces of 16 bytes) and
  // the 32 values of V_h^T (2 channel tiles x 16 keys of the lane's lane half)
  int voff[16];                                                  // row offsets of the lane's 16 key slots (2 q x 8 j), clamped into the sequence
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    int tk = base + 16 * (k >> 3) + 8 * ((k & 7) >> 2) + 4 * h + (k & 3);
    tk = tk < 0 ? 0 : (tk < p.T ? tk : p.T - 1);
    voff[k] = tk * EE;
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
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) {
    if (hd + 1 < 4) load_head(hd + 1);
    // Q_h, K_h as planes: K step ks = 2 t2 + q <- channels 64 hd + 32 t2 + 16 q + (chain order) = pieces 4 t2 + 2 q, + 1
    f16x8 qh[4], ql[4], kh[4], kl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float a8[8], b8[8];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a8[4 * u + e] = rq[hd & 1][2 * ks + u][e] * qscale; b8[4 * u + e] = rk[hd & 1][2 * ks + u][e] * qscale; }
      split8(a8, 1.f, qh[ks], ql[ks]);
      split8(b8, 1.f, kh[ks], kl[ks]);
    }
    // V_h^T fragments: (ct, q): half j of lane half h = key 16 q + 8 (j >> 2) + 4 h + (j & 3)
    f16x8 vh[2][2], vl[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = rv[hd & 1][16 * ct + 8 * q + j];
        split8(v8, 1.f, vh[ct][q], vl[ct][q]);
      }
    // S^T = K_h Q_h^T: slot e of lane (r, h) = key (e & 3) + 8 (e >> 2) + 4 h against query row r
    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      S = mma(kl[ks], qh[ks], S);
      S = mma(kh[ks], ql[ks], S);
      S = mma(kh[ks], qh[ks], S);
    }
    // band + key mask + softmax (blocks.py:252-262, :279-294): keys outside the window or the sequence are -inf, padded keys inside
    // it get -1e4, padded queries come out 0
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int d = i - r;
      const bool in = d >= -half && d <= half && ((ki >> i) & 1u);
      const float pen = ((km >> i) & 1u) ? 0.f : -1e4f;
      S[e] = in ? S[e] + pen : -INFINITY;
      mx = fmaxf(mx, S[e]);
    }
    mx = xor32_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = fast_exp(S[e] - mx); sum += S[e]; }
    const float inv = live ? 1.0f / xor32_sum(sum) : 0.f;
    f16x8 ph[2], pl[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v8[e] = live ? S[8 * q + e] * inv : 0.f;
      split8(v8, 1.f, ph[q], pl[q]);
    }
    // O_h^T = V_h^T P^T -> ctx planes of K steps 2 (2 hd + ct) + q of the projection
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x16 O;
#pragma unroll
      for (int e = 0; e < 16; ++e) O[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        O = mma(vl[ct][q], ph[q], O);
        O = mma(vh[ct][q], pl[q], O);
        O = mma(vh[ct][q], ph[q], O);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = O[8 * q + e];
        split8(v8, SA, cth[2 * (2 * hd + ct) + q], ctl[2 * (2 * hd + ct) + q]);
      }
    }
  }

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
    stage_begin();
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
}

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
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_enc_attn), hipFuncAttributeMaxDynamicSharedMemorySize, A_LDS_BYTES));
    attr_set[dev] = true;
  }
  const unsigned grid = (unsigned)(a.B * ((a.T + AWG_ROWS - 1) / AWG_ROWS));
  hipLaunchKernelGGL(k_enc_attn, dim3(grid), dim3(256), A_LDS_BYTES, stream, a);
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
