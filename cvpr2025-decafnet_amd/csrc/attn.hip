// Attention cores of the grounding path (fp32 in, fp32 out).
//
//  * xattn:  clips attend to <= 64 text tokens (MaskedMHA global branch,
//            libs/modeling/blocks.py:374-389).  K/V of one (query, head) are staged once per workgroup in
//            LDS as two fp16 planes (hi + lo, the same split the GEMMs use); a wavefront then streams
//            16-row groups of clips: q fragments are fetched two groups ahead, Q.K^T and P.V run on
//            v_mfma_f32_16x16x16_f16 with three products per step (hi.hi + hi.lo + lo.hi, fp32-class
//            accuracy), the softmax of the <= 64 scores stays in registers.  HBM traffic is one read of q
//            and one write of the context -- 8 B/channel/clip, the algorithmic minimum.
//            A VALU variant (lane owns 4 channels, online softmax) serves the shapes the MFMA tiling
//            does not cover.
//  * local:  sliding-window self attention |i-j| <= w/2 (MaskedMHA local branch, blocks.py:204-325,
//            357-373, restated as a band).  A wavefront owns one clip row; its <= w neighbour K/V
//            rows come from L1/L2; a lane owns 4 consecutive channels, a head is a group of d/4 adjacent
//            lanes, the q.k dot product is a 4-FMA partial + log2(d/4) cross-lane adds, and the softmax
//            is carried online (running max / running sum), so no score matrix is materialised.
#include "attn.h"

#include <cstdlib>
#include "common.h"

namespace dcf {

template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
  if constexpr (LPH == 1) return v;
  else return group_sum<LPH>(v);
}

// exp(x) for x <= 0 (softmax after max subtraction) as one v_exp_f32: exp2(x * log2(e)).  Relative error
// ~ (1 + |x|) * 2^-23; arguments below -126/log2(e) flush to 0 exactly like the tail of expf would round.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
  return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
}

// ------------------------------------------------------------------------------------------
// cross attention
// ------------------------------------------------------------------------------------------
// grid = (row-groups, B); block = 256 threads = 4 wavefronts; LDS = 2 * Lk * C floats.
template <int NCH, int LPH>
__global__ __launch_bounds__(256) void k_xattn_valu(XAttnArgs p) {
  extern __shared__ float smem[];
  const int C = p.C, Lk = p.Lk;
  float* Ks = smem;
  float* Vs = smem + (size_t)Lk * C;
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float scale = 1.0f / sqrtf(sqrtf((float)(C / p.heads)));   // d^-1/4 on q AND k (blocks.py:179,379)

  // stage K (pre-scaled) and V of this query
  const f32x4* Kg = reinterpret_cast<const f32x4*>(p.K + (size_t)b * Lk * C);
  const f32x4* Vg = reinterpret_cast<const f32x4*>(p.V + (size_t)b * Lk * C);
  for (int i = tid; i < Lk * C / 4; i += 256) {
    reinterpret_cast<f32x4*>(Ks)[i] = Kg[i] * scale;
    reinterpret_cast<f32x4*>(Vs)[i] = Vg[i];
  }
  __syncthreads();
  const uint8_t* kvm = p.kvmask + (size_t)b * Lk;

  for (int t = blockIdx.x * 4 + wave; t < p.T; t += gridDim.x * 4) {
    const int64_t row = (int64_t)b * p.T + t;
    Row<NCH> q;
    q.load(p.Q + row * C, C, lane);
    f32x4 acc[NCH];
    float m[NCH], l[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { q.v[j] *= scale; acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; m[j] = -INFINITY; l[j] = 0.f; }
    for (int k = 0; k < Lk; ++k) {
      if (!kvm[k]) continue;                           // masked_fill(-inf): contributes exactly 0
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = 256 * j + 4 * lane;
        const bool act = c < C;
        f32x4 kv = act ? *reinterpret_cast<const f32x4*>(Ks + (size_t)k * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        float s = head_sum<LPH>(dot4(q.v[j], kv));
        float mn = fmaxf(m[j], s);
        float corr = expf(m[j] - mn);               // exp(-inf) = 0 on the first key
        float e = expf(s - mn);
        f32x4 vv = act ? *reinterpret_cast<const f32x4*>(Vs + (size_t)k * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[j] = acc[j] * corr + e * vv;
        l[j] = l[j] * corr + e;
        m[j] = mn;
      }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) q.v[j] = acc[j] / l[j];   // all keys masked -> 0/0 = NaN, as the reference
    q.store(p.O + row * C, C, lane);
  }
}


// ------------------------------------------------------------------------------------------
// cross attention on the matrix cores, fp32 accurate (two fp16 planes per operand, three MFMA products)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 wavefronts = (64 clip rows, one head); grid = (row groups, heads, B), each workgroup
// stages K_h and V_h of its (query, head) once in LDS and then strides over row groups.
// Per wavefront (16 clip rows):
//   S^T = K_h Q^T   "swapped" product: A = K tile (16 keys x d), B = Q^T.  The D fragment then holds,
//                   per lane (row r = lane & 15, g = lane >> 4), the scores of keys 16*kt + 4*g + j:
//                   the softmax over keys is in-lane plus two cross-lane steps (xor 16, xor 32).
//   O^T = V_h^T P^T A = V^T tile (16 channels x keys), B = P^T: the B fragment of a 16-key step IS the
//                   lane's score register -- the probabilities never leave their registers.
//   Q goes global -> registers directly in B-fragment order (every byte of the q slice is read once,
//   as 64-byte pieces), O goes registers -> global as float4 (4 consecutive channels per lane).
// Arithmetic: v_mfma_f32_16x16x16_f16 on x = hi + lo (two fp16 values, 22 significant bits; below |x| = 2^-14 the
// absolute representation error is <= 2^-25), products hi*lo + lo*hi + hi*hi accumulated in fp32 -- the operand split of
// the dense convolutions (gemm_bf16s.hip) without its scaling: q, k are O(1) after the d^-1/4 scale, p is in [0, 1].
// Measured error against the fp32 reference: that of an fp32 FMA chain (tests/test_gpu_ops.py: 1e-5 at E = 1024).
// Why not the exact fp32 MFMA (v_mfma_f32_16x16x4_f32, the round-1 kernel): it retires 4 k per 32 cycles -- 64 of them per
// 16 rows x 64 channels x 33 keys = 2 048 cycles for 8 KiB of q / ctx traffic, 30 us of pure MFMA issue per SIMD at
// BASELINE config 2 against 43 us of HBM time: the core was bound by both.  Three 16-cycle fp16 products replace four
// 32-cycle fp32 ones (2.7x less matrix time), same lane layout (lane group g owns d = 16c + 4g + j of a 16-wide chunk).
typedef _Float16 xh4 __attribute__((ext_vector_type(4)));
typedef _Float16 xh2 __attribute__((ext_vector_type(2)));
typedef float xf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const f32x4& x, xh4& hi, xh4& lo) {
  const xh2 h0 = __builtin_convertvector(xf2{x.x, x.y}, xh2), h1 = __builtin_convertvector(xf2{x.z, x.w}, xh2);
  const xh2 l0 = __builtin_convertvector(xf2{x.x - (float)h0[0], x.y - (float)h0[1]}, xh2);
  const xh2 l1 = __builtin_convertvector(xf2{x.z - (float)h1[0], x.w - (float)h1[1]}, xh2);
  hi = xh4{h0[0], h0[1], h1[0], h1[1]};
  lo = xh4{l0[0], l0[1], l1[0], l1[1]};
}
__device__ __forceinline__ f32x4 mma3(const xh4& ah, const xh4& al, const xh4& bh, const xh4& bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);      // smallest terms first
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
}

template <int D16, int NKT, int REM, int QT>
__global__ __launch_bounds__(256) void k_xattn_mfma(XAttnArgs p) {
  // NKT full 16-key MFMA tiles + REM (0..2) trailing keys on the vector ALU: Lk = 33 (32 words + the
  // background token) would otherwise pay for 48 keys.  Each wavefront owns QT tiles of 16 clip rows so
  // that every K / V fragment read from LDS feeds QT MFMA groups.
  constexpr int D = 16 * D16;          // head dim
  constexpr int KP = D + 4;            // pitch (halves) of a key row: 8-byte reads of 16 rows x 2 lane groups hit distinct banks
  constexpr int VP = 16 * NKT + 4;     // pitch (halves) of a row of the transposed V image
  constexpr int KT16 = 16 * NKT;
  extern __shared__ float smem[];
  _Float16* Kh = reinterpret_cast<_Float16*>(smem);          // [KT16][KP]  hi plane of K (pre-scaled by d^-1/4)
  _Float16* Kl = Kh + KT16 * KP;                             // lo plane
  _Float16* Vh = Kl + KT16 * KP;                             // [D][VP]     hi plane of V^T: Vh[chan][key]
  _Float16* Vl = Vh + D * VP;
  float* Kr = reinterpret_cast<float*>(Vl + D * VP);         // [REM][D]    trailing keys, fp32 (vector-ALU path), pre-scaled
  float* Vr = Kr + (REM > 0 ? REM : 1) * D;                  // [REM][D]
  float* Ms = Vr + (REM > 0 ? REM : 1) * D;                  // [KT16 + REM] additive key mask (0 / -inf)
  // Context rows leave through LDS where a head's slice of a row is at least two cache lines (d = 64 / 128): a lane's 16-byte piece
  // of 16 different rows per store instruction (64 bytes per row: half a line, the other half by the next instruction) became whole
  // rows per instruction.  tools/micro/qstream.hip, the kernel's grid and maps with no arithmetic at all: 56.9 us with the direct
  // stores, 53.9 us with whole-row stores (4.71 -> 4.98 TB/s at BASELINE config 2); the same change on the LOAD side changes nothing.
  // Per wave 16 rows x CH 16-byte chunks; chunk k of row r sits at slot (k + 2 (r % 8)) % CH of its row, so that the eight lanes
  // an LDS cycle serves (rows r .. r + 7, same chunk) hit eight different bank groups.
#ifdef DCF_XATTN_DIRECT_STORE        // (A/B builds of tools/: the direct 16-byte stores)
  constexpr bool OSTAGE = false;
#else
  constexpr bool OSTAGE = D16 == 4 || D16 == 8;
#endif
  constexpr int CH = 4 * D16;                                // 16-byte chunks per row
  unsigned char* Os = reinterpret_cast<unsigned char*>(Ms + ((KT16 + REM + 3) & ~3));   // [4 waves][16 rows][D] fp32, QT = 1
  const int C = p.C, Lk = p.Lk;
  const int head = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const float scale = 1.0f / sqrtf(sqrtf((float)D));
  const bool single = p.single != 0;                  // uniform: the hi planes alone (dcf_config::attn_mode 1)

  bool nonfinite = false;                // an operand this thread split into fp16 planes was out of range / not finite
  auto track = [&](const f32x4& x) __attribute__((always_inline)) {
    const float m4 = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    nonfinite |= !(m4 <= 65504.f);       // also true for NaN
  };
  constexpr int ROWS = 64 * QT;        // clip rows per workgroup iteration
  const int n_groups = (p.T + ROWS - 1) / ROWS;
  // Q fragments are fetched TWO row groups ahead (two statically indexed register sets, the loop is unrolled by two): a wave
  // has nothing but its own requests in flight to cover the HBM latency with.
  constexpr int QA = 2;                 // row groups of q in flight per wave (3: no faster, 12 more registers)
  f32x4 qn_[QA][QT][D16];
  auto fetch_q = [&](int grp, f32x4 (&qn)[QT][D16]) __attribute__((always_inline)) {
    if (grp >= n_groups) return;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      int tt = grp * ROWS + (wave * QT + t) * 16 + r;
      tt = tt < p.T ? tt : p.T - 1;
      const float* qp = p.Q + ((int64_t)b * p.T + tt) * C + (size_t)head * D + 4 * g;
#pragma unroll
      for (int c = 0; c < D16; ++c) qn[t][c] = *reinterpret_cast<const f32x4*>(qp + 16 * c);
    }
  };
  // the first row groups' q are requested BEFORE the K / V staging: the staging's own round trips then overlap them
#pragma unroll
  for (int a = 0; a < QA; ++a) fetch_q(blockIdx.x + a * gridDim.x, qn_[a]);
  for (int i = tid; i < (KT16 + REM) * (D / 4); i += 256) {
    const int key = i / (D / 4), c4 = i % (D / 4);
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (key < Lk) {
      const size_t off = ((size_t)b * Lk + key) * C + (size_t)head * D + c4 * 4;
      kv = *reinterpret_cast<const f32x4*>(p.K + off) * scale;
      vv = *reinterpret_cast<const f32x4*>(p.V + off);
      if (key < KT16) { track(kv); track(vv); }
    }
    if (key < KT16) {
      xh4 hi, lo;
      split4(kv, hi, lo);
      *reinterpret_cast<xh4*>(Kh + key * KP + c4 * 4) = hi;
      *reinterpret_cast<xh4*>(Kl + key * KP + c4 * 4) = lo;
      split4(vv, hi, lo);
#pragma unroll
      for (int e = 0; e < 4; ++e) { Vh[(c4 * 4 + e) * VP + key] = hi[e]; Vl[(c4 * 4 + e) * VP + key] = lo[e]; }
    } else {
      *reinterpret_cast<f32x4*>(Kr + (key - KT16) * D + c4 * 4) = kv;
      *reinterpret_cast<f32x4*>(Vr + (key - KT16) * D + c4 * 4) = vv;
    }
  }
  for (int i = tid; i < KT16 + REM; i += 256) Ms[i] = (i < Lk && p.kvmask[(size_t)b * Lk + i]) ? 0.f : -INFINITY;
  __syncthreads();

  auto process = [&](int grp, f32x4 (&qn)[QT][D16]) __attribute__((always_inline)) {
    // ---- Q fragments: q[t][c] = Q[row_t][head*D + 16c + 4g .. +3], scaled, as two fp16 planes
    f32x4 q[QT][D16];
    xh4 qh[QT][D16], ql[QT][D16];
    bool live[QT];
    int64_t row[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const int tt = grp * ROWS + (wave * QT + t) * 16 + r;
      live[t] = tt < p.T;
      row[t] = (int64_t)b * p.T + (live[t] ? tt : p.T - 1);
#pragma unroll
      for (int c = 0; c < D16; ++c) {
        q[t][c] = qn[t][c] * scale;
        split4(q[t][c], qh[t][c], ql[t][c]);
      }
    }
    fetch_q(grp + QA * (int)gridDim.x, qn);          // this set is free again: request the group QA ahead
    // ---- S^T tiles
    f32x4 s[QT][NKT > 0 ? NKT : 1];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int t = 0; t < QT; ++t) s[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < D16; ++c) {
        const xh4 kh = *reinterpret_cast<const xh4*>(Kh + (16 * kt + r) * KP + 16 * c + 4 * g);
        const xh4 kl = *reinterpret_cast<const xh4*>(Kl + (16 * kt + r) * KP + 16 * c + 4 * g);
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kt] = single ? __builtin_amdgcn_mfma_f32_16x16x16f16(kh, qh[t][c], s[t][kt], 0, 0, 0) : mma3(kh, kl, qh[t][c], ql[t][c], s[t][kt]);
      }
    }
    // ---- trailing keys: full fp32 dot product per row, replicated over the 4 lane groups
    float sr[QT][REM > 0 ? REM : 1];
#pragma unroll
    for (int j = 0; j < REM; ++j) {
      float part[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) part[t] = 0.f;
#pragma unroll
      for (int c = 0; c < D16; ++c) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Kr + j * D + 16 * c + 4 * g);
#pragma unroll
        for (int t = 0; t < QT; ++t) part[t] += dot4(q[t][c], kf);
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        sr[t][j] = xor32_sum(xor16_sum(part[t])) + Ms[KT16 + j];
      }
    }
    xh4 ph[QT][NKT > 0 ? NKT : 1], pl[QT][NKT > 0 ? NKT : 1];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      // ---- softmax over keys (lane holds keys 16kt + 4g + j of its row, plus the replicated trailing keys)
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + 16 * kt + 4 * g);
        s[t][kt] += mk;
        // a q value beyond the fp16 range (hi plane inf, lo plane NaN) or a non-finite one turns EVERY score of its row into NaN
        // (a masked key gives -inf, never NaN): one compare per row instead of a range test per operand, which cost 17 % of
        // this kernel
        if (kt == 0) nonfinite |= s[t][0].x != s[t][0].x;
        mx = fmaxf(fmaxf(mx, fmaxf(s[t][kt].x, s[t][kt].y)), fmaxf(s[t][kt].z, s[t][kt].w));
      }
      if (NKT > 0) mx = xor32_max(xor16_max(mx));
#pragma unroll
      for (int j = 0; j < REM; ++j) mx = fmaxf(mx, sr[t][j]);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        s[t][kt].x = fast_exp(s[t][kt].x - mx); s[t][kt].y = fast_exp(s[t][kt].y - mx);
        s[t][kt].z = fast_exp(s[t][kt].z - mx); s[t][kt].w = fast_exp(s[t][kt].w - mx);
        sum += (s[t][kt].x + s[t][kt].y) + (s[t][kt].z + s[t][kt].w);
      }
      if (NKT > 0) sum = xor32_sum(xor16_sum(sum));
#pragma unroll
      for (int j = 0; j < REM; ++j) { sr[t][j] = fast_exp(sr[t][j] - mx); sum += sr[t][j]; }
      const float inv = 1.0f / sum;        // all keys masked: NaN row, as the reference
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) { s[t][kt] *= inv; split4(s[t][kt], ph[t][kt], pl[t][kt]); }
#pragma unroll
      for (int j = 0; j < REM; ++j) sr[t][j] *= inv;
    }
    // ---- O^T tiles and store
#pragma unroll
    for (int ct = 0; ct < D16; ++ct) {
      f32x4 o[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const xh4 vh = *reinterpret_cast<const xh4*>(Vh + (16 * ct + r) * VP + 16 * kt + 4 * g);
        const xh4 vl = *reinterpret_cast<const xh4*>(Vl + (16 * ct + r) * VP + 16 * kt + 4 * g);
#pragma unroll
        for (int t = 0; t < QT; ++t) o[t] = single ? __builtin_amdgcn_mfma_f32_16x16x16f16(vh, ph[t][kt], o[t], 0, 0, 0) : mma3(vh, vl, ph[t][kt], pl[t][kt], o[t]);
      }
#pragma unroll
      for (int j = 0; j < REM; ++j) {
        const f32x4 vr = *reinterpret_cast<const f32x4*>(Vr + j * D + 16 * ct + 4 * g);
#pragma unroll
        for (int t = 0; t < QT; ++t) o[t] += sr[t][j] * vr;
      }
      if constexpr (OSTAGE) {
        static_assert(!OSTAGE || QT == 1, "staged context stores: one row tile per wave");
        *reinterpret_cast<f32x4*>(Os + ((wave * 16 + r) * CH + ((4 * ct + g + 2 * (r & 7)) & (CH - 1))) * 16) = o[0];
      } else {
#pragma unroll
        for (int t = 0; t < QT; ++t)
          if (live[t]) *reinterpret_cast<f32x4*>(p.O + row[t] * C + (size_t)head * D + 4 * g + 16 * ct) = o[t];
      }
    }
    if constexpr (OSTAGE) {
      // whole rows per store instruction: lane l of instruction i takes the 16 bytes at (64 i + l) * 16 of the wave's tile
      const int t0 = grp * ROWS + wave * 16;
#pragma unroll
      for (int i = 0; i < D16; ++i) {
        const int pos = 64 * i + lane, rr = pos / CH, sl = pos & (CH - 1), kc = (sl - 2 * (rr & 7)) & (CH - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(Os + (wave * 16 * CH + pos) * 16);
        if (t0 + rr < p.T) *reinterpret_cast<f32x4*>(p.O + ((int64_t)b * p.T + t0 + rr) * C + (size_t)head * D + 4 * kc) = v;
      }
    }
  };
  for (int grp = blockIdx.x; grp < n_groups; grp += QA * gridDim.x) {
#pragma unroll
    for (int a = 0; a < QA; ++a)
      if (grp + a * (int)gridDim.x < n_groups) process(grp + a * gridDim.x, qn_[a]);
  }
  if (nonfinite && p.status) atomicOr(p.status, 1u);
}

// ------------------------------------------------------------------------------------------
// sliding-window self attention.  One wave per query row, 8 waves per SIMD.  Measured alternatives (level 0, 26 us):
// all K/V rows of the window requested up front (4 waves/SIMD) 0.107 -> 0.120 ms per step; K/V of a 16-row strip staged
// once through LDS (3 workgroups per CU) 0.157; a strip of queries per wave with the window in a register ring 0.275 ->
// 0.314 ms per five-video forward (profiles/r02_gemm_operand_stream.md).
// ------------------------------------------------------------------------------------------
constexpr int LA_QPW = 2;      // (four rows per wave at 6 waves per SIMD: 0.251 ms per five-video forward against 0.231; one row: 0.256)
// WMAX > 0: windows of at most WMAX keys.  A wave takes LA_QPW = two consecutive query rows: their windows overlap in all but one key,
// so the 2 * (WMAX + 1) K / V row loads serve both (10 KiB instead of 18 KiB per query through the vector-memory path at
// w = 9, which is what bounds this kernel).  The key loop is fully unrolled (no loop-carried registers: the rows of the
// next keys stay in flight across the score -> exp -> accumulate chain of the current one with counted waits) and capped at
// 64 registers = 8 waves per SIMD so that the scheduler cannot hoist the whole window's loads.
// WMAX = 0: any window, one query per wave, one round trip per key.
// EAGER (WMAX > 0; small grids: the upper pyramid levels of one video): every K / V row of the wave's window requested at once.  With a
// few hundred waves on the chip nothing hides a round trip (1 - 2 us from the Infinity Cache, where the projections' output lies), and the
// two-row ring above makes a wave pay five of them in a row: 6.5 -> 3.6 us per launch at <= 4 096 rows.  Same operations, same bits.
template <int NCH, int LPH, bool FULL, int WMAX, bool EAGER = false>
__global__ __launch_bounds__(256, EAGER ? 2 : (WMAX > 0 && NCH == 1 ? 8 : 4)) void k_local_attn(LocalAttnArgs p) {
  constexpr int QPW = WMAX > 0 ? LA_QPW : 1;           // query rows per wave
  const int lane = threadIdx.x & 63;
  const int wps = (p.T + QPW - 1) / QPW;               // waves per sequence
  const int64_t w = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the key loop is uniform
  if (w >= (int64_t)p.B * wps) return;
  const int C = FULL ? 256 * NCH : p.C;                 // FULL: every lane chunk exists, no per-lane channel predicate
  const int t = (int)(w % wps) * QPW;
  const int64_t base = (w / wps) * p.T, r = base + t;
  const int half = p.window / 2;
  const float scale = 1.0f / sqrtf(sqrtf((float)(C / p.heads)));
  // query z of the wave is row t + z (T may be odd: the second one may not exist); out-of-range keys are -inf (blocks.py:260-261)
  const int nq = min(QPW, p.T - t);
  const int lo = max(t - half, 0), hi = min(t + (nq - 1) + half, p.T - 1);
  // validity of the window's keys: one mask byte per lane (lane l <-> key lo + l), one ballot
  unsigned long long km = 0ull;
  if constexpr (WMAX > 0) km = __ballot(lane <= hi - lo && p.mask[base + lo + lane] != 0);
  bool live[QPW];
#pragma unroll
  for (int z = 0; z < QPW; ++z) live[z] = z < nq && p.mask[r + (z < nq ? z : 0)] != 0;   // padded query rows are forced to 0 (blocks.py:293)
  auto fetch = [&](int u, Row<NCH>& k, Row<NCH>& v) __attribute__((always_inline)) {
    k.load(p.K + (base + u) * C, C, lane);
    v.load(p.V + (base + u) * C, C, lane);
  };
  constexpr int NKB = EAGER ? WMAX + QPW - 1 : 2;
  static_assert(!EAGER || WMAX > 0, "EAGER needs a bounded window");
  Row<NCH> kb[NKB], vb[NKB];
  if constexpr (EAGER) {
#pragma unroll
    for (int i = 0; i < NKB; ++i) fetch(lo + i <= hi ? lo + i : hi, kb[i], vb[i]);
  } else if constexpr (WMAX > 0) {
    fetch(lo, kb[0], vb[0]);
    fetch(lo + 1 <= hi ? lo + 1 : hi, kb[1], vb[1]);
  }
  Row<NCH> q[QPW];
  f32x4 acc[QPW][NCH];
  float m[QPW][NCH], l[QPW][NCH];
#pragma unroll
  for (int z = 0; z < QPW; ++z) {
    q[z].load(p.Q + (r + (z < nq ? z : 0)) * C, C, lane);
#pragma unroll
    for (int j = 0; j < NCH; ++j) { q[z].v[j] *= scale; acc[z][j] = f32x4{0.f, 0.f, 0.f, 0.f}; m[z][j] = -INFINITY; l[z][j] = 0.f; }
  }
  auto key = [&](int z, bool valid, const Row<NCH>& k, const Row<NCH>& v) __attribute__((always_inline)) {
    const float pen = valid ? 0.f : -1e4f;             // padded keys get a finite -1e4 (blocks.py:279)
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float s = head_sum<LPH>(dot4(q[z].v[j], k.v[j] * scale)) + pen;
      float mn = fmaxf(m[z][j], s);
      float corr = fast_exp(m[z][j] - mn);             // exp(-inf) = 0 on the first key; v_exp_f32 instead of ~12-instruction expf
      float e = fast_exp(s - mn);
      acc[z][j] = acc[z][j] * corr + e * v.v[j];
      l[z][j] = l[z][j] * corr + e;
      m[z][j] = mn;
    }
  };
  if constexpr (WMAX > 0) {
#pragma unroll
    for (int i = 0; i < WMAX + QPW - 1; ++i) {
      const int u = lo + i;
      if (u > hi) break;
      const bool valid = ((km >> i) & 1ull) != 0;
#pragma unroll
      for (int z = 0; z < QPW; ++z)
        if (live[z] && u >= t + z - half && u <= t + z + half) key(z, valid, kb[EAGER ? i : (i & 1)], vb[EAGER ? i : (i & 1)]);
      if constexpr (!EAGER) {
        if (i + 2 < WMAX + QPW - 1 && u + 2 <= hi) fetch(u + 2, kb[i & 1], vb[i & 1]);   // this register set is free again
      }
    }
  } else {
    if (live[0]) {
      for (int u = lo; u <= hi; ++u) {
        fetch(u, kb[0], vb[0]);
        key(0, p.mask[base + u] != 0, kb[0], vb[0]);
      }
    }
  }
#pragma unroll
  for (int z = 0; z < QPW; ++z) {
    if (z >= nq) break;
    Row<NCH> o;
#pragma unroll
    for (int j = 0; j < NCH; ++j) o.v[j] = live[z] ? acc[z][j] / l[z][j] : f32x4{0.f, 0.f, 0.f, 0.f};
    o.store(p.O + (r + z) * C, C, lane);
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define DISPATCH_ATTN(KERNEL, C, heads, ...)                                                         \
  do {                                                                                               \
    DCF_CHECK((C) % (heads) == 0, "attention: C=%d not divisible by heads=%d", (int)(C), (int)(heads)); \
    const int _d = (C) / (heads);                                                                    \
    const int _n = ((C) + 255) / 256;                                                                \
    DCF_CHECK((C) % 4 == 0 && _n <= 4 && _d >= 4 && _d <= 256 && (_d & (_d - 1)) == 0,               \
              "attention: unsupported C=%d / head dim=%d (need power-of-two head dim in [4,256], C<=1024)", (int)(C), _d); \
    const int _lph = _d / 4;                                                                         \
    bool _ok = true;                                                                                 \
    switch (_n * 100 + _lph) {                                                                       \
      case 101: KERNEL(1, 1, __VA_ARGS__); break;                                                    \
      case 102: KERNEL(1, 2, __VA_ARGS__); break;                                                    \
      case 104: KERNEL(1, 4, __VA_ARGS__); break;                                                    \
      case 108: KERNEL(1, 8, __VA_ARGS__); break;                                                    \
      case 116: KERNEL(1, 16, __VA_ARGS__); break;                                                   \
      case 132: KERNEL(1, 32, __VA_ARGS__); break;                                                   \
      case 164: KERNEL(1, 64, __VA_ARGS__); break;                                                   \
      case 208: KERNEL(2, 8, __VA_ARGS__); break;                                                    \
      case 216: KERNEL(2, 16, __VA_ARGS__); break;                                                   \
      case 232: KERNEL(2, 32, __VA_ARGS__); break;                                                   \
      case 264: KERNEL(2, 64, __VA_ARGS__); break;                                                   \
      case 316: KERNEL(3, 16, __VA_ARGS__); break;                                                   \
      case 332: KERNEL(3, 32, __VA_ARGS__); break;                                                   \
      case 364: KERNEL(3, 64, __VA_ARGS__); break;                                                   \
      case 416: KERNEL(4, 16, __VA_ARGS__); break;                                                   \
      case 432: KERNEL(4, 32, __VA_ARGS__); break;                                                   \
      case 464: KERNEL(4, 64, __VA_ARGS__); break;                                                   \
      default: _ok = false;                                                                          \
    }                                                                                                \
    DCF_CHECK(_ok, "attention: no kernel for C=%d heads=%d", (int)(C), (int)(heads));               \
    DCF_HIP(hipGetLastError());                                                                      \
  } while (0)

// grids small enough that a launch is a latency chain, not a stream (k_local_attn EAGER); DCF_LA_EAGER_MAX_ROWS: developer switch
static bool local_attn_eager(int64_t rows) {
  static const long v = getenv("DCF_LA_EAGER_MAX_ROWS") ? atol(getenv("DCF_LA_EAGER_MAX_ROWS")) : 8192;
  return rows <= v;
}

#define XATTN_LAUNCH(NCH_, LPH_, grid, lds, st, a) \
  hipLaunchKernelGGL((k_xattn_valu<NCH_, LPH_>), grid, dim3(256), lds, st, a)
#define LOCAL_LAUNCH_W(NCH_, LPH_, W_, grid, st, a) do { \
    if ((a).C % 256 == 0) hipLaunchKernelGGL((k_local_attn<NCH_, LPH_, true, W_>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((k_local_attn<NCH_, LPH_, false, W_>), grid, dim3(256), 0, st, a); } while (0)
#define LOCAL_LAUNCH_EAGER(NCH_, LPH_, grid, st, a) do { \
    if ((a).C % 256 == 0) hipLaunchKernelGGL((k_local_attn<NCH_, LPH_, true, 9, true>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((k_local_attn<NCH_, LPH_, false, 9, true>), grid, dim3(256), 0, st, a); } while (0)
#define LOCAL_LAUNCH(NCH_, LPH_, grid, st, a) do { \
    if ((a).window <= 9 && NCH_ == 1 && local_attn_eager((int64_t)(a).B * (a).T)) LOCAL_LAUNCH_EAGER(1, LPH_, grid, st, a); \
    else if ((a).window <= 9) LOCAL_LAUNCH_W(NCH_, LPH_, 9, grid, st, a); \
    else if ((a).window <= 19) LOCAL_LAUNCH_W(NCH_, LPH_, 19, grid, st, a); \
    else LOCAL_LAUNCH_W(NCH_, LPH_, 0, grid, st, a); } while (0)

// the dynamic-LDS limit of one instantiation on the current device, raised once per size (the attribute belongs to the device's copy
// of the kernel; a forward under graph capture does not repeat the call)
template <int D16, int NKT, int REM, int QT>
static int xattn_raise_lds(size_t bytes) {
  static size_t have[64] = {};
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || bytes > have[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_xattn_mfma<D16, NKT, REM, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (dev >= 0 && dev < 64) have[dev] = bytes;
  }
  return 0;
}

template <int D16>
static int launch_xattn_mfma(const XAttnArgs& a, hipStream_t st) {
  // keys = 16 * nkt + rem: up to 2 trailing keys go to the vector ALU instead of a mostly empty MFMA tile
  int nkt = a.Lk / 16, rem = a.Lk % 16;
  if (rem > 2) { nkt += 1; rem = 0; }
  const int n_groups = (a.T + (D16 <= 2 ? 127 : 63)) / (D16 <= 2 ? 128 : 64);
  // enough workgroups to fill the chip, few enough that K/V staging (2*Lk*d floats) is amortised
  int gx = n_groups;
  const int per = a.heads * a.B;
  // Workgroup count: 512 are resident (186 registers: 2 per CU).  Round 3, direct context stores, interleaved A/B on one box at BASELINE
  // config 2: 1024 -> 66.3 - 70.7 us cold / 5.14 - 5.26 TB/s warm; 1536 -> 68.8 - 71.0 us; 512 and 2048 / 3072 slower; three row groups
  // of q in flight instead of two: no change (profiles/r03_notes.md).  Round 5, with the context rows leaving as whole rows through LDS, the
  // optimum moved: **768** (one and a half rounds: the second half round starts while the first is in mid-stream, so the K / V stagings and
  // the ends of the workgroups no longer coincide) 64.0 - 66.6 us cold / 52.2 - 53.7 us warm against 71.8 - 72.4 / 56.5 - 57.5 us for 1024
  // in six alternating runs on one box, 63.1 - 65.9 against 69.5 - 71.4 on another; 512 66 - 69, 640 67 - 70, 896 73 - 75, 1536 68 - 69,
  // 2048 69, 384 75 us (profiles/r05_notes.md).
  static const int wg_target = getenv("DCF_XATTN_WGS") ? atoi(getenv("DCF_XATTN_WGS")) : 768;      // (tools/: workgroup-count sweeps)
  const int cap = (wg_target + per - 1) / per;
  if (gx > cap) gx = cap;
  dim3 grid(gx, a.heads, a.B);
  constexpr int D = 16 * D16;
  // (measured at BASELINE config 2, d = 64: QT = 2 -> 216 registers, 2 waves per SIMD, 70 us against 60; non-temporal q loads /
  // context stores 72 us; heads as the fastest grid index 66 us)
  constexpr int QT = D16 <= 2 ? 2 : 1;
  // two fp16 planes of K [16 nkt][D + 4] and of V^T [D][16 nkt + 4], fp32 rows of the trailing keys (K and V), the key mask
  // (+ the staged context tile of the d = 64 / 128 instantiations: [4 waves][16 rows][D] fp32 behind the mask, 16-byte aligned)
  const size_t lds = (size_t)2 * 2 * (16 * nkt * (D + 4) + D * (16 * nkt + 4)) + ((size_t)2 * (rem > 0 ? rem : 1) * D + ((16 * nkt + rem + 3) & ~3)) * sizeof(float) +
                     ((D16 == 4 || D16 == 8) ? (size_t)4 * 16 * D * sizeof(float) : 0);
  DCF_CHECK(lds <= 160 * 1024, "xattn: K / V planes of %d keys x %d channels (+ the staged context tile) need %zu bytes of LDS (> 160 KiB)", a.Lk, D, lds);
  // (above the 64 KiB a launch gets by default the kernel's limit is raised first: d = 128 with the staged context tile, d = 256)
#define XL(NKT_, REM_) do { \
    if (lds > 64 * 1024) { if (int rc_ = xattn_raise_lds<D16, NKT_, REM_, QT>(lds)) return rc_; } \
    hipLaunchKernelGGL((k_xattn_mfma<D16, NKT_, REM_, QT>), grid, dim3(256), lds, st, a); } while (0)
  switch (nkt * 4 + rem) {
    case 0 * 4 + 1: XL(0, 1); break;
    case 0 * 4 + 2: XL(0, 2); break;
    case 1 * 4 + 0: XL(1, 0); break;
    case 1 * 4 + 1: XL(1, 1); break;
    case 1 * 4 + 2: XL(1, 2); break;
    case 2 * 4 + 0: XL(2, 0); break;
    case 2 * 4 + 1: XL(2, 1); break;
    case 2 * 4 + 2: XL(2, 2); break;
    case 3 * 4 + 0: XL(3, 0); break;
    case 3 * 4 + 1: XL(3, 1); break;
    case 3 * 4 + 2: XL(3, 2); break;
    case 4 * 4 + 0: XL(4, 0); break;
    default: DCF_CHECK(false, "xattn: Lk=%d not supported by the MFMA kernel", a.Lk);
  }
#undef XL
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_xattn(const XAttnArgs& a, hipStream_t st) {
  if (a.B * a.T <= 0) return 0;
  DCF_CHECK(a.Lk >= 1, "xattn: Lk must be >= 1");
  DCF_CHECK(a.heads >= 1 && a.C % a.heads == 0 && a.C % 4 == 0, "xattn: C=%d not divisible by heads=%d", a.C, a.heads);
  const double rows = (double)a.B * a.T;
  ProfScope prof("xattn_core", st, 4.0 * rows * a.C * a.Lk, 4.0 * (2.0 * rows * a.C + 2.0 * a.B * a.Lk * a.C));
  const int d = a.C / a.heads;
  if (a.Lk <= 64 && d % 16 == 0 && d <= 256 && (d & (d - 1)) == 0) {   // Lk in 65..66 would need nkt=4,rem>0: VALU path
    switch (d / 16) {
      case 1: return launch_xattn_mfma<1>(a, st);
      case 2: return launch_xattn_mfma<2>(a, st);
      case 4: return launch_xattn_mfma<4>(a, st);
      case 8: return launch_xattn_mfma<8>(a, st);
      default: return launch_xattn_mfma<16>(a, st);
    }
  }
  // generic VALU path (small head dims / long texts)
  size_t lds = (size_t)2 * a.Lk * a.C * sizeof(float);
  DCF_CHECK(lds <= 160 * 1024, "xattn: K/V (%d x %d) do not fit the 160 KiB LDS", a.Lk, a.C);
  int groups = (a.T + 3) / 4;
  int gx = groups < 512 ? groups : 512;     // each workgroup re-stages K/V once, then strides over rows
  dim3 grid(gx, a.B);
  DISPATCH_ATTN(XATTN_LAUNCH, a.C, a.heads, grid, lds, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Global clip self-attention (window 0): online softmax over key tiles of 64, one workgroup per (64 queries, head, sequence).
// Thread (qi = tid >> 2, part = tid & 3): the query's D channels in registers (scaled by d^-1/2: the reference scales q and k
// by d^-1/4 each, blocks.py:376-377), scores of 16 of the tile's 64 keys, a quarter of the D output channels.  K / V tiles and
// the tile's probabilities go through LDS; max / sum are combined over the four threads of a query by DPP.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_global_attn(GlobalAttnArgs p) {
  constexpr int KT = 64, KP = D + 4, DP = D / 4;
  __shared__ float Ks[KT * KP], Vs[KT * KP], Ps[64 * (KT + 4)];
  __shared__ float valid[KT];
  const int tid = threadIdx.x, qi = tid >> 2, part = tid & 3;
  const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 64;
  const int64_t base = (int64_t)b * p.T;
  const int qrow = q0 + qi < p.T ? q0 + qi : p.T - 1;
  float q[D];
  {
    const float sc = 1.0f / sqrtf((float)D);
    const float* qp = p.Q + (base + qrow) * p.C + hd * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + d);
      q[d] = v.x * sc; q[d + 1] = v.y * sc; q[d + 2] = v.z * sc; q[d + 3] = v.w * sc;
    }
  }
  float m = -INFINITY, l = 0.f, o[DP];
#pragma unroll
  for (int d = 0; d < DP; ++d) o[d] = 0.f;
  for (int k0 = 0; k0 < p.T; k0 += KT) {
    __syncthreads();                                       // the previous tile is consumed
    for (int i = tid; i < KT * (D / 4); i += 256) {
      const int j = i / (D / 4), c4 = (i - j * (D / 4)) * 4;
      const int kr = k0 + j < p.T ? k0 + j : p.T - 1;
      *reinterpret_cast<f32x4*>(Ks + j * KP + c4) = *reinterpret_cast<const f32x4*>(p.K + (base + kr) * p.C + hd * D + c4);
      *reinterpret_cast<f32x4*>(Vs + j * KP + c4) = *reinterpret_cast<const f32x4*>(p.V + (base + kr) * p.C + hd * D + c4);
    }
    if (tid < KT) valid[tid] = (k0 + tid < p.T && p.mask[base + k0 + tid]) ? 1.f : 0.f;
    __syncthreads();
    float s[16], tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = part * 16 + jj;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(Ks + j * KP + d);
        a = __builtin_fmaf(q[d], kv.x, a); a = __builtin_fmaf(q[d + 1], kv.y, a);
        a = __builtin_fmaf(q[d + 2], kv.z, a); a = __builtin_fmaf(q[d + 3], kv.w, a);
      }
      s[jj] = valid[j] != 0.f ? a : -INFINITY;
      tmax = fmaxf(tmax, s[jj]);
    }
    tmax = fmaxf(tmax, dpp_self<DPP_XOR1>(tmax));
    tmax = fmaxf(tmax, dpp_self<DPP_XOR2>(tmax));
    const float mn = fmaxf(m, tmax);
    const float corr = mn == -INFINITY ? 1.f : __expf(m - mn);     // (no valid key so far: nothing to rescale)
    float psum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float pj = s[jj] == -INFINITY ? 0.f : __expf(s[jj] - mn);
      Ps[qi * (KT + 4) + part * 16 + jj] = pj;
      psum += pj;
    }
    psum += dpp_zero<DPP_XOR1>(psum);
    psum += dpp_zero<DPP_XOR2>(psum);
    l = l * corr + psum;
    m = mn;
    __syncthreads();                                       // (the four threads of a query share a wave; the barrier keeps it simple)
#pragma unroll
    for (int d = 0; d < DP; ++d) o[d] *= corr;
    for (int j = 0; j < KT; ++j) {
      const float pj = Ps[qi * (KT + 4) + j];
#pragma unroll
      for (int d = 0; d < DP; d += 4) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(Vs + j * KP + part * DP + d);
        o[d] = __builtin_fmaf(pj, vv.x, o[d]); o[d + 1] = __builtin_fmaf(pj, vv.y, o[d + 1]);
        o[d + 2] = __builtin_fmaf(pj, vv.z, o[d + 2]); o[d + 3] = __builtin_fmaf(pj, vv.w, o[d + 3]);
      }
    }
  }
  if (q0 + qi < p.T) {
    const float inv = l > 0.f ? 1.0f / l : 0.f;            // a sequence without a valid key: zeros
    float* op = p.O + (base + q0 + qi) * p.C + hd * D + part * DP;
#pragma unroll
    for (int d = 0; d < DP; d += 4) *reinterpret_cast<f32x4*>(op + d) = f32x4{o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
  }
}

int launch_global_attn(const GlobalAttnArgs& a, hipStream_t st) {
  if ((int64_t)a.B * a.T <= 0) return 0;
  DCF_CHECK(a.heads > 0 && a.C % a.heads == 0, "global_attn: C = %d, heads = %d", a.C, a.heads);
  const int D = a.C / a.heads;
  DCF_CHECK(D == 32 || D == 64, "global_attn: head dimension %d (32 or 64)", D);
  const dim3 grid((unsigned)((a.T + 63) / 64), (unsigned)a.heads, (unsigned)a.B);
  ProfScope prof("global_attn", st, 4.0 * a.B * (double)a.T * a.T * a.C, 4.0 * 4.0 * a.B * (double)a.T * a.C);
  if (D == 64) hipLaunchKernelGGL(k_global_attn<64>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_global_attn<32>, grid, dim3(256), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_local_attn(const LocalAttnArgs& a, hipStream_t st) {
  int64_t rows = (int64_t)a.B * a.T;
  if (rows <= 0) return 0;
  DCF_CHECK(a.window >= 1 && (a.window & 1), "local_attn: window must be odd");
  const int qpw = a.window <= 19 ? LA_QPW : 1;         // query rows per wave (k_local_attn)
  dim3 grid((unsigned)(((int64_t)a.B * ((a.T + qpw - 1) / qpw) + 3) / 4));
  ProfScope prof("local_attn", st, 4.0 * rows * a.C * a.window, 4.0 * 4.0 * rows * a.C);
  DISPATCH_ATTN(LOCAL_LAUNCH, a.C, a.heads, grid, st, a);
  return 0;
}

}  // namespace dcf
