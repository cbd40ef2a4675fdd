// Head output convolutions + the iterative refinement branch (libs/modeling/model.py:442-471,
// libs/modeling/tcn.py, libs/modeling/head.py).  The output convolutions are thin (1-2 output channels) and run on the
// vector ALUs.  The 32-channel TCN layers run on the matrix cores in the f16x3 arithmetic of the dense convolutions
// (k_tcn_layer_mfma); the fp32 vector-ALU layer (k_tcn_layer) serves the other arithmetic modes.
#include "common.h"
#include "heads.h"

namespace dcf {

__device__ __forceinline__ int find_level(const LevelTable* lt, int r) {
  int l = 0;
  while (l + 1 < lt->n_levels && r >= lt->start[l + 1]) ++l;
  return l;
}

// ------------------------------------------------------------------------------------------
// k3 conv to NO (1 or 2) channels over masked input; one wavefront per row.
// ------------------------------------------------------------------------------------------
// One wavefront per strip of CO_STRIP consecutive rows: the CO_STRIP + 2 input rows are requested up front (each is
// used by three outputs) and the 2 x CO_STRIP wave reductions are independent chains.  (One row per wave waited for
// its three row loads and then for two dependent reductions: 28 us for the 288-channel regression head.)
constexpr int CO_STRIP = 8;

template <int NCH, int NO>
__global__ __launch_bounds__(256) void k_conv_out(ConvOutArgs p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CO_STRIP;
  if (r0 >= p.rows) return;
  const int C = p.C;
  Row<NCH> x[CO_STRIP + 2];
#pragma unroll
  for (int i = 0; i < CO_STRIP + 2; ++i) {
    const int r = r0 + i - 1;
    if (r >= 0 && r < p.rows) x[i].load(p.X + (int64_t)r * p.ldx, C, lane);
    else x[i].zero();
  }
  f32x4 w[NO][3][NCH];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = 256 * j + 4 * lane;
        w[o][tap][j] = c < C ? *reinterpret_cast<const f32x4*>(p.W + ((size_t)o * 3 + tap) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  float acc[CO_STRIP][NO];
#pragma unroll
  for (int i = 0; i < CO_STRIP; ++i) {
    const int r = r0 + i;
    const unsigned f = r < p.rows ? p.nbr[r] : 0u;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float a = 0.f;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const bool ok = tap == 0 ? (f & 2u) : tap == 1 ? (f & 1u) : (f & 4u);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const f32x4 xv = x[i + tap].v[j], ww = w[o][tap][j];
          d += (xv.x * ww.x + xv.y * ww.y) + (xv.z * ww.z + xv.w * ww.w);
        }
        a += ok ? d : 0.f;
      }
      acc[i][o] = a;
    }
  }
#pragma unroll
  for (int i = 0; i < CO_STRIP; ++i)
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[i][o] = wave_sum(acc[i][o]);
  if (lane < CO_STRIP && r0 + lane < p.rows) {           // lane i writes row r0 + i
    const int r = r0 + lane;
    const LevelTable* lt = p.lt;
    const int l = find_level(lt, p.row0 + r);
    int64_t dst = r;
    if (p.query_major) {
      const int rel = p.row0 + r - lt->start[l];
      const int b = rel / lt->T[l], t = rel - b * lt->T[l];
      dst = (int64_t)b * lt->S + lt->off[l] + t;
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < CO_STRIP; ++i) v = lane == i ? acc[i][o] : v;     // wave_sum results are wave uniform
      float y = v + p.bias[o];
      if (p.mode == 1) y = fmaxf(y * lt->scale[l], 0.f);
      p.out[dst * NO + o] = y;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same output convolution reading the RAW output of the trunk's last k3 convolution: LayerNorm + ReLU of a row happen in
// registers as it streams in, so the normalised trunk (188 MB per head at five videos) is neither written by a LayerNorm
// pass nor read back here.  A wave walks a strip of LC_STRIP output rows; every input row is requested two rows ahead,
// normalised once, and turned into its 3 x NO per-lane tap partials d[tap][o] (w[o][tap] . relu(ln(x_row)) over the lane's
// channels); output row i is the wave sum of d_{i-1}[0] + d_i[1] + d_{i+1}[2] under the neighbour flags of row i.  Parameters live in
// registers for the whole strip (a load inside the loop would wait for the row loads before it).
// (An earlier attempt normalised the 10 rows of k_conv_out's window on load: 0.377 + 0.141 -> 0.211 + 0.299 ms, no gain --
// two dependent reductions per row ahead of the loads it serialised.)
// ------------------------------------------------------------------------------------------
constexpr int LC_STRIP = 16;

template <int NCH, int NO>
__global__ __launch_bounds__(256) void k_ln_conv_out(ConvOutArgs p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * LC_STRIP;   // scalar strip loop
  if (r0 >= p.rows) return;
  const int C = p.C;
  const int r1 = min(r0 + LC_STRIP, p.rows);            // output rows [r0, r1)
  RowParam<NCH> lnw, lnb;
  lnw.init(p.ln_w, C, lane); lnb.init(p.ln_b, C, lane);
  f32x4 w[NO][3][NCH];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = 256 * j + 4 * lane;
        w[o][tap][j] = c < C ? *reinterpret_cast<const f32x4*>(p.W + ((size_t)o * 3 + tap) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  // flags of the strip's output rows: lane l <-> row r0 + l
  const unsigned fl = (lane < LC_STRIP && r0 + lane < p.rows) ? p.nbr[r0 + lane] : 0u;
  const unsigned long long m_self = __ballot((fl & 1u) != 0), m_left = __ballot((fl & 2u) != 0), m_right = __ballot((fl & 4u) != 0);

  auto fetch = [&](int r, Row<NCH>& x) __attribute__((always_inline)) {     // rows outside [0, rows) are never used by a flag
    x.load(p.X + (int64_t)(r < 0 ? 0 : (r < p.rows ? r : p.rows - 1)) * p.ldx, C, lane);
  };
  // per-lane partial tap products of one input row (the wave reduction happens once per output, below)
  auto taps = [&](Row<NCH>& x, float (&d)[3][NO]) __attribute__((always_inline)) {
    row_layernorm(x, C, lane, lnw, lnb);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      f32x4 v = x.v[j];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      x.v[j] = v;
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        f32x4 t = x.v[0] * w[o][tap][0];
#pragma unroll
        for (int j = 1; j < NCH; ++j) t += x.v[j] * w[o][tap][j];
        d[tap][o] = (t.x + t.y) + (t.z + t.w);
      }
  };

  Row<NCH> x0, x1, xa;
  float dp[3][NO], dc[3][NO], dn[3][NO];               // per-lane tap partials of rows i - 1, i, i + 1
  fetch(r0 - 1, xa); fetch(r0, x0); fetch(r0 + 1, x1);
  taps(xa, dp);
  fetch(r0 + 2, xa);
  taps(x0, dc);
  fetch(r0 + 3, x0);                                    // three rows in flight: i + 1 (x1), i + 2 (xa), i + 3 (x0)
  float res[NO];                                        // this lane's output row (lane l <-> row r0 + l)
#pragma unroll
  for (int o = 0; o < NO; ++o) res[o] = 0.f;
  auto emit = [&](int i, Row<NCH>& buf) __attribute__((always_inline)) {   // buf holds row i + 1; refilled with row i + 4
    taps(buf, dn);
    if (i + 4 <= r1) fetch(i + 4, buf);
    const int l = i - r0;
    const bool fl_ = (m_left >> l) & 1ull, fs_ = (m_self >> l) & 1ull, fr_ = (m_right >> l) & 1ull;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float y = wave_sum((fl_ ? dp[0][o] : 0.f) + (fs_ ? dc[1][o] : 0.f) + (fr_ ? dn[2][o] : 0.f));
      res[o] = lane == l ? y : res[o];
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int o = 0; o < NO; ++o) { dp[tap][o] = dc[tap][o]; dc[tap][o] = dn[tap][o]; }
  };
  // three row buffers in rotation, the loop unrolled by three so that their names are static
  for (int i = r0; i < r1; i += 3) {
    emit(i, x1);                                        // x1: row i + 1 -> refilled with row i + 4
    if (i + 1 < r1) emit(i + 1, xa);                    // xa: row i + 2 -> row i + 5
    if (i + 2 < r1) emit(i + 2, x0);                    // x0: row i + 3 -> row i + 6
  }
  if (lane < LC_STRIP && r0 + lane < p.rows) {          // lane l writes row r0 + l
    const int r = r0 + lane;
    const LevelTable* lt = p.lt;
    const int l = find_level(lt, p.row0 + r);
    int64_t dst = r;
    if (p.query_major) {
      const int rel = p.row0 + r - lt->start[l];
      const int b = rel / lt->T[l], t = rel - b * lt->T[l];
      dst = (int64_t)b * lt->S + lt->off[l] + t;
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float y = res[o] + p.bias[o];
      if (p.mode == 1) y = fmaxf(y * lt->scale[l], 0.f);
      p.out[dst * NO + o] = y;
    }
  }
}

// The same kernel for C = 256 NF + tail with tail <= 64 channels (the 288-channel heads: NF = 1, tail = 32): the tail is ONE
// float per lane instead of a second, 7/8 empty f32x4 chunk -- a third fewer vector instructions per row in a kernel that
// is bound by them.  Same operations per channel; the channel sums run over (chunks, then tail).
template <int NF, bool TAIL>
struct RowT {
  f32x4 v[NF];
  float t;
  __device__ __forceinline__ void load(const float* __restrict__ p, int tailc, int lane) {
#pragma unroll
    for (int j = 0; j < NF; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + 256 * j + 4 * lane);
    t = 0.f;
    if constexpr (TAIL) { if (lane < tailc) t = p[256 * NF + lane]; }
  }
};

template <int NF, bool TAIL, int NO>
__global__ __launch_bounds__(256) void k_ln_conv_out_t(ConvOutArgs p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * LC_STRIP;   // scalar strip loop
  if (r0 >= p.rows) return;
  const int C = p.C, tailc = C - 256 * NF;
  const bool tl = TAIL && lane < tailc;                 // this lane owns a tail channel
  const float inv_c = 1.0f / (float)C;
  const int r1 = min(r0 + LC_STRIP, p.rows);            // output rows [r0, r1)
  RowT<NF, TAIL> lnw, lnb, w[NO][3];
  lnw.load(p.ln_w, tailc, lane); lnb.load(p.ln_b, tailc, lane);
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) w[o][tap].load(p.W + ((size_t)o * 3 + tap) * C, tailc, lane);
  // flags of the strip's output rows: lane l <-> row r0 + l
  const unsigned fl = (lane < LC_STRIP && r0 + lane < p.rows) ? p.nbr[r0 + lane] : 0u;
  const unsigned long long m_self = __ballot((fl & 1u) != 0), m_left = __ballot((fl & 2u) != 0), m_right = __ballot((fl & 4u) != 0);

  auto fetch = [&](int r, RowT<NF, TAIL>& x) __attribute__((always_inline)) {   // rows outside [0, rows) are never used by a flag
    x.load(p.X + (int64_t)(r < 0 ? 0 : (r < p.rows ? r : p.rows - 1)) * p.ldx, tailc, lane);
  };
  // LayerNorm (two passes like the reference, blocks.py:125-131) + ReLU, then the per-lane partial tap products
  auto taps = [&](RowT<NF, TAIL>& x, float (&d)[3][NO]) __attribute__((always_inline)) {
    float s = x.t;
#pragma unroll
    for (int j = 0; j < NF; ++j) s += (x.v[j].x + x.v[j].y) + (x.v[j].z + x.v[j].w);
    const float mean = wave_sum(s) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      x.v[j] -= mean;
      const f32x4 q = x.v[j] * x.v[j];
      sq += (q.x + q.y) + (q.z + q.w);
    }
    if constexpr (TAIL) { x.t = tl ? x.t - mean : 0.f; sq += x.t * x.t; }
    const float rs = 1.0f / sqrtf(wave_sum(sq) * inv_c + 1e-5f);
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      f32x4 v = (x.v[j] * rs) * lnw.v[j] + lnb.v[j];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      x.v[j] = v;
    }
    if constexpr (TAIL) x.t = tl ? fmaxf((x.t * rs) * lnw.t + lnb.t, 0.f) : 0.f;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        f32x4 t4 = x.v[0] * w[o][tap].v[0];
#pragma unroll
        for (int j = 1; j < NF; ++j) t4 += x.v[j] * w[o][tap].v[j];
        float a = (t4.x + t4.y) + (t4.z + t4.w);
        if constexpr (TAIL) a += x.t * w[o][tap].t;
        d[tap][o] = a;
      }
  };

  RowT<NF, TAIL> x0, x1, xa;
  float dp[3][NO], dc[3][NO], dn[3][NO];               // per-lane tap partials of rows i - 1, i, i + 1
  fetch(r0 - 1, xa); fetch(r0, x0); fetch(r0 + 1, x1);
  taps(xa, dp);
  fetch(r0 + 2, xa);
  taps(x0, dc);
  fetch(r0 + 3, x0);                                    // three rows in flight: i + 1 (x1), i + 2 (xa), i + 3 (x0)
  float res[NO];                                        // this lane's output row (lane l <-> row r0 + l)
#pragma unroll
  for (int o = 0; o < NO; ++o) res[o] = 0.f;
  auto emit = [&](int i, RowT<NF, TAIL>& buf) __attribute__((always_inline)) {   // buf holds row i + 1; refilled with row i + 4
    taps(buf, dn);
    if (i + 4 <= r1) fetch(i + 4, buf);
    const int l = i - r0;
    const bool fl_ = (m_left >> l) & 1ull, fs_ = (m_self >> l) & 1ull, fr_ = (m_right >> l) & 1ull;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      const float y = wave_sum((fl_ ? dp[0][o] : 0.f) + (fs_ ? dc[1][o] : 0.f) + (fr_ ? dn[2][o] : 0.f));
      res[o] = lane == l ? y : res[o];
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
      for (int o = 0; o < NO; ++o) { dp[tap][o] = dc[tap][o]; dc[tap][o] = dn[tap][o]; }
  };
  for (int i = r0; i < r1; i += 3) {                    // three row buffers in rotation, static names
    emit(i, x1);
    if (i + 1 < r1) emit(i + 1, xa);
    if (i + 2 < r1) emit(i + 2, x0);
  }
  if (lane < LC_STRIP && r0 + lane < p.rows) {          // lane l writes row r0 + l
    const int r = r0 + lane;
    const LevelTable* lt = p.lt;
    const int l = find_level(lt, p.row0 + r);
    int64_t dst = r;
    if (p.query_major) {
      const int rel = p.row0 + r - lt->start[l];
      const int b = rel / lt->T[l], t = rel - b * lt->T[l];
      dst = (int64_t)b * lt->S + lt->off[l] + t;
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      float y = res[o] + p.bias[o];
      if (p.mode == 1) y = fmaxf(y * lt->scale[l], 0.f);
      p.out[dst * NO + o] = y;
    }
  }
}

int launch_conv_out(const ConvOutArgs& a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  const int n = (a.C + 255) / 256;
  DCF_CHECK(a.C % 4 == 0 && n >= 1 && n <= 4 && (a.NO == 1 || a.NO == 2), "conv_out: C=%d / NO=%d unsupported", a.C, a.NO);
  if (a.ln_w) {                                          // LayerNorm + ReLU of the trunk on load
    DCF_CHECK(a.ln_b, "conv_out: ln_b missing");
    const int strips = (a.rows + LC_STRIP - 1) / LC_STRIP;
    dim3 grid((strips + 3) / 4), blk(256);
    ProfScope prof("ln_conv_out", st, (8.0 + 6.0 * a.NO) * a.rows * a.C, 4.0 * a.rows * a.C);
#define LCO(NCH_)                                                                            \
    if (a.NO == 1) hipLaunchKernelGGL((k_ln_conv_out<NCH_, 1>), grid, blk, 0, st, a);        \
    else hipLaunchKernelGGL((k_ln_conv_out<NCH_, 2>), grid, blk, 0, st, a)
    const int nf = a.C / 256, tailc = a.C % 256;
#define LCT(NF_, TAIL_)                                                                            \
    if (a.NO == 1) hipLaunchKernelGGL((k_ln_conv_out_t<NF_, TAIL_, 1>), grid, blk, 0, st, a);      \
    else hipLaunchKernelGGL((k_ln_conv_out_t<NF_, TAIL_, 2>), grid, blk, 0, st, a)
    if (nf >= 1 && nf <= 2 && tailc == 0) {              // whole chunks
      if (nf == 1) { LCT(1, false); } else { LCT(2, false); }
    } else if (nf >= 1 && nf <= 2 && tailc <= 64) {      // whole chunks + a tail of one float per lane (288 = 256 + 32)
      if (nf == 1) { LCT(1, true); } else { LCT(2, true); }
    } else {
      switch (n) {
        case 1: LCO(1); break;
        case 2: LCO(2); break;
        case 3: LCO(3); break;
        default: LCO(4); break;
      }
    }
#undef LCT
#undef LCO
    DCF_HIP(hipGetLastError());
    return 0;
  }
  const int strips = (a.rows + CO_STRIP - 1) / CO_STRIP;
  dim3 grid((strips + 3) / 4), blk(256);
  ProfScope prof("conv_out", st, 6.0 * a.rows * a.C * a.NO, 4.0 * a.rows * a.C);
#define CO(NCH_)                                                                          \
  if (a.NO == 1) hipLaunchKernelGGL((k_conv_out<NCH_, 1>), grid, blk, 0, st, a);          \
  else hipLaunchKernelGGL((k_conv_out<NCH_, 2>), grid, blk, 0, st, a)
  switch (n) {
    case 1: CO(1); break;
    case 2: CO(2); break;
    case 3: CO(3); break;
    default: CO(4); break;
  }
#undef CO
  DCF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// refinement: stack the nearest-upsampled level logits (model.py:449-455) and map L -> 32
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_refine_in(RefineArgs p) {
  const int r = blockIdx.x * 64 + threadIdx.x;          // level-0 row (b, t)
  if (r >= p.B * p.T0) return;
  const LevelTable* lt = p.lt;
  const int b = r / p.T0, t = r - b * p.T0;
  const float m0 = p.mask_all[r] ? 1.f : 0.f;           // level 0 occupies rows [0, B*T0)
  float h[TCN_HID];
#pragma unroll
  for (int c = 0; c < TCN_HID; ++c) h[c] = p.b_in[c];
  for (int l = 0; l < p.n_levels; ++l) {
    // nearest source index floor(t * T_l / T_0) = t >> l  (T_0 = T_l * 2^l exactly)
    float u;
    if (p.stacked) {
      u = p.stacked[(int64_t)r * p.n_levels + l];
    } else {
      u = p.logits1[lt->start[l] + b * lt->T[l] + (t >> l)];
      if (l > 0) u *= m0;
    }
#pragma unroll
    for (int c = 0; c < TCN_HID; ++c) h[c] += p.w_in[l * TCN_HID + c] * u;
  }
  f32x4* dst = reinterpret_cast<f32x4*>(p.bufA + (int64_t)r * TCN_HID);
#pragma unroll
  for (int c = 0; c < TCN_HID / 4; ++c) dst[c] = f32x4{h[4 * c], h[4 * c + 1], h[4 * c + 2], h[4 * c + 3]};
}

__device__ __forceinline__ void load_row32(const float* __restrict__ src, float (&x)[TCN_HID]) {
  const f32x4* s = reinterpret_cast<const f32x4*>(src);
#pragma unroll
  for (int c = 0; c < TCN_HID / 4; ++c) {
    f32x4 v = s[c];
    x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
  }
}

// DilatedResidualLayer.forward (tcn.py:21-38): relu(dilated k3) -> 1x1 -> (x + out) * mask -> LayerNorm(32)
// One workgroup = 64 rows, four waves: lane = row, wave w owns output channels [8w, 8w+8) of both convolutions
// (8 accumulators per lane), so all four SIMDs of the CU work on the 64 rows (the one-wave version, 32
// accumulators per lane, ran 256 waves on 1024 SIMDs: 20 us per layer, now 4x fewer FMAs per wave).  The layer's
// 4 K weights are staged in LDS once per workgroup and read back as wave-uniform (broadcast) ds_read_b128; the
// 96 inputs of a row are parked in an LDS column so the reduction loops stay ROLLED (fully unrolled the compiler
// hoists every weight read and spills; scalar s_loads were latency bound).  Summation order per output is the
// same as a plain loop over (tap, channel), LayerNorm statistics are taken over channels 0..31 in order.
__global__ __launch_bounds__(256) void k_tcn_layer(const float* __restrict__ X, float* __restrict__ Y,
                                                    const float* __restrict__ wd, const float* __restrict__ bd,
                                                    const float* __restrict__ wp, const float* __restrict__ bp,
                                                    const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                    const uint8_t* __restrict__ mask, int B, int T0, int dil) {
  __shared__ f32x4 s_wd[3 * TCN_HID * TCN_HID / 4];
  __shared__ f32x4 s_wp[TCN_HID * TCN_HID / 4];
  __shared__ float s_x[3 * TCN_HID][64];
  __shared__ float s_h[TCN_HID][64];
  __shared__ float s_o[TCN_HID][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 3 * TCN_HID * TCN_HID / 4; i += 256) s_wd[i] = reinterpret_cast<const f32x4*>(wd)[i];
  for (int i = tid; i < TCN_HID * TCN_HID / 4; i += 256) s_wp[i] = reinterpret_cast<const f32x4*>(wp)[i];
  const int r = blockIdx.x * 64 + lane;
  const bool live = r < B * T0;
  const int t = live ? r % T0 : 0;
  // the row's 3 x 32 inputs = 24 float4; wave w stages float4 6w .. 6w+5
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int f = w * 6 + q, tap = f >> 3, c4 = f & 7;
    const int tt = t + (tap - 1) * dil;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live && tt >= 0 && tt < T0) v = *reinterpret_cast<const f32x4*>(X + (int64_t)(r + (tap - 1) * dil) * TCN_HID + c4 * 4);
    s_x[tap * TCN_HID + c4 * 4 + 0][lane] = v.x;
    s_x[tap * TCN_HID + c4 * 4 + 1][lane] = v.y;
    s_x[tap * TCN_HID + c4 * 4 + 2][lane] = v.z;
    s_x[tap * TCN_HID + c4 * 4 + 3][lane] = v.w;
  }
  __syncthreads();
  const int c0 = w * 8;
  float h[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) h[c] = bd[c0 + c];
#pragma unroll 4
  for (int k = 0; k < 3 * TCN_HID; ++k) {
    const float xv = s_x[k][lane];
    const f32x4 w0 = s_wd[k * (TCN_HID / 4) + w * 2], w1 = s_wd[k * (TCN_HID / 4) + w * 2 + 1];
    h[0] += w0.x * xv; h[1] += w0.y * xv; h[2] += w0.z * xv; h[3] += w0.w * xv;
    h[4] += w1.x * xv; h[5] += w1.y * xv; h[6] += w1.z * xv; h[7] += w1.w * xv;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) s_h[c0 + c][lane] = fmaxf(h[c], 0.f);
  __syncthreads();
  float o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = bp[c0 + c];
#pragma unroll 4
  for (int ci = 0; ci < TCN_HID; ++ci) {
    const float hv = s_h[ci][lane];
    const f32x4 w0 = s_wp[ci * (TCN_HID / 4) + w * 2], w1 = s_wp[ci * (TCN_HID / 4) + w * 2 + 1];
    o[0] += w0.x * hv; o[1] += w0.y * hv; o[2] += w0.z * hv; o[3] += w0.w * hv;
    o[4] += w1.x * hv; o[5] += w1.y * hv; o[6] += w1.z * hv; o[7] += w1.w * hv;
  }
  const float m = (live && mask[r]) ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    o[c] = (s_x[TCN_HID + c0 + c][lane] + o[c]) * m;       // residual = centre tap
    s_o[c0 + c][lane] = o[c];
  }
  __syncthreads();
  if (!live) return;
  float mean = 0.f;
#pragma unroll
  for (int c = 0; c < TCN_HID; ++c) mean += s_o[c][lane];
  mean *= (1.0f / TCN_HID);
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < TCN_HID; ++c) { const float d = s_o[c][lane] - mean; var += d * d; }
  const float rs = 1.0f / sqrtf(var * (1.0f / TCN_HID) + 1e-5f);
  f32x4* dst = reinterpret_cast<f32x4*>(Y + (int64_t)r * TCN_HID + c0);
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    f32x4 v;
    v.x = (o[4 * c] - mean) * rs * lnw[c0 + 4 * c] + lnb[c0 + 4 * c];
    v.y = (o[4 * c + 1] - mean) * rs * lnw[c0 + 4 * c + 1] + lnb[c0 + 4 * c + 1];
    v.z = (o[4 * c + 2] - mean) * rs * lnw[c0 + 4 * c + 2] + lnb[c0 + 4 * c + 2];
    v.w = (o[4 * c + 3] - mean) * rs * lnw[c0 + 4 * c + 3] + lnb[c0 + 4 * c + 3];
    dst[c] = v;
  }
}

// ------------------------------------------------------------------------------------------
// The same layer on the matrix cores (f16x3 arithmetic of the dense convolutions: every fp32 operand as hi + lo fp16
// planes, three v_mfma_f32_32x32x16_f16 products per step, operands pre-scaled by 2^4 / 2^8 -- gemm_bf16s.hip).
// A wave owns 32 consecutive rows; the products are taken TRANSPOSED, D[co][row] = sum_k W[co][k] X[k][row], so that
//   * the weights are the A operand: lane (co = lane & 31, h = lane >> 5) builds its fragments once per wave
//     (dilated conv: 6 chunks of 16 k = (tap, 16 channels); 1x1 convs: 2 chunks) and keeps them in registers,
//   * the activations are the B operand: lane (row, h) reads 8 consecutive channels of its own row -- 32-byte pieces of
//     the 128-byte rows, no LDS, no transposition,
//   * an accumulator lane holds, for its row, the 16 channels 8q + 4h + j (q, j = 0..3): element e = 4q + j.  Chunk c of
//     the NEXT product (hidden -> 1x1 conv, LayerNorm output -> conv_out) takes elements 8c .. 8c + 7 as they are, and the
//     weight fragments of that product are built with the same channel order (k position i of chunk c = channel
//     8 (2c + i / 4) + 4h + i % 4), so three chained GEMMs never leave the registers.
// The LayerNorm over the 32 channels of a row is 16 in-lane terms + one v_permlane32_swap.  LAST: refine.conv_out
// (1x1, * mask) runs in the same kernel and writes columns [E, E + 32) of the level-0 pyramid rows.
// 29 us per layer on the vector ALUs (19 TFLOP/s) -> 16 us.
// ------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr float TCN_SA = 16.f, TCN_SW = 256.f, TCN_UNSCALE = 1.f / 4096.f;

__device__ __forceinline__ void split8(const float (&x)[8], float s, h16x8& hi, h16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 hv = (_Float16)(x[i] * s);
    hi[i] = hv;
    lo[i] = (_Float16)__builtin_fmaf(x[i], s, -(float)hv);     // exact residual
  }
}
__device__ __forceinline__ f32x16 mma3(const h16x8& ah, const h16x8& al, const h16x8& bh, const h16x8& bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);     // smallest terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}
// fragments of a (ci, co) 32x32 weight for a product whose B operand is an accumulator (see above)
__device__ __forceinline__ void chain_frags(const float* __restrict__ w, int co, int h, h16x8 (&fh)[2], h16x8 (&fl)[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = w[(8 * (2 * c + i / 4) + 4 * h + i % 4) * TCN_HID + co];
    split8(x, TCN_SW, fh[c], fl[c]);
  }
}
__device__ __forceinline__ void chan4(const float* __restrict__ v, int h, f32x4 (&out)[4]) {   // v[8q + 4h .. + 3], q = 0..3
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * h);
}

// fragment pair f of a TCN layer (6 dilated-conv chunks, 2 chunks of conv_1x1, 2 of refine.conv_out) for lane (n, h)
__device__ __forceinline__ void tcn_frag(int f, const float* __restrict__ wd, const float* __restrict__ wp, const float* __restrict__ wo,
                                         int n, int h, h16x8& fh, h16x8& fl) {
  float x[8];
  if (f < 6) {                                           // chunk f = (tap, 16-channel half): k position i = channel 16 cc + 8 h + i
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = wd[((f >> 1) * TCN_HID + 16 * (f & 1) + 8 * h + i) * TCN_HID + n];
  } else {                                               // chained products: k position i of chunk c = channel 8 (2c + i / 4) + 4 h + i % 4
    const float* __restrict__ wsrc = f < 8 ? wp : wo;
    const int c = f & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = wsrc[(8 * (2 * c + i / 4) + 4 * h + i % 4) * TCN_HID + n];
  }
  split8(x, TCN_SW, fh, fl);
}

// the fragments of a layer once per model instead of once per workgroup (img [10][2][64] h16x8; pairs 8, 9 only with wo)
__global__ __launch_bounds__(64) void k_tcn_frag_image(const float* __restrict__ wd, const float* __restrict__ wp, const float* __restrict__ wo,
                                                        h16x8* __restrict__ img) {
  const int lane = threadIdx.x, f = blockIdx.x;
  if (f >= 8 && !wo) return;
  h16x8 fh, fl;
  tcn_frag(f, wd, wp, wo, lane & 31, lane >> 5, fh, fl);
  img[(f * 2 + 0) * 64 + lane] = fh;
  img[(f * 2 + 1) * 64 + lane] = fl;
}
int launch_tcn_frag_image(const float* wd, const float* wp, const float* wo, unsigned short* img, hipStream_t st) {
  DCF_CHECK(wd && wp && img, "launch_tcn_frag_image: null argument");
  hipLaunchKernelGGL(k_tcn_frag_image, dim3(10), dim3(64), 0, st, wd, wp, wo, reinterpret_cast<h16x8*>(img));
  DCF_HIP(hipGetLastError());
  return 0;
}

// frag != nullptr: the layer's fragment image (launch_tcn_frag_image); otherwise the workgroup builds the fragments itself
template <bool LAST>
__global__ __launch_bounds__(256) void k_tcn_layer_mfma(const float* __restrict__ X, float* __restrict__ Y, const h16x8* __restrict__ frag,
                                                         const float* __restrict__ wd, const float* __restrict__ bd,
                                                         const float* __restrict__ wp, const float* __restrict__ bp,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                         const uint8_t* __restrict__ mask, int B, int T0, int dil, int tiles_per_wave,
                                                         const float* __restrict__ wo, const float* __restrict__ bo,
                                                         float* __restrict__ F, int64_t ldf, int E, unsigned* __restrict__ status) {
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * 4 + wv;
  const int rows = B * T0;

  // weight fragments: the four waves of the workgroup build a quarter each (gather + split) and share them through LDS --
  // building all of them per wave cost as much as the wave's tile
  constexpr int NPAIR = LAST ? 10 : 8;                   // (hi, lo) fragment pairs: 6 dilated-conv chunks, 2 + 2 chained ones
  __shared__ h16x8 s_frag[NPAIR][2][64];
  h16x8 wd_h[6], wd_l[6], wp_h[2], wp_l[2], wo_h[2], wo_l[2];
  if (frag) {                                            // uniform
    if (wave * tiles_per_wave * 32 >= rows) return;
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) { wd_h[kc] = frag[(kc * 2) * 64 + lane]; wd_l[kc] = frag[(kc * 2 + 1) * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) { wp_h[c] = frag[((6 + c) * 2) * 64 + lane]; wp_l[c] = frag[((6 + c) * 2 + 1) * 64 + lane]; }
    if constexpr (LAST) {
#pragma unroll
      for (int c = 0; c < 2; ++c) { wo_h[c] = frag[((8 + c) * 2) * 64 + lane]; wo_l[c] = frag[((8 + c) * 2 + 1) * 64 + lane]; }
    }
  } else {
#pragma unroll
    for (int f = 0; f < NPAIR; ++f) {
      if ((f & 3) != wv) continue;
      h16x8 fh, fl;
      tcn_frag(f, wd, wp, wo, n, h, fh, fl);
      s_frag[f][0][lane] = fh;
      s_frag[f][1][lane] = fl;
    }
    __syncthreads();
    if (wave * tiles_per_wave * 32 >= rows) return;
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) { wd_h[kc] = s_frag[kc][0][lane]; wd_l[kc] = s_frag[kc][1][lane]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) { wp_h[c] = s_frag[6 + c][0][lane]; wp_l[c] = s_frag[6 + c][1][lane]; }
    if constexpr (LAST) {
#pragma unroll
      for (int c = 0; c < 2; ++c) { wo_h[c] = s_frag[8 + c][0][lane]; wo_l[c] = s_frag[8 + c][1][lane]; }
    }
  }
  f32x4 bd4[4], bp4[4], lw4[4], lb4[4], bo4[4];
  chan4(bd, h, bd4); chan4(bp, h, bp4); chan4(lnw, h, lw4); chan4(lnb, h, lb4);
  if constexpr (LAST) chan4(bo, h, bo4);

  bool bad = false;
  for (int it = 0; it < tiles_per_wave; ++it) {
    const int r = (wave * tiles_per_wave + it) * 32 + n;
    if (r - n >= rows) break;
    const bool live = r < rows;
    const int t = live ? r % T0 : 0;
    // ---- relu(dilated k3): the three taps of the row, zero outside the sequence
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 xin[3][2][2];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int tt = t + (tap - 1) * dil;
      const bool ok = live && tt >= 0 && tt < T0;
      const float* src = X + (int64_t)(ok ? r + (tap - 1) * dil : 0) * TCN_HID + 8 * h;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + 16 * cc + 4 * u);
          xin[tap][cc][u] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 res[4];                                        // residual = the row itself, in accumulator channel order
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(X + (int64_t)(live ? r : 0) * TCN_HID + 8 * q + 4 * h);
      res[q] = live ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float m = (live && mask[live ? r : 0]) ? 1.f : 0.f;
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) {
      const f32x4 a = xin[kc >> 1][kc & 1][0], b = xin[kc >> 1][kc & 1][1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      h16x8 xh, xl;
      split8(x, TCN_SA, xh, xl);
      acc = mma3(wd_h[kc], wd_l[kc], xh, xl, acc);
    }
    float hid[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = acc[e] * TCN_UNSCALE + bd4[e >> 2][e & 3];
      bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
      hid[e] = fmaxf(v, 0.f);
    }
    // ---- 1x1 conv, residual, mask
    f32x16 acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float x[8] = {hid[8 * c], hid[8 * c + 1], hid[8 * c + 2], hid[8 * c + 3], hid[8 * c + 4], hid[8 * c + 5], hid[8 * c + 6], hid[8 * c + 7]};
      h16x8 xh, xl;
      split8(x, TCN_SA, xh, xl);
      acc2 = mma3(wp_h[c], wp_l[c], xh, xl, acc2);
    }
    float o[16], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float v = acc2[e] * TCN_UNSCALE + bp4[e >> 2][e & 3];
      bad |= !(__builtin_fabsf(v) <= 3.4028234664e38f);
      o[e] = (res[e >> 2][e & 3] + v) * m;
      sum += o[e];
    }
    // ---- LayerNorm over the 32 channels of the row (lanes n and n + 32)
    const float mean = xor32_sum(sum) * (1.0f / TCN_HID);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = o[e] - mean; sq += d * d; }
    const float rs = 1.0f / sqrtf(xor32_sum(sq) * (1.0f / TCN_HID) + 1e-5f);
    float y[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) y[e] = (o[e] - mean) * rs * lw4[e >> 2][e & 3] + lb4[e >> 2][e & 3];
    if constexpr (!LAST) {
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(Y + (int64_t)r * TCN_HID + 8 * q + 4 * h) = f32x4{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
      }
    } else {
      // ---- refine.conv_out (1x1) * mask -> columns [E, E + 32) of the pyramid row
      f32x16 acc3;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float x[8] = {y[8 * c], y[8 * c + 1], y[8 * c + 2], y[8 * c + 3], y[8 * c + 4], y[8 * c + 5], y[8 * c + 6], y[8 * c + 7]};
        h16x8 xh, xl;
        split8(x, TCN_SA, xh, xl);
        acc3 = mma3(wo_h[c], wo_l[c], xh, xl, acc3);
      }
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float u = acc3[4 * q + j] * TCN_UNSCALE + bo4[q][j];
            bad |= !(__builtin_fabsf(u) <= 3.4028234664e38f);
            v[j] = u * m;
          }
          *reinterpret_cast<f32x4*>(F + (int64_t)r * ldf + E + 8 * q + 4 * h) = v;
        }
      }
    }
  }
  if (bad && status) atomicOr(status, 1u);
}

// ------------------------------------------------------------------------------------------
// The first NL layers of the TCN (dilations 1, 2, .. 2^(NL-1)) in ONE launch.  A layer by itself is latency-bound (a dependent
// load -> split -> MFMA -> LayerNorm chain per wave over rows of 128 bytes: 24 us per layer at 131 072 rows, 6.5 us at 16 384) and
// reads and writes every row; here a workgroup keeps a window of TS_ROWS consecutive rows of ONE sequence in LDS (two buffers,
// row pitch 36 floats: the 16-lane groups of a `ds_read_b128` then fall on 64 distinct banks), runs the layers on it back to
// back -- a barrier between two layers, the arithmetic of k_tcn_layer_mfma operation for operation -- and writes the rows whose
// whole receptive field (2^NL - 1 rows to either side) lay inside the window.  Rows outside the sequence are zero in every
// layer (the convolutions' zero padding), rows of the window's rim come out wrong and are never written.
// ------------------------------------------------------------------------------------------
constexpr int TS_ROWS = 256, TS_PITCH = 36, TS_MAXL = 5;
struct TcnStackArgs {
  const float* X;                      // [B*T0][32] the stack's input rows
  float* Y;                            // [B*T0][32] rows after layer NL - 1
  const h16x8* frag[TS_MAXL];          // the layers' fragment images (launch_tcn_frag_image)
  const float* bd[TS_MAXL]; const float* bp[TS_MAXL]; const float* lnw[TS_MAXL]; const float* lnb[TS_MAXL];
  const uint8_t* mask;                 // [B*T0]
  int B, T0;
  unsigned* status;
};

template <int NL>
__global__ __launch_bounds__(256) void k_tcn_stack(TcnStackArgs p) {
  static_assert(NL >= 2 && NL <= TS_MAXL, "2 .. 5 layers");
  constexpr int HALO = (1 << NL) - 1, VALID = TS_ROWS - 2 * HALO;
  extern __shared__ __attribute__((aligned(16))) float ts_lds[];               // [2][TS_ROWS][TS_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per_seq = (p.T0 + VALID - 1) / VALID;
  const int b = blockIdx.x / per_seq, tile = blockIdx.x - b * per_seq;
  const int t_lo = tile * VALID - HALO;                                        // sequence position of window row 0
  const int64_t seq0 = (int64_t)b * p.T0;
  // ---- the window: TS_ROWS x 32 floats, contiguous in memory (row pitch 32 floats): 16-byte pieces in thread order
#pragma unroll
  for (int k = 0; k < TS_ROWS * 8 / 256; ++k) {
    const int idx = k * 256 + tid, row = idx >> 3, c4 = idx & 7;
    const int t = t_lo + row;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < p.T0) v = *reinterpret_cast<const f32x4*>(p.X + (seq0 + t) * TCN_HID + 4 * c4);
    *reinterpret_cast<f32x4*>(ts_lds + row * TS_PITCH + 4 * c4) = v;
  }
  // the two 32-row blocks of this wave: window rows (2 wv + kb) 32 + n
  bool inseq[2];
  float mk[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int t = t_lo + (2 * wv + kb) * 32 + n;
    inseq[kb] = t >= 0 && t < p.T0;
    mk[kb] = (inseq[kb] && p.mask[seq0 + (inseq[kb] ? t : 0)]) ? 1.f : 0.f;
  }
  __syncthreads();
  bool bad = false;
  int cur = 0;
  // (a rolled loop: unrolled, the compiler requests every layer's fragments up front -- 512 registers and scratch; the layer's pointers
  // by a cascade of selects: a kernel-argument array indexed by a variable is copied to scratch)
  auto pick = [&](auto const (&arr)[TS_MAXL], int l) __attribute__((always_inline)) {
    auto r = arr[0];
#pragma unroll
    for (int k = 1; k < TS_MAXL; ++k) r = l == k ? arr[k] : r;
    return r;
  };
#pragma unroll 1
  for (int l = 0; l < NL; ++l) {
    const int dil = 1 << l;
    const float* in = ts_lds + cur * TS_ROWS * TS_PITCH;
    float* out = ts_lds + (cur ^ 1) * TS_ROWS * TS_PITCH;
    h16x8 wd_h[6], wd_l[6], wp_h[2], wp_l[2];
    const h16x8* frag = pick(p.frag, l);
#pragma unroll
    for (int kc = 0; kc < 6; ++kc) { wd_h[kc] = frag[(kc * 2) * 64 + lane]; wd_l[kc] = frag[(kc * 2 + 1) * 64 + lane]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) { wp_h[c] = frag[((6 + c) * 2) * 64 + lane]; wp_l[c] = frag[((6 + c) * 2 + 1) * 64 + lane]; }
    f32x4 bd4[4], bp4[4], lw4[4], lb4[4];
    chan4(pick(p.bd, l), h, bd4); chan4(pick(p.bp, l), h, bp4); chan4(pick(p.lnw, l), h, lw4); chan4(pick(p.lnb, l), h, lb4);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int i = (2 * wv + kb) * 32 + n;
      // ---- relu(dilated k3): the three taps of the row (rows beyond the window read as zero: such a row's result is not used)
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      f32x4 xin[3][2][2];
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int ii = i + (tap - 1) * dil;
        const bool ok = ii >= 0 && ii < TS_ROWS;
        const float* src = in + (ok ? ii : 0) * TS_PITCH + 8 * h;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 16 * cc + 4 * u);
            xin[tap][cc][u] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
      f32x4 res[4];                                      // residual = the row itself, in accumulator channel order
#pragma unroll
      for (int q = 0; q < 4; ++q) res[q] = *reinterpret_cast<const f32x4*>(in + i * TS_PITCH + 8 * q + 4 * h);
#pragma unroll
      for (int kc = 0; kc < 6; ++kc) {
        const f32x4 a = xin[kc >> 1][kc & 1][0], bq = xin[kc >> 1][kc & 1][1];
        const float x[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        h16x8 xh, xl;
        split8(x, TCN_SA, xh, xl);
        acc = mma3(wd_h[kc], wd_l[kc], xh, xl, acc);
      }
      float hid[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[e] * TCN_UNSCALE + bd4[e >> 2][e & 3];
        bad |= inseq[kb] && !(__builtin_fabsf(v) <= 3.4028234664e38f);
        hid[e] = fmaxf(v, 0.f);
      }
      // ---- 1x1 conv, residual, mask
      f32x16 acc2;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float x[8] = {hid[8 * c], hid[8 * c + 1], hid[8 * c + 2], hid[8 * c + 3], hid[8 * c + 4], hid[8 * c + 5], hid[8 * c + 6], hid[8 * c + 7]};
        h16x8 xh, xl;
        split8(x, TCN_SA, xh, xl);
        acc2 = mma3(wp_h[c], wp_l[c], xh, xl, acc2);
      }
      float o[16], sum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc2[e] * TCN_UNSCALE + bp4[e >> 2][e & 3];
        bad |= inseq[kb] && !(__builtin_fabsf(v) <= 3.4028234664e38f);
        o[e] = (res[e >> 2][e & 3] + v) * mk[kb];
        sum += o[e];
      }
      // ---- LayerNorm over the 32 channels of the row (lanes n and n + 32)
      const float mean = xor32_sum(sum) * (1.0f / TCN_HID);
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float d = o[e] - mean; sq += d * d; }
      const float rs = 1.0f / sqrtf(xor32_sum(sq) * (1.0f / TCN_HID) + 1e-5f);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = inseq[kb] ? (o[4 * q + j] - mean) * rs * lw4[q][j] + lb4[q][j] : 0.f;   // (outside the sequence: the next layer's zero padding)
        *reinterpret_cast<f32x4*>(out + i * TS_PITCH + 8 * q + 4 * h) = y;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  // ---- the rows whose receptive field lay inside the window
  const float* fin = ts_lds + cur * TS_ROWS * TS_PITCH;
#pragma unroll
  for (int k = 0; k < TS_ROWS * 8 / 256; ++k) {
    const int idx = k * 256 + tid, row = idx >> 3, c4 = idx & 7;
    const int t = t_lo + row;
    if (row >= HALO && row < HALO + VALID && t < p.T0)
      *reinterpret_cast<f32x4*>(p.Y + (seq0 + t) * TCN_HID + 4 * c4) = *reinterpret_cast<const f32x4*>(fin + row * TS_PITCH + 4 * c4);
  }
  if (bad && p.status) atomicOr(p.status, 1u);
}

// (timed and priced inside launch_refine's "refine_tcn" scope, which counts every layer of the branch)
template <int NL>
static int launch_tcn_stack_n(const TcnStackArgs& a, hipStream_t st) {
  constexpr int VALID = TS_ROWS - 2 * ((1 << NL) - 1);
  const unsigned grid = (unsigned)(a.B * ((a.T0 + VALID - 1) / VALID));
  const size_t lds = (size_t)2 * TS_ROWS * TS_PITCH * sizeof(float);        // 72 KiB: above the 64 KiB a kernel gets without asking
  static bool attr_set[64] = {};
  int dev = 0;
  DCF_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    DCF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tcn_stack<NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((k_tcn_stack<NL>), dim3(grid), dim3(256), lds, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

// |w| * 2^8 must stay inside fp16 for the kernel above: raises *flag otherwise (checked once per model, like the GEMM weights)
__global__ void k_f16_weight_range(const float* __restrict__ w, int n, unsigned* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(__builtin_fabsf(w[i]) * TCN_SW <= 65504.f)) atomicOr(flag, 1u);
}
int launch_f16_weight_range(const float* w, int n, unsigned* flag, hipStream_t st) {
  if (n <= 0 || !flag) return 0;
  hipLaunchKernelGGL(k_f16_weight_range, dim3((n + 255) / 256), dim3(256), 0, st, w, n, flag);
  DCF_HIP(hipGetLastError());
  return 0;
}

// conv_out (1x1, 32->32) * mask, written into columns [E, E+32) of the level-0 pyramid rows
__global__ __launch_bounds__(64) void k_refine_out(const float* __restrict__ X, const float* __restrict__ w,
                                                    const float* __restrict__ b, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ F, int64_t ldf, int E, int rows) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  float x[TCN_HID], o[TCN_HID];
  load_row32(X + (int64_t)r * TCN_HID, x);
#pragma unroll
  for (int c = 0; c < TCN_HID; ++c) o[c] = b[c];
#pragma unroll
  for (int ci = 0; ci < TCN_HID; ++ci) {
#pragma unroll
    for (int co = 0; co < TCN_HID; ++co) o[co] += w[ci * TCN_HID + co] * x[ci];
  }
  const float m = mask[r] ? 1.f : 0.f;
  f32x4* dst = reinterpret_cast<f32x4*>(F + (int64_t)r * ldf + E);
#pragma unroll
  for (int c = 0; c < TCN_HID / 4; ++c) dst[c] = f32x4{o[4 * c] * m, o[4 * c + 1] * m, o[4 * c + 2] * m, o[4 * c + 3] * m};
}

// masked_max_pool1d (blocks.py:31-47) of the 32 refined channels from level l-1 to level l:
// out = max over the VALID window entries, 0 if the window has none (see DESIGN.md on the filler).
__global__ __launch_bounds__(256) void k_refine_pool(float* __restrict__ F, int64_t ldf, int E,
                                                      const uint8_t* __restrict__ mask_in, int64_t in_row0,
                                                      int64_t out_row0, int B, int T_in) {
  const int To = T_in / 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;       // (row_out, channel quad)
  const int r = idx >> 3, cq = idx & 7;
  if (r >= B * To) return;
  const int b = r / To, i = r - b * To;
  f32x4 best = {0.f, 0.f, 0.f, 0.f};
  bool any = false;
#pragma unroll
  for (int d = -1; d <= 1; ++d) {
    const int t = 2 * i + d;
    if (t < 0 || t >= T_in) continue;
    const int64_t rin = (int64_t)b * T_in + t;
    if (!mask_in[rin]) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(F + (in_row0 + rin) * ldf + E + cq * 4);
    if (!any) best = v;
    else { best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w); }
    any = true;
  }
  *reinterpret_cast<f32x4*>(F + (out_row0 + r) * ldf + E + cq * 4) = best;
}

// All levels of that pooling chain in one launch: one workgroup per row i of the coarsest level.  It loads the
// 2W - 1 level-0 rows [iW - (W - 1), iW + W) (W = 2^(L-1)) into LDS and walks down the pyramid there; level l keeps
// W_l - 1 halo rows to the left of its W_l = W >> l owned rows (the window only reaches one row to the left), which
// neighbours recompute.  Replaces L - 1 dependent launches of k_refine_pool (7 x 4.8 us at L = 8).
__global__ __launch_bounds__(256) void k_refine_pool_all(float* __restrict__ F, int64_t ldf, int E,
                                                          const uint8_t* __restrict__ mask_all, const LevelTable* __restrict__ lt) {
  extern __shared__ float s_pool[];                    // two buffers of (2W - 1) rows x 32 channels
  const int L = lt->n_levels, B = lt->B;
  const int W = 1 << (L - 1);
  const int TL = lt->T[L - 1];
  const int b = blockIdx.x / TL, i = blockIdx.x - b * TL;
  (void)B;
  float* cur = s_pool;
  float* nxt = s_pool + (2 * W - 1) * TCN_HID;
  const int tid = threadIdx.x;
  {   // level 0 window
    const int t_lo = i * W - (W - 1);
    for (int idx = tid; idx < (2 * W - 1) * 8; idx += 256) {
      const int j = idx >> 3, cq = idx & 7;
      const int t = t_lo + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t >= 0) v = *reinterpret_cast<const f32x4*>(F + ((int64_t)lt->start[0] + (int64_t)b * lt->T[0] + t) * ldf + E + cq * 4);
      *reinterpret_cast<f32x4*>(cur + j * TCN_HID + cq * 4) = v;
    }
  }
  __syncthreads();
  for (int l = 1; l < L; ++l) {
    const int Wl = W >> l, hl = Wl - 1, Wp = W >> (l - 1), hp = Wp - 1;
    const int k_lo = i * Wl - hl;                        // first row of this level's window
    const int p_lo = i * Wp - hp;                        // first row of the previous level's window
    const uint8_t* mask_in = mask_all + lt->start[l - 1] + (int64_t)b * lt->T[l - 1];
    for (int idx = tid; idx < (2 * Wl - 1) * 8; idx += 256) {
      const int j = idx >> 3, cq = idx & 7;
      const int k = k_lo + j;
      f32x4 best = {0.f, 0.f, 0.f, 0.f};
      if (k >= 0) {
        bool any = false;
#pragma unroll
        for (int d = -1; d <= 1; ++d) {
          const int t = 2 * k + d;
          if (t < 0 || t >= lt->T[l - 1] || !mask_in[t]) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(cur + (t - p_lo) * TCN_HID + cq * 4);
          if (!any) best = v;
          else { best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w); }
          any = true;
        }
        if (j >= hl)                                     // owned row: goes to the pyramid buffer
          *reinterpret_cast<f32x4*>(F + ((int64_t)lt->start[l] + (int64_t)b * lt->T[l] + k) * ldf + E + cq * 4) = best;
      }
      *reinterpret_cast<f32x4*>(nxt + j * TCN_HID + cq * 4) = best;
    }
    __syncthreads();
    float* t_ = cur; cur = nxt; nxt = t_;
  }
}

// one level of the refined map's pooling chain by itself (dcf_hybrid_phase2 / 3: two pyramids, an exchange in the middle)
int launch_refine_pool(float* F, int64_t ldf, int E, const uint8_t* mask_in, int64_t in_row0, int64_t out_row0, int B, int T_in,
                       hipStream_t st) {
  const int n = B * (T_in / 2) * 8;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_refine_pool, dim3((n + 255) / 256), dim3(256), 0, st, F, ldf, E, mask_in, in_row0, out_row0, B, T_in);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_refine(const RefineArgs& a, const LevelTable& lt, hipStream_t st) {
  const int rows0 = a.B * a.T0;
  if (rows0 <= 0) return 0;
  DCF_CHECK(a.ldf % 4 == 0 && a.E % 4 == 0, "refine: ldf/E must be multiples of 4");
  dim3 g64((rows0 + 63) / 64), b64(64);
  ProfScope prof("refine_tcn", st, 2.0 * rows0 * (a.n_layers * 4096.0 + 32.0 * a.n_levels + 1024.0), 4.0 * rows0 * 64.0 * (a.n_layers + 2));
  hipLaunchKernelGGL(k_refine_in, g64, b64, 0, st, a);
  float* cur = a.bufA;
  float* nxt = a.bufB;
  // f16x3 mode: the layers run on the matrix cores, the last one carries conv_out
  const int tiles = (rows0 + 31) / 32;
  const int tpw = 1;                                     // 32-row tiles per wave (measured at 81 920 rows: 1 -> 0.180 ms for the branch, 2 -> 0.189,
                                                         // 4 -> 0.193, 8 -> 0.302: a tile is one dependent load -> split -> MFMA chain, more waves hide it best)
  const dim3 gm((((tiles + tpw - 1) / tpw) + 3) / 4);
  bool out_done = false;
  auto fimg = [&](int i) { return a.host_frag && a.host_frag[i] ? reinterpret_cast<const h16x8*>(a.host_frag[i]) : (const h16x8*)nullptr; };
  // the first layers (dilations 1 .. 16: a halo of 31 rows) as one launch over LDS windows; the rest -- dilations from 32 on, whose
  // halos would outgrow the window -- layer by layer
  int first = 0;
  {
    int nl = a.stack_layers < 0 ? 5 : a.stack_layers;
    if (nl > TS_MAXL) nl = TS_MAXL;
    if (nl > a.n_layers - 1) nl = a.n_layers - 1;        // (the last layer carries conv_out)
    bool have = a.f16 && nl >= 2 && a.T0 >= 1;
    for (int i = 0; i < nl && have; ++i) have = fimg(i) != nullptr;
    if (have) {
      TcnStackArgs sa{};
      sa.X = cur; sa.Y = nxt; sa.mask = a.mask_all; sa.B = a.B; sa.T0 = a.T0; sa.status = a.status;
      for (int i = 0; i < nl; ++i) { sa.frag[i] = fimg(i); sa.bd[i] = a.host_b_dil[i]; sa.bp[i] = a.host_b_pw[i]; sa.lnw[i] = a.host_ln_w[i]; sa.lnb[i] = a.host_ln_b[i]; }
      int rc;
      switch (nl) {
        case 2: rc = launch_tcn_stack_n<2>(sa, st); break;
        case 3: rc = launch_tcn_stack_n<3>(sa, st); break;
        case 4: rc = launch_tcn_stack_n<4>(sa, st); break;
        default: rc = launch_tcn_stack_n<5>(sa, st); break;
      }
      if (rc) return rc;                                // (the output buffer was not written: do not swap and carry on)
      float* t = cur; cur = nxt; nxt = t;
      first = nl;
    }
  }
  for (int i = first; i < a.n_layers; ++i) {
    DCF_CHECK(a.host_w_dil && a.host_w_dil[i], "refine: missing TCN layer %d", i);
    if (a.f16) {
      if (i + 1 < a.n_layers) {
        hipLaunchKernelGGL(k_tcn_layer_mfma<false>, gm, dim3(256), 0, st, (const float*)cur, nxt, fimg(i), a.host_w_dil[i], a.host_b_dil[i],
                           a.host_w_pw[i], a.host_b_pw[i], a.host_ln_w[i], a.host_ln_b[i], a.mask_all, a.B, a.T0, 1 << i, tpw,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (int64_t)0, 0, a.status);
      } else {
        hipLaunchKernelGGL(k_tcn_layer_mfma<true>, gm, dim3(256), 0, st, (const float*)cur, nxt, fimg(i), a.host_w_dil[i], a.host_b_dil[i],
                           a.host_w_pw[i], a.host_b_pw[i], a.host_ln_w[i], a.host_ln_b[i], a.mask_all, a.B, a.T0, 1 << i, tpw,
                           a.w_out, a.b_out, a.F, a.ldf, a.E, a.status);
        out_done = true;
      }
    } else {
      hipLaunchKernelGGL(k_tcn_layer, g64, dim3(256), 0, st, (const float*)cur, nxt, a.host_w_dil[i], a.host_b_dil[i], a.host_w_pw[i],
                         a.host_b_pw[i], a.host_ln_w[i], a.host_ln_b[i], a.mask_all, a.B, a.T0, 1 << i);
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  if (!out_done)
    hipLaunchKernelGGL(k_refine_out, g64, b64, 0, st, (const float*)cur, a.w_out, a.b_out, a.mask_all, a.F, a.ldf, a.E, rows0);
  if (a.n_levels > 1 && !a.stacked) {
    const int W = 1 << (a.n_levels - 1);
    const size_t lds = (size_t)2 * (2 * W - 1) * TCN_HID * sizeof(float);
    if (lds <= 64 * 1024 && lt.T[0] == lt.T[a.n_levels - 1] * W) {
      hipLaunchKernelGGL(k_refine_pool_all, dim3(a.B * lt.T[a.n_levels - 1]), dim3(256), lds, st, a.F, a.ldf, a.E, a.mask_all, a.lt);
    } else {                                             // very deep pyramids: level by level
      for (int l = 1; l < a.n_levels; ++l) {
        const int Tin = lt.T[l - 1];
        const int n = a.B * (Tin / 2) * 8;
        hipLaunchKernelGGL(k_refine_pool, dim3((n + 255) / 256), dim3(256), 0, st, a.F, a.ldf, a.E,
                           a.mask_all + lt.start[l - 1], (int64_t)lt.start[l - 1], (int64_t)lt.start[l], a.B, Tin);
      }
    }
  }
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
