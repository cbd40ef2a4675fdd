// Position-wise ("row") kernels of the grounding path: everything between the dense GEMMs.
//
// One 64-lane wavefront owns one clip position (a row of E channels, 16 B per lane per
// 256-channel chunk => every global access is a fully coalesced 1 KiB wave transaction)
// and walks a strip of consecutive positions so that the k3 depthwise convolutions / k3 s2
// max-pool can slide a 3-row window through registers: each input row is read and
// layer-normalised exactly once.  All of these kernels are HBM/L2-bandwidth bound.
#include <cstdlib>
#include "common.h"
#include "rowops.h"

namespace dcf {

// Output rows per wavefront: long strips amortise the window warm-up (2 extra rows), short strips expose
// more wavefronts when a pyramid level is small.
static inline int pick_strip(long rows) {
  constexpr long want = 4096;                // strips per launch; measured 1024: 2.47, 2048: 2.43, 4096: 2.40, 8192: 2.39 ms per step
  int s = 16;
  while (s > 2 && rows / s < want) s >>= 1;
  return s;
}

// ------------------------------------------------------------------------------------------
// masks
// ------------------------------------------------------------------------------------------
// nbr[r]: bit0 = row usable, bit1 = left neighbour (same batch) usable, bit2 = right.
__global__ void k_rowflags(const uint8_t* __restrict__ mask, uint8_t* __restrict__ nbr, int T, int rows) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int t = r % T;
  unsigned f = mask[r] ? 1u : 0u;
  if (t > 0 && mask[r - 1]) f |= 2u;
  if (t < T - 1 && mask[r + 1]) f |= 4u;
  nbr[r] = (uint8_t)f;
}

// All pyramid levels at once: mask_l[b][i] = mask_0[b][i << l] (l applications of the stride-2 MaskedConv1D mask rule
// mask[2i], blocks.py:101-105) and the k3 neighbour flags of every level.  rows are ordered [level][b][t]; start[l] =
// B * sum_{j<l} (T0 >> j).  One launch instead of 2L - 1.
__global__ void k_pyramid_masks(uint8_t* __restrict__ mask_all, uint8_t* __restrict__ nbr_all, int B, int T0, int L, int rows_all) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows_all) return;
  int l = 0, start = 0;
  while (l + 1 < L && r >= start + B * (T0 >> l)) { start += B * (T0 >> l); ++l; }
  const int T = T0 >> l;
  const int rel = r - start, b = rel / T, t = rel - b * T;
  const uint8_t* m0 = mask_all + (size_t)b * T0;            // level 0 occupies rows [0, B*T0)
  auto at = [&](int tt) { return m0[(size_t)tt << l] != 0; };
  const bool self = at(t);
  unsigned f = self ? 1u : 0u;
  if (t > 0 && at(t - 1)) f |= 2u;
  if (t < T - 1 && at(t + 1)) f |= 4u;
  nbr_all[r] = (uint8_t)f;
  if (l > 0) mask_all[r] = self ? 1 : 0;
}

// stride-2 MaskedConv1D mask: nearest downsample == mask[2i] (libs/modeling/blocks.py:101-105)
__global__ void k_mask_down(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int rows_out) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows_out) out[r] = in[2 * r];
}

// ------------------------------------------------------------------------------------------
// vid_map epilogue: X0[b,t,:] = m * (g * P1[t,:] + P2[t,:] + correl[b,t] * w3) + bias     (model.py:543-555)
// P1 = W[:, :D] . vid, P2 = W[:, D:2D] . shallow are query independent and computed once; w3 = W[:, 2D] is the
// column of the raw-score channel (opt.model.scat).  P1 is null for sfonly, P2 without msf, w3 without scat.
// ------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void k_vidmap_combine(const float* __restrict__ P1, const float* __restrict__ P2,
                                                         const float* __restrict__ bias, const float* __restrict__ gate,
                                                         const uint8_t* __restrict__ mask,
                                                         const float* __restrict__ w3, const float* __restrict__ correl,
                                                         float* __restrict__ X, int T, int rows, int E,
                                                         unsigned long long vmap) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  // row of the video's products: video (vmap >> 4b) & 15 of batch element b = r / T (0 for one video)
  const int t = r % T + (int)((vmap >> (4 * ((r / T) & 15))) & 15ull) * T;
  const float m = mask[r] ? 1.f : 0.f;
  const float g = gate[r];
  const float s = w3 ? correl[r] : 0.f;     // scat: the clip's raw sidekick score is one more input channel
  Row<NCH> a, b;
  if (P1 && g != 0.f) a.load(P1 + (int64_t)t * E, E, lane); else a.zero();     // (rows under a closed gate may not exist: GemmArgs::tile_skip)
  if (P2) b.load(P2 + (int64_t)t * E, E, lane); else b.zero();
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int c = 256 * j + 4 * lane;
    if (c < E) {
      f32x4 bb = *reinterpret_cast<const f32x4*>(bias + c);
      f32x4 v = g * a.v[j] + b.v[j];
      if (w3) v += s * *reinterpret_cast<const f32x4*>(w3 + c);
      a.v[j] = m * v + bb;
    }
  }
  a.store(X + (int64_t)r * E, E, lane);
}

// ------------------------------------------------------------------------------------------
// generic LayerNorm row kernel:  Y = [relu] LN(X) [+ pe[t] * mask]
// ------------------------------------------------------------------------------------------
// A wave takes LN_ROWS rows and requests all of them (and the affine parameters) before the first reduction: with one
// row per wave the kernel held 1 KiB per wave in flight and ran at 4.3 TB/s.
constexpr int LN_ROWS = 4;
template <int NCH>
__global__ __launch_bounds__(256) void k_ln(LnArgs p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * LN_ROWS;
  if (r0 >= p.rows) return;
  Row<NCH> x[LN_ROWS];
#pragma unroll
  for (int u = 0; u < LN_ROWS; ++u) {
    const int r = r0 + u < p.rows ? r0 + u : p.rows - 1;
    x[u].load(p.X + (int64_t)r * p.ldx, p.C, lane);
  }
  RowParam<NCH> w, b;
  w.init(p.skip_ln ? nullptr : p.w, p.C, lane);
  b.init(p.skip_ln ? nullptr : p.b, p.C, lane);
  const bool relu = p.relu;
#pragma unroll
  for (int u = 0; u < LN_ROWS; ++u) {
    const int r = r0 + u;
    if (r >= p.rows) break;
    if (!p.skip_ln) row_layernorm(x[u], p.C, lane, w, b);
    const float* pe = nullptr;
    if (p.pe && p.mask[r]) pe = p.pe + (int64_t)(r % p.T) * p.C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int c = 256 * j + 4 * lane;
      if (c < p.C) {
        f32x4 v = x[u].v[j];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (pe) v += *reinterpret_cast<const f32x4*>(pe + c);
        x[u].v[j] = v;
      }
    }
    x[u].store(p.Y + (int64_t)r * p.ldy, p.C, lane);
  }
}

// ------------------------------------------------------------------------------------------
// helpers for the sliding-window kernels
// ------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ void axpy3(Row<NCH>& out, const Row<NCH>& a, const Row<NCH>& b, const Row<NCH>& c,
                                      const float* __restrict__ w /* [3][C] */, int C, int lane) {
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int ch = 256 * j + 4 * lane;
    if (ch < C) {
      f32x4 w0 = *reinterpret_cast<const f32x4*>(w + ch);
      f32x4 w1 = *reinterpret_cast<const f32x4*>(w + C + ch);
      f32x4 w2 = *reinterpret_cast<const f32x4*>(w + 2 * C + ch);
      out.v[j] = w0 * a.v[j] + w1 * b.v[j] + w2 * c.v[j];
    } else {
      out.v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// depthwise k3 taps as RowParams ([3][C] in memory)
template <int NCH>
struct Taps3 {
  RowParam<NCH> t[3];
  __device__ __forceinline__ void init(const float* __restrict__ w, int C, int lane) {
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k].init(w + k * C, C, lane);
  }
};
template <int NCH>
__device__ __forceinline__ void axpy3(Row<NCH>& out, const Row<NCH>& a, const Row<NCH>& b, const Row<NCH>& c,
                                      const Taps3<NCH>& w, int C, int lane) {
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (256 * j + 4 * lane < C) out.v[j] = w.t[0].get(j, lane) * a.v[j] + w.t[1].get(j, lane) * b.v[j] + w.t[2].get(j, lane) * c.v[j];
    else out.v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Strip kernels below: a wave walks a strip of consecutive rows.  The validity of every input row of the strip (<= 34 rows)
// is fetched up front -- one mask byte per lane, one ballot -- and raw rows are requested TWO rows ahead of their use
// (unconditionally, address clamped into the sequence), so a wave keeps 2 KiB of row loads in flight across the LayerNorm
// reductions and stores of the current row instead of one dependent round trip per row, mask byte and parameter vector.
// ------------------------------------------------------------------------------------------
// TransformerDecoder front half (libs/modeling/blocks.py:632-645, :513-516):
//   q  = x * m
//   Qc = q_norm( dwconv3( ln_xattn_q(q) * m ) )          -> input of the query projection
//   Xa = adaln(q * m)  (LayerNorm without affine; the identity in xattn_mode 'affine') -> modulated later by the xattn output
// ------------------------------------------------------------------------------------------
template <int NCH, bool FULL>
__global__ __launch_bounds__(256) void k_dec_pre(DecPreArgs p) {
  const int lane = threadIdx.x & 63;
  const int STRIP = p.strip;
  const int strips_per_b = (p.T + STRIP - 1) / STRIP;
  const int s = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the strip loop is uniform
  if (s >= strips_per_b * p.B) return;
  const int b = s / strips_per_b;
  const int t0 = (s % strips_per_b) * STRIP;
  const int t1 = min(t0 + STRIP, p.T);
  const int64_t base = (int64_t)b * p.T;
  const int C = FULL ? 256 * NCH : p.C;                  // FULL: every lane chunk exists, no per-lane channel predicate

  RowParam<NCH> lnw, lnb, qnw, qnb, none;
  Taps3<NCH> dw;
  lnw.init(p.ln_q_w, C, lane); lnb.init(p.ln_q_b, C, lane);
  qnw.init(p.qn_w, C, lane); qnb.init(p.qn_b, C, lane);
  none.init(nullptr, C, lane);
  dw.init(p.dw, C, lane);

  // rows t0-1 .. t1: lane l looks at row t0 - 1 + l
  unsigned long long vm, lm = ~0ull, rm = ~0ull;
  {
    const int t = t0 - 1 + lane;
    const bool in = lane < t1 - t0 + 2 && t >= 0 && t < p.T;
    vm = __ballot(in && p.mask[base + t] != 0);
    if (p.nbr) {   // pyramid mode: rows of several sequences back to back, the flags of row t say which neighbours belong to it
      const unsigned f = in ? p.nbr[base + t] : 0u;
      lm = __ballot((f & 2u) != 0);
      rm = __ballot((f & 4u) != 0);
    }
  }
  auto fetch = [&](int t, Row<NCH>& raw) __attribute__((always_inline)) {
    const int tc = t < 0 ? 0 : (t < p.T ? t : p.T - 1);
    raw.load(p.X + (base + tc) * p.ldx, C, lane);
  };
  // normalised-and-masked row (conv input), zero outside [0,T) or where the mask is 0
  auto norm = [&](int t, Row<NCH>& raw, Row<NCH>& ln) __attribute__((always_inline)) {
    if ((vm >> (t - t0 + 1)) & 1ull) {
      ln = raw;
      row_layernorm(ln, C, lane, lnw, lnb);
    } else {
      raw.zero();
      ln.zero();
    }
  };

  Row<NCH> raw_c, raw_n, prev, cur, nxt, x0, x1;
  fetch(t0 - 1, raw_n); fetch(t0, raw_c); fetch(t0 + 1, x0); fetch(t0 + 2, x1);
  norm(t0 - 1, raw_n, prev);
  norm(t0, raw_c, cur);
  auto emit = [&](int t, Row<NCH>& buf) __attribute__((always_inline)) {
    raw_n = buf;                                           // row t + 1, requested two iterations ago
    if (t + 3 <= t1) fetch(t + 3, buf);
    norm(t + 1, raw_n, nxt);
    Row<NCH> q;
    {
      Row<NCH> pz = prev, nz = nxt;
      if (!((lm >> (t - t0 + 1)) & 1ull)) pz.zero();
      if (!((rm >> (t - t0 + 1)) & 1ull)) nz.zero();
      axpy3(q, pz, cur, nz, dw, C, lane);
    }
    row_layernorm(q, C, lane, qnw, qnb);
    q.store(p.Qc + (base + t) * (int64_t)C, C, lane);
    Row<NCH> xa = raw_c;                       // already zero where masked
    if (!p.affine) row_layernorm(xa, C, lane, none, none);        // 'affine': nn.Identity (blocks.py:625-626)
    xa.store(p.Xa + (base + t) * (int64_t)C, C, lane);
    prev = cur; cur = nxt; raw_c = raw_n;
  };
  for (int t = t0; t < t1; t += 2) {
    emit(t, x0);
    if (t + 1 < t1) emit(t + 1, x1);
  }
}

// ------------------------------------------------------------------------------------------
// TransformerDecoder middle (blocks.py:643-648):  q3 = Xa * scale + shift,  Xn = ln_ffn(q3)
// H = xattn output (rows, 2C): scale = H[:, :C], shift = H[:, C:]
// ------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void k_dec_mid(const float* __restrict__ Xa, const float* __restrict__ H,
                                                  const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                  float* __restrict__ Q3, float* __restrict__ Xn, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  Row<NCH> x, sc, sh;
  x.load(Xa + (int64_t)r * C, C, lane);
  sc.load(H + (int64_t)r * 2 * C, C, lane);
  sh.load(H + (int64_t)r * 2 * C + C, C, lane);
#pragma unroll
  for (int j = 0; j < NCH; ++j) x.v[j] = x.v[j] * sc.v[j] + sh.v[j];
  x.store(Q3 + (int64_t)r * C, C, lane);
  row_layernorm(x, C, lane, ln_w, ln_b);
  x.store(Xn + (int64_t)r * C, C, lane);
}

// ------------------------------------------------------------------------------------------
// TransformerEncoder front half (blocks.py:578-585, :462-469) for conv stride S in {1, 2}:
//   x    = x * m
//   skip = S == 2 ? masked_max_pool1d(x, m)  (blocks.py:31-47)  : x
//   xn   = ln_attn(x);  {q,k,v}c = {q,k,v}_norm( dwconv3_strideS( xn * m ) )
// Output row i reads input rows S*i-1, S*i, S*i+1.
// ------------------------------------------------------------------------------------------
template <int NCH, int S, bool FULL>
__global__ __launch_bounds__(256) void k_enc_pre(EncPreArgs p) {
  const int lane = threadIdx.x & 63;
  const int To = p.T_in / S;
  const int STRIP = p.strip;
  const int strips_per_b = (To + STRIP - 1) / STRIP;
  const int s = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the strip loop is uniform
  if (s >= strips_per_b * p.B) return;
  const int b = s / strips_per_b;
  const int i0 = (s % strips_per_b) * STRIP;
  const int i1 = min(i0 + STRIP, To);
  const int64_t ibase = (int64_t)b * p.T_in;
  const int64_t obase = (int64_t)b * To;
  const int C = FULL ? 256 * NCH : p.C;                  // FULL: every lane chunk exists, no per-lane channel predicate

  RowParam<NCH> lnw, lnb, qnw, qnb, knw, knb, vnw, vnb;
  Taps3<NCH> dwq, dwk, dwv;
  lnw.init(p.ln_w, C, lane); lnb.init(p.ln_b, C, lane);
  qnw.init(p.qn_w, C, lane); qnb.init(p.qn_b, C, lane);
  knw.init(p.kn_w, C, lane); knb.init(p.kn_b, C, lane);
  vnw.init(p.vn_w, C, lane); vnb.init(p.vn_b, C, lane);
  dwq.init(p.dw_q, C, lane); dwk.init(p.dw_k, C, lane); dwv.init(p.dw_v, C, lane);

  // input rows tb .. tl of the strip (<= 2 * 16 + 1): lane l looks at row tb + l
  const int tb = S * i0 - 1, tl = S * (i1 - 1) + 1;
  unsigned long long vm;
  {
    const int t = tb + lane;
    vm = __ballot(t <= tl && t >= 0 && t < p.T_in && p.mask_in[ibase + t] != 0);
  }
  auto fetch = [&](int t, Row<NCH>& raw) __attribute__((always_inline)) {
    const int tc = t < 0 ? 0 : (t < p.T_in ? t : p.T_in - 1);
    raw.load(p.X + (ibase + tc) * p.ldx, C, lane);
  };
  auto norm = [&](int t, Row<NCH>& raw, Row<NCH>& ln) __attribute__((always_inline)) -> bool {
    const bool valid = (vm >> (t - tb)) & 1ull;
    if (valid) {
      ln = raw;
      row_layernorm(ln, C, lane, lnw, lnb);
    } else {
      raw.zero();
      ln.zero();
    }
    return valid;
  };

  Row<NCH> rp, rc, rn, lp, lc, ln_;
  bool vp, vc = false, vn;
  auto emit = [&](int i) __attribute__((always_inline)) {
    Row<NCH> o;
    axpy3(o, lp, lc, ln_, dwq, C, lane);
    row_layernorm(o, C, lane, qnw, qnb);
    o.store(p.Qc + (obase + i) * (int64_t)C, C, lane);
    axpy3(o, lp, lc, ln_, dwk, C, lane);
    row_layernorm(o, C, lane, knw, knb);
    o.store(p.Kc + (obase + i) * (int64_t)C, C, lane);
    axpy3(o, lp, lc, ln_, dwv, C, lane);
    row_layernorm(o, C, lane, vnw, vnb);
    o.store(p.Vc + (obase + i) * (int64_t)C, C, lane);
    if constexpr (S == 2) {
      // max over the valid window entries; 0 when the output position itself is padded
      // (its mask is mask_in[2i] == vc) -- the global-min filler never wins, see DESIGN.md
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        f32x4 m = rc.v[j];
        if (vp) { m.x = fmaxf(m.x, rp.v[j].x); m.y = fmaxf(m.y, rp.v[j].y); m.z = fmaxf(m.z, rp.v[j].z); m.w = fmaxf(m.w, rp.v[j].w); }
        if (vn) { m.x = fmaxf(m.x, rn.v[j].x); m.y = fmaxf(m.y, rn.v[j].y); m.z = fmaxf(m.z, rn.v[j].z); m.w = fmaxf(m.w, rn.v[j].w); }
        o.v[j] = vc ? m : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      o.store(p.Skip + (obase + i) * (int64_t)C, C, lane);
    }
  };

  Row<NCH> x0, x1;
  if constexpr (S == 2) {
    fetch(tb, rp); fetch(tb + 1, x0); fetch(tb + 2, x1);
    vp = norm(tb, rp, lp);
    for (int i = i0; i < i1; ++i) {
      rc = x0; rn = x1;                                    // rows 2i, 2i + 1, requested one iteration ago
      if (i + 1 < i1) { fetch(2 * i + 2, x0); fetch(2 * i + 3, x1); }
      vc = norm(2 * i, rc, lc);
      vn = norm(2 * i + 1, rn, ln_);
      emit(i);
      rp = rn; lp = ln_; vp = vn;
    }
  } else {
    fetch(tb, rp); fetch(tb + 1, rc); fetch(tb + 2, x0); fetch(tb + 3, x1);
    vp = norm(tb, rp, lp);
    vc = norm(tb + 1, rc, lc);
    auto one = [&](int i, Row<NCH>& buf) __attribute__((always_inline)) {
      rn = buf;                                            // row i + 1, requested two iterations ago
      if (i + 3 <= tl) fetch(i + 3, buf);
      vn = norm(i + 1, rn, ln_);
      emit(i);
      rp = rc; lp = lc; vp = vc;
      rc = rn; lc = ln_; vc = vn;
    };
    for (int i = i0; i < i1; i += 2) {
      one(i, x0);
      if (i + 1 < i1) one(i + 1, x1);
    }
  }
}

// ------------------------------------------------------------------------------------------
// text side of the cross attention (blocks.py:639): kvn = ln_xattn_kv(text).  The encoded
// text arrives in the reference layout (1, TE, Lk) channel-major, one tensor per query;
// output is token-major (B * Lkmax, TE) zero-padded, plus the key mask.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_text_ln(TextLnArgs p) {
  const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int len = p.meta.len[b];
  float* out = p.out + ((int64_t)b * p.Lkmax + j) * p.TE;
  if (j >= len) {
    for (int c = lane; c < p.TE; c += 64) out[c] = 0.f;
    if (lane == 0 && p.kvmask) p.kvmask[b * p.Lkmax + j] = 0;
    return;
  }
  const float* src = p.meta.text[b];
  float s = 0.f;
  for (int c = lane; c < p.TE; c += 64) s += src[(int64_t)c * len + j];
  const float mean = wave_sum(s) / (float)p.TE;
  float sq = 0.f;
  for (int c = lane; c < p.TE; c += 64) { float d = src[(int64_t)c * len + j] - mean; sq += d * d; }
  const float rs = 1.0f / sqrtf(wave_sum(sq) / (float)p.TE + 1e-5f);
  for (int c = lane; c < p.TE; c += 64) out[c] = (src[(int64_t)c * len + j] - mean) * rs * p.w[c] + p.b[c];
  if (lane == 0 && p.kvmask) p.kvmask[b * p.Lkmax + j] = p.meta.text_mask[b] ? p.meta.text_mask[b][j] : 1;
}


// ------------------------------------------------------------------------------------------
// TextTransformer front end (text_net.py:166-181): MaskedConv1D 1x1 on x * mask (the bias is added at padded
// tokens too, blocks.py:63-106), + pe * mask, background token prepended, mask = cat(mask[:1], mask).
// One workgroup per output row; a few hundred KFLOP per query, T independent.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float text_embed_one(const TextEmbedArgs& p, int t, int c, float mk) {
  float acc;
  if (p.W) {
    const float* w = p.W + (int64_t)c * p.Ct;
    acc = 0.f;
    for (int k = 0; k < p.Ct; ++k) acc += w[k] * (p.tokens[(int64_t)k * p.Lq + t] * mk);
    acc += p.bias ? p.bias[c] : 0.f;
  } else {
    acc = p.tokens[(int64_t)c * p.Lq + t];               // TextIdentity without embd_fc: the raw token, not masked
  }
  if (p.pe) acc += p.pe[(int64_t)t * p.TE + c] * mk;
  return acc;
}

__global__ __launch_bounds__(256) void k_text_embed(TextEmbedArgs p) {
  const int j = blockIdx.x;                              // output row
  const int off = (p.bkgd || p.pool) ? 1 : 0;
  float* out = p.X + (int64_t)j * p.TE;
  if (j < off) {
    if (p.pool) {
      // masked_avg_pool1d (blocks.py:10-17): sum_t x[t] * mask[t] / sum_t mask[t], tokens summed in order
      float n = 0.f;
      for (int t = 0; t < p.Lq; ++t) n += (!p.mask || p.mask[t]) ? 1.f : 0.f;
      for (int c = threadIdx.x; c < p.TE; c += 256) {
        float s = 0.f;
        for (int t = 0; t < p.Lq; ++t) {
          const float mk = (!p.mask || p.mask[t]) ? 1.f : 0.f;
          s += text_embed_one(p, t, c, mk) * mk;
        }
        out[c] = s / n;
      }
    } else {
      for (int c = threadIdx.x; c < p.TE; c += 256) out[c] = p.bkgd[c];
    }
    if (threadIdx.x == 0) p.mask_out[0] = p.mask ? p.mask[0] : 1;
    return;
  }
  const int t = j - off;
  const float mk = (!p.mask || p.mask[t]) ? 1.f : 0.f;
  for (int c = threadIdx.x; c < p.TE; c += 256) out[c] = text_embed_one(p, t, c, mk);
  if (threadIdx.x == 0) p.mask_out[j] = mk != 0.f;
}

__global__ void k_mask_rows(float* __restrict__ X, const uint8_t* __restrict__ mask, int rows, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  if (!mask[i / C]) X[i] = 0.f;
}

__global__ void k_rows_to_chanmajor(const float* __restrict__ X, float* __restrict__ out, int rows, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // index into out: c * rows + r
  if (i >= rows * C) return;
  const int c = i / rows, r = i - c * rows;
  out[i] = X[(int64_t)r * C + c];
}

// A operand of a k5 / stride 2 / padding 2 MaskedConv1D (vid_net.stride > 1, video_net.py:62-70) as rows:
// col[b * T/2 + t][j * C + c] = (x * mask)[b][2 t - 2 + j][c], zero outside the sequence.  One thread per 4 channels of a tap.
__global__ void k_im2col5s2(const float* __restrict__ X, int64_t ldx, const uint8_t* __restrict__ mask, float* __restrict__ col,
                            int B, int T, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4, To = T / 2;
  if (i >= (int64_t)B * To * 5 * c4) return;
  const int c = (int)(i % c4) * 4;
  const int j = (int)((i / c4) % 5);
  const int64_t ro = i / (5 * c4);
  const int b = (int)(ro / To), t = (int)(ro - (int64_t)b * To);
  const int u = 2 * t - 2 + j;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (u >= 0 && u < T && mask[(int64_t)b * T + u]) v = *reinterpret_cast<const float4*>(X + ((int64_t)b * T + u) * ldx + c);
  *reinterpret_cast<float4*>(col + ro * 5 * C + (int64_t)j * C + c) = v;
}

// a pool_only branch layer (video_net.py:107-109): depthwise k3 MaskedConv1D, stride 1 or 2, padding 1, no bias; the output is
// NOT masked (blocks.py:99), the mask is sampled at the stride (blocks.py:101-105).  w: [3][C] (tap-major).
__global__ void k_dwconv3(const float* __restrict__ X, int64_t ldx, const uint8_t* __restrict__ mask, const float* __restrict__ w,
                          float* __restrict__ Y, int64_t ldy, int B, int T, int stride, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4, To = T / stride;
  if (i >= (int64_t)B * To * c4) return;
  const int c = (int)(i % c4) * 4;
  const int64_t ro = i / c4;
  const int b = (int)(ro / To), t = (int)(ro - (int64_t)b * To);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int u = stride * t - 1 + j;
    if (u < 0 || u >= T || !mask[(int64_t)b * T + u]) continue;
    const float4 x = *reinterpret_cast<const float4*>(X + ((int64_t)b * T + u) * ldx + c);
    const float4 ww = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c);
    acc.x = __builtin_fmaf(x.x, ww.x, acc.x); acc.y = __builtin_fmaf(x.y, ww.y, acc.y);
    acc.z = __builtin_fmaf(x.z, ww.z, acc.z); acc.w = __builtin_fmaf(x.w, ww.w, acc.w);
  }
  *reinterpret_cast<float4*>(Y + ro * ldy + c) = acc;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
#define DISPATCH_NCH(C, ...)                                                      \
  do {                                                                            \
    int _n = ((C) + 255) / 256;                                                   \
    DCF_CHECK((C) % 4 == 0 && _n >= 1 && _n <= 4, "channels=%d unsupported (need C %% 4 == 0, C <= 1024)", (int)(C)); \
    switch (_n) {                                                                 \
      case 1: { constexpr int NCH = 1; __VA_ARGS__; } break;                      \
      case 2: { constexpr int NCH = 2; __VA_ARGS__; } break;                      \
      case 3: { constexpr int NCH = 3; __VA_ARGS__; } break;                      \
      default: { constexpr int NCH = 4; __VA_ARGS__; } break;                     \
    }                                                                             \
    DCF_HIP(hipGetLastError());                                                   \
  } while (0)
// the same with `constexpr bool FULL` = "C is a whole number of 256-channel chunks" defined for the call
#define DISPATCH_NCH_FULL(C, ...)                                                 \
  do {                                                                            \
    if ((C) % 256 == 0) { constexpr bool FULL = true; DISPATCH_NCH(C, __VA_ARGS__); }    \
    else { constexpr bool FULL = false; DISPATCH_NCH(C, __VA_ARGS__); }           \
  } while (0)

int launch_rowflags(const uint8_t* mask, uint8_t* nbr, int T, int rows, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_rowflags, dim3((rows + 255) / 256), dim3(256), 0, st, mask, nbr, T, rows);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_pyramid_masks(uint8_t* mask_all, uint8_t* nbr_all, int B, int T0, int L, int rows_all, hipStream_t st) {
  if (rows_all <= 0) return 0;
  hipLaunchKernelGGL(k_pyramid_masks, dim3((rows_all + 255) / 256), dim3(256), 0, st, mask_all, nbr_all, B, T0, L, rows_all);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_mask_down(const uint8_t* in, uint8_t* out, int rows_out, hipStream_t st) {
  if (rows_out <= 0) return 0;
  hipLaunchKernelGGL(k_mask_down, dim3((rows_out + 255) / 256), dim3(256), 0, st, in, out, rows_out);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_vidmap_combine(const float* P1, const float* P2, const float* bias, const float* gate, const uint8_t* mask,
                          const float* w3, const float* correl, float* X, int T, int rows, int E, unsigned long long vmap,
                          hipStream_t st) {
  if (rows <= 0) return 0;
  ProfScope prof("vidmap_combine", st, 3.0 * rows * E, 4.0 * 3.0 * rows * E);
  DISPATCH_NCH(E, hipLaunchKernelGGL((k_vidmap_combine<NCH>), dim3((rows + 3) / 4), dim3(256), 0, st, P1, P2, bias, gate,
                                     mask, w3, correl, X, T, rows, E, vmap));
  return 0;
}

int launch_ln(const LnArgs& a, hipStream_t st) {
  if (a.rows <= 0) return 0;
  DCF_CHECK(a.ldx % 4 == 0 && a.ldy % 4 == 0, "launch_ln: row pitch must be a multiple of 4");
  DCF_CHECK(!a.pe || (a.mask && a.T > 0), "launch_ln: pe needs mask and T");
  ProfScope prof("layernorm", st, 8.0 * a.rows * a.C, 4.0 * a.rows * a.C * (a.pe ? 3.0 : 2.0));
  DISPATCH_NCH(a.C, hipLaunchKernelGGL((k_ln<NCH>), dim3((a.rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS)), dim3(256), 0, st, a));
  return 0;
}

int launch_dec_pre(const DecPreArgs& a_, hipStream_t st) {
  if (a_.B * a_.T <= 0) return 0;
  DecPreArgs a = a_;
  a.strip = pick_strip((long)a.B * a.T);
  const int STRIP = a.strip;
  int strips = a.B * ((a.T + STRIP - 1) / STRIP);
  ProfScope prof("dec_pre", st, 30.0 * a.B * a.T * a.C, 4.0 * 3.0 * a.B * a.T * a.C);
  DISPATCH_NCH_FULL(a.C, hipLaunchKernelGGL((k_dec_pre<NCH, FULL>), dim3((strips + 3) / 4), dim3(256), 0, st, a));
  return 0;
}

int launch_dec_mid(const float* Xa, const float* H, const float* ln_w, const float* ln_b, float* Q3, float* Xn, int rows,
                   int C, hipStream_t st) {
  if (rows <= 0) return 0;
  ProfScope prof("dec_mid", st, 10.0 * rows * C, 4.0 * 5.0 * rows * C);
  DISPATCH_NCH(C, hipLaunchKernelGGL((k_dec_mid<NCH>), dim3((rows + 3) / 4), dim3(256), 0, st, Xa, H, ln_w, ln_b, Q3, Xn,
                                     rows, C));
  return 0;
}

int launch_enc_pre(const EncPreArgs& a_, int stride, hipStream_t st) {
  DCF_CHECK(stride == 1 || stride == 2, "enc_pre: stride %d unsupported", stride);
  DCF_CHECK(a_.T_in % stride == 0, "enc_pre: T_in %% stride != 0");
  if (a_.B * a_.T_in <= 0) return 0;
  EncPreArgs a = a_;
  int To = a.T_in / stride;
  a.strip = pick_strip((long)a.B * To);
  const int STRIP = a.strip;
  int strips = a.B * ((To + STRIP - 1) / STRIP);
  ProfScope prof(stride == 1 ? "enc_pre_s1" : "enc_pre_s2", st, 50.0 * a.B * To * a.C,
                 4.0 * a.C * a.B * ((double)a.T_in + (stride == 1 ? 3.0 : 4.0) * To));
  if (stride == 1) {
    DISPATCH_NCH_FULL(a.C, hipLaunchKernelGGL((k_enc_pre<NCH, 1, FULL>), dim3((strips + 3) / 4), dim3(256), 0, st, a));
  } else {
    DCF_CHECK(a.Skip, "enc_pre: stride 2 needs a skip buffer");
    DISPATCH_NCH_FULL(a.C, hipLaunchKernelGGL((k_enc_pre<NCH, 2, FULL>), dim3((strips + 3) / 4), dim3(256), 0, st, a));
  }
  return 0;
}

int launch_text_embed(const TextEmbedArgs& a, hipStream_t st) {
  DCF_CHECK(a.Lq >= 1 && a.Ct >= 1 && a.TE >= 1, "text_embed: empty input");
  const int Lk = a.Lq + ((a.bkgd || a.pool) ? 1 : 0);
  DCF_CHECK(a.W || a.Ct == a.TE, "text_embed: no embedding weight needs in_dim == embd_dim");
  hipLaunchKernelGGL(k_text_embed, dim3(Lk), dim3(256), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_im2col5s2(const float* X, int64_t ldx, const uint8_t* mask, float* col, int B, int T, int C, hipStream_t st) {
  DCF_CHECK(B > 0 && T > 0 && T % 2 == 0 && C % 4 == 0 && ldx % 4 == 0, "launch_im2col5s2: T must be even, C and the row pitch multiples of 4");
  const int64_t n = (int64_t)B * (T / 2) * 5 * (C / 4);
  ProfScope prof("im2col5s2", st, 0.0, 4.0 * (double)B * T * C * 3.5);
  hipLaunchKernelGGL(k_im2col5s2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, ldx, mask, col, B, T, C);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_dwconv3(const float* X, int64_t ldx, const uint8_t* mask, const float* w, float* Y, int64_t ldy, int B, int T, int stride,
                   int C, hipStream_t st) {
  DCF_CHECK(B > 0 && T > 0 && (stride == 1 || stride == 2) && T % stride == 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0,
            "launch_dwconv3: stride 1 or 2 dividing T, C and the row pitches multiples of 4");
  const int64_t n = (int64_t)B * (T / stride) * (C / 4);
  ProfScope prof("dwconv3", st, 6.0 * (double)B * (T / stride) * C, 4.0 * (double)B * T * C * (1.0 + 1.0 / stride));
  hipLaunchKernelGGL(k_dwconv3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, ldx, mask, w, Y, ldy, B, T, stride, C);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_mask_rows(float* X, const uint8_t* mask, int rows, int C, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_mask_rows, dim3((rows * C + 255) / 256), dim3(256), 0, st, X, mask, rows, C);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_rows_to_chanmajor(const float* X, float* out, int rows, int C, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(k_rows_to_chanmajor, dim3((rows * C + 255) / 256), dim3(256), 0, st, X, out, rows, C);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_text_ln(const TextLnArgs& a, int B, hipStream_t st) {
  if (B <= 0 || a.Lkmax <= 0) return 0;
  hipLaunchKernelGGL(k_text_ln, dim3(a.Lkmax, B), dim3(64), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
