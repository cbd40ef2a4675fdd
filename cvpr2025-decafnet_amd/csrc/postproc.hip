// Proposal decoding and 1-D (soft-)NMS on the GPU.
//
//   collect : Evaluator._collect_segments (libs/worker_v2.py:1131-1187)
//   nms     : nms_1d_cpu      (libs/nms/src/nms_cpu.cpp:20-63)
//   softnms : softnms_1d_cpu  (libs/nms/src/nms_cpu.cpp:72-172)
//   voting  : segment_voting  (libs/nms/nms.py:64-103)
//
// n <= a few thousand candidates per query, so each problem is latency bound; one 1024-thread
// workgroup per query keeps everything (keys, segments, alive flags) in LDS and replaces the
// reference's O(n^2) scalar loops by n short parallel steps.  Index-producing arithmetic (IoU,
// thresholds, decay) uses exactly the reference's fp32 operation order so that the returned int64
// indices are bit-identical; ties in the sorts are broken by the lower original index.
#include <type_traits>
#include "common.h"
#include "postproc.h"

namespace dcf {

constexpr int NT = 1024;          // threads per workgroup
constexpr int NW = NT / 64;

// order-preserving map float -> uint32 (larger float => larger key), total order incl. negatives
__device__ __forceinline__ uint32_t fkey(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// exclusive prefix sum of one int per thread over the workgroup; also returns the total.  Inside a wave: six DPP adds
// (row shifts 1, 2, 4, 8, then lane 15 / lane 31 broadcast into the following rows); across waves every thread adds up the
// wave totals itself.  (The first version -- six ds_bpermute steps and thread 0 walking the 16 wave totals through LDS --
// cost ~3 k cycles per call with 1023 threads waiting, ten calls per decoded video.)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_zero_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int NWV = NW>
__device__ __forceinline__ int block_exscan(int v, int* s_wave /* [NWV+1] */, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int inc = v;
  inc += dpp_zero_i<0x111>(inc);                     // row_shr:1
  inc += dpp_zero_i<0x112>(inc);                     // row_shr:2
  inc += dpp_zero_i<0x114>(inc);                     // row_shr:4
  inc += dpp_zero_i<0x118>(inc);                     // row_shr:8  -> inclusive within each 16-lane row
  inc += dpp_zero_i<DPP_BCAST15, 0xA>(inc);          // rows 1, 3 += total of rows 0, 2
  inc += dpp_zero_i<DPP_BCAST31, 0xC>(inc);          // rows 2, 3 += total of rows 0 + 1
  __syncthreads();                                   // the previous use of s_wave has been read
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NWV; ++i) { const int t = s_wave[i]; tot += t; before += i < w ? t : 0; }
  total = tot;
  return before + inc - v;
}

// the same for a 0/1 flag per thread: ballot + population counts, no cross-lane shuffles, every thread adds up the wave totals
__device__ __forceinline__ int block_excount(bool f, int* s_wave /* [NW+1] */, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long m = __ballot(f);
  const int within = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();                                  // the previous use of s_wave has been read
  if (lane == 0) s_wave[w] = __popcll(m);
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) { const int t = s_wave[i]; tot += t; before += i < w ? t : 0; }
  total = tot;
  return before + within;
}

// Bitonic sort, descending, of the n_pad (power of two, <= 4096) 64-bit keys in key[] (entries beyond the live ones hold 0).
// The keys live in REGISTERS: wave w owns the block [w * EW, (w + 1) * EW), EW = 64 * EPL, lane l holds keys w * EW + 64 m + l.
// A compare-exchange step with distance j is then
//   j <  64 : an exchange between lanes l and l ^ j -- DPP quad permutes / row mirrors, v_permlane16/32_swap: VALU speed,
//   j <  EW : between two registers of the same lane,
//   j >= EW : between waves, through key[] and two workgroup barriers
// -- 10 barrier steps out of 66 for 2048 keys.  (All steps through LDS with a barrier each took 30 us of the 112 us proposal
// decoding of a 32 640-point video; wave-local steps through LDS without barriers were no faster: 39 us.)
template <int J>
__device__ __forceinline__ uint32_t xor_lane(uint32_t x) {
  const int v = (int)x;
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(v, v, DPP_XOR1, 0xf, 0xf, false);
  if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(v, v, DPP_XOR2, 0xf, 0xf, false);
  if constexpr (J == 4) {                               // l ^ 4 = reverse the quad of (reverse the half row)
    const int t = __builtin_amdgcn_update_dpp(v, v, DPP_HALF_MIRROR, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, t, 0x1B, 0xf, 0xf, false);       // quad_perm [3,2,1,0]
  }
  if constexpr (J == 8) {                               // l ^ 8 = reverse the half row of (reverse the row)
    const int t = __builtin_amdgcn_update_dpp(v, v, DPP_MIRROR, 0xf, 0xf, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, t, DPP_HALF_MIRROR, 0xf, 0xf, false);
  }
  if constexpr (J == 16) {
    auto t = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // {r0,r0,r2,r2} / {r1,r1,r3,r3}
    return ((threadIdx.x >> 4) & 1) ? t[0] : t[1];
  }
  if constexpr (J == 32) {
    auto t = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // {lo,lo} / {hi,hi}
    return ((threadIdx.x >> 5) & 1) ? t[0] : t[1];
  }
  return x;
}
template <int J>
__device__ __forceinline__ unsigned long long xor_lane64(unsigned long long x) {
  return ((unsigned long long)xor_lane<J>((uint32_t)(x >> 32)) << 32) | xor_lane<J>((uint32_t)x);
}
// element i keeps the larger of (its key, its partner's key) if it is the lower index of a descending pair or the higher
// index of an ascending one
__device__ __forceinline__ unsigned long long cmpx_keep(unsigned long long v, unsigned long long pv, int i, int j, int k) {
  const bool take_max = ((i & j) == 0) == ((i & k) == 0);
  const bool gt = v > pv;
  return (take_max == gt) ? v : pv;
}

template <int EPL>
__device__ __forceinline__ void bitonic_desc_regs(unsigned long long* key, int n_pad) {
  constexpr int EW = 64 * EPL;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n_eff = n_pad < 64 ? 64 : n_pad;            // a single wave sorts 64 slots, the padding slots hold 0
  const bool active = w * EW < n_eff;
  unsigned long long v[EPL];
  int idx[EPL];
  __syncthreads();                                      // key[] is complete
#pragma unroll
  for (int m = 0; m < EPL; ++m) {
    idx[m] = w * EW + m * 64 + lane;
    v[m] = (active && idx[m] < n_pad) ? key[idx[m]] : 0ull;
  }
  for (int k = 2; k <= n_eff; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= EW) {
        __syncthreads();                                // the readers of the previous exchange are done
        if (active) {
#pragma unroll
          for (int m = 0; m < EPL; ++m) key[idx[m]] = v[m];
        }
        __syncthreads();
        if (active) {
#pragma unroll
          for (int m = 0; m < EPL; ++m) v[m] = cmpx_keep(v[m], key[idx[m] ^ j], idx[m], j, k);
        }
      } else if (j >= 64) {
        if constexpr (EPL > 1) {
          auto pair = [&](auto lo_c, auto hi_c) __attribute__((always_inline)) {   // register indices known at compile time
            constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
            const unsigned long long a = v[LO], b = v[HI];
            v[LO] = cmpx_keep(a, b, idx[LO], j, k);
            v[HI] = cmpx_keep(b, a, idx[HI], j, k);
          };
          using std::integral_constant;
          if (j == 64) {
            pair(integral_constant<int, 0>{}, integral_constant<int, 1>{});
            if constexpr (EPL == 4) pair(integral_constant<int, 2>{}, integral_constant<int, 3>{});
          } else {                                      // j == 128, EPL == 4
            if constexpr (EPL == 4) {
              pair(integral_constant<int, 0>{}, integral_constant<int, 2>{});
              pair(integral_constant<int, 1>{}, integral_constant<int, 3>{});
            }
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < EPL; ++m) {
          unsigned long long pv;
          switch (j) {
            case 1: pv = xor_lane64<1>(v[m]); break;
            case 2: pv = xor_lane64<2>(v[m]); break;
            case 4: pv = xor_lane64<4>(v[m]); break;
            case 8: pv = xor_lane64<8>(v[m]); break;
            case 16: pv = xor_lane64<16>(v[m]); break;
            default: pv = xor_lane64<32>(v[m]); break;
          }
          v[m] = cmpx_keep(v[m], pv, idx[m], j, k);
        }
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int m = 0; m < EPL; ++m)
      if (idx[m] < n_pad) key[idx[m]] = v[m];
  }
  __syncthreads();
}
__device__ __forceinline__ void bitonic_desc(unsigned long long* key, int n_pad) {
  if (n_pad <= 64 * NW) bitonic_desc_regs<1>(key, n_pad);
  else if (n_pad <= 128 * NW) bitonic_desc_regs<2>(key, n_pad);
  else bitonic_desc_regs<4>(key, n_pad);
}

// Stable LSD radix sort of n <= RS_MAX 64-bit keys by their upper 32 bits, DESCENDING, in four 8-bit passes (a bitonic
// network spends n log^2 n / 2 compare-exchanges: 64 k cycles for 2048 keys on one CU; this is ~10x less work).  Element
// e = m * NT + tid (m < RS_M).  Per pass: the lanes of a wave that share a digit find each other with eight ballots
// (rank = population count of the peers in lower lanes, the first peer records the group size in table[digit][m][wave]);
// one workgroup scan of the table in (digit descending, m, wave) order turns the sizes into output offsets; scatter.
// The input order is kept among equal digits, so keys that arrive in candidate-index order leave ordered by (score
// descending, index ascending).  key / key2 ping-pong; the result is in key.
constexpr int RS_M = 2;
constexpr int RS_MAX = RS_M * NT;
constexpr int RS_TABLE = 256 * RS_M * NW;              // 8192 counters

__device__ __forceinline__ void radix_sort_desc(unsigned long long* key, unsigned long long* key2, int* table, int n, int* s_wave) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  constexpr int EPT = RS_TABLE / NT;                   // table entries per thread in the scan (8)
  unsigned long long* src = key;
  unsigned long long* dst = key2;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 32 + 8 * pass;
    __syncthreads();                                   // src is complete, the table is free
#pragma unroll
    for (int z = 0; z < EPT; ++z) table[tid * EPT + z] = 0;
    __syncthreads();
    unsigned long long kv[RS_M];
    int slot[RS_M], rank[RS_M];
#pragma unroll
    for (int m = 0; m < RS_M; ++m) {
      const int e = m * NT + tid;
      const bool valid = e < n;
      kv[m] = valid ? src[e] : 0ull;
      const int d = (int)((kv[m] >> shift) & 0xffull);
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1;
        const unsigned long long bb = __ballot(bit);
        peers &= bit ? bb : ~bb;
      }
      rank[m] = __popcll(peers & below);
      slot[m] = ((255 - d) * RS_M + m) * NW + w;
      if (valid && rank[m] == 0) table[slot[m]] = __popcll(peers);
      if (!valid) slot[m] = -1;
    }
    __syncthreads();
    int loc[EPT], sum = 0;
#pragma unroll
    for (int z = 0; z < EPT; ++z) { loc[z] = sum; sum += table[tid * EPT + z]; }
    int tot;
    const int base = block_exscan(sum, s_wave, tot);
    __syncthreads();
#pragma unroll
    for (int z = 0; z < EPT; ++z) table[tid * EPT + z] = base + loc[z];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < RS_M; ++m)
      if (slot[m] >= 0) dst[table[slot[m]] + rank[m]] = kv[m];
    unsigned long long* t = src; src = dst; dst = t;
  }
  __syncthreads();
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// ------------------------------------------------------------------------------------------
// collect segments
// ------------------------------------------------------------------------------------------
constexpr int CAND_CAP = 4096;    // >= pre_nms_topk

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// CACHE: S <= KPT * NT points: a thread keeps its KPT keys (points tid, tid + NT, ...: coalesced loads) in registers for the
// three selection passes and the compaction: no re-reads from global memory, and the ordered compaction takes its offsets
// from ONE scan of the (chunk, wave) count table per class instead of two workgroup scans per 1024-point chunk (41 us of
// the 112 us this kernel took at 32 640 points)
constexpr int KPT = 32;
template <bool CACHE>
__global__ __launch_bounds__(NT) void k_collect(CollectArgs p) {
  __shared__ unsigned long long key[CAND_CAP];
  __shared__ unsigned long long key2[RS_MAX];
  __shared__ int hist[RS_TABLE];                       // radix-select histogram (2048), compaction tables, sort table
  __shared__ int s_wave[NW + 1];
  __shared__ uint32_t s_prefix;
  __shared__ int s_need;
  const int q = blockIdx.x, tid = threadIdx.x;
  const int S = p.S;
  const float* logits = p.logits + (size_t)q * S;
  const float* offs = p.offsets + (size_t)q * S * 2;
  const uint8_t* mask = p.masks + (size_t)q * S;
  uint32_t* skey = p.keys + (size_t)q * S;

  // (1) score = sigmoid(logit) [* ext] * mask, keep > thresh
  // ext_scores are max-pooled (k3, s2, p1, -inf padding) once per level (worker_v2.py:1150-1156); a maximum of maxima is
  // the maximum over the union of the windows, which for point j of level l is the level-0 range
  // [j 2^l - (2^l - 1), j 2^l + (2^l - 1)] clipped to the video: read straight from the level-0 row, no pyramid buffer.
  const float* ext = p.ext ? p.ext + (size_t)q * p.T : nullptr;
  int cnt = 0;
  uint32_t kreg[CACHE ? KPT : 1];
#pragma unroll
  for (int c = 0; c < (CACHE ? KPT : 1); ++c) kreg[c] = 0u;
  auto score = [&](int i) __attribute__((always_inline)) -> uint32_t {
    float s = sigmoidf_(logits[i]);
    if (ext) {
      int l = 0;
      while (l + 1 < p.n_levels && i >= p.off[l + 1]) ++l;
      const int c = (i - p.off[l]) << l, h = (1 << l) - 1;
      const int lo = max(c - h, 0), hi = min(c + h, p.T - 1);
      float e = ext[lo];
      for (int t = lo + 1; t <= hi; ++t) e = fmaxf(e, ext[t]);
      s *= e;
    }
    s *= (mask[i] ? 1.f : 0.f);
    return (s > p.pre_nms_thresh) ? __float_as_uint(s) : 0u;   // positive floats order as uints
  };
  if constexpr (CACHE) {
#pragma unroll
    for (int c = 0; c < KPT; ++c) {
      const int i = c * NT + tid;
      if (i < S) { kreg[c] = score(i); cnt += kreg[c] != 0; }
    }
  } else {
    for (int i = tid; i < S; i += NT) {
      const uint32_t k = score(i);
      skey[i] = k;
      cnt += k != 0;
    }
  }
  int n_cand;
  block_exscan(cnt, s_wave, n_cand);
  const int K = min(min(n_cand, p.pre_nms_topk), CAND_CAP);

  // (2) radix-select the K-th largest key: 11 + 11 + 10 bits
  uint32_t prefix = 0, pmask = 0;
  int need = K;                                   // how many still to take from the current bucket
  if (K > 0) {
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      const int nb = 1 << bits[pass];
      for (int i = tid; i < nb; i += NT) hist[i] = 0;
      __syncthreads();
      if constexpr (CACHE) {
#pragma unroll
        for (int c = 0; c < KPT; ++c) {
          const uint32_t k = kreg[c];
          if (k && (k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1);
        }
      } else {
        for (int i = tid; i < S; i += NT) {
          uint32_t k = skey[i];
          if (k && (k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1);
        }
      }
      __syncthreads();
      // the bucket holding the need-th largest key: the highest b with sum_{j >= b} hist[j] >= need.  Thread t owns buckets
      // 2t, 2t + 1 (the serial walk down the 2048 buckets by one thread was 60 us per pass)
      {
        const int b1 = 2 * tid + 1, b0 = 2 * tid;
        const int h1 = b1 < nb ? hist[b1] : 0, h0 = b0 < nb ? hist[b0] : 0;
        int tot;
        const int below = block_exscan(h0 + h1, s_wave, tot);        // keys in buckets < 2t
        const int above1 = tot - below - h0 - h1;                    // keys in buckets > 2t + 1
        const int above0 = above1 + h1;
        if (b1 < nb && above1 < need && above1 + h1 >= need) { s_prefix = prefix | ((uint32_t)b1 << shifts[pass]); s_need = need - above1; }
        if (b0 < nb && above0 < need && above0 + h0 >= need) { s_prefix = prefix | ((uint32_t)b0 << shifts[pass]); s_need = need - above0; }
      }
      __syncthreads();
      prefix = s_prefix;
      need = s_need;
      pmask |= (uint32_t)(nb - 1) << shifts[pass];
      __syncthreads();
    }
  }
  const uint32_t kth = prefix;                    // exact K-th largest key; `need` ties are taken in index order

  // (3) ordered compaction of {key > kth} (there are K - need of them) plus the first `need` {key == kth}
  const int n_greater = K - need;
  int run_g = 0, run_e = 0;
  auto place = [&](int i, uint32_t k) __attribute__((always_inline)) {
    const int g = k > kth, e = (k == kth && k != 0);
    int tg, te;
    const int pg = block_excount(g != 0, s_wave, tg);
    const int pe = block_excount(e != 0, s_wave, te);
    if (g) key[run_g + pg] = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    if (e && run_e + pe < need) key[n_greater + run_e + pe] = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    run_g += tg;
    run_e += te;
  };
  if (K > 0) {
    if constexpr (CACHE) {
      // counts per (chunk c, wave w) in point order c * NW + w, one table per class (the histogram array is free now)
      int* tab_g = hist;
      int* tab_e = hist + KPT * NW;
      const int lane = tid & 63, w = tid >> 6;
      const unsigned long long below = (1ull << lane) - 1ull;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < KPT; ++c) {
        const unsigned long long mg = __ballot(kreg[c] > kth), me = __ballot(kreg[c] == kth && kreg[c] != 0);
        if (lane == 0) { tab_g[c * NW + w] = __popcll(mg); tab_e[c * NW + w] = __popcll(me); }
      }
      __syncthreads();
      int tg, te;
      const int sg = block_exscan(tid < KPT * NW ? tab_g[tid] : 0, s_wave, tg);
      const int se = block_exscan(tid < KPT * NW ? tab_e[tid] : 0, s_wave, te);
      __syncthreads();
      if (tid < KPT * NW) { tab_g[tid] = sg; tab_e[tid] = se; }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < KPT; ++c) {
        const uint32_t k = kreg[c];
        const bool g = k > kth, e = k == kth && k != 0;
        const unsigned long long mg = __ballot(g), me = __ballot(e);
        const unsigned long long v = ((unsigned long long)k << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)(c * NT + tid));
        if (g) key[tab_g[c * NW + w] + __popcll(mg & below)] = v;
        if (e) { const int pe = tab_e[c * NW + w] + __popcll(me & below); if (pe < need) key[n_greater + pe] = v; }
      }
    } else {
      for (int c0 = 0; c0 < S; c0 += NT) place(c0 + tid, c0 + tid < S ? skey[c0 + tid] : 0u);
    }
  }
  __syncthreads();
  // (4) sort: score descending, ties by lower candidate index (stable argsort); the candidates are in index order here
  if (K <= RS_MAX) {
    radix_sort_desc(key, key2, hist, K, s_wave);
  } else {
    const int n_pad = next_pow2(max(K, 1));
    for (int i = K + tid; i < n_pad; i += NT) key[i] = 0ull;
    bitonic_desc(key, n_pad);
  }

  // (5) decode + length filter, order preserved
  float* out_segs = p.segs + (size_t)q * p.pre_nms_topk * 2;
  float* out_scores = p.scores + (size_t)q * p.pre_nms_topk;
  int run = 0;
  for (int c0 = 0; c0 < K; c0 += NT) {
    const int j = c0 + tid;
    int keep = 0;
    float left = 0.f, right = 0.f, sc = 0.f;
    if (j < K) {
      const unsigned long long kk = key[j];
      const int i = (int)(0xFFFFFFFFu - (uint32_t)(kk & 0xFFFFFFFFu));
      sc = __uint_as_float((uint32_t)(kk >> 32));
      int l = 0;
      while (l + 1 < p.n_levels && i >= p.off[l + 1]) ++l;
      const float stride = (float)(1 << l);
      const float ctr = (float)(i - p.off[l]) * stride;         // PtGenerator point (model.py:703-723)
      left = ctr - offs[2 * i] * stride;
      right = ctr + offs[2 * i + 1] * stride;
      keep = (right - left) > p.seg_len_thresh;
    }
    int tot;
    const int pos = block_excount(keep != 0, s_wave, tot);
    if (keep) {
      out_segs[2 * (run + pos)] = left;
      out_segs[2 * (run + pos) + 1] = right;
      out_scores[run + pos] = sc;
    }
    run += tot;
  }
  if (tid == 0) p.counts[q] = run;
}

bool collect_needs_scratch(int S) { return S > KPT * NT; }

int launch_collect(const CollectArgs& a, int nq, hipStream_t st) {
  if (nq <= 0) return 0;
  DCF_CHECK(a.pre_nms_topk >= 1 && a.pre_nms_topk <= CAND_CAP, "collect: pre_nms_topk=%d exceeds %d", a.pre_nms_topk, CAND_CAP);
  DCF_CHECK(a.n_levels >= 1 && a.n_levels <= 16, "collect: bad n_levels");
  ProfScope prof("collect_segments", st, 0.0, 4.0 * 4.0 * nq * a.S);
  if (!collect_needs_scratch(a.S)) hipLaunchKernelGGL(k_collect<true>, dim3(nq), dim3(NT), 0, st, a);
  else hipLaunchKernelGGL(k_collect<false>, dim3(nq), dim3(NT), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// hard NMS
// ------------------------------------------------------------------------------------------
constexpr int NMS_BIG_MAX = 1 << 22;  // 32-bit positions, a few hundred MB of scratch at most
constexpr int NMS_CAP = 4096;     // candidates per query that fit one workgroup's LDS; beyond that the BIG instantiations

// plain bitonic network over keys in global memory (n_pad a power of two, any size): one compare-exchange per element
// pair and step, a workgroup barrier per step.  Only the n > NMS_CAP kernels use it.
__device__ __forceinline__ void bitonic_desc_global(unsigned long long* key, int n_pad) {
  __syncthreads();
  for (int k = 2; k <= n_pad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pad; i += NT) {
        const int o = i ^ j;
        if (o > i) {
          const unsigned long long a = key[i], b = key[o];
          const bool desc = (i & k) == 0;
          if (desc ? a < b : a > b) { key[i] = b; key[o] = a; }
        }
      }
      __syncthreads();
    }
}

// per-query scratch of the BIG kernels (bytes, 16-byte aligned pieces)
static inline size_t nms_big_scratch(int n) {
  size_t np = 1;
  while (np < (size_t)n) np <<= 1;
  return np * 8 + (size_t)n * 12 + (((size_t)n + 15) & ~(size_t)15);
}
static inline size_t softnms_big_scratch(int n) { return (size_t)n * 24; }

// BIG = false: every array in LDS (n <= NMS_CAP).  BIG = true: the same algorithm with the arrays in a global scratch
// block (n > NMS_CAP: the reference takes any n, nms_cpu.cpp:20-63); one workgroup per query either way, a workgroup
// barrier orders its global writes for its own later reads.
template <bool BIG>
__global__ __launch_bounds__(NT) void k_nms(NmsArgs p, unsigned char* scratch, size_t scratch_per_q) {
  __shared__ unsigned long long s_key[BIG ? 1 : NMS_CAP];
  __shared__ float s_x1[BIG ? 1 : NMS_CAP], s_x2[BIG ? 1 : NMS_CAP], s_ar[BIG ? 1 : NMS_CAP];
  __shared__ unsigned char s_alive[BIG ? 1 : NMS_CAP];
  __shared__ int s_wave[NW + 1];
  const int q = blockIdx.x, tid = threadIdx.x;
  const int n = p.counts ? min(p.counts[q], p.n_max) : p.n_max;
  const float* segs = p.segs + (size_t)q * p.stride * 2;
  const float* scores = p.scores + (size_t)q * p.stride;
  long long* out = p.keep + (size_t)q * p.stride;
  if (n <= 0) { if (tid == 0) p.keep_counts[q] = 0; return; }
  const int n_pad = next_pow2(n);
  unsigned long long* key;
  float *x1, *x2, *ar;
  unsigned char* alive;
  if constexpr (BIG) {
    unsigned char* base = scratch + (size_t)q * scratch_per_q;
    const int n_cap = p.n_max;
    int np_cap = 1;
    while (np_cap < n_cap) np_cap <<= 1;
    key = reinterpret_cast<unsigned long long*>(base);
    x1 = reinterpret_cast<float*>(base + (size_t)np_cap * 8);
    x2 = x1 + n_cap; ar = x2 + n_cap;
    alive = reinterpret_cast<unsigned char*>(ar + n_cap);
  } else {
    key = s_key; x1 = s_x1; x2 = s_x2; ar = s_ar; alive = s_alive;
  }
  for (int i = tid; i < n_pad; i += NT)
    key[i] = i < n ? (((unsigned long long)fkey(scores[i]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i)) : 0ull;
  if constexpr (BIG) bitonic_desc_global(key, n_pad);
  else bitonic_desc(key, n_pad);
  for (int a = tid; a < n; a += NT) {
    const int i = (int)(0xFFFFFFFFu - (uint32_t)(key[a] & 0xFFFFFFFFu));
    const float l = segs[2 * i], r = segs[2 * i + 1];
    x1[a] = l; x2[a] = r;
    ar[a] = (r - l) + 1e-6f;                         // areas = x2 - x1 + 1e-6  (nms_cpu.cpp:31)
    alive[a] = 1;
  }
  __syncthreads();
  // Greedy suppression in blocks of 64 candidates (sorted order).  The kept set of a greedy NMS does not depend on the
  // order in which suppressions are applied, only on "i kept => every later j with IoU(i, j) >= thr dies":
  //   1. all threads build the 64 x 64 suppression bit matrix of the block (four IoUs each), wave 0 resolves the block
  //      with a scalar scan over readlane'd rows (no barrier per candidate: the one-candidate-per-barrier loop this
  //      replaces took 0.87 ms at n = 2000);
  //   2. every thread tests the later candidates against the (<= 64) kept members of the block.
  // Four barriers per 64 candidates.  IoU arithmetic as nms_cpu.cpp:38-56 (same fp32 expressions).
  __shared__ int s_kept[64];
  __shared__ int s_nkept;
  __shared__ unsigned s_row[64][2];
  const bool nonpos_thr = !(p.iou_thresh > 0.f);       // then 0 >= thr holds and disjoint pairs suppress too
  for (int a0 = 0; a0 < n; a0 += 64) {
    // 1a. the 64 x 64 bit matrix with all threads: thread -> (row i, four columns j)
    if (tid < 128) s_row[tid >> 1][tid & 1] = 0u;
    __syncthreads();
    {
      const int i = tid >> 4, j0 = (tid & 15) * 4;
      const int gi = a0 + i;
      if (gi < n) {
        const float ix1 = x1[gi], ix2 = x2[gi], ia = ar[gi];
        unsigned bits = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q, gj = a0 + j;
          if (j > i && gj < n) {
            const float xx1 = fmaxf(ix1, x1[gj]);
            const float xx2 = fminf(ix2, x2[gj]);
            const float inter = fmaxf(0.f, xx2 - xx1);
            if (inter > 0.f || nonpos_thr) {             // disjoint pairs: ovr = 0 < thr, no division needed
              const float ovr = inter / (ia + ar[gj] - inter);
              if (ovr >= p.iou_thresh) bits |= 1u << (j & 31);
            }
          }
        }
        if (bits) atomicOr(&s_row[i][j0 >> 5], bits);
      }
    }
    __syncthreads();
    // 1b. wave 0 resolves the block
    if (tid < 64) {
      const int i = a0 + tid;
      const bool valid = i < n;
      const bool al = valid && alive[i];
      const unsigned rlo = s_row[tid][0], rhi = s_row[tid][1];
      const unsigned long long alive_mask = __ballot(al);
      unsigned long long removed = ~alive_mask, keepmask = 0ull;
#pragma unroll
      for (int k = 0; k < 64; ++k) {                   // scalar scan: rows come from lane k
        const unsigned long long rk = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)rhi, k) << 32) |
                                      (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)rlo, k);   // readlane returns int: no sign extension
        if (!((removed >> k) & 1ull)) { keepmask |= 1ull << k; removed |= rk; }
      }
      const bool kept = (keepmask >> tid) & 1ull;
      if (valid) alive[i] = kept ? 1 : 0;
      if (kept) s_kept[__popcll(keepmask & ((1ull << tid) - 1ull))] = i;
      if (tid == 0) s_nkept = __popcll(keepmask);
    }
    __syncthreads();
    const int nk = s_nkept;
    if (nk > 0) {
      for (int b = a0 + 64 + tid; b < n; b += NT) {
        if (!alive[b]) continue;
        const float bx1 = x1[b], bx2 = x2[b], ba = ar[b];
        for (int k = 0; k < nk; ++k) {
          const int a = s_kept[k];
          const float xx1 = fmaxf(x1[a], bx1);
          const float xx2 = fminf(x2[a], bx2);
          const float inter = fmaxf(0.f, xx2 - xx1);
          if (inter > 0.f || nonpos_thr) {
            const float ovr = inter / (ar[a] + ba - inter);
            if (ovr >= p.iou_thresh) { alive[b] = 0; break; }
          }
        }
      }
    }
    __syncthreads();
  }
  int run = 0;
  for (int c0 = 0; c0 < n; c0 += NT) {
    const int a = c0 + tid;
    const int k = a < n ? alive[a] : 0;
    int tot;
    const int pos = block_exscan(k, s_wave, tot);
    if (k) out[run + pos] = (long long)(0xFFFFFFFFu - (uint32_t)(key[a] & 0xFFFFFFFFu));
    run += tot;
  }
  if (tid == 0) p.keep_counts[q] = run;
}

int launch_nms(const NmsArgs& a, int nq, hipStream_t st) {
  if (nq <= 0) return 0;
  DCF_CHECK(a.n_max >= 0 && a.n_max <= NMS_BIG_MAX, "nms: n=%d exceeds %d", a.n_max, NMS_BIG_MAX);
  DCF_CHECK(a.stride >= a.n_max, "nms: stride < n_max");
  ProfScope prof("nms_1d", st, 0.0, 0.0);
  if (a.n_max <= NMS_CAP) {
    hipLaunchKernelGGL(k_nms<false>, dim3(nq), dim3(NT), 0, st, a, (unsigned char*)nullptr, (size_t)0);
    DCF_HIP(hipGetLastError());
    return 0;
  }
  // more candidates than one workgroup's LDS holds: stream-ordered scratch, same kernel over global arrays
  const size_t per_q = (nms_big_scratch(a.n_max) + 255) & ~(size_t)255;
  unsigned char* scratch = nullptr;
  DCF_HIP(hipMallocAsync((void**)&scratch, per_q * nq, st));
  hipLaunchKernelGGL(k_nms<true>, dim3(nq), dim3(NT), 0, st, a, scratch, per_q);
  const hipError_t e = hipGetLastError();
  DCF_HIP(hipFreeAsync(scratch, st));
  DCF_HIP(e);
  return 0;
}

// ------------------------------------------------------------------------------------------
// soft NMS
// ------------------------------------------------------------------------------------------
// expf exactly as the reference's C library computes it (softnms_1d_cpu's weight is std::exp of a float, nms_cpu.cpp:147): glibc 2.27+ /
// ARM optimized-routines `expf` -- double arithmetic, z = x N / ln 2, k = round(z) through the 1.5 x 2^52 shift, r = z - k,
// s = 2^(k / N) from a 32-entry table of bit patterns, a cubic in r, one rounding to float at the end.  Restated here operation for operation
// (the three multiply-adds fused, as the x86-64 build the fixtures come from selects on FMA hardware; with or without fusion the float
// result is the same on 200 000 random arguments in (-40, 0], checked against libm in the build container: tools/expf_model_check.py).  The
// double-precision exp rounded to float that stood here is the CORRECTLY rounded value, which glibc's expf (< 0.502 ulp) misses in ~1 of
// 2 000 arguments: 65 of 71 742 dets elements of the known-answer fixtures were 1 - 6 ulp off (a score is decayed several times); with
// this function they are bit-identical (tests/test_gpu_ops.py: torch.equal on the dets since round 5).
__device__ const unsigned long long expf_tab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
template <typename Tab>
__device__ __forceinline__ float expf_glibc(float x, Tab tab) {      // tab: expf_tab, or a copy of it nearer by
  const double xd = (double)x;
  const unsigned abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
  if (abstop >= 0x42bu) return (float)exp(xd);          // |x| >= 88, inf, NaN: the library's special cases (not reached by -ovr^2 / sigma for sigma > 0.012)
  const double inv_ln2_n = 0x1.71547652b82fep+0 * 32.0, shift = 0x1.8p+52;
  const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
  double z = inv_ln2_n * xd;
  double kd = z + shift;
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= shift;
  const double r = z - kd;
  unsigned long long t = tab[ki & 31ull];
  t += ki << 47;
  const double s = __longlong_as_double((long long)t);
  z = __builtin_fma(c0, r, c1);
  const double r2 = r * r;
  double y = __builtin_fma(c2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

// The arg-max of soft NMS as ONE unsigned 64-bit maximum: high word = the score's bits mapped so that unsigned order is float order
// (-0 counted as +0, as the reference's `>` does), low word = ~position, so the larger key is the larger score and, between equal scores,
// the LOWER position -- "first maximum wins" (nms_cpu.cpp:107-113).  0 = no candidate (below the key of any score that is not a NaN).
// Branch-free: the struct-and-`if` form of this compiled to a dozen exec-mask branches per position (profiles/r06_notes.md section 5).
__device__ __forceinline__ unsigned long long snms_key(float v, int pos) {
  unsigned b = __float_as_uint(v);
  b = v == 0.f ? 0u : b;
  const unsigned srt = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
  return ((unsigned long long)srt << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)pos);
}
__device__ __forceinline__ int snms_pos(unsigned long long key) { return (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)); }
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k) {         // lanes not written keep their own value
  const int lo = (int)(unsigned)(k & 0xFFFFFFFFull), hi = (int)(unsigned)(k >> 32);
  const unsigned l2 = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const unsigned h2 = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)h2 << 32) | l2;
}
// the wave's largest key in every lane: the DPP steps of wave_max (common.h)
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  k = umax64(k, dpp_u64<DPP_XOR1>(k));
  k = umax64(k, dpp_u64<DPP_XOR2>(k));
  k = umax64(k, dpp_u64<DPP_HALF_MIRROR>(k));
  k = umax64(k, dpp_u64<DPP_MIRROR>(k));
  k = umax64(k, dpp_u64<DPP_BCAST15, 0xA>(k));
  k = umax64(k, dpp_u64<DPP_BCAST31, 0xC>(k));
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k & 0xFFFFFFFFull), 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// SNT threads: a pick is two barriers, a handful of LDS round trips and one pass over the positions behind it -- four waves synchronise
// faster than sixteen, sixteen pass over 4 096 positions faster than four: launch_softnms picks 256 / 512 / 1024 by the candidate count
// (tools/softnms_time.py, same box: n = 512 0.69 / 0.73 / 0.73 ms, n = 2 000 3.39 / 3.00 / 3.05, n = 4 096 8.07 / 6.18 / 5.89).
template <bool BIG, int SNT>
__global__ __launch_bounds__(SNT) void k_softnms(SoftNmsArgs p, unsigned char* scratch, size_t scratch_per_q) {
  constexpr int SNW = SNT / 64;
  __shared__ float s_x1[BIG ? 1 : NMS_CAP], s_x2[BIG ? 1 : NMS_CAP], s_sc[BIG ? 1 : NMS_CAP], s_ar[BIG ? 1 : NMS_CAP];
  __shared__ int s_ind[BIG ? 1 : NMS_CAP];
  __shared__ int s_slot[BIG ? 1 : NMS_CAP];
  __shared__ int s_wave[SNW + 1];
  __shared__ unsigned long long s_key[SNW];              // the waves' best (score, position) keys for the coming pick
  __shared__ unsigned long long s_tab[32];               // expf_tab in LDS: the look-up sits in the dependent chain of every decay
  __shared__ int s_dead2[2], s_cnt2[2];                  // per pick parity: "a score fell below min_score", length of the overlap list
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = p.counts ? min(p.counts[q], p.n_max) : p.n_max;
  const float* segs = p.segs + (size_t)q * p.stride * 2;
  const float* scores = p.scores + (size_t)q * p.stride;
  float* dets = p.dets + (size_t)q * p.stride * 3;
  long long* out = p.inds + (size_t)q * p.stride;
  if (n <= 0) { if (tid == 0) p.out_counts[q] = 0; return; }
  float *x1, *x2, *sc, *ar;
  int *ind, *slot;
  if constexpr (BIG) {
    float* base = reinterpret_cast<float*>(scratch + (size_t)q * scratch_per_q);
    const int n_cap = p.n_max;
    x1 = base; x2 = x1 + n_cap; sc = x2 + n_cap; ar = sc + n_cap;
    ind = reinterpret_cast<int*>(ar + n_cap); slot = ind + n_cap;
  } else {
    x1 = s_x1; x2 = s_x2; sc = s_sc; ar = s_ar; ind = s_ind; slot = s_slot;
  }
  if (tid < 32) s_tab[tid] = expf_tab[tid];
  for (int i = tid; i < n; i += SNT) {
    const float l = segs[2 * i], r = segs[2 * i + 1];
    x1[i] = l; x2[i] = r; sc[i] = scores[i]; ar[i] = (r - l) + 1e-6f; ind[i] = i;
  }
  __syncthreads();
  int nsegs = n;
  const int iters = p.max_iters > 0 ? p.max_iters : n;
  int i = 0;
  // TWO barriers per pick (round 5: four; round 4: five with a twelve-step ds_bpermute butterfly):
  //   [A] the waves' best (score, position) keys over [i, nsegs) are in LDS -- every thread folded them together while it DECAYED the scores
  //       for the previous pick (each score is looked at then anyway), so a pick needs no pass of its own over the scores; EVERY thread reduces
  //       the SNW keys, so all know the pick `mp`; every thread reads the picked segment (slot mp) and the one it changes places with (slot i);
  //       the threads then look at the segments behind i -- the thread that meets position mp works on the values of slot i it read above
  //       (the slot's content after the swap) -- and list those the pick overlaps (counter of this pick's parity);
  //   [B] every thread has read slots i and mp: thread 0 writes the swap and the detection; the listed segments are decayed; the flag of this
  //       pick's parity is raised if a score fell below min_score; the waves publish their best for the next pick.
  // The counter / flag of the OTHER parity are cleared behind [B] (their readers are past [A]).  A raised flag sends the next iteration
  // through the pruning pass first (uniformly), which moves segments: the keys are then taken again.
  // The passes over the positions are branch-free (keys, selects, one `if` around the list entry): written with an `if` per case and a
  // (score, position) struct, a pick spent 1.25 of its 3.3 us (n = 2 000, four waves) in the exec-mask branches of the scan and 0.5 us in
  // the wave reduction of the pairs (profiles/r06_notes.md section 5; several positions per trip, ballots instead of the atomic, a list
  // per wave, the segment carried in the list entry and a move-only path for a single pruned segment were each measured slower or equal).
  auto publish_best = [&](unsigned long long best) __attribute__((always_inline)) {
    best = wave_max_key(best);
    if (lane == 0) s_key[w] = best;
  };
  {
    unsigned long long best = 0;
    for (int pos = tid; pos < nsegs; pos += SNT) best = umax64(best, snms_key(sc[pos], pos));
    publish_best(best);
  }
  if (tid == 0) { s_dead2[0] = s_dead2[1] = 0; s_cnt2[0] = s_cnt2[1] = 0; }
  int dead_cnt = 0;                                      // scores of this thread's share that fell below min_score in the last decay
  const bool all_touch = p.method != 2 && !(p.iou_thresh > 0.f);
  for (;;) {
    __syncthreads();                                                                      // [A]
    const int cur = i & 1, prv = cur ^ 1;
    if (i > 0 && s_dead2[prv]) {                         // uniform: the decay behind pick i - 1 left scores below min_score
      int n_dead;
      block_exscan<SNW>(dead_cnt, s_wave, n_dead);
      // ---- emulate the sequential "swap with the last segment" pruning (nms_cpu.cpp:157-165):
      // survivors keep their slots; the k-th dead slot (left to right) inside the new range is
      // filled by the k-th surviving segment counted from the right end of the old range.
      const int first = i;
      const int n_alive = (nsegs - first) - n_dead;
      const int new_n = first + n_alive;
      // each thread owns a contiguous run of positions so that prefix counts are ordered
      const int span = nsegs - first;
      const int per = (span + SNT - 1) / SNT;
      const int lo = first + tid * per, hi = min(lo + per, nsegs);
      int c_dead_left = 0, c_alive_right = 0;
      for (int pos = lo; pos < hi; ++pos) {
        const bool dead = sc[pos] < p.min_score;
        if (pos < new_n) c_dead_left += dead; else c_alive_right += !dead;
      }
      int tot_d, tot_a;
      int pd = block_exscan<SNW>(c_dead_left, s_wave, tot_d);
      int pa = block_exscan<SNW>(c_alive_right, s_wave, tot_a);
      for (int pos = lo; pos < hi; ++pos) {
        if (pos < new_n && sc[pos] < p.min_score) slot[pd++] = pos;
      }
      __syncthreads();
      for (int pos = lo; pos < hi; ++pos) {
        if (pos >= new_n && !(sc[pos] < p.min_score)) {
          // rank from the right = tot_a - 1 - (rank from the left)
          const int dst = slot[tot_a - 1 - pa];
          ++pa;
          x1[dst] = x1[pos]; x2[dst] = x2[pos]; sc[dst] = sc[pos]; ar[dst] = ar[pos]; ind[dst] = ind[pos];
        }
      }
      __syncthreads();
      nsegs = new_n;
      unsigned long long best = 0;                       // segments moved: the keys over [i, nsegs) again
      for (int pos = i + tid; pos < nsegs; pos += SNT) best = umax64(best, snms_key(sc[pos], pos));
      publish_best(best);
      __syncthreads();
    }
    if (!(i < nsegs && i < iters)) break;                // (uniform)
    unsigned long long bb = s_key[0];
#pragma unroll
    for (int k = 1; k < SNW; ++k) bb = umax64(bb, s_key[k]);
    const int mp = snms_pos(bb);
    // the pick (slot mp) and the segment that moves into its slot (slot i): nms_cpu.cpp:115-133
    const float ix1 = x1[mp], ix2 = x2[mp], isc = sc[mp], ia = ar[mp];
    const int iind = ind[mp];
    const float ox1 = x1[i], ox2 = x2[i], osc = sc[i], oar = ar[i];
    const int oind = ind[i];
    // ---- decay every later segment (nms_cpu.cpp:137-155).  Without overlap the weight is exactly 1 -- exp(-0) for the Gaussian, below
    // any positive threshold for the other two -- and score * 1 is the score: only the segments the pick touches need the division and
    // the double-precision exp.  They are few (tens of 2 000) but scattered, so four of five waves would run the exp for a lane or
    // two: their positions go to a list (`slot`, free outside the pruning pass) and ceil(count / 64) waves work it off behind [B].
    dead_cnt = 0;
    bool fix_mp = false;
    unsigned long long best = 0;                         // this thread's share of the NEXT pick's argmax, over [i + 1, nsegs)
    // (SCAN_U positions per trip, their LDS reads going out unconditionally from clamped positions, then the selects, then the list
    // entries: measured with 1 / 2 / 3 / 4 / 8 -- one position per trip is the fastest at every thread count, and this form of it is 7 %
    // faster than a plain `for (pos...; pos < nsegs; ...)` with the reads inside)
    constexpr int SCAN_U = 1;
    for (int base = i + 1 + tid; base < nsegs; base += SCAN_U * SNT) {
      float qx1[SCAN_U], qx2[SCAN_U], qsc[SCAN_U];
      bool touch[SCAN_U];
#pragma unroll
      for (int u = 0; u < SCAN_U; ++u) {
        const int pos = min(base + u * SNT, nsegs - 1);
        qx1[u] = x1[pos]; qx2[u] = x2[pos]; qsc[u] = sc[pos];
      }
#pragma unroll
      for (int u = 0; u < SCAN_U; ++u) {
        const int pos = base + u * SNT;
        const bool valid = pos < nsegs;
        const bool moved = pos == mp;                    // this slot now holds what slot i held
        const float px1 = moved ? ox1 : qx1[u], px2 = moved ? ox2 : qx2[u], psc = moved ? osc : qsc[u];
        const float inter = fmaxf(0.f, fminf(ix2, px2) - fmaxf(ix1, px1));
        touch[u] = valid & ((inter > 0.f) | all_touch);
        const bool keep = valid & !touch[u];             // untouched: the score stays, it competes for the next pick as it is
        fix_mp = fix_mp | (keep & moved);                // (written behind [B]: thread 0 still reads the pick's score from that slot)
        dead_cnt += (int)(keep & (psc < p.min_score));
        best = umax64(best, keep ? snms_key(psc, pos) : 0ull);
      }
#pragma unroll
      for (int u = 0; u < SCAN_U; ++u)
        if (touch[u]) slot[atomicAdd(&s_cnt2[cur], 1)] = base + u * SNT;
    }
    __syncthreads();                                                                      // [B]
    if (fix_mp) sc[mp] = osc;
    if (tid == 0) {
      dets[i * 3 + 0] = ix1; dets[i * 3 + 1] = ix2; dets[i * 3 + 2] = isc;
      x1[i] = ix1; x2[i] = ix2; sc[i] = isc; ar[i] = ia; ind[i] = iind;
      if (mp != i) { x1[mp] = ox1; x2[mp] = ox2; ar[mp] = oar; ind[mp] = oind; }         // (its score: by the thread that decays position mp)
      s_cnt2[prv] = 0; s_dead2[prv] = 0;
    }
    const int n_touch = s_cnt2[cur];
    for (int k = tid; k < n_touch; k += SNT) {
      const int pos = slot[k];
      const bool moved = pos == mp;
      const float px1 = moved ? ox1 : x1[pos], px2 = moved ? ox2 : x2[pos], par = moved ? oar : ar[pos], psc = moved ? osc : sc[pos];
      const float xx1 = fmaxf(ix1, px1);
      const float xx2 = fminf(ix2, px2);
      const float inter = fmaxf(0.f, xx2 - xx1);
      const float ovr = inter / (ia + par - inter);
      float weight = 1.f;
      if (p.method == 0) { if (ovr >= p.iou_thresh) weight = 0.f; }
      else if (p.method == 1) { if (ovr >= p.iou_thresh) weight = 1.f - ovr; }
      else if (p.method == 2) {
        weight = expf_glibc(-(ovr * ovr) / p.sigma, s_tab);  // std::exp(float) of the reference's C library, bit for bit
      }
      const float s_ = psc * weight;
      sc[pos] = s_;
      dead_cnt += (int)(s_ < p.min_score);
      best = umax64(best, snms_key(s_, pos));
    }
    if (dead_cnt) s_dead2[cur] = 1;
    publish_best(best);
    ++i;
  }
  const int n_out = (p.max_iters > 0) ? min(i, nsegs) : nsegs;
  for (int k = tid; k < n_out; k += SNT) out[k] = (long long)ind[k];
  if (tid == 0) p.out_counts[q] = n_out;
}

int launch_softnms(const SoftNmsArgs& a, int nq, hipStream_t st) {
  if (nq <= 0) return 0;
  DCF_CHECK(a.n_max >= 0 && a.n_max <= NMS_BIG_MAX, "softnms: n=%d exceeds %d", a.n_max, NMS_BIG_MAX);
  DCF_CHECK(a.method >= 0 && a.method <= 2, "softnms: method must be 0, 1 or 2");
  DCF_CHECK(a.stride >= a.n_max, "softnms: stride < n_max");
  ProfScope prof("softnms_1d", st, 0.0, 0.0);
  if (a.n_max <= NMS_CAP) {
    if (a.n_max <= 768) hipLaunchKernelGGL((k_softnms<false, 256>), dim3(nq), dim3(256), 0, st, a, (unsigned char*)nullptr, (size_t)0);
    else if (a.n_max <= 3072) hipLaunchKernelGGL((k_softnms<false, 512>), dim3(nq), dim3(512), 0, st, a, (unsigned char*)nullptr, (size_t)0);
    else hipLaunchKernelGGL((k_softnms<false, 1024>), dim3(nq), dim3(1024), 0, st, a, (unsigned char*)nullptr, (size_t)0);
    DCF_HIP(hipGetLastError());
    return 0;
  }
  const size_t per_q = (softnms_big_scratch(a.n_max) + 255) & ~(size_t)255;
  unsigned char* scratch = nullptr;
  DCF_HIP(hipMallocAsync((void**)&scratch, per_q * nq, st));
  hipLaunchKernelGGL((k_softnms<true, 1024>), dim3(nq), dim3(1024), 0, st, a, scratch, per_q);
  const hipError_t e = hipGetLastError();
  DCF_HIP(hipFreeAsync(scratch, st));
  DCF_HIP(e);
  return 0;
}

// ------------------------------------------------------------------------------------------
// segment voting: one wavefront per NMS survivor
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_voting(VotingArgs p) {
  const int q = blockIdx.y, i = blockIdx.x, lane = threadIdx.x;
  const int n1 = p.n1_counts ? min(p.n1_counts[q], p.n1_max) : p.n1_max;
  const int n2 = p.n2_counts ? min(p.n2_counts[q], p.n2_max) : p.n2_max;
  if (i >= n1) return;
  const float* ns = p.nms_segs + ((size_t)q * p.n1_stride + i) * p.nms_ld;
  const float* as = p.all_segs + (size_t)q * p.n2_stride * 2;
  const float* sc = p.all_scores + (size_t)q * p.n2_stride;
  const float a0 = ns[0], a1 = ns[1];
  float sw = 0.f, sl = 0.f, sr = 0.f;
  for (int j = lane; j < n2; j += 64) {
    const float b0 = as[2 * j], b1 = as[2 * j + 1];
    const float left = fmaxf(a0, b0), right = fminf(a1, b1);
    const float ov = fmaxf(right - left, 0.f);
    const float uni = (a1 - a0) + (b1 - b0) - ov;
    const float iou = ov / uni;
    const float wgt = (iou >= p.iou_thresh) ? sc[j] : 0.f;
    sw += wgt; sl += wgt * b0; sr += wgt * b1;
  }
  sw = wave_sum(sw); sl = wave_sum(sl); sr = wave_sum(sr);
  if (lane == 0) {
    float* o = p.out + ((size_t)q * p.n1_stride + i) * 2;
    o[0] = sl / sw;
    o[1] = sr / sw;
  }
}

int launch_voting(const VotingArgs& a, int nq, hipStream_t st) {
  if (nq <= 0 || a.n1_max <= 0) return 0;
  hipLaunchKernelGGL(k_voting, dim3(a.n1_max, nq), dim3(64), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
