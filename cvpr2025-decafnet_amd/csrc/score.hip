// Sidekick/expert clip scoring and the block top-k gate.
//
// Scoring (libs/modeling/model.py:500-505) is one streaming pass over the (D, T) shallow feature
// matrix (64 MiB at D=1024, T=16384): lanes run along T (16 B per lane, 1 KiB per wave
// transaction), the D channels are cut into SCORE_SLICES slices so that >= 1024 wavefronts are in
// flight, and the per-slice partial sums are combined in a fixed order (deterministic: the gate
// below is a discrete decision, so no float atomics).
//
// Gate (model.py:531-541): ceil-mode block mean over `sn` clips -> ascending rank -> keep the top
// int(sratio * n) blocks -> nearest-neighbour upsample back to clips.  One workgroup per query,
// rank-by-counting in LDS (n <= 8192 blocks).  Ties are broken by block index (stable); the
// reference's argsort leaves tie order unspecified.
#include "common.h"
#include "score.h"

namespace dcf {

// entry v of a kernel-argument array for a wave-uniform v: a chain of selects over static indices (a dynamically indexed
// by-value argument array is copied to scratch memory first)
template <typename T>
__device__ __forceinline__ T pick(const T (&a)[SCORE_MAXVID], int v) {
  T r = a[0];
#pragma unroll
  for (int i = 1; i < SCORE_MAXVID; ++i) r = v == i ? a[i] : r;
  return r;
}

// grid = total queries; video of query row q: the last v with qoff[v] <= q
__global__ __launch_bounds__(256) void k_text_cls_norm(ScoreArgs p) {
  __shared__ float red[4];
  const int q = blockIdx.x, tid = threadIdx.x;
  int v = 0;
#pragma unroll
  for (int i = 1; i < SCORE_MAXVID; ++i) v = (i < p.nvid && p.qoff[i] <= q) ? i : v;
  const float* cls = pick(p.text_cls, v) + (size_t)(q - pick(p.qoff, v)) * p.D;
  const int D = p.D;
  float s = 0.f;
  for (int c = tid; c < D; c += 256) { float x = cls[c]; s += x * x; }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const float inv = p.norm ? 1.0f / (nrm + 1e-4f) : 1.0f;
  for (int c = tid; c < D; c += 256) p.tn[(size_t)q * D + c] = cls[c] * inv;
}

// grid = (ceil(T/256), SCORE_SLICES, nvid * query chunks); block = 64 (one wave, 4 clips per lane)
template <int NQ>
__global__ __launch_bounds__(64) void k_sidekick_partial(ScoreArgs p) {
  const int lane = threadIdx.x;
  const int t = (blockIdx.x * 64 + lane) * 4;
  const int slice = blockIdx.y;
  const int vid = blockIdx.z % p.nvid, q0 = (blockIdx.z / p.nvid) * SCORE_MAXQ;
  const int nqv = pick(p.nq, vid) - q0;                 // queries of this video left for this chunk (may be <= 0 or > NQ)
  if (nqv <= 0) return;
  const int qrow = pick(p.qoff, vid) + q0;              // first of them in tn / correl
  const float* shallow = pick(p.shallow, vid);
  const float* tn = p.tn + (size_t)qrow * p.D;
  const int cps = (p.D + SCORE_SLICES - 1) / SCORE_SLICES;
  const int c0 = slice * cps, c1 = min(c0 + cps, p.D);
  const bool vec = (p.T & 3) == 0;
  f32x4 ss = {0.f, 0.f, 0.f, 0.f};
  f32x4 dot[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) dot[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t < p.T) {
    int c = c0;
    if (vec) {
      // eight channel rows requested before the first is used: a load per iteration left the loop waiting for one round trip
      // per channel (35 us for 64 MiB)
      for (; c + 8 <= c1; c += 8) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const f32x4*>(shallow + (size_t)(c + u) * p.T + t);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          ss += x[u] * x[u];
#pragma unroll
          for (int q = 0; q < NQ; ++q) dot[q] += x[u] * tn[(size_t)(q < nqv ? q : 0) * p.D + c + u];
        }
      }
    }
    for (; c < c1; ++c) {
      const float* src = shallow + (size_t)c * p.T + t;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (vec) x = *reinterpret_cast<const f32x4*>(src);
      else { x.x = src[0]; if (t + 1 < p.T) x.y = src[1]; if (t + 2 < p.T) x.z = src[2]; if (t + 3 < p.T) x.w = src[3]; }
      ss += x * x;
#pragma unroll
      for (int q = 0; q < NQ; ++q) dot[q] += x * tn[(size_t)(q < nqv ? q : 0) * p.D + c];   // wave-uniform scalar load
    }
    float* dst = p.partial + ((size_t)slice * (p.NQ + p.nvid) + qrow + vid) * p.T + t;      // the video's ss row
    const int n = min(4, p.T - t);
    for (int i = 0; i < n; ++i) {
      if (q0 == 0) dst[i] = ss[i];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqv) dst[(size_t)(1 + q) * p.T + i] = dot[q][i];
    }
  }
}

// grid = (ceil(T/256), nvid)
__global__ __launch_bounds__(256) void k_sidekick_final(ScoreArgs p) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= p.T) return;
  const int vid = blockIdx.y;
  const int nq = pick(p.nq, vid), qoff = pick(p.qoff, vid);
  const size_t rows = p.NQ + p.nvid, row0 = qoff + vid;
  float ss = 0.f;
  for (int s = 0; s < SCORE_SLICES; ++s) ss += p.partial[((size_t)s * rows + row0) * p.T + t];
  const float inv = p.norm ? 1.0f / (sqrtf(ss) + 1e-4f) : 1.0f;
  for (int q = 0; q < nq; ++q) {
    float d = 0.f;
    for (int s = 0; s < SCORE_SLICES; ++s) d += p.partial[((size_t)s * rows + row0 + 1 + q) * p.T + t];
    p.correl[(size_t)(qoff + q) * p.T + t] = d * inv;
  }
}

// the normalised text vectors alone (tn): the scores themselves then ride on the channel-major vid_map GEMMs (GemmArgs::score_out)
int launch_text_cls_norm(const ScoreArgs& a, hipStream_t st) {
  if (a.NQ <= 0 || a.nvid <= 0) return 0;
  DCF_CHECK(a.nvid <= SCORE_MAXVID, "sidekick: %d videos per launch > %d", a.nvid, SCORE_MAXVID);
  hipLaunchKernelGGL(k_text_cls_norm, dim3(a.NQ), dim3(256), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

int launch_sidekick(const ScoreArgs& a, hipStream_t st) {
  if (a.NQ <= 0 || a.T <= 0 || a.nvid <= 0) return 0;
  DCF_CHECK(a.nvid <= SCORE_MAXVID, "sidekick: %d videos per launch > %d", a.nvid, SCORE_MAXVID);
  int maxq = 0;
  for (int v = 0; v < a.nvid; ++v) maxq = a.nq[v] > maxq ? a.nq[v] : maxq;
  const int chunks = (maxq + SCORE_MAXQ - 1) / SCORE_MAXQ;
  ProfScope prof("sidekick_score", st, 2.0 * (a.NQ + a.nvid) * a.D * a.T, 4.0 * (double)a.D * a.T * a.nvid * chunks);
  hipLaunchKernelGGL(k_text_cls_norm, dim3(a.NQ), dim3(256), 0, st, a);
  dim3 grid((a.T + 255) / 256, SCORE_SLICES, a.nvid * chunks);
  switch (maxq < SCORE_MAXQ ? maxq : SCORE_MAXQ) {
    case 1: hipLaunchKernelGGL(k_sidekick_partial<1>, grid, dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_sidekick_partial<2>, grid, dim3(64), 0, st, a); break;
    case 3: hipLaunchKernelGGL(k_sidekick_partial<3>, grid, dim3(64), 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_sidekick_partial<4>, grid, dim3(64), 0, st, a); break;
    case 5: hipLaunchKernelGGL(k_sidekick_partial<5>, grid, dim3(64), 0, st, a); break;
    case 6: hipLaunchKernelGGL(k_sidekick_partial<6>, grid, dim3(64), 0, st, a); break;
    case 7: hipLaunchKernelGGL(k_sidekick_partial<7>, grid, dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL(k_sidekick_partial<8>, grid, dim3(64), 0, st, a); break;
  }
  hipLaunchKernelGGL(k_sidekick_final, dim3((a.T + 255) / 256, a.nvid), dim3(256), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
constexpr int GATE_MAX_BLOCKS = 8192;

constexpr int GATE_STAGE = 24576;     // floats of correl staged in LDS per pass (96 KiB)

__global__ __launch_bounds__(1024) void k_gate(GateArgs p) {
  __shared__ __attribute__((aligned(16))) float pooled[GATE_MAX_BLOCKS];
  __shared__ uint8_t sel[GATE_MAX_BLOCKS];
  __shared__ float stage[GATE_STAGE];
  __shared__ int s_len;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* correl = p.correl + (size_t)(p.q0 + b) * p.T;
  const uint8_t* vid_mask = p.vid_mask + (size_t)((p.vmap >> (4 * (b & 15))) & 15ull) * p.T;
  // vid_len = vid_masks.sum()  (model.py:531), computed on device: no host sync
  if (tid == 0) s_len = 0;
  __syncthreads();
  int cnt = 0;
  {
    const int T16 = p.T & ~15;                        // 16 mask bytes per load (the mask pointer is 16-byte aligned
    const bool al = (reinterpret_cast<uintptr_t>(vid_mask) & 15) == 0;   // for torch storage; else byte loop)
    int t0 = 0;
    if (al) {
      for (int t = tid * 16; t < T16; t += 1024 * 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(vid_mask + t);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s8 = 0; s8 < 32; s8 += 8) cnt += ((w[i] >> s8) & 0xffu) ? 1 : 0;
      }
      t0 = T16;
    }
    for (int t = t0 + tid; t < p.T; t += 1024) cnt += vid_mask[t] ? 1 : 0;
  }
  cnt = (int)wave_sum((float)cnt);
  if ((tid & 63) == 0 && cnt) atomicAdd(&s_len, cnt);
  __syncthreads();
  const int len = s_len;
  const int n = (len + p.sn - 1) / p.sn;             // avg_pool1d(ceil_mode=True)
  // block means: sequential fp32 sum then one division by the true window size (ATen avg_pool2d).  The scores are
  // staged through LDS (coalesced) so that the 60 dependent adds of a block do not each wait for a global load
  // (29 -> 12 us at T = 16384); blocks wider than the stage buffer are summed from global memory.
  const int per = p.sn <= GATE_STAGE ? GATE_STAGE / p.sn : 0;     // whole blocks per pass
  if (per > 0) {
    for (int i0 = 0; i0 < n; i0 += per) {
      const int e0 = i0 * p.sn, e1 = min((i0 + per) * p.sn, len);
      __syncthreads();
      for (int t = e0 + tid; t < e1; t += 1024) stage[t - e0] = correl[t];
      __syncthreads();
      for (int i = i0 + tid; i < min(i0 + per, n); i += 1024) {
        const int s0 = i * p.sn, s1 = min(s0 + p.sn, len);
        float acc = 0.f;
        for (int t = s0; t < s1; ++t) acc += stage[t - e0];
        pooled[i] = acc / (float)(s1 - s0);
      }
    }
  } else {
    for (int i = tid; i < n; i += 1024) {
      const int s0 = i * p.sn, s1 = min(s0 + p.sn, len);
      float acc = 0.f;
      for (int t = s0; t < s1; ++t) acc += correl[t];
      pooled[i] = acc / (float)(s1 - s0);
    }
  }
  __syncthreads();
  const int k = (int)(p.sratio * (double)n);          // int(ratio * n): truncation of the double product
  for (int i = tid; i < n; i += 1024) {
    unsigned char s = 1;
    if (k > 0) {                                      // ranked[-0:] keeps everything (model.py:535)
      const float v = pooled[i];
      int rank = 0;
      int j = 0;
      for (; j + 4 <= n; j += 4) {                     // broadcast 16-byte LDS reads, four compares each
        const f32x4 u = *reinterpret_cast<const f32x4*>(&pooled[j]);
        rank += (u.x < v || (u.x == v && j < i)) ? 1 : 0;
        rank += (u.y < v || (u.y == v && j + 1 < i)) ? 1 : 0;
        rank += (u.z < v || (u.z == v && j + 2 < i)) ? 1 : 0;
        rank += (u.w < v || (u.w == v && j + 3 < i)) ? 1 : 0;
      }
      for (; j < n; ++j) {
        const float u = pooled[j];
        rank += (u < v || (u == v && j < i)) ? 1 : 0;
      }
      s = rank >= n - k;
    }
    sel[i] = s;
  }
  __syncthreads();
  // nearest upsample (ATen upsample_nearest1d): identity / >>1 / min(floor(t * float(n/len)), n-1)
  const float scale = (len > 0) ? (float)n / (float)len : 0.f;
  for (int t = tid; t < p.T; t += 1024) {
    float g = 0.f;
    if (t < len) {
      int src;
      if (len == n) src = t;
      else if (len == 2 * n) src = t >> 1;
      else src = min((int)floorf((float)t * scale), n - 1);
      g = sel[src] ? 1.f : 0.f;
    }
    p.gate[(size_t)b * p.T + t] = g;
    const bool m = vid_mask[t] != 0;
    p.mask_out[(size_t)b * p.T + t] = p.msf ? m : (m && g != 0.f);
  }
  // which 64-clip row tiles this query keeps at all (GemmArgs::tile_skip): a wave's 64 lanes are the clips of one tile
  if (p.tile_flags) {
    for (int t0 = (tid & ~63); t0 < p.nflags * 64; t0 += 1024) {
      const int t = t0 + (tid & 63);
      float g = 0.f;
      if (t < len) {
        int src;
        if (len == n) src = t;
        else if (len == 2 * n) src = t >> 1;
        else src = min((int)floorf((float)t * scale), n - 1);
        g = sel[src] ? 1.f : 0.f;
      }
      const unsigned long long keep = __ballot(g != 0.f);
      if ((tid & 63) == 0) p.tile_flags[(size_t)b * p.nflags + (t0 >> 6)] = keep ? 1 : 0;
    }
  }
}

int launch_gate(const GateArgs& a, hipStream_t st) {
  if (a.B <= 0) return 0;
  DCF_CHECK(a.sn >= 1, "gate: sn must be >= 1");
  DCF_CHECK((a.T + a.sn - 1) / a.sn <= GATE_MAX_BLOCKS, "gate: more than %d pooling blocks (T=%d, sn=%d)", GATE_MAX_BLOCKS, a.T, a.sn);
  ProfScope prof("gate_topk", st, 0.0, 4.0 * 2.0 * a.B * a.T);
  hipLaunchKernelGGL(k_gate, dim3(a.B), dim3(1024), 0, st, a);
  DCF_HIP(hipGetLastError());
  return 0;
}

}  // namespace dcf
