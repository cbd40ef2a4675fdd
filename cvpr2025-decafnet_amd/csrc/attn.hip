// Attention cores of the grounding path (fp32).
//
//  * xattn:  clips attend to <= 64 text tokens (MaskedMHA global branch,
//            libs/modeling/blocks.py:374-389).  K/V of one query (Lk x C, <= 64 KiB each) are staged
//            once per workgroup in LDS; every wavefront then streams clip rows: 1 KiB coalesced
//            read of q, 1 KiB coalesced write of the context -- 8 B/channel/clip of HBM traffic,
//            which is the algorithmic minimum.
//  * local:  sliding-window self attention |i-j| <= w/2 (MaskedMHA local branch, blocks.py:204-325,
//            357-373, restated as a band).  A wavefront owns one clip row; its <= w neighbour K/V
//            rows come from L1/L2.
//
// In both kernels a lane owns 4 consecutive channels of the row, a head is a group of d/4 adjacent
// lanes, the q.k dot product is a 4-FMA partial + log2(d/4) cross-lane adds, and the softmax is
// carried online (running max / running sum), so no score matrix is ever materialised.
#include "attn.h"
#include "common.h"

namespace dcf {

template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
  if constexpr (LPH == 1) return v;
  else return group_sum<LPH>(v);
}

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
// cross attention on the matrix cores (fp32 MFMA 16x16x4, exact fp32)
// ------------------------------------------------------------------------------------------
// Workgroup = 4 wavefronts = (64 clip rows, one head); grid = (row groups, heads, B), each workgroup
// stages K_h and V_h of its (query, head) once in LDS and then strides over row groups.
// Per wavefront (16 clip rows):
//   S^T = K_h Q^T   "swapped" product: A = K tile (16 keys x d), B = Q^T.  The D fragment then holds,
//                   per lane (row r = lane & 15, g = lane >> 4), the scores of keys 16*kt + 4*g + j:
//                   the softmax over keys is in-lane plus two cross-lane steps (xor 16, xor 32).
//   O^T = V_h^T P^T A = V^T tile (16 channels x keys), B = P^T: the B fragment of k-step j IS the
//                   lane's score register j -- the probabilities never leave their registers.
//   Q goes global -> registers directly in B-fragment order (every byte of the q slice is read once,
//   as 64-byte pieces), O goes registers -> global as float4 (4 consecutive channels per lane).
// The k index of each 16-wide chunk is permuted (lane group g owns d = 16c + 4g + j for step j) for
// both operands alike, so one ds_read_b128 / global_load_dwordx4 feeds 4 MFMAs.
template <int D16, int NKT>
__global__ __launch_bounds__(256) void k_xattn_mfma(XAttnArgs p) {
  constexpr int D = 16 * D16;          // head dim
  constexpr int PITCH = D + 4;         // LDS row pitch in floats
  constexpr int LKP = 16 * NKT;        // padded key count
  extern __shared__ float smem[];
  float* Ks = smem;                    // [LKP][PITCH], pre-scaled by d^-1/4
  float* Vs = smem + LKP * PITCH;      // [LKP][PITCH]
  float* Ms = Vs + LKP * PITCH;        // [LKP] additive key mask (0 / -inf)
  const int C = p.C, Lk = p.Lk;
  const int head = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const float scale = 1.0f / sqrtf(sqrtf((float)D));

  for (int i = tid; i < LKP * (D / 4); i += 256) {
    const int key = i / (D / 4), c4 = i % (D / 4);
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (key < Lk) {
      const size_t off = ((size_t)b * Lk + key) * C + (size_t)head * D + c4 * 4;
      kv = *reinterpret_cast<const f32x4*>(p.K + off) * scale;
      vv = *reinterpret_cast<const f32x4*>(p.V + off);
    }
    *reinterpret_cast<f32x4*>(Ks + key * PITCH + c4 * 4) = kv;
    *reinterpret_cast<f32x4*>(Vs + key * PITCH + c4 * 4) = vv;
  }
  for (int i = tid; i < LKP; i += 256) Ms[i] = (i < Lk && p.kvmask[(size_t)b * Lk + i]) ? 0.f : -INFINITY;
  __syncthreads();

  const int n_groups = (p.T + 63) / 64;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int t = grp * 64 + wave * 16 + r;
    const bool live = t < p.T;
    const int64_t row = (int64_t)b * p.T + (live ? t : p.T - 1);
    // ---- Q fragment: q[c] = Q[row][head*D + 16c + 4g .. +3], scaled
    f32x4 q[D16];
    const float* qp = p.Q + row * C + (size_t)head * D + 4 * g;
#pragma unroll
    for (int c = 0; c < D16; ++c) q[c] = *reinterpret_cast<const f32x4*>(qp + 16 * c) * scale;
    // ---- S^T tiles
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < D16; ++c) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + r) * PITCH + 16 * c + 4 * g);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, q[c].x, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, q[c].y, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, q[c].z, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, q[c].w, s[kt], 0, 0, 0);
      }
    }
    // ---- softmax over keys (lane holds keys 16kt + 4g + j of its row)
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const f32x4 mk = *reinterpret_cast<const f32x4*>(Ms + 16 * kt + 4 * g);
      s[kt] += mk;
      mx = fmaxf(fmaxf(mx, fmaxf(s[kt].x, s[kt].y)), fmaxf(s[kt].z, s[kt].w));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt].x = expf(s[kt].x - mx); s[kt].y = expf(s[kt].y - mx);
      s[kt].z = expf(s[kt].z - mx); s[kt].w = expf(s[kt].w - mx);
      sum += (s[kt].x + s[kt].y) + (s[kt].z + s[kt].w);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;          // all keys masked: exp(nan) -> NaN row, as the reference
    // ---- O^T tiles and store
    float* op = p.O + row * C + (size_t)head * D + 4 * g;
#pragma unroll
    for (int ct = 0; ct < D16; ++ct) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        const float* vp = Vs + (16 * kt + 4 * g) * PITCH + 16 * ct + r;
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], s[kt].x, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[PITCH], s[kt].y, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[2 * PITCH], s[kt].z, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[3 * PITCH], s[kt].w, o, 0, 0, 0);
      }
      if (live) *reinterpret_cast<f32x4*>(op + 16 * ct) = o * inv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// sliding-window self attention
// ------------------------------------------------------------------------------------------
template <int NCH, int LPH>
__global__ __launch_bounds__(256) void k_local_attn(LocalAttnArgs p) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)p.B * p.T) return;
  const int C = p.C;
  const int t = (int)(r % p.T);
  const int64_t base = r - t;
  const int half = p.window / 2;
  const float scale = 1.0f / sqrtf(sqrtf((float)(C / p.heads)));
  Row<NCH> q;
  if (!p.mask[r]) {                                    // padded query rows are forced to 0 (blocks.py:293)
    q.zero();
    q.store(p.O + r * C, C, lane);
    return;
  }
  q.load(p.Q + r * C, C, lane);
  f32x4 acc[NCH];
  float m[NCH], l[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) { q.v[j] *= scale; acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; m[j] = -INFINITY; l[j] = 0.f; }
  const int lo = max(t - half, 0), hi = min(t + half, p.T - 1);   // out-of-range keys are -inf (blocks.py:260-261)
  for (int u = lo; u <= hi; ++u) {
    const float pen = p.mask[base + u] ? 0.f : -1e4f;  // padded keys get a finite -1e4 (blocks.py:279)
    Row<NCH> k, v;
    k.load(p.K + (base + u) * C, C, lane);
    v.load(p.V + (base + u) * C, C, lane);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      float s = head_sum<LPH>(dot4(q.v[j], k.v[j] * scale)) + pen;
      float mn = fmaxf(m[j], s);
      float corr = expf(m[j] - mn);
      float e = expf(s - mn);
      acc[j] = acc[j] * corr + e * v.v[j];
      l[j] = l[j] * corr + e;
      m[j] = mn;
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) q.v[j] = acc[j] / l[j];
  q.store(p.O + r * C, C, lane);
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

#define XATTN_LAUNCH(NCH_, LPH_, grid, lds, st, a) \
  hipLaunchKernelGGL((k_xattn_valu<NCH_, LPH_>), grid, dim3(256), lds, st, a)
#define LOCAL_LAUNCH(NCH_, LPH_, grid, st, a) \
  hipLaunchKernelGGL((k_local_attn<NCH_, LPH_>), grid, dim3(256), 0, st, a)

template <int D16>
static int launch_xattn_mfma(const XAttnArgs& a, hipStream_t st) {
  const int nkt = (a.Lk + 15) / 16;
  const int n_groups = (a.T + 63) / 64;
  // enough workgroups to fill the chip, few enough that K/V staging (2*Lk*d floats) is amortised
  int gx = n_groups;
  const int per = a.heads * a.B;
  const int cap = (2048 + per - 1) / per;
  if (gx > cap) gx = cap;
  dim3 grid(gx, a.heads, a.B);
  const size_t lds = (size_t)(2 * 16 * nkt * (16 * D16 + 4) + 16 * nkt) * sizeof(float);
  switch (nkt) {
    case 1: hipLaunchKernelGGL((k_xattn_mfma<D16, 1>), grid, dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL((k_xattn_mfma<D16, 2>), grid, dim3(256), lds, st, a); break;
    case 3: hipLaunchKernelGGL((k_xattn_mfma<D16, 3>), grid, dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((k_xattn_mfma<D16, 4>), grid, dim3(256), lds, st, a); break;
  }
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
  if (a.Lk <= 64 && d % 16 == 0 && d <= 256 && (d & (d - 1)) == 0) {
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

int launch_local_attn(const LocalAttnArgs& a, hipStream_t st) {
  int64_t rows = (int64_t)a.B * a.T;
  if (rows <= 0) return 0;
  DCF_CHECK(a.window >= 1 && (a.window & 1), "local_attn: window must be odd");
  dim3 grid((unsigned)((rows + 3) / 4));
  ProfScope prof("local_attn", st, 4.0 * rows * a.C * a.window, 4.0 * 4.0 * rows * a.C);
  DISPATCH_ATTN(LOCAL_LAUNCH, a.C, a.heads, grid, st, a);
  return 0;
}

}  // namespace dcf
