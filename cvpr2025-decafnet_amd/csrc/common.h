// Shared device/host helpers for the DeCafNet gfx950 kernels.
//
// Layout convention used by every kernel in this directory ("token-major"):
//   an activation is a row-major matrix [rows][ld] of fp32, one row per clip position,
//   rows ordered [batch b][position t]  (row = b * T + t), channels contiguous.
// The reference keeps (bs, C, T) channel-major tensors; the only channel-major buffers we
// touch are the user's inputs (vid, shallow_vid, text), which the first kernels transpose
// on the fly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DCF_WAVE 64

namespace dcf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- wave-level reductions (64 lanes) -------------------------------------------------
// Cross-lane adds go through DPP (data-parallel primitives: quad_perm / row mirrors / row broadcasts),
// which cost a few cycles each; __shfl_xor lowers to ds_bpermute (an LDS crossbar round trip of
// ~100 cycles per step) and made every LayerNorm latency bound.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_zero(float v) {   // lanes not written by the DPP move read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_self(float v) {   // lanes not written keep their own value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <- lane 7-i of its 8-lane half row
constexpr int DPP_MIRROR = 0x140;       // lane i <- lane 15-i of its 16-lane row
constexpr int DPP_BCAST15 = 0x142;      // lane 15 of row r -> every lane of row r+1
constexpr int DPP_BCAST31 = 0x143;      // lane 31 -> every lane of rows 2,3

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_zero<DPP_XOR1>(v);
  v += dpp_zero<DPP_XOR2>(v);
  v += dpp_zero<DPP_HALF_MIRROR>(v);
  v += dpp_zero<DPP_MIRROR>(v);           // every lane: sum of its 16-lane row
  v += dpp_zero<DPP_BCAST15, 0xA>(v);     // rows 1,3 += rows 0,2
  v += dpp_zero<DPP_BCAST31, 0xC>(v);     // rows 2,3 += row 1 total; lane 63 holds the wave sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_self<DPP_XOR1>(v));
  v = fmaxf(v, dpp_self<DPP_XOR2>(v));
  v = fmaxf(v, dpp_self<DPP_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_self<DPP_MIRROR>(v));
  v = fmaxf(v, dpp_self<DPP_BCAST15, 0xA>(v));
  v = fmaxf(v, dpp_self<DPP_BCAST31, 0xC>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// xor-16 / xor-32 butterflies across the four 16-lane rows of a wave with the gfx950 lane-swap
// instructions (VALU speed; the ds_bpermute behind __shfl_xor costs an LDS round trip):
//   v_permlane16_swap a, b : swaps the odd rows of a with the even rows of b
//   v_permlane32_swap a, b : swaps the upper half of a with the lower half of b
// With a == b == v the two results hold {r0,r0,r2,r2}/{r1,r1,r3,r3} resp. {lo,lo}/{hi,hi}.
__device__ __forceinline__ float xor16_sum(float v) {
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// sum within aligned groups of G lanes (G power of two <= 64); every lane of the group gets the sum
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G == 64) return wave_sum(v);
  if constexpr (G >= 2) v += dpp_zero<DPP_XOR1>(v);
  if constexpr (G >= 4) v += dpp_zero<DPP_XOR2>(v);
  if constexpr (G >= 8) v += dpp_zero<DPP_HALF_MIRROR>(v);
  if constexpr (G >= 16) v += dpp_zero<DPP_MIRROR>(v);
  if constexpr (G >= 32) v = xor16_sum(v);
  return v;
}

// NOTE: register arrays must use the native ext_vector type f32x4, never HIP's float4 struct:
// struct copies lower to cross-address-space memcpy, which blocks SROA and spills to scratch.

// ---- a row of up to 4*256 channels held by one wave: lane owns f32x4 chunks -----------
// chunk j of lane l covers channels [256*j + 4*l, 256*j + 4*l + 4)
template <int NCH>
struct Row {
  f32x4 v[NCH];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int j = 0; j < NCH; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __device__ __forceinline__ void load(const float* __restrict__ p, int C, int lane) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int c = 256 * j + 4 * lane;
      v[j] = (c < C) ? *reinterpret_cast<const f32x4*>(p + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ p, int C, int lane) const {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int c = 256 * j + 4 * lane;
      if (c < C) *reinterpret_cast<f32x4*>(p + c) = v[j];
    }
  }
  __device__ __forceinline__ float sum() const {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    return wave_sum(s);
  }
};

// LayerNorm over the channels of a Row (biased variance, eps inside sqrt; two-pass like the
// reference's channel LayerNorm, libs/modeling/blocks.py:125-131).  Inactive chunks hold 0
// and are excluded from the statistics.
template <int NCH>
__device__ __forceinline__ void row_layernorm(Row<NCH>& r, int C, int lane, const float* __restrict__ w,
                                              const float* __restrict__ b, float eps = 1e-5f) {
  const float inv_c = 1.0f / (float)C;
  float mean = r.sum() * inv_c;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int c = 256 * j + 4 * lane;
    if (c < C) {
      r.v[j].x -= mean; r.v[j].y -= mean; r.v[j].z -= mean; r.v[j].w -= mean;
      sq += (r.v[j].x * r.v[j].x + r.v[j].y * r.v[j].y) + (r.v[j].z * r.v[j].z + r.v[j].w * r.v[j].w);
    }
  }
  float var = wave_sum(sq) * inv_c;
  float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int c = 256 * j + 4 * lane;
    if (c < C) {
      f32x4 x = r.v[j];
      x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs;
      if (w != nullptr) {
        f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        f32x4 bb = *reinterpret_cast<const f32x4*>(b + c);
        x.x = x.x * ww.x + bb.x; x.y = x.y * ww.y + bb.y; x.z = x.z * ww.z + bb.z; x.w = x.w * ww.w + bb.w;
      }
      r.v[j] = x;
    }
  }
}

// A per-channel parameter vector (LayerNorm weight / bias, one tap of a depthwise convolution) of a strip kernel: rows of one
// chunk (C <= 256) keep it in registers for the whole strip, so the strip loop issues no load but its row prefetches --
// a parameter load inside the loop would make every s_waitcnt on it wait for the older, slower row loads too (a wave's
// vector-memory results come back in order).  Wider rows read it at the point of use (L1 hits).
template <int NCH>
struct RowParam {
  static constexpr bool CACHED = NCH == 1;
  Row<NCH> r;
  const float* p;
  __device__ __forceinline__ void init(const float* __restrict__ p_, int C, int lane) {
    p = p_;
    if constexpr (CACHED) { if (p_) r.load(p_, C, lane); else r.zero(); }
  }
  __device__ __forceinline__ f32x4 get(int j, int lane) const {
    if constexpr (CACHED) return r.v[j];
    else return *reinterpret_cast<const f32x4*>(p + 256 * j + 4 * lane);
  }
};

// row_layernorm with the affine parameters as RowParam (w.p == nullptr: no affine); same operation order, same bits
template <int NCH>
__device__ __forceinline__ void row_layernorm(Row<NCH>& r, int C, int lane, const RowParam<NCH>& w, const RowParam<NCH>& b,
                                              float eps = 1e-5f) {
  const float inv_c = 1.0f / (float)C;
  float mean = r.sum() * inv_c;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int c = 256 * j + 4 * lane;
    if (c < C) {
      r.v[j].x -= mean; r.v[j].y -= mean; r.v[j].z -= mean; r.v[j].w -= mean;
      sq += (r.v[j].x * r.v[j].x + r.v[j].y * r.v[j].y) + (r.v[j].z * r.v[j].z + r.v[j].w * r.v[j].w);
    }
  }
  float var = wave_sum(sq) * inv_c;
  float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int c = 256 * j + 4 * lane;
    if (c < C) {
      f32x4 x = r.v[j];
      x.x *= rs; x.y *= rs; x.z *= rs; x.w *= rs;
      if (w.p != nullptr) {
        const f32x4 ww = w.get(j, lane), bb = b.get(j, lane);
        x.x = x.x * ww.x + bb.x; x.y = x.y * ww.y + bb.y; x.z = x.z * ww.z + bb.z; x.w = x.w * ww.w + bb.w;
      }
      r.v[j] = x;
    }
  }
}

// exact (erf) GELU, nn.GELU default (libs/modeling/blocks.py:531): 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |x| P(|x|) with
// P = 0.5 (1 - erf(|x| / sqrt 2)) = 0.5 poly(t) e^{-z^2}, z = |x| / sqrt 2, t = 1 / (1 + 0.3275911 z): erf by Abramowitz & Stegun
// 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level) on one v_exp_f32 and one v_rcp_f32 instead of ocml's erff (~40
// instructions; it was 10-13 % of the FFN fc GEMM).  In this form there is no sign select and no cancellation on either side
// (P <= 0.5), and the whole function is 15 vector instructions: z carries the sqrt(log2 e) of the exponential, the 0.5 rides in the
// polynomial's coefficients, and the last step is one fma on max(x, 0).  The FFN kernels (ffn_chain.hip) run the same steps on
// 16 x (their operand scale); scaling by a power of two commutes with every rounding here, so they agree with this function bit
// for bit.
constexpr float GELU_CZ = 0.849321800288019f;          // sqrt(log2(e) / 2): zs = |x| GELU_CZ = z sqrt(log2 e)
constexpr float GELU_CT = 0.2727374808792225f;         // 0.3275911 / sqrt(log2 e): 1 + 0.3275911 z = 1 + GELU_CT zs
constexpr float GELU_A1 = 0.5f * 0.254829592f, GELU_A2 = 0.5f * -0.284496736f, GELU_A3 = 0.5f * 1.421413741f,
                GELU_A4 = 0.5f * -1.453152027f, GELU_A5 = 0.5f * 1.061405429f;
__device__ __forceinline__ float gelu_half_poly(float t) {      // 0.5 (a1 t + a2 t^2 + ... + a5 t^5)
  return t * (GELU_A1 + t * (GELU_A2 + t * (GELU_A3 + t * (GELU_A4 + t * GELU_A5))));
}
// One-pass LayerNorm statistics (sum, sum of squares carried between kernels: var = E[x^2] - mean^2) lose ~1e-7 mean^2 / var of the
// variance to cancellation -- fp32 level while a row's mean does not dwarf its spread, which holds for every activation the
// synthetic and the reference-generated fixtures produce (|mean| / sigma <= 3), but is an assumption about a real checkpoint.  It is
// guarded, not assumed: a consumer that meets a row with mean^2 > LN_ILL_RATIO (var + eps) -- |mean| > 8 sigma, relative error of rstd
// beyond ~1e-5 -- raises bit 1 of the sticky numerics word (dcf_numerics_status: 16); the host then switches the model to the
// standalone two-pass LayerNorm launches (dcf_model_set_ln_carry) and repeats the forward.  All-zero (masked) rows never trip it.
constexpr float LN_ILL_RATIO = 64.f;
__device__ __forceinline__ bool ln_ill(float mean, float var) { return mean * mean > LN_ILL_RATIO * (var + 1e-5f); }

__device__ __forceinline__ float relu_max(float x) {             // max(x, 0) as ONE v_max_f32 (fmaxf puts a canonicalising max in front)
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ float gelu_erf(float x) {
  const float zs = fabsf(x) * GELU_CZ;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(zs, GELU_CT, 1.0f));
  const float ex = __builtin_amdgcn_exp2f(-zs * zs);
  return __builtin_fmaf(-fabsf(x), gelu_half_poly(t) * ex, relu_max(x));
}

}  // namespace dcf

// ---- optional per-launch timing (HIP events on the launch stream; off by default) --------------
namespace dcf {
struct ProfScope {
  int idx;
  hipStream_t st;
  // flops / bytes are the ALGORITHMIC work of this launch, not measured traffic
  ProfScope(const char* name, hipStream_t st, double flops, double bytes);
  ~ProfScope();
};
}  // namespace dcf

// ---- host-side error plumbing -----------------------------------------------------------
namespace dcf {
void set_error(const char* fmt, ...);
}
#define DCF_CHECK(cond, ...)            \
  do {                                  \
    if (!(cond)) {                      \
      dcf::set_error(__VA_ARGS__);      \
      return -1;                        \
    }                                   \
  } while (0)
#define DCF_HIP(expr)                                                              \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      dcf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -1;                                                                   \
    }                                                                              \
  } while (0)
