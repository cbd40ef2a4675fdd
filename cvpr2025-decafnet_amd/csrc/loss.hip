// Point losses of the training objective, forward values only (libs/modeling/loss.py; Trainer.forward_backward,
// libs/worker_v2.py:441-461).  Elementwise arithmetic in the reference's fp32 operation order; the boolean-mask selections of
// the caller (`logits[fpn_masks]`, `offsets[pos_masks]`) are a byte mask here instead of a compaction; sums are taken in a
// fixed order (chunk partials, then one workgroup over the partials) so a value does not depend on scheduling.
#include "../../include/decafnet_hip.h"
#include "common.h"

namespace dcf {

constexpr int LOSS_NT = 256;
constexpr int LOSS_PER_BLOCK = LOSS_NT * 16;       // elements per workgroup of the partial-sum pass

// sigmoid_focal_loss, loss.py:5-57
__device__ __forceinline__ float focal_elem(float x, float t, float alpha, float gamma, bool smoothing) {
  const float mask = t >= 0.5f ? 1.f : 0.f;                       // positive mask (loss.py:38)
  const float p = 1.f / (1.f + expf(-x));                         // torch.sigmoid
  const float p_t = smoothing ? p * t + (1.f - p) * (1.f - t) : p * mask + (1.f - p) * (1.f - mask);
  // F.binary_cross_entropy_with_logits(x, t) = (1 - t) x + log(1 + exp(-x)), the log term formed without overflow
  const float ce = (1.f - t) * x + (fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))));
  const float m = 1.f - p_t;
  float loss = ce * (gamma == 2.f ? m * m : powf(m, gamma));
  if (alpha >= 0.f) loss = (alpha * mask + (1.f - alpha) * (1.f - mask)) * loss;
  return loss;
}

// ctr_giou_loss (kind 0, loss.py:60-109) / ctr_diou_loss (kind 1, loss.py:111-166) of one (left, right) offset pair
__device__ __forceinline__ float iou_elem(float lp, float rp, float lg, float rg, int kind, float eps) {
  const float lkis = fminf(lp, lg), rkis = fminf(rp, rg);
  const float intsctk = rkis + lkis;
  const float unionk = (lp + rp) + (lg + rg) - intsctk;
  const float iouk = intsctk / fmaxf(unionk, eps);
  float loss = 1.0f - iouk;
  if (kind == 1) {
    const float len_c = fmaxf(lp, lg) + fmaxf(rp, rg);             // smallest enclosing segment
    const float rho = 0.5f * (rp - lp - rg + lg);                  // offset between the centres
    const float q = rho / fmaxf(len_c, eps);
    loss = loss + q * q;
  }
  return loss;
}

struct LossArgs {
  const float* a;          // inputs / input_offsets
  const float* b;          // targets / target_offsets
  const uint8_t* select;
  long long n;
  float alpha, gamma, eps;
  int smoothing, kind;
  float* elem;
  float* part_sum;         // [blocks]
  int* part_cnt;           // [blocks]
};

template <bool IOU>
__global__ __launch_bounds__(LOSS_NT) void k_loss_partial(LossArgs p) {
  __shared__ float s_sum[LOSS_NT / 64];
  __shared__ int s_cnt[LOSS_NT / 64];
  const long long base = (long long)blockIdx.x * LOSS_PER_BLOCK;
  float acc = 0.f;
  int cnt = 0;
  for (int k = 0; k < LOSS_PER_BLOCK / LOSS_NT; ++k) {
    const long long i = base + (long long)k * LOSS_NT + threadIdx.x;
    if (i >= p.n) break;
    const bool sel = !p.select || p.select[i] != 0;
    float v = 0.f;
    if (sel) {
      if constexpr (IOU) v = iou_elem(p.a[2 * i], p.a[2 * i + 1], p.b[2 * i], p.b[2 * i + 1], p.kind, p.eps);
      else v = focal_elem(p.a[i], p.b[i], p.alpha, p.gamma, p.smoothing != 0);
      acc += v;
      ++cnt;
    }
    if (p.elem) p.elem[i] = v;
  }
  acc = wave_sum(acc);
  const float c = wave_sum((float)cnt);                            // <= 4096 per block: exact in fp32
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = acc; s_cnt[threadIdx.x >> 6] = (int)c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    int n = 0;
    for (int w = 0; w < LOSS_NT / 64; ++w) { s += s_sum[w]; n += s_cnt[w]; }
    p.part_sum[blockIdx.x] = s;
    p.part_cnt[blockIdx.x] = n;
  }
}

// one workgroup: thread t adds the partials t, t + NT, ... in order, then the threads are added in order
__global__ __launch_bounds__(LOSS_NT) void k_loss_final(const float* __restrict__ part_sum, const int* __restrict__ part_cnt, int blocks,
                                                        float* __restrict__ sum_out, int* __restrict__ count_out) {
  __shared__ float s_sum[LOSS_NT];
  __shared__ int s_cnt[LOSS_NT];
  float s = 0.f;
  int n = 0;
  for (int i = threadIdx.x; i < blocks; i += LOSS_NT) { s += part_sum[i]; n += part_cnt[i]; }
  s_sum[threadIdx.x] = s;
  s_cnt[threadIdx.x] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    float ts = 0.f;
    int tn = 0;
    for (int i = 0; i < LOSS_NT; ++i) { ts += s_sum[i]; tn += s_cnt[i]; }
    if (sum_out) *sum_out = ts;
    if (count_out) *count_out = tn;
  }
}

template <bool IOU>
static int run_loss(LossArgs a, float* sum_out, int32_t* count_out, hipStream_t st) {
  if (a.n <= 0) {
    if (sum_out) DCF_HIP(hipMemsetAsync(sum_out, 0, sizeof(float), st));
    if (count_out) DCF_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t), st));
    return 0;
  }
  const int blocks = (int)((a.n + LOSS_PER_BLOCK - 1) / LOSS_PER_BLOCK);
  char* scratch = nullptr;
  DCF_HIP(hipMallocAsync((void**)&scratch, (size_t)blocks * (sizeof(float) + sizeof(int)), st));
  a.part_sum = reinterpret_cast<float*>(scratch);
  a.part_cnt = reinterpret_cast<int*>(scratch + (size_t)blocks * sizeof(float));
  ProfScope prof(IOU ? "ctr_iou_loss" : "sigmoid_focal_loss", st, 0.0, (IOU ? 16.0 : 8.0) * (double)a.n);
  hipLaunchKernelGGL(k_loss_partial<IOU>, dim3(blocks), dim3(LOSS_NT), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && (sum_out || count_out)) {
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(LOSS_NT), 0, st, (const float*)a.part_sum, (const int*)a.part_cnt, blocks, sum_out, (int*)count_out);
    e = hipGetLastError();
  }
  DCF_HIP(hipFreeAsync(scratch, st));
  DCF_HIP(e);
  return 0;
}

}  // namespace dcf

extern "C" {

int dcf_sigmoid_focal_loss(const float* inputs, const float* targets, const uint8_t* select, int64_t n, float alpha, float gamma,
                           int32_t smoothing, float* elem_out, float* sum_out, int32_t* count_out, void* stream) {
  DCF_CHECK(n >= 0 && (n == 0 || (inputs && targets)) && (elem_out || sum_out || count_out), "dcf_sigmoid_focal_loss: bad arguments");
  dcf::LossArgs a{inputs, targets, select, (long long)n, alpha, gamma, 0.f, smoothing, 0, elem_out, nullptr, nullptr};
  return dcf::run_loss<false>(a, sum_out, count_out, (hipStream_t)stream);
}

int dcf_ctr_iou_loss(const float* input_offsets, const float* target_offsets, const uint8_t* select, int64_t n, int32_t kind,
                     float eps, float* elem_out, float* sum_out, int32_t* count_out, void* stream) {
  DCF_CHECK(n >= 0 && (n == 0 || (input_offsets && target_offsets)) && (kind == 0 || kind == 1) && (elem_out || sum_out || count_out),
            "dcf_ctr_iou_loss: bad arguments");
  dcf::LossArgs a{input_offsets, target_offsets, select, (long long)n, 0.f, 0.f, eps, 0, kind, elem_out, nullptr, nullptr};
  return dcf::run_loss<true>(a, sum_out, count_out, (hipStream_t)stream);
}

}  // extern "C"
