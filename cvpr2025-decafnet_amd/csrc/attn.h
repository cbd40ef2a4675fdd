// Attention cores (attn.hip): text->clip cross attention and sliding-window clip self attention.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

struct XAttnArgs {
  const float* Q;          // [B*T][C]  projected queries (one row per clip)
  const float* K;          // [B*Lk][C] projected text keys   (zero padded to Lk)
  const float* V;          // [B*Lk][C]
  const uint8_t* kvmask;   // [B*Lk]    1 = valid key
  float* O;                // [B*T][C]  context, heads concatenated along C
  int B, T, Lk, C, heads;
  // sticky numerics word of the model (bit 0): the MFMA core carries q d^-1/4, k d^-1/4, v as two fp16 planes whatever
  // opt.model.gemm_mode is, so an operand beyond the fp16 range (|x| > 65504) or a non-finite one is reported here instead of
  // turning into inf / NaN context rows silently; nullptr = not reported
  unsigned* status;
  int single;              // dcf_config::attn_mode 1: one fp16 product per multiply-add (hi planes only) instead of the f16x3 triple
};

struct LocalAttnArgs {
  const float* Q; const float* K; const float* V;   // [B*T][C]
  const uint8_t* mask;                               // [B*T]
  float* O;                                          // [B*T][C]
  int B, T, C, heads, window;                        // window odd
};

// global clip self-attention (vid_net.mha_win_size = 0, blocks.py:339-356, :374-393): every query attends to every valid key of
// its sequence.  Not on the hot path of the published configurations (window 9 / 19): a plain fp32 online-softmax core.
struct GlobalAttnArgs {
  const float* Q; const float* K; const float* V;   // [B*T][C]
  const uint8_t* mask;                               // [B*T] key validity
  float* O;                                          // [B*T][C]
  int B, T, C, heads;
};

int launch_xattn(const XAttnArgs& a, hipStream_t st);
int launch_global_attn(const GlobalAttnArgs& a, hipStream_t st);
int launch_local_attn(const LocalAttnArgs& a, hipStream_t st);

}  // namespace dcf
