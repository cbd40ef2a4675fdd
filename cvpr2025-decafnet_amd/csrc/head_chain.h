// A whole prediction head as ONE kernel (libs/modeling/head.py:53-64 ClsHead, :95-103 RegHead): two trunk layers
// MaskedConv1D(k3) -> LayerNorm -> ReLU and the k3 output convolution, over every row of the pyramid; f16x3 operand split.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "heads.h"

namespace dcf {

struct HeadChainArgs {
  const float* X;               // [rows][ldx] pyramid features, the head reads columns [0, C)
  int64_t ldx;
  const uint8_t* nbr;           // [rows] bit0 row usable, bit1 left neighbour usable, bit2 right (k_pyramid_masks)
  const unsigned short* W1c;    // chain image of trunk convolution 1 (launch_split_chain3)
  const unsigned short* W2c;    // ... of trunk convolution 2
  const float* ln1_w; const float* ln1_b; const float* ln2_w; const float* ln2_b;   // [C]
  const float* Wout;            // [NO][3][C] output convolution
  const float* bout;            // [NO]
  const LevelTable* lt;         // device copy: level starts, per-level scale, query-major offsets
  float* out;                   // query_major: out[(b * S + off_l + t) * NO + o], else out[row * NO + o]
  int rows, NO;
  int mode;                     // 0: raw logits   1: relu(scale_l * y)   (ConvOutArgs::mode)
  int query_major;
  unsigned* status;             // sticky numerics word (GemmArgs::status)
};

// C = 256 or 288, NO = 1 or 2; every launch covers rows [0, rows) of the pyramid; count = 1, or 2 heads of the same width on the
// same rows in one grid
bool head_chain_supports(int C, int NO);
int launch_head_chain(const HeadChainArgs* a, int count, int C, hipStream_t stream);
// halfs in the chain image of a (C, 3, C) k3 weight
size_t head_chain_image_halfs(int C);
// Wp: the packed fp32 weight [C out][3 taps][C in] (engine.hip pack3) -> img: fp16 hi / lo fragments in the order the kernel
// streams them; overflow (optional): bit 0 set if a weight leaves the scaled fp16 range
int launch_split_chain3(const float* Wp, unsigned short* img, int C, hipStream_t stream, unsigned* overflow = nullptr);

}  // namespace dcf
