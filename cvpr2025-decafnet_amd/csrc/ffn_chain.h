// FFN of a transformer block as ONE kernel (libs/modeling/blocks.py:535-538: fc -> erf GELU -> proj, with the block's
// residual / LayerScale / mask epilogue of blocks.py:589-590), f16x3 operand split, E = 256 (hidden 1024).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

struct FfnChainArgs {
  const float* X;              // [M][ldx]: the rows fc consumes (RAW rows when stats != nullptr, LayerNorm output otherwise)
  int64_t ldx;
  const unsigned short* W1s;   // fragment image of the fc weight (1024 x 256), launch_split_planes in the f16x3 mode
  const float* b1;             // [1024] fc bias (the folded bias c when stats != nullptr, GemmArgs::stats_in)
  const float* ln_s;           // [1024] row sums of the folded weight (stats != nullptr)
  const float* stats;          // [M][stats_slots] float2 (sum, sum of squares) of the raw rows, or nullptr
  int stats_slots;
  const unsigned short* W2s;   // fragment image of the proj weight (256 x 1024)
  const float* b2;             // [256]
  const float* ls;             // [256] LayerScale or nullptr (= 1)
  const float* R;              // residual rows [M][ldr]
  int64_t ldr;
  const uint8_t* rowmask;      // [M] or nullptr: the FFN output is multiplied by it before the residual (G_OUT_MASK)
  float* C;                    // [M][ldc] = R + ls * ((gelu(X W1^T + b1) W2^T + b2) * rowmask)
  int64_t ldc;
  float* stats_out;            // optional [M][256 / stats_w] float2: (sum, sum of squares) of the rows written to C
  int stats_w;
  unsigned* status;            // sticky numerics word (GemmArgs::status)
  int M;
  int variant;                 // 0: the default kernel, 1: four waves (k_ffn_chain), 2: eight waves, producer / consumer pairs (k_ffn_pair)
};

// E = 256 only; rows in tiles of 128 (one workgroup per CU: eight waves as four producer / consumer pairs, 154 KiB of LDS,
// persistent over its row tiles; or the four-wave kernel, 136 KiB)
int launch_ffn_chain(const FfnChainArgs& a, hipStream_t stream);
// stats [rows][C / stats_w] float2 <- (sum, sum of squares) of every row of X in the slot layout of GemmArgs::stats_out
int launch_row_stats(const float* X, int64_t ldx, float* stats, int rows, int C, int stats_w, hipStream_t stream);

}  // namespace dcf
