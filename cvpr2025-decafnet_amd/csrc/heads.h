// Output convolutions of the heads and the iterative-refinement TCN (heads.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

constexpr int DCF_MAX_LEVELS = 16;
constexpr int TCN_HID = 32;     // TCN(arch[-1], 32, 32, ...) is hard-wired in libs/modeling/model.py:424

// Pyramid geometry of one batched forward.  Rows of every pyramid-wide buffer are ordered
// [level l][batch b][position t]:  row = start[l] + b * T[l] + t.   Lives in device memory.
struct LevelTable {
  int n_levels, B;
  int S;                              // sum_l T[l]  (points per query)
  int T[DCF_MAX_LEVELS];
  int start[DCF_MAX_LEVELS + 1];      // start[l] = B * sum_{j<l} T[j]
  int off[DCF_MAX_LEVELS];            // off[l] = sum_{j<l} T[j]  (offset inside one query's output)
  float scale[DCF_MAX_LEVELS];        // reg_head.scales.{l}.scale
};

struct ConvOutArgs {
  const float* X; int64_t ldx;        // [rows][ldx] activations (already LN+ReLU'd)
  const uint8_t* nbr;                 // [rows] neighbour flags (MaskedConv1D masks its input)
  const float* W;                     // [NO][3][C]
  const float* bias;                  // [NO]
  const LevelTable* lt;
  float* out;
  int rows, C, NO;
  int row0;                           // pyramid row of X[0] (X, nbr and, for internal order, out are pre-offset by it)
  int mode;                           // 0: raw logits   1: relu(scale_l * y)  (RegHead, head.py:102-103)
  int query_major;                    // 0: out[row*NO+o] (internal order)   1: out[(b*S + off_l + t)*NO + o]
  const float* ln_w; const float* ln_b;   // optional: X holds the RAW output of the trunk's last convolution and the kernel applies
                                      // its LayerNorm + ReLU on load (one pass over X instead of a LayerNorm pass + this one)
};
int launch_conv_out(const ConvOutArgs& a, hipStream_t st);

struct RefineArgs {
  const float* logits1;               // [rows] first-pass logits, internal order
  const LevelTable* lt;
  const uint8_t* mask_all;            // [rows] per-level masks, internal order
  // TCN weights (repacked): in [L][32], dil[i] [3][32][32] (tap, ci, co), pw[i] [32][32] (ci, co), out [32][32]
  const float* w_in; const float* b_in;
  const float* const* host_w_dil; const float* const* host_b_dil;   // HOST arrays [n_layers] of device pointers
  const float* const* host_w_pw; const float* const* host_b_pw;
  const float* const* host_ln_w; const float* const* host_ln_b;
  const float* w_out; const float* b_out;
  const unsigned short* const* host_frag;   // optional HOST array [n_layers] of device pointers: the layers' MFMA fragment images
                                            // (launch_tcn_frag_image; the last layer's with w_out); null = built per workgroup
  float* bufA; float* bufB;           // [B*T0][32] ping-pong
  float* F; int64_t ldf; int E;       // pyramid feature buffer; refined logits go to columns [E, E+32)
  int B, T0, n_levels, n_layers;
  const float* stacked;               // optional [B*T0][n_levels] TCN input given directly (dcf_op_tcn); then logits1 / lt are
                                      // not read and nothing is pooled down a pyramid
  int f16;                            // 1: the layers run on the matrix cores in the f16x3 arithmetic of the dense convolutions
                                      // (weights range-checked with launch_f16_weight_range), 0: fp32 on the vector ALUs
  unsigned* status;                   // f16: sticky numerics word (bit 0 raised on a non-finite intermediate), may be null
  int stack_layers;                   // f16 with fragment images: how many leading TCN layers run as ONE launch over LDS windows (k_tcn_stack):
                                      // -1 = default (5: dilations 1 .. 16), 0 / 1 = none, at most 5 and never the last layer
};
int launch_refine(const RefineArgs& a, const LevelTable& host_lt, hipStream_t st);
// masked max-pool (k3, s2) of the 32 refined channels (columns [E, E + 32) of F) from the rows of one level to the next
int launch_refine_pool(float* F, int64_t ldf, int E, const uint8_t* mask_in, int64_t in_row0, int64_t out_row0, int B, int T_in,
                       hipStream_t st);
// raises *flag if a weight of the TCN does not fit the scaled fp16 range of the f16 mode (|w| < 255.9)
int launch_f16_weight_range(const float* w, int n, unsigned* flag, hipStream_t st);
// fp16 hi / lo MFMA fragments of one TCN layer's weights (dilated conv [3][32][32], conv_1x1 [32][32], optionally refine.conv_out
// [32][32] for the last layer) in the order k_tcn_layer_mfma consumes them: img [10][2][64][8] halfs (20 KiB)
constexpr int TCN_FRAG_HALFS = 10 * 2 * 64 * 8;
int launch_tcn_frag_image(const float* wd, const float* wp, const float* wo, unsigned short* img, hipStream_t st);

}  // namespace dcf
