// The attention half of a TransformerEncoder (vid_net layer, libs/modeling/blocks.py:578-586 with ConvAttNLayer :462-473 and the
// local branch of MaskedMHA :357-373) as two kernels, E = 256, 4 heads of 64 channels, window <= 9, f16x3 operand split:
//
//   k_enc_qkv   (stride 1) x -> ln_attn -> three depthwise k3 convolutions (lane shifts) -> q / k / v_norm -> query / key / value
//               projections, chained on chip (enc_chain.hip); writes Q, K, V.  Replaces k_enc_pre + the grouped q / k / v GEMM: the
//               three normalised conv outputs (3 rows written, 3 read back) never reach memory.
//   k_enc_attn  Q, K, V -> sliding-window attention on the matrix cores (S^T = K Q^T over the wave's own 32 keys + an 8-key halo
//               tile, band mask + softmax in registers, O^T = V^T P^T) -> attn.proj -> x' = skip * mask + ls_attn * (.) and the row
//               statistics of x' for the folded ln_ffn.  Replaces k_local_attn + the projection GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

struct EncQkvArgs {
  const float* X;               // [B*T_in][ldx] the layer's input rows
  int64_t ldx;
  const uint8_t* mask_in;       // [B*T_in]
  const float* ln_w; const float* ln_b;          // ln_attn [256]
  const float* dw[3];           // depthwise k3 weights of q / k / v_conv, [3][256] each (engine.hip pack3)
  // q / k / v_norm folded into the projections (engine.hip fold_ln): W' = W diag(g), s[n] = sum_k W'[n][k], c[n] = b[n] + sum_k beta[k] W[n][k]
  const unsigned short* W[3];   // chain images of the folded attn.attn.query / key / value.weight (256 x 256), launch_split_chain1
  const float* fs[3];           // [256] s
  const float* fc[3];           // [256] c
  float* out[3];                // Q, K, V [B*T_in][256]
  int B, T_in;
  unsigned* status;             // sticky numerics word (GemmArgs::status)
};

bool enc_chain_supports(int E, int heads, int win);
int launch_enc_qkv(const EncQkvArgs& a, hipStream_t stream);

struct EncAttnArgs {
  const float* Q; const float* K; const float* V;    // [B*T][256] projected rows
  const uint8_t* mask;          // [B*T] row validity at this level
  const unsigned short* Wp;     // chain image of attn.attn.proj.weight (256 x 256), launch_split_chain1
  const float* bp;              // [256]
  const float* ls;              // [256] drop_path_attn.scale (LayerScale)
  const float* R;               // [B*T][ldr] skip rows: the layer's input (stride 1) or its masked max-pool (stride 2)
  int64_t ldr;
  float* Y;                     // [B*T][ldy] out: x' = R * mask + ls * (proj(ctx) + bp)       (blocks.py:586)
  int64_t ldy;
  float* stats_out;             // optional [B*T][256 / stats_w] float2 (sum, sum of squares) of the Y rows (slot 0 carries the row)
  int stats_w;
  int B, T, win;                // window size (odd, <= 9)
  unsigned* status;
  int attn_single;              // dcf_config::attn_mode 1: the window attention's two products as one fp16 product each (hi planes)
};
int launch_enc_attn(const EncAttnArgs& a, hipStream_t stream);

}  // namespace dcf
