// The attention half of a TransformerDecoder (fusion layer) as ONE kernel, libs/modeling/blocks.py:632-646 with
// ConvXAttNLayer (:513-520) and the global branch of MaskedMHA (:348-356, :374-393):
//   q   = x * mask
//   qc  = q_norm(dwconv3(ln_xattn_q(q) * mask))            query side of the cross attention
//   Q   = query(qc);  ctx = softmax(Q K^T / sqrt(d)) V     K, V: the projected text of the row's query (<= 64 tokens)
//   h   = proj(ctx) = (scale, shift);  q3 = adaln(q) * scale + shift        ('affine': q itself instead of adaln(q))
// and the (sum, sum of squares) of every q3 row for the LayerNorm that the FFN kernel folds (GemmArgs::stats_in).
// E = 256, 4 heads of 64 channels, f16x3 operand split.  Replaces k_dec_pre + the query GEMM + k_xattn_mfma + the AdaLN GEMM:
// one read and one write of a row instead of five and five.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

struct DecChainArgs {
  const float* X;               // [B*T][ldx] the decoder's input rows (raw stream)
  int64_t ldx;
  const uint8_t* mask;          // [B*T] row validity
  const float* ln_q_w; const float* ln_q_b;      // ln_xattn_q [256]
  const float* dw;              // depthwise k3 weight [3][256] (engine.hip pack3)
  const float* qn_w; const float* qn_b;          // xattn.q_norm [256]
  const unsigned short* Wq;     // chain image of xattn.xattn.query.weight (256 x 256), launch_split_chain1
  const float* bq;              // [256]
  const unsigned short* KV;     // [B] fragment images of the projected text keys / values (launch_kv_image)
  const float* kmask;           // [B][64] additive key mask: 0 valid, -inf masked or beyond the text
  const unsigned short* Wp;     // chain image of xattn.xattn.proj.weight with rows in (32 scale, 32 shift) blocks (512 x 256)
  const float* bp;              // [512] its bias in the same row order
  float* Q3;                    // [B*T][ldq] out: adaln(q) * scale + shift
  int64_t ldq;
  float* stats_out;             // [B*T][256 / stats_w] float2 (sum, sum of squares) of the Q3 rows (slot 0 carries the row)
  int stats_w;
  int B, T;
  int affine;                   // fusion.xattn_mode == 'affine' (blocks.py:623-626)
  int lk2;                      // 32-key tiles of the text: 1 (Lk <= 32) or 2 (Lk <= 64)
  unsigned* status;             // sticky numerics word (GemmArgs::status)
  int attn_single;              // dcf_config::attn_mode 1: S^T = K Q^T and O^T = V^T P^T as one fp16 product each (hi planes)
};

// true if the kernel covers this decoder shape
bool dec_chain_supports(int E, int heads, int Lk);
int launch_dec_chain(const DecChainArgs& a, hipStream_t stream);

// W [N][256] fp32 -> fp16 hi / lo A fragments in chain order (K step kk, lane (h, r), half j: W[32 n32 + r][16 kk + 8 (j >> 2) + 4 h
// + (j & 3)] * 2^8); N a multiple of 32.  overflow (optional): bit 0 set if a weight leaves the scaled fp16 range.
size_t chain1_image_halfs(int N, int K);
int launch_split_chain1(const float* W, unsigned short* img, int N, int K, hipStream_t stream, unsigned* overflow = nullptr);

// K, V [B*Lk][256] fp32 (projected text, zero padded to Lk rows per query), kvmask [B*Lk] -> per query the A fragments of
// S^T = K Q^T (K pre-scaled by d^-1/4) and O^T = V^T P^T as two unscaled fp16 planes, and the additive key mask.
size_t kv_image_halfs(int lk2);          // per query
int launch_kv_image(const float* K, const float* V, const uint8_t* kvmask, int B, int Lk, int lk2, unsigned short* img, float* kmask,
                    hipStream_t stream);

}  // namespace dcf
