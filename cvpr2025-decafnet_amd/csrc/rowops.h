// Launchers of the position-wise kernels (rowops.hip).  Token-major activations, see common.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

constexpr int DCF_MAX_BATCH = 64;   // queries processed together in one batched forward

struct LnArgs {
  const float* X; int64_t ldx;
  float* Y; int64_t ldy;
  const float* w; const float* b;   // affine (C) or nullptr
  int rows, C;
  int relu;                         // apply ReLU after LN
  int skip_ln;                      // 1: no normalisation at all (used for a bare '+ pe * mask')
  const float* pe;                  // optional (T, C) position encoding added where mask != 0
  const uint8_t* mask;              // [rows] (needed with pe)
  int T;                            // rows per batch element (pe index = row % T)
};

struct DecPreArgs {
  const float* X; int64_t ldx;      // [B*T][ldx]
  const uint8_t* mask;              // [B*T]
  const float* ln_q_w; const float* ln_q_b;
  const float* dw;                  // depthwise k3 weight repacked [3][C]
  const float* qn_w; const float* qn_b;
  float* Qc; float* Xa;             // [B*T][C]
  int B, T, C;
  int strip;                        // rows per wavefront (filled in by the launcher)
  const uint8_t* nbr;               // optional [B*T] neighbour flags: several sequences laid back to back (pyramid)
  int affine;                       // fusion.xattn_mode == 'affine' (blocks.py:623-626): Xa = q * m, no LayerNorm
};

struct EncPreArgs {
  const float* X; int64_t ldx;      // [B*T_in][ldx]
  const uint8_t* mask_in;           // [B*T_in]
  const float* ln_w; const float* ln_b;
  const float* dw_q; const float* dw_k; const float* dw_v;     // [3][C] each
  const float* qn_w; const float* qn_b; const float* kn_w; const float* kn_b; const float* vn_w; const float* vn_b;
  float* Qc; float* Kc; float* Vc;  // [B*T_out][C]
  float* Skip;                      // [B*T_out][C] (stride 2 only)
  int B, T_in, C;
  int strip;                        // output rows per wavefront (filled in by the launcher)
};

struct TextMeta {                   // passed BY VALUE inside TextLnArgs (kernel argument, 208 bytes at 8 queries)
  const float* text[DCF_MAX_BATCH];        // (TE, len) channel-major per query
  const uint8_t* text_mask[DCF_MAX_BATCH]; // (len) or nullptr
  int len[DCF_MAX_BATCH];
};

// text_net front end (text_net.py:158-181): tokens (C_t, Lq) channel-major -> rows [Lk][TE] token-major
struct TextEmbedArgs {
  const float* tokens;              // (C_t, Lq)
  const uint8_t* mask;              // (Lq) or nullptr (= all valid)
  const float* W; const float* bias;  // embd_fc (TE, C_t), (TE)
  const float* pe;                  // (>= Lq, TE) token-major or nullptr
  const float* bkgd;                // (TE) or nullptr
  float* X;                         // [Lk][TE]
  uint8_t* mask_out;                // [Lk]
  int Ct, Lq, TE;
  int pool;                         // TextIdentity + AttNPool1D: row 0 = masked mean of the embedded tokens (blocks.py:405)
};
int launch_text_embed(const TextEmbedArgs& a, hipStream_t st);
int launch_mask_rows(float* X, const uint8_t* mask, int rows, int C, hipStream_t st);                 // X[r][:] *= mask[r]
int launch_rows_to_chanmajor(const float* X, float* out, int rows, int C, hipStream_t st);           // out[c][r] = X[r][c]

struct TextLnArgs {
  TextMeta meta;
  float* out;                       // [B*Lkmax][TE]
  uint8_t* kvmask;                  // [B*Lkmax] or nullptr
  const float* w; const float* b;   // ln_xattn_kv affine (TE)
  int Lkmax, TE;
};

int launch_rowflags(const uint8_t* mask, uint8_t* nbr, int T, int rows, hipStream_t st);
int launch_pyramid_masks(uint8_t* mask_all, uint8_t* nbr_all, int B, int T0, int L, int rows_all, hipStream_t st);
int launch_mask_down(const uint8_t* in, uint8_t* out, int rows_out, hipStream_t st);
// vid_net.stride > 1: the row matrix [B*T/2][5 C] of a k5 / stride 2 / padding 2 MaskedConv1D's masked input (video_net.py:62-70)
int launch_im2col5s2(const float* X, int64_t ldx, const uint8_t* mask, float* col, int B, int T, int C, hipStream_t st);
// vid_net.pool_only: depthwise k3 MaskedConv1D, stride 1 / 2, w [3][C], output not masked (video_net.py:107-109, blocks.py:99)
int launch_dwconv3(const float* X, int64_t ldx, const uint8_t* mask, const float* w, float* Y, int64_t ldy, int B, int T, int stride,
                   int C, hipStream_t st);
int launch_vidmap_combine(const float* P1, const float* P2, const float* bias, const float* gate, const uint8_t* mask,
                          const float* w3, const float* correl, float* X, int T, int rows, int E, unsigned long long vmap,
                          hipStream_t st);
int launch_ln(const LnArgs& a, hipStream_t st);
int launch_dec_pre(const DecPreArgs& a, hipStream_t st);
int launch_dec_mid(const float* Xa, const float* H, const float* ln_w, const float* ln_b, float* Q3, float* Xn, int rows,
                   int C, hipStream_t st);
int launch_enc_pre(const EncPreArgs& a, int stride, hipStream_t st);
int launch_text_ln(const TextLnArgs& a, int B, hipStream_t st);

}  // namespace dcf
