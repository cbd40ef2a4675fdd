// fp32 MFMA GEMM used for every dense 1x1 / k3 convolution of the grounding path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

enum GemmAMode {
  A_ROWS = 0,      // A[m][k] = X[m*lda + k]                       (token-major activation)
  A_ROWS_TAP3 = 1, // A[m][tap*cin + c] = X[(m+tap-1)*lda + c] if neighbour usable else 0  (k3 conv, pad 1)
  A_CHANMAJOR = 2, // A[m][k] = X[k*lda + m]                       (reference (C,T) input layout)
};

enum GemmFlags {
  G_GELU = 1,       // exact erf GELU on (acc + bias)
  G_RELU = 2,
  G_RES = 4,        // C = R * (G_RES_MASK ? mask : 1) + ls[n] * h,   h = (acc + bias) * (G_OUT_MASK ? mask : 1)
  G_RES_MASK = 8,
  G_OUT_MASK = 16,
  G_AMASK = 32,     // A_ROWS: multiply A rows by rowmask on load (MaskedConv1D's x * mask)
  G_TAPSLAB = 128,  // A_ROWS_TAP3, split kernels: walk K channel slab by channel slab, taps innermost (set by launch_gemm_split)
  G_ADALN = 64,     // the N output columns are blocks of (32 scale columns, 32 shift columns of the same channels) -- weight
                    // rows ordered like that by the caller: C[m][c] = R[m][c] * scale[m][c] + shift[m][c], the AdaLN modulation
                    // of the fusion decoder (blocks.py:643-646); C and R have N / 2 columns.  Split kernel, tiles whose waves
                    // span 64 columns only (gemm_can_fuse_adaln); no other epilogue flag
};

struct GemmArgs {
  const float* A;
  int64_t lda;
  const float* W;      // [N][ldw], K contiguous (PyTorch Conv1d weight (N, K, 1); k3 weights are repacked [N][tap][cin])
  int64_t ldw;         // row pitch of W in floats (0 = K)
  const unsigned short* Ws;  // optional: W split into pre-tiled bf16 planes (gemm_bf16s.hip); nullptr = fp32 MFMA
  const float* bias;   // [N] or nullptr
  float* C;
  int64_t ldc;
  int M, N, K;
  int cin;                 // A_ROWS_TAP3: channels per tap
  const uint8_t* rowmask;  // [M] validity of row m (G_AMASK / G_RES_MASK / G_OUT_MASK)
  const uint8_t* nbr;      // [M] A_ROWS_TAP3: bit0 self usable, bit1 left usable, bit2 right usable
  const float* R;          // residual [M][ldr]
  int64_t ldr;
  const float* ls;         // [N] layer scale (G_RES); nullptr = 1
  int flags;
  // Optional fused channel LayerNorm of the output row (bf16-split kernel only, tile must span all N columns, see
  // gemm_can_fuse_ln): Y = LN(v) * ln_w + ln_b [ReLU] [+ ln_pe[row % ln_T] * rowmask[row]], v = the epilogue value
  // that goes to C.  C may be nullptr when only the normalised rows are needed.
  const float* ln_w;       // [N] or nullptr = no fused LayerNorm
  const float* ln_b;       // [N]
  float* Y;                // [M][ldy]
  int64_t ldy;
  int ln_relu;
  const float* ln_pe;      // optional (ln_T, N) position encoding added where ln_mask != 0
  const uint8_t* ln_mask;  // [M] (with ln_pe)
  int ln_T;
  // Row statistics carried between GEMMs instead of a LayerNorm pass (split tile kernels with the 16-byte epilogue only,
  // gemm_can_carry_stats).  Producer: stats_out [M][n_out / stats_w] float2 receives, per output row and per block of
  // stats_w columns, the partial (sum, sum of squares) of the values written to C (a workgroup tile fills the first slot it
  // covers and zeroes the others, so a consumer adds up all slots of a row whatever tile produced them).  Consumer: A holds
  // the RAW rows x, W is the weight with the LayerNorm gain folded in (W'[n][k] = W[n][k] g[k]), ln_s[n] = sum_k W'[n][k],
  // bias[n] = b[n] + sum_k beta[k] W[n][k]; with (mean, rstd) of row m from stats_in (stats_slots slots, K channels)
  //   acc'[m][n] = rstd[m] (acc[m][n] - mean[m] ln_s[n])        ( = sum_k LN(x)[m][k] g[k] W[n][k] without the beta term)
  // replaces acc before bias / activation / residual: LayerNorm(x) W^T + b without reading or writing LayerNorm(x).
  float* stats_out;
  int stats_w;             // columns per statistics slot (64; a tile's BN must be a multiple)
  const float* stats_in;   // [M][stats_slots] float2
  const float* ln_s;       // [N]
  int stats_slots;
  // A_ROWS_TAP3 consumer of a LayerNorm + ReLU (head / embedding trunks: conv -> LN -> ReLU -> conv, head.py:56-58): A holds the RAW
  // output rows of the previous convolution, a_stats [M][a_stats_slots] float2 their (sum, sum of squares) slots written by
  // that GEMM's stats_out, a_ln_g / a_ln_b [cin] the LayerNorm parameters; the rows are normalised, scaled and rectified while
  // they are staged into LDS (3 vector operations per element before the operand split), so the LayerNorm pass over the rows
  // -- a read and a write of every row -- does not exist.  The split 128x256 tile kernel only (gemm_can_norm_a).
  const float* a_stats;
  int a_stats_slots;
  const float* a_ln_g;
  const float* a_ln_b;
  // A_CHANMAJOR, split tile kernels only: the sidekick scores of the clips (model.py:500-505) computed on the side while the A tile
  // of the shallow features is staged (the vid_map product W[:, D:] . shallow streams exactly the (D, T) matrix the scores need):
  //   score_out[q][m] = sum_k A[m][k] score_tn[q][k] / (||A[m][:]|| + 1e-4)       (score_norm; the plain dot product without)
  // score_tn = the (already normalised) text vectors [score_nq][K], score_nq <= GEMM_SCORE_MAXQ.  Written by the workgroups of
  // the first column tile, sums in a fixed order (deterministic: the gate is a discrete decision).  nullptr = off.
  const float* score_tn;
  float* score_out;
  int score_nq, score_norm;
  // A_CHANMAJOR, split tile kernels only: row tiles the caller does not need (the expert half of vid_map under the top-k gate,
  // model.py:531-543: `vid * all_weight` is zero outside the selected blocks).  tile_skip[q * skip_stride + t / 64] != 0 says that
  // query q of this video keeps at least one of the clips 64 (t / 64) .. + 63 (written by k_gate, GateArgs::tile_flags); a
  // workgroup whose row tile is kept by NONE of the skip_nq queries returns before its first load and leaves its tile of C
  // unwritten -- the consumer must not read rows whose gate is 0 (k_vidmap_combine does not).  nullptr = every tile.
  const uint8_t* tile_skip;
  int skip_nq, skip_stride;
  // f16x3 mode: sticky device word, bit 0 is set when an accumulator leaves the finite range (an operand overflowed the
  // fp16 range, or the inputs already held inf / NaN); nullptr = not reported
  unsigned* status;
  // f16x3 mode: power-of-two pre-scale of the A operand (0 = default 2^4: |a| < 4094, small values exact down to 2^-7);
  // 1 widens the range to |a| < 65504 at an absolute representation floor of 2^-25
  float a_scale;
};

constexpr int GEMM_ZMAX = 16;          // operand sets of one channel-major GEMM in one grid (launch_gemm_split_z)
constexpr int GEMM_SCORE_MAXQ = 4;     // queries whose sidekick scores one channel-major GEMM carries (GemmArgs::score_out)

// Launch up to 3 independent GEMMs of identical (M, N, K, mode) in one grid (blockIdx.z).
int launch_gemm(const GemmArgs* g, int count, GemmAMode mode, hipStream_t stream);

// fp32-accurate GEMM on the 16-bit matrix cores by operand splitting (gemm_bf16s.hip); nterms = 16 (f16x3: two fp16
// planes, 3 products) or 6 (bf16x6: three bf16 planes, 6 products).  The weight image must have been made for the same mode.
constexpr int GEMM_F16X3 = 16, GEMM_BF16X6 = 6;
int launch_gemm_split(const GemmArgs* g, int count, GemmAMode mode, int nterms, hipStream_t stream);
// one channel-major GEMM (base: W / Ws, shape, lda, ldc, skip_stride) over nz (A, C) pairs in one grid, each with its own row-tile
// selection skip[z] / skip_nq[z] (GemmArgs::tile_skip; skip or skip[z] null = every tile); 64 x 256 tiles, N % 256 == 0, nz <= 16
// work_fraction: what the per-launch profile counts as this launch's algorithmic work (a lower bound of the share of row tiles that run)
int launch_gemm_split_z(const GemmArgs& base, int nz, const float* const* A, float* const* C, const uint8_t* const* skip, const int* skip_nq,
                        int nterms, hipStream_t stream, double work_fraction = 1.0);
// overflow: optional device word, bit 0 set if a weight does not fit the scaled fp16 range (f16x3 only)
int launch_split_planes(const float* W, unsigned short* out, int N, int K, int64_t ldw, hipStream_t st, int nterms = GEMM_BF16X6,
                        unsigned* overflow = nullptr);
// true if launch_gemm_split can run g with its LayerNorm fused (one tile spans all N columns and the grid still
// fills the chip); otherwise the caller launches the LayerNorm kernel itself
bool gemm_can_fuse_ln(int M, int N, int K, GemmAMode mode);
// true if launch_gemm_split runs an A_ROWS GEMM of this shape with a tile kernel that implements G_ADALN (not the k-sliced
// kernel of the small grids)
bool gemm_can_fuse_adaln(int M, int N, int K);
// true if launch_gemm_split runs `count` A_ROWS GEMMs of this shape in mode `nterms` with a tile kernel whose epilogue can
// write (stats_out) / consume (stats_in) row statistics: not the k-sliced kernel, tile width a multiple of 64 columns
bool gemm_can_carry_stats(int M, int N, int K, int count, int nterms);
// true if launch_gemm_split runs an A_ROWS_TAP3 GEMM of this shape on a tile kernel that can write row statistics (stats_out)
// and apply a LayerNorm + ReLU to its A rows while staging them (a_stats): the 128x256 tile (N % 256 == 0, M >= 65536);
// *stats_w receives the slot width the producer must use
bool gemm_can_norm_a(int M, int N, int K, int nterms, int* stats_w);

}  // namespace dcf
