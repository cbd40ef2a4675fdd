// Sidekick scoring and block top-k gate (score.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

constexpr int SCORE_SLICES = 32;   // channel slices of the partial-sum pass
constexpr int SCORE_MAXQ = 8;      // queries per scoring launch

constexpr int SCORE_MAXVID = 16;   // videos per scoring launch (= videos per forward)

// Scores of the queries of up to SCORE_MAXVID videos in one pass (three launches in all).  Video v owns rows
// qoff[v] .. qoff[v] + nq[v] - 1 of tn and correl and rows qoff[v] + v .. (its sum of squares, then its queries) of every
// slice of partial.
struct ScoreArgs {
  const float* shallow[SCORE_MAXVID];   // (D, T) channel-major -- the reference layout of shallow_vid[0]
  const float* text_cls[SCORE_MAXVID];  // (nq[v], D)
  int nq[SCORE_MAXVID], qoff[SCORE_MAXVID];
  int nvid;
  float* tn;              // (NQ, D) scratch: normalised text_cls, NQ = sum of nq[v]
  float* partial;         // [SCORE_SLICES][NQ + nvid][T] scratch
  float* correl;          // (NQ, T) out
  int D, T, NQ, norm;
};
int launch_sidekick(const ScoreArgs& a, hipStream_t st);
int launch_text_cls_norm(const ScoreArgs& a, hipStream_t st);     // tn only (the scores come out of the vid_map GEMMs)
// one video
inline ScoreArgs score_args(const float* shallow, const float* text_cls, float* tn, float* partial, float* correl, int D, int T,
                            int nq, int norm) {
  ScoreArgs a{};
  a.shallow[0] = shallow; a.text_cls[0] = text_cls; a.nq[0] = nq; a.qoff[0] = 0; a.nvid = 1;
  a.tn = tn; a.partial = partial; a.correl = correl; a.D = D; a.T = T; a.NQ = nq; a.norm = norm;
  return a;
}

struct GateArgs {
  const float* correl;      // (NQ, T), row q0 + b is used for batch element b
  const uint8_t* vid_mask;  // (nvid, T) validity of each clip (a prefix); batch element b uses row (vmap >> 4b) & 15
  float* gate;              // [B*T] out: 0/1 weight per clip
  uint8_t* mask_out;        // [B*T] out: vid_mask (msf) or vid_mask & gate (no msf, model.py:544-545)
  int T, B, q0, sn, msf;
  double sratio;
  unsigned long long vmap;  // video index of batch element b in nibble b (0 for one video)
  uint8_t* tile_flags;      // optional [B][nflags]: flag t / 64 of batch element b = its gate keeps a clip of 64 (t / 64) .. + 63
  int nflags;               // (T + 63) / 64
};
int launch_gate(const GateArgs& a, hipStream_t st);

}  // namespace dcf
