// Sidekick scoring and block top-k gate (score.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

constexpr int SCORE_SLICES = 32;   // channel slices of the partial-sum pass
constexpr int SCORE_MAXQ = 8;      // queries per scoring launch

struct ScoreArgs {
  const float* shallow;   // (D, T) channel-major -- the reference layout of shallow_vid[0]
  const float* text_cls;  // (NQ, D)
  float* tn;              // (NQ, D) scratch: normalised text_cls
  float* partial;         // [SCORE_SLICES][NQ + 1][T] scratch
  float* correl;          // (NQ, T) out
  int D, T, NQ, norm;
};
int launch_sidekick(const ScoreArgs& a, hipStream_t st);

struct GateArgs {
  const float* correl;      // (NQ, T), row q0 + b is used for batch element b
  const uint8_t* vid_mask;  // (nvid, T) validity of each clip (a prefix); batch element b uses row (vmap >> 4b) & 15
  float* gate;              // [B*T] out: 0/1 weight per clip
  uint8_t* mask_out;        // [B*T] out: vid_mask (msf) or vid_mask & gate (no msf, model.py:544-545)
  int T, B, q0, sn, msf;
  double sratio;
  unsigned long long vmap;  // video index of batch element b in nibble b (0 for one video)
};
int launch_gate(const GateArgs& a, hipStream_t st);

}  // namespace dcf
