// Proposal decoding + NMS launchers (postproc.hip).  Every problem is "one query"; arrays of
// several queries are laid out with a fixed per-query stride so one launch serves a batch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dcf {

struct CollectArgs {
  const float* logits;      // [nq][S]     raw logits, query-major pyramid order
  const float* offsets;     // [nq][S][2]
  const uint8_t* masks;     // [nq][S]
  uint32_t* keys;           // [nq][S] scratch
  const float* ext;         // [nq][T] external per-clip scores (worker_v2.py:1150-1156) or nullptr
  int T;                    // level-0 length (row pitch of ext)
  int S, n_levels;
  int off[17];              // off[l] = first point of level l inside a query (off[n_levels] = S)
  float pre_nms_thresh, seg_len_thresh;
  int pre_nms_topk;
  float* segs;              // [nq][pre_nms_topk][2] out
  float* scores;            // [nq][pre_nms_topk]    out
  int* counts;              // [nq] out
};
// does launch_collect need the CollectArgs::keys scratch (nq * S words) for S points per query?  (no: the keys stay in registers)
bool collect_needs_scratch(int S);
int launch_collect(const CollectArgs& a, int nq, hipStream_t st);

struct NmsArgs {
  const float* segs;        // [nq][stride][2]
  const float* scores;      // [nq][stride]
  const int* counts;        // [nq] valid entries per query (nullptr: n_max for all)
  int n_max, stride;
  float iou_thresh;
  long long* keep;          // [nq][stride] out: kept ORIGINAL indices, descending score
  int* keep_counts;         // [nq] out
};
int launch_nms(const NmsArgs& a, int nq, hipStream_t st);

struct SoftNmsArgs {
  const float* segs; const float* scores; const int* counts;
  int n_max, stride;
  float iou_thresh, sigma, min_score;
  int method;               // 0 vanilla, 1 linear, 2 gaussian
  int max_iters;            // 0 = run to completion (reference behaviour); k > 0 = stop after k picks
  float* dets;              // [nq][stride][3] out (x1, x2, score) per pick
  long long* inds;          // [nq][stride] out
  int* out_counts;          // [nq] out
};
int launch_softnms(const SoftNmsArgs& a, int nq, hipStream_t st);

struct VotingArgs {
  const float* nms_segs; int nms_ld;     // [nq][n1_stride][nms_ld] (first two columns are the segment)
  const int* n1_counts; int n1_max, n1_stride;
  const float* all_segs; const float* all_scores; const int* n2_counts; int n2_max, n2_stride;
  float iou_thresh;
  float* out;               // [nq][n1_stride][2]
};
int launch_voting(const VotingArgs& a, int nq, hipStream_t st);

}  // namespace dcf
