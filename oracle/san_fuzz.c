/* CPU ORACLE -- test infrastructure only.  Sanitizer harness for oracle/nms_ref.c: `make -C oracle sanitize` builds this file together with
 * nms_ref.c under -fsanitize=address,undefined and runs both entry points over random candidate sets of every small size (0, 1, 2, ...),
 * heavy overlaps, equal scores and scores below the pruning threshold -- the cases that move the swap-with-last pruning and the arg-max
 * scan to the ends of their arrays.  Exit code 0 = no report from either sanitizer and the invariants below hold.
 * (GPU sanitizers are not available on this pool; the C half of the oracle is what the HIP kernels' indices are compared with bit for bit.) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t dcf_oracle_nms_1d(const float *segs, const float *scores, int64_t n, float iou_thresh, int64_t *keep);
int64_t dcf_oracle_softnms_1d(const float *segs, const float *scores, int64_t n, float *dets, float iou_thresh, float sigma, float min_score,
                              int method, int64_t *inds);

static uint32_t rng = 12345u;
static float frand(void) { rng = rng * 1664525u + 1013904223u; return (float)(rng >> 8) / 16777216.0f; }

int main(void) {
  long checks = 0;
  for (int round = 0; round < 400; ++round) {
    const int64_t n = round < 40 ? round : 1 + (int64_t)(frand() * 700.0f);
    /* exact-size allocations: an off-by-one access is a sanitizer report */
    float *segs = malloc(sizeof(float) * 2 * (size_t)(n ? n : 1));
    float *scores = malloc(sizeof(float) * (size_t)(n ? n : 1));
    float *dets = malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    int64_t *idx = malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    const float span = (round % 3 == 0) ? 20.0f : 2000.0f;          /* every third round: everything overlaps everything */
    for (int64_t i = 0; i < n; ++i) {
      const float c = frand() * span, w = 1.0f + frand() * 40.0f;
      segs[2 * i] = c - w / 2; segs[2 * i + 1] = c + w / 2;
      scores[i] = (round % 5 == 0) ? 0.5f : frand();                /* every fifth round: all scores equal */
      if (round % 7 == 0 && (i & 3) == 0) scores[i] = 0.0005f;     /* below min_score from the start */
    }
    const float thr = 0.1f + 0.8f * frand();
    int64_t k = dcf_oracle_nms_1d(segs, scores, n, thr, idx);
    if (k < 0 || k > n) { fprintf(stderr, "nms: kept %lld of %lld\n", (long long)k, (long long)n); return 1; }
    for (int64_t j = 0; j < k; ++j)
      if (idx[j] < 0 || idx[j] >= n) { fprintf(stderr, "nms: index %lld out of range\n", (long long)idx[j]); return 1; }
    for (int method = 0; method < 3; ++method) {
      k = dcf_oracle_softnms_1d(segs, scores, n, dets, thr, 0.1f + frand(), 0.001f, method, idx);
      if (k < 0 || k > n) { fprintf(stderr, "softnms: ranked %lld of %lld\n", (long long)k, (long long)n); return 1; }
      for (int64_t j = 0; j < k; ++j) {
        if (idx[j] < 0 || idx[j] >= n) { fprintf(stderr, "softnms: index %lld out of range\n", (long long)idx[j]); return 1; }
        if (j > 0 && dets[3 * j + 2] > dets[3 * (j - 1) + 2]) { fprintf(stderr, "softnms: picked scores increase at %lld\n", (long long)j); return 1; }
      }
      ++checks;
    }
    free(segs); free(scores); free(dets); free(idx);
  }
  printf("san_fuzz ok: %ld soft-NMS runs, 400 NMS runs\n", checks);
  return 0;
}
