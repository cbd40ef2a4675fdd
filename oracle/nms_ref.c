/*
 * CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of the
 * reference's 1-D NMS extension, libs/nms/src/nms_cpu.cpp.
 *
 *   dcf_oracle_nms_1d      follows nms_1d_cpu      (nms_cpu.cpp:20-63)
 *   dcf_oracle_softnms_1d  follows softnms_1d_cpu  (nms_cpu.cpp:72-172)
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks both functions against
 * tests/golden/nms_kat.npz (outputs of the reference extension compiled from
 * /root/reference by oracle/build_ref.py) and, when oracle/_ref/ is present, against
 * that extension directly on random inputs.
 *
 * One deliberate difference: the reference sorts with at::sort(descending=true), whose
 * order on tied scores is unspecified.  This oracle defines ties as "lowest original
 * index first" (a stable descending sort); fixtures are tie-free.
 *
 * Built by oracle/Makefile into oracle/libnms_oracle.so.  Only tests/, smoke() and the
 * cpu_baseline leg of bench.py may load it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float score; int64_t idx; } pair_t;

/* merge sort: stable, descending by score */
static void merge_sort_desc(pair_t *a, pair_t *tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort_desc(a, tmp, h);
    merge_sort_desc(a + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take right only if strictly greater -> stable */
        if (a[j].score > a[i].score) tmp[k++] = a[j++];
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(pair_t));
}

/* segs: (n,2) row-major fp32; scores: (n,) fp32; keep_out: room for n int64.
 * Returns the number of kept indices (original indices, in descending-score order). */
int64_t dcf_oracle_nms_1d(const float *segs, const float *scores, int64_t n,
                          float iou_thresh, int64_t *keep_out) {
    if (n <= 0) return 0;
    float *x1 = (float *)malloc((size_t)n * sizeof(float));
    float *x2 = (float *)malloc((size_t)n * sizeof(float));
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    pair_t *order = (pair_t *)malloc((size_t)n * sizeof(pair_t));
    pair_t *tmp = (pair_t *)malloc((size_t)n * sizeof(pair_t));
    unsigned char *alive = (unsigned char *)malloc((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        x1[i] = segs[2 * i];
        x2[i] = segs[2 * i + 1];
        /* Tensor - Tensor + double scalar stays fp32: (x2 - x1) + 1e-6f   (:31) */
        areas[i] = (x2[i] - x1[i]) + 1e-6f;
        order[i].score = scores[i];
        order[i].idx = i;
        alive[i] = 1;
    }
    merge_sort_desc(order, tmp, n);
    for (int64_t a = 0; a < n; a++) {
        if (!alive[a]) continue;
        int64_t i = order[a].idx;
        float ix1 = x1[i], ix2 = x2[i], iarea = areas[i];
        for (int64_t b = a + 1; b < n; b++) {
            if (!alive[b]) continue;
            int64_t j = order[b].idx;
            float xx1 = ix1 > x1[j] ? ix1 : x1[j];
            float xx2 = ix2 < x2[j] ? ix2 : x2[j];
            float inter = xx2 - xx1;
            if (!(inter > 0.f)) inter = 0.f;           /* std::max(0.f, xx2 - xx1) */
            float ovr = inter / (iarea + areas[j] - inter);
            if (ovr >= iou_thresh) alive[b] = 0;
        }
    }
    int64_t k = 0;
    for (int64_t a = 0; a < n; a++)
        if (alive[a]) keep_out[k++] = order[a].idx;
    free(x1); free(x2); free(areas); free(order); free(tmp); free(alive);
    return k;
}

/* dets: (n,3) out-param [x1, x2, score] per pick; inds_out: room for n int64.
 * method 0 vanilla, 1 linear, 2 gaussian.  Returns the surviving count. */
int64_t dcf_oracle_softnms_1d(const float *segs, const float *scores, int64_t n, float *dets,
                              float iou_thresh, float sigma, float min_score, int method,
                              int64_t *inds_out) {
    if (n <= 0) return 0;
    float *x1 = (float *)malloc((size_t)n * sizeof(float));
    float *x2 = (float *)malloc((size_t)n * sizeof(float));
    float *sc = (float *)malloc((size_t)n * sizeof(float));
    float *areas = (float *)malloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; i++) {
        x1[i] = segs[2 * i];
        x2[i] = segs[2 * i + 1];
        sc[i] = scores[i];
        areas[i] = (x2[i] - x1[i]) + 1e-6f;
        inds_out[i] = i;
    }
    int64_t nsegs = n;
    for (int64_t i = 0; i < nsegs; i++) {
        /* first maximum in [i, nsegs) wins (strict <)                      (:104-114) */
        float max_score = sc[i];
        int64_t max_pos = i;
        for (int64_t pos = i + 1; pos < nsegs; pos++) {
            if (max_score < sc[pos]) { max_score = sc[pos]; max_pos = pos; }
        }
        float ix1 = x1[max_pos], ix2 = x2[max_pos], iscore = sc[max_pos], iarea = areas[max_pos];
        int64_t iind = inds_out[max_pos];
        dets[i * 3 + 0] = ix1; dets[i * 3 + 1] = ix2; dets[i * 3 + 2] = iscore;
        x1[max_pos] = x1[i]; x2[max_pos] = x2[i]; sc[max_pos] = sc[i];
        areas[max_pos] = areas[i]; inds_out[max_pos] = inds_out[i];
        x1[i] = ix1; x2[i] = ix2; sc[i] = iscore; areas[i] = iarea; inds_out[i] = iind;

        int64_t pos = i + 1;
        while (pos < nsegs) {
            float xx1 = ix1 > x1[pos] ? ix1 : x1[pos];
            float xx2 = ix2 < x2[pos] ? ix2 : x2[pos];
            float inter = xx2 - xx1;
            if (!(inter > 0.f)) inter = 0.f;
            float ovr = inter / (iarea + areas[pos] - inter);
            float weight = 1.f;
            if (method == 0) {
                if (ovr >= iou_thresh) weight = 0.f;
            } else if (method == 1) {
                if (ovr >= iou_thresh) weight = 1.f - ovr;
            } else if (method == 2) {
                weight = expf(-(ovr * ovr) / sigma);
            }
            sc[pos] *= weight;
            if (sc[pos] < min_score) {                                   /* (:157-165) */
                x1[pos] = x1[nsegs - 1]; x2[pos] = x2[nsegs - 1]; sc[pos] = sc[nsegs - 1];
                areas[pos] = areas[nsegs - 1]; inds_out[pos] = inds_out[nsegs - 1];
                nsegs -= 1;
                pos -= 1;
            }
            pos += 1;
        }
    }
    free(x1); free(x2); free(sc); free(areas);
    return nsegs;
}
