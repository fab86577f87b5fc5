/*
 * trex_split.c -- CPU restatement of TRex's threshold search that splits merged blobs.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows (reference = /root/reference, read-only):
 *   SplitBlob::apply_threshold            Application/src/tracker/tracking/SplitBlob.cpp:130-179
 *   SplitBlob::evaluate_result_multiple   :193-255
 *   Run<>::perform / check_viable_option  :258-337
 *   SplitBlob::split                      :419-800   (blob_split_algorithm threshold / threshold_approximate; the watershed
 *                                                      algorithm needs cv::watershed and is not restated)
 * Not in the tree (commons): pixel::threshold_blob(cache, blob, diff_px, threshold) -- restated as "keep a pixel iff its
 * difference value >= threshold" like the background overload pinned by Tests/test_pixels.cpp; Float2_t taken as float.
 *
 * Blobs of >= 16000 pixels are searched by 4 pool threads in the reference (:735-760).  With blob_split_algorithm = threshold
 * (`accurate`) every thread only stops at the first ABORT / KEEP_ABORT of its own arithmetic progression, ABORT is monotone in the
 * threshold, so the smallest KEEP_ABORT threshold is always visited: the result equals the sequential scan restated here.
 * With threshold_approximate the threaded result depends on thread timing; the sequential form (:719-726) is restated for all sizes.
 */
#include "trex_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { A_KEEP = 0, A_KEEP_ABORT = 1, A_REMOVE = 2, A_ABORT = 3, A_TOO_FEW = 4, A_SKIP = 5, A_NO_CHANCE = 6 };

typedef struct {
    const oracle_run* runs; int32_t n_runs; const uint8_t* pixels; const uint8_t* bg; int32_t bg_stride, width, height, method, connectivity;
    const oracle_split_params* P;
    int32_t presumed_nr;
    float sqrcm, first_size, max_size;
    int min_pixel, max_pixel;
    int best_threshold;              /* best_match.threshold (:542-548) */
    int n_best;                      /* number of blobs saved with it (after evaluate's removals) */
    double best_min_size;            /* the removal bound that was applied to them */
    int tried;
    uint8_t cache[512];              /* Run::results */
    int run_best;                    /* Run::best */
} split_state;

static int in_range_of_one(const oracle_split_params* P, float cmsq) {      /* core/SizeFilters.cpp:36-53, scale_factor -1 */
    if (P->n_ranges == 0) return 1;
    for (int i = 0; i < P->n_ranges; ++i) if ((double)cmsq >= P->ranges[2 * i] && (double)cmsq < P->ranges[2 * i + 1]) return 1;
    return 0;
}
static void max_range(const oracle_split_params* P, double* start, double* end) {   /* SizeFilters::add, :12-18 */
    *start = -1; *end = -1;
    for (int i = 0; i < P->n_ranges; ++i) {
        if (*start == -1 || P->ranges[2 * i] < *start) *start = P->ranges[2 * i];
        if (*end == -1 || P->ranges[2 * i + 1] > *end) *end = P->ranges[2 * i + 1];
    }
}

static int cmp_desc(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? 1 : (x > y ? -1 : 0);
}

/* try_threshold (:514-556): apply_threshold + evaluate_result_multiple; returns the action */
static int try_threshold(split_state* S, int threshold) {
    const oracle_split_params* P = S->P;
    const int initial = threshold == -1;
    if (initial) threshold = S->P->initial_threshold;
    /* apply_threshold (:130-179): the first call clamps the threshold to the smallest difference value of the blob */
    int applied = threshold;
    if (initial && applied < S->min_pixel) applied = S->min_pixel;
    oracle_frame* f = oracle_threshold_blob(S->runs, S->n_runs, S->pixels, S->bg, S->bg_stride, S->width, S->height, S->method, applied,
                                            S->connectivity);
    int32_t nb, nr, np;
    oracle_frame_counts(f, &nb, &nr, &np);
    oracle_blob* blobs = (oracle_blob*)malloc(sizeof(oracle_blob) * (size_t)(nb > 0 ? nb : 1));
    oracle_run* runs = (oracle_run*)malloc(sizeof(oracle_run) * (size_t)(nr > 0 ? nr : 1));
    uint8_t* px = (uint8_t*)malloc((size_t)(np > 0 ? np : 1));
    oracle_frame_copy(f, blobs, runs, px);
    oracle_frame_free(f);
    /* sorted by (num_pixels, blob_id) descending (:169-172); only the sizes enter the evaluation */
    uint64_t* key = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(nb > 0 ? nb : 1));
    for (int i = 0; i < nb; ++i) key[i] = ((uint64_t)blobs[i].n_pixels << 32) | blobs[i].bid;
    qsort(key, (size_t)nb, sizeof(uint64_t), cmp_desc);
    S->max_size = (float)(nb ? (uint32_t)(key[0] >> 32) : 0u) * S->sqrcm;
    ++S->tried;

    /* evaluate_result_multiple (:193-255) */
    int action;
    size_t pixels = 0;
    for (int i = 0; i < nb; ++i) pixels += (uint32_t)(key[i] >> 32);
    int n_kept = nb;
    double bound = 0;
    if ((float)pixels * S->sqrcm < P->blob_split_max_shrink * S->first_size) action = A_ABORT;
    else {
        double ms, me;
        max_range(P, &ms, &me);
        if (P->n_ranges) bound = ms * P->blob_split_global_shrink_limit;
        else bound = (double)((float)pixels * S->sqrcm * P->blob_split_max_shrink);
        n_kept = 0;
        for (int i = 0; i < nb; ++i) {
            const float fsize = (float)(uint32_t)(key[i] >> 32) * S->sqrcm;
            if (!((double)fsize < bound)) key[n_kept++] = key[i];
        }
        size_t valid = 0; int has_min = 0; size_t min_size = 0;
        for (size_t i = 0; i < (size_t)S->presumed_nr && i < (size_t)n_kept; ++i) {
            const size_t n = (uint32_t)(key[i] >> 32);
            if (!has_min || n < min_size) { min_size = n; has_min = 1; }
            if (in_range_of_one(P, (float)n * S->sqrcm)) ++valid;
        }
        if (P->n_ranges && has_min && (double)((float)min_size * S->sqrcm) > me) action = A_REMOVE;
        else if (valid < (size_t)S->presumed_nr) action = A_TOO_FEW;
        else action = A_KEEP_ABORT;
    }
    if (S->first_size == 0) S->first_size = S->max_size;
    if (action == A_KEEP || action == A_KEEP_ABORT) {
        if (S->best_threshold == -1 || threshold < S->best_threshold) {
            S->best_threshold = threshold; S->n_best = n_kept; S->best_min_size = bound;
        }
    }
    free(key); free(blobs); free(runs); free(px);
    return action;
}

/* Run<false>::perform (:318-337) */
static int perform(split_state* S, int threshold) {
    int action = threshold >= 0 && threshold < 512 ? S->cache[threshold] : A_NO_CHANCE;
    if (action == A_NO_CHANCE) action = try_threshold(S, threshold);
    if (action == A_KEEP || action == A_KEEP_ABORT) {
        if (S->run_best == -1 || threshold < S->run_best) { S->run_best = threshold; if (threshold >= 0 && threshold < 512) S->cache[threshold] = (uint8_t)action; }
    }
    return action;
}

/* work.operator()<false> (:609-706) */
static void work_approximate(split_state* S, int begin_threshold, int thread_index) {
    const int segments = 3, sampling_runs = 2, step = segments * sampling_runs;
    const int start = begin_threshold, end = S->max_pixel;
    const int fs_start = start, fs_end = start + (int)((end - start) * 0.3);
    for (int offset = 0; offset < sampling_runs; ++offset) {
        if (S->run_best != -1) break;
        for (int threshold = fs_start + thread_index * sampling_runs + offset; threshold < fs_end; threshold += step) {
            if (S->run_best != -1 && threshold >= S->run_best) break;
            const int action = perform(S, threshold);
            if (action == A_ABORT || action == A_KEEP_ABORT) { if (action == A_KEEP_ABORT) return; break; }
        }
    }
    if (S->run_best != -1) return;
    const int use_step = step / 2;
    for (int threshold = fs_end + thread_index; threshold < end; threshold += use_step) {
        if (S->run_best != -1 && threshold >= S->run_best) break;
        const int action = perform(S, threshold);
        if (action == A_ABORT || action == A_KEEP_ABORT) { if (action == A_KEEP_ABORT) return; break; }
    }
}

void oracle_split_search(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels, const uint8_t* bg, int32_t bg_stride,
                         int32_t width, int32_t height, int32_t method, int32_t connectivity, const oracle_split_params* P,
                         int32_t presumed_nr, oracle_split_info* out) {
    split_state S;
    memset(&S, 0, sizeof(S));
    S.runs = runs; S.n_runs = n_runs; S.pixels = pixels; S.bg = bg; S.bg_stride = bg_stride; S.width = width; S.height = height;
    S.method = method; S.connectivity = connectivity; S.P = P; S.presumed_nr = presumed_nr;
    S.sqrcm = P->cm_per_pixel * P->cm_per_pixel;
    S.best_threshold = -1; S.run_best = -1;
    memset(S.cache, A_NO_CHANCE, sizeof(S.cache));
    /* difference values of the blob's pixels (:131-160): min_pixel starts at 254, max_pixel at 0 */
    S.min_pixel = 254; S.max_pixel = 0;
    size_t npx = 0;
    {
        const uint8_t* px = pixels;
        for (int32_t i = 0; i < n_runs; ++i)
            for (int x = runs[i].x0; x <= runs[i].x1; ++x, ++px, ++npx) {
                const int b = bg ? bg[(size_t)runs[i].y * bg_stride + x] : 0;
                const int d = method == 0 ? abs(b - *px) : (method == 1 ? (b - *px > 0 ? b - *px : 0) : *px);
                if (d < S.min_pixel) S.min_pixel = d;
                if (d > S.max_pixel) S.max_pixel = d;
            }
    }
    memset(out, 0, sizeof(*out));
    out->threshold = -1; out->effective_threshold = -1;
    if (P->algorithm == 0) { out->initial_action = A_SKIP; return; }                 /* blob_split_algorithm none (:421-422) */

    int action = try_threshold(&S, -1);                                              /* :558 */
    out->initial_action = action;
    double ms, me;
    max_range(P, &ms, &me);
    if (action != A_KEEP && action != A_KEEP_ABORT
        && (P->n_ranges == 0 || (double)((float)npx * S.sqrcm) < me * 100)) {        /* :560-563 */
        if (presumed_nr > 1) {
            const int begin_threshold = P->initial_threshold > S.min_pixel ? P->initial_threshold : S.min_pixel;   /* :592 */
            if (P->algorithm == 1) {                                                 /* complete search (:711-717) */
                for (int i = begin_threshold; i < S.max_pixel; ++i) {
                    const int a = perform(&S, i);
                    if (a == A_ABORT || a == A_KEEP_ABORT) break;
                }
            } else {                                                                 /* :719-726 */
                for (int i = 0; i < 3; ++i) { if (S.run_best != -1) break; work_approximate(&S, begin_threshold, i); }
            }
        }
    }
    out->threshold = S.best_threshold;
    if (S.best_threshold != -1) {
        out->effective_threshold = S.best_threshold > S.min_pixel ? S.best_threshold : S.min_pixel;
        out->n_result = S.n_best;
        out->min_size_bound = S.best_min_size;
    }
    out->n_tried = S.tried;
    out->min_pixel = S.min_pixel; out->max_pixel = S.max_pixel;
    out->first_size = S.first_size;
}
