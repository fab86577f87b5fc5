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


/* ------------------------------------------------------------------------------------------------------------------------------
 * HistorySplit's per-frame decision: which blobs of a frame are split, and into how many.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows HistorySplit::HistorySplit, Application/src/tracker/tracking/HistorySplit.cpp:52-312 (apply_manual_matches :8-50), on flat
 * arrays: blobs are 0 .. n_blobs - 1 (standing for pv::bid), individuals 0 .. n_fish - 1 (Idx_t; a negative entry of a blob's
 * individuals = the invalid Idx_t a manual split leaves there, skipped like :101-102).
 *   map_off / map_fish    PPFrame::blob_mappings (PPFrame.h:68): per blob the individuals mapped to it -- a std::set, ascending
 *   pair_off / pair_blob / pair_d   PPFrame::paired (:69): per individual its (blob, distance) edges, walked in the order given
 *   streak                frame.cached(fdx)->valid_frame_streak (:139-147), threshold < 0 = track_history_split_threshold invalid
 * The reference walks two robin_hood maps (blob_mappings at :76, probs_per_fish at :266) in HASH order.  Nothing decided here depends on
 * that order except (a) the order of `centers`, which only the watershed algorithm reads, and (b) exact ties of two distances for one
 * blob (:206: the individual assigned first keeps it).  Here: blobs ascending, individuals in the order the clique search meets them.
 * Output: number[b] / allow_less[b] = expect[b] (0 = not in `expect`), big[b] = in big_blobs; center_off / center_fish = per blob the
 * individuals whose last_positions :281-291 appends to expect[b].centers, in that order.  Returns the number of big blobs.
 * ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct { int fish; float d; } hs_assign;
int32_t oracle_history_split(int32_t n_blobs, int32_t n_fish, const int32_t* map_off, const int32_t* map_fish,
                             const int32_t* pair_off, const int32_t* pair_blob, const float* pair_d,
                             const int32_t* streak, int32_t split_threshold, const int32_t* manual, int32_t n_manual, int32_t history_split_on,
                             int32_t* number, uint8_t* allow_less, uint8_t* big, int32_t* center_off, int32_t* center_fish) {
    int32_t n_big = 0;
    uint8_t* walked = (uint8_t*)calloc((size_t)n_blobs + 1, 1);
    /* centers as (blob, fish) pairs in the order they are appended; gathered per blob at the end */
    int32_t* cb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * n_fish + 2)); int32_t* cf = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * n_fish + 2)); int32_t nc = 0;
    for (int b = 0; b < n_blobs; ++b) { number[b] = 0; allow_less[b] = 0; big[b] = 0; }
    for (int i = 0; i < n_manual; ++i) {                                    /* :18-36 */
        const int b = manual[i];
        if (b < 0 || b >= n_blobs) continue;                                /* !bdx.valid() / !frame.has_bdx(bdx) */
        if (!big[b]) { big[b] = 1; ++n_big; }
        number[b] = 2; allow_less[b] = 0; walked[b] = 1;
    }
    if (history_split_on) {                                                 /* :63-68 */
        uint8_t* in_b = (uint8_t*)malloc((size_t)n_blobs + 1); uint8_t* in_f = (uint8_t*)malloc((size_t)n_fish + 1);
        int32_t* qb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_blobs + 1)); int32_t* fl = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_fish + 1));
        hs_assign* ab = (hs_assign*)malloc(sizeof(hs_assign) * (size_t)(n_blobs + 1));       /* assign_blob */
        int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_fish + 1));             /* first alternative left of probs_per_fish[f] (index into its sorted edges) */
        int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * (size_t)(pair_off[n_fish] + 1));   /* per individual its edges sorted by (distance, blob): the std::set of :233-236 */
        int32_t* q = (int32_t*)malloc(sizeof(int32_t) * (size_t)(4 * (pair_off[n_fish] + n_fish) + 16));
        for (int b0 = 0; b0 < n_blobs; ++b0) {                              /* :76 */
            if (map_off[b0 + 1] - map_off[b0] <= 1) continue;               /* :79 */
            if (walked[b0]) continue;                                       /* :82 */
            memset(in_b, 0, (size_t)n_blobs); memset(in_f, 0, (size_t)n_fish);
            int qh = 0, qt = 0, nb = 0, nf = 0;
            qb[qt++] = b0;                                                   /* :91-92: the start blob is NOT inserted into available_bdx / already_walked here */
            while (qh < qt) {
                const int c = qb[qh++];
                for (int k = map_off[c]; k < map_off[c + 1]; ++k) {
                    const int f = map_fish[k];
                    if (f < 0) continue;                                    /* :101 */
                    if (split_threshold >= 0) {                             /* :104-148: the streak is cached per individual, the test is the same every time */
                        const int len = streak[f] > 0 ? streak[f] : -1;
                        if (len < 0 || len < split_threshold) continue;
                    }
                    for (int e = pair_off[f]; e < pair_off[f + 1]; ++e) {   /* :151-157 */
                        const int b = pair_blob[e];
                        if (!in_b[b]) { if (qt <= n_blobs) qb[qt++] = b; in_b[b] = 1; ++nb; walked[b] = 1; }
                    }
                    if (!in_f[f]) { in_f[f] = 1; fl[nf++] = f; }            /* :159 */
                }
            }
            if (nf <= nb) continue;                                         /* :172 */
            for (int b = 0; b < n_blobs; ++b) ab[b].fish = -1;
            int ch = 0, ct = 0;
            for (int i = 0; i < nf; ++i) {                                  /* :227-246 */
                const int f = fl[i], e0 = pair_off[f], e1 = pair_off[f + 1];
                cur[f] = e0;
                if (e0 == e1) { cur[f] = -1; continue; }                    /* pairs.empty(): no entry in probs_per_fish */
                for (int e = e0; e < e1; ++e) ord[e] = e;
                for (int a = e0 + 1; a < e1; ++a) {                         /* insertion sort by (d, blob) */
                    const int v = ord[a]; int z = a - 1;
                    while (z >= e0 && (pair_d[ord[z]] > pair_d[v] || (pair_d[ord[z]] == pair_d[v] && pair_blob[ord[z]] > pair_blob[v]))) { ord[z + 1] = ord[z]; --z; }
                    ord[z + 1] = v;
                }
                int w = e0;                                                 /* a std::set: equal (d, blob) entries collapse */
                for (int a = e0 + 1; a < e1; ++a) if (pair_d[ord[a]] != pair_d[ord[w]] || pair_blob[ord[a]] != pair_blob[ord[w]]) ord[++w] = ord[a];
                for (int a = w + 1; a < e1; ++a) ord[a] = -1;
                q[ct++] = f;
            }
            while (ch < ct) {                                               /* :248-257 */
                const int f = q[ch++];
                const int e1 = pair_off[f + 1];
                if (cur[f] < 0 || cur[f] >= e1 || ord[cur[f]] < 0) continue;        /* combinations.empty() */
                const int e = ord[cur[f]], b = pair_blob[e];                /* check_combinations :190-224 */
                const float d = pair_d[e];
                int done = 0;
                if (ab[b].fish < 0) { ab[b].fish = f; ab[b].d = d; done = 1; }
                else if (ab[b].fish != f) {
                    if (!(ab[b].d <= d)) { const int o = ab[b].fish; ab[b].fish = f; ab[b].d = d; q[ct++] = o; done = 1; }
                }
                if (!done) { ++cur[f]; q[ct++] = f; }                       /* :221 erase + :256 push again */
            }
            for (int i = 0; i < nf; ++i) {                                  /* :266-303 */
                const int f = fl[i], e0 = pair_off[f], e1 = pair_off[f + 1];
                if (cur[f] < 0) continue;                                   /* not in probs_per_fish */
                if (cur[f] < e1 && ord[cur[f]] >= 0) continue;              /* alternatives left (:272) */
                const int mx = pair_blob[ord[e0]];                          /* assign_fish[fdx] = the closest (:241) */
                if (ab[mx].fish >= 0) {                                     /* :285-294 */
                    ++number[mx]; cb[nc] = mx; cf[nc++] = ab[mx].fish; ab[mx].fish = -1;
                }
                ++number[mx]; cb[nc] = mx; cf[nc++] = f;                    /* :296-301 */
                if (!big[mx]) { big[mx] = 1; ++n_big; }
            }
        }
        free(in_b); free(in_f); free(qb); free(fl); free(ab); free(cur); free(ord); free(q);
    }
    int32_t o = 0;
    for (int b = 0; b < n_blobs; ++b) { center_off[b] = o; for (int i = 0; i < nc; ++i) if (cb[i] == b) center_fish[o++] = cf[i]; }
    center_off[n_blobs] = o;
    free(walked); free(cb); free(cf);
    return n_big;
}
