/*
 * trex_oracle.h -- CPU restatement ("oracle") of the TRex detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under trex_amd/ (the product) may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU baseline.
 *
 * PARITY STATUS (see DESIGN.md "Oracle"):
 *   - detect stage (RawProcessing::generate_binary + CPULabeling::run) lives in the
 *     un-vendored submodule Application/src/commons (.gitmodules:1-3, empty dir, pinned
 *     SHA unknown).  Restated here from the call sites
 *     (Application/src/tracker/python/BackgroundSubtraction.cpp:126-347) and the published
 *     OpenCV semantics it uses.  Unit-level parity of the detect stage is UNPINNED;
 *     it is anchored by (a) scipy.ndimage cross-checks and (b) the reference's end-to-end
 *     golden CSVs (videos/compare_data_automatic) -- see tests/test_golden_e2e.py.
 *   - track-stage threshold semantics (keep pixel iff diff >= threshold, difference
 *     methods absolute/sign/none) ARE pinned by literal vectors of
 *     Application/Tests/test_pixels.cpp:981-1071,1611-1821 (tests/test_oracle_golden.py).
 */
#ifndef TREX_ORACLE_H
#define TREX_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_params {
    int32_t width, height;
    int32_t threshold;           /* detect_threshold (grabber/misc/default_config.cpp:98) */
    int32_t threshold_maximum;   /* threshold_maximum (:99); <255 => inRange[thr,max]      */
    int32_t enable_difference;   /* enable_difference (:126)                               */
    int32_t absolute_difference; /* detect_threshold_is_absolute (core/default_config.cpp:1168) */
    int32_t image_invert;        /* image_invert (:157)                                    */
    int32_t inclusive;           /* 1 (default of make_params): diff >= thr, core/default_config.cpp:1168; 0: diff > thr (cv::threshold THRESH_BINARY) */
    int32_t zero_is_background;  /* 1: output = grey under mask, CCL labels non-zero       */
    int32_t connectivity;        /* 8 (run overlap +-1) or 4                               */
    int32_t dilation_size;       /* core/default_config.cpp:1163 */
    int32_t use_closing;         /* :1164 */
    int32_t closing_size;        /* :1165 */
    int32_t n_ranges;            /* detect_size_filter; 0 => accept all (SizeFilters.cpp:38) */
    double  cm_per_pixel;
    double  ranges[16];          /* [start,end) pairs */
} oracle_params;

/* 8-byte run: HorizontalLine{y,x0,x1}, x1 inclusive (pv.cpp:509) */
typedef struct oracle_run { uint16_t x0, x1, y, pad; } oracle_run;

/* identical layout to trexhip_blob (include/trexhip.h) so tests compare raw bytes */
typedef struct oracle_blob {
    uint32_t run_begin, n_runs;
    uint32_t pix_begin, n_pixels;
    uint16_t x0, y0, x1, y1;     /* inclusive bounding box */
    uint32_t bid;                /* pv::bid hash of the first run */
    uint32_t px_min_max;         /* min | max << 8 */
    uint32_t parent;             /* re-threshold: index of the detect blob, else 0xFFFFFFFF */
    uint32_t flags;              /* re-threshold: size class 0 in range / 1 below / 2 big */
    uint64_t m10, m01;           /* sum x, sum y over pixels */
    uint64_t m20, m11, m02;      /* sum x^2, sum x*y, sum y^2 */
    uint64_t sp, spx, spy;       /* sum p, sum p*x, sum p*y (p = grey value) */
} oracle_blob;

typedef struct oracle_frame oracle_frame;

/* detect stage: BackgroundSubtraction::apply body for one gray frame */
oracle_frame* oracle_segment(const uint8_t* frame, const uint8_t* bg, const oracle_params* p);
void oracle_frame_counts(const oracle_frame* f, int32_t* n_blobs, int32_t* n_runs, int32_t* n_pixels);
void oracle_frame_copy(const oracle_frame* f, oracle_blob* blobs, oracle_run* runs, uint8_t* pixels);
void oracle_frame_free(oracle_frame* f);

/* pixel passes alone (for morphology / mask tests): out = binary image as fed to CCL */
void oracle_generate_binary(const uint8_t* frame, const uint8_t* bg, uint8_t* out, const oracle_params* p);

/* bounded-sample CPU baseline: n frames (contiguous), `threads` OpenMP threads;
 * returns total number of blobs (so the work cannot be optimised away) */
int64_t oracle_segment_batch(const uint8_t* frames, int32_t n, const uint8_t* bg,
                             const oracle_params* p, int32_t threads);

/* track stage (Tracker::prefilter -> pixel::threshold_blob, test_pixels.cpp semantics):
 * method 0 = absolute |bg-p|, 1 = sign max(0,bg-p), 2 = none (p).  keep iff diff >= threshold.
 * in: runs+pixels of ONE blob; out: thresholded runs+pixels (not re-labelled).
 * returns number of output runs; out arrays must hold n_pixels_in entries (worst case). */
int32_t oracle_line_without_grid(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels,
                                 const uint8_t* bg, int32_t bg_stride, int32_t method, int32_t threshold,
                                 oracle_run* out_runs, uint8_t* out_pixels, int32_t* n_out_pixels);

/* threshold_blob: line_without_grid + re-label into sub-blobs (8-connectivity),
 * returned as an oracle_frame in the full-frame coordinate system */
oracle_frame* oracle_threshold_blob(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels,
                                    const uint8_t* bg, int32_t bg_stride, int32_t width, int32_t height,
                                    int32_t method, int32_t threshold, int32_t connectivity);

/* Tracker::prefilter arithmetic for a whole frame (tracking/Tracker.cpp:765-912): threshold_blob on every blob of
 * `detect` (an oracle_segment result), sub-blobs in raster order of their first run, each with parent = index of
 * its detect blob and flags = size class against `ranges` (cm^2, half open). */
oracle_frame* oracle_rethreshold_frame(const oracle_frame* detect, const uint8_t* frame, const uint8_t* bg, int32_t width,
                                       int32_t height, int32_t method, int32_t threshold, int32_t connectivity,
                                       const double* ranges, int32_t n_ranges, double cm_per_pixel, int32_t invert);
/* normalised crops (FilterCache.cpp:21-115,276-288): transforms are row-major 2x3 float [m0 m1 m2; m3 m4 m5] */
void oracle_normalize_transform(const float* tr6, float midline_length, int32_t use_legacy, int32_t out_w, int32_t out_h, float scale, float* M6);
void oracle_moments_transform(const oracle_blob* B, float* tr6);
void oracle_warp_affine_u8(const uint8_t* src, int32_t sw, int32_t sh, const float* M6, uint8_t* dst, int32_t dw, int32_t dh);
void oracle_warp_affine_nearest_u8(const uint8_t* src, int32_t sw, int32_t sh, const float* M6, uint8_t* dst, int32_t dw, int32_t dh);
uint32_t oracle_bid(uint32_t x0, uint32_t x1, uint32_t y, uint32_t n_runs);
/* colour encodings: layout pinned by test_pixels.cpp:629-795; encoding 0 gray, 1 r3g3b2, 2 rgb8 */
uint8_t oracle_vec_to_r3g3b2(uint8_t c0, uint8_t c1, uint8_t c2);
void oracle_r3g3b2_to_vec(uint8_t code, uint8_t* out3);
uint8_t oracle_bgr2gray(uint8_t b, uint8_t g, uint8_t r);
int32_t oracle_line_without_grid_enc(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels, int32_t pixel_enc,
                                     const uint8_t* bg, int32_t bg_stride_px, int32_t bg_enc, int32_t method, int32_t threshold,
                                     oracle_run* out_runs, uint8_t* out_pixels, int32_t* n_out_pixels);

/* blob splitting by threshold search (tracking/SplitBlob.cpp:130-255,419-800); trex_split.c */
typedef struct oracle_split_params {
    int32_t initial_threshold;        /* (calculate_posture ? max(track_threshold, track_posture_threshold) : track_threshold) + 1, :512 */
    int32_t algorithm;                /* blob_split_algorithm: 0 none, 1 threshold, 2 threshold_approximate */
    float blob_split_max_shrink, blob_split_global_shrink_limit;
    float cm_per_pixel;
    int32_t n_ranges;                 /* track_size_filter */
    double ranges[16];
} oracle_split_params;
typedef struct oracle_split_info {
    int32_t threshold;                /* best_match.threshold or -1 */
    int32_t effective_threshold;      /* the threshold apply_threshold really used for it (>= min_pixel) */
    int32_t initial_action, n_result, n_tried, min_pixel, max_pixel;
    float first_size;
    double min_size_bound;            /* blobs below it were removed from the result (evaluate_result_multiple) */
} oracle_split_info;
void oracle_split_search(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels, const uint8_t* bg, int32_t bg_stride,
                         int32_t width, int32_t height, int32_t method, int32_t connectivity, const oracle_split_params* P,
                         int32_t presumed_nr, oracle_split_info* out);

/* HistorySplit's per-frame decision (tracking/HistorySplit.cpp:52-312), see trex_split.c */
int32_t oracle_history_split(int32_t n_blobs, int32_t n_fish, const int32_t* map_off, const int32_t* map_fish,
                             const int32_t* pair_off, const int32_t* pair_blob, const float* pair_d,
                             const int32_t* streak, int32_t split_threshold, const int32_t* manual, int32_t n_manual, int32_t history_split_on,
                             int32_t* number, uint8_t* allow_less, uint8_t* big, int32_t* center_off, int32_t* center_fish);

#ifdef __cplusplus
}
#endif
#endif
