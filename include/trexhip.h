/*
 * trexhip.h -- C ABI of libtrexhip: the MI355X (gfx950) implementation of TRex's per-frame
 * detection + identity hot path.  extern "C", plain pointers and sizes only.
 *
 * Each entry point names the reference interface (file:line under /root/reference) whose work
 * it replaces; INTEGRATION.md shows the binding a TRex maintainer adds on the C++ side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative TREXHIP_E_* code; the message is kept
 *     per thread and read with trexhip_last_error().  No exception crosses this ABI.
 *   - one ctx per (host thread, device); calls on one ctx are not re-entrant.
 *   - "_device" variants take device pointers (HBM-resident data) and only enqueue work on the
 *     ctx stream; the plain variants take host pointers and include the PCIe copies.
 *   - all result pointers handed out stay owned by the ctx and are valid until the next call
 *     of the same function on that ctx.
 */
#ifndef TREXHIP_H
#define TREXHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TREXHIP_ABI_VERSION 6

enum {
    TREXHIP_OK = 0,
    TREXHIP_E_INVALID = -1,    /* bad argument / not initialised (e.g. no background yet)   */
    TREXHIP_E_DEVICE = -2,     /* HIP runtime error                                         */
    TREXHIP_E_CAPACITY = -3,   /* a frame exceeded max_runs / pooled output capacity         */
    TREXHIP_E_UNSUPPORTED = -4, /* a setting of the reference that this library does not implement was switched on   */
    TREXHIP_E_NOMEM = -5
};

/* per-frame status bits in trexhip_frame_info.flags */
#define TREXHIP_FRAME_OVERFLOW_RUNS   1u
#define TREXHIP_FRAME_OVERFLOW_OUTPUT 2u

/* Settings the hot path reads (names = TRex setting names). */
typedef struct trexhip_params {
    int32_t device;              /* HIP device ordinal                                         */
    int32_t width, height;       /* frame size in pixels, < 65535 (pv.cpp:601-602)             */
    int32_t max_batch;           /* frames per segment call                                    */
    int32_t max_runs;            /* capacity: raw horizontal lines per frame                   */
    int32_t max_blobs;           /* capacity: kept blobs per frame (pool = max_batch * this)   */
    int32_t max_pixels;          /* capacity: kept foreground pixels per frame (pooled)        */
    /* RawProcessing::generate_binary (BackgroundSubtraction.cpp:209) */
    int32_t threshold;           /* detect_threshold            grabber/misc/default_config.cpp:98  */
    int32_t threshold_maximum;   /* threshold_maximum           :99 (<255 => inRange[thr,max])      */
    int32_t enable_difference;   /* enable_difference           :126                                */
    int32_t absolute_difference; /* detect_threshold_is_absolute core/default_config.cpp:1168       */
    int32_t image_invert;        /* image_invert                :1159                               */
    int32_t inclusive;           /* 1 (default): keep diff >= thr; 0: diff > thr (a plain cv::threshold).  UNPINNED: the detect-stage comparison
                                  * happens inside RawProcessing::generate_binary (BackgroundSubtraction.cpp:209), whose source is not in the
                                  * reference tree.  The default follows the documented wording ("disregards any pixel |p| < threshold",
                                  * core/default_config.cpp:1168 -- the words of the track-stage rule :1167, which Tests/test_pixels.cpp:1026-1059
                                  * pins as >=); the reference's golden CSVs cannot tell the two apart.  A maintainer who
                                  * knows generate_binary to be strict sets 0 here (HipBackgroundSubtraction::Settings::inclusive = false). */
    int32_t zero_is_background;  /* 1 (default): a masked pixel of grey value 0 is background        */
    /* CPULabeling::run (BackgroundSubtraction.cpp:216) */
    int32_t connectivity;        /* 8 (default) or 4                                                 */
    int32_t dilation_size;       /* core/default_config.cpp:1163                                     */
    int32_t use_closing;         /* :1164                                                            */
    int32_t closing_size;        /* :1165                                                            */
    /* size filter (BackgroundSubtraction.cpp:259, core/SizeFilters.cpp:37-53) */
    int32_t n_ranges;            /* detect_size_filter, 0 = accept all                               */
    double  cm_per_pixel;
    double  ranges[16];          /* [start,end) pairs in cm^2                                        */
    /* meta_encoding of the produced pixel arrays (Background::meta_encoding(), BackgroundSubtraction.cpp:132,151-186):
     * TREXHIP_ENC_GRAY 1 B/px grey value; TREXHIP_ENC_R3G3B2 1 B/px colour code (convert_to_r3g3b2, layout pinned by
     * Tests/test_pixels.cpp:629-795); TREXHIP_ENC_RGB8 3 B/px in the input's memory order (BGRA2BGR).  The colour encodings
     * need colour input (trexhip_segment_color*); detection itself always works on cv::cvtColor(BGR2GRAY) (or color_channel). */
    int32_t pixel_encoding;
    /* Pre-processing options of RawProcessing::generate_binary that are NOT implemented (their arithmetic lives in the un-vendored
     * commons and nothing in the tree pins it).  They are part of the parameter block so that a caller can hand over what TRex's
     * settings say: any non-default value makes trexhip_create fail with TREXHIP_E_UNSUPPORTED -- never a silently different mask.
     *   image_adjust (+ image_contrast_increase / image_brightness_increase / image_square_brightness), blur_difference,
     *   equalize_histogram, correct_luminance      grabber/misc/default_config.cpp:121-129
     *   use_adaptive_threshold (+ adaptive_threshold_scale)                         core/default_config.cpp:1161-1162 */
    int32_t image_adjust, blur_difference, equalize_histogram, correct_luminance, use_adaptive_threshold;
    /* host tiles of a colour format with a gray / binary pixel_encoding (trexhip_segment_color): 0 (default) = the upload threads reduce
     * the tile to gray while they copy it into the pinned ring (a quarter / a third of the bytes cross PCIe; same fixed-point formula),
     * 1 = the colour tile is uploaded and reduced on the device.  The colour encodings always upload the colour tile. */
    int32_t device_color_reduce;
    int32_t reserved_[1];
} trexhip_params;
enum { TREXHIP_ENC_GRAY = 0, TREXHIP_ENC_R3G3B2 = 1, TREXHIP_ENC_RGB8 = 2 };   /* order of cmn::meta_encoding_t */

/* size class of a re-thresholded blob against track_size_filter (tracking/Tracker.cpp:864-912) */
#define TREXHIP_BLOB_IN_RANGE    0u   /* fish_size.in_range_of_one -> commit                     */
#define TREXHIP_BLOB_BELOW_RANGE 1u   /* recount < max_range().start -> filter_out(OutsideRange) */
#define TREXHIP_BLOB_BIG         2u   /* otherwise -> big_blob                                    */

/* HorizontalLine{y,x0,x1}, x1 inclusive (pv.cpp:509); 8 bytes */
typedef struct trexhip_run { uint16_t x0, x1, y, pad; } trexhip_run;

/* One blob = one blob::Pair (lines + pixels) plus the reductions the tracker asks of it
 * (pv::Blob::calculate_moments, Individual::weighted_centroid -- Individual.cpp:2414-2440).
 * Sums are exact integers so host-side float results do not depend on summation order. */
typedef struct trexhip_blob {
    uint32_t run_begin, n_runs;     /* index into the frame's run array (sorted by (y,x0))   */
    uint32_t pix_begin, n_pixels;   /* index into the frame's pixel array (gathered in run order) */
    uint16_t x0, y0, x1, y1;        /* inclusive bounding box                                */
    uint32_t bid;                   /* pv::bid of the blob (13/13/6-bit hash of first line)  */
    uint32_t px_min_max;            /* min | max << 8 of the grey values                     */
    uint32_t parent;                /* re-threshold results: pooled index of the detect blob; else 0xFFFFFFFF */
    uint32_t flags;                 /* re-threshold results: TREXHIP_BLOB_* size class; else 0 */
    uint64_t m10, m01;              /* sum x, sum y                                          */
    uint64_t m20, m11, m02;         /* sum x^2, sum x*y, sum y^2                             */
    uint64_t sp, spx, spy;          /* sum p, sum p*x, sum p*y                               */
} trexhip_blob;

typedef struct trexhip_frame_info {
    uint32_t n_blobs, n_runs, n_pixels;   /* kept (after size filter)                        */
    uint32_t blob_begin, run_begin, pix_begin; /* offsets of this frame in the pooled arrays */
    uint32_t n_raw_runs, n_raw_blobs;     /* before filtering                                */
    uint32_t flags;                       /* TREXHIP_FRAME_*                                 */
    uint32_t reserved[3];
} trexhip_frame_info;

/* Host view of one segmented batch (pooled arrays, pinned host memory owned by the ctx). */
typedef struct trexhip_batch_result {
    int32_t n_frames;
    uint32_t total_blobs, total_runs, total_pixels;
    const trexhip_frame_info* frames;   /* [n_frames]                                        */
    const trexhip_blob* blobs;          /* [total_blobs]; run_begin/pix_begin are FRAME-relative */
    const trexhip_run* runs;            /* [total_runs]                                      */
    const uint8_t* pixels;              /* [total_pixels * pixel_channels]                   */
    uint32_t pixel_channels;            /* bytes per pixel: 1 (gray, r3g3b2) or 3 (rgb8); pix_begin / n_pixels count PIXELS */
    uint32_t reserved_;
} trexhip_batch_result;

/* Device view of the same tables (for downstream device stages and torch interop). */
typedef struct trexhip_device_view {
    const trexhip_frame_info* frames;
    const trexhip_blob* blobs;
    const trexhip_run* runs;
    const uint8_t* pixels;
    const uint32_t* totals;             /* [3] total blobs, runs, pixels                     */
    const uint32_t* blob_frame;         /* [pool] frame index of each pooled blob            */
} trexhip_device_view;

typedef struct trexhip_ctx trexhip_ctx;

int trexhip_abi_version(void);
const char* trexhip_last_error(void);
void trexhip_default_params(trexhip_params* p, int32_t width, int32_t height);

/* BackgroundSubtraction::BackgroundSubtraction / Detection::init (python/Detection.cpp:16-58) */
int trexhip_create(const trexhip_params* p, trexhip_ctx** out);
/* BackgroundSubtraction::deinit (BackgroundSubtraction.cpp:118-120) */
void trexhip_destroy(trexhip_ctx* ctx);
/* The settings the reference re-reads on EVERY apply() (python/BackgroundSubtraction.cpp:137-143: cm_per_pixel, detect_size_filter;
 * RawProcessing reads its thresholds per call as well): trexhip_get_live_params copies the context's current values,
 * trexhip_update_params replaces them for every segment call that follows (the kernels take the settings by value at launch: this is a
 * host-side update, no device work; not to be called while another thread is inside a segment call of the same ctx).  Settings that size
 * or lay out buffers (frame size, capacities, pixel_encoding, morphology sizes) stay fixed for the life of the context.  More than 8
 * ranges or image_invert with a colour pixel_encoding are refused like in trexhip_create; the context keeps its previous values then. */
typedef struct trexhip_live_params {
    int32_t threshold, threshold_maximum, inclusive;                     /* as in trexhip_params */
    int32_t enable_difference, absolute_difference, image_invert, zero_is_background;
    int32_t n_ranges;                                                    /* detect_size_filter, 0 = accept all, at most 8 */
    double  cm_per_pixel;
    double  ranges[16];                                                  /* [start,end) pairs in cm^2 */
} trexhip_live_params;
int trexhip_get_live_params(trexhip_ctx* ctx, trexhip_live_params* out);
int trexhip_update_params(trexhip_ctx* ctx, const trexhip_live_params* lp);
/* use an external hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream */
int trexhip_set_stream(trexhip_ctx* ctx, void* hip_stream);

/* BackgroundSubtraction::set_background -> Data::set (BackgroundSubtraction.cpp:86-101) */
int trexhip_set_background(trexhip_ctx* ctx, const uint8_t* gray, int32_t stride);
int trexhip_set_background_device(trexhip_ctx* ctx, const uint8_t* d_gray);
/* Background(image, meta_encoding_t::rgb8): a BGR / BGRA background.  Detection and the track-stage thresholds keep using its
 * cv::cvtColor(BGR2GRAY) (or the picked color_channel, as in trexhip_segment_color), which this call also installs as the gray
 * background; the colour image stays resident for the per-channel background-difference crops of the rgb8 pixel encoding
 * (imageFromLines' `differences`, Application/Tests/test_pixels.cpp:1381-1479). */
int trexhip_set_background_color(trexhip_ctx* ctx, const uint8_t* bgr, int32_t stride_bytes, int32_t channels, int32_t color_channel);
int trexhip_set_background_color_device(trexhip_ctx* ctx, const uint8_t* d_bgr, int32_t channels, int32_t color_channel);

/* background model from n sampled gray frames in HBM ("next" row of SURVEY.md 8f: Segmenter::trigger_average_generator,
 * ui/Segmenter.cpp:467-566; averaging_method grabber/misc/default_config.cpp:131): method 0 = mean (float accumulation in
 * sample order, rounded half-to-even), 1 = max, 2 = min, 3 = mode (most frequent value of the pixel, the smallest wins a tie; at most
 * 255 samples); the result becomes the context's background. */
int trexhip_generate_average_device(trexhip_ctx* ctx, const uint8_t* d_frames, int32_t n, int32_t method);
int trexhip_get_background(trexhip_ctx* ctx, uint8_t* gray, int32_t stride);

/* BackgroundSubtraction::apply(std::vector<TileImage>&&) hot loop (BackgroundSubtraction.cpp:146-316):
 * generate_binary + CPULabeling::run + size filter for n gray frames. */
int trexhip_segment_device(trexhip_ctx* ctx, const uint8_t* d_frames, int32_t n);
int trexhip_segment(trexhip_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t n);
/* the same for the BGR / BGRA tile images TRex actually hands over (BackgroundSubtraction.cpp:162-180):
 * channels 3 or 4; color_channel < 0 (or >= channels) = cv::cvtColor(BGR2GRAY / BGRA2GRAY), else that channel */
int trexhip_segment_color(trexhip_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t n,
                          int32_t channels, int32_t color_channel);
/* the same for n contiguous colour frames already in HBM ([n][height][width][channels]) */
int trexhip_segment_color_device(trexhip_ctx* ctx, const uint8_t* d_color_frames, int32_t n, int32_t channels, int32_t color_channel);
int trexhip_pixel_channels(trexhip_ctx* ctx);   /* bytes per pixel of the pixel arrays and channels of the crops: 1 or 3 (rgb8) */
/* device buffers for callers that do not link the HIP runtime themselves (the C++ adapters in trex_amd/host): plain
 * hipMalloc / hipFree / stream-ordered device-to-host and host-to-device copies (synchronous on return) on the context's device and stream */
int trexhip_device_alloc(trexhip_ctx* ctx, size_t bytes, void** out_device_ptr);
int trexhip_device_free(trexhip_ctx* ctx, void* device_ptr);
int trexhip_copy_to_host(trexhip_ctx* ctx, void* host_dst, const void* device_src, size_t bytes);
int trexhip_copy_to_device(trexhip_ctx* ctx, void* device_dst, const void* host_src, size_t bytes);
/* wait for the last segment call and copy its tables to pinned host memory.  Batches of up to 16 frames (TRex's default detect_batch_size is 1,
 * core/default_config.cpp:1113) leave the device as ONE kernel that writes the filled parts of all five tables into the pinned mirrors + one
 * stream synchronize; larger batches as two rounds of DMA copies (frame table and totals, then the tables at their exact sizes). */
int trexhip_fetch(trexhip_ctx* ctx, trexhip_batch_result* out);
int trexhip_device_view_get(trexhip_ctx* ctx, trexhip_device_view* out);
int trexhip_synchronize(trexhip_ctx* ctx);

/* ---- .pv frame bodies ------------------------------------------------------------------------------
 * pv::Frame::serialize (ProcessedVideo/pv.cpp:666-703) for every frame of the last fetched batch, on the device, in the layout
 * pv::Frame::read_from accepts for file version V_6 (pv.cpp:296-420) -- the newest one whose line type is in the reference tree
 * (LegacyShortHorizontalLine, pv.h:17-52; >= V_7 use commons' ShortHorizontalLine): u8 compression_flag = 0, u64 timestamp, u16 n,
 * n x {u16 start_y, u16 mask_size, mask_size x {u16 x0, u16 x1 << 1 | eol}, 1 byte per pixel}.  Gray pixel arrays, width <= 32768.
 *   timestamps  host array [n_frames] (relative to the file header's timestamp) or NULL (zeros)
 *   d_out       the bodies back to back, in frame order;  d_offsets [n_frames + 1] byte offsets (d_offsets[n_frames] = total bytes;
 *               when that exceeds `capacity` nothing valid was written: call again with a larger buffer)
 * Compression and the index table: trexhip_pv_write_frames below (host side); the file header (pv.cpp:842-990) is not produced. */
int trexhip_pack_frames_v6_device(trexhip_ctx* ctx, const uint64_t* timestamps, uint8_t* d_out, size_t capacity, uint64_t* d_offsets);

/* ---- .pv data section: per-frame LZO1X compression + index table (host side, no GPU needed) -------
 * trexhip_pv_write_frames = the tail of pv::Frame::serialize (pv.cpp:705-772: a pack of >= 15000 bytes -- every pack when
 * always_compress, which is what the rgb8 encoding does -- goes through LZO1X-1 and is kept compressed if 8 + compressed < uncompressed)
 * + pv::File::add_individual (pv.cpp:1488-1496: u8 compression_flag, then either the pack or u32 compressed size, u32 uncompressed size,
 * compressed bytes; the frame's offset in the file goes to the index table) for the frames of trexhip_pack_frames_v6_device copied to
 * the host (`bodies` + `offsets`, each starting with its compression_flag 0).
 *   file_offset  where `out` will sit in the file (the data section starts right behind the header, pv.cpp:1079-1095)
 *   out          capacity >= sum of the bodies is always enough;  *out_bytes = bytes written
 *   index_table  [n_frames] u64 file offsets = what pv::Header::update writes as the index table (pv.cpp:1181-1192) and
 *                Header::read loads (pv.cpp:986-990)
 * The stream is read back by pv::Frame::read_from's lzo1x_decompress (pv.cpp:316-340).  The compressor is this library's own LZO1X
 * encoder, not a byte-for-byte lzo1x_1_compress (ProcessedVideo/lzo/minilzo.c): a reader only ever sees the decompressed pack.
 * Still blocked on the un-vendored commons: the header's strings / cv::Size encoding (DataFormat) and the >= V_7 line type. */
size_t trexhip_lzo1x_bound(size_t n);            /* pv.cpp:712 OUT_LEN: n + n / 16 + 64 + 3 */
int trexhip_lzo1x_compress(const uint8_t* in, size_t n, uint8_t* out, size_t capacity, size_t* out_len);
int trexhip_pv_write_frames(const uint8_t* bodies, const uint64_t* offsets, int32_t n_frames, int32_t always_compress, uint64_t file_offset,
                            uint8_t* out, size_t capacity, uint64_t* index_table, size_t* out_bytes);

/* ---- track-stage re-threshold -----------------------------------------------------------------
 * Tracker::prefilter's arithmetic (tracking/Tracker.cpp:765-849): for every kept blob of the last segmented
 * batch, pixel::threshold_blob(blob, threshold, background) -- keep a pixel iff diff >= threshold with
 * method 0 = |bg-p| (track_threshold_is_absolute), 1 = max(bg-p,0) (signed), 2 = p (no background
 * subtraction); runs split where pixels fail, survivors re-labelled into sub-blobs.  Nothing is dropped:
 * each sub-blob carries `parent` (pooled index of its detect blob) and `flags` (TREXHIP_BLOB_* class
 * against size_ranges = track_size_filter, cm^2), so the host applies prefilter's remaining policy
 * (recount = sum of n_pixels over a parent's sub-blobs).  Results are a second table set. */
int trexhip_rethreshold_device(trexhip_ctx* ctx, int32_t threshold, int32_t method, const double* size_ranges, int32_t n_ranges);
/* the same with one threshold per detect blob (device array indexed by pooled blob index; negative = skip the blob):
 * the building block of SplitBlob::apply_threshold (tracking/SplitBlob.cpp:130-164), which tries different thresholds on
 * different merged blobs; the search over thresholds stays with the caller */
int trexhip_rethreshold_per_blob_device(trexhip_ctx* ctx, int32_t threshold, const int32_t* d_blob_thresholds, int32_t method,
                                        const double* size_ranges, int32_t n_ranges);
int trexhip_fetch_rethreshold(trexhip_ctx* ctx, trexhip_batch_result* out);

/* ---- splitting merged blobs: SplitBlob's threshold search -----------------------------------------------
 * SplitBlob::split (tracking/SplitBlob.cpp:419-800) for blob_split_algorithm threshold / threshold_approximate: for every detect blob
 * of the last batch with presumed_nr[blob] > 0 (PrefilterBlobs::split_big's split_expectation::number, PrefilterBlobs.cpp:217-236;
 * <= 0 = not a candidate) find the smallest threshold at which pixel::threshold_blob on the blob's difference values
 * (apply_threshold :130-179) is accepted by evaluate_result_multiple (:193-255).  The reference re-labels the blob on the CPU for
 * every tried threshold; here one wave per blob keeps the difference values in LDS.  Large blobs, which the reference searches
 * with 4 pool threads (:735-760), get the deterministic sequential result (identical for `threshold`, see oracle/trex_split.c).
 *   d_thresholds [n_blobs] int32: the threshold to hand to trexhip_rethreshold_per_blob_device (which then yields the sub-blobs
 *                                 threshold_blob returns), or -1 when the blob cannot be split / is no candidate
 *   d_info       [n_blobs]: what the search saw.  Of the sub-blobs, those with n_pixels * cm^2 < min_size_bound are not part
 *                           of SplitBlob::split's result (:204-221); the rest is returned sorted by (num_pixels, blob_id) descending
 *                           and shifted by -bounds().pos() (:166-172), which the caller does on the fetched table.
 * Needs a fetched batch (n_blobs = total_blobs).  Three size classes by LDS need (2048 / 16384 / 61440 pixels per blob); blobs with
 * more than 61440 pixels or 2048 lines (or 4096 lines after thresholding) report status 2 and no threshold. */
typedef struct trexhip_split_params {
    int32_t track_threshold;                 /* core/default_config.cpp track_threshold            */
    int32_t track_posture_threshold;
    int32_t calculate_posture;               /* initial threshold = (calculate_posture ? max(both) : track_threshold) + 1 (:512) */
    int32_t algorithm;                       /* blob_split_algorithm: 0 none, 1 threshold (default, :923), 2 threshold_approximate; 3 fill (cv::watershed, :419-485) and 4 fill_approximate are refused with TREXHIP_E_UNSUPPORTED */
    float   blob_split_max_shrink;           /* :921 (0.2) */
    float   blob_split_global_shrink_limit;  /* :922 (0.2) */
    int32_t n_ranges;                        /* track_size_filter, cm^2, at most 8 ranges */
    int32_t reserved_;
    double  size_ranges[16];
} trexhip_split_params;
typedef struct trexhip_split_info {
    int32_t threshold;                       /* best_match.threshold as the reference records it, or -1 */
    int32_t effective_threshold;             /* the threshold apply_threshold really used for it (clamped to the blob's smallest difference) */
    int32_t status;                          /* 0 searched, 2 capacity, 3 not a candidate / frame overflowed */
    int32_t initial_action;                  /* split::Action of the first try: 1 KEEP_ABORT, 2 REMOVE, 3 ABORT, 4 TOO_FEW, 5 SKIP */
    int32_t n_result;                        /* blobs SplitBlob::split returns */
    int32_t n_evaluated;                     /* labelling passes spent on the blob */
    int32_t min_pixel, max_pixel;            /* range of the difference values (:142-156) */
    float   first_size;                      /* size of the biggest sub-blob at the first try, cm^2 (:530-531) */
    float   reserved_;
    double  min_size_bound;
} trexhip_split_info;
void trexhip_default_split_params(trexhip_split_params* p);
int trexhip_split_search_device(trexhip_ctx* ctx, const trexhip_split_params* sp, int32_t method, const int32_t* d_presumed_nr,
                                int32_t n_blobs, int32_t* d_thresholds, trexhip_split_info* d_info);

/* ---- posture (outline -> midline) -----------------------------------------------------------------
 * posture::calculate_posture (tracking/Posture.cpp:305-399) for every blob of a table of the last batch
 * (table 0 = detect blobs, 1 = re-thresholded sub-blobs, i.e. the caller picks track_posture_threshold through
 * trexhip_rethreshold_device).  One pass per call: the reference's retry loop "threshold += 2 until a midline is found"
 * (:331-381) only ever re-runs blobs too small for a midline and ends in its first-outline fallback, which is what one pass
 * returns for them (status 3 / 4 with the outline) -- shown on the CPU restatement, tests/test_posture_oracle.py.
 * Outputs (caller-owned device memory, pooled order like trexhip_fetch):
 *   outline  [n_blobs][max_points] float2   resampled, smoothed, EFT-approximated outline rotated so that point 0 is
 *                                           the tail (what Outline holds after calculate_midline), relative to the
 *                                           blob's bounds().pos()
 *   segments [n_blobs][max_points/2+1] float4 = MidlineSegment{pos.x, pos.y, height, l_length} (Outline.h:241-250)
 *   info     [n_blobs] status 0 ok / 1 empty / 2 capacity (more than max_points traced outline points, 2048 lines or 1022 rows) /
 *                      3 no curvature peak / 4 too few midline segments */
typedef struct trexhip_posture_params {
    float   outline_resample;               /* core/default_config.cpp:898  (1)    */
    int32_t outline_smooth_samples;         /* :890 (4)                            */
    int32_t outline_smooth_step;            /* :889 (1)                            */
    int32_t outline_approximate;            /* :888 (3)                            */
    float   outline_curvature_range_ratio;  /* :891 (0.03)                         */
    float   midline_walk_offset;            /* :892 (0.025)                        */
    int32_t max_points;                     /* capacity per blob (outline points; the traced lattice outline has 2 per pixel edge), even, 8..4096 */
    /* not implemented, refused with TREXHIP_E_UNSUPPORTED when switched on (never silently ignored):
     *   posture_closing_steps > 0 (Posture.cpp:335, morphological closing inside threshold_get_biggest_blob),
     *   peak_mode = broad (1; Outline.cpp:627-661; 0 = pointy, the default :897),
     * posture_direction_smoothing is not used by the posture call: the window is the tracker's (Individual::calculate_previous_vector);
     * with a value > 1 the reference hands the resulting movement direction to Midline::post_process -- trexhip_midline_movement_device */
    int32_t posture_closing_steps, peak_mode, posture_direction_smoothing;
} trexhip_posture_params;
typedef struct trexhip_posture_info { int32_t status, n_outline, n_segments, tail_index, head_index, n_traced, reserved[2]; } trexhip_posture_info;
void trexhip_default_posture_params(trexhip_posture_params* p);
int trexhip_posture_device(trexhip_ctx* ctx, int32_t table, const trexhip_posture_params* pp, int32_t n_blobs,
                           float* d_outline, float* d_segments, trexhip_posture_info* d_info);

/* posture::calculate_posture WITH its retry loop (Posture.cpp:305-399), for every DETECT blob of the last fetched batch: threshold =
 * track_posture_threshold; repeat { biggest sub-blob at that threshold (pixel::threshold_get_biggest_blob; method as in
 * trexhip_rethreshold_device) -> outline relative to the original blob -> midline; done on success; else threshold += 2 } until the
 * sub-blob has fewer than max(1, pixels / 10) pixels or threshold >= track_posture_threshold + 100; without success the first outline
 * that could be traced is returned without a midline (status of the last attempt, n_segments 0).  Outputs as trexhip_posture_device
 * (pooled order of the detect table); d_threshold_used / d_iterations ([n_blobs] int32, may be NULL) = the threshold whose result is
 * returned (-1: none) and the attempts made.  Uses (overwrites) the context's re-threshold tables. */
int trexhip_posture_auto_device(trexhip_ctx* ctx, const trexhip_posture_params* pp, int32_t method, int32_t track_posture_threshold,
                                int32_t n_blobs, float* d_outline, float* d_segments, trexhip_posture_info* d_info,
                                int32_t* d_threshold_used, int32_t* d_iterations);

/* Midline::post_process (no movement information, posture_direction_smoothing <= 1; Outline.cpp:895-1060) followed by
 * Midline::normalize() (Outline.cpp:1270-1454; call site Individual.cpp:1369-1372) for every blob of a posture call.
 *   d_segments: the segments buffer of trexhip_posture_device (same max_points); post-processed IN PLACE (head part straightened)
 *   d_midline : [n_blobs][midline_resolution] float4 = MidlineSegment{pos.x,pos.y,height,l_length}, head at the origin
 *   info      : status 0 ok / 1 no midline / 2 resampling did not yield midline_resolution points (normalize() == nullptr);
 *               len, angle, offset = Midline::len() / angle() / offset() (blob-local coordinates, like the outline) */
typedef struct trexhip_midline_params {
    int32_t midline_resolution;             /* core/default_config.cpp:894 (25)   */
    float   midline_stiff_percentage;       /* :893 (0.15)                        */
    int32_t midline_invert;                 /* :901 (false)                       */
    int32_t midline_start_with_head;        /* :900 (false)                       */
} trexhip_midline_params;
typedef struct trexhip_midline_info { int32_t status, n; float len, angle, offx, offy; int32_t reserved[2]; } trexhip_midline_info;
void trexhip_default_midline_params(trexhip_midline_params* p);
int trexhip_midline_device(trexhip_ctx* ctx, const trexhip_midline_params* mp, int32_t n_blobs, int32_t max_points,
                           const trexhip_posture_info* d_posture_info, float* d_segments, float* d_midline,
                           trexhip_midline_info* d_midline_info);
/* The same with MovementInformation::direction per blob (posture_direction_smoothing > 1: Individual.cpp:1364-1369 hands
 * calculate_previous_vector(frame) to post_process): d_movement_direction [n_blobs][2] float (x, y), (0, 0) = no information for that
 * blob; NULL = trexhip_midline_device.  A midline whose direction (Midline::midline_direction, Outline.cpp:870-887) points against the
 * movement -- acos(-direction . movement) < acos(direction . movement), :937 -- is turned round; trexhip_midline_info.reserved[0] = 1 for
 * such a blob (`_inverted_because_previous`): the caller swaps its head and tail index (:959).  The vector itself is tracker state (the
 * previous frames' velocities): the caller computes it. */
int trexhip_midline_movement_device(trexhip_ctx* ctx, const trexhip_midline_params* mp, int32_t n_blobs, int32_t max_points,
                                    const trexhip_posture_info* d_posture_info, float* d_segments, float* d_midline,
                                    trexhip_midline_info* d_midline_info, const float* d_movement_direction);

/* ---- crops ------------------------------------------------------------------------------------
 * constraints::diff_image (tracking/FilterCache.cpp:265-294): one out_w x out_h uint8 crop per blob of the
 * last segmented batch, pooled order (blob i of trexhip_fetch == crop i).  n_blobs = total_blobs of that
 * batch ([n][out_h][out_w][3] for the rgb8 pixel encoding, colour codes warped with nearest neighbour for r3g3b2; the difference modes of rgb8 are per-channel
 * differences against the colour background of trexhip_set_background_color, r3g3b2 crops hold raw codes only).  normalization: individual_image_normalization none or
 * moments (posture / legacy: next function).  difference: 0 = grey
 * values, 1 = |bg - p|, 2 = max(bg - p, 0)  (track_background_subtraction, FilterCache.cpp:171-175). */
enum { TREXHIP_NORMALIZE_NONE = 0, TREXHIP_NORMALIZE_MOMENTS = 1, TREXHIP_NORMALIZE_POSTURE = 2 };
int trexhip_crops_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                         int32_t normalization, int32_t difference);

/* posture / legacy normalisation (FilterCache.cpp:267-274): the caller supplies, per blob in pooled order, the 2x3 row-major
 * float matrix of Midline::transform(normalize).toCV() (Outline.cpp:1237-1255) and the (median) midline length; the library
 * applies normalize_image (FilterCache.cpp:21-115): translate(size/2) . scale(individual_image_scale) .
 * translate(len*0.4 | legacy: (-len/2, 0)) . tr, then cv::warpAffine INTER_LINEAR in OpenCV's 8-bit fixed point. */
int trexhip_crops_transformed_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                                     const float* transforms, const float* midline_lengths, int32_t use_legacy,
                                     float image_scale, int32_t difference);
/* the same with tr = Midline::transform(posture | legacy) built from trexhip_midline_device's info (device pointer, pooled
 * order; blobs without a midline get an all-zero crop: diff_image returns nullptr for them, FilterCache.cpp:268-270).
 * midline_lengths: host array, the individuals' median midline length per blob (FilterCache.cpp:272), or NULL to use
 * each blob's own Midline::len(). */
int trexhip_crops_posture_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                                 const trexhip_midline_info* d_midline_info, const float* midline_lengths, int32_t use_legacy,
                                 float image_scale, int32_t difference);

/* ---- identity network (V118_3) -------------------------------------------------------------
 * VINetwork::load_weights (ml/VisualIdentification.cpp) / visual_recognition_torch.py:841-921: takes the
 * flat fp32 blob described in trex_amd/weights.py (state_dict order; tools/convert_weights.py makes it
 * from a TRex <base>_dict.pth).  BatchNorm is folded and the tensors repacked on load. */
int trexhip_load_weights(trexhip_ctx* ctx, const void* blob, size_t bytes);
int trexhip_num_classes(trexhip_ctx* ctx);
int trexhip_network_channels(trexhip_ctx* ctx);   /* channels of the crops the loaded network expects (1 or 3); 0 without weights */
/* arithmetic of conv1..fc1.  All modes but BF16X3 meet the 1e-4 softmax bar against the fp32 reference network:
 *   TREXHIP_CNN_FP16X3 (default): every fp32 operand as two fp16 pieces (22 mantissa bits), 3 piece products per product on the
 *       fp16 matrix cores, fp32 accumulate; an activation outside the fp16 range raises a device flag and the layer stack is
 *       re-run by the BF16X6 kernels (never a silent wrong answer);
 *   TREXHIP_CNN_BF16X6: three bf16 pieces, 6 piece products;  TREXHIP_CNN_FP32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *   TREXHIP_CNN_BF16X3: three piece products only (~2^-16 relative per product) -- for experiments. */
enum { TREXHIP_CNN_FP32 = 0, TREXHIP_CNN_BF16X6 = 1, TREXHIP_CNN_BF16X3 = 2, TREXHIP_CNN_FP16X3 = 3 };
int trexhip_set_identity_precision(trexhip_ctx* ctx, int32_t mode);
/* VINetwork::probabilities (ml/VisualIdentification.cpp:440-458) -> predict_numpy
 * (visual_recognition_torch.py:290-352): crops are uint8 NHWC [n][80][80][C] (values 0..255, no
 * scaling), probs is [n][classes] float32 softmax rows.  d_logits may be NULL. */
int trexhip_identify_device(trexhip_ctx* ctx, const uint8_t* d_crops, int32_t n, float* d_probs, float* d_logits);
int trexhip_identify(trexhip_ctx* ctx, const uint8_t* crops, int32_t n, float* probs);
/* What the fp16 range guard of the LAST identify call on this context did (waits for the context's stream): *rerun_crops = crops
 * whose layer stack was re-run by the BF16X6 kernels (0 = none; the whole batch when the flag came from a kernel that does not
 * know the crop: *whole_batch = 1 -- also when other crops WERE named in the same batch).  Per crop since round 5: one out-of-range crop no
 * longer re-runs the batch.  Two documented exceptions to "a crop's probabilities do not depend on its batch": (1) the fused conv1 + conv2 kernels
 * attribute an out-of-range value to the crops that share its pass -- the crop itself and at most one neighbour in the batch --, and (2) more than
 * 1024 named crops re-run the whole batch; a crop that is re-run although it was in range gets the BF16X6 result, which differs from its FP16X3
 * result by about 2e-6 on the softmax (both inside the 1e-4 bar).  The answer refers to the last identify call even if
 * trexhip_set_identity_precision was called since; zeros before the first call and behind a call in another precision mode.  No reference
 * counterpart (the reference's torch network has no range limit); either pointer may be NULL. */
int trexhip_identify_guard_stats(trexhip_ctx* ctx, uint32_t* rerun_crops, uint32_t* whole_batch);

/* ---- multi-GPU hand-off --------------------------------------------------------------------
 * Fixed-size per-blob identity table of the last batch, written to caller-owned device memory
 * (e.g. a torch tensor that is then all-gathered over RCCL/xGMI to rank 0, whose sequential matcher
 * consumes it -- Tracker::predicted, tracking/Tracker.cpp:237-247).  Row = 8 x u32 header
 * {global frame index = frame_base + frame, pv::bid, n_pixels, x0|y0<<16, x1|y1<<16, centroid x (f32),
 * centroid y (f32), valid} followed by `classes` float probabilities.  Rows >= n_blobs are zeroed. */
int trexhip_export_id_table_device(trexhip_ctx* ctx, const float* d_probs, int32_t n_blobs, int32_t classes,
                                   uint32_t frame_base, void* d_table, int32_t max_rows);
/* the full record of one blob for the all-gather: row = 16 x u32 {the 8 header words above, central second moments per pixel
 * mu20, mu11, mu02 (f32), Midline::len / angle / offset.x / offset.y (f32) and the midline status (-1 when no posture is handed in)}
 * + `classes` probabilities + midline_resolution x {x, y, height} (f32) of the normalised midline (zero unless status 0).
 * d_midline / d_midline_info = outputs of trexhip_midline_device, or both NULL. */
int trexhip_export_id_table_ex_device(trexhip_ctx* ctx, const float* d_probs, int32_t n_blobs, int32_t classes, uint32_t frame_base,
                                      const float* d_midline, const trexhip_midline_info* d_midline_info, int32_t midline_resolution,
                                      void* d_table, int32_t max_rows);

/* The collective of the frame-sharded path, owned by the library: one process per GPU, frames dealt to ranks in blocks, and after
 * every block each rank's table goes to rank 0 (grouped ncclSend / ncclRecv over RCCL -- xGMI inside a node -- on the context's
 * stream, i.e. ordered behind trexhip_export_id_table*_device): a GATHER, because only rank 0's sequential matcher reads the tables.
 *   trexhip_comm_unique_id   rank 0: 128 bytes (ncclUniqueId) to hand to every rank by whatever side channel the host program has
 *   trexhip_comm_create      every rank, collectively (ncclCommInitRank on the context's device); world = 1 needs no id and no RCCL
 *   trexhip_comm_gather_device  `bytes` from every rank's d_send land at d_recv_rank0 + rank * bytes on rank 0 (world x bytes there;
 *                            ignored on the other ranks).  Enqueued on the context's stream; trexhip_synchronize waits for it.
 *   trexhip_comm_gather_device_on  the same on the stream of ANOTHER context of the same device: software-pipelined contexts (one per
 *                            batch in flight) share ONE communicator -- one ncclCommInitRank per process instead of one per context.
 *                            RCCL orders the operations of a communicator by their issue order, so every rank must issue its gathers in
 *                            the same sequence (one host thread driving the contexts in a fixed rotation does).
 * RCCL is loaded with dlopen on first use (an instance already in the process is shared). */
typedef struct trexhip_comm trexhip_comm;
int trexhip_comm_unique_id(void* id128);
int trexhip_comm_create(trexhip_ctx* ctx, const void* id128, int32_t rank, int32_t world, trexhip_comm** out);
void trexhip_comm_destroy(trexhip_comm* comm);
int trexhip_comm_rank(trexhip_comm* comm);
int trexhip_comm_world(trexhip_comm* comm);
int trexhip_comm_gather_device(trexhip_comm* comm, const void* d_send, size_t bytes, void* d_recv_rank0);
int trexhip_comm_gather_device_on(trexhip_comm* comm, trexhip_ctx* stream_ctx, const void* d_send, size_t bytes, void* d_recv_rank0);
/* collective health check: an all-reduce (sum) of 1 over the communicator -> the number of ranks that took part (= world when every
 * rank of the frame-sharded job is alive and on this communicator); synchronous, on the communicator's context's stream */
int trexhip_comm_count_ranks(trexhip_comm* comm, int32_t* ranks_seen);

/* live HIP-event timing of the dominant kernels on the ctx stream (bench.py roofline):
 * stage ids TREXHIP_STAGE_* ; returns accumulated milliseconds and launch count since reset */
enum { TREXHIP_STAGE_ROWS = 0, TREXHIP_STAGE_SEGMENT_ALL = 1, TREXHIP_STAGE_CONV2 = 2, TREXHIP_STAGE_CONV3 = 3,
       TREXHIP_STAGE_CNN_ALL = 4, TREXHIP_STAGE_CROPS = 5, TREXHIP_STAGE_POSTURE = 6,
       /* host-pointer entry points (trexhip_segment, trexhip_segment_color): per FRAME, always collected -- host milliseconds spent copying
        * pageable tiles into the pinned ring, and DMA milliseconds (HIP events on the copy stream).  The two legs overlap each other. */
       TREXHIP_STAGE_UPLOAD_COPY = 8, TREXHIP_STAGE_UPLOAD_DMA = 9, TREXHIP_STAGE_COUNT = 10 };
int trexhip_profile_enable(trexhip_ctx* ctx, int32_t on);
int trexhip_profile_read(trexhip_ctx* ctx, int32_t stage, double* total_ms, int64_t* launches);
int trexhip_profile_reset(trexhip_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Training step of the identity network.
 * Replaces the body of the batch loop of train(), Application/src/tracker/python/visual_recognition_torch.py:1137-1158
 * (forward in training mode, nn.CrossEntropyLoss, backward, torch.optim.Adam(lr).step -- criterion / optimizer :1420-1421), for
 * V118_3 (visual_identification_network_torch.py:184-258) in fp32: the reference's arithmetic on every device but 'cuda', where it
 * additionally wraps the step in autocast + GradScaler (:1066-1072).  What stays on the host side of the boundary, as in the
 * reference: the data loader with its augmentation (:158-188, :1325-1336), epochs, validation, ReduceLROnPlateau (-> set_lr) and
 * early stopping (:1160-1283).
 *   trexhip_trainer_create     weights = the blob of trexhip_load_weights (state_dict order, running statistics included); the
 *                              trainer keeps parameters, gradients and Adam moments in HBM
 *   trexhip_train_step_device  d_inputs [n][80][80][channels] float32 in [0, 255] (NHWC, what TRexImageDataset yields), d_targets
 *                              [n] class indices.  d_keep_masks: null = the library draws the dropout masks (counter-based hash of
 *                              seed, step, index); else n*16 + n*64 + n*128 + n*100 bytes, 1 = keep: the masks of Dropout2d after
 *                              block 1, 2, 3 ([n][C]) and of the Dropout after fc1 ([n][100]) -- how the parity tests inject what
 *                              the reference drew.  loss / correct (host, optional): mean cross entropy and the number of samples
 *                              whose arg-max equals the target; passing either makes the call synchronise.  A target outside
 *                              0..classes-1 (train() asserts it, :1112) is flagged by the device: that call (if it synchronises)
 *                              and every later read / export return TREXHIP_E_INVALID.  The step is ordered on the context's
 *                              stream like every other call; inside, weight packing, mask drawing and the weight gradients run on a
 *                              second stream owned by the trainer and join the context's stream in front of the parameter update
 *   trexhip_trainer_export     the current weights as a blob for trexhip_load_weights (the reference hands its state_dict back)
 *   trexhip_trainer_read       one tensor in torch's layout; tensor = index in state_dict order (0 conv1.weight ... 23 fc2.bias,
 *                              running statistics included), kind 0 parameter, 1 gradient of the last step, 2 / 3 Adam moments
 * ------------------------------------------------------------------------------------------------ */
typedef struct trexhip_trainer trexhip_trainer;
typedef struct {
    int32_t max_batch;          /* largest n of a step (VINetwork: 64..128, ml/VisualIdentification.cpp:112-118)               */
    float lr;                   /* learning_rate, 0.001 (visual_identification_network.py:151)                                */
    float beta1, beta2, eps;    /* torch.optim.Adam defaults 0.9, 0.999, 1e-8                                                */
    float bn_momentum;          /* nn.BatchNorm2d default 0.1                                                                */
    float dropout;              /* 0.05 for all four dropout layers (visual_identification_network_torch.py:189-208)         */
    int32_t precision;          /* arithmetic of the conv2 / conv3 forward, data-gradient and weight-gradient convolutions: 0 = fp16 two-piece split on the
                                   16-bit matrix cores (22-bit operands, fp32 accumulate, per-tensor power-of-two scales: the inference
                                   path's arithmetic), 1 = exact fp32 MFMA.  Everything else is fp32 either way                     */
    uint64_t seed;              /* of the library's own dropout masks                                                        */
} trexhip_train_params;
size_t trexhip_weight_blob_bytes(int32_t classes, int32_t channels);
int trexhip_trainer_create(trexhip_ctx* ctx, const void* blob, size_t bytes, const trexhip_train_params* params, trexhip_trainer** out);
void trexhip_trainer_destroy(trexhip_trainer* trainer);
int trexhip_trainer_set_lr(trexhip_trainer* trainer, float lr);
int64_t trexhip_trainer_steps(trexhip_trainer* trainer);
int trexhip_train_step_device(trexhip_trainer* trainer, const float* d_inputs, const int32_t* d_targets, int32_t n, const uint8_t* d_keep_masks,
                              float* loss, int32_t* correct);
/* model.eval() forward + mean cross entropy + arg-max count of one validation batch (train(), :1171-1190: what ReduceLROnPlateau and the
 * early-stopping callback are fed); running statistics, nothing dropped; changes nothing in the trainer */
int trexhip_train_eval_device(trexhip_trainer* trainer, const float* d_inputs, const int32_t* d_targets, int32_t n, float* loss, int32_t* correct);
int trexhip_train_eval(trexhip_trainer* trainer, const float* inputs, const int32_t* targets, int32_t n, float* loss, int32_t* correct);
/* the same step from host memory (what the reference's DataLoader yields); targets are range-checked like train() does (:1112) */
int trexhip_train_step(trexhip_trainer* trainer, const float* inputs, const int32_t* targets, int32_t n, const uint8_t* keep_masks, float* loss,
                       int32_t* correct);
int trexhip_trainer_read(trexhip_trainer* trainer, int32_t tensor, int32_t kind, float* out, size_t count);
int trexhip_trainer_export(trexhip_trainer* trainer, void* blob, size_t capacity, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
