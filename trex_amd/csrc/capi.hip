// capi.hip -- C ABI of libtrexhip (include/trexhip.h): context, memory, background, segment, fetch.
#include "internal.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

namespace trexhip {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

static void stage_fold(Stage& s, bool all) {
    // fold finished intervals into the totals; `all` waits for the outstanding ones too
    size_t keep = 0;
    for (size_t i = 0; i < s.pending.size(); ++i) {
        EvPair p = s.pending[i];
        hipError_t q = all ? hipEventSynchronize(p.b) : hipEventQuery(p.b);
        float ms = 0.f;
        if (q == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            s.total_ms += ms; s.launches++;
            s.freelist.push_back(p);
        } else if (q == hipErrorNotReady) {
            s.pending[keep++] = p;
        } else {
            (void)hipGetLastError();
            s.freelist.push_back(p);
        }
    }
    s.pending.resize(keep);
}
void stage_begin(trexhip_ctx* ctx, int stage) {
    if (!ctx->profiling) return;
    Stage& s = ctx->stages[stage];
    if (s.pending.size() > 1024) stage_fold(s, false);
    EvPair p;
    if (!s.freelist.empty()) { p = s.freelist.back(); s.freelist.pop_back(); }
    else {
        // timing events only: no system-scope fence when one is recorded (hipEventDisableSystemFence -- "can be used for events that are only being used
        // to measure timing ... avoiding the cost of cache writeback and invalidation, and the performance impact of those actions on the execution of
        // following work", hip_runtime_api.h).  A default event between the pixel pass and the labelling kernel wrote back and invalidated the L2 the
        // labelling kernel was about to read (round 6: the detect pass read 280 us through its own timers and 270 us on the wall clock without them)
        static const unsigned flags = std::getenv("TREXHIP_TIMER_EVENT_FLAGS") ? (unsigned)std::strtoul(std::getenv("TREXHIP_TIMER_EVENT_FLAGS"), nullptr, 0) : (unsigned)hipEventDisableSystemFence;
        hipEventCreateWithFlags(&p.a, flags); hipEventCreateWithFlags(&p.b, flags);
    }
    hipEventRecord(p.a, ctx->stream);
    s.cur = p;
}
void stage_end(trexhip_ctx* ctx, int stage) {
    if (!ctx->profiling) return;
    Stage& s = ctx->stages[stage];
    if (!s.cur.a) return;
    hipEventRecord(s.cur.b, ctx->stream);
    s.pending.push_back(s.cur);
    s.cur = EvPair();
}
void stage_read(trexhip_ctx* ctx, int stage) { stage_fold(ctx->stages[stage], true); }
void stage_free(trexhip_ctx* ctx) {
    for (auto& s : ctx->stages) {
        stage_fold(s, true);
        for (auto& p : s.freelist) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
        s.freelist.clear();
    }
}

template <typename T>
static int dmalloc(T** p, size_t count) {
    if (count == 0) count = 1;
    TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    return TREXHIP_OK;
}
template <typename T>
static int hmalloc(T** p, size_t count) {
    if (count == 0) count = 1;
    TH_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(p), count * sizeof(T), hipHostMallocDefault));
    return TREXHIP_OK;
}

static int fill_cfg(trexhip_ctx* ctx) {
    const trexhip_params& p = ctx->p;
    SegCfg& c = ctx->cfg;
    c.W = p.width; c.H = p.height; c.B = p.max_batch; c.ctr_frames = p.max_batch; c.R = p.max_runs;
    c.T = p.height * TREXHIP_ROW_SLOT + p.max_runs;
    const int thr = p.threshold < 0 ? -p.threshold : p.threshold;   // abs(threshold), as the reference does
    if (p.threshold_maximum < 255) { c.tmin = thr; c.tmax = p.threshold_maximum; }   // cv::inRange
    else { c.tmin = p.inclusive ? thr : thr + 1; c.tmax = 255; }                      // inclusive (default): |p| < threshold is disregarded; 0: strict, cv::threshold
    c.enable_diff = p.enable_difference; c.absdiff = p.absolute_difference;
    c.invert = p.image_invert; c.zero_bg = p.zero_is_background;
    c.slack = p.connectivity == 4 ? 0 : 1;
    c.n_ranges = p.n_ranges;
    c.sqcm = (float)(p.cm_per_pixel * p.cm_per_pixel);
    for (int i = 0; i < 16; ++i) c.ranges[i] = p.ranges[i];
    c.pool_blobs = (uint32_t)p.max_batch * (uint32_t)p.max_blobs;
    c.pool_runs = (uint32_t)p.max_batch * (uint32_t)p.max_runs;
    c.pool_pixels = (uint32_t)p.max_batch * (uint32_t)p.max_pixels;
    c.cap_blobs = (uint32_t)p.max_blobs; c.cap_pixels = (uint32_t)p.max_pixels;
    return TREXHIP_OK;
}

}  // namespace trexhip

using namespace trexhip;

static void pass2_free(trexhip_ctx* ctx);
extern "C" int trexhip_rethreshold_per_blob_device(trexhip_ctx* ctx, int32_t threshold, const int32_t* d_blob_thresholds, int32_t method,
                                                   const double* size_ranges, int32_t n_ranges);

extern "C" {

int trexhip_abi_version(void) { return TREXHIP_ABI_VERSION; }
const char* trexhip_last_error(void) { return g_error.c_str(); }

void trexhip_default_params(trexhip_params* p, int32_t width, int32_t height) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->device = 0;
    p->width = width; p->height = height;
    p->max_batch = 16;
    p->max_runs = 65536;
    p->max_blobs = 2048;
    p->max_pixels = 1 << 20;
    p->threshold = 15; p->threshold_maximum = 255;
    p->enable_difference = 1; p->absolute_difference = 1;
    p->image_invert = 0; p->zero_is_background = 1;
    // keep iff |diff| >= detect_threshold: the rule the reference documents for the detect stage (core/default_config.cpp:1168,
    // docs/parameters_trex.rst:624-629 "disregards any pixel |p| < threshold") in the words of the track-stage rule (:1167), which
    // Tests/test_pixels.cpp:1026-1059 pins as >=.  0 selects the strict comparison (cv::threshold's THRESH_BINARY).
    p->inclusive = 1;
    p->connectivity = 8;
    p->dilation_size = 0; p->use_closing = 0; p->closing_size = 3;
    p->n_ranges = 0; p->cm_per_pixel = 1.0;
    p->pixel_encoding = TREXHIP_ENC_GRAY;
}

int trexhip_create(const trexhip_params* p, trexhip_ctx** out) {
    if (!p || !out) { set_error("trexhip_create: null argument"); return TREXHIP_E_INVALID; }
    *out = nullptr;
    if (p->width <= 0 || p->height <= 0 || p->width >= 65535 || p->height >= 65535) {
        set_error("trexhip_create: frame size must be in 1..65534 (pv.cpp:601-602)"); return TREXHIP_E_INVALID;
    }
    if (p->max_batch <= 0 || p->max_runs <= 0 || p->max_blobs <= 0 || p->max_pixels <= 0) {
        set_error("trexhip_create: capacities must be positive"); return TREXHIP_E_INVALID;
    }
    if ((uint64_t)p->max_batch * (uint64_t)p->max_pixels > 0xffffffffull || (uint64_t)p->max_batch * (uint64_t)p->max_runs > 0xffffffffull ||
        (uint64_t)p->max_batch * (uint64_t)p->max_blobs > 0xffffffffull) {
        set_error("trexhip_create: max_batch * max_pixels / max_runs / max_blobs must stay below 2^32 (pooled tables are indexed with 32 bits)"); return TREXHIP_E_INVALID;
    }
    if (p->connectivity != 8 && p->connectivity != 4) { set_error("trexhip_create: connectivity must be 4 or 8"); return TREXHIP_E_INVALID; }
    {   // settings of the reference that are not implemented: refuse, never diverge silently
        const struct { int32_t v; const char* name; } off[] = {{p->image_adjust, "image_adjust"}, {p->blur_difference, "blur_difference"},
            {p->equalize_histogram, "equalize_histogram"}, {p->correct_luminance, "correct_luminance"}, {p->use_adaptive_threshold, "use_adaptive_threshold"}};
        for (const auto& o : off)
            if (o.v != 0) {
                set_error(std::string("trexhip_create: ") + o.name + " is not implemented by this backend (RawProcessing's optional pre-processing); switch it off or use the CPU detector");
                return TREXHIP_E_UNSUPPORTED;
            }
    }
    if (p->n_ranges < 0 || p->n_ranges > 8) { set_error("trexhip_create: n_ranges must be 0..8"); return TREXHIP_E_INVALID; }
    if ((p->use_closing && (p->closing_size < 1 || p->closing_size > 15)) || p->dilation_size > 7 || p->dilation_size < -7) {
        set_error("trexhip_create: structuring elements larger than 15x15 are not supported (closing_size <= 15, |dilation_size| <= 7)");
        return TREXHIP_E_UNSUPPORTED;
    }
    if (p->pixel_encoding < 0 || p->pixel_encoding > 2) { set_error("trexhip_create: pixel_encoding must be TREXHIP_ENC_GRAY / R3G3B2 / RGB8"); return TREXHIP_E_INVALID; }
    if (p->pixel_encoding != TREXHIP_ENC_GRAY && p->image_invert) {
        set_error("trexhip_create: image_invert with a colour pixel_encoding is not supported"); return TREXHIP_E_UNSUPPORTED;
    }
    int ndev = 0;
    TH_CHECK_HIP(hipGetDeviceCount(&ndev));
    if (p->device < 0 || p->device >= ndev) { set_error("trexhip_create: no such HIP device"); return TREXHIP_E_DEVICE; }
    TH_CHECK_HIP(hipSetDevice(p->device));
    trexhip_ctx* ctx = new (std::nothrow) trexhip_ctx();
    if (!ctx) { set_error("out of host memory"); return TREXHIP_E_NOMEM; }
    ctx->p = *p;
    ctx->pix_ch = p->pixel_encoding == TREXHIP_ENC_RGB8 ? 3 : 1;
    fill_cfg(ctx);
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && cus > 0) ctx->n_cus = cus; }
    // tuning knobs of the default build only choose between schedules / tilings that give identical results.  The knobs that
    // stop a kernel half-way or skip work (profiling aids) exist only in a -DTREXHIP_DEV_KNOBS build.
    if (const char* e = std::getenv("TREXHIP_ROWS_ORDER")) ctx->tune_rows_order = std::atoi(e) & (1 | 4 | 8 | 1024 | 2048);
    if (const char* e = std::getenv("TREXHIP_ROWS_K")) ctx->tune_rows_k = std::atoi(e);
    if (const char* e = std::getenv("TREXHIP_CONV_GEOM")) ctx->tune_conv_geom = std::atoi(e);
#ifdef TREXHIP_DEV_KNOBS
    if (const char* e = std::getenv("TREXHIP_ROWS_ORDER")) ctx->tune_rows_order = std::atoi(e);
    if (const char* e = std::getenv("TREXHIP_CCL_STOP")) ctx->tune_ccl_stop = std::atoi(e);
#endif
    if (const char* e = std::getenv("TREXHIP_SEG_GROUPS")) ctx->tune_seg_groups = std::atoi(e);
    if (const char* e = std::getenv("TREXHIP_SEG_SCHEME")) ctx->tune_seg_scheme = std::atoi(e);
    if (const char* e = std::getenv("TREXHIP_CCL_BANDS")) ctx->tune_ccl_bands = std::atoi(e);     // same tables whatever the bands
    if (const char* e = std::getenv("TREXHIP_CCL_INST")) ctx->tune_ccl_inst = std::atoi(e);      // which k_ccl_lds instance goes first: same results either way
    if (const char* e = std::getenv("TREXHIP_ROWS_BLOCKS")) { ctx->tune_rows_blocks = std::atoi(e) > 0 ? std::atoi(e) : 8192; ctx->tune_rows_blocks_set = true; }
    const size_t B = p->max_batch, H = p->height, W = p->width, R = p->max_runs, NB = p->max_blobs, P = p->max_pixels;
    int rc = TREXHIP_OK;
#define TRY(x) do { if (rc == TREXHIP_OK) rc = (x); } while (0)
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); delete ctx; return TREXHIP_E_DEVICE; }
    ctx->stream = ctx->own_stream;
    TRY(dmalloc(&ctx->d_bg, H * W + 16));
    TRY(dmalloc(&ctx->d_ctr, B * TREXHIP_CTR_STRIDE + 4));
    if (rc == TREXHIP_OK && hipMemset(ctx->d_ctr, 0, sizeof(uint32_t) * (B * TREXHIP_CTR_STRIDE + 4)) != hipSuccess) { set_error("hipMemset of the counters failed"); rc = TREXHIP_E_DEVICE; }   // every pass leaves the counters zero (launch_segment)
    TRY(dmalloc(&ctx->d_band_fail, B));
    if (rc == TREXHIP_OK && hipMemset(ctx->d_band_fail, 0, sizeof(uint32_t) * B) != hipSuccess) { set_error("hipMemset of the band flags failed"); rc = TREXHIP_E_DEVICE; }
    TRY(dmalloc(&ctx->d_row_cnt, B * H));
    TRY(dmalloc(&ctx->d_row_off, B * H));
    TRY(dmalloc(&ctx->d_row_base, B * (H + 1)));
    TRY(dmalloc(&ctx->d_tmp_runs, B * (H * TREXHIP_ROW_SLOT + R)));
    TRY(dmalloc(&ctx->d_raster, B * R));
    TRY(dmalloc(&ctx->d_parent, B * R));
    TRY(dmalloc(&ctx->d_root_ord, B * R));
    TRY(dmalloc(&ctx->d_cnt_runs, B * R));
    TRY(dmalloc(&ctx->d_cnt_px, B * R));
    TRY(dmalloc(&ctx->d_cur_run, B * R));
    TRY(dmalloc(&ctx->d_pix_begin, B * R));
    TRY(dmalloc(&ctx->d_blob_map, B * R));
    TRY(dmalloc(&ctx->d_info, B));
    TRY(dmalloc(&ctx->d_blobs, B * NB));
    TRY(dmalloc(&ctx->d_blob_frame, B * NB));
    TRY(dmalloc(&ctx->d_runs, B * R));
    TRY(dmalloc(&ctx->d_pixels, B * P * ctx->pix_ch));
    if (p->use_closing || p->dilation_size != 0) {
        const size_t WBw = (W + 31) / 32;
        TRY(dmalloc(&ctx->d_bits[0], B * H * WBw + 4));
        TRY(dmalloc(&ctx->d_bits[1], B * H * WBw + 4));
    }
    TRY(hmalloc(&ctx->h_info, B));
    TRY(hmalloc(&ctx->h_totals, 4));
    TRY(hmalloc(&ctx->h_ccl_hint, 2));
    if (rc == TREXHIP_OK) ctx->h_ccl_hint[0] = ctx->h_ccl_hint[1] = 0u;
    TRY(hmalloc(&ctx->h_blobs, B * NB));
    TRY(hmalloc(&ctx->h_runs, B * R));
    TRY(hmalloc(&ctx->h_pixels, B * P * ctx->pix_ch));
#undef TRY
    if (rc != TREXHIP_OK) { trexhip_destroy(ctx); return rc; }
    *out = ctx;
    return TREXHIP_OK;
}

void trexhip_destroy(trexhip_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->p.device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    net_free(ctx);
    pass2_free(ctx);
    void* dev[] = {ctx->d_bg, ctx->d_staging, ctx->d_ctr, ctx->d_band_fail, ctx->d_row_cnt, ctx->d_row_off, ctx->d_row_base, ctx->d_tmp_runs,
                   ctx->d_raster, ctx->d_parent, ctx->d_root_ord, ctx->d_cnt_runs, ctx->d_cnt_px, ctx->d_cur_run,
                   ctx->d_pix_begin, ctx->d_blob_map, ctx->d_info, ctx->d_blobs, ctx->d_blob_frame, ctx->d_runs, ctx->d_pixels, ctx->d_color, ctx->d_bits[0], ctx->d_bits[1], ctx->d_warp, ctx->d_bg_color, ctx->d_len, ctx->d_auto};
    for (void* p : dev) if (p) hipFree(p);
    upload_free(ctx);
    void* host[] = {ctx->h_info, ctx->h_totals, ctx->h_blobs, ctx->h_runs, ctx->h_pixels, ctx->h_staging, ctx->h_color, ctx->h_ccl_hint};
    for (void* p : host) if (p) hipHostFree(p);
    stage_free(ctx);
    if (ctx->aux_stream) {
        (void)hipStreamDestroy(ctx->aux_stream);
        for (hipEvent_t e : ctx->ev_grp) if (e) (void)hipEventDestroy(e);
    }
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int trexhip_pixel_channels(trexhip_ctx* ctx) { return ctx ? ctx->pix_ch : 0; }

int trexhip_device_alloc(trexhip_ctx* ctx, size_t bytes, void** out_device_ptr) {
    if (!ctx || !out_device_ptr) { set_error("trexhip_device_alloc: null argument"); return TREXHIP_E_INVALID; }
    *out_device_ptr = nullptr;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    TH_CHECK_HIP(hipMalloc(out_device_ptr, bytes ? bytes : 1));
    return TREXHIP_OK;
}

int trexhip_device_free(trexhip_ctx* ctx, void* device_ptr) {
    if (!ctx) { set_error("null ctx"); return TREXHIP_E_INVALID; }
    if (!device_ptr) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    TH_CHECK_HIP(hipFree(device_ptr));
    return TREXHIP_OK;
}

int trexhip_copy_to_host(trexhip_ctx* ctx, void* host_dst, const void* device_src, size_t bytes) {
    if (!ctx || (bytes && (!host_dst || !device_src))) { set_error("trexhip_copy_to_host: null argument"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (bytes) TH_CHECK_HIP(hipMemcpyAsync(host_dst, device_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return TREXHIP_OK;
}

int trexhip_copy_to_device(trexhip_ctx* ctx, void* device_dst, const void* host_src, size_t bytes) {
    if (!ctx || (bytes && (!device_dst || !host_src))) { set_error("trexhip_copy_to_device: null argument"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (bytes) TH_CHECK_HIP(hipMemcpyAsync(device_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return TREXHIP_OK;
}

int trexhip_get_live_params(trexhip_ctx* ctx, trexhip_live_params* out) {
    if (!ctx || !out) { set_error("trexhip_get_live_params: null argument"); return TREXHIP_E_INVALID; }
    const trexhip_params& p = ctx->p;
    std::memset(out, 0, sizeof(*out));
    out->threshold = p.threshold; out->threshold_maximum = p.threshold_maximum; out->inclusive = p.inclusive;
    out->enable_difference = p.enable_difference; out->absolute_difference = p.absolute_difference;
    out->image_invert = p.image_invert; out->zero_is_background = p.zero_is_background;
    out->n_ranges = p.n_ranges; out->cm_per_pixel = p.cm_per_pixel;
    for (int i = 0; i < 16; ++i) out->ranges[i] = p.ranges[i];
    return TREXHIP_OK;
}

int trexhip_update_params(trexhip_ctx* ctx, const trexhip_live_params* lp) {
    if (!ctx || !lp) { set_error("trexhip_update_params: null argument"); return TREXHIP_E_INVALID; }
    if (lp->n_ranges < 0 || lp->n_ranges > 8) { set_error("trexhip_update_params: n_ranges must be 0..8 (detect_size_filter with more than 8 ranges is not supported)"); return TREXHIP_E_INVALID; }
    if (!(lp->cm_per_pixel > 0.0)) { set_error("trexhip_update_params: cm_per_pixel must be positive"); return TREXHIP_E_INVALID; }
    if (ctx->p.pixel_encoding != TREXHIP_ENC_GRAY && lp->image_invert) {
        set_error("trexhip_update_params: image_invert with a colour pixel_encoding is not supported"); return TREXHIP_E_UNSUPPORTED;
    }
    trexhip_params& p = ctx->p;
    p.threshold = lp->threshold; p.threshold_maximum = lp->threshold_maximum; p.inclusive = lp->inclusive;
    p.enable_difference = lp->enable_difference; p.absolute_difference = lp->absolute_difference;
    p.image_invert = lp->image_invert; p.zero_is_background = lp->zero_is_background;
    p.n_ranges = lp->n_ranges; p.cm_per_pixel = lp->cm_per_pixel;
    for (int i = 0; i < 16; ++i) p.ranges[i] = i < 2 * lp->n_ranges ? lp->ranges[i] : 0.0;
    return fill_cfg(ctx);
}

int trexhip_set_stream(trexhip_ctx* ctx, void* hip_stream) {
    if (!ctx) { set_error("null ctx"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return TREXHIP_OK;
}

int trexhip_set_background(trexhip_ctx* ctx, const uint8_t* gray, int32_t stride) {
    if (!ctx || !gray) { set_error("trexhip_set_background: null argument"); return TREXHIP_E_INVALID; }
    if (stride < ctx->p.width) { set_error("trexhip_set_background: stride < width"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    TH_CHECK_HIP(hipMemcpy2DAsync(ctx->d_bg, ctx->p.width, gray, stride, ctx->p.width, ctx->p.height,
                                  hipMemcpyHostToDevice, ctx->stream));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->has_bg = true;
    return TREXHIP_OK;
}

int trexhip_set_background_device(trexhip_ctx* ctx, const uint8_t* d_gray) {
    if (!ctx || !d_gray) { set_error("trexhip_set_background_device: null argument"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    TH_CHECK_HIP(hipMemcpyAsync(ctx->d_bg, d_gray, (size_t)ctx->p.width * ctx->p.height, hipMemcpyDeviceToDevice, ctx->stream));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->has_bg = true;
    return TREXHIP_OK;
}

static int check_segment_args(trexhip_ctx* ctx, const void* frames, int32_t n) {
    if (!ctx || !frames) { set_error("trexhip_segment: null argument"); return TREXHIP_E_INVALID; }
    if (!ctx->has_bg) { set_error("trexhip_segment: background image not set"); return TREXHIP_E_INVALID; }
    if (n < 0 || n > ctx->p.max_batch) { set_error("trexhip_segment: n outside 0..max_batch"); return TREXHIP_E_INVALID; }
    return TREXHIP_OK;
}

int trexhip_segment_device(trexhip_ctx* ctx, const uint8_t* d_frames, int32_t n) {
    int rc = check_segment_args(ctx, d_frames, n);
    if (rc) return rc;
    if (ctx->pix_ch != 1 || ctx->p.pixel_encoding != TREXHIP_ENC_GRAY) { set_error("trexhip_segment_device: a colour pixel_encoding needs colour input (trexhip_segment_color*)"); return TREXHIP_E_INVALID; }
    ctx->d_color_src = nullptr; ctx->color_ch = 0;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (n == 0) { ctx->last_n = 0; ctx->fetched = false; return TREXHIP_OK; }
    return launch_segment(ctx, d_frames, n);
}

int trexhip_segment(trexhip_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t n) {
    int rc = check_segment_args(ctx, frames, n);
    if (rc) return rc;
    if (ctx->p.pixel_encoding != TREXHIP_ENC_GRAY) { set_error("trexhip_segment: a colour pixel_encoding needs colour input (trexhip_segment_color*)"); return TREXHIP_E_INVALID; }
    ctx->d_color_src = nullptr; ctx->color_ch = 0;
    if (stride < ctx->p.width) { set_error("trexhip_segment: stride < width"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (n == 0) { ctx->last_n = 0; ctx->fetched = false; return TREXHIP_OK; }
    const size_t W = ctx->p.width, H = ctx->p.height;
    if (!ctx->d_staging) {
        rc = dmalloc(&ctx->d_staging, (size_t)ctx->p.max_batch * W * H + 16);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i)
        if (!frames[i]) { set_error("trexhip_segment: null frame pointer"); return TREXHIP_E_INVALID; }
    // pageable frame -> pinned ring slot (host threads) -> HBM, one async DMA per frame overlapping the next frame's staging (upload.hip)
    rc = upload_frames(ctx, frames, n, H, W, (size_t)stride, ctx->d_staging, nullptr);
    if (rc) return rc;
    return launch_segment(ctx, ctx->d_staging, n);
}

int trexhip_synchronize(trexhip_ctx* ctx) {
    if (!ctx) { set_error("null ctx"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return TREXHIP_OK;
}


}  // extern "C"
namespace trexhip {
// A batch of a few frames (TRex's default detect_batch_size is ONE frame per call) leaves the device as ONE launch that writes the frame table, the
// totals and exactly the filled parts of the blob / line / pixel tables into the context's pinned host mirrors, followed by ONE stream synchronize.
// The DMA path below needs two round trips (frame table and totals first, then three copies whose sizes the totals give): five small copies with
// 10-17 us of host latency between them -- more than the detect kernels of one frame take (profiles/r06_batch1_timeline.txt).  4-byte words: every
// table is a multiple of 4 bytes except the pixels, whose tail bytes go out one by one.
__global__ __launch_bounds__(256) void k_export(const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ totals,
                                                const trexhip_blob* __restrict__ blobs, const trexhip_run* __restrict__ runs, const uint8_t* __restrict__ pixels,
                                                trexhip_frame_info* __restrict__ h_info, uint32_t* __restrict__ h_totals, trexhip_blob* __restrict__ h_blobs,
                                                trexhip_run* __restrict__ h_runs, uint8_t* __restrict__ h_pixels, const int n,
                                                const uint32_t pool_blobs, const uint32_t pool_runs, const uint32_t pool_pixels, const int pix_ch) {
    const uint32_t tb = min(totals[0], pool_blobs), tr = min(totals[1], pool_runs), tp = min(totals[2], pool_pixels);
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gstep = (size_t)gridDim.x * 256;
    auto words = [&](const void* src, void* dst, const size_t nw) {
        const uint32_t* a = static_cast<const uint32_t*>(src);
        uint32_t* b = static_cast<uint32_t*>(dst);
        for (size_t i = gtid; i < nw; i += gstep) b[i] = a[i];
    };
    words(info, h_info, (size_t)n * sizeof(trexhip_frame_info) / 4);
    if (gtid < 4) h_totals[gtid] = totals[gtid];
    words(blobs, h_blobs, (size_t)tb * sizeof(trexhip_blob) / 4);
    words(runs, h_runs, (size_t)tr * sizeof(trexhip_run) / 4);
    const size_t pb = (size_t)tp * pix_ch;
    words(pixels, h_pixels, pb / 4);
    if (gtid < (pb & 3)) h_pixels[(pb & ~(size_t)3) + gtid] = pixels[(pb & ~(size_t)3) + gtid];
}
static_assert(sizeof(trexhip_frame_info) % 4 == 0 && sizeof(trexhip_blob) % 4 == 0 && sizeof(trexhip_run) % 4 == 0, "k_export copies 4-byte words");
static constexpr int EXPORT_MAX_FRAMES = 16;
static int export_small(trexhip_ctx* ctx, int n) {
    hipLaunchKernelGGL(k_export, dim3(32), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_ctr + (size_t)ctx->p.max_batch * TREXHIP_CTR_STRIDE, ctx->d_blobs,
                       ctx->d_runs, ctx->d_pixels, ctx->h_info, ctx->h_totals, ctx->h_blobs, ctx->h_runs, ctx->h_pixels, n,
                       ctx->cfg.pool_blobs, ctx->cfg.pool_runs, ctx->cfg.pool_pixels, ctx->pix_ch);
    TH_CHECK_HIP(hipGetLastError());
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return TREXHIP_OK;
}
}  // namespace trexhip
extern "C" {

int trexhip_fetch(trexhip_ctx* ctx, trexhip_batch_result* out) {
    if (!ctx || !out) { set_error("trexhip_fetch: null argument"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    std::memset(out, 0, sizeof(*out));
    const int n = ctx->last_n;
    out->n_frames = n;
    out->frames = ctx->h_info; out->blobs = ctx->h_blobs; out->runs = ctx->h_runs; out->pixels = ctx->h_pixels;
    out->pixel_channels = (uint32_t)ctx->pix_ch; out->reserved_ = 0;
    if (n == 0) return TREXHIP_OK;
    hipStream_t s = ctx->stream;
    static const bool export_env = !(std::getenv("TREXHIP_EXPORT") && std::atoi(std::getenv("TREXHIP_EXPORT")) == 0);
    const bool by_kernel = export_env && n <= EXPORT_MAX_FRAMES;      // a few frames: one launch + one synchronize (k_export)
    if (by_kernel) { int rce = export_small(ctx, n); if (rce) return rce; }
    else {
        TH_CHECK_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(trexhip_frame_info) * n, hipMemcpyDeviceToHost, s));
        TH_CHECK_HIP(hipMemcpyAsync(ctx->h_totals, ctx->d_ctr + (size_t)ctx->p.max_batch * TREXHIP_CTR_STRIDE, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
        TH_CHECK_HIP(hipStreamSynchronize(s));
    }
    bool pending = false;
    for (int i = 0; i < n; ++i) pending |= ctx->h_info[i].reserved[0] == 1u;
    if (pending) {   // frames with more runs than fit in LDS: finish them with the global-memory chain
        int rc2 = launch_pending(ctx);
        if (rc2) return rc2;
        if (by_kernel) { int rce = export_small(ctx, n); if (rce) return rce; }
        else {
            TH_CHECK_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(trexhip_frame_info) * n, hipMemcpyDeviceToHost, s));
            TH_CHECK_HIP(hipMemcpyAsync(ctx->h_totals, ctx->d_ctr + (size_t)ctx->p.max_batch * TREXHIP_CTR_STRIDE, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
            TH_CHECK_HIP(hipStreamSynchronize(s));
        }
    }
    // frames that overflowed the pool reserved nothing valid; clamp the copies to the pools
    const uint32_t tb = ctx->h_totals[0] < ctx->cfg.pool_blobs ? ctx->h_totals[0] : ctx->cfg.pool_blobs;
    const uint32_t tr = ctx->h_totals[1] < ctx->cfg.pool_runs ? ctx->h_totals[1] : ctx->cfg.pool_runs;
    const uint32_t tp = ctx->h_totals[2] < ctx->cfg.pool_pixels ? ctx->h_totals[2] : ctx->cfg.pool_pixels;
    if (!by_kernel) {
        if (tb) TH_CHECK_HIP(hipMemcpyAsync(ctx->h_blobs, ctx->d_blobs, sizeof(trexhip_blob) * tb, hipMemcpyDeviceToHost, s));
        if (tr) TH_CHECK_HIP(hipMemcpyAsync(ctx->h_runs, ctx->d_runs, sizeof(trexhip_run) * tr, hipMemcpyDeviceToHost, s));
        if (tp) TH_CHECK_HIP(hipMemcpyAsync(ctx->h_pixels, ctx->d_pixels, (size_t)tp * ctx->pix_ch, hipMemcpyDeviceToHost, s));
        TH_CHECK_HIP(hipStreamSynchronize(s));
    }
    out->total_blobs = tb; out->total_runs = tr; out->total_pixels = tp;
    ctx->fetched = true;
    int rc = TREXHIP_OK;
    for (int i = 0; i < n; ++i)
        if (ctx->h_info[i].flags) {
            char buf[160];
            std::snprintf(buf, sizeof(buf), "frame %d of the batch exceeded capacity (flags=%u, raw runs=%u): raise max_runs/max_blobs/max_pixels",
                          i, ctx->h_info[i].flags, ctx->h_info[i].n_raw_runs);
            set_error(buf);
            rc = TREXHIP_E_CAPACITY;
        }
    return rc;
}

static int pass2_alloc(trexhip_ctx* ctx) {
    Pass2& q = ctx->pass2;
    if (q.allocated) return TREXHIP_OK;
    const size_t B = ctx->p.max_batch, H = ctx->p.height, R = ctx->p.max_runs, NB = ctx->p.max_blobs, P = ctx->p.max_pixels;
    int rc = TREXHIP_OK;
#define TRY(x) do { if (rc == TREXHIP_OK) rc = (x); } while (0)
    TRY(dmalloc(&q.d_sub_cnt, B * R)); TRY(dmalloc(&q.d_sub_base, B * R)); TRY(dmalloc(&q.d_row_base, B * (H + 1)));
    TRY(dmalloc(&q.d_row_cnt, B * H)); TRY(dmalloc(&q.d_run_parent, B * R)); TRY(dmalloc(&q.d_raster, B * R));
    TRY(dmalloc(&q.d_parent, B * R)); TRY(dmalloc(&q.d_root_ord, B * R)); TRY(dmalloc(&q.d_cnt_runs, B * R));
    TRY(dmalloc(&q.d_cnt_px, B * R)); TRY(dmalloc(&q.d_cur_run, B * R)); TRY(dmalloc(&q.d_pix_begin, B * R));
    TRY(dmalloc(&q.d_blob_map, B * R)); TRY(dmalloc(&q.d_totals, 4)); TRY(dmalloc(&q.d_info, B));
    TRY(dmalloc(&q.d_blobs, B * NB)); TRY(dmalloc(&q.d_blob_frame, B * NB)); TRY(dmalloc(&q.d_runs, B * R)); TRY(dmalloc(&q.d_pixels, B * P * ctx->pix_ch));
    TRY(hmalloc(&q.h_info, B)); TRY(hmalloc(&q.h_totals, 4)); TRY(hmalloc(&q.h_blobs, B * NB)); TRY(hmalloc(&q.h_runs, B * R)); TRY(hmalloc(&q.h_pixels, B * P * ctx->pix_ch));
#undef TRY
    q.allocated = rc == TREXHIP_OK;
    return rc;
}

static void pass2_free(trexhip_ctx* ctx) {
    Pass2& q = ctx->pass2;
    void* dev[] = {q.d_sub_cnt, q.d_sub_base, q.d_row_base, q.d_row_cnt, q.d_run_parent, q.d_raster, q.d_parent, q.d_root_ord, q.d_cnt_runs,
                   q.d_cnt_px, q.d_cur_run, q.d_pix_begin, q.d_blob_map, q.d_totals, q.d_info, q.d_blobs, q.d_blob_frame, q.d_runs, q.d_pixels};
    for (void* p : dev) if (p) hipFree(p);
    void* host[] = {q.h_info, q.h_totals, q.h_blobs, q.h_runs, q.h_pixels};
    for (void* p : host) if (p) hipHostFree(p);
    q = Pass2();
}

int trexhip_rethreshold_device(trexhip_ctx* ctx, int32_t threshold, int32_t method, const double* size_ranges, int32_t n_ranges) {
    return trexhip_rethreshold_per_blob_device(ctx, threshold, nullptr, method, size_ranges, n_ranges);
}

int trexhip_rethreshold_per_blob_device(trexhip_ctx* ctx, int32_t threshold, const int32_t* d_blob_thresholds, int32_t method,
                                        const double* size_ranges, int32_t n_ranges) {
    if (!ctx) { set_error("trexhip_rethreshold_device: null ctx"); return TREXHIP_E_INVALID; }
    if (method < 0 || method > 2) { set_error("trexhip_rethreshold_device: method must be 0 (absolute), 1 (sign) or 2 (none)"); return TREXHIP_E_INVALID; }
    if (n_ranges < 0 || n_ranges > 8 || (n_ranges && !size_ranges)) { set_error("trexhip_rethreshold_device: bad size ranges"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0) { set_error("trexhip_rethreshold_device: no segmented batch"); return TREXHIP_E_INVALID; }
    if (!ctx->fetched) { set_error("trexhip_rethreshold_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    int rc = pass2_alloc(ctx);
    if (rc) return rc;
    return launch_rethreshold(ctx, threshold, method, size_ranges, n_ranges, d_blob_thresholds);
}

int trexhip_fetch_rethreshold(trexhip_ctx* ctx, trexhip_batch_result* out) {
    if (!ctx || !out) { set_error("trexhip_fetch_rethreshold: null argument"); return TREXHIP_E_INVALID; }
    Pass2& q = ctx->pass2;
    if (!q.allocated || q.valid_n == 0) { set_error("trexhip_fetch_rethreshold: no re-thresholded batch"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    std::memset(out, 0, sizeof(*out));
    const int n = q.valid_n;
    out->n_frames = n;
    out->frames = q.h_info; out->blobs = q.h_blobs; out->runs = q.h_runs; out->pixels = q.h_pixels;
    out->pixel_channels = (uint32_t)ctx->pix_ch; out->reserved_ = 0;
    hipStream_t s = ctx->stream;
    TH_CHECK_HIP(hipMemcpyAsync(q.h_info, q.d_info, sizeof(trexhip_frame_info) * n, hipMemcpyDeviceToHost, s));
    TH_CHECK_HIP(hipMemcpyAsync(q.h_totals, q.d_totals, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, s));
    TH_CHECK_HIP(hipStreamSynchronize(s));
    const uint32_t tb = q.h_totals[0] < ctx->cfg.pool_blobs ? q.h_totals[0] : ctx->cfg.pool_blobs;
    const uint32_t tr = q.h_totals[1] < ctx->cfg.pool_runs ? q.h_totals[1] : ctx->cfg.pool_runs;
    const uint32_t tp = q.h_totals[2] < ctx->cfg.pool_pixels ? q.h_totals[2] : ctx->cfg.pool_pixels;
    if (tb) TH_CHECK_HIP(hipMemcpyAsync(q.h_blobs, q.d_blobs, sizeof(trexhip_blob) * tb, hipMemcpyDeviceToHost, s));
    if (tr) TH_CHECK_HIP(hipMemcpyAsync(q.h_runs, q.d_runs, sizeof(trexhip_run) * tr, hipMemcpyDeviceToHost, s));
    if (tp) TH_CHECK_HIP(hipMemcpyAsync(q.h_pixels, q.d_pixels, (size_t)tp * ctx->pix_ch, hipMemcpyDeviceToHost, s));
    TH_CHECK_HIP(hipStreamSynchronize(s));
    q.fetched = true;
    out->total_blobs = tb; out->total_runs = tr; out->total_pixels = tp;
    for (int i = 0; i < n; ++i)
        if (q.h_info[i].flags) { set_error("a frame exceeded capacity during re-threshold: raise max_runs/max_blobs/max_pixels"); return TREXHIP_E_CAPACITY; }
    return TREXHIP_OK;
}

int trexhip_device_view_get(trexhip_ctx* ctx, trexhip_device_view* out) {
    if (!ctx || !out) { set_error("trexhip_device_view_get: null argument"); return TREXHIP_E_INVALID; }
    out->frames = ctx->d_info; out->blobs = ctx->d_blobs; out->runs = ctx->d_runs; out->pixels = ctx->d_pixels;
    out->totals = ctx->d_ctr + (size_t)ctx->p.max_batch * TREXHIP_CTR_STRIDE; out->blob_frame = ctx->d_blob_frame;
    return TREXHIP_OK;
}

int trexhip_debug_read(trexhip_ctx* ctx, unsigned long long* out, int32_t n) {   /* dev only: phase stamps of k_ccl_lds */
    if (!ctx || !out) return TREXHIP_E_INVALID;
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    TH_CHECK_HIP(hipMemcpy(out, ctx->d_cnt_px, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    return TREXHIP_OK;
}

int trexhip_profile_enable(trexhip_ctx* ctx, int32_t on) {
    if (!ctx) { set_error("null ctx"); return TREXHIP_E_INVALID; }
    ctx->profiling = on != 0;
    return TREXHIP_OK;
}

int trexhip_profile_read(trexhip_ctx* ctx, int32_t stage, double* total_ms, int64_t* launches) {
    if (!ctx || stage < 0 || stage >= TREXHIP_STAGE_COUNT) { set_error("trexhip_profile_read: bad argument"); return TREXHIP_E_INVALID; }
    if (stage == TREXHIP_STAGE_UPLOAD_COPY || stage == TREXHIP_STAGE_UPLOAD_DMA) {     // host-input legs (upload.hip), per frame
        Uploader& u = ctx->up;
        if (stage == TREXHIP_STAGE_UPLOAD_DMA)
            for (int k = 0; k < UP_SLOTS; ++k)
                if (u.busy[k] && hipEventQuery(u.ev_done[k]) == hipSuccess) {
                    float ms = 0.f;
                    if (hipEventElapsedTime(&ms, u.ev_start[k], u.ev_done[k]) == hipSuccess) { u.dma_ms += ms; u.dma_n += u.frames_in[k]; }
                    u.busy[k] = false;
                }
        if (total_ms) *total_ms = stage == TREXHIP_STAGE_UPLOAD_COPY ? u.copy_ms : u.dma_ms;
        if (launches) *launches = stage == TREXHIP_STAGE_UPLOAD_COPY ? u.copy_n : u.dma_n;
        return TREXHIP_OK;
    }
    stage_read(ctx, stage);
    Stage& s = ctx->stages[stage];
    if (total_ms) *total_ms = s.total_ms;
    if (launches) *launches = s.launches;
    return TREXHIP_OK;
}

int trexhip_profile_reset(trexhip_ctx* ctx) {
    if (!ctx) { set_error("null ctx"); return TREXHIP_E_INVALID; }
    for (int i = 0; i < TREXHIP_STAGE_COUNT; ++i) { stage_read(ctx, i); ctx->stages[i].total_ms = 0.0; ctx->stages[i].launches = 0; }
    for (int k = 0; k < UP_SLOTS; ++k)                       // uploads still in flight belong to the period before the reset
        if (ctx->up.busy[k]) { (void)hipEventSynchronize(ctx->up.ev_done[k]); ctx->up.busy[k] = false; }
    ctx->up.copy_ms = ctx->up.dma_ms = 0.0; ctx->up.copy_n = ctx->up.dma_n = 0;
    return TREXHIP_OK;
}

}  // extern "C"
