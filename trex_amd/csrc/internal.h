// internal.h -- shared declarations of libtrexhip (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/trexhip.h"

#define TREXHIP_ROW_SLOT 16
#define TREXHIP_CTR_STRIDE 32

namespace trexhip {

void set_error(const std::string& msg);

#define TH_CHECK_HIP(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::trexhip::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));          \
            return TREXHIP_E_DEVICE;                                                          \
        }                                                                                     \
    } while (0)

// wave-uniform constants handed to every segment kernel by value
struct SegCfg {
    int W, H, B;          // frame size, frames in this launch
    int ctr_frames;       // frames the per-frame counter array was sized for (max_batch): the pooled totals live behind them
    int R;                // raw run capacity per frame
    int T;                // entries per frame in the tmp run array = H*ROW_SLOT + R
    int tmin;             // smallest difference value that passes the threshold
    int tmax;             // largest passing value (255 unless threshold_maximum < 255)
    int enable_diff, absdiff, invert, zero_bg;
    int slack;            // 1 for 8-connectivity, 0 for 4
    int n_ranges;
    float sqcm;
    double ranges[16];
    uint32_t pool_blobs, pool_runs, pool_pixels;
    uint32_t cap_blobs, cap_pixels;   // per frame (max_blobs, max_pixels): a frame beyond them fails alone, before it reserves pooled space
};

// live kernel timing: event pairs are recorded on the ctx stream and only read (folded) on
// trexhip_profile_read, so enabling profiling never adds a host sync to the timed region
struct EvPair { hipEvent_t a = nullptr, b = nullptr; };
struct Stage {
    std::vector<EvPair> pending, freelist;
    EvPair cur;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace trexhip

namespace trexhip {
// second CCL pass (track-stage re-threshold): its own run-level state and pooled outputs
struct Pass2 {
    bool allocated = false;
    int valid_n = 0;
    bool fetched = false;               // the host tables below hold the last re-threshold batch
    uint32_t *d_sub_cnt = nullptr, *d_sub_base = nullptr, *d_row_base = nullptr, *d_row_cnt = nullptr, *d_run_parent = nullptr;
    trexhip_run* d_raster = nullptr;
    uint32_t *d_parent = nullptr, *d_root_ord = nullptr, *d_cnt_runs = nullptr, *d_cnt_px = nullptr, *d_cur_run = nullptr,
             *d_pix_begin = nullptr, *d_totals = nullptr, *d_blob_frame = nullptr;
    int32_t* d_blob_map = nullptr;
    trexhip_frame_info* d_info = nullptr;
    trexhip_blob* d_blobs = nullptr;
    trexhip_run* d_runs = nullptr;
    uint8_t* d_pixels = nullptr;
    trexhip_frame_info* h_info = nullptr;
    uint32_t* h_totals = nullptr;
    trexhip_blob* h_blobs = nullptr;
    trexhip_run* h_runs = nullptr;
    uint8_t* h_pixels = nullptr;
};
}

namespace trexhip {
// host tiles -> HBM (upload.hip): ring of pinned slots, copy stream, host copy threads, timings of the two legs
enum { UP_SLOTS = 3 };
static constexpr size_t UP_CHUNK_BYTES = (size_t)32 << 20;   // frames per DMA: about this many bytes
struct Uploader {
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_start[UP_SLOTS] = {}, ev_done[UP_SLOTS] = {};
    bool busy[UP_SLOTS] = {};
    int frames_in[UP_SLOTS] = {};
    uint8_t* ring = nullptr;            // UP_SLOTS x slot_bytes, pinned
    size_t slot_bytes = 0;
    void* pool = nullptr;               // CopyPool
    double copy_ms = 0.0, dma_ms = 0.0;
    int64_t copy_n = 0, dma_n = 0;
};
}

struct trexhip_ctx {
    trexhip_params p;
    trexhip::SegCfg cfg;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool has_bg = false;
    int last_n = 0;
    int batch_invert = 0, batch_zero_bg = 0;      // image_invert / zero_is_background the LAST segmented batch was taken with: its crops, re-threshold and
                                                  // split search keep them although trexhip_update_params may have changed the live settings since
    bool fetched = true;

    // device memory (layout: DESIGN.md "Data layout in HBM")
    uint8_t* d_bg = nullptr;
    uint8_t* d_staging = nullptr;       // frames uploaded by the host-pointer API
    const uint8_t* d_frames = nullptr;  // frames of the last segment call
    uint32_t* d_ctr = nullptr;          // [B*CTR_STRIDE] per-frame overflow counters (128 B apart) + [4] pooled totals
    uint32_t* d_band_fail = nullptr;    // [B] a band of the frame held more lines than k_ccl_band's workgroup takes (k_ccl_lds reads and clears it)
    uint32_t* d_row_cnt = nullptr;      // [B*H]
    uint32_t* d_row_off = nullptr;      // [B*H]   offset of the row's runs in tmp order
    uint32_t* d_row_base = nullptr;     // [B*(H+1)] exclusive scan of row_cnt = raster index
    uint32_t* d_tmp_runs = nullptr;     // [B*T]   x0 | x1 << 16; T = H*ROW_SLOT (row slots) + R (overflow area)
    trexhip_run* d_raster = nullptr;    // [B*R]   runs in raster order
    uint32_t* d_parent = nullptr;       // [B*R]   union-find parent -> root label
    uint32_t* d_root_ord = nullptr;     // [B*R]   ordinal of the raw blob rooted at run r
    uint32_t* d_cnt_runs = nullptr;     // [B*R]   per raw blob
    uint32_t* d_cnt_px = nullptr;       // [B*R]
    uint32_t* d_cur_run = nullptr;      // [B*R]   frame-relative run cursor per raw blob
    uint32_t* d_pix_begin = nullptr;    // [B*R]   frame-relative pixel begin per raw blob
    int32_t* d_blob_map = nullptr;      // [B*R]   raw blob ordinal -> kept index or -1
    trexhip_frame_info* d_info = nullptr;  // [B]
    trexhip_blob* d_blobs = nullptr;       // [B*NB] pooled
    uint32_t* d_blob_frame = nullptr;      // [B*NB]
    trexhip_run* d_runs = nullptr;         // [B*R] pooled, grouped by blob
    uint8_t* d_pixels = nullptr;           // [B*P] pooled

    // pinned host mirrors for trexhip_fetch
    trexhip_frame_info* h_info = nullptr;
    uint32_t* h_totals = nullptr;
    trexhip_blob* h_blobs = nullptr;
    trexhip_run* h_runs = nullptr;
    uint8_t* h_pixels = nullptr;
    uint8_t* h_staging = nullptr;       // pinned upload buffer
    double* d_warp = nullptr;           // per-blob inverse affine maps of the normalised crops
    int warp_cap = 0;
    float* d_len = nullptr;             // per-blob (median) midline lengths handed in by the caller
    int len_cap = 0;
    void* d_auto = nullptr;             // scratch of trexhip_posture_auto_device (thresholds, selections, first outlines)
    size_t auto_cap = 0;
    uint32_t* d_bits[2] = {nullptr, nullptr};   // 1 bit/pixel masks for the optional morphology [B][H][ceil(W/32)]
    uint8_t* d_color = nullptr;         // BGR/BGRA frames of the colour-input API
    uint8_t* h_color = nullptr;

    void* net = nullptr;                // trexhip::Net (cnn.hip)
    trexhip::Pass2 pass2;
    int cnn_mode = TREXHIP_CNN_FP16X3;  // TREXHIP_CNN_*: fp32-class arithmetic on the fp16 matrix cores, range-guarded

    bool profiling = false;
    // tuning knobs (env TREXHIP_ROWS_ORDER / TREXHIP_ROWS_BLOCKS override the defaults)
    int tune_rows_order = 0;
    int tune_rows_blocks = 8192;
    int tune_rows_k = 0;                // frames per wave of k_rows32b (TREXHIP_ROWS_K; 0 = default 8)
    bool tune_rows_blocks_set = false;  // TREXHIP_ROWS_BLOCKS given: no automatic grid for the wide pixel pass
    int tune_conv_geom = 0;             // dev only: alternative conv tilings (TREXHIP_CONV_GEOM)
    // hipFuncSetAttribute is per device: one process may drive several devices through several contexts
    bool attr_cnn = false, attr_ccl = false, attr_split = false;
    bool ctr_dirty = false;             // a detect pass was queued but not to its end: the per-frame overflow counters may be non-zero (launch_segment zeroes them first)
    int attr_posture_bytes = 0;
    int pix_ch = 1;                     // bytes per output pixel (pixel_encoding)
    const uint8_t* d_color_src = nullptr; // colour frames of the last segment_color* call ([n][H][W][color_ch])
    int color_ch = 0;
    uint8_t* d_bg_color = nullptr;      // colour background ([H][W][bg_color_ch]) for the difference crops of the rgb8 encoding
    int bg_color_ch = 0;
    int n_cus = 256;                    // compute units of the device (persistent kernels size their grids with it)
    int tune_seg_groups = 1;            // >1: pixel pass of frame group g+1 on the caller stream, labelling of g on an auxiliary stream (TREXHIP_SEG_GROUPS); measured SLOWER (cross-stream events cost 30-50 us each: 159 -> 266 us at 2 groups), kept off
    hipStream_t aux_stream = nullptr;   // labelling + gather of a group while the next group's pixel pass runs
    hipEvent_t ev_grp[10] = {};
    int tune_seg_scheme = 0;            // TREXHIP_SEG_SCHEME: how the groups use the two streams (launch_segment)
    int tune_ccl_stop = 0;              // dev only: stop k_ccl_lds after phase N (TREXHIP_CCL_STOP)
    int tune_ccl_bands = -1;            // dev only: several workgroups per frame (TREXHIP_CCL_BANDS; -1 = by the frames per launch and the hint word, 0 never, n always n bands)
    int tune_ccl_inst = 0;              // dev only: the k_ccl_lds instance that goes first (TREXHIP_CCL_INST; 0 = by the hint words)
    uint32_t* h_ccl_hint = nullptr;     // [2] pinned, written by k_ccl_lds: a frame had more lines than the S / the M instance holds
    uint32_t ccl_calls = 0;
    trexhip::Stage stages[TREXHIP_STAGE_COUNT];
    trexhip::Uploader up;
};

namespace trexhip {
int launch_segment(trexhip_ctx* ctx, const uint8_t* d_frames, int n);
void stage_begin(trexhip_ctx* ctx, int stage);
void stage_end(trexhip_ctx* ctx, int stage);
void net_free(trexhip_ctx* ctx);
int launch_pending(trexhip_ctx* ctx);
int launch_morphology(trexhip_ctx* ctx, const uint8_t* d_frames, int n, const uint32_t** result);
int launch_rethreshold(trexhip_ctx* ctx, int thr, int method, const double* ranges, int n_ranges, const int32_t* d_blob_thr);
int launch_to_gray(trexhip_ctx* ctx, const uint8_t* d_color, uint8_t* d_gray, size_t npix, int channels, int color_channel);
void upload_free(trexhip_ctx* ctx);
int upload_frames(trexhip_ctx* ctx, const uint8_t* const* frames, int n, size_t rows, size_t row_bytes, size_t stride, uint8_t* d_dst,
                  const std::function<int(int, int)>& after_chunk, int reduce_channels = 0, int color_channel = -1);
}
