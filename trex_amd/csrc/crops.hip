// crops.hip -- blob -> 80x80 identity-network input ("individual_image_size"), gathered on the device.
//
// Replaces constraints::diff_image with individual_image_normalization = none
//   Application/src/tracker/tracking/FilterCache.cpp:265-294 -> calculate_diff_image :157-235
// (imageFromLines paints the blob's pixels -- or, with track_background_subtraction, its background
// differences -- into its bounding box; the box is then centre-padded with zeros / centre-cropped to
// the output size: pad = out - size, right = pad/2, left = pad - right (:184-194); cut likewise (:212-226)).
// One workgroup = one blob of the last segmented batch, in pooled order.
#include "internal.h"

namespace trexhip {

__global__ __launch_bounds__(256) void k_crops_none(const SegCfg c, const uint8_t* __restrict__ frames,
                                                    const uint8_t* __restrict__ bg,
                                                    const trexhip_frame_info* __restrict__ info,
                                                    const uint32_t* __restrict__ blob_frame,
                                                    const trexhip_blob* __restrict__ blobs,
                                                    const trexhip_run* __restrict__ runs, uint8_t* __restrict__ crops,
                                                    int OW, int OH, int diff_mode /*0 raw, 1 |bg-p|, 2 max(bg-p,0)*/) {
    const uint32_t bi = blockIdx.x;
    uint8_t* out = crops + (size_t)bi * OW * OH;
    for (int i = threadIdx.x * 16; i < OW * OH; i += 256 * 16) *reinterpret_cast<uint4*>(out + i) = make_uint4(0, 0, 0, 0);
    const uint32_t f = blob_frame[bi];
    if (f >= (uint32_t)c.B) return;
    const trexhip_frame_info fi = info[f];
    if (fi.flags) return;
    const trexhip_blob B = blobs[bi];
    const int bw = B.x1 - B.x0 + 1, bh = B.y1 - B.y0 + 1;
    // FilterCache.cpp:184-194 / :212-226 -- the same split for padding and for cutting
    int sx, sy;
    if (bw <= OW) { const int T = OW - bw; sx = T - T / 2; } else { const int T = bw - OW; sx = -(T - T / 2); }
    if (bh <= OH) { const int T = OH - bh; sy = T - T / 2; } else { const int T = bh - OH; sy = -(T - T / 2); }
    __syncthreads();
    const trexhip_run* rr = runs + fi.run_begin + B.run_begin;
    const uint8_t* img = frames + (size_t)f * c.H * c.W;
    for (uint32_t i = threadIdx.x >> 4; i < B.n_runs; i += 16) {       // 16 lanes per run
        const trexhip_run q = rr[i];
        const int oy = (int)q.y - (int)B.y0 + sy;
        if (oy < 0 || oy >= OH) continue;
        for (int x = q.x0 + (threadIdx.x & 15); x <= q.x1; x += 16) {
            const int ox = x - (int)B.x0 + sx;
            if (ox < 0 || ox >= OW) continue;
            int p = img[(size_t)q.y * c.W + x];
            if (c.invert) p = 255 - p;
            if (diff_mode) {
                const int b = bg[(size_t)q.y * c.W + x];
                p = diff_mode == 1 ? abs(b - p) : max(b - p, 0);
            }
            out[oy * OW + ox] = (uint8_t)p;
        }
    }
}

int launch_crops(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode) {
    if (n <= 0) return TREXHIP_OK;
    SegCfg c = ctx->cfg;
    c.B = ctx->last_n;
    stage_begin(ctx, TREXHIP_STAGE_CROPS);
    hipLaunchKernelGGL(k_crops_none, dim3(n), dim3(256), 0, ctx->stream, c, ctx->d_frames, ctx->d_bg, ctx->d_info,
                       ctx->d_blob_frame, ctx->d_blobs, ctx->d_runs, d_crops, OW, OH, diff_mode);
    stage_end(ctx, TREXHIP_STAGE_CROPS);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

}  // namespace trexhip

using namespace trexhip;

extern "C" {

int trexhip_crops_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                         int32_t normalization, int32_t difference) {
    if (!ctx || !d_crops) { set_error("trexhip_crops_device: null argument"); return TREXHIP_E_INVALID; }
    if (normalization != TREXHIP_NORMALIZE_NONE) {
        set_error("trexhip_crops_device: only individual_image_normalization=none is implemented on the device");
        return TREXHIP_E_UNSUPPORTED;
    }
    if (out_w <= 0 || out_h <= 0 || (out_w * out_h) % 16 != 0) { set_error("trexhip_crops_device: output size must be a multiple of 16 bytes"); return TREXHIP_E_INVALID; }
    if (difference < 0 || difference > 2) { set_error("trexhip_crops_device: difference must be 0,1,2"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_crops_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0) { set_error("trexhip_crops_device: no segmented batch"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    return launch_crops(ctx, d_crops, n_blobs, out_w, out_h, difference);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// per-blob identity table: the fixed-size record that frame-sharded ranks all-gather over xGMI for
// the sequential matcher on rank 0 (consumer: Tracker::predicted, Tracker.cpp:237-247)
// row = 8 header words + C probabilities:
//   [0] global frame index  [1] pv::bid  [2] n_pixels  [3] x0 | y0<<16  [4] x1 | y1<<16
//   [5] centroid x (float)  [6] centroid y (float)     [7] 1 = valid row
// ------------------------------------------------------------------------------------------------
namespace trexhip {
__global__ __launch_bounds__(256) void k_id_table(const trexhip_frame_info* __restrict__ info,
                                                  const uint32_t* __restrict__ blob_frame,
                                                  const trexhip_blob* __restrict__ blobs, const float* __restrict__ probs,
                                                  int n, int C, int B, uint32_t frame_base, uint32_t* __restrict__ table,
                                                  int max_rows) {
    const int row = blockIdx.x;
    uint32_t* out = table + (size_t)row * (8 + C);
    bool valid = row < n;
    uint32_t f = 0;
    if (valid) { f = blob_frame[row]; valid = f < (uint32_t)B && info[f].flags == 0; }
    if (threadIdx.x == 0) {
        uint32_t hdr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (valid) {
            const trexhip_blob b = blobs[row];
            hdr[0] = frame_base + f; hdr[1] = b.bid; hdr[2] = b.n_pixels;
            hdr[3] = b.x0 | ((uint32_t)b.y0 << 16); hdr[4] = b.x1 | ((uint32_t)b.y1 << 16);
            hdr[5] = __float_as_uint((float)((double)b.m10 / (double)b.n_pixels));
            hdr[6] = __float_as_uint((float)((double)b.m01 / (double)b.n_pixels));
            hdr[7] = 1u;
        }
        for (int i = 0; i < 8; ++i) out[i] = hdr[i];
    }
    for (int c = threadIdx.x; c < C; c += 256)
        out[8 + c] = (valid && probs) ? __float_as_uint(probs[(size_t)row * C + c]) : 0u;
}
}  // namespace trexhip

extern "C" int trexhip_export_id_table_device(trexhip_ctx* ctx, const float* d_probs, int32_t n_blobs, int32_t classes,
                                              uint32_t frame_base, void* d_table, int32_t max_rows) {
    if (!ctx || !d_table) { trexhip::set_error("trexhip_export_id_table_device: null argument"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || max_rows < n_blobs || classes < 0) { trexhip::set_error("trexhip_export_id_table_device: need 0 <= n_blobs <= max_rows"); return TREXHIP_E_INVALID; }
    if (max_rows == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipLaunchKernelGGL(trexhip::k_id_table, dim3(max_rows), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs,
                       d_probs, n_blobs, classes, ctx->last_n, frame_base, static_cast<uint32_t*>(d_table), max_rows);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}
