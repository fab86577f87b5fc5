// crops.hip -- blob -> 80x80 identity-network input ("individual_image_size"), gathered on the device.
//
// Replaces constraints::diff_image with individual_image_normalization = none
//   Application/src/tracker/tracking/FilterCache.cpp:265-294 -> calculate_diff_image :157-235
// (imageFromLines paints the blob's pixels -- or, with track_background_subtraction, its background
// differences -- into its bounding box; the box is then centre-padded with zeros / centre-cropped to
// the output size: pad = out - size, right = pad/2, left = pad - right (:184-194); cut likewise (:212-226)).
// One workgroup = one blob of the last segmented batch, in pooled order.
#include "internal.h"
#include "affine.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace trexhip {

__global__ __launch_bounds__(256) void k_crops_none(const SegCfg c, const uint8_t* __restrict__ frames,
                                                    const uint8_t* __restrict__ bg,
                                                    const trexhip_frame_info* __restrict__ info,
                                                    const uint32_t* __restrict__ blob_frame,
                                                    const trexhip_blob* __restrict__ blobs,
                                                    const trexhip_run* __restrict__ runs, uint8_t* __restrict__ crops,
                                                    int OW, int OH, int diff_mode /*0 raw, 1 |bg-p|, 2 max(bg-p,0)*/,
                                                    const uint8_t* __restrict__ color, int color_ch, int och /*1 grey or r3g3b2 code, 3 rgb8*/, int enc,
                                                    const uint8_t* __restrict__ bgc, int bgc_ch) {
    const uint32_t bi = blockIdx.x;
    uint8_t* out = crops + (size_t)bi * OW * OH * och;
    for (int i = threadIdx.x * 16; i < OW * OH * och; i += 256 * 16) *reinterpret_cast<uint4*>(out + i) = make_uint4(0, 0, 0, 0);
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
            if (och == 3) {                                   // rgb8: the colour pixel itself (imageFromLines, Tests/test_pixels.cpp:1381-1460)
                const uint8_t* s = color + (((size_t)f * c.H + q.y) * c.W + x) * color_ch;
                uint8_t* d = out + ((size_t)oy * OW + ox) * 3;
                if (diff_mode) {                              // per-channel background difference (imageFromLines' `differences`, Tests/test_pixels.cpp:1462-1479)
                    const uint8_t* b3 = bgc + ((size_t)q.y * c.W + x) * bgc_ch;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) d[ch] = (uint8_t)(diff_mode == 1 ? abs((int)b3[ch] - (int)s[ch]) : max((int)b3[ch] - (int)s[ch], 0));
                } else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
                continue;
            }
            if (enc == TREXHIP_ENC_R3G3B2) {                  // the colour code of the pixel (convert_to_r3g3b2)
                const uint8_t* s = color + (((size_t)f * c.H + q.y) * c.W + x) * color_ch;
                out[oy * OW + ox] = (uint8_t)(((s[0] >> 6) << 6) | ((s[1] >> 5) << 3) | (s[2] >> 5));
                continue;
            }
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

int launch_crops_warp(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode, const float* tr6, const float* lengths,
                      bool legacy, float scale);
int launch_crops_warp_device(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode, const trexhip_midline_info* d_minfo,
                             const float* d_lengths, bool legacy, float scale);
int check_colour_difference(trexhip_ctx* ctx, int difference, const char* who);

int launch_crops(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode) {
    if (n <= 0) return TREXHIP_OK;
    SegCfg c = ctx->cfg;
    c.invert = ctx->batch_invert; c.zero_bg = ctx->batch_zero_bg;
    c.B = ctx->last_n;
    stage_begin(ctx, TREXHIP_STAGE_CROPS);
    hipLaunchKernelGGL(k_crops_none, dim3(n), dim3(256), 0, ctx->stream, c, ctx->d_frames, ctx->d_bg, ctx->d_info,
                       ctx->d_blob_frame, ctx->d_blobs, ctx->d_runs, d_crops, OW, OH, diff_mode, ctx->d_color_src, ctx->color_ch,
                       ctx->p.pixel_encoding == TREXHIP_ENC_RGB8 ? 3 : 1, ctx->p.pixel_encoding, ctx->d_bg_color, ctx->bg_color_ch);
    stage_end(ctx, TREXHIP_STAGE_CROPS);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// background-difference crops of a colour encoding: rgb8 with a colour background (trexhip_set_background_color)
int check_colour_difference(trexhip_ctx* ctx, int difference, const char* who) {
    if (ctx->p.pixel_encoding == TREXHIP_ENC_GRAY || difference == 0) return TREXHIP_OK;
    if (ctx->p.pixel_encoding != TREXHIP_ENC_RGB8) {
        set_error(std::string(who) + ": r3g3b2 crops hold the colour codes: background-difference crops are not implemented for them"); return TREXHIP_E_UNSUPPORTED;
    }
    if (!ctx->d_bg_color) { set_error(std::string(who) + ": background-difference crops of rgb8 pixels need the colour background (trexhip_set_background_color)"); return TREXHIP_E_INVALID; }
    return TREXHIP_OK;
}

}  // namespace trexhip

using namespace trexhip;

extern "C" {

int trexhip_crops_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                         int32_t normalization, int32_t difference) {
    if (!ctx || !d_crops) { set_error("trexhip_crops_device: null argument"); return TREXHIP_E_INVALID; }
    if (int rc = check_colour_difference(ctx, difference, "trexhip_crops_device")) return rc;
    if (ctx->p.pixel_encoding == TREXHIP_ENC_RGB8 && (out_w * out_h * 3) % 16 != 0) { set_error("trexhip_crops_device: out_w*out_h*3 must be a multiple of 16"); return TREXHIP_E_UNSUPPORTED; }
    if (normalization != TREXHIP_NORMALIZE_NONE && normalization != TREXHIP_NORMALIZE_MOMENTS) {
        set_error("trexhip_crops_device: posture / legacy normalisation need the caller's Midline::transform: use trexhip_crops_transformed_device");
        return TREXHIP_E_UNSUPPORTED;
    }
    if (out_w <= 0 || out_h <= 0 || (out_w * out_h) % 16 != 0) { set_error("trexhip_crops_device: output size must be a multiple of 16 bytes"); return TREXHIP_E_INVALID; }
    if (difference < 0 || difference > 2) { set_error("trexhip_crops_device: difference must be 0,1,2"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_crops_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0) { set_error("trexhip_crops_device: no segmented batch"); return TREXHIP_E_INVALID; }
    if (!ctx->fetched) { set_error("trexhip_crops_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (normalization == TREXHIP_NORMALIZE_MOMENTS)      // transform from the blob table's integer moments, built on the device (midline.hip)
        return launch_crops_warp_device(ctx, d_crops, n_blobs, out_w, out_h, difference, nullptr, nullptr, false, 1.0f);
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
    if (!ctx->fetched) { trexhip::set_error("trexhip_export_id_table_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipLaunchKernelGGL(trexhip::k_id_table, dim3(max_rows), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs,
                       d_probs, n_blobs, classes, ctx->last_n, frame_base, static_cast<uint32_t*>(d_table), max_rows);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// the full per-blob record of SURVEY.md 8(e): header + second moments + midline pose, probabilities, normalised midline points
namespace trexhip {
__global__ __launch_bounds__(256) void k_id_table_ex(const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ blob_frame,
                                                     const trexhip_blob* __restrict__ blobs, const float* __restrict__ probs,
                                                     const float4* __restrict__ midline, const trexhip_midline_info* __restrict__ minfo,
                                                     int n, int C, int R, int B, uint32_t frame_base, uint32_t* __restrict__ table) {
    const int row = blockIdx.x;
    const int rowlen = 16 + C + 3 * R;
    uint32_t* out = table + (size_t)row * rowlen;
    bool valid = row < n;
    uint32_t f = 0;
    if (valid) { f = blob_frame[row]; valid = f < (uint32_t)B && info[f].flags == 0; }
    int mstatus = -1;
    if (threadIdx.x == 0) {
        uint32_t hdr[16];
        for (int i = 0; i < 16; ++i) hdr[i] = 0;
        if (valid) {
            const trexhip_blob b = blobs[row];
            const double np_ = (double)b.n_pixels, cx = (double)b.m10 / np_, cy = (double)b.m01 / np_;
            hdr[0] = frame_base + f; hdr[1] = b.bid; hdr[2] = b.n_pixels;
            hdr[3] = b.x0 | ((uint32_t)b.y0 << 16); hdr[4] = b.x1 | ((uint32_t)b.y1 << 16);
            hdr[5] = __float_as_uint((float)cx); hdr[6] = __float_as_uint((float)cy); hdr[7] = 1u;
            hdr[8] = __float_as_uint((float)((double)b.m20 / np_ - cx * cx));          // central second moments per pixel
            hdr[9] = __float_as_uint((float)((double)b.m11 / np_ - cx * cy));
            hdr[10] = __float_as_uint((float)((double)b.m02 / np_ - cy * cy));
            hdr[15] = 0xffffffffu;                                                      // midline status: -1 = no posture handed in
            if (minfo) {
                const trexhip_midline_info m = minfo[row];
                hdr[11] = __float_as_uint(m.len); hdr[12] = __float_as_uint(m.angle);
                hdr[13] = __float_as_uint(m.offx); hdr[14] = __float_as_uint(m.offy); hdr[15] = (uint32_t)m.status;
            }
        }
        for (int i = 0; i < 16; ++i) out[i] = hdr[i];
    }
    if (valid && minfo) mstatus = minfo[row].status;
    for (int c = threadIdx.x; c < C; c += 256)
        out[16 + c] = (valid && probs) ? __float_as_uint(probs[(size_t)row * C + c]) : 0u;
    for (int i = threadIdx.x; i < R; i += 256) {
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && midline && mstatus == 0) p = midline[(size_t)row * R + i];          // MidlineSegment{pos.x, pos.y, height, l_length}
        out[16 + C + 3 * i] = __float_as_uint(p.x); out[16 + C + 3 * i + 1] = __float_as_uint(p.y); out[16 + C + 3 * i + 2] = __float_as_uint(p.z);
    }
}
}  // namespace trexhip

extern "C" int trexhip_export_id_table_ex_device(trexhip_ctx* ctx, const float* d_probs, int32_t n_blobs, int32_t classes, uint32_t frame_base,
                                                 const float* d_midline, const trexhip_midline_info* d_midline_info, int32_t midline_resolution,
                                                 void* d_table, int32_t max_rows) {
    if (!ctx || !d_table) { trexhip::set_error("trexhip_export_id_table_ex_device: null argument"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || max_rows < n_blobs || classes < 0 || midline_resolution < 0 || midline_resolution > 256) { trexhip::set_error("trexhip_export_id_table_ex_device: need 0 <= n_blobs <= max_rows, 0 <= midline_resolution <= 256"); return TREXHIP_E_INVALID; }
    if ((d_midline == nullptr) != (d_midline_info == nullptr)) { trexhip::set_error("trexhip_export_id_table_ex_device: midline points and infos come together"); return TREXHIP_E_INVALID; }
    if (max_rows == 0) return TREXHIP_OK;
    if (!ctx->fetched) { trexhip::set_error("trexhip_export_id_table_ex_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipLaunchKernelGGL(trexhip::k_id_table_ex, dim3(max_rows), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs, d_probs,
                       reinterpret_cast<const float4*>(d_midline), d_midline_info, n_blobs, classes, midline_resolution, ctx->last_n, frame_base,
                       static_cast<uint32_t*>(d_table));
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// colour reduce in front of the detect stage (BackgroundSubtraction.cpp:162-180): cv::cvtColor
// BGR2GRAY / BGRA2GRAY (8-bit fixed point: (B*1868 + G*9617 + R*4899 + 8192) >> 14) or a channel pick
// (color_channel).  4 pixels per thread, 12/16-byte loads, 4-byte stores.
// ------------------------------------------------------------------------------------------------
namespace trexhip {
template <int CH>
__global__ __launch_bounds__(256) void k_to_gray(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t npix4,
                                                 int color_channel) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix4) return;
    uint32_t w[CH];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src) + i * CH;
#pragma unroll
    for (int k = 0; k < CH; ++k) w[k] = s[k];
    uint32_t out = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t c[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int byte = p * CH + k;
            c[k] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xffu;
        }
        uint32_t g;
        if (color_channel >= 0) g = c[color_channel & 3];
        else g = (__umul24(c[0], 1868u) + __umul24(c[1], 9617u) + __umul24(c[2], 4899u) + 8192u) >> 14;   // 24-bit multiplies: full rate (v_mul_lo_u32 is quarter rate)
        out |= g << (8 * p);
    }
    reinterpret_cast<uint32_t*>(dst)[i] = out;
}

// the same, 16 pixels per thread: CH 16-byte loads, one 16-byte store (4 pixels per thread -- 16 bytes in, 4 out -- moved the 5.4 GB of a
// 256-frame BGRA batch at 3.9 TB/s)
template <int CH>
__global__ __launch_bounds__(256) void k_to_gray16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t npix16, int color_channel) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix16) return;
    uint32_t w[4 * CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) { const uint4 v = src[i * CH + k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        uint32_t c[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int byte = p * CH + k;
            c[k] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xffu;
        }
        uint32_t g;
        if (color_channel >= 0) g = c[color_channel & 3];
        else g = (__umul24(c[0], 1868u) + __umul24(c[1], 9617u) + __umul24(c[2], 4899u) + 8192u) >> 14;
        out[p >> 2] |= g << (8 * (p & 3));
    }
    dst[i] = make_uint4(out[0], out[1], out[2], out[3]);
}

int launch_to_gray(trexhip_ctx* ctx, const uint8_t* d_color, uint8_t* d_gray, size_t npix, int channels, int color_channel) {
    if (npix % 16 == 0 && ((reinterpret_cast<uintptr_t>(d_color) | reinterpret_cast<uintptr_t>(d_gray)) & 15) == 0) {
        const size_t n16 = npix / 16;
        const dim3 grid16((unsigned)((n16 + 255) / 256));
        if (channels == 3) hipLaunchKernelGGL((k_to_gray16<3>), grid16, dim3(256), 0, ctx->stream, reinterpret_cast<const uint4*>(d_color), reinterpret_cast<uint4*>(d_gray), n16, color_channel);
        else               hipLaunchKernelGGL((k_to_gray16<4>), grid16, dim3(256), 0, ctx->stream, reinterpret_cast<const uint4*>(d_color), reinterpret_cast<uint4*>(d_gray), n16, color_channel);
        TH_CHECK_HIP(hipGetLastError());
        return TREXHIP_OK;
    }
    const size_t n4 = npix / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (channels == 3) hipLaunchKernelGGL((k_to_gray<3>), grid, dim3(256), 0, ctx->stream, d_color, d_gray, n4, color_channel);
    else               hipLaunchKernelGGL((k_to_gray<4>), grid, dim3(256), 0, ctx->stream, d_color, d_gray, n4, color_channel);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}
}  // namespace trexhip

extern "C" int trexhip_segment_color(trexhip_ctx* ctx, const uint8_t* const* frames, int32_t stride, int32_t n,
                                     int32_t channels, int32_t color_channel) {
    using namespace trexhip;
    if (!ctx || !frames) { set_error("trexhip_segment_color: null argument"); return TREXHIP_E_INVALID; }
    if (!ctx->has_bg) { set_error("trexhip_segment_color: background image not set"); return TREXHIP_E_INVALID; }
    if (n < 0 || n > ctx->p.max_batch) { set_error("trexhip_segment_color: n outside 0..max_batch"); return TREXHIP_E_INVALID; }
    if (channels != 3 && channels != 4) {   // BackgroundSubtraction.cpp:179 throws for anything else
        set_error("Invalid number of channels in input image for the network."); return TREXHIP_E_INVALID;
    }
    if (color_channel >= channels) color_channel = -1;                 // BackgroundSubtraction.cpp:163-164
    const size_t W = ctx->p.width, H = ctx->p.height;
    if ((W * H) % 4 != 0) { set_error("trexhip_segment_color: width*height must be a multiple of 4"); return TREXHIP_E_UNSUPPORTED; }
    if ((size_t)stride < W * channels) { set_error("trexhip_segment_color: stride < width*channels"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (n == 0) { ctx->last_n = 0; ctx->fetched = false; return TREXHIP_OK; }
    if (!ctx->d_staging) {
        TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_staging), (size_t)ctx->p.max_batch * W * H + 16));
    }
    for (int i = 0; i < n; ++i)
        if (!frames[i]) { set_error("trexhip_segment_color: null frame pointer"); return TREXHIP_E_INVALID; }
    const size_t row = W * channels, fpx = W * H;
    if (ctx->p.pixel_encoding == TREXHIP_ENC_GRAY && !ctx->p.device_color_reduce) {
        // gray / binary pixel arrays never look at the colour tile again: the upload threads reduce it while they copy (hostcvt.cpp,
        // the same fixed-point formula as k_to_gray), so a quarter / a third of the bytes cross PCIe
        int rc = upload_frames(ctx, frames, n, H, row, (size_t)stride, ctx->d_staging, nullptr, channels, color_channel);
        if (rc) return rc;
        ctx->d_color_src = nullptr; ctx->color_ch = 0;
        return launch_segment(ctx, ctx->d_staging, n);
    }
    if (!ctx->d_color) TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_color), (size_t)ctx->p.max_batch * W * H * 4 + 16));
    // frame by frame: pageable tile -> pinned ring slot (host threads) -> HBM (DMA on the copy stream) -> cv::cvtColor on the compute
    // stream as soon as the frame has landed; the next frame's staging and DMA run meanwhile (upload.hip)
    int rc = upload_frames(ctx, frames, n, H, row, (size_t)stride, ctx->d_color, [&](int i0, int cnt) {
        return launch_to_gray(ctx, ctx->d_color + (size_t)i0 * H * row, ctx->d_staging + (size_t)i0 * fpx, fpx * (size_t)cnt, channels, color_channel);
    });
    if (rc) return rc;
    ctx->d_color_src = ctx->d_color; ctx->color_ch = channels;       // the colour pixel encodings gather from here
    return launch_segment(ctx, ctx->d_staging, n);
}

extern "C" int trexhip_segment_color_device(trexhip_ctx* ctx, const uint8_t* d_color_frames, int32_t n, int32_t channels, int32_t color_channel) {
    using namespace trexhip;
    if (!ctx || !d_color_frames) { set_error("trexhip_segment_color_device: null argument"); return TREXHIP_E_INVALID; }
    if (!ctx->has_bg) { set_error("trexhip_segment_color_device: background image not set"); return TREXHIP_E_INVALID; }
    if (n < 0 || n > ctx->p.max_batch) { set_error("trexhip_segment_color_device: n outside 0..max_batch"); return TREXHIP_E_INVALID; }
    if (channels != 3 && channels != 4) { set_error("Invalid number of channels in input image for the network."); return TREXHIP_E_INVALID; }
    if (color_channel >= channels) color_channel = -1;
    const size_t W = ctx->p.width, H = ctx->p.height;
    if ((W * H) % 4 != 0) { set_error("trexhip_segment_color_device: width*height must be a multiple of 4"); return TREXHIP_E_UNSUPPORTED; }
    if (reinterpret_cast<uintptr_t>(d_color_frames) & 3) { set_error("trexhip_segment_color_device: colour frames must be 4-byte aligned"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (n == 0) { ctx->last_n = 0; ctx->fetched = false; return TREXHIP_OK; }
    if (!ctx->d_staging) {
        TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_staging), (size_t)ctx->p.max_batch * W * H + 16));
    }
    int rc = launch_to_gray(ctx, d_color_frames, ctx->d_staging, (size_t)n * W * H, channels, color_channel);
    if (rc) return rc;
    ctx->d_color_src = d_color_frames; ctx->color_ch = channels;
    return launch_segment(ctx, ctx->d_staging, n);
}

// Background(image, meta_encoding_t::rgb8): the colour background stays resident for the per-channel difference crops, the detection
// and the track threshold keep working on its cv::cvtColor(BGR2GRAY) (Tests/test_pixels.cpp:1385-1400 builds both from one image and
// expects the same masks)
static int set_background_color_common(trexhip_ctx* ctx, int32_t channels, int32_t color_channel) {
    using namespace trexhip;
    const size_t W = ctx->p.width, H = ctx->p.height;
    int rc = launch_to_gray(ctx, ctx->d_bg_color, ctx->d_bg, W * H, channels, color_channel >= channels ? -1 : color_channel);
    if (rc) return rc;
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->bg_color_ch = channels;
    ctx->has_bg = true;
    return TREXHIP_OK;
}
static int alloc_background_color(trexhip_ctx* ctx, int32_t channels) {
    using namespace trexhip;
    const size_t W = ctx->p.width, H = ctx->p.height;
    if (channels != 3 && channels != 4) { set_error("trexhip_set_background_color: channels must be 3 (BGR) or 4 (BGRA)"); return TREXHIP_E_INVALID; }
    if ((W * H) % 4 != 0) { set_error("trexhip_set_background_color: width*height must be a multiple of 4"); return TREXHIP_E_UNSUPPORTED; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (ctx->d_bg_color && ctx->bg_color_ch != channels) { (void)hipFree(ctx->d_bg_color); ctx->d_bg_color = nullptr; ctx->bg_color_ch = 0; }
    if (!ctx->d_bg_color) TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_bg_color), W * H * (size_t)channels));
    return TREXHIP_OK;
}
extern "C" int trexhip_set_background_color(trexhip_ctx* ctx, const uint8_t* bgr, int32_t stride, int32_t channels, int32_t color_channel) {
    using namespace trexhip;
    if (!ctx || !bgr) { set_error("trexhip_set_background_color: null argument"); return TREXHIP_E_INVALID; }
    if (stride < ctx->p.width * channels) { set_error("trexhip_set_background_color: stride < width * channels"); return TREXHIP_E_INVALID; }
    if (int rc = alloc_background_color(ctx, channels)) return rc;
    TH_CHECK_HIP(hipMemcpy2DAsync(ctx->d_bg_color, (size_t)ctx->p.width * channels, bgr, stride, (size_t)ctx->p.width * channels, ctx->p.height,
                                  hipMemcpyHostToDevice, ctx->stream));
    return set_background_color_common(ctx, channels, color_channel);
}
extern "C" int trexhip_set_background_color_device(trexhip_ctx* ctx, const uint8_t* d_bgr, int32_t channels, int32_t color_channel) {
    using namespace trexhip;
    if (!ctx || !d_bgr) { set_error("trexhip_set_background_color_device: null argument"); return TREXHIP_E_INVALID; }
    if (int rc = alloc_background_color(ctx, channels)) return rc;
    TH_CHECK_HIP(hipMemcpyAsync(ctx->d_bg_color, d_bgr, (size_t)ctx->p.width * ctx->p.height * channels, hipMemcpyDeviceToDevice, ctx->stream));
    return set_background_color_common(ctx, channels, color_channel);
}

// ------------------------------------------------------------------------------------------------
// background model from sampled frames (Segmenter::trigger_average_generator, ui/Segmenter.cpp:467-566 ->
// VideoSource::generate_average [commons]; settings averaging_method / average_samples,
// grabber/misc/default_config.cpp:131-132): per pixel mean (float accumulation in sample order, rounded
// half-to-even like cv::Mat::convertTo), max or min over n gray frames.  16 pixels per thread.
// ------------------------------------------------------------------------------------------------
namespace trexhip {
__global__ __launch_bounds__(256) void k_average(const uint8_t* __restrict__ frames, uint8_t* __restrict__ bg, size_t npix16,
                                                 size_t frame_stride, int n, int method) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix16) return;
    float acc[16];
    uint32_t ext[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) ext[w] = method == 2 ? 0xffffffffu : 0u;
    for (int f = 0; f < n; ++f) {
        const uint4 v = *reinterpret_cast<const uint4*>(frames + (size_t)f * frame_stride + i * 16);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t p = (w4[w] >> (8 * b)) & 0xffu;
                if (method == 0) acc[4 * w + b] += (float)p;
                else {
                    const uint32_t c = (ext[w] >> (8 * b)) & 0xffu;
                    const uint32_t r = method == 1 ? max(c, p) : min(c, p);
                    ext[w] = (ext[w] & ~(0xffu << (8 * b))) | (r << (8 * b));
                }
            }
    }
    if (method == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float m = acc[4 * w + b] / (float)n;
                o |= (uint32_t)min(max((int)rintf(m), 0), 255) << (8 * b);
            }
            ext[w] = o;
        }
    }
    *reinterpret_cast<uint4*>(bg + i * 16) = make_uint4(ext[0], ext[1], ext[2], ext[3]);
}
// averaging_method = mode: the most frequent grey value of every pixel over the samples (the smallest one wins a tie).  One thread per
// pixel, its 256 one-byte bins in LDS (64 KB per workgroup of 256 pixels; the caller feeds at most 255 samples per bin overflow-free --
// n <= 255 is checked), bins laid out [value][thread] so that a wave's increments fall into 64 different banks.
__global__ __launch_bounds__(256) void k_average_mode(const uint8_t* __restrict__ frames, uint8_t* __restrict__ bg, size_t npix, size_t frame_stride, int n) {
    extern __shared__ uint8_t hist[];                                   // [256 values][256 threads]
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * 256 + tid;
    for (int k = tid; k < 256 * 256 / 4; k += 256) reinterpret_cast<uint32_t*>(hist)[k] = 0u;
    __syncthreads();
    if (i < npix)
        for (int f = 0; f < n; ++f) {
            const uint32_t p = frames[(size_t)f * frame_stride + i];
            hist[p * 256u + tid] += 1;
        }
    if (i >= npix) return;
    int best = 0, cnt = hist[tid];
    for (int v = 1; v < 256; ++v) { const int c = hist[v * 256 + tid]; if (c > cnt) { cnt = c; best = v; } }
    bg[i] = (uint8_t)best;
}
}  // namespace trexhip

extern "C" int trexhip_generate_average_device(trexhip_ctx* ctx, const uint8_t* d_frames, int32_t n, int32_t method) {
    using namespace trexhip;
    if (!ctx || !d_frames) { set_error("trexhip_generate_average_device: null argument"); return TREXHIP_E_INVALID; }
    if (n < 1) { set_error("trexhip_generate_average_device: need at least one sample"); return TREXHIP_E_INVALID; }
    if (method < 0 || method > 3) { set_error("trexhip_generate_average_device: method must be 0 mean, 1 max, 2 min or 3 mode"); return TREXHIP_E_INVALID; }
    if (method == 3 && n > 255) { set_error("trexhip_generate_average_device: averaging_method mode takes at most 255 samples (one-byte bins)"); return TREXHIP_E_UNSUPPORTED; }
    const size_t npix = (size_t)ctx->p.width * ctx->p.height;
    if (npix % 16 != 0 || (reinterpret_cast<uintptr_t>(d_frames) & 15)) { set_error("trexhip_generate_average_device: width*height must be a multiple of 16 and the frames 16-byte aligned"); return TREXHIP_E_UNSUPPORTED; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    const size_t n16 = npix / 16;
    if (method == 3) {
        static bool attr = false;
        if (!attr) { TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_average_mode), hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); attr = true; }
        hipLaunchKernelGGL(k_average_mode, dim3((unsigned)((npix + 255) / 256)), dim3(256), 65536, ctx->stream, d_frames, ctx->d_bg, npix, npix, n);
    } else
    hipLaunchKernelGGL(k_average, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, ctx->stream, d_frames, ctx->d_bg, n16, npix, n, method);
    TH_CHECK_HIP(hipGetLastError());
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->has_bg = true;
    return TREXHIP_OK;
}

extern "C" int trexhip_get_background(trexhip_ctx* ctx, uint8_t* gray, int32_t stride) {
    using namespace trexhip;
    if (!ctx || !gray || !ctx->has_bg) { set_error("trexhip_get_background: no background"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipMemcpy2D(gray, stride, ctx->d_bg, ctx->p.width, ctx->p.width, ctx->p.height, hipMemcpyDeviceToHost));
    return TREXHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// normalised crops: individual_image_normalization = moments / posture / legacy
// (constraints::diff_image -> calculate_normalized_diff_image -> normalize_image, tracking/FilterCache.cpp:21-115,265-294)
//   image of the blob in its bounding box (imageFromLines) --cv::warpAffine(INTER_LINEAR, BORDER_CONSTANT)--> out_w x out_h
// The per-blob 2x3 transform is composed on the host exactly like normalize_image does (translate(size/2), scale,
// translate(len*0.4 | -len/2), combine(tr)), inverted in double like cv::warpAffine, and the device does OpenCV's 8-bit
// fixed-point bilinear (AB_BITS 10, INTER_BITS 5, coefficient bits 15): integer arithmetic only, so the result does not
// depend on device float rounding.  tr = rotate(-orientation + 45 deg) . translate(-size/2) for `moments`
// (FilterCache.cpp:276-288; orientation from the blob's central moments), or Midline::transform(...) supplied by the
// caller for `posture` / `legacy` (Outline.cpp:1237-1255).
// ------------------------------------------------------------------------------------------------
namespace trexhip {

static constexpr int W_NR = 2048;      // lines of one blob held in LDS
static constexpr int W_OUT = 256;      // output rows / columns with tabulated fixed-point terms
static constexpr int W_IMG = 16384;    // bounding boxes up to this many pixels are painted into LDS

__global__ __launch_bounds__(256) void k_crops_warp(const SegCfg c, const uint8_t* __restrict__ frames, const uint8_t* __restrict__ bg,
                                                    const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ blob_frame,
                                                    const trexhip_blob* __restrict__ blobs, const trexhip_run* __restrict__ runs,
                                                    const double* __restrict__ minv /*[n][6] inverse maps*/, uint8_t* __restrict__ crops,
                                                    int OW, int OH, int diff_mode, const uint8_t* __restrict__ color, int color_ch,
                                                    int och /*1 grey or r3g3b2 code, 3 rgb8 (channels warped independently)*/, int enc,
                                                    const uint8_t* __restrict__ bgc, int bgc_ch) {
    __shared__ uint32_t s_runs[W_NR];
    __shared__ int s_row[1024 + 2];
    const uint32_t bi = blockIdx.x;
    uint8_t* out = crops + (size_t)bi * OW * OH * och;
    const uint32_t f = blob_frame[bi];
    bool ok = f < (uint32_t)c.B;
    trexhip_frame_info fi = {};
    if (ok) { fi = info[f]; ok = fi.flags == 0; }
    trexhip_blob B = {};
    if (ok) { B = blobs[bi]; ok = B.n_runs <= (uint32_t)W_NR && (B.y1 - B.y0 + 1) <= 1024; }
    if (!ok) { for (int i = threadIdx.x; i < OW * OH * och; i += 256) out[i] = 0; return; }
    const trexhip_run* rr = runs + fi.run_begin + B.run_begin;
    const int y0 = B.y0, rows = B.y1 - B.y0 + 1;
    for (int i = threadIdx.x; i < (int)B.n_runs; i += 256) {
        const trexhip_run q = rr[i];
        s_runs[i] = (uint32_t)q.x0 | ((uint32_t)q.x1 << 16);
        if (i == 0 || rr[i - 1].y != q.y) s_row[q.y - y0] = i;
    }
    if (threadIdx.x == 0) s_row[rows] = (int)B.n_runs;
    __syncthreads();
    const double* M = minv + (size_t)bi * 6;
    const double m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], m4 = M[4], m5 = M[5];
    const int sw = B.x1 - B.x0 + 1, sh = rows, plane = sw * sh;
    const uint8_t* img = frames + (size_t)f * c.H * c.W;
    const uint8_t* cimg = enc != TREXHIP_ENC_GRAY ? color + (size_t)f * c.H * c.W * color_ch : nullptr;
    const bool nearest = enc == TREXHIP_ENC_R3G3B2;          // colour codes are not interpolated (FilterCache.cpp:70-73: INTER_NEAREST)
    // value of the blob image at a member pixel: grey (raw or background difference) or one colour channel
    auto source = [&](int ay, int ax, int ch) -> int {
        if (och == 3) {
            const int p3 = cimg[((size_t)ay * c.W + ax) * color_ch + ch];
            if (!diff_mode) return p3;
            const int b3 = bgc[((size_t)ay * c.W + ax) * bgc_ch + ch];                  // per-channel background difference
            return diff_mode == 1 ? abs(b3 - p3) : max(b3 - p3, 0);
        }
        if (nearest) { const uint8_t* s = cimg + ((size_t)ay * c.W + ax) * color_ch; return ((s[0] >> 6) << 6) | ((s[1] >> 5) << 3) | (s[2] >> 5); }
        int p = img[(size_t)ay * c.W + ax];
        if (c.invert) p = 255 - p;
        if (diff_mode) { const int bgv = bg[(size_t)ay * c.W + ax]; p = diff_mode == 1 ? abs(bgv - p) : max(bgv - p, 0); }
        return p;
    };
    // imageFromLines: the blob's pixels on black over its bounding box.  Small boxes are painted into LDS once (then the four
    // taps of every output pixel are plain LDS reads); larger ones test line membership per tap.
    __shared__ uint8_t s_img[W_IMG];
    const bool staged = plane * och <= W_IMG;
    if (staged) {
        for (int i = threadIdx.x; i < (plane * och + 3) / 4; i += 256) reinterpret_cast<uint32_t*>(s_img)[i] = 0u;
        __syncthreads();
        for (int r = threadIdx.x >> 4; r < (int)B.n_runs; r += 16) {      // 16 lanes per line: its pixel loads are independent
            const uint32_t q = s_runs[r];
            const int xa = (int)(q & 0xffffu), xb = (int)(q >> 16), yy = (int)rr[r].y;
            for (int ax = xa + (int)(threadIdx.x & 15); ax <= xb; ax += 16)
                for (int ch = 0; ch < och; ++ch) s_img[ch * plane + (yy - y0) * sw + (ax - B.x0)] = (uint8_t)source(yy, ax, ch);
        }
        __syncthreads();
    }
    // cv::warpAffine's fixed-point coordinates: a row term and a column term per axis (AB_BITS 10), once per row / column
    __shared__ int s_rx[W_OUT], s_ry[W_OUT], s_cx[W_OUT], s_cy[W_OUT];
    const bool tabled = OW <= W_OUT && OH <= W_OUT;
    if (tabled) {
        const int rd = nearest ? 512 : 16;                      // round_delta: AB_SCALE/2 (nearest) or AB_SCALE/INTER_TAB_SIZE/2
        for (int i = threadIdx.x; i < OH; i += 256) { s_rx[i] = __double2int_rn((m1 * i + m2) * 1024.0) + rd; s_ry[i] = __double2int_rn((m4 * i + m5) * 1024.0) + rd; }
        for (int i = threadIdx.x; i < OW; i += 256) { s_cx[i] = __double2int_rn(m0 * i * 1024.0); s_cy[i] = __double2int_rn(m3 * i * 1024.0); }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < OW * OH; i += 256) {
        const int y = i / OW, x = i - y * OW;
        int X, Y;
        if (tabled) { X = s_rx[y] + s_cx[x]; Y = s_ry[y] + s_cy[x]; }
        else {
            const int rd = nearest ? 512 : 16;
            X = __double2int_rn((m1 * y + m2) * 1024.0) + rd + __double2int_rn(m0 * x * 1024.0);
            Y = __double2int_rn((m4 * y + m5) * 1024.0) + rd + __double2int_rn(m3 * x * 1024.0);
        }
        if (nearest) {                                           // X >> AB_BITS is the source pixel
            const int nx = X >> 10, ny = Y >> 10;
            int p = 0;
            if (nx >= 0 && nx < sw && ny >= 0 && ny < sh) {
                if (staged) p = s_img[ny * sw + nx];
                else {
                    const int ax = nx + B.x0;
                    for (int r = s_row[ny]; r < s_row[ny + 1]; ++r) { const uint32_t q = s_runs[r]; if (ax >= (int)(q & 0xffffu) && ax <= (int)(q >> 16)) { p = source(ny + y0, ax, 0); break; } }
                }
            }
            out[i] = (uint8_t)p;
            continue;
        }
        X >>= 5; Y >>= 5;
        const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
        if (sx < -1 || sx >= sw || sy < -1 || sy >= sh) { for (int ch = 0; ch < och; ++ch) out[i * och + ch] = 0; continue; }   // all four taps outside the box
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = sx + (k & 1), yy = sy + (k >> 1);
            in[k] = xx >= 0 && xx < sw && yy >= 0 && yy < sh;
            if (in[k] && !staged) {
                const int ax = xx + B.x0;
                bool member = false;
                for (int r = s_row[yy]; r < s_row[yy + 1]; ++r) { const uint32_t q = s_runs[r]; if (ax >= (int)(q & 0xffffu) && ax <= (int)(q >> 16)) { member = true; break; } }
                in[k] = member;
            }
        }
        const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
        for (int ch = 0; ch < och; ++ch) {
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = sx + (k & 1), yy = sy + (k >> 1);
                v[k] = !in[k] ? 0 : staged ? (int)s_img[ch * plane + yy * sw + xx] : source(yy + y0, xx + B.x0, ch);
            }
            out[i * och + ch] = (uint8_t)((v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15);
        }
    }
}

static int ensure_warp(trexhip_ctx* ctx, int n) {
    if (ctx->warp_cap >= n) return TREXHIP_OK;
    if (ctx->d_warp) (void)hipFree(ctx->d_warp);
    ctx->d_warp = nullptr; ctx->warp_cap = 0;
    TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_warp), (size_t)n * 6 * sizeof(double)));
    ctx->warp_cap = n;
    return TREXHIP_OK;
}

// the warp itself, from the per-blob inverse maps in ctx->d_warp
int launch_crops_warp_maps(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode) {
    SegCfg c = ctx->cfg;
    c.invert = ctx->batch_invert; c.zero_bg = ctx->batch_zero_bg;
    c.B = ctx->last_n;
    stage_begin(ctx, TREXHIP_STAGE_CROPS);
    hipLaunchKernelGGL(k_crops_warp, dim3(n), dim3(256), 0, ctx->stream, c, ctx->d_frames, ctx->d_bg, ctx->d_info, ctx->d_blob_frame,
                       ctx->d_blobs, ctx->d_runs, ctx->d_warp, d_crops, OW, OH, diff_mode, ctx->d_color_src, ctx->color_ch,
                       ctx->p.pixel_encoding == TREXHIP_ENC_RGB8 ? 3 : 1, ctx->p.pixel_encoding, ctx->d_bg_color, ctx->bg_color_ch);
    stage_end(ctx, TREXHIP_STAGE_CROPS);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// caller-supplied Midline::transform matrices (host arrays): composed and inverted on the host, uploaded (the copy from pageable
// memory is staged by the runtime before the call returns, so the stack vector may go)
int launch_crops_warp(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode, const float* tr6, const float* lengths,
                      bool legacy, float scale) {
    if (int rc = check_colour_difference(ctx, diff_mode, "normalised crops")) return rc;
    std::vector<double> minv((size_t)n * 6);
    for (int i = 0; i < n; ++i) {
        Aff tr;
        std::memcpy(tr.m, tr6 + (size_t)i * 6, sizeof(tr.m));
        compose_and_invert(tr, lengths ? lengths[i] : 0.f, legacy, OW, OH, scale, &minv[(size_t)i * 6]);
    }
    if (int rc = ensure_warp(ctx, n)) return rc;
    TH_CHECK_HIP(hipMemcpyAsync(ctx->d_warp, minv.data(), minv.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));        // minv lives on this stack frame
    return launch_crops_warp_maps(ctx, d_crops, n, OW, OH, diff_mode);
}

int warp_reserve(trexhip_ctx* ctx, int n) { return ensure_warp(ctx, n); }

}  // namespace trexhip

extern "C" int trexhip_crops_transformed_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                                                const float* transforms, const float* midline_lengths, int32_t use_legacy,
                                                float image_scale, int32_t difference) {
    using namespace trexhip;
    if (!ctx || !d_crops || !transforms) { set_error("trexhip_crops_transformed_device: null argument"); return TREXHIP_E_INVALID; }
    if (out_w <= 0 || out_h <= 0 || difference < 0 || difference > 2) { set_error("trexhip_crops_transformed_device: bad argument"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0 || !ctx->fetched) { set_error("trexhip_crops_transformed_device: segment and fetch a batch first"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_crops_transformed_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    return launch_crops_warp(ctx, d_crops, n_blobs, out_w, out_h, difference, transforms, midline_lengths, use_legacy != 0, image_scale);
}
