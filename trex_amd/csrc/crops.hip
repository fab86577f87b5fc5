// crops.hip -- blob -> 80x80 identity-network input ("individual_image_size"), gathered on the device.
//
// Replaces constraints::diff_image with individual_image_normalization = none
//   Application/src/tracker/tracking/FilterCache.cpp:265-294 -> calculate_diff_image :157-235
// (imageFromLines paints the blob's pixels -- or, with track_background_subtraction, its background
// differences -- into its bounding box; the box is then centre-padded with zeros / centre-cropped to
// the output size: pad = out - size, right = pad/2, left = pad - right (:184-194); cut likewise (:212-226)).
// One workgroup = one blob of the last segmented batch, in pooled order.
#include "internal.h"
#include <cstring>

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
    if (!ctx->fetched) { set_error("trexhip_crops_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
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
    if (!ctx->fetched) { trexhip::set_error("trexhip_export_id_table_device: call trexhip_fetch on the segmented batch first"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipLaunchKernelGGL(trexhip::k_id_table, dim3(max_rows), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs,
                       d_probs, n_blobs, classes, ctx->last_n, frame_base, static_cast<uint32_t*>(d_table), max_rows);
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
        else g = (c[0] * 1868u + c[1] * 9617u + c[2] * 4899u + 8192u) >> 14;
        out |= g << (8 * p);
    }
    reinterpret_cast<uint32_t*>(dst)[i] = out;
}

int launch_to_gray(trexhip_ctx* ctx, const uint8_t* d_color, uint8_t* d_gray, size_t npix, int channels, int color_channel) {
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
        TH_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_staging), (size_t)ctx->p.max_batch * W * H, hipHostMallocDefault));
    }
    if (!ctx->d_color) {
        TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_color), (size_t)ctx->p.max_batch * W * H * 4 + 16));
        TH_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_color), (size_t)ctx->p.max_batch * W * H * 4, hipHostMallocDefault));
    }
    TH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    const size_t row = W * channels;
    for (int i = 0; i < n; ++i) {
        if (!frames[i]) { set_error("trexhip_segment_color: null frame pointer"); return TREXHIP_E_INVALID; }
        uint8_t* dst = ctx->h_color + (size_t)i * H * row;
        for (size_t y = 0; y < H; ++y) std::memcpy(dst + y * row, frames[i] + y * (size_t)stride, row);
    }
    TH_CHECK_HIP(hipMemcpyAsync(ctx->d_color, ctx->h_color, (size_t)n * H * row, hipMemcpyHostToDevice, ctx->stream));
    int rc = launch_to_gray(ctx, ctx->d_color, ctx->d_staging, (size_t)n * W * H, channels, color_channel);
    if (rc) return rc;
    return launch_segment(ctx, ctx->d_staging, n);
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
}  // namespace trexhip

extern "C" int trexhip_generate_average_device(trexhip_ctx* ctx, const uint8_t* d_frames, int32_t n, int32_t method) {
    using namespace trexhip;
    if (!ctx || !d_frames) { set_error("trexhip_generate_average_device: null argument"); return TREXHIP_E_INVALID; }
    if (n < 1) { set_error("trexhip_generate_average_device: need at least one sample"); return TREXHIP_E_INVALID; }
    if (method < 0 || method > 2) { set_error("trexhip_generate_average_device: averaging_method mode is not implemented (0 mean, 1 max, 2 min)"); return TREXHIP_E_UNSUPPORTED; }
    const size_t npix = (size_t)ctx->p.width * ctx->p.height;
    if (npix % 16 != 0 || (reinterpret_cast<uintptr_t>(d_frames) & 15)) { set_error("trexhip_generate_average_device: width*height must be a multiple of 16 and the frames 16-byte aligned"); return TREXHIP_E_UNSUPPORTED; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    const size_t n16 = npix / 16;
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
