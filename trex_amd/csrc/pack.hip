// pack.hip -- pv::Frame::serialize for the frames of a segmented batch, on the device.
//
// Layout = what pv::Frame::read_from accepts for a file of version V_6 (ProcessedVideo/pv.cpp:296-420; serialize :666-703), the newest
// layout whose line type is in the tree: LegacyShortHorizontalLine {u16 x0, u16 x1 << 1 | eol} (pv.h:17-52; used for version < V_7,
// pv.cpp:377-388), 64-bit timestamps (V_4, pv.h:59-64), uncompressed frames (compression_flag 0: pv.cpp:313-316 reads the body in place):
//     u8  compression_flag = 0
//     u64 timestamp (relative to the header's)          u16 n
//     n x { u16 start_y, u16 mask_size, mask_size x {u16 x0, u16 (x1 << 1) | eol}, pixels (1 byte each, gray) }
// eol marks the last line of an image row; the reader counts y up from start_y at every eol (pv.h:20-23,46-49).  Versions >= V_7 use
// commons' ShortHorizontalLine, whose bit layout is not in the tree; the file HEADER (pv.cpp:842-990, DataFormat strings / cv::Size)
// is not written here either -- both stay out until they can be pinned.
// One workgroup per frame: blob sizes -> block scan -> every blob's lines and pixels are copied by the whole workgroup.
#include "internal.h"

namespace trexhip {

__device__ __forceinline__ uint32_t body_bytes(const trexhip_blob& b) { return 4u + 4u * b.n_runs + b.n_pixels; }

// per frame: serialized size; then an exclusive scan over the frames (one workgroup: batches are a few hundred frames)
__global__ __launch_bounds__(256) void k_pack_sizes(const trexhip_frame_info* __restrict__ info, const trexhip_blob* __restrict__ blobs, const int n,
                                                    unsigned long long* __restrict__ offsets) {
    __shared__ unsigned long long s_part[256];
    __shared__ unsigned long long s_run;
    if (threadIdx.x == 0) s_run = 0ull;
    __syncthreads();
    for (int f0 = 0; f0 < n; f0 += 256) {
        const int f = f0 + (int)threadIdx.x;
        unsigned long long sz = 0ull;
        if (f < n) {
            sz = 11ull;
            const trexhip_frame_info fi = info[f];
            if (fi.flags == 0)
                for (uint32_t k = 0; k < fi.n_blobs; ++k) sz += body_bytes(blobs[fi.blob_begin + k]);
        }
        s_part[threadIdx.x] = sz;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = s_run;
            for (int k = 0; k < 256 && f0 + k < n; ++k) { const unsigned long long v = s_part[k]; offsets[f0 + k] = run; run += v; }
            s_run = run;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = s_run;
}

__global__ __launch_bounds__(256) void k_pack(const trexhip_frame_info* __restrict__ info, const trexhip_blob* __restrict__ blobs,
                                              const trexhip_run* __restrict__ runs, const uint8_t* __restrict__ pixels,
                                              const unsigned long long* __restrict__ offsets, const unsigned long long* __restrict__ timestamps,
                                              const unsigned long long capacity, uint8_t* __restrict__ out) {
    __shared__ uint32_t s_scan[256];
    __shared__ uint32_t s_base;
    const int f = blockIdx.x, tid = threadIdx.x;
    const unsigned long long o0 = offsets[f];
    if (offsets[f + 1] > capacity) return;                       // the caller sees offsets[n] > capacity and retries with a larger buffer
    uint8_t* o = out + o0;
    const trexhip_frame_info fi = info[f];
    const uint32_t nb = fi.flags == 0 ? fi.n_blobs : 0u;
    if (tid == 0) {
        o[0] = 0;                                                // compression_flag
        const unsigned long long ts = timestamps ? timestamps[f] : 0ull;
        for (int k = 0; k < 8; ++k) o[1 + k] = (uint8_t)(ts >> (8 * k));
        o[9] = (uint8_t)(nb & 0xffu); o[10] = (uint8_t)(nb >> 8);
        s_base = 11u;
    }
    __syncthreads();
    for (uint32_t b0 = 0; b0 < nb; b0 += 256) {
        const uint32_t b = b0 + (uint32_t)tid;
        const uint32_t sz = b < nb ? body_bytes(blobs[fi.blob_begin + b]) : 0u;
        s_scan[tid] = sz;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                      // inclusive scan of the chunk's blob sizes
            const uint32_t v = tid >= d ? s_scan[tid - d] : 0u;
            __syncthreads();
            s_scan[tid] += v;
            __syncthreads();
        }
        const uint32_t chunk_base = s_base;
        const uint32_t cnt = min(256u, nb - b0);
        for (uint32_t k = 0; k < cnt; ++k) {                     // every blob of the chunk, by the whole workgroup
            const trexhip_blob B = blobs[fi.blob_begin + b0 + k];
            uint8_t* p = o + chunk_base + (k ? s_scan[k - 1] : 0u);
            const trexhip_run* rr = runs + fi.run_begin + B.run_begin;
            if (tid == 0) {
                const uint32_t sy = B.n_runs ? rr[0].y : 0u;
                p[0] = (uint8_t)(sy & 0xffu); p[1] = (uint8_t)(sy >> 8);
                p[2] = (uint8_t)(B.n_runs & 0xffu); p[3] = (uint8_t)(B.n_runs >> 8);
            }
            for (uint32_t j = tid; j < B.n_runs; j += 256) {
                const trexhip_run q = rr[j];
                const uint32_t eol = (j + 1 == B.n_runs || rr[j + 1].y != q.y) ? 1u : 0u;
                const uint32_t x1 = ((uint32_t)q.x1 << 1) | eol;
                uint8_t* l = p + 4 + 4 * j;
                l[0] = (uint8_t)(q.x0 & 0xffu); l[1] = (uint8_t)(q.x0 >> 8); l[2] = (uint8_t)(x1 & 0xffu); l[3] = (uint8_t)(x1 >> 8);
            }
            const uint8_t* px = pixels + fi.pix_begin + B.pix_begin;
            uint8_t* q = p + 4 + 4 * B.n_runs;
            for (uint32_t j = tid; j < B.n_pixels; j += 256) q[j] = px[j];
        }
        __syncthreads();
        if (tid == 0) s_base = chunk_base + s_scan[cnt - 1];
        __syncthreads();
    }
}

}  // namespace trexhip

using namespace trexhip;

extern "C" int trexhip_pack_frames_v6_device(trexhip_ctx* ctx, const uint64_t* timestamps, uint8_t* d_out, size_t capacity, uint64_t* d_offsets) {
    if (!ctx || !d_out || !d_offsets) { set_error("trexhip_pack_frames_v6_device: null argument"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0 || !ctx->fetched) { set_error("trexhip_pack_frames_v6_device: segment and fetch a batch first"); return TREXHIP_E_INVALID; }
    if (ctx->p.pixel_encoding != TREXHIP_ENC_GRAY) { set_error("trexhip_pack_frames_v6_device: the V_6 layout holds one byte per pixel (gray); colour encodings came with V_12"); return TREXHIP_E_UNSUPPORTED; }
    if (ctx->p.width > 32768) { set_error("trexhip_pack_frames_v6_device: LegacyShortHorizontalLine holds x1 < 32768 (pv.h:36)"); return TREXHIP_E_UNSUPPORTED; }
    if (ctx->p.max_blobs > 65535) { set_error("trexhip_pack_frames_v6_device: a frame holds at most 65535 objects (u16 n, pv.cpp:686)"); return TREXHIP_E_UNSUPPORTED; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    const int n = ctx->last_n;
    unsigned long long* d_ts = nullptr;
    if (timestamps) {
        if (ctx->len_cap < 2 * n) {                                 // the context's small scratch buffer (floats): 2 per frame hold a u64
            if (ctx->d_len) (void)hipFree(ctx->d_len);
            ctx->d_len = nullptr; ctx->len_cap = 0;
            TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_len), (size_t)2 * n * sizeof(float)));
            ctx->len_cap = 2 * n;
        }
        d_ts = reinterpret_cast<unsigned long long*>(ctx->d_len);
        TH_CHECK_HIP(hipMemcpyAsync(d_ts, timestamps, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(k_pack_sizes, dim3(1), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blobs, n, reinterpret_cast<unsigned long long*>(d_offsets));
    hipLaunchKernelGGL(k_pack, dim3(n), dim3(256), 0, ctx->stream, ctx->d_info, ctx->d_blobs, ctx->d_runs, ctx->d_pixels,
                       reinterpret_cast<const unsigned long long*>(d_offsets), d_ts, (unsigned long long)capacity, d_out);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}
