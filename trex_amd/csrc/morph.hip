// morph.hip -- optional morphology of the detect stage on 1-bit-per-pixel masks.
//
// RawProcessing::generate_binary applies, after the threshold, `use_closing` (cv::dilate then cv::erode with a
// closing_size ellipse) and `dilation_size` (dilate for > 0, erode for < 0, ellipse of size 2|d|+1)
// (settings: Application/src/tracker/core/default_config.cpp:1163-1165; body in the un-vendored commons --
// semantics chosen where the reference is unpinned: DESIGN.md section 2).
//
//   k_threshold_bits  the k_rows pixel test, but the 16-bit lane masks are packed into a bit image
//   k_morph_bits      binary dilate / erode with a structuring element given as one column span per row
//                     (every row of an OpenCV ellipse is a contiguous span); a workgroup stages a tile of the
//                     bit image plus its halo in LDS and works on 32 pixels per word with funnel shifts
// k_rows then takes the bit image instead of thresholding (segment.hip).
#include "internal.h"

namespace trexhip {

// same per-pixel decision as segment.hip (kept textually separate: different translation unit)
__device__ __forceinline__ uint32_t m_exact4(uint32_t a, uint32_t b, const SegCfg& c) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = (a >> (8 * i)) & 0xff, bg = (b >> (8 * i)) & 0xff;
        const int d = !c.enable_diff ? px : (c.absdiff ? abs(bg - px) : max(bg - px, 0));
        m |= (uint32_t)(d >= c.tmin && d <= c.tmax) << i;      // zero_is_background is applied after the morphology
    }
    return m;
}

// one wave = 1024 pixels of one row; lane l decides pixels [16l, 16l+16) and even lanes store 32-bit words
__global__ __launch_bounds__(256) void k_threshold_bits(const uint8_t* __restrict__ frames, const uint8_t* __restrict__ bg,
                                                        const SegCfg c, uint32_t* __restrict__ bits, int WB) {
    const int lane = threadIdx.x & 63;
    const int nch = (c.W + 1023) / 1024;
    const uint32_t task = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t ntask = (uint32_t)c.B * c.H * nch;
    if (task >= ntask) return;
    const uint32_t ch = task % nch, row = task / nch;          // row = f*H + y
    const uint32_t y = row % c.H;
    const int x = ch * 1024 + lane * 16;
    uint32_t m = 0;
    if (x < c.W) {
        const uint8_t* fp = frames + (size_t)row * c.W + x;
        const uint8_t* bp = bg + (size_t)y * c.W + x;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (x + 4 * w + i < c.W) { a |= (uint32_t)fp[4 * w + i] << (8 * i); b |= (uint32_t)bp[4 * w + i] << (8 * i); }
            if (c.invert) a = ~a;
            if (!c.enable_diff) b = 0;
            uint32_t mm = m_exact4(a, b, c);
            if (x + 4 * w + 4 > c.W) mm &= (1u << max(c.W - x - 4 * w, 0)) - 1u;
            m |= mm << (4 * w);
        }
    }
    const uint32_t hi = __shfl_down(m, 1);
    if ((lane & 1) == 0 && x < c.W) bits[(size_t)row * WB + x / 32] = m | (hi << 16);
}

struct MorphElem { int k; int8_t j1[16], j2[16]; };   // row i of the element covers columns [j1, j2); empty if j2 <= j1

// tile: TH rows x 30 output words per workgroup (32 words incl. one halo word each side), 256 threads
static constexpr int MT_H = 8, MT_W = 30, MT_A = 7;   // halo of up to 7 rows/pixels => elements up to 15x15

__global__ __launch_bounds__(256) void k_morph_bits(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int W, int H,
                                                    int WB, const MorphElem el, const int dilate) {
    __shared__ uint32_t tile[MT_H + 2 * MT_A][MT_W + 2];
    const int f = blockIdx.z;
    const int wy0 = blockIdx.y * MT_H, wx0 = blockIdx.x * MT_W;
    const uint32_t* src = in + (size_t)f * H * WB;
    const uint32_t outside = dilate ? 0u : 0xffffffffu;         // pixels beyond the image never decide (OpenCV default border)
    for (int i = threadIdx.x; i < (MT_H + 2 * MT_A) * (MT_W + 2); i += 256) {
        const int ty = i / (MT_W + 2), tx = i - ty * (MT_W + 2);
        const int y = wy0 + ty - MT_A, wx = wx0 + tx - 1;
        uint32_t v = outside;
        if (y >= 0 && y < H && wx >= 0 && wx < WB) {
            v = src[(size_t)y * WB + wx];
            const int valid = W - wx * 32;                      // bits past the last column of the image
            if (valid < 32) v = dilate ? (v & ((1u << valid) - 1u)) : (v | ~((1u << valid) - 1u));
        }
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int a = el.k / 2;
    for (int i = threadIdx.x; i < MT_H * MT_W; i += 256) {
        const int ty = i / MT_W, tx = i - ty * MT_W;
        const int y = wy0 + ty, wx = wx0 + tx;
        if (y >= H || wx >= WB) continue;
        uint32_t acc = dilate ? 0u : 0xffffffffu;
        for (int r = 0; r < el.k; ++r) {
            const int j1 = el.j1[r], j2 = el.j2[r];
            if (j2 <= j1) continue;
            const uint32_t* row = tile[ty + MT_A + r - a];
            const uint32_t L = row[tx], C = row[tx + 1], R = row[tx + 2];
            for (int j = j1; j < j2; ++j) {
                const int dx = j - a;                            // out(x) takes in(x + dx)
                uint32_t v;
                if (dx == 0) v = C;
                else if (dx > 0) v = (C >> dx) | (R << (32 - dx));
                else v = (C << -dx) | (L >> (32 + dx));
                acc = dilate ? (acc | v) : (acc & v);
            }
        }
        const int valid = W - wx * 32;
        if (valid < 32) acc &= (1u << valid) - 1u;
        out[((size_t)f * H + y) * WB + wx] = acc;
    }
}

// OpenCV getStructuringElement(MORPH_ELLIPSE, Size(k,k)) as per-row spans (OpenCV's published algorithm: morph.dispatch.cpp)
static MorphElem ellipse_spans(int k) {
    MorphElem e{};
    e.k = k;
    const int r = k / 2, c = k / 2;
    const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < k; ++i) {
        int j1 = 0, j2 = 0;
        const int dy = i - r;
        if (abs(dy) <= r) {
            const int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < k ? c + dx + 1 : k;
        }
        e.j1[i] = (int8_t)j1; e.j2[i] = (int8_t)j2;
    }
    return e;
}

// frames -> thresholded, morphed bit image in ctx->d_bits[result]; returns the buffer index (0/1) or < 0
int launch_morphology(trexhip_ctx* ctx, const uint8_t* d_frames, int n, const uint32_t** result) {
    SegCfg c = ctx->cfg;
    c.B = n;
    const int W = c.W, H = c.H, WB = (W + 31) / 32;
    hipStream_t s = ctx->stream;
    const int nch = (W + 1023) / 1024;
    const unsigned ntask = (unsigned)n * H * nch;
    hipLaunchKernelGGL(k_threshold_bits, dim3((ntask + 3) / 4), dim3(256), 0, s, d_frames, ctx->d_bg, c, ctx->d_bits[0], WB);
    int cur = 0;
    const dim3 grid((WB + MT_W - 1) / MT_W, (H + MT_H - 1) / MT_H, n);
    auto pass = [&](const MorphElem& e, int dilate) {
        hipLaunchKernelGGL(k_morph_bits, grid, dim3(256), 0, s, ctx->d_bits[cur], ctx->d_bits[cur ^ 1], W, H, WB, e, dilate);
        cur ^= 1;
    };
    if (ctx->p.use_closing && ctx->p.closing_size > 0) {
        const MorphElem e = ellipse_spans(ctx->p.closing_size);
        pass(e, 1);
        pass(e, 0);
    }
    if (ctx->p.dilation_size != 0) {
        const int d = ctx->p.dilation_size < 0 ? -ctx->p.dilation_size : ctx->p.dilation_size;
        pass(ellipse_spans(2 * d + 1), ctx->p.dilation_size > 0);
    }
    TH_CHECK_HIP(hipGetLastError());
    *result = ctx->d_bits[cur];
    return TREXHIP_OK;
}

}  // namespace trexhip
