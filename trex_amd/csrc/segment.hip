// segment.hip -- detect stage on gfx950: background subtraction + threshold + horizontal-line
// extraction + connected components + size filter + blob gather.
//
// Replaces the body of BackgroundSubtraction::apply (reference:
// Application/src/tracker/python/BackgroundSubtraction.cpp:146-316), i.e.
// RawProcessing::generate_binary (:209) and CPULabeling::run (:216) plus the size filter (:245-291).
//
// Design (DESIGN.md "Kernels"): ONE streaming pass over the pixels (k_rows), everything after it
// works on horizontal lines ("runs"), which are ~1000x fewer than pixels:
//
//   k_rows      1 wave = 1 image row; 16 B/lane coalesced loads of frame and background,
//               v_sad_u8 reject of 4-pixel groups, per-lane 16-bit masks, run starts/ends by
//               bit tricks + wave prefix sum, one atomic per row to reserve output space
//   k_rowscan   per frame exclusive scan of runs-per-row  -> raster index of every row
//   k_link      per row: copy runs to raster order, union with the touching runs of the row above
//   k_flatten   per row: parent -> root label
//   k_blobs     per frame: number blobs in raster order, count, size-filter, reserve pooled
//               output, stable scatter of the runs into per-blob sorted lists
//   k_gather    1 wave = 1 kept blob: gather grey values, bounding box, integer moments, bid
#include "internal.h"
#include <cstddef>
#include <cstdlib>

namespace trexhip {

static constexpr int WAVE = 64;
static constexpr int ROW_SLOT = TREXHIP_ROW_SLOT;   // runs per row kept in the row's own slot
static constexpr int CTR_STRIDE = TREXHIP_CTR_STRIDE;   // per-frame counters live 128 B apart

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

// inclusive prefix sum over groups of LPB lanes with DPP row shifts (lanes outside a row read 0) and the row broadcasts of gfx9
template <int LPB> __device__ __forceinline__ uint32_t group_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    if constexpr (LPB >= 32) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    if constexpr (LPB == 64) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
// inclusive prefix sum inside each row of 16 lanes
__device__ __forceinline__ uint32_t row_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return group_incl_scan<64>(v); }

// exclusive scan over a block of up to 1024 threads; lds must hold >= 16 uint32
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();                       // lds reuse
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    // every wave scans the (<= 16) wave totals itself: lanes 0..nw-1 hold them
    const uint32_t part = lane < nw ? lds[lane] : 0u;
    const uint32_t pi = row_incl_scan(part);
    total = (uint32_t)__builtin_amdgcn_readlane((int)pi, (int)nw - 1);
    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)(pi - part), __builtin_amdgcn_readfirstlane((int)wave));
    return off + incl - v;
}

// three exclusive scans at once (one barrier pair)
__device__ __forceinline__ void block_excl_scan3(uint32_t a, uint32_t b, uint32_t c3, uint32_t* lds, uint32_t* ex, uint32_t* tot) {
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t ia = wave_incl_scan(a), ib = wave_incl_scan(b), ic = wave_incl_scan(c3);
    __syncthreads();
    if (lane == 63) { lds[wave] = ia; lds[16 + wave] = ib; lds[32 + wave] = ic; }
    __syncthreads();
    const uint32_t in[3] = {ia, ib, ic}, v[3] = {a, b, c3};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const uint32_t part = lane < nw ? lds[16 * q + lane] : 0u;
        const uint32_t pi = row_incl_scan(part);
        tot[q] = (uint32_t)__builtin_amdgcn_readlane((int)pi, (int)nw - 1);
        ex[q] = (uint32_t)__builtin_amdgcn_readlane((int)(pi - part), __builtin_amdgcn_readfirstlane((int)wave)) + in[q] - v[q];
    }
}

__device__ __forceinline__ uint32_t wmax32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// k_rows
// ---------------------------------------------------------------------------------------------

// exact per-pixel decision for one 32-bit word (4 pixels); a is already inverted when image_invert
__device__ __forceinline__ uint32_t exact4(uint32_t a, uint32_t b, const SegCfg& c) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = (a >> (8 * i)) & 0xff;
        const int bg = (b >> (8 * i)) & 0xff;
        int d;
        if (!c.enable_diff) d = px;
        else if (c.absdiff) d = abs(bg - px);
        else d = max(bg - px, 0);
        bool pass = d >= c.tmin && d <= c.tmax;
        if (c.zero_bg) pass = pass && px != 0;
        m |= (uint32_t)pass << i;
    }
    return m;
}

// The common settings (background subtraction on, no image_invert, no threshold_maximum, threshold >= 1)
// as compile-time modes (zero_is_background as a mode bit): the generic exact4 computes both difference rules and both bounds for every byte and selects (12 VALU
// instructions per pixel; it is executed by the whole wave as soon as one lane's word could pass), the specialised one needs 5.
//   MODE & 3 = 1: |bg - px| >= tmin (track_absolute_difference), 2: bg - px >= tmin;  MODE & 4: and px != 0 (zero_is_background)
template <int MODE>
__device__ __forceinline__ uint32_t exact4_fast(uint32_t a, uint32_t b, int tmin) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = (a >> (8 * i)) & 0xff;
        const int bg = (b >> (8 * i)) & 0xff;
        const int d = bg - px;
        // |d| >= t  <=>  d is outside [-(t - 1), t - 1]  <=>  (unsigned)(d + t - 1) > 2 (t - 1): one add and one compare
        bool pass = (MODE & 3) == 1 ? (uint32_t)(d + tmin - 1) > (uint32_t)(2 * (tmin - 1)) : d >= tmin;
        if (MODE & 4) pass = pass && px != 0;                 // zero_is_background
        m |= (uint32_t)pass << i;
    }
    return m;
}

template <int MODE>
__device__ __forceinline__ uint32_t mask16_fast(uint4 a, uint4 b, const SegCfg& c) {
    const uint32_t s0 = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
    const uint32_t s1 = __builtin_amdgcn_sad_u8(a.y, b.y, 0u);
    const uint32_t s2 = __builtin_amdgcn_sad_u8(a.z, b.z, 0u);
    const uint32_t s3 = __builtin_amdgcn_sad_u8(a.w, b.w, 0u);
    const uint32_t t = (uint32_t)c.tmin;
    uint32_t m = 0;
    if (max(max(s0, s1), max(s2, s3)) >= t) {
        if (s0 >= t) m |= exact4_fast<MODE>(a.x, b.x, c.tmin);
        if (s1 >= t) m |= exact4_fast<MODE>(a.y, b.y, c.tmin) << 4;
        if (s2 >= t) m |= exact4_fast<MODE>(a.z, b.z, c.tmin) << 8;
        if (s3 >= t) m |= exact4_fast<MODE>(a.w, b.w, c.tmin) << 12;
    }
    return m;
}

// 16 pixels of one lane -> 16-bit foreground mask.  A 4-pixel word is only evaluated exactly when
// the sum of its absolute differences (one v_sad_u8) could reach the threshold.
__device__ __forceinline__ uint32_t mask16(uint4 a, uint4 b, const SegCfg& c) {
    if (c.invert) { a.x = ~a.x; a.y = ~a.y; a.z = ~a.z; a.w = ~a.w; }
    if (!c.enable_diff) { b.x = b.y = b.z = b.w = 0; }
    const uint32_t s0 = __builtin_amdgcn_sad_u8(a.x, b.x, 0u);
    const uint32_t s1 = __builtin_amdgcn_sad_u8(a.y, b.y, 0u);
    const uint32_t s2 = __builtin_amdgcn_sad_u8(a.z, b.z, 0u);
    const uint32_t s3 = __builtin_amdgcn_sad_u8(a.w, b.w, 0u);
    const uint32_t t = (uint32_t)c.tmin;
    uint32_t m = 0;
    if (max(max(s0, s1), max(s2, s3)) >= t) {
        if (s0 >= t) m |= exact4(a.x, b.x, c);
        if (s1 >= t) m |= exact4(a.y, b.y, c) << 4;
        if (s2 >= t) m |= exact4(a.z, b.z, c) << 8;
        if (s3 >= t) m |= exact4(a.w, b.w, c) << 12;
    }
    return m;
}

template <bool ALIGNED>
__device__ __forceinline__ uint4 load16(const uint8_t* p, int x, int W) {
    if (ALIGNED) {
        return *reinterpret_cast<const uint4*>(p + x);
    } else {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (x + i < W) w[i >> 2] |= (uint32_t)p[x + i] << (8 * (i & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// One wave = one image row at a time; waves walk the (frame,row) tasks with a grid stride and
// prefetch the next row's 16-byte loads before they process the current one.
//   order 0: frame index fastest (concurrent waves share background rows)
//   order 1: row index fastest   (each frame is streamed front to back)
template <int NCH, bool ALIGNED>
__global__ __launch_bounds__(256) void k_rows(const uint8_t* __restrict__ frames,
                                              const uint8_t* __restrict__ bg, const SegCfg c, const int order,
                                              uint32_t* __restrict__ frame_ctr,
                                              uint32_t* __restrict__ row_cnt,
                                              uint32_t* __restrict__ row_off,
                                              uint32_t* __restrict__ tmp_runs, const uint32_t* __restrict__ bits, const uint32_t f0) {
    const int lane = lane_id();
    // the pooled totals of the pass (blobs, runs, pixels) start at zero: written here, ahead of the labelling kernel in stream order (the
    // per-frame counters are handed back zeroed by k_ccl_lds, so a pass needs no memset)
    if (f0 == 0u && !(order & (1 << 30)) && blockIdx.x == 0 && threadIdx.x < 4) frame_ctr[(size_t)c.ctr_frames * CTR_STRIDE + threadIdx.x] = 0u;
    const int W = c.W;
    const int WB = (W + 31) / 32;
    const uint32_t ntask = (uint32_t)c.B * (uint32_t)c.H;
    const uint32_t nwave = gridDim.x * 4u;
    uint32_t task = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (task >= ntask) return;

    uint4 a[NCH], b[NCH];
    // (frame, row) of a task without a division per row: the wave's first task is divided once, every further one is the previous
    // plus a constant step with one carry (the two udiv sequences were ~40 % of the instructions of a row without foreground)
    const bool frame_fastest = (order & 1) == 0;
    const uint32_t modulus = frame_fastest ? (uint32_t)c.B : (uint32_t)c.H;           // c.B = frames of this launch
    const uint32_t step_lo = nwave % modulus, step_hi = nwave / modulus;
    uint32_t cur_lo = task % modulus, cur_hi = task / modulus;                      // lo = fastest index, hi = the other
    auto advance = [&](uint32_t& lo, uint32_t& hi) { lo += step_lo; hi += step_hi; if (lo >= modulus) { lo -= modulus; ++hi; } };
    auto issue = [&](uint32_t lo, uint32_t hi) {
        const uint32_t f = f0 + (frame_fastest ? lo : hi);
        const uint32_t y = frame_fastest ? hi : lo;
        const uint8_t* fp = frames + ((size_t)f * c.H + y) * W;
        const uint8_t* bp = bg + (size_t)y * W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int x = ch * 1024 + lane * 16;
            if (x < W) {
                a[ch] = load16<ALIGNED>(fp, x, W);
                b[ch] = load16<ALIGNED>(bp, x, W);
            } else {
                a[ch] = make_uint4(0, 0, 0, 0);
                b[ch] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    issue(cur_lo, cur_hi);
    for (; task < ntask; task += nwave) {
        const uint32_t f = f0 + (frame_fastest ? cur_lo : cur_hi);
        const uint32_t y = frame_fastest ? cur_hi : cur_lo;
        advance(cur_lo, cur_hi);                             // now the next task's coordinates
        uint32_t m[NCH];
        bool any = false;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int x = ch * 1024 + lane * 16;
            uint32_t mm = 0;
            if (x < W) {
                if (bits) {                       // thresholded + morphed mask computed by morph.hip; still "grey under mask"
                    const uint16_t* bh = reinterpret_cast<const uint16_t*>(bits + ((size_t)f * c.H + y) * WB);
                    mm = bh[x >> 4];
                    if (c.zero_bg) {
                        const uint32_t w4[4] = {a[ch].x, a[ch].y, a[ch].z, a[ch].w};
                        uint32_t nz = 0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            uint32_t px = (w4[q >> 2] >> (8 * (q & 3))) & 0xffu;
                            if (c.invert) px = 255u - px;
                            nz |= (uint32_t)(px != 0u) << q;
                        }
                        mm &= nz;
                    }
                } else
#ifdef TREXHIP_DEV_KNOBS
                mm = (order & 512) ? ((a[ch].x ^ b[ch].x) == 0x12345u) : mask16(a[ch], b[ch], c);
#else
                mm = mask16(a[ch], b[ch], c);
#endif
                if (!ALIGNED && x + 16 > W) mm &= (1u << (W - x)) - 1u;
            }
            m[ch] = mm;
            any |= mm != 0;
        }
#ifdef TREXHIP_DEV_KNOBS
        if (order & 256) any = false;
#endif
        // registers a/b are free again: prefetch the next row while this one is finished
        if (task + nwave < ntask) issue(cur_lo, cur_hi);

        const size_t ri = (size_t)f * c.H + y;
        if (!__any(any)) {                       // most rows: no foreground at all
            if (lane == 0) { row_cnt[ri] = 0; row_off[ri] = 0; }
            continue;
        }
        // run starts / ends per chunk.  bit j of en = "a run ended at pixel j-1".
        uint32_t st[NCH], en[NCH], pre[NCH];     // pre = exclusive prefix (starts | ends << 16) incl. earlier chunks
        uint32_t carry = 0, tot = 0;             // carry: last pixel of previous chunk set; tot: packed row totals
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const uint32_t mm = m[ch];
            if (__ballot(mm != 0) == 0 && carry == 0) { st[ch] = 0; en[ch] = 0; pre[ch] = tot; continue; }
            uint32_t up = __shfl_up(mm >> 15, 1);
            if (lane == 0) up = carry;
            const uint32_t prevmask = ((mm << 1) | (up & 1u)) & 0xffffu;
            const uint32_t s = mm & ~prevmask;
            const uint32_t e = ~mm & prevmask;
            const uint32_t v = __popc(s) | (__popc(e) << 16);
            const uint32_t incl = wave_incl_scan(v);
            st[ch] = s; en[ch] = e; pre[ch] = tot + incl - v;
            tot += __shfl(incl, 63);
            carry = __shfl(mm >> 15, 63) & 1u;
        }
        const uint32_t n_starts = tot & 0xffffu;     // runs in this row (== ends + carry)
        // output space: every row owns a fixed slot of ROW_SLOT runs (no atomics: device-scope atomics on a
        // handful of counters cap at ~90 ops/us per word and were 80% of this kernel); only rows with more
        // runs than that reserve space in the per-frame overflow area with one atomic.
        uint32_t base = y * (uint32_t)ROW_SLOT;
        if (n_starts > (uint32_t)ROW_SLOT) {
            if (lane == 0) base = (uint32_t)c.H * ROW_SLOT + atomicAdd(&frame_ctr[f * CTR_STRIDE], n_starts);
            base = __shfl(base, 0);
        }
        if (lane == 0) { row_cnt[ri] = n_starts; row_off[ri] = base; }
        uint16_t* out = reinterpret_cast<uint16_t*>(tmp_runs + (size_t)f * c.T);
        const uint32_t R = (uint32_t)c.T;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int xb = ch * 1024 + lane * 16;
            uint32_t s = st[ch], e = en[ch];
            uint32_t ks = base + (pre[ch] & 0xffffu), ke = base + (pre[ch] >> 16);
            while (s) {
                const int j = __ffs(s) - 1; s &= s - 1;
                if (ks < R) out[2 * ks] = (uint16_t)(xb + j);
                ++ks;
            }
            while (e) {
                const int j = __ffs(e) - 1; e &= e - 1;
                if (ke < R) out[2 * ke + 1] = (uint16_t)(xb + j - 1);
                ++ke;
            }
        }
        if (carry && lane == 0) {                // run open at the end of the row closes at W-1
            const uint32_t ke = base + (tot >> 16);
            if (ke < R) out[2 * ke + 1] = (uint16_t)(W - 1);
        }
    }
}

// The same pass with 32 pixels per lane (two 16-byte loads per array): one wave covers 2048 pixels per chunk, so a 2048-wide row is
// ONE chunk -- one ballot / scan / extraction round per row instead of two.  Aligned frames of a width that is a multiple of 32, no
// morphology mask; everything else takes k_rows above.
// masks of one row-task: 32 pixels per lane per 2048-pixel chunk
template <int NCH, int MODE>
__device__ __forceinline__ bool rows32_masks(const uint4 (&a)[NCH][2], const uint4 (&b)[NCH][2], const SegCfg& c, const int lane, const int W,
                                             const int order, uint32_t (&m)[NCH]) {
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int x = ch * 2048 + lane * 32;
        uint32_t mm = 0;
        if (x < W) {
#ifdef TREXHIP_DEV_KNOBS
            if (order & 4096) mm = ((a[ch][0].x ^ b[ch][0].x ^ a[ch][1].y ^ b[ch][1].w) == 0x12345u);      // profiling aid: stream only
            else
#endif
            if constexpr (MODE == 0) mm = mask16(a[ch][0], b[ch][0], c) | (mask16(a[ch][1], b[ch][1], c) << 16);
            else mm = mask16_fast<MODE>(a[ch][0], b[ch][0], c) | (mask16_fast<MODE>(a[ch][1], b[ch][1], c) << 16);
        }
        m[ch] = mm;
        any |= mm != 0;
    }
#ifdef TREXHIP_DEV_KNOBS
    if (order & 8192) any = false;                                                                            // profiling aid: no run extraction
#endif
    return any;
}

// run extraction of one row-task from its masks: row_cnt / row_off and the (x0, x1) pairs in the row's slots (or the overflow area)
template <int NCH>
__device__ __forceinline__ void rows32_emit(const uint32_t (&m)[NCH], const bool any, const SegCfg& c, const uint32_t f, const uint32_t y, const int lane,
                                            const int W, uint32_t* __restrict__ frame_ctr, uint32_t* __restrict__ row_cnt,
                                            uint32_t* __restrict__ row_off, uint32_t* __restrict__ tmp_runs) {
    const size_t ri = (size_t)f * c.H + y;
    if (!__any(any)) {
        if (lane == 0) { row_cnt[ri] = 0; row_off[ri] = 0; }
        return;
    }
    uint32_t st[NCH], en[NCH], pre[NCH];
    uint32_t carry = 0, tot = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const uint32_t mm = m[ch];
        if (NCH > 1 && __ballot(mm != 0) == 0 && carry == 0) { st[ch] = 0; en[ch] = 0; pre[ch] = tot; continue; }
        uint32_t up = __shfl_up(mm >> 31, 1);
        if (lane == 0) up = carry;
        const uint32_t prevmask = (mm << 1) | (up & 1u);
        const uint32_t s = mm & ~prevmask;
        const uint32_t e = ~mm & prevmask;
        const uint32_t v = __popc(s) | (__popc(e) << 16);
        const uint32_t incl = wave_incl_scan(v);
        st[ch] = s; en[ch] = e; pre[ch] = tot + incl - v;
        tot += __shfl(incl, 63);
        carry = __shfl(mm >> 31, 63) & 1u;
    }
    const uint32_t n_starts = tot & 0xffffu;
    uint32_t base = y * (uint32_t)ROW_SLOT;
    if (n_starts > (uint32_t)ROW_SLOT) {
        if (lane == 0) base = (uint32_t)c.H * ROW_SLOT + atomicAdd(&frame_ctr[f * CTR_STRIDE], n_starts);
        base = __shfl(base, 0);
    }
    if (lane == 0) { row_cnt[ri] = n_starts; row_off[ri] = base; }
    uint16_t* out = reinterpret_cast<uint16_t*>(tmp_runs + (size_t)f * c.T);
    const uint32_t R = (uint32_t)c.T;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int xb = ch * 2048 + lane * 32;
        uint32_t s = st[ch], e = en[ch];
        uint32_t ks = base + (pre[ch] & 0xffffu), ke = base + (pre[ch] >> 16);
        while (s) {
            const int j = __ffs(s) - 1; s &= s - 1;
            if (ks < R) out[2 * ks] = (uint16_t)(xb + j);
            ++ks;
        }
        while (e) {
            const int j = __ffs(e) - 1; e &= e - 1;
            if (ke < R) out[2 * ke + 1] = (uint16_t)(xb + j - 1);
            ++ke;
        }
    }
    if (carry && lane == 0) {
        const uint32_t ke = base + (tot >> 16);
        if (ke < R) out[2 * ke + 1] = (uint16_t)(W - 1);
    }
}

template <int NCH, int MODE = 0>
__global__ __launch_bounds__(256) void k_rows32(const uint8_t* __restrict__ frames,
                                                const uint8_t* __restrict__ bg, const SegCfg c, const int order,
                                                uint32_t* __restrict__ frame_ctr,
                                                uint32_t* __restrict__ row_cnt,
                                                uint32_t* __restrict__ row_off,
                                                uint32_t* __restrict__ tmp_runs, const uint32_t f0) {
    const int lane = lane_id();
    // the pooled totals of the pass (blobs, runs, pixels) start at zero: written here, ahead of the labelling kernel in stream order (the
    // per-frame counters are handed back zeroed by k_ccl_lds, so a pass needs no memset)
    if (f0 == 0u && !(order & (1 << 30)) && blockIdx.x == 0 && threadIdx.x < 4) frame_ctr[(size_t)c.ctr_frames * CTR_STRIDE + threadIdx.x] = 0u;
    const int W = c.W;
    const uint32_t ntask = (uint32_t)c.B * (uint32_t)c.H;
    const uint32_t nwave = gridDim.x * 4u;
    uint32_t task = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (task >= ntask) return;

    uint4 a[NCH][2], b[NCH][2];
    const bool frame_fastest = (order & 1) == 0;
    const uint32_t modulus = frame_fastest ? (uint32_t)c.B : (uint32_t)c.H;
    const uint32_t step_lo = nwave % modulus, step_hi = nwave / modulus;
    uint32_t cur_lo = task % modulus, cur_hi = task / modulus;
    auto advance = [&](uint32_t& lo, uint32_t& hi) { lo += step_lo; hi += step_hi; if (lo >= modulus) { lo -= modulus; ++hi; } };
    auto issue = [&](uint32_t lo, uint32_t hi) {
        const uint32_t f = f0 + (frame_fastest ? lo : hi);
        const uint32_t y = frame_fastest ? hi : lo;
        const uint8_t* fp = frames + ((size_t)f * c.H + y) * W;
        const uint8_t* bp = bg + (size_t)y * W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int x = ch * 2048 + lane * 32;
            if (x < W) {
                a[ch][0] = *reinterpret_cast<const uint4*>(fp + x); a[ch][1] = *reinterpret_cast<const uint4*>(fp + x + 16);
                b[ch][0] = *reinterpret_cast<const uint4*>(bp + x); b[ch][1] = *reinterpret_cast<const uint4*>(bp + x + 16);
            } else {
                a[ch][0] = a[ch][1] = b[ch][0] = b[ch][1] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    issue(cur_lo, cur_hi);
    for (; task < ntask; task += nwave) {
        const uint32_t f = f0 + (frame_fastest ? cur_lo : cur_hi);
        const uint32_t y = frame_fastest ? cur_hi : cur_lo;
        advance(cur_lo, cur_hi);
        uint32_t m[NCH];
        const bool any = rows32_masks<NCH, MODE>(a, b, c, lane, W, order, m);
        if (task + nwave < ntask) issue(cur_lo, cur_hi);
        rows32_emit<NCH>(m, any, c, f, y, lane, W, frame_ctr, row_cnt, row_off, tmp_runs);
    }
}

// The wide pass with the background row held in registers: a wave takes row y of K consecutive frames (K divides the number of
// frames of the launch), so the 2 KB background row is loaded once per K frame rows instead of once per frame row -- half the load
// instructions and half the L2 -> CU traffic of the kernel above.  Wave w: row w % H, frames (w / H) * K ...
template <int NCH, int MODE = 0>
__global__ __launch_bounds__(256) void k_rows32b(const uint8_t* __restrict__ frames,
                                                 const uint8_t* __restrict__ bg, const SegCfg c, const int order, const int K,
                                                 uint32_t* __restrict__ frame_ctr,
                                                 uint32_t* __restrict__ row_cnt,
                                                 uint32_t* __restrict__ row_off,
                                                 uint32_t* __restrict__ tmp_runs, const uint32_t f0) {
    const int lane = lane_id();
    // the pooled totals of the pass (blobs, runs, pixels) start at zero: written here, ahead of the labelling kernel in stream order (the
    // per-frame counters are handed back zeroed by k_ccl_lds, so a pass needs no memset)
    if (f0 == 0u && !(order & (1 << 30)) && blockIdx.x == 0 && threadIdx.x < 4) frame_ctr[(size_t)c.ctr_frames * CTR_STRIDE + threadIdx.x] = 0u;
    const int W = c.W;
    const uint32_t groups = (uint32_t)c.B / (uint32_t)K;
    const uint32_t wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wid >= groups * (uint32_t)c.H) return;
    uint32_t y = wid / groups, fb = f0 + (wid - y * groups) * (uint32_t)K;
    // rows fastest (the default): neighbouring waves read neighbouring rows of the same frames.  Measured against frame groups fastest
    // (TREXHIP_ROWS_ORDER bit 0) alternated on three boxes: 215 against 220 us on two of them, 189 against 215 on the third
    if (!(order & 1)) { const uint32_t gq = wid / (uint32_t)c.H; y = wid - gq * (uint32_t)c.H; fb = f0 + gq * (uint32_t)K; }
    uint4 a[NCH][2], b[NCH][2];
    const uint8_t* bp = bg + (size_t)y * W;
    auto issue = [&](uint32_t f) {
        const uint8_t* fp = frames + ((size_t)f * c.H + y) * W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int x = ch * 2048 + lane * 32;
            if constexpr (NCH == 1) {
                if (x < W) { a[ch][0] = *reinterpret_cast<const uint4*>(fp + x); a[ch][1] = *reinterpret_cast<const uint4*>(fp + x + 16); }
                else a[ch][0] = a[ch][1] = make_uint4(0, 0, 0, 0);
            } else {
                // (several chunks: hipcc 7.2 crashes -- Machine Copy Propagation, or Post-RA pseudo expansion -- on the conditional 128-bit assignments
                // above; a clamped load and a word mask say the same)
                const int xs = x < W ? x : 0;
                const uint32_t km = x < W ? 0xffffffffu : 0u;
                uint4 v0 = *reinterpret_cast<const uint4*>(fp + xs), v1 = *reinterpret_cast<const uint4*>(fp + xs + 16);
                v0.x &= km; v0.y &= km; v0.z &= km; v0.w &= km; v1.x &= km; v1.y &= km; v1.z &= km; v1.w &= km;
                a[ch][0] = v0; a[ch][1] = v1;
            }
        }
    };
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int x = ch * 2048 + lane * 32;
        if constexpr (NCH == 1) {
            if (x < W) { b[ch][0] = *reinterpret_cast<const uint4*>(bp + x); b[ch][1] = *reinterpret_cast<const uint4*>(bp + x + 16); }
            else b[ch][0] = b[ch][1] = make_uint4(0, 0, 0, 0);
        } else {
            const int xs = x < W ? x : 0;
            const uint32_t km = x < W ? 0xffffffffu : 0u;
            uint4 v0 = *reinterpret_cast<const uint4*>(bp + xs), v1 = *reinterpret_cast<const uint4*>(bp + xs + 16);
            v0.x &= km; v0.y &= km; v0.z &= km; v0.w &= km; v1.x &= km; v1.y &= km; v1.z &= km; v1.w &= km;
            b[ch][0] = v0; b[ch][1] = v1;
        }
    }
    issue(fb);
    for (int k = 0; k < K; ++k) {
        uint32_t m[NCH];
        const bool any = rows32_masks<NCH, MODE>(a, b, c, lane, W, order, m);
        if (k + 1 < K) issue(fb + k + 1);
        rows32_emit<NCH>(m, any, c, fb + k, y, lane, W, frame_ctr, row_cnt, row_off, tmp_runs);
    }
}

// ---------------------------------------------------------------------------------------------
// k_rowscan: exclusive scan of runs-per-row, parent init
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rowscan(const SegCfg c, const int only_pending, const uint32_t* __restrict__ frame_ctr, const uint32_t* __restrict__ row_cnt,
                                                 uint32_t* __restrict__ row_base,
                                                 uint32_t* __restrict__ parent,
                                                 trexhip_frame_info* __restrict__ info) {
    __shared__ uint32_t lds[8];
    const int f = blockIdx.x;
    if (only_pending && info[f].reserved[0] != 1u) return;      // frame already finished by k_ccl_lds
    const uint32_t* cnt = row_cnt + (size_t)f * c.H;
    uint32_t* rb = row_base + (size_t)f * (c.H + 1);
    uint32_t running = 0;
    for (int y0 = 0; y0 < c.H; y0 += 256) {
        const int y = y0 + threadIdx.x;
        const uint32_t v = y < c.H ? cnt[y] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, lds, total);
        if (y < c.H) rb[y] = running + ex;
        running += total;
    }
    const uint32_t n = running;
    const bool overflow = n > (uint32_t)c.R || frame_ctr[f * CTR_STRIDE] > (uint32_t)c.R;
    if (threadIdx.x == 0) {
        rb[c.H] = n;
        trexhip_frame_info fi = {};
        fi.n_raw_runs = n;
        fi.flags = overflow ? TREXHIP_FRAME_OVERFLOW_RUNS : 0u;
        fi.reserved[0] = only_pending ? 2u : 0u;
        info[f] = fi;
    }
    if (!overflow) {
        uint32_t* p = parent + (size_t)f * c.R;
        for (uint32_t r = threadIdx.x; r < n; r += 256) p[r] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// union-find on run indices (root = smallest raster index of the component)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uf_find(const uint32_t* parent, uint32_t a) {
    uint32_t p = ld_relaxed(parent + a);
    while (p != a) { a = p; p = ld_relaxed(parent + a); }
    return a;
}
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(parent + b, a);   // hook the larger root under the smaller
        if (old == b) return;
        b = old;                                         // b was hooked meanwhile: merge its parent too
    }
}

// one thread = one image row: copy its runs to raster order, link with the row above
__global__ __launch_bounds__(256) void k_link(const SegCfg c, const int only_pending, const uint32_t* __restrict__ row_cnt,
                                              const uint32_t* __restrict__ row_off,
                                              const uint32_t* __restrict__ row_base,
                                              const uint32_t* __restrict__ tmp_runs,
                                              trexhip_run* __restrict__ raster, uint32_t* __restrict__ parent,
                                              const trexhip_frame_info* __restrict__ info) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= c.B * c.H) return;
    const int f = gid / c.H, y = gid - f * c.H;
    if (info[f].flags || (only_pending && info[f].reserved[0] != 2u)) return;
    const size_t ri = (size_t)f * c.H + y;
    const uint32_t cnt = row_cnt[ri];
    if (!cnt) return;
    const uint32_t* tmp = tmp_runs + (size_t)f * c.T;
    const uint32_t off = row_off[ri];
    const uint32_t base = row_base[(size_t)f * (c.H + 1) + y];
    trexhip_run* rr = raster + (size_t)f * c.R;
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t t = tmp[off + i];
        trexhip_run r; r.x0 = (uint16_t)(t & 0xffffu); r.x1 = (uint16_t)(t >> 16); r.y = (uint16_t)y; r.pad = 0;
        rr[base + i] = r;
    }
    if (y == 0) return;
    const uint32_t pcnt = row_cnt[ri - 1];
    if (!pcnt) return;
    const uint32_t poff = row_off[ri - 1];
    const uint32_t pbase = row_base[(size_t)f * (c.H + 1) + y - 1];
    uint32_t* par = parent + (size_t)f * c.R;
    uint32_t i = 0, j = 0;
    uint32_t cur = tmp[off], prv = tmp[poff];
    const int slack = c.slack;
    for (;;) {
        const int c0 = cur & 0xffffu, c1 = cur >> 16, p0 = prv & 0xffffu, p1 = prv >> 16;
        if (p1 + slack >= c0 && c1 + slack >= p0) uf_union(par, pbase + j, base + i);
        if (p1 < c1) { if (++j >= pcnt) break; prv = tmp[poff + j]; }
        else         { if (++i >= cnt) break;  cur = tmp[off + i]; }
    }
}

__global__ __launch_bounds__(256) void k_flatten(const SegCfg c, const int only_pending, const uint32_t* __restrict__ row_cnt,
                                                 const uint32_t* __restrict__ row_base,
                                                 uint32_t* __restrict__ parent,
                                                 const trexhip_frame_info* __restrict__ info) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= c.B * c.H) return;
    const int f = gid / c.H, y = gid - f * c.H;
    if (info[f].flags || (only_pending && info[f].reserved[0] != 2u)) return;
    const uint32_t cnt = row_cnt[(size_t)f * c.H + y];
    if (!cnt) return;
    const uint32_t base = row_base[(size_t)f * (c.H + 1) + y];
    uint32_t* par = parent + (size_t)f * c.R;
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t root = uf_find(par, base + i);
        __hip_atomic_store(par + base + i, root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------
// k_blobs: one block per frame
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool size_ok(uint32_t npx, const SegCfg& c) {
    if (c.n_ranges <= 0) return true;                       // SizeFilters.cpp:38
    const double v = (double)((float)npx * c.sqcm);         // BackgroundSubtraction.cpp:139,259
    for (int i = 0; i < c.n_ranges; ++i)
        if (v >= c.ranges[2 * i] && v < c.ranges[2 * i + 1]) return true;
    return false;
}

static constexpr int CURSOR_LDS = 8192;   // kept-blob run cursors held in LDS (else global)

__global__ __launch_bounds__(256) void k_blobs(const SegCfg c, const int only_pending, const trexhip_run* __restrict__ raster,
                                               const uint32_t* __restrict__ parent,
                                               uint32_t* __restrict__ root_ord, uint32_t* __restrict__ cnt_runs,
                                               uint32_t* __restrict__ cnt_px, uint32_t* __restrict__ cur_run,
                                               uint32_t* __restrict__ pix_begin, int32_t* __restrict__ blob_map, uint32_t* __restrict__ totals,
                                               trexhip_frame_info* __restrict__ info,
                                               trexhip_blob* __restrict__ blobs, uint32_t* __restrict__ blob_frame,
                                               trexhip_run* __restrict__ out_runs, const int classify,
                                               const uint32_t* __restrict__ run_parent) {
    __shared__ uint32_t lds[8];
    __shared__ uint32_t s_alloc[4];
    __shared__ uint32_t s_cursor[CURSOR_LDS];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    trexhip_frame_info fi = info[f];
    if (fi.flags || (only_pending && fi.reserved[0] != 2u)) return;
    const uint32_t n = fi.n_raw_runs;
    const size_t fo = (size_t)f * c.R;
    const trexhip_run* rr = raster + fo;
    const uint32_t* lab = parent + fo;
    uint32_t* ord = root_ord + fo;
    uint32_t* cr = cnt_runs + fo;
    uint32_t* cp = cnt_px + fo;
    uint32_t* cur = cur_run + fo;
    uint32_t* pbg = pix_begin + fo;
    int32_t* bmap = blob_map + fo;

    // A: ordinal of every raw blob = rank of its root run in raster order
    uint32_t nraw = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 256) {
        const uint32_t r = b0 + tid;
        const uint32_t flag = (r < n && lab[r] == r) ? 1u : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(flag, lds, total);
        if (flag) ord[r] = nraw + ex;
        nraw += total;
    }
    // counters live in L2 (atomics): zero them with L1-bypassing stores and drain before the barrier
    for (uint32_t o = tid; o < nraw; o += 256) {
        __hip_atomic_store(cr + o, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cp + o, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // B: runs / pixels per raw blob
    for (uint32_t r = tid; r < n; r += 256) {
        const uint32_t o = ord[lab[r]];
        const trexhip_run q = rr[r];
        atomicAdd(cr + o, 1u);
        atomicAdd(cp + o, (uint32_t)(q.x1 - q.x0 + 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // C: size filter (BackgroundSubtraction.cpp:259) + "< UINT16_MAX lines" (:306); offsets of kept blobs
    uint32_t kept = 0, kruns = 0, kpx = 0;
    for (uint32_t b0 = 0; b0 < nraw; b0 += 256) {
        const uint32_t o = b0 + tid;
        uint32_t nr = 0, np = 0, keep = 0;
        if (o < nraw) {
            nr = ld_relaxed(cr + o); np = ld_relaxed(cp + o);
            keep = ((classify || size_ok(np, c)) && nr < 65535u) ? 1u : 0u;
        }
        uint32_t t0, t1, t2;
        const uint32_t e0 = block_excl_scan(keep, lds, t0);
        const uint32_t e1 = block_excl_scan(keep ? nr : 0u, lds, t1);
        const uint32_t e2 = block_excl_scan(keep ? np : 0u, lds, t2);
        if (o < nraw) {
            bmap[o] = keep ? (int32_t)(kept + e0) : -1;
            if (keep) { cur[o] = kruns + e1; pbg[o] = kpx + e2; }
        }
        kept += t0; kruns += t1; kpx += t2;
    }
    // a frame beyond its own capacities fails alone and reserves nothing: the pool (max_batch x the per-frame capacities) then
    // always has room for the other frames of the batch
    if (kept > c.cap_blobs || kpx > c.cap_pixels) {
        fi.n_raw_blobs = nraw;
        if (tid == 0) { fi.flags |= TREXHIP_FRAME_OVERFLOW_OUTPUT; info[f] = fi; }
        return;
    }
    // reserve pooled output (blobs, runs, pixels) for this frame
    if (tid == 0) {
        const uint32_t bb = atomicAdd(totals + 0, kept);
        const uint32_t rb = atomicAdd(totals + 1, kruns);
        const uint32_t pb = atomicAdd(totals + 2, kpx);
        const bool over = bb + kept > c.pool_blobs || rb + kruns > c.pool_runs || pb + kpx > c.pool_pixels;
        s_alloc[0] = bb; s_alloc[1] = rb; s_alloc[2] = pb; s_alloc[3] = over ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t bb = s_alloc[0], rb = s_alloc[1], pb = s_alloc[2];
    fi.n_raw_blobs = nraw;
    if (s_alloc[3]) {
        // the reservation cannot be undone (other frames may have reserved behind it): leave holes that
        // k_gather and the host skip
        for (uint32_t k = tid; k < kept; k += 256)
            if (bb + k < c.pool_blobs) blob_frame[bb + k] = 0xffffffffu;
        if (tid == 0) { fi.flags |= TREXHIP_FRAME_OVERFLOW_OUTPUT; info[f] = fi; }
        return;
    }
    // blob records (counts / offsets; k_gather fills the rest)
    for (uint32_t o = tid; o < nraw; o += 256) {
        const int32_t k = bmap[o];
        if (k < 0) continue;
        trexhip_blob B = {};
        B.run_begin = cur[o] + rb;        // POOLED offsets until k_gather, which needs no frame table to find the lines, makes them frame-relative
        B.n_runs = ld_relaxed(cr + o);
        B.pix_begin = pbg[o] + pb;
        B.parent = 0xffffffffu;
        if (classify) {   // Tracker.cpp:864-912: in range -> commit, below the smallest range -> filtered out, else big blob
            const uint32_t np = ld_relaxed(cp + o);
            uint32_t cat = 0;
            if (!size_ok(np, c)) {
                double mn = c.ranges[0];
                for (int i = 1; i < c.n_ranges; ++i) mn = c.ranges[2 * i] < mn ? c.ranges[2 * i] : mn;
                cat = ((double)((float)np * c.sqcm) < mn) ? TREXHIP_BLOB_BELOW_RANGE : TREXHIP_BLOB_BIG;
            }
            B.flags = cat;
        }
        blobs[bb + k] = B;
        blob_frame[bb + k] = (uint32_t)f;
    }
    __syncthreads();
    if (run_parent)                      // re-threshold pass: a sub-blob inherits the detect blob of its root run
        for (uint32_t r = tid; r < n; r += 256)
            if (lab[r] == r) { const int32_t k = bmap[ord[r]]; if (k >= 0) blobs[bb + k].parent = run_parent[fo + r]; }
    const bool lds_cursor = kept <= (uint32_t)CURSOR_LDS;
    if (lds_cursor)
        for (uint32_t o = tid; o < nraw; o += 256) {
            const int32_t k = bmap[o];
            if (k >= 0) s_cursor[k] = cur[o];
        }
    __threadfence_block();
    __syncthreads();
    // D: stable scatter -- one wave walks the runs in raster order, 64 at a time; lanes of the same
    // blob are grouped by ballot so every run gets (cursor of its blob) + (rank inside the group)
    if (tid < 64) {
        const uint32_t lane = tid;
        trexhip_run* outr = out_runs + rb;
        for (uint32_t b0 = 0; b0 < n; b0 += 64) {
            const uint32_t r = b0 + lane;
            int32_t k = -1; uint32_t o = 0;
            trexhip_run q = {};
            if (r < n) { o = ord[lab[r]]; k = bmap[o]; q = rr[r]; }
            const bool active = k >= 0;
            uint64_t remaining = __ballot(active);
            uint32_t rank = 0, gsize = 0, leader = lane;
            while (remaining) {
                const int l = __ffsll((unsigned long long)remaining) - 1;
                const int32_t lk = __shfl(k, l);
                const uint64_t grp = __ballot(active && k == lk);
                if (active && k == lk) {
                    rank = __popcll(grp & ((1ull << lane) - 1ull));
                    gsize = __popcll(grp);
                    leader = (uint32_t)l;
                }
                remaining &= ~grp;
            }
            uint32_t pos = 0;
            if (active && leader == lane) {
                if (lds_cursor) { pos = s_cursor[k]; s_cursor[k] = pos + gsize; }
                else            { pos = cur[o];      cur[o] = pos + gsize; }
            }
            pos = __shfl(pos, leader);
            if (active) outr[pos + rank] = q;
        }
    }
    if (tid == 0) {
        fi.n_blobs = kept; fi.n_runs = kruns; fi.n_pixels = kpx;
        fi.blob_begin = bb; fi.run_begin = rb; fi.pix_begin = pb;
        info[f] = fi;
    }
}

// ---------------------------------------------------------------------------------------------
// k_ccl_lds: the whole run-level pipeline of one frame inside one workgroup's LDS (1024 threads):
// row scan -> runs to raster order -> link with the row above (union-find, LDS atomics) -> flatten ->
// blob numbering / counts -> size filter + offsets -> pooled reservation -> stable grouping of the runs
// by blob.  Frames with more than CCL_NMAX runs are marked pending (info.reserved[0] = 1) and finished by the global-memory chain above.
// LDS (20 bytes per run: a 4096 x 4096 frame of 256 individuals carries ~7.7 k runs and fits): s_run, s_par (union-find parent -> root ->
// raw blob ordinal), s_y, s_seg (u16: the runs grouped by blob) and two general arrays that change their role phase by phase:
//   s_a  row bases (P1..P3) | runs per raw blob (P5) | runs | first slot of the blob's segment << 16 (P6..) | sort keys (large blobs)
//   s_b  flatten scratch (P4) | ordinal of a root run (P5a) | pixels per raw blob (P5b) | kept index or ~0 (P6, P7) | segment cursor or ~0
// ---------------------------------------------------------------------------------------------
// Instantiated BY CAPACITY (round 6): a workgroup that holds 8160 lines needs all 160 KB of a CU and 16 waves whatever the frame carries, so one
// frame occupies one CU.  The smaller instances leave room for several frames per CU (and for the pixel pass of another group beside them):
//   L  1024 threads, 8160 lines, 160 KB  (a 4096 x 4096 frame of 256 individuals: ~7.7 k lines)
//   M   512 threads, 3840 lines,  76 KB  (two per CU; a 2048 x 2048 frame of 100 individuals: ~2.5 k lines)
//   S   256 threads, 2000 lines,  40 KB  (four per CU; 1280 x 720 with 32 blobs: ~0.7 k lines; TRex's one-frame calls)
// NT threads, NMAX lines, SA words of s_a (row bases of up to SA - 1 rows; power of two >= NMAX for the bitonic sort).  A frame with more lines
// than the instance holds is left for the L instance (info.reserved[0] = 3: launch_segment queues it behind every smaller one, it returns at
// once for frames that are done), a frame beyond L for the global-memory chain (reserved[0] = 1) as before.
static constexpr int CCL_NMAX = 8160;               // runs per frame of the largest instance (16-bit fields: < 65536; the arrays + 256 B must fit 160 KB)
static constexpr int CCL_SORT = 8192;
template <int NMAX, int SA> struct CclLds { static constexpr int BYTES = NMAX * (4 + 4 + 4 + 2 + 2) + SA * 4 + 256; };
static constexpr int CCL_LDS_BYTES = CclLds<CCL_NMAX, CCL_SORT>::BYTES;
static_assert(CCL_LDS_BYTES <= 160 * 1024, "k_ccl_lds: LDS");
static constexpr int CCL_M_NMAX = 3840, CCL_M_SA = 4096, CCL_S_NMAX = 2000, CCL_S_SA = 2048;
static_assert(2 * CclLds<CCL_M_NMAX, CCL_M_SA>::BYTES <= 160 * 1024 && 4 * CclLds<CCL_S_NMAX, CCL_S_SA>::BYTES <= 160 * 1024, "k_ccl_lds: LDS of the small instances");

// lane ^ D exchange for the sorting networks of k_ccl_lds: DPP quad permutes for D = 1, 2 (no LDS crossbar trip, no address register), ds_swizzle with an
// immediate pattern for 4 .. 16, a permute only across the halves
template <int D> __device__ __forceinline__ uint32_t ccl_xchg(uint32_t v) {
    if constexpr (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1, 0, 3, 2]
    else if constexpr (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2, 3, 0, 1]
    else if constexpr (D < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (D << 10) | 0x1f);
    else return (uint32_t)__shfl_xor((int)v, D);
}
__device__ __forceinline__ uint32_t lds_find(volatile uint32_t* par, uint32_t a) {
    uint32_t p = par[a];
    while (p != a) {
        const uint32_t g = par[p];
        if (g != p) par[a] = g;            // path halving: parents only ever move towards the root, so this is race-safe
        a = p; p = g;
    }
    return a;
}
__device__ __forceinline__ void lds_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        a = lds_find(par, a); b = lds_find(par, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(par + b, a);
        if (old == b) return;
        b = old;
    }
}

template <bool COLOUR, int LPB>
__device__ __forceinline__ void gather_blobs(const SegCfg& c, const int only_pending, const uint8_t* __restrict__ frames,
                                             const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ blob_frame,
                                             trexhip_blob* __restrict__ blobs, const trexhip_run* __restrict__ runs,
                                             uint8_t* __restrict__ pixels, const uint32_t f0, const uint32_t f1,
                                             const uint8_t* __restrict__ color, const int color_ch, const int enc_,
                                             const uint32_t bw0, const uint32_t bw_step, const uint32_t total,
                                             const int own_frame, const uint32_t own_run_begin, const uint32_t own_pix_begin);

// ---------------------------------------------------------------------------------------------
// k_ccl_band (round 6): SEVERAL workgroups per frame.  k_ccl_lds gives a frame one workgroup whatever the launch holds, so a launch of few
// frames (C5: 64 frames of 6.4 k lines on 256 CUs; TRex's default call: ONE frame) leaves most of the chip idle while each workgroup walks
// its whole frame through the two longest phases, runs into LDS and link.  Here a frame's rows are cut into bands: workgroup (frame, band)
// loads the band's lines, links them among themselves (never across the band's first row) and writes lines and band-local roots -- as
// frame-wide raster indices -- to the run-level tables.  k_ccl_lds then takes a frame in `banded` mode: instead of its phases 2 and 3 it
// loads lines and parents, links the first row of every band with the row above it (the seams: a handful of lines) and goes on with its
// flatten.  Roots are the smallest raster index of a component either way (lds_union), so every table comes out byte for byte the same.
// A band with more lines than fit (CCLB_NMAX) raises the frame's word in `band_fail`: k_ccl_lds then labels that frame alone, as before.
// ---------------------------------------------------------------------------------------------
static constexpr int CCLB_NT = 1024, CCLB_NMAX = 4096, CCLB_ROWS = 2048, CCLB_SW = CCLB_ROWS / CCLB_NT;
static constexpr int CCLB_LDS_BYTES = (CCLB_ROWS + 1) * 4 + CCLB_NMAX * (4 + 4 + 2) + 64 * 4 + 16;
__global__ __launch_bounds__(CCLB_NT) void k_ccl_band(const SegCfg c, const uint32_t* __restrict__ frame_ctr, const uint32_t* __restrict__ row_cnt,
                                                      const uint32_t* __restrict__ row_off, uint32_t* __restrict__ row_base, const uint32_t* __restrict__ tmp_runs,
                                                      trexhip_run* __restrict__ raster, uint32_t* __restrict__ parent, uint32_t* __restrict__ band_fail,
                                                      const int n_bands, const int band_rows, const int f0) {
    constexpr int NT = CCLB_NT;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* s_key = smem;                           // [CCLB_ROWS + 1] band-local index of every row's first line
    uint32_t* s_run = s_key + CCLB_ROWS + 1;
    uint32_t* s_par = s_run + CCLB_NMAX;
    uint32_t* s_misc = s_par + CCLB_NMAX;             // [64]
    uint16_t* s_y = reinterpret_cast<uint16_t*>(s_misc + 64);
    const int f = (int)blockIdx.x / n_bands + f0, band = (int)blockIdx.x % n_bands, tid = threadIdx.x;
    const int H = c.H, y0 = band * band_rows, y1 = min(H, y0 + band_rows);
    if (y0 >= H) return;
    const uint32_t* cnt = row_cnt + (size_t)f * H;
    const uint32_t* off = row_off + (size_t)f * H;
    const uint32_t* tmp = tmp_runs + (size_t)f * c.T;
    const size_t fo = (size_t)f * c.R;
    // every load of the prologue goes out first: the band's own rows (count, run offset -- one or two per thread, kept in registers like the first
    // sweeps of k_ccl_lds), the counts of all rows for the lines in front of the band and in the frame, the frame's overflow counter; then the first
    // run of each own row.  One round trip + one for the first runs, where a loop per quantity paid one each
    uint32_t pk[CCLB_SW], po[CCLB_SW], pt[CCLB_SW], pb[CCLB_SW];
#pragma unroll
    for (int j = 0; j < CCLB_SW; ++j) {
        const int y = y0 + j * NT + tid;
        pk[j] = y < y1 ? cnt[y] : 0u;
        po[j] = y < y1 ? off[y] : 0u;
        pt[j] = 0u; pb[j] = 0u;
    }
    const uint32_t fctr = frame_ctr[f * CTR_STRIDE];
    uint32_t sb = 0, sn = 0;
    for (int y = tid; y < H; y += NT) { const uint32_t v = cnt[y]; sn += v; if (y < y0) sb += v; }
#pragma unroll
    for (int j = 0; j < CCLB_SW; ++j) if (pk[j] && po[j] < (uint32_t)c.T) pt[j] = tmp[po[j]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sb += (uint32_t)__shfl_xor((int)sb, d); sn += (uint32_t)__shfl_xor((int)sn, d); }
    if ((tid & 63) == 0) { s_misc[32 + (tid >> 6)] = sb; s_misc[48 + (tid >> 6)] = sn; }
    __syncthreads();
    uint32_t base = 0, n = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { base += s_misc[32 + w]; n += s_misc[48 + w]; }
    // what k_ccl_lds refuses (overflow of the run area, more lines than LDS holds) it refuses by itself: nothing to prepare, and no table to write into.
    // Bit 1 of the frame's word tells it that nothing was prepared (it then scans the rows itself and finds the same)
    if (n > (uint32_t)c.R || fctr > (uint32_t)c.R || n > (uint32_t)CCL_NMAX) { if (tid == 0) atomicOr(band_fail + f, 2u); return; }
    uint32_t* rb = row_base + (size_t)f * (H + 1);
    uint32_t nb = 0;
#pragma unroll
    for (int j = 0; j < CCLB_SW; ++j) {
        if (y0 + j * NT >= y1) break;
        const int y = y0 + j * NT + tid;
        uint32_t total;
        const uint32_t ex = block_excl_scan(pk[j], s_misc, total);
        pb[j] = nb + ex;
        if (y < y1) { s_key[y - y0] = nb + ex; rb[y] = base + nb + ex; }      // the frame-wide raster index of the row: k_ccl_lds takes it from here instead of scanning again
        nb += total;
    }
    if (tid == 0) { s_key[y1 - y0] = nb; if (y1 == H) rb[H] = n; }
    if (nb > (uint32_t)CCLB_NMAX) { if (tid == 0) atomicOr(band_fail + f, 1u); return; }
    // the band's lines into LDS
#pragma unroll
    for (int j = 0; j < CCLB_SW; ++j) {
        const uint32_t k = pk[j], b = pb[j], o = po[j];
        const int y = y0 + j * NT + tid;
        if (!k) continue;
        s_run[b] = pt[j]; s_y[b] = (uint16_t)y; s_par[b] = b;
        for (uint32_t i = 1; i < k; ++i) { s_run[b + i] = tmp[o + i]; s_y[b + i] = (uint16_t)y; s_par[b + i] = b + i; }
    }
    __syncthreads();
    const int slack = c.slack;
    for (uint32_t r = tid; r < nb; r += NT) {
        const int y = s_y[r];
        if (y == y0) continue;                             // the band's first row: its row above belongs to another workgroup (the seam)
        uint32_t lo = s_key[y - 1 - y0];
        const uint32_t je = s_key[y - y0];
        if (lo >= je) continue;
        const uint32_t cur = s_run[r];
        const int c0 = cur & 0xffffu, c1 = cur >> 16;
        uint32_t hi = je;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int)(s_run[mid] >> 16) + slack >= c0) hi = mid; else lo = mid + 1; }
        for (uint32_t j = lo; j < je; ++j) {
            const uint32_t prv = s_run[j];
            if ((int)(prv & 0xffffu) > c1 + slack) break;
            lds_union(s_par, j, r);
        }
    }
    __syncthreads();
    for (uint32_t r = tid; r < nb; r += NT) {
        const uint32_t root = lds_find(s_par, r);          // (path halving beside it only ever moves parents towards their root)
        parent[fo + base + r] = base + root;
        trexhip_run q; q.x0 = (uint16_t)(s_run[r] & 0xffffu); q.x1 = (uint16_t)(s_run[r] >> 16); q.y = s_y[r]; q.pad = 0;
        raster[fo + base + r] = q;
    }
}

template <int NT, int NMAX, int SA>
__global__ __launch_bounds__(NT) void k_ccl_lds(const SegCfg c, uint32_t* __restrict__ frame_ctr,
                                                  const uint32_t* __restrict__ row_cnt, const uint32_t* __restrict__ row_off,
                                                  uint32_t* __restrict__ row_base, const uint32_t* __restrict__ tmp_runs,
                                                  trexhip_run* __restrict__ raster, uint32_t* __restrict__ parent,
                                                  uint32_t* __restrict__ root_ord, uint32_t* __restrict__ cur_run,
                                                  uint32_t* __restrict__ pix_begin, int32_t* __restrict__ blob_map,
                                                  uint32_t* __restrict__ totals, trexhip_frame_info* __restrict__ info,
                                                  trexhip_blob* __restrict__ blobs, uint32_t* __restrict__ blob_frame,
                                                  trexhip_run* __restrict__ out_runs, const int dbg_stop,
                                                  unsigned long long* __restrict__ dbg, const int f0,
                                                  const uint8_t* __restrict__ own_frames /* gray frames: the workgroup also gathers its frame's blobs; else null */,
                                                  uint8_t* __restrict__ own_pixels,
                                                  const int retry_only /* 1: only the frames a smaller instance left for this one */,
                                                  uint32_t* __restrict__ hint /* pinned host words: [0] a frame had more lines than S holds, [1] than M holds */,
                                                  const int band_rows /* > 0: k_ccl_band has prepared the frames in bands of this many rows */,
                                                  uint32_t* __restrict__ band_fail) {
    static_assert(NT % 64 == 0 && NT <= 1024 && (SA & (SA - 1)) == 0 && SA >= NMAX && SA % NT == 0 && NMAX <= CCL_NMAX, "k_ccl_lds: geometry");
    constexpr int NW = NT / 64;                   // waves
    constexpr int NSW = NT >= 1024 ? 2 : 4;       // row sweeps whose count / offset / raster index / first run stay in registers
#ifdef TREXHIP_DEV_KNOBS
#define CCL_STAMP(i) do { if (dbg_stop == -1 && blockIdx.x == 0 && threadIdx.x == 0) dbg[i] = __builtin_readcyclecounter(); } while (0)
#define CCL_STOP(n) do { if (dbg_stop == (n)) return; } while (0)
#else
#define CCL_STAMP(i) do { } while (0)
#define CCL_STOP(n) do { } while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* s_a = smem;                         // [SA] (see above)
    uint32_t* s_run = s_a + SA;             // x0 | x1 << 16, raster order
    uint32_t* s_par = s_run + NMAX;           // union-find parent -> root run -> raw blob ordinal
    uint32_t* s_b = s_par + NMAX;             // (see above)
    uint32_t* s_misc = s_b + NMAX;            // [64] scan scratch / broadcasts
    uint16_t* s_y = reinterpret_cast<uint16_t*>(s_misc + 64);
    uint16_t* s_seg = s_y + NMAX;             // raster indices of the kept runs, one segment per blob
    uint32_t* s_key = s_a;                        // P1..P3: raster index of every row's first run
    const int f = blockIdx.x + f0, tid = threadIdx.x;
    const int H = c.H;
    const uint32_t* cnt = row_cnt + (size_t)f * H;
    const uint32_t* off = row_off + (size_t)f * H;
    uint32_t* rb = row_base + (size_t)f * (H + 1);
    const size_t fo = (size_t)f * c.R;

    CCL_STAMP(0);
    if (retry_only && info[f].reserved[0] != 3u) return;      // (block-uniform: every thread reads the same word) the frame was finished by a smaller instance
    // the frame's overflow-area counter: thread 0 reads it and hands it back zeroed for the next pass (every path: a frame left pending for the
    // global-memory chain did not overflow, and that chain's own check then reads 0)
    if (tid == 0) {
        s_misc[60] = frame_ctr[f * CTR_STRIDE]; frame_ctr[f * CTR_STRIDE] = 0u;
        // banded (k_ccl_band ran in front): the frame's word is 0 when every band workgroup prepared its rows -- lines, band-local roots AND the rows'
        // raster indices are in the run-level tables; else (a band held too many lines, or the frame is one this kernel refuses) the frame is labelled here
        if (band_rows > 0) { s_misc[61] = band_fail[f]; band_fail[f] = 0u; }
    }
    const bool rb_lds = H < SA;              // row_base also lives in LDS (s_key is idle until P4)
    bool banded = false;
    if (band_rows > 0) { __syncthreads(); banded = s_misc[61] == 0u; }
    // P1: raster index of every row
    // (the rows of the first two sweeps -- every row of a frame up to 2048 lines -- keep their count, run offset, raster index and FIRST run in
    // registers: the offset and the run are fetched while the scan's barriers pass, and P2 starts without a global round trip)
    const uint32_t* tmp = tmp_runs + (size_t)f * c.T;
    uint32_t n = 0;
    uint32_t pk[NSW], po[NSW], pbase[NSW], pt[NSW];
    if (banded) {
#pragma unroll
        for (int j = 0; j < NSW; ++j) { pk[j] = 0u; po[j] = 0u; pbase[j] = 0u; pt[j] = 0u; }
        n = rb[H];
        if (rb_lds) for (int y = tid; y < H; y += NT) s_key[y] = rb[y];
    } else {
    {
        // every sweep's loads first (one round trip for the counts and offsets, one for the first runs), then the scans
        // (a frame whose run area overflowed carries offsets past its T words -- only the rows kernels' WRITES are bounded; such a frame is
        // refused below, but this prefetch comes before that test: never read past the area)
#pragma unroll
        for (int j = 0; j < NSW; ++j) {
            const int y = j * NT + tid;
            pk[j] = y < H ? cnt[y] : 0u;
            po[j] = y < H ? off[y] : 0u;
            pbase[j] = 0u; pt[j] = 0u;
        }
#pragma unroll
        for (int j = 0; j < NSW; ++j) if (pk[j] && po[j] < (uint32_t)c.T) pt[j] = tmp[po[j]];
#pragma unroll
        for (int j = 0; j < NSW; ++j) {
            if (j * NT >= H) break;
            const int y = j * NT + tid;
            uint32_t total;
            const uint32_t ex = block_excl_scan(pk[j], s_misc, total);
            if (y < H) { rb[y] = n + ex; if (y < SA - 1) s_key[y] = n + ex; }
            pbase[j] = n + ex;
            n += total;
        }
    }
    for (int y0 = NSW * NT; y0 < H; y0 += NT) {
        const int y = y0 + tid;
        const uint32_t v = y < H ? cnt[y] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, s_misc, total);
        if (y < H) { rb[y] = n + ex; if (y < SA - 1) s_key[y] = n + ex; }
        n += total;
    }
    }
    trexhip_frame_info fi = {};
    fi.n_raw_runs = n;
    const bool overflow = n > (uint32_t)c.R || s_misc[60] > (uint32_t)c.R;       // (behind the barriers of the scans above)
    if (!overflow && hint && tid == 0) {           // (plain stores of a constant into the context's pinned words: the host reads them before its next launch)
        if (n > (uint32_t)CCL_S_NMAX) hint[0] = 1u;
        if (n > (uint32_t)CCL_M_NMAX) hint[1] = 1u;
    }
    if (overflow || n > (uint32_t)NMAX) {
        if (tid == 0) {
            rb[H] = n;
            if (overflow) fi.flags = TREXHIP_FRAME_OVERFLOW_RUNS;
            else fi.reserved[0] = n > (uint32_t)CCL_NMAX ? 1u : 3u;   // pending: too many runs for LDS altogether (global-memory chain) / for this instance (L follows)
            info[f] = fi;
        }
        return;
    }
    if (tid == 0) { rb[H] = n; if (rb_lds) s_key[H] = n; }
    __syncthreads();
    CCL_STOP(1);
    CCL_STAMP(1);
    const int slack = c.slack;
    if (banded) {
        // banded (k_ccl_band): lines and band-local roots come from the run-level tables, only the seams are left to link
        for (uint32_t r = tid; r < n; r += NT) {
            const uint2 q = *reinterpret_cast<const uint2*>(raster + fo + r);      // x0 | x1 << 16, y | pad << 16
            s_run[r] = q.x; s_y[r] = (uint16_t)(q.y & 0xffffu); s_par[r] = parent[fo + r];
        }
        __syncthreads();
        for (int yb = band_rows; yb < H; yb += band_rows) {
            uint32_t lo0 = rb_lds ? s_key[yb - 1] : rb[yb - 1];
            const uint32_t je = rb_lds ? s_key[yb] : rb[yb], re = rb_lds ? s_key[yb + 1] : rb[yb + 1];
            if (lo0 >= je) continue;
            for (uint32_t r = je + tid; r < re; r += NT) {
                const uint32_t cur = s_run[r];
                const int c0 = cur & 0xffffu, c1 = cur >> 16;
                uint32_t lo = lo0, hi = je;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int)(s_run[mid] >> 16) + slack >= c0) hi = mid; else lo = mid + 1; }
                for (uint32_t j = lo; j < je; ++j) {
                    const uint32_t prv = s_run[j];
                    if ((int)(prv & 0xffffu) > c1 + slack) break;
                    lds_union(s_par, j, r);
                }
            }
        }
    } else {
    // P2: runs into LDS in raster order
#pragma unroll
    for (int j = 0; j < NSW; ++j) {
        const uint32_t k = pk[j], b = pbase[j], o = po[j];
        const int y = j * NT + tid;
        if (!k) continue;
        s_run[b] = pt[j]; s_y[b] = (uint16_t)y; s_par[b] = b;
        for (uint32_t i = 1; i < k; ++i) { s_run[b + i] = tmp[o + i]; s_y[b + i] = (uint16_t)y; s_par[b + i] = b + i; }
    }
    for (int y = NSW * NT + tid; y < H; y += NT) {
        const uint32_t b = rb_lds ? s_key[y] : rb[y];
        const uint32_t k = (rb_lds ? s_key[y + 1] : rb[y + 1]) - b;
        if (!k) continue;
        const uint32_t o = off[y];
        for (uint32_t i = 0; i < k; ++i) { s_run[b + i] = tmp[o + i]; s_y[b + i] = (uint16_t)y; s_par[b + i] = b + i; }
    }
    __syncthreads();
    CCL_STOP(2);
    CCL_STAMP(2);
    // P3: link every run with the touching runs of the row above (thread per run, binary search for the first candidate)
    for (uint32_t r = tid; r < n; r += NT) {
        const int y = s_y[r];
        if (y == 0) continue;
        uint32_t lo = rb_lds ? s_key[y - 1] : rb[y - 1];
        const uint32_t je = rb_lds ? s_key[y] : rb[y];
        if (lo >= je) continue;
        const uint32_t cur = s_run[r];
        const int c0 = cur & 0xffffu, c1 = cur >> 16;
        uint32_t hi = je;                                  // first run of the row above with x1 + slack >= c0
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((int)(s_run[mid] >> 16) + slack >= c0) hi = mid; else lo = mid + 1; }
        for (uint32_t j = lo; j < je; ++j) {
            const uint32_t prv = s_run[j];
            if ((int)(prv & 0xffffu) > c1 + slack) break;
            lds_union(s_par, j, r);
        }
    }
    }
    __syncthreads();
    CCL_STOP(3);
    CCL_STAMP(3);
    // P4: flatten
    for (uint32_t r = tid; r < n; r += NT) { const uint32_t root = lds_find(s_par, r); s_b[r] = root; }
    __syncthreads();
    for (uint32_t r = tid; r < n; r += NT) s_par[r] = s_b[r];
    __syncthreads();
    CCL_STOP(4);
    CCL_STAMP(4);
    // P5: blob ordinals (raster order of the root run) -> s_b at the roots; the run-level state of the re-threshold pass (root label, ordinal
    // of a root) goes to global memory here, then every run's label becomes its blob's ordinal and s_a / s_b count runs / pixels per blob
    uint32_t nraw = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += NT) {
        const uint32_t r = b0 + tid;
        const uint32_t flag = (r < n && s_par[r] == r) ? 1u : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan(flag, s_misc, total);
        if (flag) s_b[r] = nraw + ex;
        nraw += total;
    }
    __syncthreads();
    for (uint32_t r = tid; r < n; r += NT) {
        const uint32_t lab = s_par[r];
        const uint32_t o = s_b[lab];
        parent[fo + r] = lab;
        if (lab == r) root_ord[fo + r] = o;
        s_par[r] = o;                                      // (a thread reads and writes its own entry only; s_b stays as it is until the barrier)
    }
    __syncthreads();
    for (uint32_t o = tid; o < nraw; o += NT) { s_a[o] = 0; s_b[o] = 0; }
    __syncthreads();
    for (uint32_t r = tid; r < n; r += NT) {
        const uint32_t o = s_par[r];
        const uint32_t q = s_run[r];
        atomicAdd(s_a + o, 1u);
        atomicAdd(s_b + o, (q >> 16) - (q & 0xffffu) + 1u);
    }
    __syncthreads();
    CCL_STOP(5);
    CCL_STAMP(5);
    // P6: size filter, offsets of the kept blobs.  s_a[o] = runs | first slot of the segment << 16 (both < 8192), s_b[o] = kept index or ~0
    uint32_t kept = 0, kruns = 0, kpx = 0;
    uint32_t* pbg = pix_begin + fo;
    int32_t* bmap = blob_map + fo;
    for (uint32_t b0 = 0; b0 < nraw; b0 += NT) {
        const uint32_t o = b0 + tid;
        uint32_t nr = 0, np = 0, keep = 0;
        if (o < nraw) { nr = s_a[o]; np = s_b[o]; keep = (size_ok(np, c) && nr < 65535u) ? 1u : 0u; }
        uint32_t ex3[3], tot3[3];
        block_excl_scan3(keep, keep ? nr : 0u, keep ? np : 0u, s_misc, ex3, tot3);
        const uint32_t e0 = ex3[0], e1 = ex3[1], e2 = ex3[2], t0 = tot3[0], t1 = tot3[1], t2 = tot3[2];
        if (o < nraw) {
            bmap[o] = keep ? (int32_t)(kept + e0) : -1;
            s_b[o] = keep ? (kept + e0) : 0xffffffffu;
            if (keep) { s_a[o] = nr | ((kruns + e1) << 16); pbg[o] = kpx + e2; }
        }
        kept += t0; kruns += t1; kpx += t2;
    }
    __syncthreads();
    if (kept > c.cap_blobs || kpx > c.cap_pixels) {        // beyond this frame's own capacities: fails alone, reserves nothing (see k_blobs)
        fi.n_raw_blobs = nraw;
        if (tid == 0) { fi.flags |= TREXHIP_FRAME_OVERFLOW_OUTPUT; info[f] = fi; }
        return;
    }
    if (tid == 0) {
        // totals[0],[1] = blobs, runs reserved with ONE 64-bit atomic; totals[2] = pixels
        const unsigned long long br = atomicAdd(reinterpret_cast<unsigned long long*>(totals),
                                                (unsigned long long)kept | ((unsigned long long)kruns << 32));
        const uint32_t pb = atomicAdd(totals + 2, kpx);
        const uint32_t bb = (uint32_t)br, rbeg = (uint32_t)(br >> 32);
        const bool over = bb + kept > c.pool_blobs || rbeg + kruns > c.pool_runs || pb + kpx > c.pool_pixels;
        s_misc[32] = bb; s_misc[33] = rbeg; s_misc[34] = pb; s_misc[35] = over ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t bb = s_misc[32], rbeg = s_misc[33], pb = s_misc[34];
    fi.n_raw_blobs = nraw;
    if (s_misc[35]) {
        for (uint32_t k = tid; k < kept; k += NT)
            if (bb + k < c.pool_blobs) blob_frame[bb + k] = 0xffffffffu;
        if (tid == 0) { fi.flags |= TREXHIP_FRAME_OVERFLOW_OUTPUT; info[f] = fi; }
        return;
    }
    CCL_STOP(6);
    CCL_STAMP(6);
    // P7: blob records, raster-order runs for the re-threshold pass, largest kept blob (decides the grouping strategy, block-uniform)
    uint32_t mx = 0;
    for (uint32_t o = tid; o < nraw; o += NT) {
        const uint32_t k = s_b[o];
        if (k == 0xffffffffu) continue;
        const uint32_t a = s_a[o];
        trexhip_blob B = {};
        B.run_begin = (a >> 16) + rbeg;   // POOLED offsets until k_gather makes them frame-relative
        B.n_runs = a & 0xffffu;
        B.pix_begin = pbg[o] + pb;        // (written by this thread in P6)
        B.parent = 0xffffffffu;
        blobs[bb + k] = B;
        blob_frame[bb + k] = (uint32_t)f;
        mx = max(mx, a & 0xffffu);
    }
    for (uint32_t r = tid; r < n; r += NT) {
        trexhip_run q; q.x0 = (uint16_t)(s_run[r] & 0xffffu); q.x1 = (uint16_t)(s_run[r] >> 16); q.y = s_y[r]; q.pad = 0;
        raster[fo + r] = q;
    }
    CCL_STOP(7);
    CCL_STAMP(7);
    mx = wmax32(mx);
    __syncthreads();
    if ((tid & 63) == 0) s_misc[40 + (tid >> 6)] = mx;
    __syncthreads();
    mx = 0;
    for (int w = 0; w < NW; ++w) mx = max(mx, s_misc[40 + w]);
    trexhip_run* outr = out_runs + rbeg;
    if (mx <= 512u) {
        // (1) scatter raster indices into each blob's segment in arbitrary order (LDS atomics on a per-blob cursor: s_b becomes 0 for a
        //     kept blob, stays ~0 for a dropped one), (2) every segment is sorted by raster index and written straight out
        for (uint32_t o = tid; o < nraw; o += NT) if (s_b[o] != 0xffffffffu) s_b[o] = 0u;
        __syncthreads();
        for (uint32_t r = tid; r < n; r += NT) {
            const uint32_t o = s_par[r];
            if (s_b[o] == 0xffffffffu) continue;
            const uint32_t slot = atomicAdd(s_b + o, 1u);
            s_seg[(s_a[o] >> 16) + slot] = (uint16_t)r;
        }
        __syncthreads();
        CCL_STAMP(9);
        // (2) an in-register bitonic network per blob -- two blobs per wave (32 lanes each) when both have at most 32 lines, one per wave
        //     up to 64 lines, rank-by-counting through LDS beyond that.  The waves walk the RAW ordinals; a dropped blob counts 0 lines
        const uint32_t lane = tid & 63, wave = tid >> 6;
#define CCL_SORT_STEP(kk_, jj_) { const uint32_t o_ = ccl_xchg<jj_>(v); const bool lo_ = ((lane & (jj_)) == 0) == ((lane & (kk_)) == 0); v = lo_ ? min(v, o_) : max(v, o_); }
#define CCL_SORT32(v)                                                                                                   \
        CCL_SORT_STEP(2, 1) CCL_SORT_STEP(4, 2) CCL_SORT_STEP(4, 1) CCL_SORT_STEP(8, 4) CCL_SORT_STEP(8, 2) CCL_SORT_STEP(8, 1)    \
        CCL_SORT_STEP(16, 8) CCL_SORT_STEP(16, 4) CCL_SORT_STEP(16, 2) CCL_SORT_STEP(16, 1)                                        \
        CCL_SORT_STEP(32, 16) CCL_SORT_STEP(32, 8) CCL_SORT_STEP(32, 4) CCL_SORT_STEP(32, 2) CCL_SORT_STEP(32, 1)
        auto emit = [&](uint32_t beg, uint32_t e, uint32_t cntk, uint32_t r) {
            if (e < cntk) {
                trexhip_run q; q.x0 = (uint16_t)(s_run[r] & 0xffffu); q.x1 = (uint16_t)(s_run[r] >> 16); q.y = s_y[r]; q.pad = 0;
                outr[beg + e] = q;
            }
        };
        for (uint32_t k0 = wave * 2; k0 < nraw; k0 += 2 * NW) {
            const uint32_t aA = s_b[k0] != 0xffffffffu ? s_a[k0] : 0u;
            const uint32_t aB = (k0 + 1 < nraw && s_b[k0 + 1] != 0xffffffffu) ? s_a[k0 + 1] : 0u;
            const uint32_t cA = aA & 0xffffu, cB = aB & 0xffffu;
            if (cA <= 32u && cB <= 32u) {
                const uint32_t e = lane & 31u;
                const uint32_t beg = (lane < 32 ? aA : aB) >> 16, cntk = lane < 32 ? cA : cB;
                uint32_t v = e < cntk ? (uint32_t)s_seg[beg + e] : 0xffffffffu;
                // lanes 32..63 sort ascending as well: the direction bit of the last stage (lane & 32) is flipped for them
                CCL_SORT_STEP(2, 1) CCL_SORT_STEP(4, 2) CCL_SORT_STEP(4, 1) CCL_SORT_STEP(8, 4) CCL_SORT_STEP(8, 2) CCL_SORT_STEP(8, 1)
                CCL_SORT_STEP(16, 8) CCL_SORT_STEP(16, 4) CCL_SORT_STEP(16, 2) CCL_SORT_STEP(16, 1)
                { const uint32_t o_ = ccl_xchg<16>(v); v = (lane & 16) == 0 ? min(v, o_) : max(v, o_); }
                { const uint32_t o_ = ccl_xchg<8>(v);  v = (lane & 8) == 0 ? min(v, o_) : max(v, o_); }
                { const uint32_t o_ = ccl_xchg<4>(v);  v = (lane & 4) == 0 ? min(v, o_) : max(v, o_); }
                { const uint32_t o_ = ccl_xchg<2>(v);  v = (lane & 2) == 0 ? min(v, o_) : max(v, o_); }
                { const uint32_t o_ = ccl_xchg<1>(v);  v = (lane & 1) == 0 ? min(v, o_) : max(v, o_); }
                emit(beg, e, cntk, v);
            } else {
                for (uint32_t k = k0; k < k0 + 2 && k < nraw; ++k) {
                    const uint32_t a = k == k0 ? aA : aB;
                    const uint32_t beg = a >> 16, cntk = a & 0xffffu;
                    if (cntk == 0u) continue;
                    if (cntk <= 64u) {
                        uint32_t v = lane < cntk ? (uint32_t)s_seg[beg + lane] : 0xffffffffu;
                        CCL_SORT32(v)
                        CCL_SORT_STEP(64, 32) CCL_SORT_STEP(64, 16) CCL_SORT_STEP(64, 8) CCL_SORT_STEP(64, 4) CCL_SORT_STEP(64, 2) CCL_SORT_STEP(64, 1)
                        emit(beg, lane, cntk, v);
                    } else {
                        for (uint32_t e0 = 0; e0 < cntk; e0 += 64) {
                            const uint32_t e = e0 + lane;
                            const uint32_t mine = e < cntk ? (uint32_t)s_seg[beg + e] : 0xffffffffu;
                            uint32_t rank = 0;
                            for (uint32_t t = 0; t < cntk; ++t) rank += (uint32_t)s_seg[beg + t] < mine ? 1u : 0u;
                            if (e < cntk) emit(beg, rank, cntk, mine);
                        }
                    }
                }
            }
        }
#undef CCL_SORT32
#undef CCL_SORT_STEP
    } else {
        // a blob of more than 512 lines: one bitonic sort of (kept blob index << 13 | raster index) over the whole frame, keys in s_a
        uint32_t sn = 64;
        while (sn < n) sn <<= 1;
        uint32_t keys[SA / NT];
#pragma unroll
        for (int u = 0; u < SA / NT; ++u) {
            const uint32_t r = tid + u * NT;
            uint32_t key = 0xffffffffu;
            if (r < n) { const uint32_t k = s_b[s_par[r]]; if (k != 0xffffffffu) key = (k << 13) | r; }
            keys[u] = key;
        }
        __syncthreads();                                   // s_a's blob fields are not read any more
#pragma unroll
        for (int u = 0; u < SA / NT; ++u) { const uint32_t r = tid + u * NT; if (r < sn) s_a[r] = keys[u]; }
        __syncthreads();
        for (uint32_t k = 2; k <= sn; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = tid; i < sn; i += NT) {
                    const uint32_t x = i ^ j;
                    if (x > i) {
                        const uint32_t a = s_a[i], b2 = s_a[x];
                        const bool up = (i & k) == 0;
                        if ((a > b2) == up) { s_a[i] = b2; s_a[x] = a; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t i = tid; i < kruns; i += NT) {
            const uint32_t r = s_a[i] & 8191u;
            trexhip_run q; q.x0 = (uint16_t)(s_run[r] & 0xffffu); q.x1 = (uint16_t)(s_run[r] >> 16); q.y = s_y[r]; q.pad = 0;
            outr[i] = q;
        }
    }
    CCL_STAMP(8);
    if (tid == 0) {
        fi.n_blobs = kept; fi.n_runs = kruns; fi.n_pixels = kpx;
        fi.blob_begin = bb; fi.run_begin = rbeg; fi.pix_begin = pb;
        info[f] = fi;
    }
    if (own_frames) {
        // gray pixel arrays: the frame's blobs are gathered right here (pixels, integer moment sums, bounding box, bid) by the 16 waves of this
        // workgroup, two blobs per wave and step -- the records and lines written above are this workgroup's own (visible behind the barrier),
        // the frame and its bases are known: no second launch, no look-ups.  The records and lines were written to GLOBAL memory by other waves of
        // this workgroup: a workgroup-scope fence in front of the barrier orders those stores before the loads below (the plain
        // barrier alone relied on the CU's L1 and on the compiler not moving the loads)
        __threadfence_block();
        __syncthreads();
        // four blobs per wave (16 lanes each): the gather is a chain of dependent loads (record -> lines -> pixels) per blob and step, and a frame of
        // 100 individuals takes two steps of 64 blobs instead of four of 32 (round 6)
        gather_blobs<false, 16>(c, 0, own_frames, info, blob_frame, blobs, out_runs, own_pixels, 0u, 0xffffffffu, nullptr, 0, 0,
                                bb + (uint32_t)(tid >> 6) * 4u, 4u * NW, bb + kept, f, rbeg, pb);
    }
#undef CCL_STAMP
#undef CCL_STOP
}

// ---------------------------------------------------------------------------------------------
// k_gather: one wave per kept blob
// ---------------------------------------------------------------------------------------------
// lane exchanges inside a 32-lane half without an address register (a __shfl_xor keeps its byte address live in a VGPR; the gather kernel had 18
// of them and a quarter of its occupancy gone): ds_swizzle in bit mode takes the pattern as an immediate.  D = 32 crosses the halves (permute).
template <int D> __device__ __forceinline__ uint32_t xchg_xor(uint32_t v) {
    if constexpr (D < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (D << 10) | 0x1f);
    else return (uint32_t)__shfl_xor((int)v, D);
}
// value of lane Q of the own 32-lane half (LPB = 32) / of the wave (LPB = 64)
template <int LPB, int Q> __device__ __forceinline__ uint32_t bcast_lane(uint32_t v) {
    if constexpr (LPB == 64) return (uint32_t)__builtin_amdgcn_readlane((int)v, Q);
    else if constexpr (LPB == 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (Q << 5));
    else return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (Q << 5) | 0x10);       // LPB = 16: lane bit 4 (which group of the 32-lane half) is kept
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
// Eight per-lane values -> eight wave totals with 10 exchanges instead of 48: three transposing butterfly steps halve the number of
// values a lane carries (a lane keeps one half and hands the other half to its partner), three plain steps finish.  Afterwards every
// lane holds the total of quantity q(lane) = 4 * bit0 + 2 * bit1 + bit2 of its lane index.
template <int D> __device__ __forceinline__ uint64_t shfl_xor64(uint64_t v) {
    const uint32_t lo = xchg_xor<D>((uint32_t)v), hi = xchg_xor<D>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <int LANES = 64>
__device__ __forceinline__ uint64_t wave_sum8x64(const uint64_t (&v)[8], uint32_t lane) {
    uint64_t w[4], u[2];
    const bool b0 = lane & 1u, b1 = lane & 2u, b2 = lane & 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (b0 ? v[4 + j] : v[j]) + shfl_xor64<1>(b0 ? v[j] : v[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) u[j] = (b1 ? w[2 + j] : w[j]) + shfl_xor64<2>(b1 ? w[j] : w[2 + j]);
    uint64_t t = (b2 ? u[1] : u[0]) + shfl_xor64<4>(b2 ? u[0] : u[1]);
    t += shfl_xor64<8>(t);
    if (LANES > 16) t += shfl_xor64<16>(t);
    if (LANES > 32) t += shfl_xor64<32>(t);
    return t;
}
template <int LANES = 64>
__device__ __forceinline__ uint32_t wave_min8x32(const uint32_t (&v)[8], uint32_t lane) {
    uint32_t w[4], u[2];
    const bool b0 = lane & 1u, b1 = lane & 2u, b2 = lane & 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = min(b0 ? v[4 + j] : v[j], xchg_xor<1>(b0 ? v[j] : v[4 + j]));
#pragma unroll
    for (int j = 0; j < 2; ++j) u[j] = min(b1 ? w[2 + j] : w[j], xchg_xor<2>(b1 ? w[j] : w[2 + j]));
    uint32_t t = min(b2 ? u[1] : u[0], xchg_xor<4>(b2 ? u[0] : u[1]));
    t = min(t, xchg_xor<8>(t));
    if (LANES > 16) t = min(t, xchg_xor<16>(t));
    if (LANES > 32) t = min(t, xchg_xor<32>(t));
    return t;
}
// lane that holds quantity q after the two reductions above
static_assert(offsetof(trexhip_blob, spy) - offsetof(trexhip_blob, m10) == 56, "k_gather writes the eight sums as an array");
__device__ __forceinline__ constexpr int lane_of_q(int q) { return ((q >> 2) & 1) | (((q >> 1) & 1) << 1) | ((q & 1) << 2); }
__device__ __forceinline__ uint32_t wave_min32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d));
    return v;
}

__device__ __forceinline__ uint32_t make_bid(uint32_t x0, uint32_t x1, uint32_t y, uint32_t n) {
    // pv::bid (commons): 13/13/6 bit hash of the first line; verified on the reference's golden CSVs
    uint32_t x = x0 + (x1 - x0 + 1) / 2;
    x = min(x, 8191u); y = min(y, 8191u);
    n = n < 1u ? 1u : min(n, 63u);
    return (x << 19) | (y << 6) | n;
}

// pixel of a colour encoding: rgb8 = the first three bytes in memory order (BGRA2BGR), r3g3b2 = the code of convert_to_r3g3b2
// (first channel 2 bits on top, then 3 + 3: layout pinned by Tests/test_pixels.cpp:629-795)
__device__ __forceinline__ void store_colour(uint8_t* px, uint32_t index, const uint8_t* src, int enc) {
    const uint32_t c0 = src[0], c1 = src[1], c2 = src[2];
    if (enc == 2) { px[3 * index] = (uint8_t)c0; px[3 * index + 1] = (uint8_t)c1; px[3 * index + 2] = (uint8_t)c2; }
    else px[index] = (uint8_t)(((c0 >> 6) << 6) | ((c1 >> 5) << 3) | (c2 >> 5));
}

// LPB lanes per blob: 64 (a wave per blob) or 32 (two blobs per wave: most blobs have fewer than 32 lines, and the kernel is a chain of
// dependent loads -- blob record -> lines -> pixels -- whose throughput is the number of blobs in flight)
// The waves that call this walk the pooled blobs bw0, bw0 + bw_step, ... below `total` (BPW blobs per wave and step).  own_frame >= 0: the caller
// is the labelling workgroup of that frame (k_ccl_lds gathers its own blobs: no second launch, no look-up of the frame or of its bases)
template <bool COLOUR, int LPB>
__device__ __forceinline__ void gather_blobs(const SegCfg& c, const int only_pending, const uint8_t* __restrict__ frames,
                                             const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ blob_frame,
                                             trexhip_blob* __restrict__ blobs, const trexhip_run* __restrict__ runs,
                                             uint8_t* __restrict__ pixels, const uint32_t f0, const uint32_t f1,
                                             const uint8_t* __restrict__ color, const int color_ch, const int enc_,
                                             const uint32_t bw0, const uint32_t bw_step, const uint32_t total,
                                             const int own_frame, const uint32_t own_run_begin, const uint32_t own_pix_begin) {
    const int enc = COLOUR ? enc_ : 0;                     // the gray instantiation carries no colour addressing at all
    constexpr uint32_t BPW = 64 / LPB;                     // blobs per wave
    const uint32_t lane = lane_id();
    const uint32_t sub = lane & (LPB - 1), part = lane / LPB, part_base = part * LPB;
    for (uint32_t bw = bw0; bw < total; bw += bw_step) {
        const uint32_t bi = bw + part;
        bool active = bi < total;
        uint32_t f = 0;
        // only the three fields the kernel needs travel in registers (the whole records cost 30 registers and a quarter of the occupancy)
        struct { uint32_t run_begin, n_runs, pix_begin; } B = {0u, 0u, 0u};
        if (active) {                                            // independent of the frame table: run_begin / pix_begin are pooled offsets here
            f = own_frame >= 0 ? (uint32_t)own_frame : blob_frame[bi];
            const uint2 rn = *reinterpret_cast<const uint2*>(&blobs[bi].run_begin);
            B.run_begin = rn.x; B.n_runs = rn.y; B.pix_begin = blobs[bi].pix_begin;
        }
        active = active && (own_frame >= 0 || (f < (uint32_t)c.B && f >= f0 && f < f1));   // else: hole left by a frame that overflowed the pool / another group's frame
        struct { uint32_t run_begin, pix_begin; } fi = {own_run_begin, own_pix_begin};
        if (active && own_frame < 0) {                           // off the critical path unless only_pending
            const uint2 rp = *reinterpret_cast<const uint2*>(&info[f].run_begin);
            fi.run_begin = rp.x; fi.pix_begin = rp.y;
            if (only_pending && info[f].reserved[0] != 2u) active = false;
        }
        const uint32_t n_runs = active ? B.n_runs : 0u;
        uint32_t nr_max = n_runs;                                 // the same in every lane of a blob: the wave's maximum from one lane per blob
        if (BPW > 1) nr_max = max((uint32_t)__builtin_amdgcn_readlane((int)n_runs, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_runs, 32));
        if (BPW > 2) nr_max = max(nr_max, max((uint32_t)__builtin_amdgcn_readlane((int)n_runs, 16), (uint32_t)__builtin_amdgcn_readlane((int)n_runs, 48)));
        static_assert(BPW == 1 || BPW == 2 || BPW == 4, "k_gather: one, two or four blobs per wave");
        const trexhip_run* rr = runs + B.run_begin;
        uint8_t* px = pixels + (size_t)B.pix_begin * (enc == 2 ? 3 : 1);
        const uint8_t* img = frames + (size_t)f * c.H * c.W;
        const uint8_t* cimg = enc ? color + (size_t)f * c.H * c.W * color_ch : nullptr;   // colour source of the r3g3b2 / rgb8 pixel arrays
        uint64_t m10 = 0, m01 = 0, m20 = 0, m11 = 0, m02 = 0, sp = 0, spx = 0, spy = 0;
        uint32_t x0 = 0xffff, x1 = 0, y0 = 0xffff, y1 = 0, pmin = 255, pmax = 0, po = 0;
        for (uint32_t b0 = 0; b0 < nr_max; b0 += LPB) {
            const uint32_t i = b0 + sub;
            trexhip_run q = {};
            uint32_t len = 0;
            if (i < n_runs) { q = rr[i]; len = (uint32_t)(q.x1 - q.x0 + 1); }
            const uint32_t incl = group_incl_scan<LPB>(len);
            uint32_t off = po + incl - len;
            po += bcast_lane<LPB, LPB - 1>(incl);
            if (len) {
                x0 = min(x0, (uint32_t)q.x0); x1 = max(x1, (uint32_t)q.x1);
                y0 = min(y0, (uint32_t)q.y);  y1 = max(y1, (uint32_t)q.y);
                const uint8_t* src = img + (size_t)q.y * c.W;
                const uint64_t y = q.y;
                uint64_t rp = 0;                                       // sum of grey values of this run
                if constexpr (!COLOUR) {
                    // gray pixel arrays: 8 pixels per step as ONE unaligned 8-byte load (the lanes of a wave read different rows: every load
                    // instruction costs the texture path a cycle per lane) fetched a step ahead, one 8-byte store, and no per-pixel
                    // arithmetic: sum p by v_sad_u8, sum p * x = x * sum p + sum k * p_k by v_dot4_u32_u8, min / max on packed 16-bit
                    // lanes; sum x and sum x^2 of the run in closed form below.  All sums are exact integers: the same values.
                    const uint32_t W_ = (uint32_t)c.W, xe = (uint32_t)q.x1;
                    auto load8 = [&](uint32_t xx) -> unsigned long long {
                        unsigned long long w = 0;
                        if (xx + 8u <= W_) __builtin_memcpy(&w, src + xx, 8);
                        else { _Pragma("unroll 1") for (uint32_t k = 0; k < 8u && xx + k <= xe; ++k) w |= (unsigned long long)src[xx + k] << (8 * k); }
                        return w;
                    };
                    uint32_t x = q.x0;
                    unsigned long long cur = load8(x);
                    for (;;) {
                        const uint32_t rem = min(8u, xe + 1u - x);
                        const bool more = x + 8u <= xe;
                        unsigned long long nxt = 0;
                        if (more) nxt = load8(x + 8u);
                        const unsigned long long vm = rem == 8u ? ~0ull : ((1ull << (8u * rem)) - 1ull);
                        const unsigned long long w = (c.invert ? ~cur : cur) & vm;
                        if (rem == 8u) __builtin_memcpy(px + off, &w, 8);
                        else { _Pragma("unroll 1") for (uint32_t k = 0; k < rem; ++k) px[off + k] = (uint8_t)(w >> (8u * k)); }
                        const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
                        const uint32_t sp_ = __builtin_amdgcn_sad_u8(wh, 0u, __builtin_amdgcn_sad_u8(wl, 0u, 0u));
                        const uint32_t kp = __builtin_amdgcn_udot4(wh, 0x07060504u, __builtin_amdgcn_udot4(wl, 0x03020100u, 0u, false), false);
                        rp += sp_; spx += (uint64_t)(x * sp_ + kp);                         // x * sp_ < 8192 * 2040
                        // extrema: the bytes past the run take the value of the run's pixel at x (a valid one)
                        const unsigned long long wm = w | (((w & 0xffull) * 0x0101010101010101ull) & ~vm);
                        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                        const uint32_t e0 = (uint32_t)wm & 0x00ff00ffu, e1 = ((uint32_t)wm >> 8) & 0x00ff00ffu;
                        const uint32_t e2 = (uint32_t)(wm >> 32) & 0x00ff00ffu, e3 = ((uint32_t)(wm >> 40)) & 0x00ff00ffu;
                        const u16x2 a0 = __builtin_bit_cast(u16x2, e0), a1 = __builtin_bit_cast(u16x2, e1), a2 = __builtin_bit_cast(u16x2, e2), a3 = __builtin_bit_cast(u16x2, e3);
                        const u16x2 mn = __builtin_elementwise_min(__builtin_elementwise_min(a0, a1), __builtin_elementwise_min(a2, a3));
                        const u16x2 mxv = __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(a2, a3));
                        pmin = min(pmin, min((uint32_t)mn.x, (uint32_t)mn.y)); pmax = max(pmax, max((uint32_t)mxv.x, (uint32_t)mxv.y));
                        off += rem;
                        if (!more) break;
                        x += 8u; cur = nxt;
                    }
                    const uint32_t L32 = len, xa = q.x0;
                    const uint64_t tri = (uint64_t)((L32 - 1u) * L32 / 2u);                  // 0 + 1 + .. + (L - 1) < 2^25
                    m10 += (uint64_t)xa * L32 + tri;
                    m20 += (uint64_t)L32 * (xa * xa) + (uint64_t)xa * (2u * (uint32_t)tri) + tri * (uint64_t)(2u * L32 - 1u) / 3u;   // sum (xa + k)^2
                } else {
                // 8 pixels per step: the byte loads of a step are independent, so a run costs len/8 memory round trips, not len;
                // the sums of one step fit 32 bits (8 * 8191^2 < 2^30) and are widened once per step
                uint32_t x = q.x0;
                for (; x <= (uint32_t)q.x1; x += 8) {
                    const uint32_t rem = min(8u, (uint32_t)q.x1 + 1u - x);
                    uint32_t v[8];
                    bool stored = false;                                 // a full step's grey values go out as one 8-byte store
                    if (x + 8u <= (uint32_t)c.W) {
                        // one unaligned 8-byte load instead of eight byte loads (the bytes past the run stay inside the row): the lanes of a
                        // wave read different rows, every load instruction costs the texture path one cycle per lane, and those cycles -- not
                        // the latency -- were what the kernel took
                        unsigned long long w8;
                        __builtin_memcpy(&w8, src + x, 8);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (uint32_t)k < rem ? (uint32_t)(w8 >> (8 * k)) & 0xffu : 0u;
                        if (!enc && rem == 8u) { const unsigned long long o8 = c.invert ? ~w8 : w8; __builtin_memcpy(px + off, &o8, 8); stored = true; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (uint32_t)k < rem ? src[x + k] : 0u;
                    }
                    uint32_t s10 = 0, s20 = 0, sp_ = 0, spx_ = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if ((uint32_t)k >= rem) break;
                        uint32_t p = v[k];
                        if (c.invert) p = 255u - p;
                        const uint32_t xx = x + k;
                        if (!enc) { if (!stored) px[off + k] = (uint8_t)p; }
                        else store_colour(px, off + k, cimg + ((size_t)q.y * c.W + xx) * color_ch, enc);
                        s10 += xx; s20 += __umul24(xx, xx);                 // x < 8192, p < 256: 24-bit multiplies run at full rate
                        sp_ += p; spx_ += __umul24(p, xx);
                        pmin = min(pmin, p); pmax = max(pmax, p);
                    }
                    m10 += s10; m20 += s20; rp += sp_; spx += spx_;
                    off += rem;
                }
                }
                const uint64_t L = len;
                const uint64_t sx = (uint64_t)(q.x0 + q.x1) * L / 2;   // sum of x over the run
                m01 += y * L; m02 += y * y * L; m11 += y * sx;
                sp += rp; spy += rp * y;
            }
        }
        // all eight sums / six extrema of the blob with two transposing reductions (over the blob's LPB lanes)
        const uint64_t sums[8] = {m10, m01, m20, m11, m02, sp, spx, spy};          // = the order of the fields in trexhip_blob
        const uint64_t tot = wave_sum8x64<LPB>(sums, lane);
        const uint32_t ext[8] = {x0, y0, ~x1, ~y1, pmin, ~pmax, 0xffffffffu, 0xffffffffu};
        const uint32_t te = wave_min8x32<LPB>(ext, lane);
        x0 = bcast_lane<LPB, lane_of_q(0)>(te); y0 = bcast_lane<LPB, lane_of_q(1)>(te);
        x1 = ~bcast_lane<LPB, lane_of_q(2)>(te); y1 = ~bcast_lane<LPB, lane_of_q(3)>(te);
        pmin = bcast_lane<LPB, lane_of_q(4)>(te); pmax = ~bcast_lane<LPB, lane_of_q(5)>(te);
        if (!active) continue;
        trexhip_blob* out = blobs + bi;
        if (sub < 8) {
            const uint32_t q = 4u * (sub & 1u) + 2u * ((sub >> 1) & 1u) + ((sub >> 2) & 1u);
            (&out->m10)[q] = tot;
        }
        if (sub == 0) {
            const trexhip_run first = rr[0];
            out->n_pixels = po;
            out->run_begin = B.run_begin - fi.run_begin; out->pix_begin = B.pix_begin - fi.pix_begin;    // the ABI's frame-relative offsets
            out->x0 = (uint16_t)x0; out->y0 = (uint16_t)y0; out->x1 = (uint16_t)x1; out->y1 = (uint16_t)y1;
            out->bid = make_bid(first.x0, first.x1, first.y, B.n_runs);
            out->px_min_max = pmin | (pmax << 8);
        }
    }
}

template <bool COLOUR, int LPB>
__global__ __launch_bounds__(256) void k_gather(const SegCfg c, const int only_pending, const uint8_t* __restrict__ frames,
                                                const uint32_t* __restrict__ totals,
                                                const trexhip_frame_info* __restrict__ info,
                                                const uint32_t* __restrict__ blob_frame,
                                                trexhip_blob* __restrict__ blobs,
                                                const trexhip_run* __restrict__ runs,
                                                uint8_t* __restrict__ pixels, const uint32_t f0, const uint32_t f1,
                                                const uint8_t* __restrict__ color, const int color_ch, const int enc_) {
    constexpr uint32_t BPW = 64 / LPB;
    gather_blobs<COLOUR, LPB>(c, only_pending, frames, info, blob_frame, blobs, runs, pixels, f0, f1, color, color_ch, enc_,
                              (blockIdx.x * 4 + (threadIdx.x >> 6)) * BPW, gridDim.x * 4 * BPW, min(totals[0], c.pool_blobs), -1, 0u, 0u);
}

// ---------------------------------------------------------------------------------------------
// host side launch
#define LAUNCH_GATHER(grid_, stream_, ...) do { if (ctx->p.pixel_encoding != TREXHIP_ENC_GRAY) hipLaunchKernelGGL((k_gather<true, 64>), grid_, dim3(256), 0, stream_, __VA_ARGS__); \
                                                 else hipLaunchKernelGGL((k_gather<false, 32>), grid_, dim3(256), 0, stream_, __VA_ARGS__); } while (0)
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
static void launch_rows(int nch, dim3 grid, hipStream_t s, const uint8_t* frames, const uint8_t* bg,
                        const SegCfg& c, int order, uint32_t* ctr, uint32_t* row_cnt, uint32_t* row_off, uint32_t* tmp, const uint32_t* bits, uint32_t f0) {
    switch (nch) {
        case 1: hipLaunchKernelGGL((k_rows<1, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
        case 2: hipLaunchKernelGGL((k_rows<2, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
        case 3: hipLaunchKernelGGL((k_rows<3, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
        case 4: hipLaunchKernelGGL((k_rows<4, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
        case 5: case 6:
                hipLaunchKernelGGL((k_rows<6, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
        default: hipLaunchKernelGGL((k_rows<8, ALIGNED>), grid, dim3(256), 0, s, frames, bg, c, order, ctr, row_cnt, row_off, tmp, bits, f0); break;
    }
}

int launch_segment(trexhip_ctx* ctx, const uint8_t* d_frames, int n) {
    SegCfg c = ctx->cfg;
    c.B = n;
    ctx->batch_invert = c.invert; ctx->batch_zero_bg = c.zero_bg;      // the batch's own copy: downstream calls on this batch use it
    hipStream_t s_main = ctx->stream;
    hipStream_t s = s_main;
    const int H = c.H, W = c.W;
    const int nch = (W + 1023) / 1024;
    if (nch > 8) { set_error("frame width > 8192 is not supported yet"); return TREXHIP_E_UNSUPPORTED; }
    stage_begin(ctx, TREXHIP_STAGE_SEGMENT_ALL);
    // (no memset in the normal case: the rows kernel zeroes the pooled totals, k_ccl_lds hands every frame counter back zeroed; the array starts
    // zeroed at create.  A pass that was left half-queued by an error return below leaves the flag up: the next one starts from clean counters)
    if (ctx->ctr_dirty) TH_CHECK_HIP(hipMemsetAsync(ctx->d_ctr, 0, sizeof(uint32_t) * ((size_t)ctx->p.max_batch * CTR_STRIDE + 4), s));
    ctx->ctr_dirty = true;
    const unsigned want = (unsigned)(((size_t)H * n + 3) / 4);
    const dim3 grid_rows(want < (unsigned)ctx->tune_rows_blocks ? want : (unsigned)ctx->tune_rows_blocks);
    const bool aligned = (W % 16 == 0) && ((reinterpret_cast<uintptr_t>(d_frames) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(ctx->d_bg) & 15) == 0);
    const uint32_t* bits = nullptr;
    if (ctx->p.use_closing || ctx->p.dilation_size != 0) {
        int rcm = launch_morphology(ctx, d_frames, n, &bits);
        if (rcm) return rcm;
    }
    using LdsM = CclLds<CCL_M_NMAX, CCL_M_SA>;
    using LdsS = CclLds<CCL_S_NMAX, CCL_S_SA>;
    if (!ctx->attr_ccl) {
        TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_lds<1024, CCL_NMAX, CCL_SORT>), hipFuncAttributeMaxDynamicSharedMemorySize, CCL_LDS_BYTES));
        TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_lds<512, CCL_M_NMAX, CCL_M_SA>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsM::BYTES));
        TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_lds<256, CCL_S_NMAX, CCL_S_SA>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsS::BYTES));
        TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ccl_band), hipFuncAttributeMaxDynamicSharedMemorySize, CCLB_LDS_BYTES));
        ctx->attr_ccl = true;
    }
    // Which instance of k_ccl_lds goes first (see the table above the kernel).  Measured in round 6 (profiles/r06_ccl_by_capacity.txt): with one frame
    // per CU or fewer the 1024-thread instance is the fastest whatever the frame holds (C2, 256 frames: L 86 us per pass, S 109; C4: L 270, M 302) --
    // the phases are bound by the threads that walk the lines, not by the barriers; only when a launch brings SEVERAL frames per CU do the small
    // instances pay (C2, 1024 frames: L 327, S 317).  So: L unless the launch holds at least two frames per CU; then the smallest instance that
    // held the frames of the earlier calls, told by two pinned words the kernels write (a frame had more lines than S / than M holds).  A wrong
    // guess costs time, never results: frames the first instance cannot hold are finished by the L instance queued right behind it.  Every 64th
    // call the words are cleared, so a context whose frames got emptier finds its way back down.
    // TREXHIP_CCL_INST (dev): 1 S, 2 M, 3 L.  (M with 1024 threads and S with 512 were measured too: profiles/r06_ccl_by_capacity.txt, variants 4 and 5.)
    int inst = ctx->tune_ccl_inst;
    if ((++ctx->ccl_calls & 63) == 0) { ctx->h_ccl_hint[0] = 0u; ctx->h_ccl_hint[1] = 0u; }
    if (inst <= 0) {
        inst = 3;
        if (n >= 2 * ctx->n_cus) {
            const uint32_t over_s = __atomic_load_n(&ctx->h_ccl_hint[0], __ATOMIC_RELAXED), over_m = __atomic_load_n(&ctx->h_ccl_hint[1], __ATOMIC_RELAXED);
            inst = over_m ? 3 : (over_s ? 2 : 1);
            if (c.R <= CCL_S_NMAX) inst = 1; else if (c.R <= CCL_M_NMAX && inst > 2) inst = 2;     // max_runs itself bounds the lines of a frame
        }
    }
    uint32_t* totals = ctx->d_ctr + (size_t)ctx->p.max_batch * CTR_STRIDE;
    // The batch can be cut into groups of frames so that the labelling of one group (latency chains, one workgroup per frame) runs beside the
    // pixel pass of another (HBM-bound, every CU).  TREXHIP_SEG_GROUPS = G, TREXHIP_SEG_SCHEME:
    //   0  pixel passes on the caller's stream, labelling of group g on an auxiliary stream behind an event (rounds 4-5: slower, the L instance needs a whole CU)
    //   1  group g as a whole (pixel pass, labelling) on stream g % 2; the pixel pass of g waits for the pixel pass of g - 1 (one event per group)
    //   2  the same without events between the groups: the two streams start together, the auxiliary one at the lowest priority
    int G = ctx->tune_seg_groups;
    const int scheme = ctx->tune_seg_scheme;
    if (G > 8) G = 8;
    if (G < 1 || n < 2 * G) G = 1;
    if (G > 1 && !ctx->aux_stream) {
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        TH_CHECK_HIP(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, scheme == 0 ? phi : (scheme == 2 ? plo : 0)));
        // (the events order kernels of this device only: no system-scope fence, i.e. no write-back / invalidate of the caches, when one is recorded)
        static const unsigned ev_flags = std::getenv("TREXHIP_EVENT_FLAGS") ? (unsigned)std::strtoul(std::getenv("TREXHIP_EVENT_FLAGS"), nullptr, 0) : (hipEventDisableTiming | hipEventDisableSystemFence);
        for (int g = 0; g < 10; ++g) TH_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_grp[g], ev_flags));
    }
    int order_bits = ctx->tune_rows_order;
    if (G > 1 && scheme >= 1) {
        // the pooled totals start at zero BEFORE either stream starts (the pixel pass of frame 0 does it otherwise, in stream order ahead of its labelling)
        TH_CHECK_HIP(hipMemsetAsync(totals, 0, 16, s));
        order_bits |= 1 << 30;
        TH_CHECK_HIP(hipEventRecord(ctx->ev_grp[9], s));
        TH_CHECK_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_grp[9], 0));
    }
    const int gs = (n + G - 1) / G;
    // gray pixel arrays: k_ccl_lds gathers the blobs of the frame it has just labelled (TREXHIP_FUSE_GATHER=0: the separate k_gather launch);
    // the colour encodings keep k_gather (one wave per blob, colour addressing)
    // ... when the launch gives (nearly) every CU a frame.  With fewer frames the labelling workgroups are the only ones at work and the gather
    // is better spread over the idle CUs by its own launch (round 6, profiles/r06_fuse_gather.txt; us per detect pass, fused / separate:
    // C4 1 frame 47.7 / 39.3, 16 frames 71.0 / 59.7; C5 64 frames 347 / 313, 16 frames 170 / 134; C2 64 frames 44.8 / 45.8; C4 256 frames: fused 5 us ahead)
    static const int fuse_env = std::getenv("TREXHIP_FUSE_GATHER") ? std::atoi(std::getenv("TREXHIP_FUSE_GATHER")) : -1;
    const bool fuse_gather = ctx->p.pixel_encoding == TREXHIP_ENC_GRAY && (fuse_env >= 0 ? fuse_env != 0 : 4 * n >= 3 * ctx->n_cus);
    stage_begin(ctx, TREXHIP_STAGE_ROWS);
    for (int g = 0; g < G; ++g) {
        const int f0 = g * gs, f1 = (g + 1) * gs < n ? (g + 1) * gs : n;
        if (f0 >= f1) break;
        SegCfg cg = c;
        cg.B = f1 - f0;
        hipStream_t s = s_main;
        if (G > 1 && scheme >= 1) {
            s = (g & 1) ? ctx->aux_stream : s_main;
            if (scheme == 1 && g > 0) TH_CHECK_HIP(hipStreamWaitEvent(s, ctx->ev_grp[g - 1], 0));      // behind the previous group's pixel pass
        }
        const unsigned wantg = (unsigned)(((size_t)H * cg.B + 3) / 4);
        const int nch32 = (W + 2047) / 2048;
        const bool wide = aligned && !bits && W % 32 == 0 && nch32 <= 4 && W >= 1024 && !(ctx->tune_rows_order & 1024);   // TREXHIP_ROWS_ORDER bit 10: 16 pixels per lane
        // rows per wave: the wide kernel is fastest with ~4 rows per wave (measured at 256 frames of 2048^2: 8192 blocks 266 us, 16384: 260,
        // 32768 = 4 rows per wave: 250, 65536: 268); TREXHIP_ROWS_BLOCKS overrides
        unsigned cap = (unsigned)ctx->tune_rows_blocks;
        if (wide && !ctx->tune_rows_blocks_set) { cap = wantg / 4; if (cap < 2048u) cap = 2048u; }
        const dim3 grid_g(wantg < cap ? wantg : cap);
        if (wide) {
            // compile-time modes of the common settings (see exact4_fast); TREXHIP_ROWS_ORDER bit 11 keeps the generic kernel
            int mode = 0;
            if (cg.enable_diff && !cg.invert && cg.tmax >= 255 && cg.tmin >= 1 && !(ctx->tune_rows_order & 2048)) mode = (cg.absdiff ? 1 : 2) | (cg.zero_bg ? 4 : 0);
            // background row in registers for K frames (k_rows32b) when K divides the launch's frames; TREXHIP_ROWS_ORDER bit 2 keeps k_rows32
            int K = 0;
            if (!(ctx->tune_rows_order & 4)) { const int want = ctx->tune_rows_k > 0 ? ctx->tune_rows_k : 8; for (int k = want; k >= 2; --k) if (cg.B % k == 0) { K = k; break; } }
            const dim3 grid_b(K ? (unsigned)(((size_t)H * (cg.B / K) + 3) / 4) : 1u);
            // (k_rows32b for one and, since round 6, two 2048-pixel chunks: the two-chunk form needed its loads written without conditional 128-bit
            // assignments, on which hipcc 7.2 crashes; TREXHIP_ROWS_ORDER bit 3 keeps k_rows32 for two chunks)
#define TH_ROWS32(NCH_, MODE_) do { if (K && NCH_ == 1) hipLaunchKernelGGL((k_rows32b<1, MODE_>), grid_b, dim3(256), 0, s, d_frames, ctx->d_bg, cg, order_bits, K, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_tmp_runs, (uint32_t)f0); \
                                    else if (K && NCH_ == 2 && !(ctx->tune_rows_order & 8)) hipLaunchKernelGGL((k_rows32b<2, MODE_>), grid_b, dim3(256), 0, s, d_frames, ctx->d_bg, cg, order_bits, K, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_tmp_runs, (uint32_t)f0); \
                                    else hipLaunchKernelGGL((k_rows32<NCH_, MODE_>), grid_g, dim3(256), 0, s, d_frames, ctx->d_bg, cg, order_bits, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_tmp_runs, (uint32_t)f0); } while (0)
#define TH_ROWS32_M(NCH_) do { switch (mode) { case 1: TH_ROWS32(NCH_, 1); break; case 2: TH_ROWS32(NCH_, 2); break; case 5: TH_ROWS32(NCH_, 5); break; \
                                               case 6: TH_ROWS32(NCH_, 6); break; default: TH_ROWS32(NCH_, 0); } } while (0)
            switch (nch32) {
                case 1: TH_ROWS32_M(1); break;
                case 2: TH_ROWS32_M(2); break;
                case 3: TH_ROWS32_M(3); break;
                default: TH_ROWS32_M(4); break;
            }
#undef TH_ROWS32_M
#undef TH_ROWS32
        } else
        if (aligned) launch_rows<true>(nch, grid_g, s, d_frames, ctx->d_bg, cg, order_bits, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_tmp_runs, bits, (uint32_t)f0);
        else         launch_rows<false>(nch, grid_g, s, d_frames, ctx->d_bg, cg, order_bits, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_tmp_runs, bits, (uint32_t)f0);
        if (g == G - 1) stage_end(ctx, TREXHIP_STAGE_ROWS);
        hipStream_t t = s;
        if (G > 1 && scheme == 0) {
            TH_CHECK_HIP(hipEventRecord(ctx->ev_grp[g], s));
            TH_CHECK_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_grp[g], 0));
            t = ctx->aux_stream;
        } else if (G > 1 && scheme == 1 && g + 1 < G) TH_CHECK_HIP(hipEventRecord(ctx->ev_grp[g], s));
        // run-level CCL of every frame inside one workgroup's LDS; frames with too many runs are left pending
        // and finished by the global-memory chain in finish_segment()
        // several workgroups per frame (k_ccl_band) when the launch leaves CUs idle AND the frames are heavy: bands of at most CCLB_ROWS rows, as many as
        // there are CUs per frame (at most 8), none of fewer than 64 rows.  The extra launch + the reload cost ~10 us and save phases 2 and 3 of
        // k_ccl_lds: 29 us for a 4096 x 4096 frame of 6.4 k lines, 10 us for a 2048 x 2048 frame of 2.5 k, 2 us for C2 -- measured (us per pass, one
        // workgroup per frame / banded): C5 64 frames 325 / 313, 16 frames 133 / 119; C4 one frame 38.9 / 41.9, 64 frames 95 / 100; C2 one frame 23 / 34.
        // So: only for frames of more lines than the M instance holds (the pinned hint word the kernels of the context's earlier calls wrote).
        // TREXHIP_CCL_BANDS (dev, read at trexhip_create): 0 never, n >= 2 always n bands.  Same tables either way (tests/test_segment_gpu.py).
        int n_bands = 1;
        if (inst == 3) {
            int want = ctx->tune_ccl_bands >= 0 ? ctx->tune_ccl_bands
                                                : ((2 * (f1 - f0) <= ctx->n_cus && __atomic_load_n(&ctx->h_ccl_hint[1], __ATOMIC_RELAXED)) ? ctx->n_cus / (f1 - f0) : 0);
            if (want > 8) want = 8;
            while (want > 1 && (H + want - 1) / want < 64) --want;
            if (want >= 2 && (H + want - 1) / want <= CCLB_ROWS) n_bands = want;
        }
        const int band_rows = n_bands > 1 ? (H + n_bands - 1) / n_bands : 0;
        if (n_bands > 1)
            hipLaunchKernelGGL(k_ccl_band, dim3((f1 - f0) * n_bands), dim3(CCLB_NT), CCLB_LDS_BYTES, t, c, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_row_base, ctx->d_tmp_runs,
                               ctx->d_raster, ctx->d_parent, ctx->d_band_fail, n_bands, band_rows, f0);
#define TH_CCL(NT_, NMAX_, SA_, RETRY_) hipLaunchKernelGGL((k_ccl_lds<NT_, NMAX_, SA_>), dim3(f1 - f0), dim3(NT_), (CclLds<NMAX_, SA_>::BYTES), t, c, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_off, ctx->d_row_base, \
                           ctx->d_tmp_runs, ctx->d_raster, ctx->d_parent, ctx->d_root_ord, ctx->d_cur_run, ctx->d_pix_begin, ctx->d_blob_map,                           \
                           totals, ctx->d_info, ctx->d_blobs, ctx->d_blob_frame, ctx->d_runs, ctx->tune_ccl_stop, reinterpret_cast<unsigned long long*>(ctx->d_cnt_px), f0, \
                           fuse_gather ? d_frames : (const uint8_t*)nullptr, ctx->d_pixels, RETRY_, ctx->h_ccl_hint, (NT_) == 1024 && (NMAX_) == CCL_NMAX && !(RETRY_) ? band_rows : 0, ctx->d_band_fail)
        switch (inst) {
            case 1: TH_CCL(256, CCL_S_NMAX, CCL_S_SA, 0); break;
            case 2: TH_CCL(512, CCL_M_NMAX, CCL_M_SA, 0); break;
            default: TH_CCL(1024, CCL_NMAX, CCL_SORT, 0); break;
        }
        if (inst != 3) TH_CCL(1024, CCL_NMAX, CCL_SORT, 1);      // returns at once for every frame the first instance finished
#undef TH_CCL
        static const int gather_blocks_env = std::getenv("TREXHIP_GATHER_BLOCKS") ? std::atoi(std::getenv("TREXHIP_GATHER_BLOCKS")) : 0;
        // (eight blobs per workgroup; a launch of a few frames does not need 2048 workgroups that look at the total and leave)
        const unsigned gather_need = (unsigned)(((size_t)(f1 - f0) * ctx->p.max_blobs + 7) / 8);
        const unsigned gather_grid = gather_blocks_env > 0 ? (unsigned)gather_blocks_env : (gather_need < 16u ? 16u : (gather_need > 2048u ? 2048u : gather_need));
        if (!fuse_gather)
        LAUNCH_GATHER(dim3(G > 1 ? 256 : gather_grid), t, c, 0, d_frames, totals, ctx->d_info, ctx->d_blob_frame,
                           ctx->d_blobs, ctx->d_runs, ctx->d_pixels, (uint32_t)f0, (uint32_t)f1, ctx->d_color_src, ctx->color_ch, ctx->p.pixel_encoding);
    }
    if (G > 1) {
        TH_CHECK_HIP(hipEventRecord(ctx->ev_grp[8], ctx->aux_stream));
        TH_CHECK_HIP(hipStreamWaitEvent(s_main, ctx->ev_grp[8], 0));
    }
    stage_end(ctx, TREXHIP_STAGE_SEGMENT_ALL);
    TH_CHECK_HIP(hipGetLastError());
    ctx->ctr_dirty = false;              // every frame's labelling kernel is queued behind its rows kernel
    ctx->d_frames = d_frames;
    ctx->last_n = n;
    ctx->fetched = false;
    ctx->pass2.valid_n = 0;
    ctx->pass2.fetched = false;
    return TREXHIP_OK;
}


int launch_pending(trexhip_ctx* ctx) {
    SegCfg c = ctx->cfg;
    c.invert = ctx->batch_invert; c.zero_bg = ctx->batch_zero_bg;
    const int n = ctx->last_n;
    c.B = n;
    hipStream_t s = ctx->stream;
    const dim3 grid_r((unsigned)((n * c.H + 255) / 256));
    uint32_t* totals = ctx->d_ctr + (size_t)ctx->p.max_batch * CTR_STRIDE;
    hipLaunchKernelGGL(k_rowscan, dim3(n), dim3(256), 0, s, c, 1, ctx->d_ctr, ctx->d_row_cnt, ctx->d_row_base, ctx->d_parent, ctx->d_info);
    hipLaunchKernelGGL(k_link, grid_r, dim3(256), 0, s, c, 1, ctx->d_row_cnt, ctx->d_row_off, ctx->d_row_base,
                       ctx->d_tmp_runs, ctx->d_raster, ctx->d_parent, ctx->d_info);
    hipLaunchKernelGGL(k_flatten, grid_r, dim3(256), 0, s, c, 1, ctx->d_row_cnt, ctx->d_row_base, ctx->d_parent, ctx->d_info);
    hipLaunchKernelGGL(k_blobs, dim3(n), dim3(256), 0, s, c, 1, ctx->d_raster, ctx->d_parent, ctx->d_root_ord,
                       ctx->d_cnt_runs, ctx->d_cnt_px, ctx->d_cur_run, ctx->d_pix_begin, ctx->d_blob_map, totals, ctx->d_info,
                       ctx->d_blobs, ctx->d_blob_frame, ctx->d_runs, 0, (const uint32_t*)nullptr);
    LAUNCH_GATHER(dim3(1024), s, c, 1, ctx->d_frames, totals, ctx->d_info, ctx->d_blob_frame,
                       ctx->d_blobs, ctx->d_runs, ctx->d_pixels, 0u, (uint32_t)n, ctx->d_color_src, ctx->color_ch, ctx->p.pixel_encoding);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// track-stage re-threshold (Tracker::prefilter -> pv::Blob::recount + pixel::threshold_blob,
// tracking/Tracker.cpp:765-849; semantics pinned by Application/Tests/test_pixels.cpp: keep a pixel iff
// diff(bg, p) >= threshold, runs split where pixels fail, survivors re-labelled).
// Works on the detect pass's raster-ordered runs: count sub-runs per run -> scan -> write sub-runs
// (raster order is preserved, so the second CCL pass needs no sort) -> k_link2 -> k_flatten -> k_blobs
// (classify instead of drop) -> k_gather.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool track_pass(int p, int b, int method, int thr) {
    const int d = method == 0 ? abs(b - p) : (method == 1 ? max(b - p, 0) : p);
    return d >= thr;
}

// thread per image row: for every run of a KEPT detect blob count (write=0) or emit (write=1) its sub-runs
template <int WRITE>
__global__ __launch_bounds__(256) void k_sub(const SegCfg c, const uint8_t* __restrict__ frames, const uint8_t* __restrict__ bg,
                                             const uint32_t* __restrict__ row_base, const trexhip_run* __restrict__ raster,
                                             const uint32_t* __restrict__ label, const uint32_t* __restrict__ root_ord,
                                             const int32_t* __restrict__ blob_map, const trexhip_frame_info* __restrict__ info,
                                             const int method, const int thr_all, const int32_t* __restrict__ blob_thr,
                                             uint32_t* __restrict__ sub_cnt,
                                             const uint32_t* __restrict__ sub_base, trexhip_run* __restrict__ raster2,
                                             uint32_t* __restrict__ run_parent2) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= c.B * c.H) return;
    const int f = gid / c.H, y = gid - f * c.H;
    const trexhip_frame_info fi = info[f];
    if (fi.flags) return;
    const uint32_t* rb = row_base + (size_t)f * (c.H + 1);
    const size_t fo = (size_t)f * c.R;
    const uint8_t* img = frames + ((size_t)f * c.H + y) * c.W;
    const uint8_t* bgr = bg + (size_t)y * c.W;
    for (uint32_t r = rb[y]; r < rb[y + 1]; ++r) {
        const int32_t k = blob_map[fo + root_ord[fo + label[fo + r]]];
        uint32_t n = 0;
        // per-blob thresholds (SplitBlob::apply_threshold tries different thresholds per blob); negative = leave the blob out
        const int thr = (k >= 0 && blob_thr) ? blob_thr[fi.blob_begin + k] : thr_all;
        if (k >= 0 && thr >= 0) {
            const trexhip_run q = raster[fo + r];
            uint32_t out = WRITE ? sub_base[fo + r] : 0;
            int open = -1;
            for (int x = q.x0; x <= q.x1; ++x) {
                int p = img[x];
                if (c.invert) p = 255 - p;
                const bool ok = track_pass(p, bgr[x], method, thr);
                if (ok && open < 0) open = x;
                if (!ok && open >= 0) {
                    if (WRITE) { trexhip_run s; s.x0 = (uint16_t)open; s.x1 = (uint16_t)(x - 1); s.y = (uint16_t)y; s.pad = 0;
                                 raster2[fo + out] = s; run_parent2[fo + out] = fi.blob_begin + (uint32_t)k; ++out; }
                    ++n; open = -1;
                }
            }
            if (open >= 0) {
                if (WRITE) { trexhip_run s; s.x0 = (uint16_t)open; s.x1 = q.x1; s.y = (uint16_t)y; s.pad = 0;
                             raster2[fo + out] = s; run_parent2[fo + out] = fi.blob_begin + (uint32_t)k; }
                ++n;
            }
        }
        if (!WRITE) sub_cnt[fo + r] = n;
    }
}

// block per frame: exclusive scan of sub-run counts in raster order -> sub_base, second-pass row tables, parent init
__global__ __launch_bounds__(256) void k_sub_scan(const SegCfg c, const trexhip_frame_info* __restrict__ info1,
                                                  const uint32_t* __restrict__ row_base1, const uint32_t* __restrict__ sub_cnt,
                                                  uint32_t* __restrict__ sub_base, uint32_t* __restrict__ row_base2,
                                                  uint32_t* __restrict__ row_cnt2, uint32_t* __restrict__ parent2,
                                                  trexhip_frame_info* __restrict__ info2) {
    __shared__ uint32_t lds[8];
    const int f = blockIdx.x;
    const trexhip_frame_info fi = info1[f];
    trexhip_frame_info out = {};
    if (fi.flags) { if (threadIdx.x == 0) { out.flags = fi.flags; info2[f] = out; } return; }
    const size_t fo = (size_t)f * c.R;
    const uint32_t n = fi.n_raw_runs;
    uint32_t running = 0;
    for (uint32_t b0 = 0; b0 < n; b0 += 256) {
        const uint32_t r = b0 + threadIdx.x;
        const uint32_t v = r < n ? sub_cnt[fo + r] : 0;
        uint32_t total;
        const uint32_t ex = block_excl_scan(v, lds, total);
        if (r < n) sub_base[fo + r] = running + ex;
        running += total;
    }
    const uint32_t n2 = running;
    __threadfence_block();
    __syncthreads();
    const uint32_t* rb1 = row_base1 + (size_t)f * (c.H + 1);
    uint32_t* rb2 = row_base2 + (size_t)f * (c.H + 1);
    for (int y = threadIdx.x; y <= c.H; y += 256) {
        const uint32_t r = rb1[y];
        rb2[y] = r < n ? sub_base[fo + r] : n2;
    }
    __threadfence_block();
    __syncthreads();
    for (int y = threadIdx.x; y < c.H; y += 256) row_cnt2[(size_t)f * c.H + y] = rb2[y + 1] - rb2[y];
    const bool overflow = n2 > (uint32_t)c.R;
    if (threadIdx.x == 0) { out.n_raw_runs = n2; out.flags = overflow ? TREXHIP_FRAME_OVERFLOW_RUNS : 0u; info2[f] = out; }
    if (!overflow) for (uint32_t r = threadIdx.x; r < n2; r += 256) parent2[fo + r] = r;
}

// thread per row: union the sub-runs of row y with the touching sub-runs of row y-1 (both already in raster order)
__global__ __launch_bounds__(256) void k_link2(const SegCfg c, const uint32_t* __restrict__ row_base,
                                               const trexhip_run* __restrict__ raster, uint32_t* __restrict__ parent,
                                               const trexhip_frame_info* __restrict__ info) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= c.B * c.H) return;
    const int f = gid / c.H, y = gid - f * c.H;
    if (y == 0 || info[f].flags) return;
    const uint32_t* rb = row_base + (size_t)f * (c.H + 1);
    uint32_t j = rb[y - 1], i = rb[y];
    const uint32_t je = rb[y], ie = rb[y + 1];
    if (i >= ie || j >= je) return;
    const trexhip_run* rr = raster + (size_t)f * c.R;
    uint32_t* par = parent + (size_t)f * c.R;
    trexhip_run cur = rr[i], prv = rr[j];
    const int slack = c.slack;
    for (;;) {
        if ((int)prv.x1 + slack >= (int)cur.x0 && (int)cur.x1 + slack >= (int)prv.x0) uf_union(par, j, i);
        if (prv.x1 < cur.x1) { if (++j >= je) break; prv = rr[j]; }
        else                 { if (++i >= ie) break; cur = rr[i]; }
    }
}

int launch_rethreshold(trexhip_ctx* ctx, int thr, int method, const double* ranges, int n_ranges, const int32_t* d_blob_thr) {
    Pass2& q = ctx->pass2;
    SegCfg c = ctx->cfg;
    c.invert = ctx->batch_invert; c.zero_bg = ctx->batch_zero_bg;
    const int n = ctx->last_n;
    c.B = n;
    c.n_ranges = n_ranges;
    for (int i = 0; i < 2 * n_ranges; ++i) c.ranges[i] = ranges[i];
    hipStream_t s = ctx->stream;
    const dim3 grid_r((unsigned)((n * c.H + 255) / 256));
    TH_CHECK_HIP(hipMemsetAsync(q.d_totals, 0, sizeof(uint32_t) * 4, s));
    hipLaunchKernelGGL((k_sub<0>), grid_r, dim3(256), 0, s, c, ctx->d_frames, ctx->d_bg, ctx->d_row_base, ctx->d_raster, ctx->d_parent,
                       ctx->d_root_ord, ctx->d_blob_map, ctx->d_info, method, thr, d_blob_thr, q.d_sub_cnt, q.d_sub_base, q.d_raster, q.d_run_parent);
    hipLaunchKernelGGL(k_sub_scan, dim3(n), dim3(256), 0, s, c, ctx->d_info, ctx->d_row_base, q.d_sub_cnt, q.d_sub_base, q.d_row_base,
                       q.d_row_cnt, q.d_parent, q.d_info);
    hipLaunchKernelGGL((k_sub<1>), grid_r, dim3(256), 0, s, c, ctx->d_frames, ctx->d_bg, ctx->d_row_base, ctx->d_raster, ctx->d_parent,
                       ctx->d_root_ord, ctx->d_blob_map, ctx->d_info, method, thr, d_blob_thr, q.d_sub_cnt, q.d_sub_base, q.d_raster, q.d_run_parent);
    hipLaunchKernelGGL(k_link2, grid_r, dim3(256), 0, s, c, q.d_row_base, q.d_raster, q.d_parent, q.d_info);
    hipLaunchKernelGGL(k_flatten, grid_r, dim3(256), 0, s, c, 0, q.d_row_cnt, q.d_row_base, q.d_parent, q.d_info);
    hipLaunchKernelGGL(k_blobs, dim3(n), dim3(256), 0, s, c, 0, q.d_raster, q.d_parent, q.d_root_ord, q.d_cnt_runs, q.d_cnt_px, q.d_cur_run,
                       q.d_pix_begin, q.d_blob_map, q.d_totals, q.d_info, q.d_blobs, q.d_blob_frame, q.d_runs, 1, q.d_run_parent);
    LAUNCH_GATHER(dim3(1024), s, c, 0, ctx->d_frames, q.d_totals, q.d_info, q.d_blob_frame, q.d_blobs,
                       q.d_runs, q.d_pixels, 0u, (uint32_t)n, ctx->d_color_src, ctx->color_ch, ctx->p.pixel_encoding);
    TH_CHECK_HIP(hipGetLastError());
    q.valid_n = n;
    q.fetched = false;
    return TREXHIP_OK;
}

}  // namespace trexhip
