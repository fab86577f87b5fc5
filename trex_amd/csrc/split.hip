// split.hip -- SplitBlob's threshold search for merged blobs, one wave per blob, everything in LDS.
//
// Replaces the search of Application/src/tracker/tracking/SplitBlob.cpp:
//   SplitBlob::apply_threshold            :130-179   difference value of every blob pixel, threshold_blob at a threshold
//   SplitBlob::evaluate_result_multiple   :193-255   ABORT / REMOVE / TOO_FEW / KEEP_ABORT from the sub-blob sizes
//   SplitBlob::split                      :419-800   blob_split_algorithm threshold (complete) and threshold_approximate
// The reference re-labels the blob once per tried threshold on the CPU (up to ~250 times per blob); here the blob's
// difference values stay in LDS, every evaluation is a run-level union-find of one wave, and thresholds that do not
// change the pixel set (empty histogram bins) reuse the previous evaluation.  The kernel only finds the threshold; the
// sub-blobs themselves come from trexhip_rethreshold_per_blob_device with the thresholds written here.
// Compiled with -ffp-contract=off: the size comparisons are float products compared like the reference's.
#include "internal.h"

namespace trexhip {

// two size classes: most merged blobs are a few hundred pixels, and the kernel lives on the number of blobs in flight per CU
static constexpr int S_PX = 16384;     // pixels per blob held in LDS (large class)
static constexpr int S_RUNS = 1024;    // lines per blob
static constexpr int S_SUB = 2048;     // lines after thresholding
static constexpr int S_PX_SMALL = 2048, S_RUNS_SMALL = 256, S_SUB_SMALL = 1024;   // 2048 pixels cannot make more than 1024 lines
static constexpr int S_PX_HUGE = 61440, S_RUNS_HUGE = 2048, S_SUB_HUGE = 4096;   // two merged 600-row animals; pixel offsets stay 16-bit
constexpr int split_lds_bytes(int px, int runs, int sub) { return runs * 4 + sub * 12 + 1024 + runs * 2 + (runs + 2) * 2 + sub * 2 + 256 + px; }

enum { A_KEEP = 0, A_KEEP_ABORT = 1, A_REMOVE = 2, A_ABORT = 3, A_TOO_FEW = 4, A_SKIP = 5, A_NO_CHANCE = 6 };

struct SplitCfg {
    int W, H, B, invert, slack, method, initial_threshold, algorithm, n_ranges;
    float sqcm, max_shrink, global_shrink;
    double ranges[16];
    double max_start, max_end;         // SizeFilters::max_range()
};

__device__ __forceinline__ uint32_t wmax_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}
__device__ __forceinline__ uint32_t wmin_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}
__device__ __forceinline__ uint32_t wsum_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
    return v;
}
__device__ __forceinline__ uint32_t wexcl_scan(uint32_t v, int lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, d); if (lane >= d) x += y; }
    total = (uint32_t)__shfl((int)x, 63);
    return x - v;
}
__device__ __forceinline__ uint32_t lds_find(const volatile uint32_t* par, uint32_t a) {
    uint32_t p = par[a];
    while (p != a) { a = p; p = par[a]; }
    return a;
}
__device__ __forceinline__ void lds_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        const uint32_t old = atomicMin(par + b, a);
        if (old == b) return;
        b = old;
    }
}
__device__ __forceinline__ bool split_in_range(const SplitCfg& C, float cmsq) {     // SizeFilters::in_range_of_one, scale_factor -1
    if (C.n_ranges <= 0) return true;
    for (int i = 0; i < C.n_ranges; ++i) if ((double)cmsq >= C.ranges[2 * i] && (double)cmsq < C.ranges[2 * i + 1]) return true;
    return false;
}

template <int S_PX, int S_RUNS, int S_SUB, int MIN_PX, int MIN_RUNS>
__global__ __launch_bounds__(64) void k_split_search(const SplitCfg C, const uint8_t* __restrict__ frames, const uint8_t* __restrict__ bg,
                                                     const trexhip_frame_info* __restrict__ info, const uint32_t* __restrict__ blob_frame,
                                                     const trexhip_blob* __restrict__ blobs, const trexhip_run* __restrict__ runs,
                                                     const int32_t* __restrict__ presumed, int n_blobs, int32_t* __restrict__ out_thr,
                                                     trexhip_split_info* __restrict__ out_info) {
    // dynamic LDS (the largest size class needs more than the 64 KB a static allocation may have), carved by descending alignment
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    uint32_t* s_rx = reinterpret_cast<uint32_t*>(s_lds);
    uint32_t* s_sx = s_rx + S_RUNS;
    uint32_t* s_par = s_sx + S_SUB;
    uint32_t* s_size = s_par + S_SUB;
    uint32_t* s_hist = s_size + S_SUB;
    uint16_t* s_ry = reinterpret_cast<uint16_t*>(s_hist + 256);
    uint16_t* s_roff = s_ry + S_RUNS;
    uint16_t* s_sy = s_roff + (S_RUNS + 2);
    uint8_t* s_cache = reinterpret_cast<uint8_t*>(s_sy + S_SUB);
    uint8_t* s_d = s_cache + 256;

    const int bi = blockIdx.x, lane = threadIdx.x;
    if (bi >= n_blobs) return;
    trexhip_split_info res = {};
    res.threshold = -1; res.effective_threshold = -1; res.initial_action = A_SKIP;
    const int pn = presumed[bi];
    const uint32_t f = blob_frame[bi];
    bool ok = pn > 0 && C.algorithm != 0 && f < (uint32_t)C.B;
    trexhip_frame_info fi = {};
    if (ok) { fi = info[f]; ok = fi.flags == 0; }
    if (!ok) { if (lane == 0) { res.status = 3; out_info[bi] = res; out_thr[bi] = -1; } return; }
    const trexhip_blob Bl = blobs[bi];
    const int n_runs = (int)Bl.n_runs, npx = (int)Bl.n_pixels;
    if (npx <= MIN_PX && n_runs <= MIN_RUNS) return;           // a smaller size class handles it
    if (n_runs > S_RUNS || npx > S_PX || n_runs == 0) {
        if (n_runs && npx <= S_PX_HUGE && n_runs <= S_RUNS_HUGE && S_PX < S_PX_HUGE) {
            // left to a larger size class; the host launches the last one only when the batch holds such a blob, so mark it beyond capacity first
            if (S_PX == trexhip::S_PX && lane == 0) { res.status = 2; out_info[bi] = res; out_thr[bi] = -1; }
            return;
        }
        if (lane == 0) { res.status = n_runs ? 2 : 3; out_info[bi] = res; out_thr[bi] = -1; }
        return;
    }

    // ---- the blob's lines and pixel offsets ----
    const trexhip_run* rr = runs + fi.run_begin + Bl.run_begin;
    {
        uint32_t running = 0;
        for (int c0 = 0; c0 < n_runs; c0 += 64) {
            const int i = c0 + lane;
            uint32_t len = 0;
            if (i < n_runs) { const trexhip_run q = rr[i]; s_rx[i] = (uint32_t)q.x0 | ((uint32_t)q.x1 << 16); s_ry[i] = q.y; len = (uint32_t)q.x1 - q.x0 + 1u; }
            uint32_t total;
            const uint32_t ex = wexcl_scan(len, lane, total);
            if (i < n_runs) s_roff[i] = (uint16_t)(running + ex);
            running += total;
        }
        if (lane == 0) s_roff[n_runs] = (uint16_t)npx;        // npx <= 16384
    }
    for (int i = lane; i < 256; i += 64) { s_hist[i] = 0; s_cache[i] = A_NO_CHANCE; }
    __syncthreads();
    // ---- difference values (SplitBlob.cpp:131-160): Background::diff of the method the tracker uses ----
    {
        const uint8_t* img = frames + (size_t)f * C.H * C.W;
        for (int p = lane; p < npx; p += 64) {
            int lo = 0, hi = n_runs - 1;                      // last line whose offset <= p
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)s_roff[mid] <= p) lo = mid; else hi = mid - 1; }
            const int x = (int)(s_rx[lo] & 0xffffu) + (p - (int)s_roff[lo]), y = s_ry[lo];
            int v = img[(size_t)y * C.W + x];
            if (C.invert) v = 255 - v;
            const int b = bg[(size_t)y * C.W + x];
            const int d = C.method == 0 ? abs(b - v) : (C.method == 1 ? max(b - v, 0) : v);
            s_d[p] = (uint8_t)d;
            atomicAdd(&s_hist[d], 1u);
        }
    }
    __syncthreads();
    int min_pixel, max_pixel;
    {
        uint32_t mn = 254u, mx = 0u;                          // :142-143
        for (int i = lane; i < 256; i += 64) if (s_hist[i]) { mn = min(mn, (uint32_t)i); mx = max(mx, (uint32_t)i); }
        min_pixel = (int)wmin_u32(mn); max_pixel = (int)wmax_u32(mx);
    }

    int n_eval = 0;
    bool capacity = false;
    // one evaluation: threshold_blob(diff >= t) as a run-level union-find, then evaluate_result_multiple (:193-255)
    auto evaluate = [&](int t, float first_size, float& max_size, int& n_kept, double& bound) -> int {
        ++n_eval;
        // lines after thresholding, raster order kept
        uint32_t nsub = 0;
        for (int c0 = 0; c0 < n_runs; c0 += 64) {
            const int i = c0 + lane;
            uint32_t cnt = 0;
            int off = 0, len = 0, x0 = 0;
            if (i < n_runs) {
                off = s_roff[i]; len = (int)s_roff[i + 1] - off; x0 = (int)(s_rx[i] & 0xffffu);
                bool open = false;
                for (int j = 0; j < len; ++j) { const bool k = (int)s_d[off + j] >= t; cnt += (k && !open) ? 1u : 0u; open = k; }
            }
            uint32_t total;
            uint32_t o = nsub + wexcl_scan(cnt, lane, total);
            if (nsub + total > (uint32_t)S_SUB) { capacity = true; return A_NO_CHANCE; }
            if (cnt) {
                int start = -1;
                for (int j = 0; j < len; ++j) {
                    const bool k = (int)s_d[off + j] >= t;
                    if (k && start < 0) start = j;
                    if (!k && start >= 0) { s_sx[o] = (uint32_t)(x0 + start) | ((uint32_t)(x0 + j - 1) << 16); s_sy[o] = s_ry[i]; ++o; start = -1; }
                }
                if (start >= 0) { s_sx[o] = (uint32_t)(x0 + start) | ((uint32_t)(x0 + len - 1) << 16); s_sy[o] = s_ry[i]; }
            }
            nsub += total;
        }
        for (uint32_t i = lane; i < nsub; i += 64) { s_par[i] = i; s_size[i] = 0; }
        __syncthreads();
        // link every line with the touching lines of the row above
        for (uint32_t i = lane; i < nsub; i += 64) {
            const int y = s_sy[i];
            const int cx0 = (int)(s_sx[i] & 0xffffu), cx1 = (int)(s_sx[i] >> 16);
            int lo = 0, hi = (int)i;                          // first line with sy >= y - 1
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)s_sy[mid] < y - 1) lo = mid + 1; else hi = mid; }
            for (int j = lo; j < (int)i && (int)s_sy[j] == y - 1; ++j) {
                const int px0 = (int)(s_sx[j] & 0xffffu), px1 = (int)(s_sx[j] >> 16);
                if (px0 > cx1 + C.slack) break;
                if (px1 + C.slack >= cx0) lds_union(s_par, (uint32_t)j, i);
            }
        }
        __syncthreads();
        for (uint32_t i = lane; i < nsub; i += 64) {
            const uint32_t root = lds_find(s_par, i);
            atomicAdd(&s_size[root], (s_sx[i] >> 16) - (s_sx[i] & 0xffffu) + 1u);
        }
        __syncthreads();
        // sizes of the sub-blobs = s_size at the roots
        uint32_t pix = 0, big = 0;
        for (uint32_t i = lane; i < nsub; i += 64) if (s_par[i] == i) { pix += s_size[i]; big = max(big, s_size[i]); }
        const uint32_t pixels = wsum_u32(pix);
        max_size = (float)wmax_u32(big) * C.sqcm;             // apply_threshold's return value times sqrcm (:523-526)
        n_kept = 0; bound = 0;
        if ((float)pixels * C.sqcm < C.max_shrink * first_size) return A_ABORT;
        if (C.n_ranges > 0) bound = C.max_start * (double)C.global_shrink;
        else bound = (double)((float)pixels * C.sqcm * C.max_shrink);
        uint32_t kept = 0;
        for (uint32_t i = lane; i < nsub; i += 64) {
            if (s_par[i] != i) { s_size[i] = 0; continue; }
            if ((double)((float)s_size[i] * C.sqcm) < bound) s_size[i] = 0; else ++kept;
        }
        n_kept = (int)wsum_u32(kept);
        __syncthreads();
        // the presumed_nr largest of what is left (:225-236)
        uint32_t valid = 0, min_size = 0;
        bool has_min = false;
        for (int r = 0; r < pn; ++r) {
            uint32_t key = 0;
            for (uint32_t i = lane; i < nsub; i += 64) if (s_size[i]) key = max(key, (s_size[i] << 12) | i);
            key = wmax_u32(key);
            if (!key) break;
            const uint32_t sz = key >> 12;
            if (lane == 0) s_size[key & 4095u] = 0;
            __syncthreads();
            min_size = sz; has_min = true;                    // descending: the last taken is the smallest
            if (split_in_range(C, (float)sz * C.sqcm)) ++valid;
        }
        if (C.n_ranges > 0 && has_min && (double)((float)min_size * C.sqcm) > C.max_end) return A_REMOVE;
        if (valid < (uint32_t)pn) return A_TOO_FEW;
        return A_KEEP_ABORT;
    };

    float first_size = 0.f, max_size = 0.f;
    int n_kept = 0; double bound = 0;
    const int begin = max(C.initial_threshold, min_pixel);    // first apply_threshold clamps to min_pixel (:161); begin_threshold (:592)
    int action = evaluate(begin, 0.f, max_size, n_kept, bound);   // try_threshold(-1) (:558)
    first_size = max_size;
    res.initial_action = action;
    int best = -1, best_eff = -1;
    if (action == A_KEEP_ABORT) { best = C.initial_threshold; best_eff = begin; }
    // smallest difference value >= t: thresholds in between keep the same pixels and evaluate the same
    auto canonical = [&](int t) -> int { int c = t; while (c < 256 && s_hist[c] == 0) ++c; return c; };
    auto perform = [&](int t) -> int {                        // Run::perform (:318-337) with the evaluation memoised per pixel set
        const int c = canonical(t);
        int a = c < 256 ? (int)s_cache[c] : A_NO_CHANCE;
        if (a == A_NO_CHANCE) {
            float ms; int nk; double bd;
            a = evaluate(t, first_size, ms, nk, bd);
            if (c < 256 && lane == 0) s_cache[c] = (uint8_t)a;
            __syncthreads();
        }
        if (a == A_KEEP_ABORT && (best == -1 || t < best)) { best = t; best_eff = t; }
        return a;
    };
    const bool guard = C.n_ranges <= 0 || (double)((float)npx * C.sqcm) < C.max_end * 100.0;     // :560-563
    if (!capacity && action != A_KEEP_ABORT && guard && pn > 1) {
        if (C.algorithm == 1) {                               // complete search (:711-717)
            for (int t = begin; t < max_pixel && !capacity; ++t) {
                const int a = perform(t);
                if (a == A_ABORT || a == A_KEEP_ABORT) break;
            }
        } else {                                              // threshold_approximate, sequential form (:609-706,719-726)
            const int start = begin, end = max_pixel;
            const int fs_end = start + (int)((double)(end - start) * 0.3);
            for (int ti = 0; ti < 3 && best == -1 && !capacity; ++ti) {
                bool done = false;
                for (int offset = 0; offset < 2 && !done; ++offset) {
                    if (best != -1) break;
                    for (int t = start + ti * 2 + offset; t < fs_end && !capacity; t += 6) {
                        if (best != -1 && t >= best) break;
                        const int a = perform(t);
                        if (a == A_ABORT || a == A_KEEP_ABORT) { if (a == A_KEEP_ABORT) done = true; break; }
                    }
                }
                if (done || best != -1) continue;
                for (int t = fs_end + ti; t < end && !capacity; t += 3) {
                    if (best != -1 && t >= best) break;
                    const int a = perform(t);
                    if (a == A_ABORT || a == A_KEEP_ABORT) break;
                }
            }
        }
    }
    if (!capacity && best != -1 && action != A_KEEP_ABORT) evaluate(best_eff, first_size, max_size, n_kept, bound);   // sizes of the saved result
    if (lane == 0) {
        if (capacity) { res.status = 2; best = -1; best_eff = -1; }
        res.threshold = best; res.effective_threshold = best_eff;
        res.n_result = best != -1 ? n_kept : 0;
        res.min_size_bound = best != -1 ? bound : 0.0;
        res.n_evaluated = n_eval; res.min_pixel = min_pixel; res.max_pixel = max_pixel; res.first_size = first_size;
        out_info[bi] = res;
        out_thr[bi] = best_eff;
    }
}

int launch_split_search(trexhip_ctx* ctx, const trexhip_split_params* sp, int method, const int32_t* d_presumed, int n_blobs, int32_t* d_thr,
                        trexhip_split_info* d_info) {
    SplitCfg C = {};
    C.W = ctx->cfg.W; C.H = ctx->cfg.H; C.B = ctx->last_n; C.invert = ctx->batch_invert; C.slack = ctx->cfg.slack; C.method = method;
    C.initial_threshold = (sp->calculate_posture ? max(sp->track_threshold, sp->track_posture_threshold) : sp->track_threshold) + 1;   // :512
    C.algorithm = sp->algorithm; C.n_ranges = sp->n_ranges;
    C.sqcm = ctx->cfg.sqcm; C.max_shrink = sp->blob_split_max_shrink; C.global_shrink = sp->blob_split_global_shrink_limit;
    C.max_start = -1; C.max_end = -1;
    for (int i = 0; i < sp->n_ranges; ++i) {                  // SizeFilters::add (core/SizeFilters.cpp:12-18)
        C.ranges[2 * i] = sp->size_ranges[2 * i]; C.ranges[2 * i + 1] = sp->size_ranges[2 * i + 1];
        if (C.max_start == -1 || C.ranges[2 * i] < C.max_start) C.max_start = C.ranges[2 * i];
        if (C.max_end == -1 || C.ranges[2 * i + 1] > C.max_end) C.max_end = C.ranges[2 * i + 1];
    }
    hipLaunchKernelGGL((k_split_search<S_PX_SMALL, S_RUNS_SMALL, S_SUB_SMALL, 0, 0>), dim3((unsigned)n_blobs), dim3(64),
                       split_lds_bytes(S_PX_SMALL, S_RUNS_SMALL, S_SUB_SMALL), ctx->stream, C, ctx->d_frames,
                       ctx->d_bg, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs, ctx->d_runs, d_presumed, n_blobs, d_thr, d_info);
    hipLaunchKernelGGL((k_split_search<S_PX, S_RUNS, S_SUB, S_PX_SMALL, S_RUNS_SMALL>), dim3((unsigned)n_blobs), dim3(64),
                       split_lds_bytes(S_PX, S_RUNS, S_SUB), ctx->stream, C, ctx->d_frames,
                       ctx->d_bg, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs, ctx->d_runs, d_presumed, n_blobs, d_thr, d_info);
    // the third size class (134 KB of LDS: one blob per CU) only when the fetched tables hold a blob of that size
    bool huge = false;
    for (int i = 0; i < n_blobs && !huge; ++i) huge = ctx->h_blobs[i].n_pixels > (uint32_t)S_PX || ctx->h_blobs[i].n_runs > (uint32_t)S_RUNS;
    if (huge) {
        const int bytes = split_lds_bytes(S_PX_HUGE, S_RUNS_HUGE, S_SUB_HUGE);
        if (!ctx->attr_split) {
            TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_split_search<S_PX_HUGE, S_RUNS_HUGE, S_SUB_HUGE, S_PX, S_RUNS>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            ctx->attr_split = true;
        }
        hipLaunchKernelGGL((k_split_search<S_PX_HUGE, S_RUNS_HUGE, S_SUB_HUGE, S_PX, S_RUNS>), dim3((unsigned)n_blobs), dim3(64), bytes, ctx->stream, C,
                           ctx->d_frames, ctx->d_bg, ctx->d_info, ctx->d_blob_frame, ctx->d_blobs, ctx->d_runs, d_presumed, n_blobs, d_thr, d_info);
    }
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

}  // namespace trexhip

using namespace trexhip;

extern "C" void trexhip_default_split_params(trexhip_split_params* p) {
    if (!p) return;
    *p = trexhip_split_params{};
    p->track_threshold = 15; p->track_posture_threshold = 15; p->calculate_posture = 1; p->algorithm = 1;
    p->blob_split_max_shrink = 0.2f; p->blob_split_global_shrink_limit = 0.2f;
}

extern "C" int trexhip_split_search_device(trexhip_ctx* ctx, const trexhip_split_params* sp, int32_t method, const int32_t* d_presumed_nr,
                                           int32_t n_blobs, int32_t* d_thresholds, trexhip_split_info* d_info) {
    if (!ctx || !sp || !d_presumed_nr || !d_thresholds || !d_info) { set_error("trexhip_split_search_device: null argument"); return TREXHIP_E_INVALID; }
    if (method < 0 || method > 2) { set_error("trexhip_split_search_device: method must be 0 (absolute), 1 (signed) or 2 (none)"); return TREXHIP_E_INVALID; }
    if (sp->algorithm == 3 || sp->algorithm == 4) { set_error("trexhip_split_search_device: blob_split_algorithm = fill / fill_approximate (cv::watershed, SplitBlob.cpp:419-485) is not implemented by this backend"); return TREXHIP_E_UNSUPPORTED; }
    if (sp->algorithm < 0 || sp->algorithm > 2) { set_error("trexhip_split_search_device: algorithm must be 0 (none), 1 (threshold) or 2 (threshold_approximate)"); return TREXHIP_E_INVALID; }
    if (sp->n_ranges < 0 || sp->n_ranges > 8) { set_error("trexhip_split_search_device: at most 8 size ranges"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0 || !ctx->fetched) { set_error("trexhip_split_search_device: segment and fetch a batch first"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_split_search_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    return launch_split_search(ctx, sp, method, d_presumed_nr, n_blobs, d_thresholds, d_info);
}
