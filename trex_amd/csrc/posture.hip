// posture.hip -- posture::calculate_posture for every blob of a segmented batch, one wave per blob.
//
// Replaces Application/src/tracker/tracking/Posture.cpp:305-399 (one threshold iteration: the blob table handed in is
// either the detect table or the re-threshold table, i.e. threshold_get_biggest_blob is done by the caller choosing
// the sub-blob) and Outline.cpp:
//   outline of the blob (pixel::find_outer_points, commons)     -> trace on the half-pixel lattice (definition: DESIGN.md section 2)
//   Outline::resample              Outline.cpp:724-766            sequential float walk, lane 0, same operation order
//   smooth_outline                 Outline.cpp:330-378            one lane per point
//   Outline::offset_to_middle      Outline.cpp:454-718            clockwise test, EFT(order) -> inverse, curvature, tail/head
//   Outline::calculate_midline     Outline.cpp:768-868            two-pointer walk, lane 0
// Everything of one blob lives in the wave's slice of LDS (runs, row table, two point buffers, curvature).
// The float pipeline is compiled without FMA contraction (-ffp-contract=off in the Makefile) so resample / smooth /
// walk reproduce the CPU operation order; the transcendental part (EFT) agrees to float rounding.
#include "internal.h"
#include <algorithm>
#include <type_traits>

namespace trexhip {

static constexpr int P_NP = 4096;      // largest max_points (one blob per workgroup beyond ~1300: the LDS of a CU holds fewer large blobs)
static constexpr int P_NR = 2048;      // runs per blob held in LDS
static constexpr int P_ROWS = 1022;    // rows per blob
// bytes of LDS per wave for a given point capacity: two point buffers, curvature, arc length, runs, row table
// (nr / nrows = line and row capacity of this launch: the host sizes them to the largest blob of the fetched table, so that more
// blobs fit into a CU's LDS -- the kernel is a chain of LDS / memory latencies and lives on the number of blobs in flight)
__host__ __device__ constexpr int posture_wave_lds(int np, int nr, int nrows) { return np * 8 * 2 + np * 4 * 2 + nr * 4 + (nrows + 2) * 4; }

#ifdef TREXHIP_DEV_KNOBS
#define POSTURE_STOP(n) do { if (P.stop == (n)) return; } while (0)
#else
#define POSTURE_STOP(n) do { } while (0)
#endif
struct PostureCfg {
    float outline_resample; int smooth_samples, smooth_step, approximate;
    float curvature_range_ratio, midline_walk_offset; int max_points;
    int nr_cap, rows_cap;            // <= P_NR, P_ROWS
    int stop;                        // dev only: return after phase N (TREXHIP_POSTURE_STOP)
    int walk_group;                  // bits 8 / 16: blobs whose two-pointer walk searches at most 8 / 16 candidates leave the walk to k_posture_walk<8> / <16>
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// sin / cos of the EFT phases: the same operations in the same order as the CPU restatement's det_sincosf (tests' checker) (Cody-Waite reduction by pi/2 in three
// pieces, degree-7 / degree-8 polynomials on [-pi/4, pi/4]; this file is compiled with -ffp-contract=off): bit-identical to the CPU restatement,
// which the library sincosf (and glibc's on the other side) was not -- the tail / head choice flipped at near-ties of the curvature peaks.
__device__ __forceinline__ void det_sincosf(const float x, float& sn, float& cs) {
    const float kf = floorf(x * 0.636619772f + 0.5f);
    const int k = (int)kf;
    float r = x - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    const float z = r * r;
    float ps = -1.9515295891e-4f; ps = ps * z + 8.3321608736e-3f; ps = ps * z - 1.6666654611e-1f;
    const float s0 = r + r * z * ps;
    float pc = 2.443315711809948e-5f; pc = pc * z - 1.388731625493765e-3f; pc = pc * z + 4.166664568298827e-2f;
    const float c0 = (1.0f - 0.5f * z) + z * z * pc;
    const bool swap = k & 1;
    const float a = swap ? c0 : s0, b = swap ? s0 : c0;
    sn = (k & 2) ? -a : a;
    cs = ((k + 1) & 2) ? -b : b;
}

__device__ __forceinline__ bool in_blob(const uint32_t* s_runs, const int* s_row, int y0, int y1, int x, int y) {
    if (y < y0 || y > y1) return false;
    for (int i = s_row[y - y0]; i < s_row[y - y0 + 1]; ++i) {
        const uint32_t r = s_runs[i];
        if (x >= (int)(r & 0xffffu) && x <= (int)(r >> 16)) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_posture(const PostureCfg P, const trexhip_frame_info* __restrict__ info,
                                                 const uint32_t* __restrict__ blob_frame, const trexhip_blob* __restrict__ blobs,
                                                 const trexhip_run* __restrict__ runs, int n_blobs, int B,
                                                 float2* __restrict__ out_outline, float4* __restrict__ out_segments,
                                                 trexhip_posture_info* __restrict__ out_info,
                                                 const int32_t* __restrict__ sel, const trexhip_blob* __restrict__ origin) {
    extern __shared__ __attribute__((aligned(16))) uint8_t plds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bi = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // 4, 2 or 1 blobs per workgroup (LDS per blob)
    if (bi >= n_blobs) return;
    const int NPc = P.max_points;                           // point capacity = LDS layout stride
    uint8_t* base = plds + (size_t)wave * posture_wave_lds(NPc, P.nr_cap, P.rows_cap);
    float2* bufA = reinterpret_cast<float2*>(base);
    float2* bufB = bufA + NPc;
    float* s_curv = reinterpret_cast<float*>(bufB + NPc);
    float* s_t = s_curv + NPc;
    uint32_t* s_runs = reinterpret_cast<uint32_t*>(s_t + NPc);
    int* s_row = reinterpret_cast<int*>(s_runs + P.nr_cap);

    trexhip_posture_info res = {};
    const int cap = NPc;
    // sel: output slot bi takes blob sel[bi] of the table (negative: nothing to do for this slot); origin: the blob whose bounds().pos()
    // the coordinates are relative to (the ORIGINAL blob when a thresholded sub-blob is traced, Posture.cpp:336)
    const int si = sel ? sel[bi] : bi;
    if (si < 0) return;
    const uint32_t f = blob_frame[si];
    bool ok = f < (uint32_t)B;
    trexhip_frame_info fi = {};
    if (ok) { fi = info[f]; ok = fi.flags == 0; }
    if (!ok) { if (lane == 0) { res.status = 1; out_info[bi] = res; } return; }
    const trexhip_blob Bl = blobs[si];
    const int n_runs = (int)Bl.n_runs, y0 = Bl.y0, y1 = Bl.y1, rows = y1 - y0 + 1;
    if (n_runs == 0) { if (lane == 0) { res.status = 1; out_info[bi] = res; } return; }
    if (n_runs > P.nr_cap || rows > P.rows_cap) { if (lane == 0) { res.status = 2; out_info[bi] = res; } return; }
    const trexhip_run* rr = runs + fi.run_begin + Bl.run_begin;
    for (int i = lane; i < n_runs; i += 64) {
        const trexhip_run q = rr[i];
        s_runs[i] = (uint32_t)q.x0 | ((uint32_t)q.x1 << 16);
        if (i == 0 || rr[i - 1].y != q.y) s_row[q.y - y0] = i;
    }
    if (lane == 0) s_row[rows] = n_runs;
    __builtin_amdgcn_wave_barrier();
    const int ox = origin ? (int)origin[bi].x0 : (int)Bl.x0, oy = origin ? (int)origin[bi].y0 : (int)Bl.y0;   // coordinates relative to the blob's bounds().pos() (Posture.cpp:336)
    // blobs at most 64 pixels wide: one 64-bit occupancy word per row (in the curvature / arc-length arrays, idle until the
    // outline exists) makes the membership test of the boundary walk a single LDS read
    const int bx0 = Bl.x0;
    const bool use_bm = (int)Bl.x1 - (int)Bl.x0 < 64 && rows <= NPc;
    uint32_t* bm = reinterpret_cast<uint32_t*>(s_curv);
    if (use_bm) {
        for (int i = lane; i < rows * 2; i += 64) bm[i] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n_runs; i += 64) {
            const trexhip_run q = rr[i];
            const int a = (int)q.x0 - bx0, len = (int)q.x1 - (int)q.x0 + 1;
            const unsigned long long m = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << a;
            if ((uint32_t)m) atomicOr(&bm[2 * (q.y - y0)], (uint32_t)m);
            if ((uint32_t)(m >> 32)) atomicOr(&bm[2 * (q.y - y0) + 1], (uint32_t)(m >> 32));
        }
        __builtin_amdgcn_wave_barrier();
    }
    auto inside = [&](int x, int y) -> bool {
        if (!use_bm) return in_blob(s_runs, s_row, y0, y1, x, y);
        if (y < y0 || y > y1) return false;
        const int xr = x - bx0;
        if ((unsigned)xr >= 64u) return false;
        return (bm[2 * (y - y0) + (xr >> 5)] >> (xr & 31)) & 1u;
    };

    // ---- outline on the half-pixel lattice ----
    // The boundary is a permutation of directed pixel sides: side k of pixel p (k = 0 top, moving +x; 1 right, +y; 2 bottom, -x;
    // 3 left, -y; the blob on the right hand) is followed by the left-turn side of the pixel ahead-left if that pixel is set, else
    // the same side of the pixel ahead, else the next side of p itself -- exactly the turn rule of the sequential walk below.  All
    // sides are numbered (per row and direction: popcounts of bit masks), every lane computes successors, pointer jumping ranks
    // the cycle through the start side (top side of the first pixel), and each side emits its two points at its rank: ~100
    // dependent steps of one lane become ~10 short passes of the whole wave.  Needs the one-word-per-row bitmap, rows <= max_points/2
    // and at most max_points sides (holes included); anything else takes the sequential walk.
    int n = 0, status = 0;
    bool traced = false;
    if (use_bm && rows * 2 <= NPc) {
        uint16_t* base16 = reinterpret_cast<uint16_t*>(bufA);              // first side id per (row, direction)
        uint32_t* jmp0 = reinterpret_cast<uint32_t*>(bufB);                // next id | hops to the last side << 16, double buffered
        uint32_t* jmp1 = jmp0 + NPc;
        uint32_t* geo = reinterpret_cast<uint32_t*>(s_t);                  // x | row << 8 | direction << 16
        auto row64 = [&](int yr) -> unsigned long long {
            return (yr < 0 || yr >= rows) ? 0ull : ((unsigned long long)bm[2 * yr] | ((unsigned long long)bm[2 * yr + 1] << 32));
        };
        auto side_mask = [&](int yr, int k) -> unsigned long long {
            const unsigned long long R = row64(yr);
            return k == 0 ? R & ~row64(yr - 1) : (k == 1 ? R & ~(R >> 1) : (k == 2 ? R & ~row64(yr + 1) : R & ~(R << 1)));
        };
        const int items = rows * 4;
        uint32_t running = 0;
        for (int i0 = 0; i0 < items; i0 += 64) {
            const int it = i0 + lane;
            const uint32_t cnt = it < items ? (uint32_t)__popcll(side_mask(it >> 2, it & 3)) : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, d); if (lane >= d) incl += t; }
            if (it < items) base16[it] = (uint16_t)(running + incl - cnt);
            running += (uint32_t)__shfl((int)incl, 63);
        }
        const int E = (int)running;
        if (E <= NPc) {
            // one lane per SIDE (a lane per (row, direction) would walk up to 60 sides of a horizontal animal's top row while the others idle): the
            // (row, direction) item that owns side id e is the last one whose first id is <= e -- items write their number at their first id, a
            // running maximum fills the rest --, and the side is the (e - first id)-th set bit of the item's mask
            uint16_t* owner = reinterpret_cast<uint16_t*>(jmp1);               // free until the pointer jumping starts
            for (int e = lane; e < E; e += 64) owner[e] = 0;
            if (lane == 0) base16[items] = (uint16_t)E;
            __builtin_amdgcn_wave_barrier();
            for (int it = lane; it < items; it += 64) { const uint16_t b0 = base16[it]; if (base16[it + 1] != b0) owner[b0] = (uint16_t)it; }
            __builtin_amdgcn_wave_barrier();
            int carry = 0;
            for (int e0 = 0; e0 < E; e0 += 64) {
                const int e = e0 + lane;
                int it = e < E ? (int)owner[e] : 0;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(it, d); if (lane >= d && t > it) it = t; }
                if (carry > it) it = carry;
                carry = __shfl(it, 63);
                if (e >= E) continue;
                const int yr = it >> 2, k = it & 3;
                const unsigned long long m = side_mask(yr, k);
                int kk = e - (int)base16[it];                                       // the kk-th set bit of m
                uint32_t w = (uint32_t)m; int xr = 0;
                { const int c = __popc(w); if (kk >= c) { kk -= c; w = (uint32_t)(m >> 32); xr = 32; } }
                { const int c = __popc(w & 0xffffu); if (kk >= c) { kk -= c; w >>= 16; xr += 16; } }
                { const int c = __popc(w & 0xffu); if (kk >= c) { kk -= c; w >>= 8; xr += 8; } }
                { const int c = __popc(w & 0xfu); if (kk >= c) { kk -= c; w >>= 4; xr += 4; } }
                { const int c = __popc(w & 0x3u); if (kk >= c) { kk -= c; w >>= 2; xr += 2; } }
                if (kk >= (int)(w & 1u)) xr += 1;
                const int ddx = k == 0 ? 1 : (k == 2 ? -1 : 0), ddy = k == 1 ? 1 : (k == 3 ? -1 : 0);
                const int kl = (k + 3) & 3;
                const int lx = kl == 0 ? 1 : (kl == 2 ? -1 : 0), ly = kl == 1 ? 1 : (kl == 3 ? -1 : 0);
                int tx = xr + ddx + lx, ty = yr + ddy + ly, tk = kl;                     // ahead-left: turn left
                if (!((unsigned)tx < 64u && ((row64(ty) >> tx) & 1ull))) {
                    tx = xr + ddx; ty = yr + ddy; tk = k;                                // ahead: straight on
                    if (!((unsigned)tx < 64u && ((row64(ty) >> tx) & 1ull))) { tx = xr; ty = yr; tk = (k + 1) & 3; }   // turn right
                }
                const uint32_t tid = (uint32_t)base16[ty * 4 + tk] + (uint32_t)__popcll(side_mask(ty, tk) & ((1ull << tx) - 1ull));
                // the side whose successor is the start side (id 0: first top side of the first row) ends the cycle
                jmp0[e] = tid == 0u ? (uint32_t)e : (tid | (1u << 16));
                geo[e] = (uint32_t)xr | ((uint32_t)yr << 6) | ((uint32_t)k << 16);      // x < 64, row < 1024
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t* cur = jmp0; uint32_t* oth = jmp1;
            for (int span = 1; span < E; span <<= 1) {
                for (int e = lane; e < E; e += 64) {
                    const uint32_t v = cur[e], w = cur[v & 0xffffu];
                    oth[e] = (w & 0xffffu) | (((v >> 16) + (w >> 16)) << 16);
                }
                __builtin_amdgcn_wave_barrier();
                uint32_t* t = cur; cur = oth; oth = t;
            }
            const uint32_t first = cur[0];
            const int len = (int)(first >> 16) + 1;                                     // sides on the outer cycle
            const int nt = 2 * len;
            if (nt > NPc) { status = 2; res.n_traced = 0; }
            else {
                for (int e = lane; e < E; e += 64) {
                    const uint32_t v = cur[e];
                    if ((v & 0xffffu) != (first & 0xffffu)) continue;                    // a hole's boundary
                    const int pos = (int)(first >> 16) - (int)(v >> 16);
                    const uint32_t g = geo[e];
                    const int xr = (int)(g & 0x3fu), yr = (int)((g >> 6) & 0x3ffu), k = (int)(g >> 16);
                    const int vx = 2 * (xr + bx0 - ox) + ((k == 1 || k == 2) ? 1 : -1), vy = 2 * (yr + y0 - oy) + ((k >= 2) ? 1 : -1);   // doubled, relative to the origin
                    const int ddx = k == 0 ? 1 : (k == 2 ? -1 : 0), ddy = k == 1 ? 1 : (k == 3 ? -1 : 0);
                    bufA[2 * pos] = make_float2(0.5f * (float)vx, 0.5f * (float)vy);
                    bufA[2 * pos + 1] = make_float2(0.5f * (float)(vx + ddx), 0.5f * (float)(vy + ddy));
                }
                res.n_traced = nt;
            }
            traced = true;
        }
    }
    if (!traced && lane == 0) {
        const int fx = (int)(s_runs[0] & 0xffffu);
        const int sx = 2 * fx - 1, sy = 2 * y0 - 1;
        int vx = sx, vy = sy, dx = 1, dy = 0, nt = 0;
        do {
            if (nt + 2 > NPc) { status = 2; break; }
            bufA[nt++] = make_float2(0.5f * (float)(vx - 2 * ox), 0.5f * (float)(vy - 2 * oy));
            bufA[nt++] = make_float2(0.5f * (float)(vx + dx - 2 * ox), 0.5f * (float)(vy + dy - 2 * oy));
            vx += 2 * dx; vy += 2 * dy;
            const int lx = dy, ly = -dx, rx = -dy, ry = dx;
            if (inside((vx + dx + lx) / 2, (vy + dy + ly) / 2)) { dx = lx; dy = ly; }
            else if (inside((vx + dx + rx) / 2, (vy + dy + ry) / 2)) { }
            else { dx = rx; dy = ry; }
        } while (!(vx == sx && vy == sy && dx == 1 && dy == 0));
        res.n_traced = status ? 0 : nt;                 // an outline beyond the capacity reports no points at all
    }
    POSTURE_STOP(1);
    int nt_all = traced ? res.n_traced : __shfl(res.n_traced, 0);
    if (!traced) status = __shfl(status, 0);
    res.n_traced = nt_all;
    __builtin_amdgcn_wave_barrier();
    // Outline::resample (Outline.cpp:724-766).  Every segment of the lattice outline is exactly 0.5 long, so when 2 * resample
    // distance is a whole number k all of the reference's float arithmetic is exact and it emits the start point of every k-th
    // segment: done by all lanes.  Any other distance takes the sequential walk (same operation order as the reference).
    const float rd = P.outline_resample;
    const float k2 = rd * 2.0f;
    const bool lattice = rd > 0.f && k2 == floorf(k2) && k2 <= 1024.f;
    if (!status && lattice && nt_all > 1) {
        const int k = (int)k2;
        n = nt_all / k;
        if (n > cap) { status = 2; n = 0; }
        else {
            for (int jx = lane; jx < n; jx += 64) bufB[jx] = bufA[(jx + 1) * k - 1];
            if (n == 0) status = 1;
        }
    } else if (lane == 0) {
        const int nt = nt_all;
        if (!status) {
            if (rd <= 0.f || nt <= 1) { for (int i = 0; i < nt; ++i) bufB[i] = bufA[i]; n = nt; }
            else {
                float walked = 0.0f;
                for (int i = 0; i < nt && !status; ++i) {
                    int i1 = i + 1; if (i1 >= nt) i1 -= nt;
                    const float2 pt0 = bufA[i], pt1 = bufA[i1];
                    const float lxn = pt1.x - pt0.x, lyn = pt1.y - pt0.y;
                    const float len = sqrtf(lxn * lxn + lyn * lyn);
                    walked += len;
                    const float percent = len / rd;
                    float wp = walked / rd;
                    int offset = 0;
                    while (wp >= 1.0f) {
                        const float tt = (float)((double)offset * 1.0 / (double)percent);
                        if (n >= cap) { status = 2; break; }
                        bufB[n++] = make_float2(pt0.x + lxn * tt, pt0.y + lyn * tt);
                        offset++;
                        walked -= rd;
                        wp -= 1.0f;
                    }
                }
            }
            if (!status && n == 0) status = 1;
        }
    }
    if (!(lattice && nt_all > 1)) { n = __shfl(n, 0); status = __shfl(status, 0); }
    __builtin_amdgcn_wave_barrier();
    if (status) { if (lane == 0) { res.status = status; out_info[bi] = res; } return; }

    POSTURE_STOP(2);
    float2* pts = bufB; float2* other = bufA;
    // ---- smooth_outline (Outline.cpp:330-378): triangular weights over +-range*step ----
    if (P.smooth_samples > 0 && n > P.smooth_samples) {
        const float step_row = (float)P.smooth_samples * (float)P.smooth_step;
        // the normalised weights w[s] / sum once per blob (the curvature array is idle until the outline is final): lane s holds weight s, the sum
        // runs over them in the reference's order
        float* s_w = s_curv;
        int nw = 0; float sum = 0.f, mine = 0.f;
        for (int i = (int)-step_row; i <= step_row && nw < 33; i += P.smooth_step) { const float val = (step_row - fabsf((float)i)) / step_row; sum += val; if (nw == lane) mine = val; ++nw; }
        if (lane < nw) s_w[lane] = mine / sum;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += 64) {
            float2 pt = make_float2(0.f, 0.f); int s = 0;
            for (int j = (int)((float)i - step_row); (float)j <= (float)i + step_row; j += P.smooth_step) {
                int idx = j; while (idx < 0) idx += n; while (idx >= n) idx -= n;
                const float ww = s_w[s < 32 ? s : 32]; ++s;
                const float2 q = pts[idx];
                pt.x += q.x * ww; pt.y += q.y * ww;
            }
            other[i] = pt;
        }
        __builtin_amdgcn_wave_barrier();
        float2* t = pts; pts = other; other = t;
    }
    // ---- offset_to_middle: clockwise test ----
    {
        float part = 0.f;
        for (int i = lane; i < n; i += 64) { const float2 a = pts[i], q = pts[i + 1 == n ? 0 : i + 1]; part += a.x * q.y - q.x * a.y; }
        if (wsum(part) < 0.f) {
            for (int i = lane; i < n; i += 64) other[i] = pts[n - 1 - i];
            __builtin_amdgcn_wave_barrier();
            float2* t = pts; pts = other; other = t;
        }
    }
    // ---- EFT (order) -> inverse at n uniform parameters around the centre ----
    if (P.approximate > 0) {
        // the centre: Outline.cpp:502-505 adds the points IN ORDER, and a float sum is its order -- every lane walks the same sequence (LDS broadcasts)
        // (eight points per iteration: the loads of a block are independent of the running sums, only the adds are the chain)
        float cx = 0.f, cy = 0.f;
        int ic = 0;
        for (; ic + 8 <= n; ic += 8) {
            float2 a[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = pts[ic + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) { cx += a[k].x; cy += a[k].y; }
        }
        for (; ic < n; ++ic) { const float2 a = pts[ic]; cx += a.x; cy += a.y; }
        cx /= (float)n; cy /= (float)n;
        // cumulative arc length (wave scan over chunks of 64 segments + a running carry; the tests' CPU restatement follows exactly this order):
        // s_t[i] = arc length at the END of segment i
        float run = 0.f;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            float dt = 0.f;
            if (i < n) { const float2 a = pts[i], q = pts[i + 1 == n ? 0 : i + 1]; dt = sqrtf((q.x - a.x) * (q.x - a.x) + (q.y - a.y) * (q.y - a.y)); }
            float incl = dt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            if (i < n) s_t[i] = run + incl;
            run += __shfl(incl, 63);
        }
        __builtin_amdgcn_wave_barrier();
        const float T = s_t[n - 1];
        const float PI = 3.14159265358979323846f;
        float ca[4][4];                                           // [harmonic 1..3][a b c d]
        const int order = P.approximate;                          // 1 .. 15 (the launch refuses more); up to three harmonics: the unrolled form below
        // cos / sin of the FIRST harmonic's phase at every segment boundary once (boundary i = the start of segment i = the end of segment i - 1;
        // boundary n closes the curve: it goes to the idle curvature array), kept in the spare point buffer.  Harmonics 2 and 3 follow by the angle
        // addition formulas (c2 = c1 c1 - s1 s1, s2 = 2 (s1 c1), c3 = c2 c1 - s2 s1, s3 = s2 c1 + c2 s1): one sin / cos per boundary instead of
        // three -- the CPU restatement does the same operations in the same order
        for (int i = lane; i <= n; i += 64) {
            float sn, cs;
            det_sincosf(2.0f * PI * (float)1 * (i > 0 ? s_t[i - 1] : 0.f) / T, sn, cs);
            if (i < n) other[i] = make_float2(cs, sn);
            else { s_curv[0] = cs; s_curv[1] = sn; }
        }
        __builtin_amdgcn_wave_barrier();
        float* coef = s_curv + 2;                                 // more than three harmonics: [16][4] coefficients in the idle curvature array
        if (order > 3) {
            // `outline_approximate` is a uint8_t without an upper bound in the reference (core/default_config.cpp:888).  More than three harmonics
            // take this general form: the SAME sums per harmonic (term i to partial i % 64, xor butterfly) and the SAME recurrence for the phases
            // (h = 2 by the doubling form, h >= 3 from harmonic h - 1 by angle addition, always started at harmonic 1), three harmonics per sweep
            // over the segments -- a sweep recomputes the recurrence up to its first harmonic, which gives the values a single sweep would have had.
            const float csE = s_curv[0], snE = s_curv[1];
            for (int hb = 1; hb <= order; hb += 3) {
                float sa[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f}, sc[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
                const int hl = min(order, hb + 2);
                for (int i = lane; i < n; i += 64) {
                    const float2 a = pts[i], q = pts[i + 1 == n ? 0 : i + 1];
                    const float t0 = i > 0 ? s_t[i - 1] : 0.f, t1 = s_t[i];
                    const float dt = t1 - t0;
                    if (dt <= 0.f) continue;
                    const float2 c0 = other[i], c1 = (i + 1 < n) ? other[i + 1] : make_float2(csE, snE);
                    const float ddx = q.x - a.x, ddy = q.y - a.y;
                    const float gx = ddx / dt, gy = ddy / dt;
                    float c0h = c0.x, s0h = c0.y, c1h = c1.x, s1h = c1.y;
                    for (int h = 1; h <= hl; ++h) {
                        if (h > 1) {
                            float cn, sn2;
                            if (h == 2) { cn = c0.x * c0.x - c0.y * c0.y; sn2 = 2.0f * (c0.y * c0.x); } else { cn = c0h * c0.x - s0h * c0.y; sn2 = s0h * c0.x + c0h * c0.y; }
                            c0h = cn; s0h = sn2;
                            if (h == 2) { cn = c1.x * c1.x - c1.y * c1.y; sn2 = 2.0f * (c1.y * c1.x); } else { cn = c1h * c1.x - s1h * c1.y; sn2 = s1h * c1.x + c1h * c1.y; }
                            c1h = cn; s1h = sn2;
                        }
                        if (h >= hb) {
                            const float dc = c1h - c0h, ds = s1h - s0h;
                            const float pa = gx * dc, pb = gx * ds, pc = gy * dc, pd = gy * ds;
                            if (h == hb) { sa[0] += pa; sb[0] += pb; sc[0] += pc; sd[0] += pd; }
                            else if (h == hb + 1) { sa[1] += pa; sb[1] += pb; sc[1] += pc; sd[1] += pd; }
                            else { sa[2] += pa; sb[2] += pb; sc[2] += pc; sd[2] += pd; }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int h = hb + k;
                    if (h <= order) {
                        const float kf = T / (2.0f * (float)(h * h) * PI * PI);
                        const float va = kf * wsum(sa[k]), vb = kf * wsum(sb[k]), vc = kf * wsum(sc[k]), vd = kf * wsum(sd[k]);
                        if (lane == 0) { coef[4 * h] = va; coef[4 * h + 1] = vb; coef[4 * h + 2] = vc; coef[4 * h + 3] = vd; }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            const float csE = s_curv[0], snE = s_curv[1];
            // term i goes to partial sum i % 64 in order of i, the partials are combined by the xor butterfly (wsum): the order the CPU restatement follows too
            float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = lane; i < n; i += 64) {
                const float2 a = pts[i], q = pts[i + 1 == n ? 0 : i + 1];
                const float t0 = i > 0 ? s_t[i - 1] : 0.f, t1 = s_t[i];
                const float dt = t1 - t0;
                if (dt <= 0.f) continue;
                const float2 c0 = other[i], c1 = (i + 1 < n) ? other[i + 1] : make_float2(csE, snE);
                const float ddx = q.x - a.x, ddy = q.y - a.y;
                const float gx = ddx / dt, gy = ddy / dt;
                float c0h = c0.x, s0h = c0.y, c1h = c1.x, s1h = c1.y;
#pragma unroll
                for (int h = 1; h <= 3; ++h) {
                    if (h <= order) {
                        if (h > 1) {        // phase (h - 1) theta -> h theta at both boundaries
                            float cn, sn2;
                            if (h == 2) { cn = c0.x * c0.x - c0.y * c0.y; sn2 = 2.0f * (c0.y * c0.x); } else { cn = c0h * c0.x - s0h * c0.y; sn2 = s0h * c0.x + c0h * c0.y; }
                            c0h = cn; s0h = sn2;
                            if (h == 2) { cn = c1.x * c1.x - c1.y * c1.y; sn2 = 2.0f * (c1.y * c1.x); } else { cn = c1h * c1.x - s1h * c1.y; sn2 = s1h * c1.x + c1h * c1.y; }
                            c1h = cn; s1h = sn2;
                        }
                        const float dc = c1h - c0h, ds = s1h - s0h;
                        sa[h] += gx * dc; sb[h] += gx * ds; sc[h] += gy * dc; sd[h] += gy * ds;
                    }
                }
            }
#pragma unroll
            for (int h = 1; h <= 3; ++h) {
                if (h <= order) {
                    const float k = T / (2.0f * (float)(h * h) * PI * PI);
                    ca[h][0] = k * wsum(sa[h]); ca[h][1] = k * wsum(sb[h]); ca[h][2] = k * wsum(sc[h]); ca[h][3] = k * wsum(sd[h]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (order > 3) {
            for (int i = lane; i < n; i += 64) {
                const float tt = (float)i / (float)n;
                float x = cx, y = cy;
                float s1, c1;
                det_sincosf(2.0f * PI * (float)1 * tt, s1, c1);
                float ch = c1, sh = s1;
                for (int h = 1; h <= order; ++h) {
                    if (h > 1) {
                        float cn, sn2;
                        if (h == 2) { cn = c1 * c1 - s1 * s1; sn2 = 2.0f * (s1 * c1); } else { cn = ch * c1 - sh * s1; sn2 = sh * c1 + ch * s1; }
                        ch = cn; sh = sn2;
                    }
                    x += coef[4 * h] * ch + coef[4 * h + 1] * sh; y += coef[4 * h + 2] * ch + coef[4 * h + 3] * sh;
                }
                other[i] = make_float2(x, y);
            }
        } else
        for (int i = lane; i < n; i += 64) {
            const float tt = (float)i / (float)n;
            float x = cx, y = cy;
            float s1, c1;
            det_sincosf(2.0f * PI * (float)1 * tt, s1, c1);
            float ch = c1, sh = s1;
#pragma unroll
            for (int h = 1; h <= 3; ++h) {
                if (h <= order) {
                    if (h > 1) {
                        float cn, sn2;
                        if (h == 2) { cn = c1 * c1 - s1 * s1; sn2 = 2.0f * (s1 * c1); } else { cn = ch * c1 - sh * s1; sn2 = sh * c1 + ch * s1; }
                        ch = cn; sh = sn2;
                    }
                    x += ca[h][0] * ch + ca[h][1] * sh; y += ca[h][2] * ch + ca[h][3] * sh;
                }
            }
            other[i] = make_float2(x, y);
        }
        __builtin_amdgcn_wave_barrier();
        float2* t = pts; pts = other; other = t;
    }
    POSTURE_STOP(3);
    // ---- curvature, tail = highest peak, head = farthest peak ----
    int r = (int)(P.curvature_range_ratio * (float)n); if (r < 1) r = 1;
    for (int i = lane; i < n; i += 64) {
        // (i -+ r) mod n; r < n except for range ratios >= 1
        const int im = r < n ? (i - r < 0 ? i - r + n : i - r) : ((i - r) % n + n) % n, ip = r < n ? (i + r >= n ? i + r - n : i + r) : (i + r) % n;
        const float2 p1 = pts[im], p2 = pts[i], p3 = pts[ip];
        const float cr = (p2.x - p1.x) * (p3.y - p2.y) - (p2.y - p1.y) * (p3.x - p2.x);
        const float d12 = (p2.x - p1.x) * (p2.x - p1.x) + (p2.y - p1.y) * (p2.y - p1.y);
        const float d23 = (p3.x - p2.x) * (p3.x - p2.x) + (p3.y - p2.y) * (p3.y - p2.y);
        const float d13 = (p3.x - p1.x) * (p3.x - p1.x) + (p3.y - p1.y) * (p3.y - p1.y);
        const float den = sqrtf(d12 * d23 * d13);
        s_curv[i] = den > 0.f ? fabsf(2.0f * cr / den) : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    float best = -1.f; int tail = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float c0 = s_curv[i == 0 ? n - 1 : i - 1], c1 = s_curv[i], c2 = s_curv[i + 1 == n ? 0 : i + 1];
        if (c1 > c0 && c1 >= c2 && c1 > best) { best = c1; tail = i; }        // per lane: first index of its maximum
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d); const int ot = __shfl_xor(tail, d);
        if (ob > best || (ob == best && ot < tail)) { best = ob; tail = ot; }
    }
    if (tail == 0x7fffffff || best < 0.f) {     // no curvature peak: the outline stays available (first_outline fallback, Posture.cpp:361-368)
        float2* of = out_outline + (size_t)bi * P.max_points;
        for (int i = lane; i < n; i += 64) of[i] = pts[i];
        if (lane == 0) { res.status = 3; res.n_outline = n; out_info[bi] = res; }
        return;
    }
    float maxd = 0.f; int head = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float c0 = s_curv[i == 0 ? n - 1 : i - 1], c1 = s_curv[i], c2 = s_curv[i + 1 == n ? 0 : i + 1];
        if (!(c1 > c0 && c1 >= c2)) continue;
        float dd;
        if (i >= tail) dd = fminf(fabsf((float)(i - tail)), fabsf((float)(i - tail - n)));
        else dd = fminf(fabsf((float)(tail - i)), fabsf((float)(tail - i - n)));
        if (dd > maxd) { maxd = dd; head = i; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float od = __shfl_xor(maxd, d); const int oh = __shfl_xor(head, d);
        if (od > maxd || (od == maxd && oh < head)) { maxd = od; head = oh; }
    }
    // rotate so that the tail is point 0 (Outline.cpp:707); this is the outline the caller gets
    float2* oo = out_outline + (size_t)bi * P.max_points;
    for (int i = lane; i < n; i += 64) { const float2 v = pts[i + tail >= n ? i + tail - n : i + tail]; other[i] = v; oo[i] = v; }
    __builtin_amdgcn_wave_barrier();
    { float2* t = pts; pts = other; other = t; }
    res.n_outline = n; res.tail_index = 0;
    res.head_index = head == 0x7fffffff ? -1 : ((head - tail) % n + n) % n;
    if (n <= 1) { if (lane == 0) { res.status = 1; out_info[bi] = res; } return; }
    POSTURE_STOP(4);
    {
        // the walk keeps 3 .. a few lanes of a wave busy (max_offset candidates per search) and is bound by instruction issue: blobs whose
        // searches fit a small lane group are finished by k_posture_walk, several blobs per wave (n_segments = -1 marks them)
        float mo0 = P.midline_walk_offset * (float)n; if (mo0 < 3.0f) mo0 = 3.0f;
        const int m0 = (int)mo0;      // tag -G / 8 of the smallest launched group that holds the searches
        const int tag = ((P.walk_group & 8) && m0 <= 8) ? -1 : (((P.walk_group & 16) && m0 <= 16) ? -2 : 0);
        if (tag) {
            if (lane == 0) { res.n_segments = tag; res.status = 0; out_info[bi] = res; }
            return;
        }
    }
    // ---- the two-pointer walk (Outline.cpp:790-857): control flow is wave-uniform, the max_offset candidates of each
    // search are evaluated one per lane and reduced to the FIRST minimum (the sequential `len < min_d` rule).  The kernel is bound
    // by instruction issue, so the loop only keeps what the next iteration depends on: the pair of outline indices of every
    // segment goes to LDS and the segments themselves (two square roots each) are computed by all lanes afterwards. ----
    {
        const int L = n;
        int idx_r = 1, idx_l = -1, ns = 0;
        float mo = P.midline_walk_offset * (float)L; if (mo < 3.0f) mo = 3.0f;
        const int max_offset = (int)mo;
        float4* so = out_segments + (size_t)bi * (P.max_points / 2 + 1);
        const float BIG = 3.402823466e38f;
        uint32_t* s_pair = reinterpret_cast<uint32_t*>(s_curv);          // curvature is done: right index (0xffff = none) | left index << 16
        const int seg_cap = P.max_points / 2 + 1;
        auto walk = [&](auto REDC) {
            constexpr int RED = decltype(REDC)::value;                   // lanes taking part in one search (power of two >= max_offset)
            while (idx_r < L + idx_l) {
                // the three LDS reads of an iteration depend only on the indices of the previous one: one round trip, not four
                const bool cand_r = lane < max_offset && idx_r + lane < L, cand_l = lane < max_offset && idx_l - lane > -L;
                const float2 pl0 = pts[L + idx_l];
                const float2 cr = cand_r ? pts[idx_r + lane] : make_float2(0.f, 0.f);
                const float2 cl = cand_l ? pts[L + idx_l - lane] : make_float2(0.f, 0.f);
                float len = BIG; int idx = 0x7fffffff;
                if (cand_r) {
                    const float ddx = cr.x - pl0.x, ddy = cr.y - pl0.y;
                    len = sqrtf(ddx * ddx + ddy * ddy); idx = idx_r + lane;
                    if (!(len < BIG)) { len = BIG; idx = 0x7fffffff; }
                }
#pragma unroll
                for (int d = 1; d < RED; d <<= 1) {
                    const float ol = __shfl_xor(len, d); const int oi = __shfl_xor(idx, d);
                    if (ol < len || (ol == len && oi < idx)) { len = ol; idx = oi; }
                }
                // the winner's point comes out of its lane's registers (v_readlane), and the state stays wave-uniform
                float2 pt_r = make_float2(0.f, 0.f);
                uint32_t used_r = 0xffffu;
                const int idx0 = __builtin_amdgcn_readfirstlane(idx);
                if (idx0 != 0x7fffffff) {
                    const int wl = idx0 - __builtin_amdgcn_readfirstlane(idx_r);
                    pt_r.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cr.x), wl));
                    pt_r.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cr.y), wl));
                    idx_r = idx0; used_r = (uint32_t)idx0;
                }
                float len2 = BIG; int key = 0x7fffffff;
                if (cand_l) {
                    const float ddx = pt_r.x - cl.x, ddy = pt_r.y - cl.y;
                    len2 = sqrtf(ddx * ddx + ddy * ddy); key = lane;
                    if (!(len2 < BIG)) { len2 = BIG; key = 0x7fffffff; }
                }
#pragma unroll
                for (int d = 1; d < RED; d <<= 1) {
                    const float ol = __shfl_xor(len2, d); const int ok2 = __shfl_xor(key, d);
                    if (ol < len2 || (ol == len2 && ok2 < key)) { len2 = ol; key = ok2; }
                }
                const int key0 = __builtin_amdgcn_readfirstlane(key);
                if (key0 != 0x7fffffff) idx_l -= key0;
                if (lane == 0 && ns < seg_cap) s_pair[ns] = used_r | ((uint32_t)(L + idx_l) << 16);
                ++ns;
                idx_r++; idx_l--;
            }
        };
        if (max_offset <= 64) {
            if (max_offset <= 4) walk(std::integral_constant<int, 4>{});
            else if (max_offset <= 8) walk(std::integral_constant<int, 8>{});
            else if (max_offset <= 16) walk(std::integral_constant<int, 16>{});
            else walk(std::integral_constant<int, 64>{});
            __builtin_amdgcn_wave_barrier();
            const int nseg = ns < seg_cap ? ns : seg_cap;
            for (int i = lane; i < nseg; i += 64) {
                const uint32_t pr = s_pair[i];
                const float2 pt_r = (pr & 0xffffu) == 0xffffu ? make_float2(0.f, 0.f) : pts[pr & 0xffffu];
                const float2 pt_l = pts[pr >> 16];
                const float lx = pt_r.x - pt_l.x, ly = pt_r.y - pt_l.y;
                const float mx = pt_l.x + lx * 0.5f, my = pt_l.y + ly * 0.5f;
                so[i] = make_float4(mx, my, sqrtf(lx * lx + ly * ly), sqrtf((mx - pt_l.x) * (mx - pt_l.x) + (my - pt_l.y) * (my - pt_l.y)));
            }
        } else {
        // more candidates than lanes (midline_walk_offset * points > 64): several passes per search
        int red = 64;
        while (idx_r < L + idx_l) {
            float2 pt_r = make_float2(0.f, 0.f); float2 pt_l = pts[L + idx_l];
            int min_idx = -1; float best1 = BIG;
            for (int i0 = 0; i0 < max_offset; i0 += 64) {
                const int i = i0 + lane;
                float len = BIG; int idx = 0x7fffffff;
                if (i < max_offset && idx_r + i < L) {
                    const float2 pt = pts[idx_r + i];
                    const float ddx = pt.x - pt_l.x, ddy = pt.y - pt_l.y;
                    len = sqrtf(ddx * ddx + ddy * ddy); idx = idx_r + i;
                    if (!(len < BIG)) { len = BIG; idx = 0x7fffffff; }
                }
                float bl = len; int bidx = idx;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const float ol = __shfl_xor(bl, d); const int oi = __shfl_xor(bidx, d); if (ol < bl || (ol == bl && oi < bidx)) { bl = ol; bidx = oi; } }
                if (bidx != 0x7fffffff && bl < best1) { best1 = bl; min_idx = bidx; }    // strict `<`: earlier candidates keep ties
            }
            if (min_idx != -1) { pt_r = pts[min_idx]; idx_r = min_idx; }
            int min_idx2 = 1; float best2 = BIG;
            for (int i0 = 0; i0 < max_offset; i0 += 64) {
                const int i = i0 + lane;
                float len = BIG; int key = 0x7fffffff;
                if (i < max_offset && idx_l - i > -L) {
                    const float2 pt = pts[L + idx_l - i];
                    const float ddx = pt_r.x - pt.x, ddy = pt_r.y - pt.y;
                    len = sqrtf(ddx * ddx + ddy * ddy); key = i;
                    if (!(len < BIG)) { len = BIG; key = 0x7fffffff; }
                }
                float bl = len; int bk = key;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const float ol = __shfl_xor(bl, d); const int ok2 = __shfl_xor(bk, d); if (ol < bl || (ol == bl && ok2 < bk)) { bl = ol; bk = ok2; } }
                if (bk != 0x7fffffff && bl < best2) { best2 = bl; min_idx2 = idx_l - bk; }
            }
            if (min_idx2 != 1) { pt_l = pts[L + min_idx2]; idx_l = min_idx2; }
            const float lx = pt_r.x - pt_l.x, ly = pt_r.y - pt_l.y;
            const float mx = pt_l.x + lx * 0.5f, my = pt_l.y + ly * 0.5f;
            if (lane == 0 && ns <= P.max_points / 2)
                so[ns] = make_float4(mx, my, sqrtf(lx * lx + ly * ly), sqrtf((mx - pt_l.x) * (mx - pt_l.x) + (my - pt_l.y) * (my - pt_l.y)));
            ++ns;
            idx_r++; idx_l--;
        }
        (void)red;
        }
        if (lane == 0) {
            res.n_segments = ns;
            res.status = ns <= 2 ? 4 : 0;
            out_info[bi] = res;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_posture_walk: the two-pointer walk of k_posture (Outline.cpp:790-857) for the blobs it tagged (n_segments == -G / 8), 64 / G blobs per wave.
// The walk is a chain of ~40 dependent steps per blob with 3 .. a few candidates per search: one blob per wave leaves the vector unit nearly
// empty and the phase bound by instruction issue.  Here a group of G lanes owns a blob -- its outline (written by k_posture, tail at point 0)
// in the group's LDS, the max_offset <= G candidates of a search one per lane -- and a step costs the wave ~100 instructions for 64 / G blobs:
// the minimum distance over the group is three or four v_min_u32 with a DPP operand (distances are >= +0, their bit patterns order like the
// values), the FIRST minimum (the sequential `len < min_d` rule) is the lowest set bit of the group's part of a ballot, and the winning point
// is read back from LDS.  The same float operations in the same order as the in-kernel walk; groups whose blobs take fewer steps idle.
// ------------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ uint32_t dpp_min_u32(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)v, CTRL, 0xf, 0xf, false);
    return o < v ? o : v;
}
template <int G> __device__ __forceinline__ uint32_t group_min_u32(uint32_t v) {
    v = dpp_min_u32<0xB1>(v);                              // quad_perm [1,0,3,2]
    v = dpp_min_u32<0x4E>(v);                              // quad_perm [2,3,0,1]
    if (G >= 8) v = dpp_min_u32<0x141>(v);                 // row_half_mirror: the other quad of the 8
    if (G >= 16) v = dpp_min_u32<0x140>(v);                // row_mirror: the other half of the 16
    return v;
}
template <int G>
__global__ __launch_bounds__(64) void k_posture_walk(const float walk_offset, const int max_points, const int n_blobs, const float2* __restrict__ outline,
                                                     float4* __restrict__ out_segments, trexhip_posture_info* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wlds[];
    constexpr int BPW = 64 / G;
    const int lane = threadIdx.x, g = lane % G, grp = lane / G, gbase = lane - g;
    const int bi = blockIdx.x * BPW + grp;
    const int seg_cap = max_points / 2 + 1;
    const size_t per = (size_t)max_points * 8 + (((size_t)seg_cap * 4 + 7) & ~(size_t)7);
    float2* pts = reinterpret_cast<float2*>(wlds + (size_t)grp * per);
    uint32_t* s_pair = reinterpret_cast<uint32_t*>(pts + max_points);
    int L = 0;                                             // 0: not this kernel's blob (every lane stays, the wave loads the outlines together)
    if (bi < n_blobs) {
        const trexhip_posture_info pi = info[bi];
        if (pi.status == 0 && pi.n_segments == -(G / 8)) L = pi.n_outline;
    }
    // ---- the outlines: the whole wave loads one blob's points at a time, the first 128 points of all blobs in flight together
    {
        float2 v0[BPW], v1[BPW];
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int Lb = __builtin_amdgcn_readlane(L, b * G);
            const float2* src = outline + (size_t)(blockIdx.x * BPW + b) * max_points;
            v0[b] = lane < Lb ? src[lane] : make_float2(0.f, 0.f);
            v1[b] = lane + 64 < Lb ? src[lane + 64] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int Lb = __builtin_amdgcn_readlane(L, b * G);
            float2* dst = reinterpret_cast<float2*>(wlds + (size_t)b * per);
            if (lane < Lb) dst[lane] = v0[b];
            if (lane + 64 < Lb) dst[lane + 64] = v1[b];
            if (Lb > 128) {
                const float2* src = outline + (size_t)(blockIdx.x * BPW + b) * max_points;
                for (int i = 128 + lane; i < Lb; i += 64) dst[i] = src[i];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                       // one wave: its LDS operations stay in order
    int idx_r = 1, idx_l = -1, ns = 0;
    float mo = walk_offset * (float)L; if (mo < 3.0f) mo = 3.0f;
    const int max_offset = (int)mo;                        // <= G (k_posture's own test)
    const uint32_t BIGU = 0x7f7fffffu;                     // FLT_MAX: "no candidate"
    const bool in_search = g < max_offset;
    while (idx_r < L + idx_l) {
        const int il = L + idx_l;
        const bool cand_r = in_search && idx_r + g < L, cand_l = in_search && il - g > 0;
        const float2 pl0 = pts[il];
        const float2 cr = pts[cand_r ? idx_r + g : 0];
        const float2 cl = pts[cand_l ? il - g : 0];
        uint32_t lb = BIGU;
        {
            const float ddx = cr.x - pl0.x, ddy = cr.y - pl0.y;
            const uint32_t b = __builtin_bit_cast(uint32_t, sqrtf(ddx * ddx + ddy * ddy));
            if (cand_r && b < BIGU) lb = b;                // NaN / inf patterns are above FLT_MAX: !(len < FLT_MAX) never wins
        }
        const uint32_t m1 = group_min_u32<G>(lb);
        const uint64_t w1 = __ballot(lb == m1 && lb != BIGU);
        const uint32_t g1 = (uint32_t)(w1 >> gbase) & ((1u << G) - 1u);
        float2 pt_r = make_float2(0.f, 0.f);
        uint32_t used_r = 0xffffu;
        if (g1) { idx_r += __builtin_ctz(g1); used_r = (uint32_t)idx_r; pt_r = pts[idx_r]; }   // the same in every lane of the group
        uint32_t lb2 = BIGU;
        {
            const float ddx = pt_r.x - cl.x, ddy = pt_r.y - cl.y;
            const uint32_t b = __builtin_bit_cast(uint32_t, sqrtf(ddx * ddx + ddy * ddy));
            if (cand_l && b < BIGU) lb2 = b;
        }
        const uint32_t m2 = group_min_u32<G>(lb2);
        const uint64_t w2 = __ballot(lb2 == m2 && lb2 != BIGU);
        const uint32_t g2 = (uint32_t)(w2 >> gbase) & ((1u << G) - 1u);
        if (g2) idx_l -= __builtin_ctz(g2);
        if (g == 0 && ns < seg_cap) s_pair[ns] = used_r | ((uint32_t)(L + idx_l) << 16);
        ++ns;
        idx_r++; idx_l--;
    }
    if (L == 0) return;
    __builtin_amdgcn_wave_barrier();
    float4* so = out_segments + (size_t)bi * seg_cap;
    const int nseg = ns < seg_cap ? ns : seg_cap;
    for (int i = g; i < nseg; i += G) {
        const uint32_t pr = s_pair[i];
        const float2 pt_r = (pr & 0xffffu) == 0xffffu ? make_float2(0.f, 0.f) : pts[pr & 0xffffu];
        const float2 pt_l = pts[pr >> 16];
        const float lx = pt_r.x - pt_l.x, ly = pt_r.y - pt_l.y;
        const float mx = pt_l.x + lx * 0.5f, my = pt_l.y + ly * 0.5f;
        so[i] = make_float4(mx, my, sqrtf(lx * lx + ly * ly), sqrtf((mx - pt_l.x) * (mx - pt_l.x) + (my - pt_l.y) * (my - pt_l.y)));
    }
    if (g == 0) { info[bi].n_segments = ns; info[bi].status = ns <= 2 ? 4 : 0; }
}

// which lane groups of k_posture_walk a launch uses (bits 8, 16): groups of 8 take the blobs whose searches fit them (twice the blobs per wave),
// groups of 16 the rest up to 16 candidates and are not launched when the parameters cannot need them; searches of more than 16 candidates stay
// inside k_posture, and so does everything when the outlines of a wave's blobs do not fit 64 KB of LDS.  (Groups of 4 -- 16 blobs per wave, 40 KB
// of LDS at 256 points -- leave one wave per SIMD and measured slower than groups of 8: 62 vs 50 us per 25 600 blobs.)
static size_t posture_walk_lds(int max_points) { return (size_t)max_points * 8 + ((((size_t)max_points / 2 + 1) * 4 + 7) & ~(size_t)7); }
static int posture_walk_groups(const trexhip_posture_params* pp) {
    float mo = pp->midline_walk_offset * (float)pp->max_points; if (mo < 3.0f) mo = 3.0f;
    const size_t per = posture_walk_lds(pp->max_points);
    const bool a8 = per * 8 <= 64 * 1024, a16 = per * 4 <= 64 * 1024;
    return (a8 ? 8 : 0) | (a16 && ((int)mo > 8 || !a8) ? 16 : 0);
}
static void launch_posture_walk(hipStream_t s, const trexhip_posture_params* pp, int groups, int n_blobs, float* d_outline, float* d_segments, trexhip_posture_info* d_info) {
    if (n_blobs <= 0) return;
    const size_t per = posture_walk_lds(pp->max_points);
#define PW_LAUNCH(G_) do { const int bpw = 64 / G_; const size_t lds = per * (size_t)bpw; \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_posture_walk<G_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_posture_walk<G_>), dim3((unsigned)((n_blobs + bpw - 1) / bpw)), dim3(64), lds, s, pp->midline_walk_offset, pp->max_points, n_blobs, \
                           reinterpret_cast<const float2*>(d_outline), reinterpret_cast<float4*>(d_segments), d_info); } while (0)
    if (groups & 8) PW_LAUNCH(8);
    if (groups & 16) PW_LAUNCH(16);
#undef PW_LAUNCH
}

}  // namespace trexhip

using namespace trexhip;

extern "C" int trexhip_posture_device(trexhip_ctx* ctx, int32_t table, const trexhip_posture_params* pp, int32_t n_blobs,
                                      float* d_outline, float* d_segments, trexhip_posture_info* d_info) {
    if (!ctx || !pp || !d_outline || !d_segments || !d_info) { set_error("trexhip_posture_device: null argument"); return TREXHIP_E_INVALID; }
    if (pp->max_points < 8 || pp->max_points > P_NP || (pp->max_points & 1)) { set_error("trexhip_posture_device: max_points must be even and in 8..4096"); return TREXHIP_E_INVALID; }
    if (pp->outline_smooth_samples < 0 || pp->outline_smooth_samples * (pp->outline_smooth_step > 0 ? pp->outline_smooth_step : 1) > 16 || pp->outline_smooth_step < 1) {
        set_error("trexhip_posture_device: outline_smooth_samples*outline_smooth_step must be <= 16"); return TREXHIP_E_UNSUPPORTED;
    }
    // (outline_approximate is a uint8_t without an upper bound in the reference, core/default_config.cpp:888: up to 15 harmonics here; beyond three the
    // general form of k_posture, whose [16][4] coefficients live in the curvature array -- 66 floats)
    if (pp->outline_approximate > 15 || (pp->outline_approximate > 3 && pp->max_points < 128)) {
        set_error("trexhip_posture_device: outline_approximate > 15 (or > 3 with max_points < 128) is not supported"); return TREXHIP_E_UNSUPPORTED;
    }
    // settings whose arithmetic is not built (it lives in the un-vendored commons and nothing in the tree pins it): refuse
    if (pp->posture_closing_steps != 0) { set_error("trexhip_posture_device: posture_closing_steps > 0 is not implemented (closing inside pixel::threshold_get_biggest_blob, Posture.cpp:335)"); return TREXHIP_E_UNSUPPORTED; }
    if (pp->peak_mode != 0) { set_error("trexhip_posture_device: peak_mode = broad is not implemented (needs periodic::find_peaks' peak ranges / integrals, Outline.cpp:627-661); only pointy"); return TREXHIP_E_UNSUPPORTED; }
    if (pp->posture_direction_smoothing < 0) { set_error("trexhip_posture_device: posture_direction_smoothing must not be negative"); return TREXHIP_E_INVALID; }   // (> 1: the caller hands the movement direction to trexhip_midline_movement_device)
    if (!ctx->d_frames || ctx->last_n == 0 || !ctx->fetched) { set_error("trexhip_posture_device: segment and fetch a batch first"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_posture_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    const trexhip_frame_info* info = ctx->d_info; const uint32_t* bf = ctx->d_blob_frame; const trexhip_blob* bl = ctx->d_blobs; const trexhip_run* ru = ctx->d_runs;
    if (table == 1) {
        if (!ctx->pass2.allocated || ctx->pass2.valid_n == 0) { set_error("trexhip_posture_device: no re-thresholded batch"); return TREXHIP_E_INVALID; }
        info = ctx->pass2.d_info; bf = ctx->pass2.d_blob_frame; bl = ctx->pass2.d_blobs; ru = ctx->pass2.d_runs;
    } else if (table != 0) { set_error("trexhip_posture_device: table must be 0 (detect) or 1 (re-threshold)"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    // line / row capacity of this launch from the fetched host tables (full capacity when they are not at hand)
    int nr_cap = P_NR, rows_cap = P_ROWS;
    {
        const trexhip_blob* hb = table == 0 ? ctx->h_blobs : ctx->pass2.h_blobs;
        const bool fetched = table == 0 ? ctx->fetched : ctx->pass2.fetched;
        if (hb && fetched) {
            uint32_t mr = 1, mrows = 1;
            for (int i = 0; i < n_blobs; ++i) {
                const trexhip_blob& b = hb[i];
                if (b.n_runs <= (uint32_t)P_NR && (uint32_t)(b.y1 - b.y0 + 1) <= (uint32_t)P_ROWS) {   // larger ones report status 2 anyway
                    mr = b.n_runs > mr ? b.n_runs : mr;
                    const uint32_t rws = (uint32_t)(b.y1 - b.y0 + 1);
                    mrows = rws > mrows ? rws : mrows;
                }
            }
            nr_cap = (int)((mr + 31u) & ~31u); rows_cap = (int)((mrows + 29u) / 32u * 32u + 30u);      // (rows_cap + 2) * 4 stays 16-byte friendly
            if (nr_cap > P_NR) nr_cap = P_NR;
            if (rows_cap > P_ROWS) rows_cap = P_ROWS;
        }
    }
    // blobs per workgroup: 4 while their LDS fits a CU comfortably, else 2 or 1 (large max_points / large blobs)
    const int wave_lds = posture_wave_lds(pp->max_points, nr_cap, rows_cap);
    int wpb = 4;
    while (wpb > 1 && wpb * wave_lds > 150 * 1024) wpb >>= 1;
    const int lds_bytes = wpb * wave_lds;
    if (lds_bytes > ctx->attr_posture_bytes) {
        TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_posture), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        ctx->attr_posture_bytes = lds_bytes;
    }
    PostureCfg P{pp->outline_resample, pp->outline_smooth_samples, pp->outline_smooth_step, pp->outline_approximate,
                 pp->outline_curvature_range_ratio, pp->midline_walk_offset, pp->max_points, nr_cap, rows_cap, 0, posture_walk_groups(pp)};
#ifdef TREXHIP_DEV_KNOBS
    if (const char* e = std::getenv("TREXHIP_POSTURE_STOP")) P.stop = std::atoi(e);
    if (const char* e = std::getenv("TREXHIP_POSTURE_WALK_GROUP")) P.walk_group = std::atoi(e);
#endif
    stage_begin(ctx, TREXHIP_STAGE_POSTURE);
    hipLaunchKernelGGL(k_posture, dim3((n_blobs + wpb - 1) / wpb), dim3(wpb * 64), lds_bytes, ctx->stream, P, info, bf, bl, ru, n_blobs, ctx->last_n,
                       reinterpret_cast<float2*>(d_outline), reinterpret_cast<float4*>(d_segments), d_info, (const int32_t*)nullptr, (const trexhip_blob*)nullptr);
    launch_posture_walk(ctx->stream, pp, P.walk_group, n_blobs, d_outline, d_segments, d_info);
    stage_end(ctx, TREXHIP_STAGE_POSTURE);
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

// ---- posture::calculate_posture with its retry loop (Posture.cpp:305-399) ---------------------------------------------------------
// Per detect blob: threshold = track_posture_threshold; repeat { biggest sub-blob at `threshold` (pixel::threshold_get_biggest_blob:
// re-threshold + largest pixel count, the first wins a tie) -> outline relative to the ORIGINAL blob -> resample -> midline; success
// ends the loop; else remember the first outline, threshold += 2, stop when that sub-blob had fewer than max(1, initial / 10) pixels or
// threshold >= track_posture_threshold + 100 }; without success the first outline is returned without a midline (:383-391).
// On the device every round is: one per-blob re-threshold pass over the still active blobs (trexhip_rethreshold_per_blob_device),
// k_auto_pick / k_auto_sel (biggest sub-blob per detect blob by a 64-bit atomic max), k_posture on the selected sub-blobs, k_auto_step
// (one wave per blob: bookkeeping, first-outline copy); the host only reads the number of blobs still active.
namespace trexhip {
__global__ void k_auto_init(const int n, const int tpt, int32_t* thr, int32_t* first_n, int32_t* first_thr, int32_t* used, int32_t* iters,
                            unsigned long long* best, trexhip_posture_info* info) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    thr[b] = tpt; first_n[b] = 0; first_thr[b] = -1; used[b] = -1; iters[b] = 0; best[b] = 0ull;
    trexhip_posture_info z = {}; z.status = 1; info[b] = z;
}
__global__ void k_auto_pick(const trexhip_blob* __restrict__ sub, const uint32_t* __restrict__ totals2, const uint32_t pool, const int n,
                            unsigned long long* best) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const uint32_t tot = totals2[0] < pool ? totals2[0] : pool;
    if (s >= tot) return;
    const trexhip_blob B = sub[s];
    if (B.parent >= (uint32_t)n || B.n_pixels == 0) return;                       // holes of frames that overflowed carry no parent
    atomicMax(best + B.parent, ((unsigned long long)B.n_pixels << 32) | (unsigned long long)(0xffffffffu - s));   // most pixels, then the lowest index
}
__global__ void k_auto_sel(const int n, const int32_t* thr, const unsigned long long* best, int32_t* sel) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    sel[b] = (thr[b] >= 0 && best[b] != 0ull) ? (int32_t)(0xffffffffu - (uint32_t)best[b]) : -1;
}
__global__ __launch_bounds__(256) void k_auto_step(const int n, const int tpt, const int max_points, const trexhip_blob* __restrict__ detect,
                                                   int32_t* thr, int32_t* sel, unsigned long long* best, int32_t* first_n, int32_t* first_thr,
                                                   int32_t* used, int32_t* iters, float2* first, float2* outline, trexhip_posture_info* info,
                                                   uint32_t* active, const trexhip_blob* __restrict__ sub, const int nr_cap, const int rows_cap) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const int t = thr[b];
    if (t < 0) return;
    const int s = sel[b];
    if (s == -2) return;                                   // already handled in this round (the round is being repeated with full LDS capacities)
    const uint32_t count = s >= 0 ? (uint32_t)(best[b] >> 32) : 0u;
    trexhip_posture_info last = info[b];
    if (s < 0) { trexhip_posture_info z = {}; z.status = 1; last = z; }
    if (s >= 0 && last.status == 2 && (nr_cap < P_NR || rows_cap < P_ROWS)) {
        // the launch's LDS capacities are an ESTIMATE from the parent blobs (a thresholded line can split into more than two); a sub-blob
        // beyond the estimate but within the kernel's real limits is not a failed attempt: leave the blob as it is and ask for a repeat
        const int sr = (int)sub[s].n_runs, srows = sub[s].y1 - sub[s].y0 + 1;
        if ((sr > nr_cap || srows > rows_cap) && sr <= P_NR && srows <= P_ROWS) {
            if (lane == 0) atomicAdd(active + 1, 1u);
            return;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { iters[b] += 1; best[b] = 0ull; sel[b] = -2; }
    if (s >= 0 && last.status == 0) { if (lane == 0) { used[b] = t; thr[b] = -1; } return; }      // a midline at the lowest possible threshold
    float2* mine = outline + (size_t)b * max_points;
    float2* keep = first + (size_t)b * max_points;
    int fn = first_n[b];
    if (s >= 0 && fn == 0 && last.n_outline > 0) {                                    // the first outline that could be traced
        for (int i = lane; i < last.n_outline; i += 64) keep[i] = mine[i];
        fn = last.n_outline;
        if (lane == 0) { first_n[b] = fn; first_thr[b] = t; }
    }
    const uint32_t initial = detect[b].n_pixels;
    const uint32_t minimum = initial / 10u > 1u ? initial / 10u : 1u;
    const int tn = t + 2;
    if (count < minimum || tn >= tpt + 100) {                                         // give up: the first outline, no midline
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < fn; i += 64) mine[i] = keep[i];
        if (lane == 0) {
            last.n_outline = fn; last.n_segments = 0;
            if (last.status == 0) last.status = 1;
            info[b] = last;
            used[b] = fn > 0 ? (first_thr[b] >= 0 ? first_thr[b] : t) : -1;
            thr[b] = -1;
        }
        return;
    }
    if (lane == 0) { thr[b] = tn; atomicAdd(active, 1u); }
}
}  // namespace trexhip

extern "C" int trexhip_posture_auto_device(trexhip_ctx* ctx, const trexhip_posture_params* pp, int32_t method, int32_t track_posture_threshold,
                                           int32_t n_blobs, float* d_outline, float* d_segments, trexhip_posture_info* d_info,
                                           int32_t* d_threshold_used, int32_t* d_iterations) {
    if (!ctx || !pp || !d_outline || !d_segments || !d_info) { set_error("trexhip_posture_auto_device: null argument"); return TREXHIP_E_INVALID; }
    if (method < 0 || method > 2) { set_error("trexhip_posture_auto_device: method must be 0 (absolute), 1 (sign) or 2 (none)"); return TREXHIP_E_INVALID; }
    if (track_posture_threshold < 0 || track_posture_threshold > 255) { set_error("trexhip_posture_auto_device: track_posture_threshold must be 0..255"); return TREXHIP_E_INVALID; }
    // the argument checks of the single pass (capacities, refused settings) by a zero-blob call of it
    if (int rc = trexhip_posture_device(ctx, 0, pp, 0, d_outline, d_segments, d_info)) return rc;
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_posture_auto_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    const int n = n_blobs, MPt = pp->max_points;
    // scratch of the loop, kept by the context
    const size_t need = (size_t)n * (6 * sizeof(int32_t) + sizeof(unsigned long long) + (size_t)MPt * sizeof(float2)) + 64;
    if (ctx->auto_cap < need) {
        if (ctx->d_auto) (void)hipFree(ctx->d_auto);
        ctx->d_auto = nullptr; ctx->auto_cap = 0;
        TH_CHECK_HIP(hipMalloc(&ctx->d_auto, need));
        ctx->auto_cap = need;
    }
    uint8_t* base = static_cast<uint8_t*>(ctx->d_auto);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(base); base += (size_t)n * 8;
    float2* first = reinterpret_cast<float2*>(base); base += (size_t)n * MPt * sizeof(float2);
    int32_t* thr = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    int32_t* sel = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    int32_t* first_n = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    int32_t* first_thr = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    int32_t* used = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    int32_t* iters = reinterpret_cast<int32_t*>(base); base += (size_t)n * 4;
    uint32_t* active = reinterpret_cast<uint32_t*>(base);
    hipStream_t s = ctx->stream;
    const dim3 g256((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(k_auto_init, g256, dim3(256), 0, s, n, (int)track_posture_threshold, thr, first_n, first_thr, used, iters, best, d_info);
    // LDS capacities of the posture launches: a thresholded sub-blob has at most its parent's rows, and more lines only where a line splits
    int nr_cap = P_NR, rows_cap = P_ROWS;
    if (ctx->h_blobs && ctx->fetched) {
        uint32_t mr = 1, mrows = 1;
        for (int i = 0; i < n; ++i) {
            const trexhip_blob& b = ctx->h_blobs[i];
            const uint32_t rws = (uint32_t)(b.y1 - b.y0 + 1);
            if (rws <= (uint32_t)P_ROWS) { mr = std::max(mr, std::min<uint32_t>(b.n_runs * 2u, (uint32_t)P_NR)); mrows = std::max(mrows, rws); }
        }
        nr_cap = (int)std::min<uint32_t>((mr + 31u) & ~31u, (uint32_t)P_NR);
        rows_cap = (int)std::min<uint32_t>((mrows + 29u) / 32u * 32u + 30u, (uint32_t)P_ROWS);
    }
    PostureCfg P{pp->outline_resample, pp->outline_smooth_samples, pp->outline_smooth_step, pp->outline_approximate,
                 pp->outline_curvature_range_ratio, pp->midline_walk_offset, pp->max_points, nr_cap, rows_cap, 0, posture_walk_groups(pp)};
#ifdef TREXHIP_DEV_KNOBS
    if (const char* e = std::getenv("TREXHIP_POSTURE_WALK_GROUP")) P.walk_group = std::atoi(e);
#endif
    int wpb = 1, lds_bytes = 0;
    auto size_launch = [&]() -> int {
        const int wave_lds = posture_wave_lds(MPt, P.nr_cap, P.rows_cap);
        wpb = 4;
        while (wpb > 1 && wpb * wave_lds > 150 * 1024) wpb >>= 1;
        lds_bytes = wpb * wave_lds;
        if (lds_bytes > ctx->attr_posture_bytes) {
            TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_posture), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            ctx->attr_posture_bytes = lds_bytes;
        }
        return TREXHIP_OK;
    };
    if (int rc = size_launch()) return rc;
    for (int round = 0; round < 51; ++round) {                        // thresholds t, t+2, ..., below t+100
        int rc = trexhip_rethreshold_per_blob_device(ctx, 0, thr, method, nullptr, 0);     // negative entries (finished blobs) are skipped
        if (rc) return rc;
        const Pass2& q = ctx->pass2;
        hipLaunchKernelGGL(k_auto_pick, dim3((ctx->cfg.pool_blobs + 255u) / 256u), dim3(256), 0, s, q.d_blobs, q.d_totals, ctx->cfg.pool_blobs, n, best);
        hipLaunchKernelGGL(k_auto_sel, g256, dim3(256), 0, s, n, thr, best, sel);
        uint32_t still[2] = {0, 0};
        for (int attempt = 0; attempt < 2; ++attempt) {               // attempt 1: only if a sub-blob did not fit the estimated LDS capacities
            TH_CHECK_HIP(hipMemsetAsync(active, 0, 8, s));
            stage_begin(ctx, TREXHIP_STAGE_POSTURE);
            hipLaunchKernelGGL(k_posture, dim3((n + wpb - 1) / wpb), dim3(wpb * 64), lds_bytes, s, P, q.d_info, q.d_blob_frame, q.d_blobs, q.d_runs, n, ctx->last_n,
                               reinterpret_cast<float2*>(d_outline), reinterpret_cast<float4*>(d_segments), d_info, sel, ctx->d_blobs);
            launch_posture_walk(s, pp, P.walk_group, n, d_outline, d_segments, d_info);
            stage_end(ctx, TREXHIP_STAGE_POSTURE);
            hipLaunchKernelGGL(k_auto_step, dim3((n + 3) / 4), dim3(256), 0, s, n, (int)track_posture_threshold, MPt, ctx->d_blobs, thr, sel, best, first_n, first_thr,
                               used, iters, first, reinterpret_cast<float2*>(d_outline), d_info, active, q.d_blobs, P.nr_cap, P.rows_cap);
            TH_CHECK_HIP(hipGetLastError());
            TH_CHECK_HIP(hipMemcpyAsync(still, active, 8, hipMemcpyDeviceToHost, s));
            TH_CHECK_HIP(hipStreamSynchronize(s));
            if (still[1] == 0) break;
            P.nr_cap = P_NR; P.rows_cap = P_ROWS;                      // from here on the kernel's real limits (fewer blobs per CU)
            if (int rc2 = size_launch()) return rc2;
        }
        if (still[0] == 0) break;
    }
    if (d_threshold_used) TH_CHECK_HIP(hipMemcpyAsync(d_threshold_used, used, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    if (d_iterations) TH_CHECK_HIP(hipMemcpyAsync(d_iterations, iters, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return TREXHIP_OK;
}

extern "C" void trexhip_default_posture_params(trexhip_posture_params* p) {
    if (!p) return;
    p->outline_resample = 1.0f; p->outline_smooth_samples = 4; p->outline_smooth_step = 1; p->outline_approximate = 3;
    p->outline_curvature_range_ratio = 0.03f; p->midline_walk_offset = 0.025f; p->max_points = 512;
    p->posture_closing_steps = 0; p->peak_mode = 0; p->posture_direction_smoothing = 0;
}
