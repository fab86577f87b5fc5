/*
 * trex_posture.c -- CPU restatement of posture::calculate_posture for ONE blob (TEST INFRASTRUCTURE ONLY).
 *
 * Follows Application/src/tracker/tracking/Posture.cpp:305-399 and Outline.cpp:
 *   outline of the (thresholded) blob   pixel::find_outer_points          [commons, NOT IN TREE]
 *   Outline::resample                   Outline.cpp:724-766               in tree, pinned by Tests/test_outlines.cpp:53-95
 *   smooth_outline / Outline::smooth    Outline.cpp:330-378,380-..        in tree
 *   Outline::offset_to_middle           Outline.cpp:454-718               in tree, but built on periodic::* [commons, NOT IN TREE]:
 *       differentiate_and_test_clockwise, eft, ieft, curvature, find_peaks
 *   Outline::calculate_midline          Outline.cpp:768-868               in tree (two-pointer walk)
 *
 * PARITY STATUS: resample / smooth / the midline walk follow in-tree source line by line.  The pieces that live in
 * the un-vendored commons are restated from their published algorithms and are UNPINNED:
 *   - outline = outer boundary of the pixel set walked along pixel edges clockwise (image coordinates), emitting
 *     every pixel-corner and every edge midpoint (half-pixel lattice), 8-connected turns; first vertex = top-left
 *     corner of the first pixel of the first line;
 *   - elliptic Fourier transform = Kuhl & Giardina (1982) in the form used by the pyefd package, order
 *     `outline_approximate`, inverse sampled at N uniform parameters around `center` (mean of the points);
 *   - curvature = Menger curvature over i-r, i, i+r (the formula left commented out in Outline.cpp:305-312), absolute;
 *   - peaks = strict/plateau-start local maxima of that curvature; tail = highest peak (peak_mode pointy,
 *     Outline.cpp:624-625), head = peak farthest from the tail in circular index distance (:668-681).
 * All arithmetic in float (Float2_t), no FMA contraction (compiled with -ffp-contract=off).
 */
#include "trex_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } v2;

/* ---- blob membership through a row table -------------------------------------------------------- */
typedef struct { const oracle_run* runs; int n; int y0, y1; int* row_start; } rowtab;

static int in_blob(const rowtab* t, int x, int y) {
    if (y < t->y0 || y > t->y1) return 0;
    for (int i = t->row_start[y - t->y0]; i < t->row_start[y - t->y0 + 1]; ++i)
        if (x >= t->runs[i].x0 && x <= t->runs[i].x1) return 1;
    return 0;
}

/* outer boundary on the half-pixel lattice; returns number of points or -1 if cap is exceeded */
static int trace_outline(const oracle_run* runs, int n_runs, int ox, int oy, v2* out, int cap) {
    rowtab t; t.runs = runs; t.n = n_runs; t.y0 = runs[0].y; t.y1 = runs[n_runs - 1].y;
    const int rows = t.y1 - t.y0 + 1;
    t.row_start = (int*)malloc((size_t)(rows + 1) * sizeof(int));
    int r = 0;
    for (int y = 0; y < rows; ++y) { t.row_start[y] = r; while (r < n_runs && runs[r].y == t.y0 + y) ++r; }
    t.row_start[rows] = n_runs;
    /* doubled coordinates: pixel (x,y) has corners (2x+-1, 2y+-1) */
    const int sx = 2 * runs[0].x0 - 1, sy = 2 * runs[0].y - 1;
    int vx = sx, vy = sy, dx = 1, dy = 0, n = 0;
    do {
        if (n + 2 > cap) { free(t.row_start); return -1; }
        out[n].x = 0.5f * (float)(vx - 2 * ox); out[n].y = 0.5f * (float)(vy - 2 * oy); ++n;                 /* corner   */
        out[n].x = 0.5f * (float)(vx + dx - 2 * ox); out[n].y = 0.5f * (float)(vy + dy - 2 * oy); ++n;       /* midpoint */
        vx += 2 * dx; vy += 2 * dy;
        const int lx = dy, ly = -dx, rx = -dy, ry = dx;          /* left / right of the walking direction (y points down) */
        const int plx = (vx + dx + lx) / 2, ply = (vy + dy + ly) / 2;
        const int prx = (vx + dx + rx) / 2, pry = (vy + dy + ry) / 2;
        if (in_blob(&t, plx, ply)) { dx = lx; dy = ly; }         /* diagonal neighbours count: 8-connectivity */
        else if (in_blob(&t, prx, pry)) { /* straight on */ }
        else { dx = rx; dy = ry; }
    } while (!(vx == sx && vy == sy && dx == 1 && dy == 0));
    free(t.row_start);
    return n;
}

/* ---- Outline::resample (Outline.cpp:724-766) ------------------------------------------------------ */
int oracle_outline_resample(const float* pts_xy, int L, float resampling_distance, float* out_xy, int cap) {
    if (resampling_distance <= 0 || L <= 1) { memcpy(out_xy, pts_xy, (size_t)L * 2 * sizeof(float)); return L; }
    const v2* p = (const v2*)pts_xy; v2* o = (v2*)out_xy;
    float walked = 0.0f; int n = 0;
    for (int i = 0; i < L; ++i) {
        int i1 = i + 1; if (i1 >= L) i1 -= L;
        const v2 pt0 = p[i], pt1 = p[i1];
        const v2 line = { pt1.x - pt0.x, pt1.y - pt0.y };
        const float len = sqrtf(line.x * line.x + line.y * line.y);
        walked += len;
        const float percent = len / resampling_distance;
        float walked_percent = walked / resampling_distance;
        int offset = 0;
        while (walked_percent >= 1.0) {
            const float tt = (float)((double)offset * 1.0 / (double)percent);
            if (n >= cap) return -1;
            o[n].x = pt0.x + line.x * tt; o[n].y = pt0.y + line.y * tt; ++n;
            offset++;
            walked -= resampling_distance;
            walked_percent -= 1.0f;
        }
    }
    return n;
}

/* ---- smooth_outline (Outline.cpp:330-378) ----------------------------------------------------------- */
static int smooth_outline(const v2* p, int L, int range, int step, v2* out) {
    if (!(L > range)) return 0;
    const float step_row = (float)range * (float)step;
    float w[64]; int nw = 0; float sum = 0;
    for (int i = (int)-step_row; i <= step_row; i += step) { const float val = (step_row - fabsf((float)i)) / step_row; sum += val; w[nw++] = val; }
    for (int i = 0; i < nw; ++i) w[i] /= sum;
    for (int i = 0; i < L; ++i) {
        v2 pt = {0, 0}; int s = 0;
        for (long j = (long)((float)i - step_row); j <= (float)i + step_row; j += step) {
            long idx = j; while (idx < 0) idx += L; while (idx >= L) idx -= L;
            pt.x += p[idx].x * w[s]; pt.y += p[idx].y * w[s]; ++s;
        }
        out[i] = pt;
    }
    return 1;
}

/* ---- periodic::* restatements (unpinned, see header) ----------------------------------------------
 * periodic::eft / ieft live in the un-vendored commons: what is summed is restated, the ORDER of the float sums and the sin / cos
 * implementation are not known.  Both are therefore fixed here in a form the device reproduces operation by operation
 * (trex_amd/csrc/posture.hip, same constants, no FMA contraction on either side), so that device and oracle agree bit for bit:
 *   - det_sincosf: Cody-Waite reduction by pi/2 in three pieces + the classic degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4]
 *     (about 1 ulp; glibc's sinf / cosf differ from it in the last bit here and there);
 *   - arc length: inclusive Hillis-Steele scan over blocks of 64 segments + a running carry (the wave scan of the device);
 *   - coefficient sums: 64 interleaved partial sums (term i goes to partial i % 64, in order of i), combined by the butterfly
 *     v[l] += v[l ^ d], d = 32 .. 1.
 * The centre is NOT part of this: Outline.cpp:502-505 (in tree) sums the points in order, and so do both sides. */
static void det_sincosf(const float x, float* sn, float* cs) {
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
    switch (k & 3) {
        case 0: *sn = s0; *cs = c0; break;
        case 1: *sn = c0; *cs = -s0; break;
        case 2: *sn = -s0; *cs = -c0; break;
        default: *sn = -c0; *cs = s0; break;
    }
}
static float butterfly64(float* v) {
    for (int d = 32; d >= 1; d >>= 1) {
        float w[64];
        for (int l = 0; l < 64; ++l) w[l] = v[l] + v[l ^ d];
        memcpy(v, w, sizeof(w));
    }
    return v[0];
}
static void eft_ieft(const v2* p, int N, int order, v2 center, v2* out) {
    float* t = (float*)malloc((size_t)(N + 1) * sizeof(float));          /* t[i + 1] = arc length at the END of segment i */
    t[0] = 0;
    float run = 0;
    for (int i0 = 0; i0 < N; i0 += 64) {
        float incl[64];
        for (int l = 0; l < 64; ++l) {
            const int i = i0 + l;
            incl[l] = 0;
            if (i < N) { const v2 q = p[(i + 1) % N]; const float dx = q.x - p[i].x, dy = q.y - p[i].y; incl[l] = sqrtf(dx * dx + dy * dy); }
        }
        for (int d = 1; d < 64; d <<= 1) {
            float w[64];
            for (int l = 0; l < 64; ++l) w[l] = l >= d ? incl[l] + incl[l - d] : incl[l];
            memcpy(incl, w, sizeof(w));
        }
        for (int l = 0; l < 64 && i0 + l < N; ++l) t[i0 + l + 1] = run + incl[l];
        run = run + incl[63];
    }
    const float T = t[N];
    float a[16], b[16], c[16], d[16];
    const float PI = 3.14159265358979323846f;
    float* cs = (float*)malloc((size_t)(N + 1) * 2 * sizeof(float));
    /* cos / sin of the FIRST harmonic's phase at every boundary; harmonics 2 and 3 by the angle addition formulas, written out in the order the
     * device uses (c2 = c1 c1 - s1 s1, s2 = 2 (s1 c1), c3 = c2 c1 - s2 s1, s3 = s2 c1 + c2 s1) */
    for (int i = 0; i <= N; ++i) det_sincosf(2.0f * PI * (float)1 * t[i] / T, &cs[2 * i + 1], &cs[2 * i]);
    float sa[16][64], sb[16][64], sc[16][64], sd[16][64];
    memset(sa, 0, sizeof(sa)); memset(sb, 0, sizeof(sb)); memset(sc, 0, sizeof(sc)); memset(sd, 0, sizeof(sd));
    /* (order <= 15 here: oracle_posture refuses more, like trexhip_posture_device.  Harmonic 2 by the doubling form, every further one from its
     * predecessor by angle addition: the device's general form (more than three harmonics, three per sweep) recomputes exactly this sequence) */
    for (int i = 0; i < N; ++i) {
        const v2 q = p[(i + 1) % N];
        const float dx = q.x - p[i].x, dy = q.y - p[i].y;
        const float dt = t[i + 1] - t[i];
        if (dt <= 0) continue;
        const float c0 = cs[2 * i], s0 = cs[2 * i + 1], c1 = cs[2 * (i + 1)], s1 = cs[2 * (i + 1) + 1];
        const float gx = dx / dt, gy = dy / dt;
        float c0h = c0, s0h = s0, c1h = c1, s1h = s1;
        const int l = i & 63;
        for (int n = 1; n <= order; ++n) {
            if (n > 1) {
                float cn, sn;
                if (n == 2) { cn = c0 * c0 - s0 * s0; sn = 2.0f * (s0 * c0); } else { cn = c0h * c0 - s0h * s0; sn = s0h * c0 + c0h * s0; }
                c0h = cn; s0h = sn;
                if (n == 2) { cn = c1 * c1 - s1 * s1; sn = 2.0f * (s1 * c1); } else { cn = c1h * c1 - s1h * s1; sn = s1h * c1 + c1h * s1; }
                c1h = cn; s1h = sn;
            }
            const float dc = c1h - c0h, ds = s1h - s0h;
            sa[n][l] += gx * dc; sb[n][l] += gx * ds; sc[n][l] += gy * dc; sd[n][l] += gy * ds;
        }
    }
    for (int n = 1; n <= order; ++n) {
        const float k = T / (2.0f * (float)(n * n) * PI * PI);
        a[n] = k * butterfly64(sa[n]); b[n] = k * butterfly64(sb[n]); c[n] = k * butterfly64(sc[n]); d[n] = k * butterfly64(sd[n]);
    }
    for (int k = 0; k < N; ++k) {
        const float tt = (float)k / (float)N;
        float x = center.x, y = center.y;
        float s1, c1;
        det_sincosf(2.0f * PI * (float)1 * tt, &s1, &c1);
        float ch = c1, sh = s1;
        for (int n = 1; n <= order; ++n) {
            if (n > 1) {
                float cn, sn;
                if (n == 2) { cn = c1 * c1 - s1 * s1; sn = 2.0f * (s1 * c1); } else { cn = ch * c1 - sh * s1; sn = sh * c1 + ch * s1; }
                ch = cn; sh = sn;
            }
            x += a[n] * ch + b[n] * sh; y += c[n] * ch + d[n] * sh;
        }
        out[k].x = x; out[k].y = y;
    }
    free(cs);
    free(t);
}

/* The NAIVE reading of the same formulas (round 6): what somebody who had never seen the device would write down from the
 * definition of the elliptic Fourier transform -- libm's sinf / cosf of every harmonic's own phase, arc length and coefficient sums
 * accumulated sequentially in float, any order up to 15 (`outline_approximate` is a uint8_t without an upper bound, core/default_config.cpp:888).
 * It shares NOTHING with the mirror above but the formulas.  tests/test_posture_oracle.py and tests/test_posture_gpu.py bound the distance
 * between the two (and so between the device and this one): a future change of the shared operation order cannot drift unseen. */
static void eft_ieft_naive(const v2* p, int N, int order, v2 center, v2* out) {
    float* t = (float*)malloc((size_t)(N + 1) * sizeof(float));
    t[0] = 0;
    for (int i = 0; i < N; ++i) {
        const v2 q = p[(i + 1) % N];
        const float dx = q.x - p[i].x, dy = q.y - p[i].y;
        t[i + 1] = t[i] + sqrtf(dx * dx + dy * dy);
    }
    const float T = t[N];
    const float PI = 3.14159265358979323846f;
    if (order > 15) order = 15;
    float a[16], b[16], c[16], d[16];
    for (int n = 1; n <= order; ++n) {
        float sa = 0, sb = 0, sc = 0, sd = 0;
        for (int i = 0; i < N; ++i) {
            const v2 q = p[(i + 1) % N];
            const float dx = q.x - p[i].x, dy = q.y - p[i].y;
            const float dt = t[i + 1] - t[i];
            if (dt <= 0) continue;
            const float p0 = 2.0f * PI * (float)n * t[i] / T, p1 = 2.0f * PI * (float)n * t[i + 1] / T;
            const float dc = cosf(p1) - cosf(p0), ds = sinf(p1) - sinf(p0);
            sa += dx / dt * dc; sb += dx / dt * ds; sc += dy / dt * dc; sd += dy / dt * ds;
        }
        const float k = T / (2.0f * (float)(n * n) * PI * PI);
        a[n] = k * sa; b[n] = k * sb; c[n] = k * sc; d[n] = k * sd;
    }
    for (int k = 0; k < N; ++k) {
        const float tt = (float)k / (float)N;
        float x = center.x, y = center.y;
        for (int n = 1; n <= order; ++n) {
            const float ph = 2.0f * PI * (float)n * tt;
            x += a[n] * cosf(ph) + b[n] * sinf(ph); y += c[n] * cosf(ph) + d[n] * sinf(ph);
        }
        out[k].x = x; out[k].y = y;
    }
    free(t);
}

static void curvature_abs(const v2* p, int N, int r, float* out) {
    for (int i = 0; i < N; ++i) {
        const v2 p1 = p[((i - r) % N + N) % N], p2 = p[i], p3 = p[(i + r) % N];
        const float cr = (p2.x - p1.x) * (p3.y - p2.y) - (p2.y - p1.y) * (p3.x - p2.x);
        const float d12 = (p2.x - p1.x) * (p2.x - p1.x) + (p2.y - p1.y) * (p2.y - p1.y);
        const float d23 = (p3.x - p2.x) * (p3.x - p2.x) + (p3.y - p2.y) * (p3.y - p2.y);
        const float d13 = (p3.x - p1.x) * (p3.x - p1.x) + (p3.y - p1.y) * (p3.y - p1.y);
        const float den = sqrtf(d12 * d23 * d13);
        out[i] = den > 0 ? fabsf(2.0f * cr / den) : 0.0f;
    }
}

/* the two-pointer walk of Outline::calculate_midline (Outline.cpp:790-857) alone: outline points (tail = point 0) -> midline
 * segments {pos.x, pos.y, height, l_length}; returns their number.  `segments` must hold n entries. */
int oracle_midline_walk(const float* pts_xy, int32_t n, float midline_walk_offset, float* segments) {
    const v2* pts = (const v2*)pts_xy;
    const int L = n;
    int idx_r = 1, idx_l = -1, ns = 0;
    float mo = midline_walk_offset * (float)L; if (mo < 3.0f) mo = 3.0f;
    const int max_offset = (int)mo;
    while (idx_r < L + idx_l) {
        v2 pt_r = {0, 0}; v2 pt_l = pts[L + idx_l];
        float min_d = 3.402823466e38f; int min_idx = -1;
        for (int i = 0; i < max_offset; ++i) {
            if (idx_r + i >= L) break;
            const v2 pt = pts[idx_r + i];
            const float dx = pt.x - pt_l.x, dy = pt.y - pt_l.y, len = sqrtf(dx * dx + dy * dy);
            if (len < min_d) { min_d = len; min_idx = idx_r + i; }
        }
        if (min_idx != -1) { pt_r = pts[min_idx]; idx_r = min_idx; }
        min_d = 3.402823466e38f; min_idx = 1;
        for (int i = 0; i < max_offset; ++i) {
            if (idx_l - i <= -L) break;
            const v2 pt = pts[L + idx_l - i];
            const float dx = pt_r.x - pt.x, dy = pt_r.y - pt.y, len = sqrtf(dx * dx + dy * dy);
            if (len < min_d) { min_d = len; min_idx = idx_l - i; }
        }
        if (min_idx != 1) { pt_l = pts[L + min_idx]; idx_l = min_idx; }
        const float lx = pt_r.x - pt_l.x, ly = pt_r.y - pt_l.y;
        const v2 m = { pt_l.x + lx * 0.5f, pt_l.y + ly * 0.5f };
        segments[4 * ns + 0] = m.x; segments[4 * ns + 1] = m.y;
        segments[4 * ns + 2] = sqrtf(lx * lx + ly * ly);
        segments[4 * ns + 3] = sqrtf((m.x - pt_l.x) * (m.x - pt_l.x) + (m.y - pt_l.y) * (m.y - pt_l.y));
        ++ns;
        idx_r++; idx_l--;
    }
    return ns;
}

typedef struct oracle_posture_params {
    float outline_resample; int32_t outline_smooth_samples, outline_smooth_step, outline_approximate;
    float outline_curvature_range_ratio, midline_walk_offset; int32_t max_points;
} oracle_posture_params;
/* peak_best / peak_runner_up: curvature at the chosen tail and the largest curvature anywhere farther than the curvature range from it
 * (test infrastructure: when the two are within float rounding of each other the tail is a coin flip, and a test can tell a legitimate
 * tie from a wrong choice); peak_margin: by how much the tail's curvature exceeds the larger of its two neighbours, relative -- the
 * local-maximum test itself (c[i] > c[i-1] && c[i] >= c[i+1]) is a coin flip on a flat top */
typedef struct oracle_posture_info { int32_t status, n_outline, n_segments, tail_index, head_index, n_traced; float peak_best, peak_runner_up, peak_margin; } oracle_posture_info;

/* status: 0 ok, 1 empty blob/outline, 2 capacity, 3 no curvature peak, 4 too few midline segments, 5 refused (outline_approximate > 15 in the mirrored form) */
static int posture_impl(const oracle_run* runs, int32_t n_runs, int32_t origin_x, int32_t origin_y, const oracle_posture_params* P,
                        float* outline_xy, float* segments /* pos.x pos.y height l_length */, oracle_posture_info* info, const int naive) {
    memset(info, 0, sizeof(*info));
    if (!naive && P->outline_approximate > 15) { info->status = 5; return 5; }     /* refuse like the device (posture.hip), never cap silently */
    if (n_runs <= 0) { info->status = 1; return 1; }
    const int cap = P->max_points;
    v2* A = (v2*)malloc((size_t)cap * sizeof(v2)); v2* B = (v2*)malloc((size_t)cap * sizeof(v2));
    float* curv = (float*)malloc((size_t)cap * sizeof(float));
    int rc = 0;
    int n = trace_outline(runs, n_runs, origin_x, origin_y, A, cap);
    if (n < 0) { rc = 2; goto done; }
    info->n_traced = n;
    n = oracle_outline_resample((float*)A, n, P->outline_resample, (float*)B, cap);          /* Posture.cpp:355 */
    if (n < 0) { rc = 2; goto done; }
    if (n == 0) { rc = 1; goto done; }
    /* Outline::calculate_midline (Outline.cpp:768-868) */
    v2* pts = B; v2* other = A;
    if (P->outline_smooth_samples > 0 && smooth_outline(pts, n, P->outline_smooth_samples, P->outline_smooth_step, other)) { v2* t = pts; pts = other; other = t; }
    {   /* offset_to_middle (Outline.cpp:454-718) */
        float sum = 0;                                                              /* clockwise test (shoelace, y down) */
        for (int i = 0; i < n; ++i) { const v2 q = pts[(i + 1) % n]; sum += pts[i].x * q.y - q.x * pts[i].y; }
        if (sum < 0) for (int i = 0; i < n / 2; ++i) { v2 t = pts[i]; pts[i] = pts[n - 1 - i]; pts[n - 1 - i] = t; }
        if (P->outline_approximate > 0) {
            v2 center = {0, 0};
            for (int i = 0; i < n; ++i) { center.x += pts[i].x; center.y += pts[i].y; }
            center.x /= (float)n; center.y /= (float)n;
            if (naive) eft_ieft_naive(pts, n, P->outline_approximate, center, other);
            else eft_ieft(pts, n, P->outline_approximate, center, other);
            v2* t = pts; pts = other; other = t;
        }
        int r = (int)(P->outline_curvature_range_ratio * (float)n); if (r < 1) r = 1;
        curvature_abs(pts, n, r, curv);
        int tail = -1; float best = -1;
        for (int i = 0; i < n; ++i) {
            const float c0 = curv[(i - 1 + n) % n], c1 = curv[i], c2 = curv[(i + 1) % n];
            if (c1 > c0 && c1 >= c2 && c1 > best) { best = c1; tail = i; }
        }
        if (tail < 0) { rc = 3; memcpy(outline_xy, pts, (size_t)n * sizeof(v2)); info->n_outline = n; goto done; }   /* the outline stays available (first_outline fallback, Posture.cpp:361-368) */
        {
            float runner = 0.f;
            for (int i = 0; i < n; ++i) {
                int dd = i > tail ? i - tail : tail - i; if (n - dd < dd) dd = n - dd;
                if (dd > r && curv[i] > runner) runner = curv[i];
            }
            info->peak_best = best; info->peak_runner_up = runner;
            { const float c0 = curv[(tail - 1 + n) % n], c2 = curv[(tail + 1) % n]; info->peak_margin = best > 0.f ? (best - (c0 > c2 ? c0 : c2)) / best : 0.f; }
        }
        int head = -1; float maxd = 0;
        for (int i = 0; i < n; ++i) {
            const float c0 = curv[(i - 1 + n) % n], c1 = curv[i], c2 = curv[(i + 1) % n];
            if (!(c1 > c0 && c1 >= c2)) continue;
            float dd;
            if (i >= tail) dd = fminf(fabsf((float)(i - tail)), fabsf((float)(i - tail - n)));
            else dd = fminf(fabsf((float)(tail - i)), fabsf((float)(tail - i - n)));
            if (dd > maxd) { maxd = dd; head = i; }
        }
        /* rotate so that the tail is point 0 (Outline.cpp:707) */
        for (int i = 0; i < n; ++i) other[i] = pts[(i + tail) % n];
        { v2* t = pts; pts = other; other = t; }
        info->tail_index = 0;
        info->head_index = head < 0 ? -1 : ((head - tail) % n + n) % n;
    }
    memcpy(outline_xy, pts, (size_t)n * sizeof(v2));
    info->n_outline = n;
    if (n <= 1) { rc = 1; goto done; }
    {   /* the two-pointer walk (Outline.cpp:790-857) */
        const int ns = oracle_midline_walk((const float*)pts, n, P->midline_walk_offset, segments);
        info->n_segments = ns;
        if (ns <= 2) rc = 4;
    }
done:
    info->status = rc;
    free(A); free(B); free(curv);
    return rc;
}

int oracle_posture(const oracle_run* runs, int32_t n_runs, int32_t origin_x, int32_t origin_y, const oracle_posture_params* P,
                   float* outline_xy, float* segments, oracle_posture_info* info) {
    return posture_impl(runs, n_runs, origin_x, origin_y, P, outline_xy, segments, info, 0);
}
/* the same pipeline with the naive EFT (libm, sequential sums, any order): the independent yardstick */
int oracle_posture_naive(const oracle_run* runs, int32_t n_runs, int32_t origin_x, int32_t origin_y, const oracle_posture_params* P,
                         float* outline_xy, float* segments, oracle_posture_info* info) {
    return posture_impl(runs, n_runs, origin_x, origin_y, P, outline_xy, segments, info, 1);
}

/* outline tracing alone (tests) */
int oracle_trace_outline(const oracle_run* runs, int32_t n_runs, int32_t ox, int32_t oy, float* out_xy, int32_t cap) {
    if (n_runs <= 0) return 0;
    return trace_outline(runs, n_runs, ox, oy, (v2*)out_xy, cap);
}

/* ---- Midline::post_process (movement = none) + Midline::normalize (Outline.cpp:895-1060, 1270-1454) ---------------------
 * in : raw midline of calculate_midline, tail first; out: `resolution` segments, head at the origin, rotated by -angle+pi
 * (Individual.cpp:1369-1372: post_process, then normalize()).  Float2_t = float, the `double` accumulators of normalize
 * are kept.  UNPINNED pieces (commons, NOT IN TREE): Vec2::normalize() of a zero vector returns zero; Vec2 * double is
 * computed in double and narrowed per component; the axis loop of post_process stops before segments().at(size)
 * (which would throw in the reference).  status: 0 ok, 1 <2 segments / zero length, 2 resampling did not give `resolution`
 * points (normalize returns nullptr, :1378-1380). */
typedef struct oracle_midline_info { int32_t status, n; float len, angle, offx, offy; int32_t reserved[2]; } oracle_midline_info;

static v2 v2norm(v2 a) { const float L = sqrtf(a.x * a.x + a.y * a.y); v2 r = {0, 0}; if (L > 0) { r.x = a.x / L; r.y = a.y / L; } return r; }
static float v2len(float x, float y) { return sqrtf(x * x + y * y); }

static float midline_calculate_angle(const float* s4, int n, float stiff) {            /* Outline.cpp:1114-1124 */
    if (n < 2) return 0;
    float center = (float)(n - 2) - (float)n * stiff; if (center < 0) center = 0;
    const int start = (int)center;
    const float rest = center - (float)start;
    const int s1 = start + 1 < n ? start + 1 : n - 1;
    const float lx = s4[4 * (n - 1)] - (s4[4 * start] * (1 - rest) + s4[4 * s1] * rest);
    const float ly = s4[4 * (n - 1) + 1] - (s4[4 * start + 1] * (1 - rest) + s4[4 * s1 + 1] * rest);
    return atan2f(ly, lx);
}

/* Midline::midline_direction (Outline.cpp:870-887): mean of the first max(1, size * midline_stiff_percentage) segment steps, normalised.
 * UNPINNED detail: cmn::max(int, float) is taken to return the float (then narrowed to long_t) */
static v2 midline_direction(const float* s4, int n, float stiff) {
    const float sf = (float)n * stiff;
    const long samples = (long)(sf > 1.f ? sf : 1.f);
    v2 d = {0, 0};
    long counted = 0;
    for (long i = 0; i < samples && i + 1 < (long)n; ++i, ++counted) { d.x += s4[4 * (i + 1)] - s4[4 * i]; d.y += s4[4 * (i + 1) + 1] - s4[4 * i + 1]; }
    if (counted > 0) { d.x /= (float)counted; d.y /= (float)counted; d = v2norm(d); }
    return d;
}

/* post_process with MovementInformation::direction = (mvx, mvy) (Outline.cpp:905-961; (0,0) = no movement information, the case of
 * posture_direction_smoothing <= 1, Individual.cpp:1366-1368): the midline is turned round when its direction points against the
 * movement, *flipped = `_inverted_because_previous` (the caller swaps head_index / tail_index, :959) */
int oracle_midline_post_process_mv(float* s4, int n, float stiff, int invert, int start_with_head, float mvx, float mvy, int* flipped) {      /* :895-1060 */
    if (flipped) *flipped = 0;
    if (n <= 2) return 1;
    int needs_invert = !invert;
    if (mvx != 0.f || mvy != 0.f) {
        v2 d = midline_direction(s4, n, stiff);
        if (!needs_invert) { d.x = -d.x; d.y = -d.y; }
        const float against = (-d.x) * mvx + (-d.y) * mvy, along = d.x * mvx + d.y * mvy;
        if (acosf(against) < acosf(along)) { needs_invert = !needs_invert; if (flipped) *flipped = 1; }
    }
    int rev = needs_invert ? !start_with_head : start_with_head;
    if (rev) for (int i = 0; i < n / 2; ++i) for (int k = 0; k < 4; ++k) { float t = s4[4 * i + k]; s4[4 * i + k] = s4[4 * (n - 1 - i) + k]; s4[4 * (n - 1 - i) + k] = t; }
    if (stiff > 0) {
        float cf = roundf((float)n * stiff) + 1; if ((float)n - 1 < cf) cf = (float)n - 1;
        const int center = (int)cf;
        const v2 cp = { s4[4 * center], s4[4 * center + 1] };
        double eo = (double)center + ((double)n * 0.1 > 0.0 ? (double)n * 0.1 : 0.0); if ((double)n < eo) eo = (double)n;
        const int extra = (int)eo;
        v2 axis = {0, 0}; unsigned count = 0;
        for (int i = center; i < extra && i + 1 < n; ++i) {
            v2 d = { s4[4 * i] - s4[4 * (i + 1)], s4[4 * i + 1] - s4[4 * (i + 1) + 1] };
            d = v2norm(d); axis.x += d.x; axis.y += d.y; ++count;
        }
        if (count > 0) { axis.x /= (float)count; axis.y /= (float)count; }
        /* copy.at(i) - copy.at(i-1): lengths of the ORIGINAL segments */
        float* L = (float*)malloc((size_t)(center + 1) * sizeof(float));
        for (int i = center; i > 0; --i) L[i] = v2len(s4[4 * i] - s4[4 * (i - 1)], s4[4 * i + 1] - s4[4 * (i - 1) + 1]);
        for (int i = center; i > 0; --i) {
            const v2 p1 = { s4[4 * i], s4[4 * i + 1] };
            v2 dc = { s4[4 * (i - 1)] - cp.x, s4[4 * (i - 1) + 1] - cp.y };
            dc = v2norm(dc);
            /* (direction_to_center + axis) * 0.5 : Vec2 * double */
            v2 t = { (float)((double)(dc.x + axis.x) * 0.5), (float)((double)(dc.y + axis.y) * 0.5) };
            t = v2norm(t);
            s4[4 * (i - 1)] = p1.x + L[i] * t.x; s4[4 * (i - 1) + 1] = p1.y + L[i] * t.y;
        }
        free(L);
    }
    for (int i = 0; i < n / 2; ++i) for (int k = 0; k < 4; ++k) { float t = s4[4 * i + k]; s4[4 * i + k] = s4[4 * (n - 1 - i) + k]; s4[4 * (n - 1 - i) + k] = t; }   /* :1057 */
    return 0;
}
int oracle_midline_post_process(float* s4, int n, float stiff, int invert, int start_with_head) {      /* no movement information */
    return oracle_midline_post_process_mv(s4, n, stiff, invert, start_with_head, 0.f, 0.f, (int*)0);
}

int oracle_midline_normalize(const float* s4, int n, int resolution, float stiff, float* out4, oracle_midline_info* info) {   /* :1270-1454 */
    memset(info, 0, sizeof(*info));
    if (n < 2) { info->status = 1; return 1; }
    double len = 0.0;
    for (int i = 1; i < n; ++i) len += v2len(s4[4 * i] - s4[4 * (i - 1)], s4[4 * i + 1] - s4[4 * (i - 1) + 1]);
    if (len == 0.0) { info->status = 1; return 1; }
    const int max_segments = resolution - 1;
    const double step = len / (double)max_segments;
    float* red = (float*)malloc((size_t)(n + resolution + 8) * 4 * sizeof(float));
    int nr = 0, cap = n + resolution + 8;
    memcpy(red, s4, 4 * sizeof(float)); nr = 1;
    int index = 0;
    double last_pt_distance = 0.0, distance;
    for (distance = 0.0; distance <= len && index < n - 1;) {
        while (distance - last_pt_distance < step && index < n - 1) {
            const float local_d = v2len(s4[4 * (index + 1)] - s4[4 * index], s4[4 * (index + 1) + 1] - s4[4 * index + 1]);
            distance += local_d;
            index++;
        }
        float off = (float)(distance - last_pt_distance);
        if (off < step) break;
        while (off >= step) {
            off = (float)((double)off - step);
            if (nr >= cap) break;
            /* index > 0 always holds here (the inner while advanced it at least once or the loop broke) */
            const float* s0 = s4 + 4 * (index - 1); const float* s1 = s4 + 4 * index;
            const float lx = s1[0] - s0[0], ly = s1[1] - s0[1];
            const float local_d = v2len(lx, ly);
            float percent = off;
            if (local_d > 0) percent /= local_d;
            percent = 1.f - percent;
            float* o = red + 4 * nr++;
            o[0] = s0[0] + lx * percent; o[1] = s0[1] + ly * percent;
            o[2] = (float)((double)(s0[2] * percent) + (double)s1[2] * (1.0 - (double)percent));
            o[3] = s0[3] > s1[3] ? s0[3] : s1[3];
            const double q = 1.0 - (double)percent;
            last_pt_distance = distance - (double)v2len((float)((double)lx * q), (float)((double)ly * q));
        }
    }
    {
        const float dx = red[4 * (nr - 1)] - s4[4 * (n - 1)], dy = red[4 * (nr - 1) + 1] - s4[4 * (n - 1) + 1];
        if (v2len(dx, dy) >= 0.01f && nr < cap) { memcpy(red + 4 * nr, s4 + 4 * (n - 1), 4 * sizeof(float)); ++nr; }
    }
    info->n = nr;
    if (nr != resolution) { free(red); info->status = 2; return 2; }
    {
        float percent = v2len(red[4] - red[0], red[5] - red[1]);
        if (len > 0) percent = (float)((double)percent / len);
        red[2] = (float)((double)(red[4 + 2] * percent) + (double)red[2] * (1.0 - (double)percent));
    }
    len = 0.0;
    for (int i = 1; i < nr; ++i) len += v2len(red[4 * i] - red[4 * (i - 1)], red[4 * i + 1] - red[4 * (i - 1) + 1]);
    const float ang0 = midline_calculate_angle(red, nr, stiff);
    const float angle = (float)(-(double)ang0 + 3.14159265358979323846);          /* Float2_t(-a + M_PI): double add, narrowed */
    const float offx = red[4 * (nr - 1)], offy = red[4 * (nr - 1) + 1];
    /* tf.rotate(DEGREE(angle)); tf.translate(-offx,-offy) */
    const float deg = angle * 180.f / 3.14159265358979323846f;      /* DEGREE(), as in oracle_moments_transform */
    const float rad = deg * 3.141592654f / 180.f;
    const float c = cosf(rad), s = sinf(rad);
    const float m0 = c, m1 = -s, m2 = c * -offx + -s * -offy, m3 = s, m4 = c, m5 = s * -offx + c * -offy;
    float fx = 0, fy = 0;
    for (int i = nr - 1, k = 0; i >= 0; --i, ++k) {
        const float x = red[4 * i], y = red[4 * i + 1];
        float px = m0 * x + m1 * y + m2, py = m3 * x + m4 * y + m5;
        if (k == 0) { fx = px; fy = py; }
        out4[4 * k] = px - fx; out4[4 * k + 1] = py - fy; out4[4 * k + 2] = red[4 * i + 2]; out4[4 * k + 3] = red[4 * i + 3];
    }
    info->len = (float)len; info->angle = ang0; info->offx = offx; info->offy = offy;
    free(red);
    return 0;
}

/* Midline::transform(normalization) (Outline.cpp:1237-1255), front() = 0: translate(-front) . rotate(DEGREE(-angle + pi/4 | pi)) . translate(-offset) */
void oracle_midline_transform(float angle, float offx, float offy, int legacy, float* tr6) {
    const float a = (float)(-(double)angle + (legacy ? 3.14159265358979323846 : 3.14159265358979323846 * 0.25));
    const float deg = a * 180.f / 3.14159265358979323846f;
    const float rad = deg * 3.141592654f / 180.f;
    const float c = (float)cos((double)rad), s = (float)sin((double)rad);    /* correctly rounded, like trex_oracle.c aff_rotate_deg */
    tr6[0] = c; tr6[1] = -s; tr6[2] = c * -offx + -s * -offy;
    tr6[3] = s; tr6[4] = c;  tr6[5] = s * -offx + c * -offy;
}


/* ---- posture::calculate_posture with its threshold retry loop (Posture.cpp:305-399) -------------------------------------------
 * threshold = track_posture_threshold; repeat: biggest sub-blob of the blob at `threshold` (pixel::threshold_get_biggest_blob,
 * commons: restated as threshold_blob + largest pixel count, first wins ties; posture_closing_steps = 0), coordinates relative
 * to the ORIGINAL blob's bounds().pos() (:336), outline -> resample -> calculate_midline; success returns at once, otherwise
 * threshold += 2 until the thresholded blob has fewer than max(1, initial/10) pixels or threshold >= start + 100.  When no
 * threshold works the first outline that could be traced is returned without a midline (:383-391).
 * info->status: 0 ok, else the status of the LAST attempt; *threshold_used = threshold of the returned result (success) or of the
 * first outline (fallback) or -1; *iterations = attempts made. */
int oracle_posture_auto(const oracle_run* runs, int32_t n_runs, const uint8_t* pixels, const uint8_t* bg, int32_t bg_stride,
                        int32_t width, int32_t height, int32_t method, int32_t start_threshold, const oracle_posture_params* P,
                        float* outline_xy, float* segments, oracle_posture_info* info, int32_t* threshold_used, int32_t* iterations) {
    memset(info, 0, sizeof(*info));
    *threshold_used = -1; *iterations = 0;
    if (n_runs <= 0) { info->status = 1; return 1; }
    int ox = 65535, oy = 65535; uint64_t initial = 0;
    for (int i = 0; i < n_runs; ++i) {
        if (runs[i].x0 < ox) ox = runs[i].x0;
        if (runs[i].y < oy) oy = runs[i].y;
        initial += (uint64_t)(runs[i].x1 - runs[i].x0 + 1);
    }
    const uint64_t minimum = initial / 10u > 1u ? initial / 10u : 1u;
    float* first = NULL; int first_n = 0, first_thr = -1;
    int threshold = start_threshold, rc = 1;
    oracle_posture_info last; memset(&last, 0, sizeof(last)); last.status = 1;
    for (;;) {
        oracle_frame* fr = oracle_threshold_blob(runs, n_runs, pixels, bg, bg_stride, width, height, method, threshold, 8);
        int nb, nr, np;
        oracle_frame_counts(fr, &nb, &nr, &np);
        uint64_t count = 0;
        ++*iterations;
        if (nb > 0) {
            oracle_blob* bl = (oracle_blob*)malloc((size_t)nb * sizeof(oracle_blob));
            oracle_run* rr = (oracle_run*)malloc((size_t)(nr > 0 ? nr : 1) * sizeof(oracle_run));
            uint8_t* px = (uint8_t*)malloc((size_t)(np > 0 ? np : 1));
            oracle_frame_copy(fr, bl, rr, px);
            int best = 0;
            for (int k = 1; k < nb; ++k) if (bl[k].n_pixels > bl[best].n_pixels) best = k;
            count = bl[best].n_pixels;
            rc = oracle_posture(rr + bl[best].run_begin, (int32_t)bl[best].n_runs, ox, oy, P, outline_xy, segments, &last);
            free(bl); free(rr); free(px);
            if (rc == 0) { oracle_frame_free(fr); *info = last; *threshold_used = threshold; free(first); return 0; }
            if (!first && last.n_outline > 0) {
                first = (float*)malloc((size_t)last.n_outline * 2 * sizeof(float));
                memcpy(first, outline_xy, (size_t)last.n_outline * 2 * sizeof(float));
                first_n = last.n_outline; first_thr = threshold;
            }
        } else { rc = 1; memset(&last, 0, sizeof(last)); last.status = 1; }
        oracle_frame_free(fr);
        threshold += 2;
        if (count < minimum || threshold >= start_threshold + 100) break;
    }
    *info = last;
    info->n_segments = 0;
    info->n_outline = 0;
    if (first) {
        memcpy(outline_xy, first, (size_t)first_n * 2 * sizeof(float));
        info->n_outline = first_n; *threshold_used = first_thr;
        free(first);
    }
    if (info->status == 0) info->status = 1;
    return info->status;
}
