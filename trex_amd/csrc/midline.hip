// midline.hip -- Midline::post_process (with or without MovementInformation::direction: the movement-history flip of Outline.cpp:905-961) and
// Midline::normalize() for every blob of a posture call (Outline.cpp:895-1060, 1270-1454; call site
// Individual.cpp:1369-1372), plus the posture / legacy crop transforms built from the result (Outline.cpp:1237-1255).
// One lane per blob: the work is a serial walk over <= max_points/2+1 segments and 25 output points.
// Segments are post-processed IN PLACE (the head part is straightened); the normalised midline (`resolution` points,
// head at the origin, rotated by -angle+pi) goes to `mid`.  Float2_t = float with the reference's double accumulators;
// compiled without FMA contraction like posture.hip.
#include "internal.h"
#include "affine.h"
#include <cmath>
#include <vector>

namespace trexhip {

struct MidlineCfg { int resolution; float stiff; int invert, start_with_head, stride; };

static constexpr int M_CAP = 48;        // segments per blob worked on in LDS (48 x 64 lanes x 16 B = 48 KB per workgroup)
__device__ __forceinline__ float vlen(float x, float y) { return sqrtf(x * x + y * y); }
__device__ __forceinline__ float2 vnorm(float x, float y) { const float L = vlen(x, y); return L > 0 ? make_float2((x / L), (y / L)) : make_float2(0.f, 0.f); }

__global__ __launch_bounds__(64) void k_midline(const MidlineCfg C, const trexhip_posture_info* __restrict__ pinfo, float4* __restrict__ segs,
                                                int n_blobs, float4* __restrict__ mid, trexhip_midline_info* __restrict__ minfo,
                                                const float2* __restrict__ move_dir /* MovementInformation::direction per blob, or null */) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_blobs) return;
    trexhip_midline_info I = {};
    float4* out = mid + (size_t)b * C.resolution;
    const trexhip_posture_info pi = pinfo[b];
    const int n = pi.n_segments;
    if (pi.status != 0 || n <= 2) { I.status = 1; minfo[b] = I; return; }
    // The algorithm below walks the blob's segment list about six times, one lane per blob: done in global memory every access of a wave
    // touches 64 cache lines.  Lists of at most M_CAP segments are worked on in the lane's LDS column (one pass in, one pass out).
    float4* Sg = segs + (size_t)b * C.stride;
    __shared__ float4 s_seg[M_CAP * 64];
    const bool in_lds = n <= M_CAP;
    if (in_lds) for (int i = 0; i < n; ++i) s_seg[i * 64 + threadIdx.x] = Sg[i];
    const int sstride = in_lds ? 64 : 1;
    float4* Sb = in_lds ? s_seg + threadIdx.x : Sg;          // generic pointer: LDS column or the global list
#define S(i_) Sb[(i_) * sstride]
    // the movement-history flip (Outline.cpp:905-961): with movement information, a midline whose direction (mean of its first
    // max(1, n * stiff) steps, Midline::midline_direction :870-887) points against the movement is turned round
    bool needs_invert = !C.invert;
    if (move_dir) {
        const float2 mv = move_dir[b];
        if (mv.x != 0.f || mv.y != 0.f) {
            const float sf = (float)n * C.stiff;
            const long samples = (long)(sf > 1.f ? sf : 1.f);
            float dx = 0.f, dy = 0.f;
            long counted = 0;
            for (long i = 0; i < samples && i + 1 < (long)n; ++i, ++counted) { const float4 a = S(i), q = S(i + 1); dx += q.x - a.x; dy += q.y - a.y; }
            if (counted > 0) { dx = dx / (float)counted; dy = dy / (float)counted; const float2 dn = vnorm(dx, dy); dx = dn.x; dy = dn.y; }
            if (!needs_invert) { dx = -dx; dy = -dy; }
            const float against = (-dx) * mv.x + (-dy) * mv.y, along = dx * mv.x + dy * mv.y;
            if (acosf(against) < acosf(along)) { needs_invert = !needs_invert; I.reserved[0] = 1; }      // `_inverted_because_previous`: the caller swaps head / tail index
        }
    }
    const bool rev = needs_invert ? (C.start_with_head == 0) : (C.start_with_head != 0);
#define PS(i) S(rev ? n - 1 - (i) : (i))
    if (C.stiff > 0) {
        float cf = roundf((float)n * C.stiff) + 1.f; if ((float)n - 1.f < cf) cf = (float)n - 1.f;
        const int center = (int)cf;
        const float4 cpt = PS(center);
        double eo = (double)center + ((double)n * 0.1 > 0.0 ? (double)n * 0.1 : 0.0); if ((double)n < eo) eo = (double)n;
        const int extra = (int)eo;
        float ax = 0.f, ay = 0.f; unsigned count = 0;
        for (int i = center; i < extra && i + 1 < n; ++i) {
            const float4 a = PS(i), q = PS(i + 1);
            const float2 d = vnorm(a.x - q.x, a.y - q.y);
            ax += d.x; ay += d.y; ++count;
        }
        if (count > 0) { ax = (ax / (float)count); ay = (ay / (float)count); }
        float ox = cpt.x, oy = cpt.y;          // original position of segment i
        float px = cpt.x, py = cpt.y;          // current (possibly moved) position of segment i
        for (int i = center; i > 0; --i) {
            float4 p0 = PS(i - 1);
            const float L = vlen(ox - p0.x, oy - p0.y);
            const float2 dc = vnorm(p0.x - cpt.x, p0.y - cpt.y);
            const float2 t = vnorm((float)((double)(dc.x + ax) * 0.5), (float)((double)(dc.y + ay) * 0.5));
            ox = p0.x; oy = p0.y;
            p0.x = px + L * t.x; p0.y = py + L * t.y;
            PS(i - 1) = p0;
            px = p0.x; py = p0.y;
        }
    }
#undef PS
    if (!rev) for (int i = 0; i < n / 2; ++i) { const float4 t = S(i); S(i) = S(n - 1 - i); S(n - 1 - i) = t; }
    if (in_lds) for (int i = 0; i < n; ++i) Sg[i] = S(i);     // the post-processed list is an output (in place)
    // ---- normalize --------------------------------------------------------------------------------
    double len = 0.0;
    {
        float4 a = S(0);
        for (int i = 1; i < n; ++i) { const float4 q = S(i); len += (double)vlen(q.x - a.x, q.y - a.y); a = q; }
    }
    if (len == 0.0) { I.status = 1; minfo[b] = I; return; }
    const int R = C.resolution;
    const double step = len / (double)(R - 1);
    int nr = 1, index = 0;
    float4 last = S(0);
    out[0] = last;
    double last_pt_distance = 0.0, distance = 0.0;
    while (distance <= len && index < n - 1) {
        while (distance - last_pt_distance < step && index < n - 1) {
            const float4 a = S(index), q = S(index + 1);
            distance += (double)vlen(q.x - a.x, q.y - a.y);
            index++;
        }
        float off = (float)(distance - last_pt_distance);
        if ((double)off < step) break;
        const float4 s0 = S(index - 1), s1 = S(index);
        const float lx = s1.x - s0.x, ly = s1.y - s0.y;
        const float local_d = vlen(lx, ly);
        while ((double)off >= step) {
            off = (float)((double)off - step);
            float percent = off;
            if (local_d > 0) percent = (percent / local_d);
            percent = 1.f - percent;
            float4 o;
            o.x = s0.x + lx * percent; o.y = s0.y + ly * percent;
            o.z = (float)((double)(s0.z * percent) + (double)s1.z * (1.0 - (double)percent));
            o.w = s0.w > s1.w ? s0.w : s1.w;
            if (nr < R) out[nr] = o;
            ++nr; last = o;
            const double q = 1.0 - (double)percent;
            last_pt_distance = distance - (double)vlen((float)((double)lx * q), (float)((double)ly * q));
            if (nr > n + R + 8) break;
        }
    }
    {
        const float4 e = S(n - 1);
        if (vlen(last.x - e.x, last.y - e.y) >= 0.01f) { if (nr < R) out[nr] = e; ++nr; }
    }
    I.n = nr;
    if (nr != R) { I.status = 2; minfo[b] = I; return; }
    {
        const float4 r0 = out[0], r1 = out[1];
        float percent = vlen(r1.x - r0.x, r1.y - r0.y);
        if (len > 0) percent = (float)((double)percent / len);
        out[0].z = (float)((double)(r1.z * percent) + (double)r0.z * (1.0 - (double)percent));
    }
    len = 0.0;
    {
        float4 a = out[0];
        for (int i = 1; i < R; ++i) { const float4 q = out[i]; len += (double)vlen(q.x - a.x, q.y - a.y); a = q; }
    }
    float ang0 = 0.f;
    {   // Midline::calculate_angle (Outline.cpp:1114-1124)
        float center = (float)(R - 2) - (float)R * C.stiff; if (center < 0) center = 0;
        const int start = (int)center;
        const float rest = center - (float)start;
        const int s1i = start + 1 < R ? start + 1 : R - 1;
        const float4 e = out[R - 1], a = out[start], q = out[s1i];
        const float lx = e.x - (a.x * (1 - rest) + q.x * rest), ly = e.y - (a.y * (1 - rest) + q.y * rest);
        ang0 = atan2f(ly, lx);
    }
    const float angle = (float)(-(double)ang0 + 3.14159265358979323846);
    const float4 A = out[R - 1];
    const float deg = angle * 180.f / 3.14159265358979323846f;
    const float rad = deg * 3.141592654f / 180.f;
    const float c = cosf(rad), s = sinf(rad);
    const float m2 = c * -A.x + -s * -A.y, m5 = s * -A.x + c * -A.y;
    // rotated = reduced reversed (head first), then shifted so that the front is the origin
    const float fx = c * A.x + -s * A.y + m2, fy = s * A.x + c * A.y + m5;
    for (int i = 0; i < (R + 1) / 2; ++i) {
        const float4 a = out[i], q = out[R - 1 - i];
        out[R - 1 - i] = make_float4((c * a.x + -s * a.y + m2) - fx, (s * a.x + c * a.y + m5) - fy, a.z, a.w);
        out[i] = make_float4((c * q.x + -s * q.y + m2) - fx, (s * q.x + c * q.y + m5) - fy, q.z, q.w);
    }
    I.len = (float)len; I.angle = ang0; I.offx = A.x; I.offy = A.y;
    minfo[b] = I;
#undef S
}

int launch_crops_warp_maps(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode);
int warp_reserve(trexhip_ctx* ctx, int n);
int check_colour_difference(trexhip_ctx* ctx, int difference, const char* who);

// per-blob inverse affine maps of the normalised crops, on the device (no host round trip): one thread per blob.
//   minfo != null: posture / legacy -- Midline::transform from the midline pose; blobs without a midline get a map that sends every
//                  output pixel far outside the image (all-zero crop; diff_image returns nullptr for them, FilterCache.cpp:268-270)
//   minfo == null: moments -- orientation from the integer moments of the blob table
// This file is compiled with -ffp-contract=off: the floats equal the host's and the CPU checker's.
__global__ __launch_bounds__(64) void k_warp_maps(const trexhip_midline_info* __restrict__ minfo, const float* __restrict__ lengths,
                                                  const trexhip_blob* __restrict__ blobs, const int n, const int legacy, const int OW, const int OH,
                                                  const float scale, double* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double* m = out + (size_t)i * 6;
    Aff tr;
    float len = 0.f;
    if (minfo) {
        const trexhip_midline_info mi = minfo[i];
        if (mi.status != 0) { m[0] = m[1] = m[3] = m[4] = 0; m[2] = m[5] = -1.0e6; return; }
        tr = midline_transform(mi.angle, mi.offx, mi.offy, legacy != 0);
        len = lengths ? lengths[i] : mi.len;
    } else {
        const trexhip_blob B = blobs[i];
        tr = moments_transform((float)B.n_pixels, (float)B.m10, (float)B.m01, (float)B.m20, (float)B.m11, (float)B.m02,
                               (float)(B.x1 - B.x0 + 1), (float)(B.y1 - B.y0 + 1));
    }
    compose_and_invert(tr, len, legacy != 0, OW, OH, scale, m);
}

int launch_crops_warp_device(trexhip_ctx* ctx, uint8_t* d_crops, int n, int OW, int OH, int diff_mode, const trexhip_midline_info* d_minfo,
                             const float* d_lengths, bool legacy, float scale) {
    if (int rc = check_colour_difference(ctx, diff_mode, "normalised crops")) return rc;
    if (int rc = warp_reserve(ctx, n)) return rc;
    hipLaunchKernelGGL(k_warp_maps, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_minfo, d_lengths, ctx->d_blobs, n, legacy ? 1 : 0, OW, OH, scale, ctx->d_warp);
    TH_CHECK_HIP(hipGetLastError());
    return launch_crops_warp_maps(ctx, d_crops, n, OW, OH, diff_mode);
}

}  // namespace trexhip

using namespace trexhip;

extern "C" int trexhip_midline_device(trexhip_ctx* ctx, const trexhip_midline_params* mp, int32_t n_blobs, int32_t max_points,
                                      const trexhip_posture_info* d_posture_info, float* d_segments, float* d_midline,
                                      trexhip_midline_info* d_midline_info) {
    return trexhip_midline_movement_device(ctx, mp, n_blobs, max_points, d_posture_info, d_segments, d_midline, d_midline_info, nullptr);
}

extern "C" int trexhip_midline_movement_device(trexhip_ctx* ctx, const trexhip_midline_params* mp, int32_t n_blobs, int32_t max_points,
                                               const trexhip_posture_info* d_posture_info, float* d_segments, float* d_midline,
                                               trexhip_midline_info* d_midline_info, const float* d_movement_direction) {
    if (!ctx || !mp || !d_posture_info || !d_segments || !d_midline || !d_midline_info) { set_error("trexhip_midline_device: null argument"); return TREXHIP_E_INVALID; }
    if (mp->midline_resolution < 3 || mp->midline_resolution > 256) { set_error("trexhip_midline_device: midline_resolution must be in 3..256"); return TREXHIP_E_INVALID; }
    if (!(mp->midline_stiff_percentage >= 0.f) || mp->midline_stiff_percentage >= 1.f) { set_error("trexhip_midline_device: midline_stiff_percentage must be in [0,1)"); return TREXHIP_E_INVALID; }
    if (max_points < 8 || max_points > 4096 || (max_points & 1)) { set_error("trexhip_midline_device: max_points must match the posture call"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0) { set_error("trexhip_midline_device: negative n_blobs"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    const MidlineCfg C{mp->midline_resolution, mp->midline_stiff_percentage, mp->midline_invert, mp->midline_start_with_head, max_points / 2 + 1};
    hipLaunchKernelGGL(k_midline, dim3((n_blobs + 63) / 64), dim3(64), 0, ctx->stream, C, d_posture_info, reinterpret_cast<float4*>(d_segments),
                       n_blobs, reinterpret_cast<float4*>(d_midline), d_midline_info, reinterpret_cast<const float2*>(d_movement_direction));
    TH_CHECK_HIP(hipGetLastError());
    return TREXHIP_OK;
}

extern "C" void trexhip_default_midline_params(trexhip_midline_params* p) {
    if (!p) return;
    p->midline_resolution = 25; p->midline_stiff_percentage = 0.15f; p->midline_invert = 0; p->midline_start_with_head = 0;
}

extern "C" int trexhip_crops_posture_device(trexhip_ctx* ctx, uint8_t* d_crops, int32_t n_blobs, int32_t out_w, int32_t out_h,
                                            const trexhip_midline_info* d_midline_info, const float* midline_lengths, int32_t use_legacy,
                                            float image_scale, int32_t difference) {
    if (!ctx || !d_crops || !d_midline_info) { set_error("trexhip_crops_posture_device: null argument"); return TREXHIP_E_INVALID; }
    if (out_w <= 0 || out_h <= 0 || difference < 0 || difference > 2) { set_error("trexhip_crops_posture_device: bad argument"); return TREXHIP_E_INVALID; }
    if (!ctx->d_frames || ctx->last_n == 0 || !ctx->fetched) { set_error("trexhip_crops_posture_device: segment and fetch a batch first"); return TREXHIP_E_INVALID; }
    if (n_blobs < 0 || (uint32_t)n_blobs > ctx->cfg.pool_blobs) { set_error("trexhip_crops_posture_device: n_blobs outside the blob pool"); return TREXHIP_E_INVALID; }
    if (n_blobs == 0) return TREXHIP_OK;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    // the transforms are built on the device from the midline poses (k_warp_maps): no copy of the poses to the host, no stream sync.
    // The optional per-blob (median) midline lengths are a host array: they are uploaded into a buffer of the context.
    const float* d_len = nullptr;
    if (midline_lengths) {
        if (ctx->len_cap < n_blobs) {
            if (ctx->d_len) (void)hipFree(ctx->d_len);
            ctx->d_len = nullptr; ctx->len_cap = 0;
            TH_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->d_len), (size_t)n_blobs * sizeof(float)));
            ctx->len_cap = n_blobs;
        }
        TH_CHECK_HIP(hipMemcpyAsync(ctx->d_len, midline_lengths, (size_t)n_blobs * sizeof(float), hipMemcpyHostToDevice, ctx->stream));   // pageable source: staged before the call returns
        d_len = ctx->d_len;
    }
    return launch_crops_warp_device(ctx, d_crops, n_blobs, out_w, out_h, difference, d_midline_info, d_len, use_legacy != 0, image_scale);
}
