// affine.h -- gui::Transform as normalize_image / Midline::transform use it (FilterCache.cpp:50-63, Outline.cpp:1237-1255): the
// SFML-style 2x3 affine in float (post-multiply), shared by the host path (caller-supplied transforms) and the device kernel that
// builds the per-blob inverse maps (midline.hip, compiled without FMA contraction so that both give the same floats).
// cos / sin / atan2 are the correctly rounded float results (computed in double and narrowed): the reference's libm is not part of
// any parity contract, and a correctly rounded value is the one definition that the CPU checker, the host path and the device can all meet.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

namespace trexhip {

struct Aff { float m[6]; };   // [m0 m1 m2; m3 m4 m5]

__host__ __device__ inline Aff aff_mul(const Aff& a, const Aff& b) {
    Aff c;
    c.m[0] = a.m[0] * b.m[0] + a.m[1] * b.m[3]; c.m[1] = a.m[0] * b.m[1] + a.m[1] * b.m[4]; c.m[2] = a.m[0] * b.m[2] + a.m[1] * b.m[5] + a.m[2];
    c.m[3] = a.m[3] * b.m[0] + a.m[4] * b.m[3]; c.m[4] = a.m[3] * b.m[1] + a.m[4] * b.m[4]; c.m[5] = a.m[3] * b.m[2] + a.m[4] * b.m[5] + a.m[5];
    return c;
}
__host__ __device__ inline Aff aff_translate(const Aff& a, float x, float y) { const Aff t = {{1, 0, x, 0, 1, y}}; return aff_mul(a, t); }
__host__ __device__ inline Aff aff_scale(const Aff& a, float s) { const Aff t = {{s, 0, 0, 0, s, 0}}; return aff_mul(a, t); }
__host__ __device__ inline float cos_cr(float x) { return (float)cos((double)x); }
__host__ __device__ inline float sin_cr(float x) { return (float)sin((double)x); }
__host__ __device__ inline float atan2_cr(float y, float x) { return (float)atan2((double)y, (double)x); }
__host__ __device__ inline Aff aff_rotate_deg(const Aff& a, float deg) {
    const float rad = deg * 3.141592654f / 180.f;
    const float c = cos_cr(rad), s = sin_cr(rad);
    const Aff t = {{c, -s, 0, s, c, 0}};
    return aff_mul(a, t);
}

// Midline::transform(posture | legacy) with front() = 0 (never set outside the legacy file reader, Output.cpp:406):
// rotate(DEGREE(-angle + pi/4 | pi)) . translate(-offset)
__host__ __device__ inline Aff midline_transform(float angle, float offx, float offy, bool legacy) {
    const float a = (float)(-(double)angle + (legacy ? 3.14159265358979323846 : 3.14159265358979323846 * 0.25));
    const float deg = a * 180.f / 3.14159265358979323846f;
    const float rad = deg * 3.141592654f / 180.f;
    const float c = cos_cr(rad), s = sin_cr(rad);
    Aff t;
    t.m[0] = c; t.m[1] = -s; t.m[2] = c * -offx + -s * -offy;
    t.m[3] = s; t.m[4] = c;  t.m[5] = s * -offx + c * -offy;
    return t;
}

// individual_image_normalization = moments (FilterCache.cpp:276-288): rotate(DEGREE(-orientation + pi/4)) . translate(-size/2),
// pv::Blob::orientation() = 0.5 * atan2(2 mu11, mu20 - mu02) from the blob's central moments
__host__ __device__ inline Aff moments_transform(float n_pixels, float m10, float m01, float m20, float m11, float m02, float bw, float bh) {
    const float cx = m10 / n_pixels, cy = m01 / n_pixels;
    const float mu20 = m20 / n_pixels - cx * cx, mu02 = m02 / n_pixels - cy * cy, mu11 = m11 / n_pixels - cx * cy;
    const float orientation = 0.5f * atan2_cr(2.f * mu11, mu20 - mu02);
    const float angle = (-orientation + 3.14159265358979323846f * 0.25f) * 180.f / 3.14159265358979323846f;
    Aff t = {{1, 0, 0, 0, 1, 0}};
    t = aff_rotate_deg(t, angle);
    return aff_translate(t, -bw * 0.5f, -bh * 0.5f);
}

// forward transform of normalize_image (FilterCache.cpp:50-63) and its inverse as cv::warpAffine computes it (double)
__host__ __device__ inline void compose_and_invert(const Aff& tr, float midline_length, bool legacy, int OW, int OH, float scale, double* out6) {
    Aff t = {{1, 0, 0, 0, 1, 0}};
    t = aff_translate(t, (float)OW * 0.5f, (float)OH * 0.5f);
    t = aff_scale(t, scale);
    if (legacy) t = aff_translate(t, -midline_length * 0.5f, 0.f);
    else        t = aff_translate(t, midline_length * 0.4f, midline_length * 0.4f);
    t = aff_mul(t, tr);
    double M[6];
    for (int i = 0; i < 6; ++i) M[i] = t.m[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; ++i) out6[i] = M[i];
}

}  // namespace trexhip
