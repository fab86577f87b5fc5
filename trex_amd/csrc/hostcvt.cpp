// hostcvt.cpp -- host side of the as-deployed boundary (plain C++, no device code): cv::cvtColor(BGR2GRAY / BGRA2GRAY) or the
// color_channel pick of BackgroundSubtraction.cpp:162-180, applied by the upload threads WHILE they move a pageable tile into the pinned
// ring.  The threads read every byte of the tile anyway; writing one byte per pixel instead of 3 / 4 cuts the PCIe transfer -- the
// bound of that path -- to a third / a quarter.  Same 14-bit fixed-point formula as the device kernel k_to_gray (crops.hip), i.e. as
// OpenCV's 8-bit path: (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14.  Only for the gray / binary pixel encodings; the colour
// encodings need the colour tile in HBM and keep the device-side reduction.
#include <cstddef>
#include <cstdint>
#include <immintrin.h>

namespace {
bool use_nt();

// BGRA -> gray with explicit SIMD (the compiler's vectorisation of the 4-byte-strided scalar loop below runs at a fraction of a core's
// memory bandwidth, and the upload threads' copy leg is what bounds the as-deployed path).  Same arithmetic bit for bit: bytes widened to
// 16 bit, pmaddwd against (1868, 9617, 4899, 0) gives B*1868 + G*9617 and R*4899 per pixel as two 32-bit sums, added, + 8192, >> 14.
__attribute__((target("avx512f,avx512bw")))
size_t to_gray4_avx512(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t n) {
    const bool nt = (reinterpret_cast<uintptr_t>(d) & 15) == 0 && n >= 1024 && use_nt();
    const __m512i w = _mm512_set1_epi64((long long)((uint64_t)1868 | ((uint64_t)9617 << 16) | ((uint64_t)4899 << 32)));
    const __m512i rnd = _mm512_set1_epi64(8192);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m512i v = _mm512_loadu_si512(s + 4 * i);                              // 16 pixels
        const __m512i lo = _mm512_cvtepu8_epi16(_mm512_castsi512_si256(v));           // pixels 0..7 as 16-bit B G R A
        const __m512i hi = _mm512_cvtepu8_epi16(_mm512_extracti64x4_epi64(v, 1));     // pixels 8..15
        __m512i a = _mm512_madd_epi16(lo, w), b = _mm512_madd_epi16(hi, w);           // per pixel: (B*1868 + G*9617, R*4899)
        a = _mm512_add_epi64(_mm512_and_si512(a, _mm512_set1_epi64(0xffffffffll)), _mm512_srli_epi64(a, 32));
        b = _mm512_add_epi64(_mm512_and_si512(b, _mm512_set1_epi64(0xffffffffll)), _mm512_srli_epi64(b, 32));
        a = _mm512_srli_epi64(_mm512_add_epi64(a, rnd), 14);
        b = _mm512_srli_epi64(_mm512_add_epi64(b, rnd), 14);
        const __m128i q = _mm_unpacklo_epi64(_mm512_cvtepi64_epi8(a), _mm512_cvtepi64_epi8(b));
        // the pinned ring is written once and read by the DMA engine only: streaming stores skip the read-for-ownership of the line
        if (nt) _mm_stream_si128(reinterpret_cast<__m128i*>(d + i), q); else _mm_storeu_si128(reinterpret_cast<__m128i*>(d + i), q);
    }
    if (nt) _mm_sfence();
    return i;
}

__attribute__((target("avx2")))
size_t to_gray4_avx2(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t n) {
    const bool nt = (reinterpret_cast<uintptr_t>(d) & 15) == 0 && n >= 1024 && use_nt();
    const __m256i w = _mm256_set1_epi64x((long long)((uint64_t)1868 | ((uint64_t)9617 << 16) | ((uint64_t)4899 << 32)));
    const __m256i rnd = _mm256_set1_epi32(8192);
    const __m256i order = _mm256_setr_epi32(0, 1, 4, 5, 2, 3, 6, 7);
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 4 * i));        // pixels 0..7
        const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + 4 * i + 32));   // pixels 8..15
        const __m256i a0 = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(v0)), w);       // px 0..3: two sums each
        const __m256i a1 = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_extracti128_si256(v0, 1)), w);  // px 4..7
        const __m256i b0 = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(v1)), w);
        const __m256i b1 = _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_extracti128_si256(v1, 1)), w);
        // hadd works inside the 128-bit halves: (px0 px1 px4 px5 | px2 px3 px6 px7) -> put back in order
        __m256i x = _mm256_permutevar8x32_epi32(_mm256_hadd_epi32(a0, a1), order);
        __m256i y = _mm256_permutevar8x32_epi32(_mm256_hadd_epi32(b0, b1), order);
        x = _mm256_srli_epi32(_mm256_add_epi32(x, rnd), 14);
        y = _mm256_srli_epi32(_mm256_add_epi32(y, rnd), 14);
        __m256i p = _mm256_packus_epi32(x, y);                     // 16-bit: (x0..3 y0..3 | x4..7 y4..7)
        p = _mm256_permute4x64_epi64(p, 0xd8);                     // (x0..3 x4..7 | y0..3 y4..7)
        const __m128i q = _mm_packus_epi16(_mm256_castsi256_si128(p), _mm256_extracti128_si256(p, 1));
        if (nt) _mm_stream_si128(reinterpret_cast<__m128i*>(d + i), q); else _mm_storeu_si128(reinterpret_cast<__m128i*>(d + i), q);
    }
    if (nt) _mm_sfence();
    return i;
}

enum { ISA_SCALAR = 0, ISA_AVX2 = 1, ISA_AVX512 = 2 };
}  // namespace
#include <cstdlib>
namespace {
bool use_nt() { static const bool v = [] { const char* e = std::getenv("TREXHIP_HOST_NT"); return !(e && std::atoi(e) == 0); }(); return v; }
int pick_isa() {
    static const int isa = [] {
        __builtin_cpu_init();
        if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw")) return (int)ISA_AVX512;
        if (__builtin_cpu_supports("avx2")) return (int)ISA_AVX2;
        return (int)ISA_SCALAR;
    }();
    return isa;
}

template <int CH>
__attribute__((target_clones("avx2", "default")))
void to_gray(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const uint32_t b = s[CH * i], g = s[CH * i + 1], r = s[CH * i + 2];
        d[i] = (uint8_t)((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14);
    }
}

template <int CH>
__attribute__((target_clones("avx2", "default")))
void pick(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t n, int c) {
    for (size_t i = 0; i < n; ++i) d[i] = s[CH * i + c];
}

}  // namespace

extern "C" void trexhip_host_reduce_row(const uint8_t* src, uint8_t* dst, size_t npix, int channels, int color_channel) {
    if (color_channel >= 0 && color_channel < channels) {
        if (channels == 3) pick<3>(src, dst, npix, color_channel); else pick<4>(src, dst, npix, color_channel);
    } else {
        if (channels == 3) { to_gray<3>(src, dst, npix); return; }
        size_t done = 0;
        const int isa = pick_isa();
        if (isa == ISA_AVX512) done = to_gray4_avx512(src, dst, npix);
        else if (isa == ISA_AVX2) done = to_gray4_avx2(src, dst, npix);
        to_gray<4>(src + 4 * done, dst + done, npix - done);
    }
}

// dev / test hook: the same reduction with a forced instruction set (0 scalar, 1 AVX2, 2 AVX-512), -1 when the CPU lacks it
extern "C" int trexhip_host_reduce_row_isa(const uint8_t* src, uint8_t* dst, size_t npix, int isa) {
    __builtin_cpu_init();
    size_t done = 0;
    if (isa == ISA_AVX512) { if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw"))) return -1; done = to_gray4_avx512(src, dst, npix); }
    else if (isa == ISA_AVX2) { if (!__builtin_cpu_supports("avx2")) return -1; done = to_gray4_avx2(src, dst, npix); }
    to_gray<4>(src + 4 * done, dst + done, npix - done);
    return 0;
}
