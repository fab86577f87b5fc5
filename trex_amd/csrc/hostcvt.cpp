// hostcvt.cpp -- host side of the as-deployed boundary (plain C++, no device code): cv::cvtColor(BGR2GRAY / BGRA2GRAY) or the
// color_channel pick of BackgroundSubtraction.cpp:162-180, applied by the upload threads WHILE they move a pageable tile into the pinned
// ring.  The threads read every byte of the tile anyway; writing one byte per pixel instead of 3 / 4 cuts the PCIe transfer -- the
// bound of that path -- to a third / a quarter.  Same 14-bit fixed-point formula as the device kernel k_to_gray (crops.hip), i.e. as
// OpenCV's 8-bit path: (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14.  Only for the gray / binary pixel encodings; the colour
// encodings need the colour tile in HBM and keep the device-side reduction.
#include <cstddef>
#include <cstdint>

namespace {

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
        if (channels == 3) to_gray<3>(src, dst, npix); else to_gray<4>(src, dst, npix);
    }
}
