// conv_f32.h -- 5x5 'same' convolution as 25 shifted GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
// Shared by the inference path's fp32 mode (cnn.hip: folded BN + ReLU + 2x2 max-pool in the epilogue) and by the training step
// (train.hip: raw convolution output + bias; the same kernel with flipped / transposed weights is the data gradient).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trexhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { CONV_EPI_POOL = 0, CONV_EPI_RAW = 1 };

// ------------------------------------------------------------------------------------------------
// conv 5x5 'same' + folded BN + ReLU + maxpool2 as 25 shifted GEMMs on fp32 MFMA
// ------------------------------------------------------------------------------------------------
template <int CI, int CO, int S, int ROWS, int CIC>
struct ConvGeom {
    static constexpr int PW = S + 4, PH = ROWS + 4;
    static constexpr int STR = CIC + 1;                               // odd pixel stride: conflict-free b32 reads
    static constexpr int RP0 = PW * STR;
    static constexpr int RP = RP0 + ((16 - (RP0 % 32)) + 32) % 32;    // row pitch == 16 (mod 32): rows y, y+1 use disjoint banks
    static constexpr int PATCH = PH * RP;                             // floats
    static constexpr int BT = CIC * CO;                               // floats per weight tile
    static constexpr int NPIX = ROWS * S;
    static constexpr int MT = (NPIX + 31) / 32;
    static constexpr int NT = CO / 32;
    static constexpr int WM = 8 / NT;                                 // wave groups along M
    static constexpr int TPW = (MT + WM - 1) / WM;                    // M tiles per wave
    static constexpr int LDS_BYTES = (PATCH + 2 * BT) * 4;
    static constexpr int BPC = S / ROWS;                              // blocks per crop
};

template <int CI, int CO, int S, int ROWS, int CIC, int EPI = CONV_EPI_POOL, int COUT = CO>
__global__ __launch_bounds__(512) void k_conv5(const float* __restrict__ in /*[N][S][S][CI]*/,
                                               const float* __restrict__ wp /*[CI/CIC][25][CIC][CO]*/,
                                               const float* __restrict__ bias /*[CO]*/,
                                               float* __restrict__ out /*POOL: [N][S/2][S/2][CO]; RAW: [N][S][S][COUT]*/) {
    using G = ConvGeom<CI, CO, S, ROWS, CIC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;
    float* Bs = lds + G::PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = wave % G::NT, mg = wave / G::NT;
    const int crop = blockIdx.x / G::BPC, row0 = (blockIdx.x % G::BPC) * ROWS;
    constexpr int WR = S / 2;                                          // pool windows per row

    int aoff[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) {
        const int p = (mg + G::WM * m) * 32 + j;                       // window-major pixel index
        int off = 0;
        if (p < G::NPIX) {
            const int wi = p >> 2, sub = p & 3;
            const int wy = wi / WR, wx = wi - wy * WR;
            off = (2 * wy + (sub >> 1)) * G::RP + (2 * wx + (sub & 1)) * G::STR;
        }
        aoff[m] = off + h;
    }
    f32x16 acc[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const float* inc = in + (size_t)crop * S * S * CI;
    constexpr int Q = CIC / 4;                                          // float4 per pixel per chunk
    constexpr int BV = G::BT / 4;                                       // float4 per weight tile
    constexpr int BPT = (BV + 511) / 512;                               // float4 per thread per weight tile
    for (int cc = 0; cc < CI / CIC; ++cc) {
        __syncthreads();                                                // previous chunk's readers are done
        for (int idx = tid; idx < G::PH * G::PW * Q; idx += 512) {
            const int q = idx % Q, px = idx / Q;
            const int py = px / G::PW, pxx = px - py * G::PW;
            const int iy = row0 + py - 2, ix = pxx - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                 // zero padding ('same')
            if (iy >= 0 && iy < S && ix >= 0 && ix < S)
                v = *reinterpret_cast<const float4*>(inc + ((size_t)iy * S + ix) * CI + cc * CIC + q * 4);
            float* d = patch + py * G::RP + pxx * G::STR + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        const float4* wsrc = reinterpret_cast<const float4*>(wp + (size_t)cc * 25 * G::BT);
        for (int i = tid; i < BV; i += 512) reinterpret_cast<float4*>(Bs)[i] = wsrc[i];
        __syncthreads();
        for (int tap = 0; tap < 25; ++tap) {
            const int buf = tap & 1;
            float4 nb[BPT];
            if (tap < 24) {
#pragma unroll
                for (int u = 0; u < BPT; ++u) {
                    const int i = tid + u * 512;
                    if (i < BV) nb[u] = wsrc[(size_t)(tap + 1) * BV + i];
                }
            }
            const int tapoff = (tap / 5) * G::RP + (tap % 5) * G::STR;
            const float* bsrc = Bs + buf * G::BT + h * CO + n * 32 + j;
            const float* asrc = patch + tapoff;
#pragma unroll
            for (int t = 0; t < CIC / 2; ++t) {
                const float b = bsrc[2 * t * CO];
#pragma unroll
                for (int m = 0; m < G::TPW; ++m) {
                    const float a = asrc[aoff[m] + 2 * t];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
                }
            }
            if (tap < 24) {
#pragma unroll
                for (int u = 0; u < BPT; ++u) {
                    const int i = tid + u * 512;
                    if (i < BV) reinterpret_cast<float4*>(Bs + (buf ^ 1) * G::BT)[i] = nb[u];
                }
            }
            __syncthreads();
        }
    }
    // epilogue: lane (j,h) holds for g=0..3 the 4 pixels of pool window (tile*8 + 2g + h), channel n*32+j
    const int co = n * 32 + j;
    if constexpr (EPI == CONV_EPI_RAW) {
        // raw output (+ bias when given): accumulator 4g+q of window wi is pixel (2 wy + (q >> 1), 2 wx + (q & 1))
        if (co >= COUT) return;
        const float bz = bias ? bias[co] : 0.f;
        float* oc = out + (size_t)crop * S * S * COUT;
#pragma unroll
        for (int m = 0; m < G::TPW; ++m) {
            const int mt = mg + G::WM * m;
            if (mt >= G::MT) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int wi = mt * 8 + 2 * g + h;
                if (wi >= G::NPIX / 4) continue;
                const int wy = wi / WR, wx = wi % WR;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    oc[((size_t)(row0 + 2 * wy + (q >> 1)) * S + 2 * wx + (q & 1)) * COUT + co] = acc[m][4 * g + q] + bz;
            }
        }
    } else {
        const float bz = bias[co];
        float* oc = out + (size_t)crop * WR * WR * CO;
#pragma unroll
        for (int m = 0; m < G::TPW; ++m) {
            const int mt = mg + G::WM * m;
            if (mt >= G::MT) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int wi = mt * 8 + 2 * g + h;
                if (wi >= G::NPIX / 4) continue;
                const float v = fmaxf(fmaxf(acc[m][4 * g], acc[m][4 * g + 1]), fmaxf(acc[m][4 * g + 2], acc[m][4 * g + 3]));
                const int wy = row0 / 2 + wi / WR, wx = wi % WR;
                oc[((size_t)wy * WR + wx) * CO + co] = fmaxf(v + bz, 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the same convolution for 16 output channels (conv2's data gradient) on v_mfma_f32_16x16x4_f32: 16 pixels x 16 channels per tile, so
// that no half of a 32-wide tile multiplies padding.  Raw output only.  Lane l of a step: A = pixel l % 16, input channel 4 t + l / 16;
// B = the same input channel, output channel l % 16; accumulator r = pixel 4 (l / 16) + r, channel l % 16.
// ------------------------------------------------------------------------------------------------
typedef float f32x4a __attribute__((ext_vector_type(4)));

template <int CI, int S, int ROWS, int CIC>
struct ConvGeom16 {
    static constexpr int PW = S + 4, PH = ROWS + 4;
    static constexpr int STR = CIC + 1;                               // odd pixel stride
    static constexpr int RP = PW * STR;
    static constexpr int PATCH = PH * RP;
    static constexpr int BT = CIC * 16;
    static constexpr int NPIX = ROWS * S;
    static constexpr int MT = NPIX / 16, TPW = (MT + 7) / 8;
    static constexpr int LDS_BYTES = (PATCH + 2 * BT) * 4;
    static constexpr int BPC = S / ROWS;
    static_assert(NPIX % 16 == 0 && S % ROWS == 0 && CIC % 4 == 0 && BT / 4 <= 512, "shapes");
};

template <int CI, int S, int ROWS, int CIC>
__global__ __launch_bounds__(512) void k_conv5_n16(const float* __restrict__ in /*[N][S][S][CI]*/, const float* __restrict__ wp /*[CI/CIC][25][CIC][16]*/,
                                                   float* __restrict__ out /*[N][S][S][16]*/) {
    using G = ConvGeom16<CI, S, ROWS, CIC>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;
    float* Bs = lds + G::PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int crop = blockIdx.x / G::BPC, row0 = (blockIdx.x % G::BPC) * ROWS;
    int aoff[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) {
        int p = (wave + 8 * m) * 16 + i16;
        p = p < G::NPIX ? p : G::NPIX - 1;
        aoff[m] = (p / S) * G::RP + (p % S) * G::STR + kq;
    }
    f32x4a acc[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) acc[m] = f32x4a{0.f, 0.f, 0.f, 0.f};
    const float* inc = in + (size_t)crop * S * S * CI;
    constexpr int Q = CIC / 4, BV = G::BT / 4;
    for (int cc = 0; cc < CI / CIC; ++cc) {
        __syncthreads();                                                // previous chunk's readers are done
        for (int idx = tid; idx < G::PH * G::PW * Q; idx += 512) {
            const int q = idx % Q, px = idx / Q;
            const int py = px / G::PW, pxx = px - py * G::PW;
            const int iy = row0 + py - 2, ix = pxx - 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                 // zero padding ('same')
            if (iy >= 0 && iy < S && ix >= 0 && ix < S)
                v = *reinterpret_cast<const float4*>(inc + ((size_t)iy * S + ix) * CI + cc * CIC + q * 4);
            float* d = patch + py * G::RP + pxx * G::STR + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        const float4* wsrc = reinterpret_cast<const float4*>(wp + (size_t)cc * 25 * G::BT);
        if (tid < BV) reinterpret_cast<float4*>(Bs)[tid] = wsrc[tid];
        __syncthreads();
        for (int tap = 0; tap < 25; ++tap) {
            const int buf = tap & 1;
            float4 nb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tap < 24 && tid < BV) nb = wsrc[(size_t)(tap + 1) * BV + tid];
            const float* bsrc = Bs + buf * G::BT + kq * 16 + i16;
            const float* asrc = patch + (tap / 5) * G::RP + (tap % 5) * G::STR;
#pragma unroll
            for (int t = 0; t < CIC / 4; ++t) {
                const float b = bsrc[4 * t * 16];
#pragma unroll
                for (int m = 0; m < G::TPW; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(asrc[aoff[m] + 4 * t], b, acc[m], 0, 0, 0);
            }
            if (tap < 24 && tid < BV) reinterpret_cast<float4*>(Bs + (buf ^ 1) * G::BT)[tid] = nb;
            __syncthreads();
        }
    }
    float* oc = out + ((size_t)crop * S * S + (size_t)row0 * S) * 16;
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) {
        const int mt = wave + 8 * m;
        if (mt >= G::MT) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) oc[(size_t)(mt * 16 + 4 * kq + r) * 16 + i16] = acc[m][r];
    }
}

}  // namespace trexhip
