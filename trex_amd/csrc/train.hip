// train.hip -- one optimizer step of the identity network V118_3 on gfx950, fp32.
//
// Replaces the body of train()'s batch loop
//   Application/src/tracker/python/visual_recognition_torch.py:1137-1158
//       outputs = model(inputs); loss = CrossEntropyLoss(outputs, targets); loss.backward(); optimizer.step(); zero_grad()
//   (criterion / Adam(lr) :1420-1421; autocast + GradScaler are enabled for device == 'cuda' only, :1066-1072 -- fp32 is the
//   reference's own arithmetic everywhere else and the parity bar here)
// for the network visual_identification_network_torch.py:184-258 in TRAINING mode:
//   [conv5x5 'same' -> BatchNorm2d (batch statistics; running statistics updated, momentum 0.1) -> ReLU -> MaxPool2 ->
//    Dropout2d(0.05)] x3 -> flatten (NCHW order) -> fc1 -> LayerNorm(100) -> ReLU -> Dropout(0.05) -> fc2 -> cross entropy (mean).
// Inputs are what TRexImageDataset yields (:158-188): NHWC float32 in [0, 255], integer class labels.
//
// Data (all NHWC fp32 in HBM, sized for max_batch at creation): z_i = raw convolution outputs (kept for the backward pass, turned
// into dz_i in place), a_i = pooled + dropped activations, da_i = their gradients.  Parameters, gradients and the two Adam moments are
// four flat arrays with one layout (the kernels' layouts, not torch's: conv [ci chunk][tap][ci][co], fc1 [hw][c][o]), so the
// optimizer is one element-wise launch; import / export / read permute on the host.
// Kernels: convolutions and data gradients = k_conv5 (conv_f32.h, fp32 MFMA, raw epilogue; the data gradient is the same kernel on
// flipped + transposed weights); weight gradients = k_t_wgrad (fp32 MFMA, operands straight from L2: one MFMA k-step = two pixels,
// A = activations at the tap's shift, B = dz); batch-norm statistics and their backward sums in double; everything else VALU.
// Every reduction has a fixed order: two runs of a step give identical bits.
#include "internal.h"
#include "conv_f32.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <string>
#include <vector>

namespace trexhip {

enum { T_C1W, T_C1B, T_G1, T_BE1, T_RM1, T_RV1, T_C2W, T_C2B, T_G2, T_BE2, T_RM2, T_RV2, T_C3W, T_C3B, T_G3, T_BE3, T_RM3, T_RV3,
       T_F1W, T_F1B, T_LNG, T_LNB, T_F2W, T_F2B, T_COUNT };

static constexpr float EPS_BN = 1e-5f, EPS_LN = 1e-5f;

// ------------------------------------------------------------------------------------------------
// conv1 forward: [n][80][80][CH] float -> z1 [n][80][80][16] = conv + bias.  Block = 4 rows of one crop, one pixel per thread.
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(320) void k_t_conv1(const float* __restrict__ x, const float* __restrict__ w /*[CH][25][16]*/,
                                                 const float* __restrict__ b, float* __restrict__ z, double* __restrict__ stat_partial /*[block][2][16] or null*/) {
    __shared__ float xs[8 * 84 * CH];
    __shared__ __attribute__((aligned(16))) float ws[CH * 25 * 16];
    const int tid = threadIdx.x, crop = blockIdx.x / 20, row0 = (blockIdx.x % 20) * 4;
    for (int idx = tid; idx < 8 * 84 * CH; idx += 320) {
        const int c = idx % CH, px = (idx / CH) % 84, py = idx / (CH * 84);
        const int iy = row0 + py - 2, ix = px - 2;
        xs[idx] = (iy >= 0 && iy < 80 && ix >= 0 && ix < 80) ? x[(((size_t)crop * 80 + iy) * 80 + ix) * CH + c] : 0.f;
    }
    for (int idx = tid; idx < CH * 25 * 16; idx += 320) ws[idx] = w[idx];
    __syncthreads();
    const int y = tid / 80, xx = tid % 80;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll 1
    for (int c = 0; c < CH; ++c)
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky)
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const float v = xs[((y + ky) * 84 + xx + kx) * CH + c];
                const float4* wv = reinterpret_cast<const float4*>(ws + (c * 25 + ky * 5 + kx) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = wv[q];
                    acc[4 * q] += v * t.x; acc[4 * q + 1] += v * t.y; acc[4 * q + 2] += v * t.z; acc[4 * q + 3] += v * t.w;
                }
            }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] += b[k];
    float4* o = reinterpret_cast<float4*>(z + (((size_t)crop * 80 + row0 + y) * 80 + xx) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    if (!stat_partial) return;                                // (uniform)
    // the block's column sums and sums of squares for the BN that follows.  Per wave a reduce-scatter butterfly: at distance 32, 16, 8, 4 a lane
    // keeps half of its channels and hands the other half to its partner, then two plain steps: a lane ends with the sum of channel
    // 8 b5 + 4 b4 + 2 b3 + b2 (b = bits of its number; 17 exchanges per plane instead of 96).  The five waves are added in order through LDS
    __shared__ double sred[5][2][16];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = pl ? (double)acc[k] * acc[k] : (double)acc[k];
#define T_RS(N, D)                                                                                       \
        {                                                                                                \
            const bool hi = (lane & D) != 0;                                                             \
            _Pragma("unroll") for (int k = 0; k < N; ++k) {                                              \
                const double mine = hi ? v[k + N] : v[k], other = hi ? v[k] : v[k + N];                  \
                v[k] = mine + __shfl_xor(other, D);                                                      \
            }                                                                                            \
        }
        T_RS(8, 32) T_RS(4, 16) T_RS(2, 8) T_RS(1, 4)
#undef T_RS
        v[0] += __shfl_xor(v[0], 2);
        v[0] += __shfl_xor(v[0], 1);
        if ((lane & 3) == 0) sred[wave][pl][((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)] = v[0];
    }
    __syncthreads();
    if (tid < 32) {
        const int pl = tid >> 4, c = tid & 15;
        double t = 0;
#pragma unroll
        for (int wv = 0; wv < 5; ++wv) t += sred[wv][pl][c];
        stat_partial[((size_t)blockIdx.x * 2 + pl) * 16 + c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// Reductions over the batch (BN statistics, BN backward sums, bias gradients): every block of the producing kernel leaves its column sums
// as doubles in partial[block][NP][C]; k_t_red_finalize (one workgroup per channel: one L2 round trip, not a chain of them) adds them in
// a fixed order -- same bits every run -- and does what the sum is for.  (Forming the totals in the producing kernel's last block was
// tried: the device-scope fences it needs write back / invalidate the L2 per block and cost 25-110 us per launch.)
// ------------------------------------------------------------------------------------------------
enum { FIN_BN_STATS = 0, FIN_BN_BWD = 1, FIN_BIAS = 2 };
struct FinArgs {
    double count; float momentum;                         // FIN_BN_STATS
    float *o0, *o1, *o2, *o3;                             // FIN_BN_STATS: mean, invstd, run_mean, run_var; FIN_BN_BWD: sums[2][C], d_gamma, d_beta; FIN_BIAS: d_bias
    const int32_t* refused;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_t_red_finalize(const double* __restrict__ partial, const int nblocks, const int C, const FinArgs A) {
    constexpr int NP = MODE == FIN_BIAS ? 1 : 2;
    __shared__ double sh[NP][4];
    const int c = blockIdx.x, tid = threadIdx.x;
    double t[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) t[pl] = 0;
    for (int b = tid; b < nblocks; b += 256)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) t[pl] += partial[((size_t)b * NP + pl) * C + c];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) t[pl] += __shfl_xor(t[pl], d);
        if ((tid & 63) == 0) sh[pl][tid >> 6] = t[pl];
    }
    __syncthreads();
    if (tid) return;
    double tot[NP];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) tot[pl] = ((sh[pl][0] + sh[pl][1]) + sh[pl][2]) + sh[pl][3];
    if constexpr (MODE == FIN_BN_STATS) {                    // mean, invstd (biased variance, like the normalisation uses); running statistics with the unbiased one
        const double m = tot[0] / A.count;
        double var = tot[1] / A.count - m * m;
        if (var < 0) var = 0;
        A.o0[c] = (float)m;
        A.o1[c] = (float)(1.0 / sqrt(var + (double)EPS_BN));
        const double unbiased = A.count > 1 ? var * A.count / (A.count - 1) : var;
        if (*A.refused) return;                              // a refused step leaves the running statistics alone (k_t_check_targets)
        A.o2[c] = (1.f - A.momentum) * A.o2[c] + A.momentum * (float)m;
        A.o3[c] = (1.f - A.momentum) * A.o3[c] + A.momentum * (float)unbiased;
    } else if constexpr (MODE == FIN_BN_BWD) {               // sums[0][C] = sum dy (= d beta), sums[1][C] = sum dy x-hat (= d gamma)
        const float sd = (float)tot[0], sg = (float)tot[1];
        A.o0[c] = sd; A.o0[C + c] = sg;
        A.o2[c] = sd; A.o1[c] = sg;
    } else {
        A.o0[c] = (float)tot[0];
    }
}

// a block's column sums: the RL row-lanes of every channel added in fixed order -> partial[block][plane][C]
template <int C, int NP>
__device__ __forceinline__ void block_columns(const double (&acc)[NP][4], double* sh /*[256 / (C / 4) * C]*/, double* partial) {
    constexpr int LC = C / 4, RL = 256 / LC;
    const int tid = threadIdx.x, cl = tid % LC, rl = tid / LC;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[rl * C + cl * 4 + k] = acc[pl][k];
        __syncthreads();
        if (tid < C) {
            double t = 0;
            for (int r = 0; r < RL; ++r) t += sh[r * C + tid];
            partial[((size_t)blockIdx.x * NP + pl) * C + tid] = t;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// batch statistics of z[rows][C]: per-channel sum and sum of squares, double accumulation -> partial[block][2][C] (FIN_BN_STATS)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_t_bn_stats(const float* __restrict__ d, size_t rows, double* __restrict__ partial) {
    constexpr int LC = C / 4, RL = 256 / LC;
    __shared__ double sh[RL * C];
    const int tid = threadIdx.x, cl = tid % LC, rl = tid / LC;
    double a[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (size_t r = (size_t)blockIdx.x * RL + rl; r < rows; r += (size_t)gridDim.x * RL) {
        const float4 v = *reinterpret_cast<const float4*>(d + r * C + cl * 4);
        a[0][0] += v.x; a[0][1] += v.y; a[0][2] += v.z; a[0][3] += v.w;
        a[1][0] += (double)v.x * v.x; a[1][1] += (double)v.y * v.y; a[1][2] += (double)v.z * v.z; a[1][3] += (double)v.w * v.w;
    }
    block_columns<C, 2>(a, sh, partial);
}

// eval mode: normalise with the running statistics
__global__ void k_t_bn_from_running(const float* __restrict__ run_mean, const float* __restrict__ run_var, int C, float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    invstd[c] = (float)(1.0 / sqrt((double)run_var[c] + (double)EPS_BN));
}

// ------------------------------------------------------------------------------------------------
// BN (batch statistics) + ReLU + 2x2 max-pool + channel dropout: z [n][S][S][C] -> a [n][S/2][S/2][C]
// ------------------------------------------------------------------------------------------------
struct PoolWin { float4 y[4]; };

template <int C>
__device__ __forceinline__ void load_window(const float* __restrict__ z, int S, int crop, int py, int px, int c4, float4 v[4]) {
    const float* base = z + (((size_t)crop * S + 2 * py) * S + 2 * px) * C + c4 * 4;
    v[0] = *reinterpret_cast<const float4*>(base);
    v[1] = *reinterpret_cast<const float4*>(base + C);
    v[2] = *reinterpret_cast<const float4*>(base + (size_t)S * C);
    v[3] = *reinterpret_cast<const float4*>(base + (size_t)S * C + C);
}

__device__ __forceinline__ float f4get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ void f4set(float4& v, int k, float x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }

template <int C>
__global__ __launch_bounds__(256) void k_t_bn_pool(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ g, const float* __restrict__ be, const uint8_t* __restrict__ keep /*[n][C]*/,
                                                   float scale, float* __restrict__ a, int n, int S) {
    const int H = S / 2;
    const size_t total = (size_t)n * H * H * (C / 4);
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = idx % (C / 4);
    const size_t pos = idx / (C / 4);
    const int px = pos % H, py = (pos / H) % H, crop = pos / ((size_t)H * H);
    float4 v[4];
    load_window<C>(z, S, crop, py, px, c4, v);
    float4 out;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        const float alpha = invstd[c] * g[c], beta = be[c] - mean[c] * alpha;
        float m = f4get(v[0], k) * alpha + beta;
#pragma unroll
        for (int q = 1; q < 4; ++q) m = fmaxf(m, f4get(v[q], k) * alpha + beta);
        m = fmaxf(m, 0.f);
        f4set(out, k, keep[(size_t)crop * C + c] ? m * scale : 0.f);
    }
    *reinterpret_cast<float4*>(a + pos * C + c4 * 4) = out;
}

// the gradient that reaches the BN output of one pooled element: through dropout, the pool's arg-max (first maximum in scan order,
// like torch's max_pool2d) and the ReLU gate.  Returns the arg-max position and x-hat there.
__device__ __forceinline__ float pooled_grad(const float zq[4], float mean, float inv, float alpha, float beta, float da, bool kept, float scale,
                                             int* arg, float* xhat) {
    float best = zq[0] * alpha + beta;
    int bi = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float y = zq[q] * alpha + beta;
        if (y > best) { best = y; bi = q; }
    }
    *arg = bi;
    *xhat = (zq[bi] - mean) * inv;
    return (kept && best > 0.f) ? da * scale : 0.f;
}

// sums of dy and dy * x-hat over (n, y, x) per channel (BN backward), from the pooled gradient: partial[block][2][C] (FIN_BN_BWD)
template <int C>
__global__ __launch_bounds__(256) void k_t_pool_bwd_stats(const float* __restrict__ da, const float* __restrict__ z, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ g, const float* __restrict__ be,
                                                          const uint8_t* __restrict__ keep, float scale, int n, int S, double* __restrict__ partial) {
    constexpr int LC = C / 4, RL = 256 / LC;
    __shared__ double sh[RL * C];
    const int tid = threadIdx.x, cl = tid % LC, rl = tid / LC;
    const int H = S / 2;
    const size_t rows = (size_t)n * H * H;
    float mn[4], iv[4], al[4], bt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cl * 4 + k;
        mn[k] = mean[c]; iv[k] = invstd[c]; al[k] = iv[k] * g[c]; bt[k] = be[c] - mn[k] * al[k];
    }
    double a[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (size_t r = (size_t)blockIdx.x * RL + rl; r < rows; r += (size_t)gridDim.x * RL) {
        const int px = r % H, py = (r / H) % H, crop = r / ((size_t)H * H);
        float4 v[4];
        load_window<C>(z, S, crop, py, px, cl, v);
        const float4 d = *reinterpret_cast<const float4*>(da + r * C + cl * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float zq[4] = {f4get(v[0], k), f4get(v[1], k), f4get(v[2], k), f4get(v[3], k)};
            int arg; float xh;
            const float gy = pooled_grad(zq, mn[k], iv[k], al[k], bt[k], f4get(d, k), keep[(size_t)crop * C + cl * 4 + k] != 0, scale, &arg, &xh);
            a[0][k] += gy; a[1][k] += (double)gy * xh;
        }
    }
    block_columns<C, 2>(a, sh, partial);
}

// dz = gamma * invstd * (dy - mean(dy) - x-hat * mean(dy x-hat)), written over z (dy is non-zero at the pool's arg-max only); the column sums
// of what was written are the convolution bias's gradient (rounding noise around 0: the bias feeds a BatchNorm): partial[block][1][C]
// (FIN_BIAS).  A thread's elements lie a grid-stride apart (a multiple of C / 4: it stays on its four channels)
template <int C>
__global__ __launch_bounds__(256) void k_t_bn_bwd(const float* __restrict__ da, float* __restrict__ z, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ g, const float* __restrict__ be,
                                                  const uint8_t* __restrict__ keep, float scale, const float* __restrict__ sums, float inv_count, int n, int S,
                                                  double* __restrict__ partial) {
    constexpr int LC = C / 4, RL = 256 / LC;
    __shared__ double sh[RL * C];
    const int H = S / 2;
    const size_t total = (size_t)n * H * H * LC;
    const int c4 = threadIdx.x % LC;
    float mn[4], iv[4], al[4], bt[4], m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        mn[k] = mean[c]; iv[k] = invstd[c]; al[k] = iv[k] * g[c]; bt[k] = be[c] - mn[k] * al[k];
        m1[k] = sums[c] * inv_count; m2[k] = sums[C + c] * inv_count;
    }
    double a[1][4] = {{0, 0, 0, 0}};
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t pos = idx / LC;
        const int px = pos % H, py = (pos / H) % H, crop = pos / ((size_t)H * H);
        float4 v[4];
        load_window<C>(z, S, crop, py, px, c4, v);
        const float4 d = *reinterpret_cast<const float4*>(da + pos * C + c4 * 4);
        float4 o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float zq[4] = {f4get(v[0], k), f4get(v[1], k), f4get(v[2], k), f4get(v[3], k)};
            int arg; float xh;
            const float gy = pooled_grad(zq, mn[k], iv[k], al[k], bt[k], f4get(d, k), keep[(size_t)crop * C + c4 * 4 + k] != 0, scale, &arg, &xh);
            float w = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xhat = (zq[q] - mn[k]) * iv[k];
                const float dz = al[k] * ((q == arg ? gy : 0.f) - m1[k] - xhat * m2[k]);
                f4set(o[q], k, dz);
                w += dz;
            }
            a[0][k] += w;
        }
        float* base = z + (((size_t)crop * S + 2 * py) * S + 2 * px) * C + c4 * 4;
        *reinterpret_cast<float4*>(base) = o[0];
        *reinterpret_cast<float4*>(base + C) = o[1];
        *reinterpret_cast<float4*>(base + (size_t)S * C) = o[2];
        *reinterpret_cast<float4*>(base + (size_t)S * C + C) = o[3];
    }
    block_columns<C, 1>(a, sh, partial);
}

// ------------------------------------------------------------------------------------------------
// fc1: h[n][100] = a3[n][hw][c] . W1c[hw][c][o], as 100 partial planes (one per hw) summed in fixed order by the head kernel.
// Block = one position x 64 samples on fp32 MFMA (32x32x2: one step contracts two channels); wave = 32 outputs (the last wave: 4), two sample
// tiles.  (The VALU form -- 8 x 4 outputs per thread, 12 LDS reads per 32 multiply-adds, one wave per SIMD -- took 30 us.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_t_fc1(const float* __restrict__ a3 /*[n][100][128]*/, const float* __restrict__ w /*[100][128][100]*/,
                                               float* __restrict__ hpart /*[100][n][100]*/, int n) {
    __shared__ float As[64 * 129];                                    // [sample][channel], pitch 129: the 32 rows of a fragment on 32 banks
    __shared__ __attribute__((aligned(16))) float Ws[128 * 100];      // [channel][output]
    const int tid = threadIdx.x, hw = blockIdx.x, n0 = blockIdx.y * 64;
    for (int idx = tid; idx < 64 * 32; idx += 256) {
        const int i = idx >> 5, q = idx & 31;
        const float4 v = (n0 + i < n) ? *reinterpret_cast<const float4*>(a3 + ((size_t)(n0 + i) * 100 + hw) * 128 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = As + i * 129 + q * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    {
        const float4* src = reinterpret_cast<const float4*>(w + (size_t)hw * 12800);
        for (int idx = tid; idx < 3200; idx += 256) reinterpret_cast<float4*>(Ws)[idx] = src[idx];
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int o = wave * 32 + j;
    const bool ov = o < 100;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float* a0 = As + j * 129 + h;
    const float* a1 = As + (32 + j) * 129 + h;
    const float* bp = Ws + h * 100 + (ov ? o : 0);
#pragma unroll 8
    for (int t = 0; t < 64; ++t) {
        const float bv = ov ? bp[2 * t * 100] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[2 * t], bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[2 * t], bv, acc1, 0, 0, 0);
    }
    if (!ov) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;                 // accumulator r of lane (j, h) is row 8 (r / 4) + 4 h + r % 4
        if (n0 + i < n) hpart[((size_t)hw * n + n0 + i) * 100 + o] = acc0[r];
        if (n0 + 32 + i < n) hpart[((size_t)hw * n + n0 + 32 + i) * 100 + o] = acc1[r];
    }
}

// block reduction of one float over 128 threads, fixed order
__device__ __forceinline__ float block_sum128(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    return red[0];
}
__device__ __forceinline__ float block_max128(float v, float* red) {
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    return red[0];
}

// one block per sample: fc1 bias + LayerNorm + ReLU + dropout + fc2 + softmax cross entropy, and the way back down to d(fc1 output)
__global__ __launch_bounds__(128) void k_t_head(const float* __restrict__ hpart, int n, const float* __restrict__ b1, const float* __restrict__ lng,
                                                const float* __restrict__ lnb, const uint8_t* __restrict__ keep /*[n][100]*/, float scale,
                                                const float* __restrict__ w2 /*[C][100]*/, const float* __restrict__ b2, const int32_t* __restrict__ targets,
                                                int classes, float* __restrict__ xhat /*[n][100]*/, float* __restrict__ hd /*[n][100]*/,
                                                float* __restrict__ dl /*[n][C]*/, float* __restrict__ dy /*[n][100]*/, float* __restrict__ dh /*[n][100]*/,
                                                float* __restrict__ loss /*[n]*/, int32_t* __restrict__ correct /*[n]*/, int32_t* __restrict__ bad_target) {
    __shared__ float red[128];
    __shared__ float s_d[100];
    __shared__ float s_dl[1024];
    __shared__ int s_arg[128];
    const int tid = threadIdx.x, s = blockIdx.x;
    const bool act = tid < 100;
    float h = 0.f;
    if (act) {
        h = b1[tid];
        for (int hw = 0; hw < 100; ++hw) h += hpart[((size_t)hw * n + s) * 100 + tid];
    }
    const float mean = block_sum128(act ? h : 0.f, red) * 0.01f;
    const float dv = act ? h - mean : 0.f;
    const float var = block_sum128(dv * dv, red) * 0.01f;
    const float rstd = 1.0f / sqrtf(var + EPS_LN);
    float xh = 0.f, y = 0.f, d = 0.f;
    bool kp = false;
    if (act) {
        xh = dv * rstd;
        y = xh * lng[tid] + lnb[tid];
        kp = keep[(size_t)s * 100 + tid] != 0;
        d = kp ? fmaxf(y, 0.f) * scale : 0.f;
        s_d[tid] = d;
        xhat[(size_t)s * 100 + tid] = xh;
        hd[(size_t)s * 100 + tid] = d;
    }
    __syncthreads();
    // logits
    float lmax = -INFINITY;
    int larg = 0x7fffffff;
    for (int c = tid; c < classes; c += 128) {
        float acc = b2[c];
        const float* wr = w2 + (size_t)c * 100;
        for (int j = 0; j < 100; ++j) acc += s_d[j] * wr[j];
        s_dl[c] = acc;
        if (acc > lmax) { lmax = acc; larg = c; }
    }
    const float gmax = block_max128(lmax, red);
    __syncthreads();
    s_arg[tid] = (lmax == gmax) ? larg : 0x7fffffff;       // first index of the maximum, like torch.argmax
    __syncthreads();
    for (int st = 64; st > 0; st >>= 1) {
        if (tid < st) s_arg[tid] = min(s_arg[tid], s_arg[tid + st]);
        __syncthreads();
    }
    int target = targets[s];
    if (target < 0 || target >= classes) target = 0;       // (memory safety only: k_t_check_targets has already refused this step)
    float se = 0.f;
    for (int c = tid; c < classes; c += 128) se += expf(s_dl[c] - gmax);
    const float sum = block_sum128(se, red);
    const float lse = gmax + logf(sum);
    if (tid == 0) {
        loss[s] = lse - s_dl[target];
        correct[s] = s_arg[0] == target ? 1 : 0;
    }
    __syncthreads();
    const float invn = 1.0f / (float)n;
    for (int c = tid; c < classes; c += 128) {
        const float p = expf(s_dl[c] - lse);
        const float gl = (p - (c == target ? 1.f : 0.f)) * invn;
        s_dl[c] = gl;
        dl[(size_t)s * classes + c] = gl;
    }
    __syncthreads();
    float gy = 0.f;
    if (act) {
        float dd = 0.f;
        for (int c = 0; c < classes; ++c) dd += s_dl[c] * w2[(size_t)c * 100 + tid];
        gy = (kp && y > 0.f) ? dd * scale : 0.f;              // through dropout and the ReLU gate: gradient at the LayerNorm output
        dy[(size_t)s * 100 + tid] = gy;
    }
    // LayerNorm backward: dh = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)), dxh = gy * gamma
    const float dxh = act ? gy * lng[tid] : 0.f;
    const float m1 = block_sum128(dxh, red) * 0.01f;
    const float m2 = block_sum128(dxh * xh, red) * 0.01f;
    if (act) dh[(size_t)s * 100 + tid] = rstd * (dxh - m1 - xh * m2);
}

// parameter gradients of the head: fc2 weight / bias, LayerNorm gamma / beta, fc1 bias; one thread per element, samples in order
__global__ __launch_bounds__(256) void k_t_head_grads(const float* __restrict__ dl, const float* __restrict__ hd, const float* __restrict__ dy,
                                                      const float* __restrict__ xhat, const float* __restrict__ dh, int n, int classes,
                                                      float* __restrict__ g_w2, float* __restrict__ g_b2, float* __restrict__ g_lng, float* __restrict__ g_lnb,
                                                      float* __restrict__ g_b1) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int nw = classes * 100;
    // sums over the samples in order, eight samples' operands in flight (one dependent L2 round trip per sample otherwise)
#define HG_SUM(expr_a_, expr_b_)                                                                                                 \
    for (int s0 = 0; s0 < n; s0 += 8) {                                                                                          \
        float va_[8], vb_[8];                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) { const int s = s0 + k < n ? s0 + k : n - 1; va_[k] = (expr_a_); vb_[k] = (expr_b_); } \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) if (s0 + k < n) acc += va_[k] * vb_[k];                                     \
    }
    if (idx < nw) {
        const int c = idx / 100, j = idx % 100;
        float acc = 0.f;
        HG_SUM(dl[(size_t)s * classes + c], hd[(size_t)s * 100 + j])
        g_w2[idx] = acc;
    } else if (idx < nw + classes) {
        const int c = idx - nw;
        float acc = 0.f;
        HG_SUM(dl[(size_t)s * classes + c], 1.f)
        g_b2[c] = acc;
    } else if (idx < nw + classes + 300) {
        const int k3 = idx - nw - classes, j = k3 % 100, which = k3 / 100;
        float acc = 0.f;
        if (which == 0) { HG_SUM(dy[(size_t)s * 100 + j], xhat[(size_t)s * 100 + j]) g_lng[j] = acc; }
        else if (which == 1) { HG_SUM(dy[(size_t)s * 100 + j], 1.f) g_lnb[j] = acc; }
        else { HG_SUM(dh[(size_t)s * 100 + j], 1.f) g_b1[j] = acc; }
    }
#undef HG_SUM
}

// fc1 weight gradient: dW1c[hw][c][o] = sum_n a3[n][hw][c] * dh[n][o] -- per spatial position a 128 x 100 matrix = A^T D with the batch as
// the contraction: fp32 MFMA, one 32x32x2 step contracts two samples; block = one position, wave = 32 channels x all outputs (4 tiles,
// the last one 4 columns wide); samples in order, so the same bits every run
__global__ __launch_bounds__(256) void k_t_fc1_wgrad(const float* __restrict__ a3, const float* __restrict__ dh, float* __restrict__ gw, int n) {
    const int tid = threadIdx.x, lane = tid & 63, mt = tid >> 6, j = lane & 31, h = lane >> 5, hw = blockIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    const float* ap = a3 + (size_t)hw * 128 + mt * 32 + j;
    const bool last_ok = 96 + j < 100;
    for (int s0 = 0; s0 < n; s0 += 2) {
        const int s = s0 + h;
        const bool ok = s < n;
        const float av = ok ? ap[(size_t)s * 12800] : 0.f;
        float bv[4];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) bv[nt] = ok ? dh[(size_t)s * 100 + nt * 32 + j] : 0.f;
        bv[3] = (ok && last_ok) ? dh[(size_t)s * 100 + 96 + j] : 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nt], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int o = nt * 32 + j;
        if (o >= 100) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = mt * 32 + 8 * (r / 4) + 4 * h + (r % 4);         // accumulator r of lane (j, h) is row 8 (r / 4) + 4 h + r % 4
            gw[((size_t)hw * 128 + c) * 100 + o] = acc[nt][r];
        }
    }
}

// fc1 data gradient: da3[n][hw][c] = sum_o dh[n][o] * W1c[hw][c][o].  Block = one position x 32 samples on fp32 MFMA (one step contracts two
// outputs), wave = 32 channels
__global__ __launch_bounds__(256) void k_t_fc1_dgrad(const float* __restrict__ dh, const float* __restrict__ w, float* __restrict__ da3, int n) {
    __shared__ float Ws[128 * 101];                                   // [channel][output], pitch 101: a fragment's 32 channels on 32 banks
    __shared__ float Ds[32 * 101];                                    // [sample][output]
    const int tid = threadIdx.x, hw = blockIdx.x, n0 = blockIdx.y * 32;
    for (int idx = tid; idx < 128 * 100; idx += 256) Ws[(idx / 100) * 101 + idx % 100] = w[(size_t)hw * 12800 + idx];
    for (int idx = tid; idx < 32 * 100; idx += 256) Ds[(idx / 100) * 101 + idx % 100] = (n0 + idx / 100) < n ? dh[(size_t)n0 * 100 + idx] : 0.f;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int c = wave * 32 + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = Ds + j * 101 + h;
    const float* bp = Ws + c * 101 + h;
#pragma unroll 10
    for (int t = 0; t < 50; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * t], bp[2 * t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nn = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (nn < n) da3[((size_t)nn * 100 + hw) * 128 + c] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------
// conv2 / conv3 forward and data gradients on the 16-bit matrix cores (precision 0, the default): the inference path's arithmetic --
// x = h1 + h2 with two fp16 pieces (22 mantissa bits), products h2 g1, h1 g2, h1 g1 on v_mfma_f32_32x32x16_f16, fp32 accumulation --
// at a fifth of the fp32-MFMA time (k_conv5<RAW>, precision 1).  fp16 cannot hold every fp32 value, so both operands carry a power-of-two
// scale: the inputs one per workgroup and 16 / 32-channel chunk, from the largest |value| of the patch it stages (largest scaled input in
// [2^13, 2^14)); the weights one per layer, from the pack kernel (largest scaled weight in [2^7, 2^8)).  Nothing overflows, and what
// underflows is below 2^-27 of the patch's / the layer's largest entry.  25 shifted GEMMs: patch in LDS, weight fragments L2 -> VGPR, no barrier inside a chunk.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pow2_scale_for(const uint32_t maxbits, const int target_exp) {   // 2^k with max * 2^k in [2^target, 2^(target + 1))
    if (maxbits == 0u) return 1.f;
    int e = (int)((maxbits >> 23) & 0xffu) - 127;
    int k = target_exp - e;
    k = k < -100 ? -100 : (k > 100 ? 100 : k);
    return __uint_as_float((uint32_t)(127 + k) << 23);
}
__device__ __forceinline__ void split2h_t(const float x, uint32_t& h1, uint32_t& h2) {
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)(x - (float)a);
    h1 = (uint32_t)__builtin_bit_cast(uint16_t, a); h2 = (uint32_t)__builtin_bit_cast(uint16_t, b);
}

template <int CI, int CO, int S, int ROWS, int CIC>
struct ConvGeomH {
    static constexpr int PW = S + 4, PH = ROWS + 4;
    static constexpr int PSTRIDE = CIC * 2 + 16;                      // bytes per pixel (data + 16 pad: odd multiple of 16)
    static constexpr int PATCH = PH * PW * PSTRIDE;                   // bytes per piece
    static constexpr int BT = 2 * (CIC / 8) * CO * 16;                // bytes per weight tile: 2 pieces x CIC / 8 k-octets x CO x 16 B
    static constexpr int NPIX = ROWS * S;
    static constexpr int MT = (NPIX + 31) / 32, NT = CO / 32, WM = 8 / NT, TPW = (MT + WM - 1) / WM;
    static constexpr int LDS_BYTES = 2 * PATCH;
    static constexpr int BPC = S / ROWS;
    static_assert(CO % 32 == 0 && 8 % NT == 0 && CIC % 16 == 0 && CI % CIC == 0 && LDS_BYTES <= 160 * 1024, "shapes");
};

template <int CI, int CO, int S, int ROWS, int CIC, int COUT>
__global__ __launch_bounds__(512) void k_t_conv5_h2(const float* __restrict__ in /*[N][S][S][CI]*/, const uint4* __restrict__ wp /*[CI/CIC][25][2][CIC/8][CO] x 16 B*/,
                                                    const float* __restrict__ bias /*[COUT] or null*/, float* __restrict__ out /*[N][S][S][COUT]*/,
                                                    const float* __restrict__ w_scale, double* __restrict__ stat_partial /*[block][2][CO] or null*/) {
    using G = ConvGeomH<CI, CO, S, ROWS, CIC>;
    constexpr int Q4 = CIC / 4, KO = CIC / 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t hlds[];
    uint8_t* patch = hlds;                       // 2 pieces
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int n = wave % G::NT, mg = wave / G::NT;
    const int crop = blockIdx.x / G::BPC, row0 = (blockIdx.x % G::BPC) * ROWS;
    __shared__ float s_red[8];
    float in_scale = 1.f;                                             // of the chunk staged last
    int aoff[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) {
        int p = (mg + G::WM * m) * 32 + j;                             // row-major pixel of the band
        p = p < G::NPIX ? p : G::NPIX - 1;
        aoff[m] = ((p / S) * G::PW + p % S) * G::PSTRIDE + h * 16;
    }
    f32x16 acc[G::TPW];
#pragma unroll
    for (int m = 0; m < G::TPW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* inc = in + (size_t)crop * S * S * CI;
    constexpr int NITEM = G::PH * G::PW * Q4, NLD = (NITEM + 511) / 512;  // Q4 float4 per pixel; float4 per thread and chunk
    constexpr int KS = CIC / 16, BV = G::BT / 16;
    const uint4* wl = wp + (h * CO + n * 32 + j);                      // this lane's fragments: + tap * BV + 2 ks CO (+ KO CO: second piece)
    for (int cc = 0; cc < CI / CIC; ++cc) {
        const uint4* wsrc = wl + (size_t)cc * 25 * BV;
        uint4 bq[3][KS][2];
        auto fetch = [&](uint4 (&d)[KS][2], const int tap) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { d[ks][0] = wsrc[tap * BV + 2 * ks * CO]; d[ks][1] = wsrc[tap * BV + 2 * ks * CO + KO * CO]; }
        };
        fetch(bq[0], 0);
        fetch(bq[1], 1);
        __syncthreads();
        // the chunk's patch: loaded into registers, its largest |value| found (wave shuffle + 8 LDS words), scaled by the power of two that
        // puts that value into [2^13, 2^14), split and stored.  A chunk with another scale than the one before rescales the accumulators
        // (a power of two: exact)
        float4 v[NLD];
        float lm = 0.f;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * 512;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < NITEM) {
                const int q = idx % Q4, px = idx / Q4;
                const int py = px / G::PW, pxx = px - py * G::PW;
                const int iy = row0 + py - 2, ix = pxx - 2;
                if (iy >= 0 && iy < S && ix >= 0 && ix < S) v[u] = *reinterpret_cast<const float4*>(inc + ((size_t)iy * S + ix) * CI + cc * CIC + q * 4);
                lm = fmaxf(lm, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) lm = fmaxf(lm, __shfl_xor(lm, d));
        if (lane == 0) s_red[wave] = lm;
        __syncthreads();
        float bm = s_red[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) bm = fmaxf(bm, s_red[w]);
        const float sc = pow2_scale_for(__float_as_uint(bm), 13);
        if (cc > 0 && sc != in_scale) {
            const float f = sc / in_scale;
#pragma unroll
            for (int m = 0; m < G::TPW; ++m) acc[m] *= f;
        }
        in_scale = sc;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + u * 512;
            if (idx < NITEM) {
                const int q = idx % Q4, px = idx / Q4;
                uint32_t a1[4], a2[4];
                split2h_t(v[u].x * sc, a1[0], a2[0]); split2h_t(v[u].y * sc, a1[1], a2[1]);
                split2h_t(v[u].z * sc, a1[2], a2[2]); split2h_t(v[u].w * sc, a1[3], a2[3]);
                uint8_t* d = patch + px * G::PSTRIDE + q * 8;
                *reinterpret_cast<uint2*>(d) = make_uint2(a1[0] | (a1[1] << 16), a1[2] | (a1[3] << 16));
                *reinterpret_cast<uint2*>(d + G::PATCH) = make_uint2(a2[0] | (a2[1] << 16), a2[2] | (a2[3] << 16));
            }
        }
        __syncthreads();
        // the tap loop runs without barriers: the patch is read-only until the next chunk, and each wave takes its weight fragments L2 -> VGPR
        // straight from the packed image, two taps ahead (through LDS they cost a barrier per tap: 1.4 us per tap against 0.7 of MFMAs)
#pragma unroll
        for (int tap = 0; tap < 25; ++tap) {
            if (tap + 2 < 25) fetch(bq[(tap + 2) % 3], tap + 2);
            const uint8_t* asrc = patch + ((tap / 5) * G::PW + (tap % 5)) * G::PSTRIDE;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                              // one 16-deep MFMA step per 16 input channels
                const h16x8 b1 = __builtin_bit_cast(h16x8, bq[tap % 3][ks][0]);
                const h16x8 b2 = __builtin_bit_cast(h16x8, bq[tap % 3][ks][1]);
#pragma unroll
                for (int m = 0; m < G::TPW; ++m) {
                    const h16x8 p1 = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(asrc + aoff[m] + ks * 32));
                    const h16x8 p2 = __builtin_bit_cast(h16x8, *reinterpret_cast<const uint4*>(asrc + aoff[m] + ks * 32 + G::PATCH));
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, b1, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b2, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, b1, acc[m], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                                      // (the epilogue's statistics reuse the patch)
    const int co = n * 32 + j;
    if (co >= COUT && !(COUT == CO && stat_partial)) return;
    const float out_scale = 1.0f / (in_scale * *w_scale);             // powers of two: exact
    const float bz = bias ? bias[co] : 0.f;
    float* oc = out + ((size_t)crop * S * S + (size_t)row0 * S) * COUT;
    double sv = 0, sq = 0;                                            // this lane's share of the channel's sum and sum of squares (the BN that follows)
#pragma unroll
    for (int m = 0; m < G::TPW; ++m) {
        const int mt = mg + G::WM * m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = mt * 32 + 8 * (r / 4) + 4 * h + (r % 4);        // accumulator r of lane (j, h) is row 8 (r / 4) + 4 h + r % 4
            if (p < G::NPIX) {
                const float v = acc[m][r] * out_scale + bz;
                oc[(size_t)p * COUT + co] = v;
                if (COUT == CO) { sv += v; sq += (double)v * v; }
            }
        }
    }
    if constexpr (COUT == CO) {
        if (!stat_partial) return;                                    // (uniform)
        // the 2 WM lanes of a channel through LDS (the patch is dead: the tap loop ended on a barrier), added in a fixed order
        double* sh = reinterpret_cast<double*>(hlds);
        const int slot = mg * 2 + h;
        sh[slot * CO + co] = sv;
        sh[(2 * G::WM + slot) * CO + co] = sq;
        __syncthreads();
        if (tid < 2 * CO) {
            const int pl = tid / CO, c = tid % CO;
            double t = 0;
#pragma unroll
            for (int k = 0; k < 2 * G::WM; ++k) t += sh[(pl * 2 * G::WM + k) * CO + c];
            stat_partial[((size_t)blockIdx.x * 2 + pl) * CO + c] = t;
        }
    }
}

// weights of one layer -> the B-operand image of k_t_conv5_h2, both directions in one launch: [dir][k / CICK][tap][piece][(k % CICK) / 8][n] x 16 B
// (8 consecutive k per 16 bytes).  dir 0 = forward: k = ci, n = co, w(ci, tap, co); dir 1 = data gradient: k = co, n = ci (padded to NP1
// columns), w(ci, 24 - tap, co).  Parameter layout of w: [ci / CICP][tap][ci % CICP][co].  Every block finds the tensor's largest |w| itself
// (the same value in every block), block 0 writes the scale.
__global__ __launch_bounds__(1024) void k_t_pack_h2(const float* __restrict__ w, const int CI, const int CO, const int CICP, const int CICK0, const int CICK1,
                                                    const int NP1, uint4* __restrict__ out0, uint4* __restrict__ out1, float* __restrict__ w_scale) {
    __shared__ float red[1024];
    const int total4 = 25 * CI * CO / 4;
    float m = 0.f;
#pragma unroll 4
    for (int i = threadIdx.x; i < total4; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(w)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]); __syncthreads(); }
    const float sc = pow2_scale_for(__float_as_uint(red[0]), 7);
    if (blockIdx.x == 0 && threadIdx.x == 0) *w_scale = sc;
    auto wat = [&](int ci, int tap, int co) -> float { return w[(((size_t)(ci / CICP) * 25 + tap) * CICP + ci % CICP) * CO + co]; };
    const int n0 = (CI / 8) * 25 * CO, n1 = (CO / 8) * 25 * NP1;         // 16-byte units per piece
    for (int u = blockIdx.x * 1024 + threadIdx.x; u < n0 + n1; u += gridDim.x * 1024) {
        const bool d1 = u >= n0;
        const int v = d1 ? u - n0 : u;
        const int K = d1 ? CO : CI, N = d1 ? NP1 : CO, CK = d1 ? CICK1 : CICK0;
        const int nn = v % N, ko = (v / N) % (CK / 8), tap = (v / (N * (CK / 8))) % 25, cc = v / (N * (CK / 8) * 25);
        (void)K;
        uint32_t h1[8], h2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = cc * CK + ko * 8 + e;
            float x = 0.f;
            if (!d1) x = wat(k, tap, nn);
            else if (nn < CI) x = wat(nn, 24 - tap, k);
            split2h_t(x * sc, h1[e], h2[e]);
        }
        uint4* o = (d1 ? out1 : out0) + ((size_t)(cc * 25 + tap) * 2 * (CK / 8) + ko) * N + nn;
        o[0] = make_uint4(h1[0] | (h1[1] << 16), h1[2] | (h1[3] << 16), h1[4] | (h1[5] << 16), h1[6] | (h1[7] << 16));
        o[(size_t)(CK / 8) * N] = make_uint4(h2[0] | (h2[1] << 16), h2[2] | (h2[3] << 16), h2[4] | (h2[5] << 16), h2[6] | (h2[7] << 16));
    }
}

// ------------------------------------------------------------------------------------------------
// convolution weight gradient on fp32 MFMA: dW[tap][ci][co] = sum over (crop, y, x) a[crop][y+ky-2][x+kx-2][ci] * dz[crop][y][x][co].
// One 32x32x2 MFMA step contracts over two neighbouring pixels: A = activations at the taps' shifts, B = dz.
//   * block = one kernel row ky x one slice of CIS input channels x one slice of COS output channels x one share of the work units
//     (unit = ROWS output rows of one crop).  It stages the unit's dz slice [ROWS * S pixels][COS] and the ROWS activation rows the kernel
//     row needs ([ROWS][S][CIS], rows outside the crop skipped) in LDS with 16-byte loads -- round 2 fetched every MFMA operand as a
//     4-byte load straight from L2 and ran at half its MFMA count -- and all 8 waves read their operands from there;
//   * M space of a kernel row = [5 kx][CIS] rows in MT tiles of 32 (conv3: CIS = 32, tile = kx; conv2: CIS = 16, 80 rows in 3 tiles);
//     a wave holds MT x (COS / 32) accumulator tiles and takes every 8th pixel pair; one dz fragment serves the MT shifted A fragments;
//   * the waves' sums are added in wave order through LDS: part[share][tap][ci][co], summed over the shares by k_t_wgrad_reduce --
//     the same bits every run.
// ------------------------------------------------------------------------------------------------
template <int CI, int CO, int S, int CIS, int COS, int MT, int ROWS>
struct WgradGeom {
    static constexpr int NT = COS / 32, NB = S / ROWS, WAVES = 8;
    static constexpr int NCS = CI / CIS, NOS = CO / COS, TYPES = 5 * NCS * NOS;
    static constexpr int DZ_FLOATS = ROWS * S * COS, A_FLOATS = ROWS * S * CIS, UNIT_FLOATS = DZ_FLOATS + A_FLOATS;
    static constexpr int SUM_FLOATS = MT * NT * 16 * 64;
    static constexpr int LDS_BYTES = (2 * UNIT_FLOATS > SUM_FLOATS ? 2 * UNIT_FLOATS : SUM_FLOATS) * 4;
    static_assert(S % ROWS == 0 && S % 2 == 0 && CI % CIS == 0 && CO % COS == 0 && COS % 32 == 0 && MT * 32 >= 5 * CIS, "shapes");
    static_assert(LDS_BYTES <= 160 * 1024 && (DZ_FLOATS * 4) % 1024 == 0 && 1024 % (COS * 4) == 0 && 1024 % (CIS * 4) == 0, "LDS");
};

// one LDS-DMA instruction: 64 lanes x 16 bytes from the lanes' global addresses to LDS [lds_addr, lds_addr + 1024).  Raw, so that the compiler's
// wait-count pass does not know of it: it cannot tell the two LDS buffers apart and would drain the DMA in front of every LDS read.
__device__ __forceinline__ void dma16_raw(const void* gptr, const uint32_t lds_addr) {
    uint32_t keep;                                                         // M0 is the compiler's: handed back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(gptr) : "memory");
}

template <int CI, int CO, int S, int CIS, int COS, int MT, int ROWS>
__global__ __launch_bounds__(512) void k_t_wgrad(const float* __restrict__ a /*[n][S][S][CI]*/, const float* __restrict__ dz /*[n][S][S][CO]*/,
                                                 float* __restrict__ part, int n) {
    using G = WgradGeom<CI, CO, S, CIS, COS, MT, ROWS>;
    constexpr int NT = G::NT, NB = G::NB, WAVES = G::WAVES;
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 31, h = lane >> 5;
    const int type = blockIdx.x, ky = type / (G::NCS * G::NOS), cs = (type / G::NOS) % G::NCS, os = type % G::NOS;
    const int share = blockIdx.y, shares = gridDim.y, units = n * NB;
    int kx[MT], cc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int mi = mt * 32 + j;
        kx[mt] = mi / CIS; cc[mt] = mi % CIS;                              // rows past the 5th tap stay zero
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    // a unit travels L2 -> the other LDS buffer by LDS-DMA while the MFMAs of the previous unit run: 1-KB pieces (4 dz pixels, or
    // 1024 / (4 CIS) activation pixels), dealt out to the 8 waves.  Activation rows outside the crop are fetched from a clamped row
    // and never read.
    const uint32_t lds0 = (uint32_t)(uintptr_t)wg_lds;
#define WG_FETCH(u_, buf_)                                                                                                       \
    do {                                                                                                                         \
        const int crop_ = (u_) / NB, y0_ = ((u_) % NB) * ROWS;                                                                   \
        const uint8_t* src_ = reinterpret_cast<const uint8_t*>(dz + ((size_t)crop_ * S * S + (size_t)y0_ * S) * CO + os * COS);  \
        const uint32_t dst_ = lds0 + (buf_) * G::UNIT_FLOATS * 4;                                                                \
        _Pragma("unroll 1") for (int k_ = wave; k_ < G::DZ_FLOATS * 4 / 1024; k_ += WAVES)                                        \
            dma16_raw(src_ + (size_t)(k_ * (1024 / (COS * 4)) + lane / (COS / 4)) * (CO * 4) + (lane % (COS / 4)) * 16, dst_ + k_ * 1024); \
        const uint8_t* asrc_ = reinterpret_cast<const uint8_t*>(a + (size_t)crop_ * S * S * CI + cs * CIS);                       \
        constexpr int PPI_ = 1024 / (CIS * 4);                           /* pixels per piece */                                  \
        _Pragma("unroll 1") for (int k_ = wave; k_ < (G::A_FLOATS * 4 + 1023) / 1024; k_ += WAVES) {                              \
            const int px_ = k_ * PPI_ + lane / (CIS / 4);                                                                        \
            if (px_ < ROWS * S) {                                                                                                \
                int ys_ = y0_ + px_ / S + ky - 2;                                                                                \
                ys_ = ys_ < 0 ? 0 : (ys_ >= S ? S - 1 : ys_);                                                                    \
                dma16_raw(asrc_ + ((size_t)ys_ * S + px_ % S) * (CI * 4) + (lane % (CIS / 4)) * 16, dst_ + G::DZ_FLOATS * 4 + k_ * 1024); \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    int u = share;
    if (u < units) WG_FETCH(u, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (; u < units; u += shares) {
        const int un = u + shares;
        if (un < units) WG_FETCH(un, cur ^ 1);                             // that buffer's readers finished before the previous barrier
        const float* dzl = wg_lds + cur * G::UNIT_FLOATS;
        const float* al = dzl + G::DZ_FLOATS;
        // rows of the unit whose activation row y + ky - 2 lies inside the crop (the others add nothing)
        const int y0 = (u % NB) * ROWS;
        int r_lo = 2 - ky - y0, r_hi = S + 2 - ky - y0;
        r_lo = r_lo < 0 ? 0 : r_lo; r_hi = r_hi > ROWS ? ROWS : r_hi;
        const int nq = (r_hi - r_lo) * (S / 2);
        // operands of pixel pair q: the dz fragments and the MT shifted activation fragments (both pixels of a pair lie in one row: S is even)
#define WG_OPS(q_, av_, bv_)                                                                                                     \
        do {                                                                                                                     \
            const int r_ = r_lo + (q_) / (S / 2), x_ = 2 * ((q_) % (S / 2)) + h;                                                 \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bv_[nt] = dzl[(r_ * S + x_) * COS + nt * 32 + j];                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                                  \
                const int xs_ = x_ + kx[mt] - 2;                                                                                 \
                const float v_ = al[(r_ * S + (xs_ < 0 ? 0 : (xs_ >= S ? S - 1 : xs_))) * CIS + cc[mt]];                          \
                av_[mt] = (kx[mt] < 5 && xs_ >= 0 && xs_ < S) ? v_ : 0.f;                                                        \
            }                                                                                                                    \
        } while (0)
        float av[MT], bv[NT];
        if (wave < nq) WG_OPS(wave, av, bv);
        for (int q = wave; q < nq; q += WAVES) {
            float avn[MT], bvn[NT];
            const int qn = q + WAVES < nq ? q + WAVES : q;                 // the next pair's operands are read under this pair's MFMAs
            WG_OPS(qn, avn, bvn);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = avn[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bvn[nt];
        }
#undef WG_OPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // this wave's pieces of the next unit have landed ...
        __syncthreads();                                                   // ... everyone's, and everyone is done with this unit
        cur ^= 1;
    }
#undef WG_FETCH
    // the waves' sums are added in wave order through LDS: one partial per block, the same bits every run
    __syncthreads();
    float* sum = wg_lds;
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* d = sum + ((mt * NT + nt) * 16 + r) * 64 + lane;
                        *d = w == 0 ? acc[mt][nt][r] : *d + acc[mt][nt][r];
                    }
        }
        __syncthreads();
    }
    float* pp = part + (size_t)share * 25 * CI * CO;
    for (int e = tid; e < G::SUM_FLOATS; e += WAVES * 64) {
        const int ln = e & 63, r = (e >> 6) & 15, t = e >> 10, nt = t % NT, mt = t / NT;
        const int mi = mt * 32 + 8 * (r / 4) + 4 * (ln >> 5) + (r % 4);     // accumulator r of lane (j, h) is row 8 (r / 4) + 4 h + r % 4
        const int k = mi / CIS, c = cs * CIS + mi % CIS;
        if (k < 5) pp[((size_t)(ky * 5 + k) * CI + c) * CO + os * COS + nt * 32 + (ln & 31)] = sum[e];
    }
}

// ------------------------------------------------------------------------------------------------
// the same weight gradients in the fp16 two-piece split arithmetic (precision 0): v_mfma_f32_32x32x16_f16 contracts SIXTEEN pixels per step
// (three products per term), a fifth of the fp32-MFMA time -- the kernel is then bound by staging, so a workgroup stages a unit once for
// ALL 25 taps:
//   * block = a slice of CIS input channels x a slice of 64 output channels x a share of the units (unit = ROWS output rows of a crop);
//     10 waves = 5 kernel rows x 2 output-channel tiles, a wave holds the 5 kx tiles (conv2: 80 rows = 5 kx x 16 channels in 3 tiles) of
//     its kernel row and walks over ALL pixel segments of the unit: no sum across waves;
//   * operands: a [ROWS + 4 rows][x + 2 .. halo][CIS] and dz [ROWS][x (padded to 8)][64] as two fp16 planes each, [pixel][channel] as they lie
//     in memory; the MFMA wants 8 consecutive PIXELS per lane for one channel -- ds_read_b64_tr_b16 delivers exactly that transpose (a
//     16-lane group reads 4 pixels x 16 channels; lane l supplies pixel l / 4, channels 4 (l % 4) .. and receives channel l of the 4 pixels);
//     a K step = two 8-pixel segments (lanes 0..31 / 32..63), shifted by kx - 2 along x for the A operand (zero halo columns), rows
//     outside the crop are zero rows;
//   * fp16 range: a unit's activations and gradients are scaled by powers of two from their largest |value| (kept while that stays inside
//     [2^10, 2^15)); a change rescales the accumulators (exact).
// ------------------------------------------------------------------------------------------------
typedef __fp16 fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h16x8 tr_frag(const uint8_t* p, const int second_off) {
    struct { fh4 lo, hi; } v;
    v.lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(p));
    v.hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)(p + second_off));
    return __builtin_bit_cast(h16x8, v);
}

template <int CI, int CO, int S, int CIS, int ROWS>
struct WgradHGeom {
    static constexpr int COS = 64, NT = 2, MT = (5 * CIS + 31) / 32, WAVES = 10, NTHR = WAVES * 64;
    static constexpr int SEGS = (S + 7) / 8, DW = SEGS * 8, PW = DW + 4, AR = ROWS + 4;
    static constexpr int A_BYTES = AR * PW * CIS * 2, DZ_BYTES = ROWS * DW * COS * 2;          // per piece
    static constexpr int LDS_BYTES = 2 * A_BYTES + 2 * DZ_BYTES;
    static constexpr int NSEG = ROWS * SEGS;
    static constexpr int NCS = CI / CIS, NOS = CO / COS, TYPES = NCS * NOS, NB = S / ROWS;
    static constexpr int NA4 = AR * S * CIS / 4, ND4 = ROWS * S * COS / 4, NLA = (NA4 + NTHR - 1) / NTHR, NLD = (ND4 + NTHR - 1) / NTHR;
    static_assert(NSEG % 2 == 0 && S % ROWS == 0 && CI % CIS == 0 && CO % COS == 0 && CIS % 16 == 0 && LDS_BYTES % 16 == 0 && LDS_BYTES <= 160 * 1024, "shapes");
};

template <int CI, int CO, int S, int CIS, int ROWS>
__global__ __launch_bounds__(640) void k_t_wgrad_h2(const float* __restrict__ a /*[n][S][S][CI]*/, const float* __restrict__ dz /*[n][S][S][CO]*/,
                                                    float* __restrict__ part, int n) {
    using G = WgradHGeom<CI, CO, S, CIS, ROWS>;
    constexpr int COS = G::COS, MT = G::MT, PW = G::PW, DW = G::DW, SEGS = G::SEGS;
    extern __shared__ __attribute__((aligned(16))) uint8_t wh_lds[];
    __shared__ float s_red[G::WAVES][2];
    uint8_t* const ap = wh_lds;                                          // two pieces, A_BYTES apart
    uint8_t* const dp = wh_lds + 2 * G::A_BYTES;                         // two pieces, DZ_BYTES apart
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5, g16 = (lane >> 4) & 1, l16 = lane & 15;
    const int ky = wave % 5, ntw = wave / 5;
    const int cs = blockIdx.x / G::NOS, os = blockIdx.x % G::NOS;
    const int share = blockIdx.y, shares = gridDim.y, units = n * G::NB;
    for (int i = tid; i < G::LDS_BYTES / 16; i += G::NTHR) reinterpret_cast<uint4*>(wh_lds)[i] = make_uint4(0, 0, 0, 0);   // halo and pad columns stay zero
    // lane constants of the transposing reads: tile i = rows i * 32 + 16 g16 + (l16): kernel column kx = row / CIS, channel = row % CIS
    int a_lane[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m0 = i * 32 + 16 * g16;
        int kx = m0 / CIS;
        kx = kx < 5 ? kx : 4;                                            // rows past the fifth tap are never written out
        a_lane[i] = (((l16 >> 2) + kx) * CIS + m0 % CIS + 4 * (l16 & 3)) * 2;
    }
    const int d_lane = ((l16 >> 2) * COS + ntw * 32 + 16 * g16 + 4 * (l16 & 3)) * 2;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float sa = 1.f, sd = 1.f;                                            // scales of the unit staged last
    for (int u = share; u < units; u += shares) {
        const int crop = u / G::NB, y0 = (u % G::NB) * ROWS;
        const float* asrc = a + (size_t)crop * S * S * CI + cs * CIS;
        const float* dsrc = dz + ((size_t)crop * S * S + (size_t)y0 * S) * CO + os * COS;
        float4 va[G::NLA], vd[G::NLD];
        float ma = 0.f, md = 0.f;
#pragma unroll
        for (int k = 0; k < G::NLA; ++k) {
            const int idx = tid + k * G::NTHR;
            va[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < G::NA4) {
                const int c4 = idx % (CIS / 4), x = (idx / (CIS / 4)) % S, ar = idx / (CIS / 4 * S);
                const int y = y0 - 2 + ar;
                if (y >= 0 && y < S) va[k] = *reinterpret_cast<const float4*>(asrc + ((size_t)y * S + x) * CI + c4 * 4);
                ma = fmaxf(ma, fmaxf(fmaxf(fabsf(va[k].x), fabsf(va[k].y)), fmaxf(fabsf(va[k].z), fabsf(va[k].w))));
            }
        }
#pragma unroll
        for (int k = 0; k < G::NLD; ++k) {
            const int idx = tid + k * G::NTHR;
            vd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < G::ND4) {
                const int c4 = idx % (COS / 4), px = idx / (COS / 4);
                vd[k] = *reinterpret_cast<const float4*>(dsrc + (size_t)px * CO + c4 * 4);
                md = fmaxf(md, fmaxf(fmaxf(fabsf(vd[k].x), fabsf(vd[k].y)), fmaxf(fabsf(vd[k].z), fabsf(vd[k].w))));
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, d)); md = fmaxf(md, __shfl_xor(md, d)); }
        __syncthreads();                                                 // the previous unit's readers are done (and s_red is free)
        if (lane == 0) { s_red[wave][0] = ma; s_red[wave][1] = md; }
        __syncthreads();
        ma = s_red[0][0]; md = s_red[0][1];
#pragma unroll
        for (int w = 1; w < G::WAVES; ++w) { ma = fmaxf(ma, s_red[w][0]); md = fmaxf(md, s_red[w][1]); }
        // keep a scale while the unit's largest value stays inside [2^10, 2^15) with it: few rescalings of the accumulators
        float na = sa, nd = sd;
        if (!(ma * sa >= 1024.f && ma * sa < 32768.f) && ma > 0.f) na = pow2_scale_for(__float_as_uint(ma), 13);
        if (!(md * sd >= 1024.f && md * sd < 32768.f) && md > 0.f) nd = pow2_scale_for(__float_as_uint(md), 13);
        if (na != sa || nd != sd) {
            const float f = (na / sa) * (nd / sd);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] *= f;
            sa = na; sd = nd;
        }
#pragma unroll
        for (int k = 0; k < G::NLA; ++k) {
            const int idx = tid + k * G::NTHR;
            if (idx < G::NA4) {
                const int c4 = idx % (CIS / 4), x = (idx / (CIS / 4)) % S, ar = idx / (CIS / 4 * S);
                uint32_t p1[4], p2[4];
                split2h_t(va[k].x * sa, p1[0], p2[0]); split2h_t(va[k].y * sa, p1[1], p2[1]);
                split2h_t(va[k].z * sa, p1[2], p2[2]); split2h_t(va[k].w * sa, p1[3], p2[3]);
                uint8_t* d = ap + ((ar * PW + x + 2) * CIS + c4 * 4) * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16));
                *reinterpret_cast<uint2*>(d + G::A_BYTES) = make_uint2(p2[0] | (p2[1] << 16), p2[2] | (p2[3] << 16));
            }
        }
#pragma unroll
        for (int k = 0; k < G::NLD; ++k) {
            const int idx = tid + k * G::NTHR;
            if (idx < G::ND4) {
                const int c4 = idx % (COS / 4), px = idx / (COS / 4), x = px % S, r = px / S;
                uint32_t p1[4], p2[4];
                split2h_t(vd[k].x * sd, p1[0], p2[0]); split2h_t(vd[k].y * sd, p1[1], p2[1]);
                split2h_t(vd[k].z * sd, p1[2], p2[2]); split2h_t(vd[k].w * sd, p1[3], p2[3]);
                uint8_t* d = dp + ((r * DW + x) * COS + c4 * 4) * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16));
                *reinterpret_cast<uint2*>(d + G::DZ_BYTES) = make_uint2(p2[0] | (p2[1] << 16), p2[2] | (p2[3] << 16));
            }
        }
        __syncthreads();
        for (int p = 0; p < G::NSEG / 2; ++p) {
            const int sg = 2 * p + h, r = sg / SEGS, x0 = 8 * (sg - r * SEGS);
            const uint8_t* dseg = dp + (r * DW + x0) * COS * 2 + d_lane;
            const uint8_t* aseg = ap + ((r + ky) * PW + x0) * CIS * 2;
            const h16x8 b1 = tr_frag(dseg, 4 * COS * 2), b2 = tr_frag(dseg + G::DZ_BYTES, 4 * COS * 2);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const h16x8 a1 = tr_frag(aseg + a_lane[i], 4 * CIS * 2), a2 = tr_frag(aseg + a_lane[i] + G::A_BYTES, 4 * CIS * 2);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[i], 0, 0, 0);
            }
        }
    }
    const float inv = 1.0f / (sa * sd);
    const int co = os * COS + ntw * 32 + j;
    float* pp = part + (size_t)share * 25 * CI * CO;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = i * 32 + 8 * (r / 4) + 4 * h + (r % 4);            // accumulator r of lane (j, h) is row 8 (r / 4) + 4 h + r % 4
            const int kx = m / CIS, c = cs * CIS + m % CIS;
            if (kx < 5) pp[((size_t)(ky * 5 + kx) * CI + c) * CO + co] = acc[i][r] * inv;
        }
}

// sum of the partials in order -> gradient in the parameter layout [ci / CIC][tap][ci % CIC][co]
__global__ __launch_bounds__(256) void k_t_wgrad_reduce(const float* __restrict__ part, int nparts, int CI, int CO, int CIC, float* __restrict__ g) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = 25 * CI * CO;
    if (idx >= total) return;
    const int co = idx % CO, cic = (idx / CO) % CIC, tap = (idx / (CO * CIC)) % 25, cc = idx / (CO * CIC * 25);
    const size_t src = ((size_t)tap * CI + cc * CIC + cic) * CO + co;
    float acc = 0.f;
    for (int p0 = 0; p0 < nparts; p0 += 8) {                            // eight loads in flight, added in part order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p0 + k < nparts ? part[(size_t)(p0 + k) * total + src] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (p0 + k < nparts) acc += v[k];
    }
    g[idx] = acc;
}

// conv1 weight gradient (VALU): block = 8 rows of one crop -> part[block][c][tap][16].  Thread = (2 of the 8 rows, kernel row ky, output channel):
// per pixel one dz value and one new window value from LDS feed the five kx of the kernel row (a thread per tap read two LDS words per multiply
// and the kernel ran at the LDS rate); the four row pairs are added in order through LDS
template <int CH>
__global__ __launch_bounds__(512) void k_t_wgrad1(const float* __restrict__ x /*[n][80][80][CH]*/, const float* __restrict__ dz /*[n][80][80][16]*/,
                                                  float* __restrict__ part) {
    __shared__ float xs[12 * 84 * CH];
    __shared__ float ds[640 * 16];
    const int tid = threadIdx.x, crop = blockIdx.x / 10, row0 = (blockIdx.x % 10) * 8;
    for (int idx = tid; idx < 12 * 84 * CH; idx += 512) {
        const int c = idx % CH, px = (idx / CH) % 84, py = idx / (CH * 84);
        const int iy = row0 + py - 2, ix = px - 2;
        xs[idx] = (iy >= 0 && iy < 80 && ix >= 0 && ix < 80) ? x[(((size_t)crop * 80 + iy) * 80 + ix) * CH + c] : 0.f;
    }
    {
        const float4* src = reinterpret_cast<const float4*>(dz + ((size_t)crop * 80 + row0) * 80 * 16);
        for (int idx = tid; idx < 640 * 4; idx += 512) reinterpret_cast<float4*>(ds)[idx] = src[idx];
    }
    __syncthreads();
    const int co = tid & 15, ky = (tid >> 4) % 5, g = tid / 80;
    float acc[CH][5];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[c][k] = 0.f;
    if (tid < 320) {
#pragma unroll 1
        for (int yy = 0; yy < 2; ++yy) {
            const int y = g * 2 + yy;
            const float* xr = xs + (y + ky) * 84 * CH;
            const float* dr = ds + y * 80 * 16 + co;
            float win[CH][5];
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int k = 1; k < 5; ++k) win[c][k] = xr[(k - 1) * CH + c];
#pragma unroll 5
            for (int xx = 0; xx < 80; ++xx) {
                const float d = dr[xx * 16];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) win[c][k] = win[c][k + 1];
                    win[c][4] = xr[(xx + 4) * CH + c];
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[c][k] += win[c][k] * d;
                }
            }
        }
    }
    __syncthreads();                                                   // ds becomes the row pairs' exchange: [g][c][tap][co]
    if (tid < 320)
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int k = 0; k < 5; ++k) ds[((g * CH + c) * 25 + ky * 5 + k) * 16 + co] = acc[c][k];
    __syncthreads();
    if (tid >= 400) return;
#pragma unroll
    for (int c = 0; c < CH; ++c)
        part[(size_t)blockIdx.x * CH * 400 + c * 400 + tid] = ((ds[c * 400 + tid] + ds[(CH + c) * 400 + tid]) + ds[(2 * CH + c) * 400 + tid]) + ds[(3 * CH + c) * 400 + tid];
}

// out[i] = sum over the parts, fixed order: 16 lanes per element take every 16th part, then the lane sums are added in order
__global__ __launch_bounds__(256) void k_t_reduce(const float* __restrict__ part, int nparts, int count, float* __restrict__ out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x, e = tid & 15, ln = tid >> 4, idx = blockIdx.x * 16 + e;
    float acc = 0.f;
    if (idx < count) {
        for (int p0 = ln; p0 < nparts; p0 += 128) {                     // eight loads in flight, added in part order
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = p0 + 16 * k < nparts ? part[(size_t)(p0 + 16 * k) * count + idx] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (p0 + 16 * k < nparts) acc += v[k];
        }
    }
    sh[ln * 16 + e] = acc;
    __syncthreads();
    if (tid < 16 && idx < count) {
        float r = 0.f;
        for (int k = 0; k < 16; ++k) r += sh[k * 16 + tid];
        out[idx] = r;
    }
}

// weights of the data-gradient convolutions: input channels = the forward layer's outputs, taps flipped.
// wb[cc'][t][cic'][co'] = w_fwd(ci = co', co = cc' * CICB + cic', tap = 24 - t); forward layout [ci / CIC][tap][ci % CIC][co]
__global__ __launch_bounds__(256) void k_t_repack_bwd(const float* __restrict__ wf, int CI, int CO, int CIC, int COP /*padded output channels*/,
                                                      int CICB /*input-channel chunk of the data-gradient kernel*/, float* __restrict__ wb) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = 25 * CO * COP;
    if (idx >= total) return;
    const int cop = idx % COP, cicp = (idx / COP) % CICB, t = (idx / (COP * CICB)) % 25, ccp = idx / (COP * CICB * 25);
    float v = 0.f;
    if (cop < CI) {
        const int ci = cop, co = ccp * CICB + cicp, tap = 24 - t;
        v = wf[(((size_t)(ci / CIC) * 25 + tap) * CIC + ci % CIC) * CO + co];
    }
    wb[idx] = v;
}

struct AdamSkip { uint32_t lo[6], hi[6]; };

// train() asserts 0 <= target < classes before it touches the model (visual_recognition_torch.py:1109-1112).  Targets handed over in device
// memory cannot be checked by the host without a synchronisation, so the step checks them first, on the device: refused[0] counts the steps
// that were refused -- this one if a target is out of range, and every following one until the host has reported the error (sticky) -- and
// the kernels that change the trainer's state (running statistics, Adam) return when it is set.  refused[1]: the same for an eval batch.
__global__ __launch_bounds__(256) void k_t_check_targets(const int32_t* __restrict__ targets, int n, int classes, int32_t* __restrict__ refused, int eval) {
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    bool bad = false;
    for (int i = threadIdx.x; i < n; i += 256) bad |= targets[i] < 0 || targets[i] >= classes;
    if (bad) s_bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (eval) { if (s_bad) refused[1] = 1; }
        else if (s_bad || refused[0]) refused[0] += 1;
    }
}

// torch.optim.Adam (_single_tensor_adam, no weight decay / amsgrad): m.lerp_(g, 1 - b1); v = v * b2 + (1 - b2) g g;
// p += -step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps)
__global__ __launch_bounds__(256) void k_t_adam(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M, float* __restrict__ V,
                                                uint32_t total, AdamSkip skip, float w1, float b2, float omb2, float step_size, float bc2_sqrt, float eps,
                                                const int32_t* __restrict__ refused) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total || *refused) return;                    // a refused step updates nothing
#pragma unroll
    for (int k = 0; k < 6; ++k)
        if (i >= skip.lo[k] && i < skip.hi[k]) return;          // running statistics are buffers, not parameters
    const float g = G[i];
    float m = M[i], v = V[i];
    m = m + w1 * (g - m);
    v = v * b2 + (omb2 * g) * g;
    M[i] = m; V[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    P[i] = P[i] + (-step_size) * (m / denom);
}

__global__ __launch_bounds__(64) void k_t_loss(const float* __restrict__ loss, const int32_t* __restrict__ correct, int n, float* __restrict__ out /*[2]*/) {
    double s = 0;                                            // one wave: lane sums in index order, then a fixed butterfly
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 64) { s += loss[i]; c += correct[i]; }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { s += __shfl_xor(s, d); c += __shfl_xor(c, d); }
    if (threadIdx.x == 0) { out[0] = (float)(s / n); out[1] = (float)c; }
}

// keep masks when the caller injects none: one counter-based hash per (step, layer, sample, channel)
__global__ __launch_bounds__(256) void k_t_masks(uint8_t* __restrict__ keep, size_t count, uint64_t seed, uint64_t step, float p_drop) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    uint64_t zed = seed + 0x9E3779B97F4A7C15ull * (step * 0x100000001B3ull + i + 1);
    zed = (zed ^ (zed >> 30)) * 0xBF58476D1CE4E5B9ull;
    zed = (zed ^ (zed >> 27)) * 0x94D049BB133111EBull;
    zed ^= zed >> 31;
    const float u = (float)(zed >> 40) * (1.0f / 16777216.0f);
    keep[i] = u >= p_drop ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Trainer {
    trexhip_ctx* ctx = nullptr;
    int classes = 0, CH = 1, max_n = 0;
    trexhip_train_params p{};
    int64_t step = 0;
    size_t off[T_COUNT + 1] = {}, cnt[T_COUNT] = {};
    size_t total = 0;
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
    float *z1 = nullptr, *a1 = nullptr, *z2 = nullptr, *a2 = nullptr, *z3 = nullptr, *a3 = nullptr, *da3 = nullptr, *da2 = nullptr, *da1 = nullptr;
    float *wb3 = nullptr, *wb2 = nullptr, *part = nullptr, *stat = nullptr /*3 x (mean, invstd, sums[2]) x 128*/;
    // precision 0: fp16 two-piece operand images of the conv2 / conv3 weights (forward, data gradient) and their scales
    uint4 *wh2f = nullptr, *wh2b = nullptr, *wh3f = nullptr, *wh3b = nullptr;
    float* wh_scale = nullptr;
    float *hpart = nullptr, *xhat = nullptr, *hd = nullptr, *dl = nullptr, *dy = nullptr, *dh = nullptr, *loss = nullptr, *out2 = nullptr;
    int32_t* correct = nullptr;
    int32_t* bad_target = nullptr;       // set by k_t_head when a target is outside 0..classes-1; sticky until read
    double* red = nullptr;
    uint8_t* keep = nullptr;
    float* x_stage = nullptr;            // host-pointer entry point: inputs, targets and injected masks staged here
    int32_t* y_stage = nullptr;
    uint8_t* keep_stage = nullptr;
    size_t part_floats = 0;
    float* part1 = nullptr;              // conv1's weight-gradient partials (its kernels run beside the side stream's, which own `part`)
    // the training step forks: weight packing, the head's / fc1's / the convolutions' weight gradients and the loss sum run on `side`, beside the
    // chain that the next kernel waits for (data gradients, BN backward); the step joins again in front of Adam
    hipStream_t side = nullptr;
    hipEvent_t ev[8] = {};
    bool masks_on_side = false;          // this step's keep masks are being drawn on `side` (ev[6])
    double* red_bias = nullptr;          // [3][BWD_BLOCKS][128]: k_t_bn_bwd's column sums, finalized on `side`
    bool attr = false;
    std::vector<void*> allocs;
};

static constexpr int RED_BLOCKS = 256, BWD_BLOCKS = 1024;   // grids of the reduction kernels (partials: red[BWD_BLOCKS][2][128])
using WG3 = WgradGeom<64, 128, 20, 32, 64, 5, 10>;   // conv3: block type = kernel row x 32-channel half x 64-channel half (20 types), unit = 10 rows of a crop
using WG2 = WgradGeom<16, 64, 40, 16, 64, 3, 5>;     // conv2: block type = kernel row (5 types), unit = 5 rows of a crop
using WH3 = WgradHGeom<64, 128, 20, 32, 10>;        // precision 0: conv3 4 block types x 64 shares, conv2 1 x 256
using WH2 = WgradHGeom<16, 64, 40, 16, 10>;
static constexpr int SHARES3H = 64, SHARES2H = 256;
static constexpr int SHARES3 = 12, SHARES2 = 51;     // 20 x 12 = 240 and 5 x 51 = 255 workgroups: one round on 256 CUs

static size_t tensor_count(int t, int classes, int CH) {
    switch (t) {
        case T_C1W: return (size_t)16 * CH * 25;
        case T_C1B: case T_G1: case T_BE1: case T_RM1: case T_RV1: return 16;
        case T_C2W: return (size_t)64 * 16 * 25;
        case T_C2B: case T_G2: case T_BE2: case T_RM2: case T_RV2: return 64;
        case T_C3W: return (size_t)128 * 64 * 25;
        case T_C3B: case T_G3: case T_BE3: case T_RM3: case T_RV3: return 128;
        case T_F1W: return (size_t)100 * 12800;
        case T_F1B: case T_LNG: case T_LNB: return 100;
        case T_F2W: return (size_t)classes * 100;
        case T_F2B: return classes;
    }
    return 0;
}

// index inside the tensor in the kernels' layout of element `i` of the torch layout
static size_t to_internal(int t, size_t i, int CH) {
    switch (t) {
        case T_C1W: { const size_t tap = i % 25, c = (i / 25) % CH, co = i / (25 * (size_t)CH); return (c * 25 + tap) * 16 + co; }
        case T_C2W: { const size_t tap = i % 25, ci = (i / 25) % 16, co = i / 400; return (tap * 16 + ci) * 64 + co; }
        case T_C3W: { const size_t tap = i % 25, ci = (i / 25) % 64, co = i / 1600; return (((ci / 32) * 25 + tap) * 32 + ci % 32) * 128 + co; }
        case T_F1W: { const size_t k = i % 12800, o = i / 12800, c = k / 100, hw = k % 100; return (hw * 128 + c) * 100 + o; }
        default: return i;
    }
}

template <class T>
static int dev_alloc(Trainer* t, T** p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) { set_error("trexhip_trainer_create: out of device memory"); return TREXHIP_E_NOMEM; }
    t->allocs.push_back(q);
    *p = static_cast<T*>(q);
    return TREXHIP_OK;
}

static void trainer_free(Trainer* t) {
    if (!t) return;
    if (t->side) (void)hipStreamSynchronize(t->side);                     // (a step joins its side stream before Adam; this is for a half-queued one)
    for (void* q : t->allocs) (void)hipFree(q);
    for (hipEvent_t e : t->ev) if (e) (void)hipEventDestroy(e);
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    delete t;
}

using G2F = ConvGeom<16, 64, 40, 20, 16>;
using G3F = ConvGeom<64, 128, 20, 10, 32>;     // 10-row bands: 2 blocks per crop (a training batch is 64..128 crops, the chip has 256 CUs; 4-row bands measured no faster)
using G3B = ConvGeom<128, 64, 20, 10, 32>;
using G2B = ConvGeom16<64, 40, 20, 16>;
using H2F = ConvGeomH<16, 64, 40, 20, 16>;       // precision 0 (fp16 two-piece split): conv2 / conv3 forward, conv3 / conv2 data gradient
using H3F = ConvGeomH<64, 128, 20, 10, 32>;
using H3B = ConvGeomH<128, 64, 20, 10, 32>;
using H2B = ConvGeomH<64, 32, 40, 20, 16>;       // 16 output channels in a 32-wide tile          // conv2's data gradient: 16 output channels on 16x16x4 tiles, two blocks per crop

template <int CH>
static void launch_layer1(Trainer* t, hipStream_t s, const float* x, int n, bool train) {
    hipLaunchKernelGGL((k_t_conv1<CH>), dim3(n * 20), dim3(320), 0, s, x, t->P + t->off[T_C1W], t->P + t->off[T_C1B], t->z1, train ? t->red : nullptr);
}
template <int CH>
static void launch_wgrad1(Trainer* t, hipStream_t s, const float* x, int n) {
    hipLaunchKernelGGL((k_t_wgrad1<CH>), dim3(n * 10), dim3(512), 0, s, x, t->z1, t->part1);
    const int count = CH * 400;
    hipLaunchKernelGGL(k_t_reduce, dim3((count + 15) / 16), dim3(256), 0, s, t->part1, n * 10, count, t->G + t->off[T_C1W]);
}

template <int C>
static void bn_forward(Trainer* t, hipStream_t s, int layer, const float* z, float* a, int n, int S, int tg, int tb, int trm, int trv, const uint8_t* keep,
                       float scale, int have_partials /*workgroups of the convolution that left their column sums in t->red; 0 = none did*/) {
    float* mean = t->stat + layer * 512;
    float* invstd = mean + 128;
    const size_t rows = (size_t)n * S * S;
    if (!have_partials) hipLaunchKernelGGL((k_t_bn_stats<C>), dim3(RED_BLOCKS), dim3(256), 0, s, z, rows, t->red);
    hipLaunchKernelGGL((k_t_red_finalize<FIN_BN_STATS>), dim3(C), dim3(256), 0, s, t->red, have_partials ? have_partials : RED_BLOCKS, C,
                       FinArgs{(double)rows, t->p.bn_momentum, mean, invstd, t->P + t->off[trm], t->P + t->off[trv], t->bad_target});
    const size_t total = (size_t)n * (S / 2) * (S / 2) * (C / 4);
    hipLaunchKernelGGL((k_t_bn_pool<C>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, mean, invstd, t->P + t->off[tg], t->P + t->off[tb], keep, scale,
                       a, n, S);
}

// da (pooled gradient) -> dz written over z; gradients of gamma, beta and of the convolution bias
template <int C>
static int bn_backward(Trainer* t, hipStream_t s, int layer, const float* da, float* z, int n, int S, int tg, int tb, int tcb, const uint8_t* keep, float scale) {
    float* mean = t->stat + layer * 512;
    float* invstd = mean + 128;
    float* sums = mean + 256;
    hipLaunchKernelGGL((k_t_pool_bwd_stats<C>), dim3(RED_BLOCKS), dim3(256), 0, s, da, z, mean, invstd, t->P + t->off[tg], t->P + t->off[tb], keep, scale, n, S,
                       t->red);
    hipLaunchKernelGGL((k_t_red_finalize<FIN_BN_BWD>), dim3(C), dim3(256), 0, s, t->red, RED_BLOCKS, C,
                       FinArgs{0.0, 0.f, sums, t->G + t->off[tg], t->G + t->off[tb], nullptr, nullptr});
    const size_t total = (size_t)n * (S / 2) * (S / 2) * (C / 4);
    const float inv_count = (float)(1.0 / ((double)n * S * S));
    // whole rounds: every workgroup takes the same number of elements, all of them resident at once
    const unsigned nb0 = (unsigned)((total + 255) / 256), per = (nb0 + BWD_BLOCKS - 1) / BWD_BLOCKS, nb = (nb0 + per - 1) / per;
    hipLaunchKernelGGL((k_t_bn_bwd<C>), dim3(nb), dim3(256), 0, s, da, z, mean, invstd, t->P + t->off[tg], t->P + t->off[tb], keep, scale, sums, inv_count, n, S,
                       t->red_bias + (size_t)layer * BWD_BLOCKS * 128);
    return (int)nb;
}
// the convolution bias's gradient from k_t_bn_bwd's column sums (nothing but Adam waits for it: the caller puts it on the side stream)
static void bias_finalize(Trainer* t, hipStream_t w, int layer, int nb, int C, int tcb) {
    hipLaunchKernelGGL((k_t_red_finalize<FIN_BIAS>), dim3(C), dim3(256), 0, w, t->red_bias + (size_t)layer * BWD_BLOCKS * 128, nb, C,
                       FinArgs{0.0, 0.f, t->G + t->off[tcb], nullptr, nullptr, nullptr, nullptr});
}

// steps with a target outside 0..classes-1 were refused on the device (k_t_check_targets) and have changed nothing: report them at the next
// synchronising call, take them back out of the step count (Adam's bias correction, the dropout counter) and go on
static int check_targets(Trainer* t) {
    int32_t bad[2] = {0, 0};
    TH_CHECK_HIP(hipMemcpy(bad, t->bad_target, 8, hipMemcpyDeviceToHost));
    if (bad[0] || bad[1]) {
        TH_CHECK_HIP(hipMemset(t->bad_target, 0, 8));
        if (bad[0]) t->step -= bad[0];
        set_error(bad[0] ? std::string("training step: a target class index was outside 0..classes-1; that step (and the ") + std::to_string(bad[0] - 1) +
                               " queued behind it) was refused, parameters, moments and running statistics are unchanged"
                         : std::string("evaluation batch: a target class index was outside 0..classes-1"));
        return TREXHIP_E_INVALID;
    }
    return TREXHIP_OK;
}

// forward pass up to the per-sample loss / arg-max (and, as a by-product of k_t_head, the gradient at the fc1 output).  train = batch
// statistics + dropout masks `keep`; eval = running statistics, nothing dropped (model.eval(), visual_recognition_torch.py:1171-1185)
static void trainer_forward(Trainer* t, hipStream_t s, const float* x, const int32_t* targets, int n, const uint8_t* keep, float scale, bool train,
                            hipStream_t side = nullptr) {
    const uint8_t *k1 = keep, *k2 = keep + (size_t)n * 16, *k3 = keep + (size_t)n * 80, *k4 = keep + (size_t)n * 208;
    float* P = t->P;
    const size_t* o = t->off;
    const bool h2 = t->p.precision == 0;
    if (h2) {
        // this step's weights as fp16 two-piece operand images (both directions); conv2 is the first to need them
        hipStream_t ps = side ? side : s;
        hipLaunchKernelGGL(k_t_pack_h2, dim3(10), dim3(1024), 0, ps, P + o[T_C2W], 16, 64, 16, 16, 16, 32, t->wh2f, t->wh2b, t->wh_scale);
        hipLaunchKernelGGL(k_t_pack_h2, dim3(50), dim3(1024), 0, ps, P + o[T_C3W], 64, 128, 32, 32, 32, 64, t->wh3f, t->wh3b, t->wh_scale + 1);
        if (side) (void)hipEventRecord(t->ev[1], side);
    }
    auto block = [&](auto tag, int layer, const float* z, float* a, int S, int tg, int tb, int trm, int trv, const uint8_t* kp, int have_partials) {
        constexpr int C = decltype(tag)::value;
        if (train) { bn_forward<C>(t, s, layer, z, a, n, S, tg, tb, trm, trv, kp, scale, have_partials); return; }
        float* mean = t->stat + layer * 512;
        float* invstd = mean + 128;
        hipLaunchKernelGGL(k_t_bn_from_running, dim3(1), dim3(128), 0, s, P + o[trm], P + o[trv], C, mean, invstd);
        const size_t total = (size_t)n * (S / 2) * (S / 2) * (C / 4);
        hipLaunchKernelGGL((k_t_bn_pool<C>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, mean, invstd, P + o[tg], P + o[tb], kp, scale, a, n, S);
    };
    if (t->CH == 1) launch_layer1<1>(t, s, x, n, train); else launch_layer1<3>(t, s, x, n, train);
    if (side && t->masks_on_side) (void)hipStreamWaitEvent(s == hipStreamLegacy ? nullptr : s, t->ev[6], 0);
    block(std::integral_constant<int, 16>{}, 0, t->z1, t->a1, 80, T_G1, T_BE1, T_RM1, T_RV1, k1, n * 20);
    if (h2 && side) (void)hipStreamWaitEvent(s == hipStreamLegacy ? nullptr : s, t->ev[1], 0);
    if (h2) hipLaunchKernelGGL((k_t_conv5_h2<16, 64, 40, 20, 16, 64>), dim3(n * H2F::BPC), dim3(512), H2F::LDS_BYTES, s, t->a1, t->wh2f, P + o[T_C2B], t->z2, t->wh_scale, train ? t->red : nullptr);
    else hipLaunchKernelGGL((k_conv5<16, 64, 40, 20, 16, CONV_EPI_RAW, 64>), dim3(n * G2F::BPC), dim3(512), G2F::LDS_BYTES, s, t->a1, P + o[T_C2W], P + o[T_C2B], t->z2);
    block(std::integral_constant<int, 64>{}, 1, t->z2, t->a2, 40, T_G2, T_BE2, T_RM2, T_RV2, k2, h2 ? n * H2F::BPC : 0);
    if (h2) hipLaunchKernelGGL((k_t_conv5_h2<64, 128, 20, 10, 32, 128>), dim3(n * H3F::BPC), dim3(512), H3F::LDS_BYTES, s, t->a2, t->wh3f, P + o[T_C3B], t->z3, t->wh_scale + 1, train ? t->red : nullptr);
    else hipLaunchKernelGGL((k_conv5<64, 128, 20, 10, 32, CONV_EPI_RAW, 128>), dim3(n * G3F::BPC), dim3(512), G3F::LDS_BYTES, s, t->a2, P + o[T_C3W], P + o[T_C3B], t->z3);
    block(std::integral_constant<int, 128>{}, 2, t->z3, t->a3, 20, T_G3, T_BE3, T_RM3, T_RV3, k3, h2 ? n * H3F::BPC : 0);
    hipLaunchKernelGGL(k_t_fc1, dim3(100, (n + 63) / 64), dim3(256), 0, s, t->a3, P + o[T_F1W], t->hpart, n);
    hipLaunchKernelGGL(k_t_head, dim3(n), dim3(128), 0, s, t->hpart, n, P + o[T_F1B], P + o[T_LNG], P + o[T_LNB], k4, scale, P + o[T_F2W], P + o[T_F2B], targets,
                       t->classes, t->xhat, t->hd, t->dl, t->dy, t->dh, t->loss, t->correct, t->bad_target);
}

static int trainer_attrs(Trainer* t) {
    if (t->attr) return TREXHIP_OK;
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5<16, 64, 40, 20, 16, CONV_EPI_RAW, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, G2F::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5<64, 128, 20, 10, 32, CONV_EPI_RAW, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, G3F::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5<128, 64, 20, 10, 32, CONV_EPI_RAW, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, G3B::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv5_n16<64, 40, 20, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, G2B::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_conv5_h2<16, 64, 40, 20, 16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, H2F::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_conv5_h2<64, 128, 20, 10, 32, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, H3F::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_conv5_h2<128, 64, 20, 10, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, H3B::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_conv5_h2<64, 32, 40, 20, 16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, H2B::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_wgrad<64, 128, 20, 32, 64, 5, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, WG3::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_wgrad<16, 64, 40, 16, 64, 3, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, WG2::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_wgrad_h2<64, 128, 20, 32, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, WH3::LDS_BYTES));
    TH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_t_wgrad_h2<16, 64, 40, 16, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, WH2::LDS_BYTES));
    t->attr = true;
    return TREXHIP_OK;
}

// model.eval() forward + mean cross entropy + arg-max count of one validation batch (train(), :1171-1190); changes nothing in the trainer
static int trainer_eval(Trainer* t, const float* x, const int32_t* targets, int n, float* h_loss, int32_t* h_correct) {
    trexhip_ctx* ctx = t->ctx;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipStream_t s = ctx->stream;
    { const int rc = trainer_attrs(t); if (rc) return rc; }
    TH_CHECK_HIP(hipMemsetAsync(t->keep, 1, (size_t)n * 308, s));                 // nothing dropped
    hipLaunchKernelGGL(k_t_check_targets, dim3(1), dim3(256), 0, s, targets, n, t->classes, t->bad_target, 1);
    trainer_forward(t, s, x, targets, n, t->keep, 1.0f, false);
    hipLaunchKernelGGL(k_t_loss, dim3(1), dim3(64), 0, s, t->loss, t->correct, n, t->out2);
    TH_CHECK_HIP(hipGetLastError());
    float two[2];
    TH_CHECK_HIP(hipMemcpyAsync(two, t->out2, sizeof(two), hipMemcpyDeviceToHost, s));
    TH_CHECK_HIP(hipStreamSynchronize(s));
    if (h_loss) *h_loss = two[0];
    if (h_correct) *h_correct = (int32_t)two[1];
    return check_targets(t);
}

static int trainer_step(Trainer* t, const float* x, const int32_t* targets, int n, const uint8_t* d_keep, float* h_loss, int32_t* h_correct) {
    trexhip_ctx* ctx = t->ctx;
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    hipStream_t s = ctx->stream;
    { const int rc = trainer_attrs(t); if (rc) return rc; }
    const float scale = 1.0f / (1.0f - t->p.dropout);
    const uint8_t* keep = d_keep;
    float* P = t->P;
    float* G = t->G;
    const size_t* o = t->off;
    // the weight-gradient side of the step.  Events go through the null stream when the caller's stream is the hipStreamLegacy handle (this
    // runtime faults in a wait on an event recorded on the handle itself; in this library the two name the same stream); the per-thread
    // handle has no such stand-in: the step then stays on the one stream
    const bool forked = s != hipStreamPerThread;
    hipStream_t w = forked ? t->side : s;
    hipStream_t es = s == hipStreamLegacy ? nullptr : s;
    auto fork = [&](int e) { if (forked) { (void)hipEventRecord(t->ev[e], es); (void)hipStreamWaitEvent(w, t->ev[e], 0); } };
    fork(0);                                                                // behind whatever wrote the parameters last
    t->masks_on_side = forked && !keep;
    if (!keep) {
        const size_t count = (size_t)n * 308;
        hipLaunchKernelGGL(k_t_masks, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, w, t->keep, count, t->p.seed, (uint64_t)t->step, t->p.dropout);
        if (forked) (void)hipEventRecord(t->ev[6], w);
        keep = t->keep;
    }
    hipLaunchKernelGGL(k_t_check_targets, dim3(1), dim3(256), 0, s, targets, n, t->classes, t->bad_target, 0);
    trainer_forward(t, s, x, targets, n, keep, scale, true, forked ? w : nullptr);
    fork(2);
    const uint8_t *k1 = keep, *k2 = keep + (size_t)n * 16, *k3 = keep + (size_t)n * 80;
    // ---- backward
    {
        const int cnt = t->classes * 100 + t->classes + 300;
        hipLaunchKernelGGL(k_t_head_grads, dim3((cnt + 255) / 256), dim3(256), 0, w, t->dl, t->hd, t->dy, t->xhat, t->dh, n, t->classes, G + o[T_F2W], G + o[T_F2B],
                           G + o[T_LNG], G + o[T_LNB], G + o[T_F1B]);
    }
    hipLaunchKernelGGL(k_t_fc1_wgrad, dim3(100), dim3(256), 0, w, t->a3, t->dh, G + o[T_F1W], n);
    hipLaunchKernelGGL(k_t_loss, dim3(1), dim3(64), 0, w, t->loss, t->correct, n, t->out2);
    hipLaunchKernelGGL(k_t_fc1_dgrad, dim3(100, (n + 31) / 32), dim3(256), 0, s, t->dh, P + o[T_F1W], t->da3, n);
    // block 3
    const bool h2 = t->p.precision == 0;
    const int nb3 = bn_backward<128>(t, s, 2, t->da3, t->z3, n, 20, T_G3, T_BE3, T_C3B, k3, scale);
    fork(3);
    bias_finalize(t, w, 2, nb3, 128, T_C3B);
    {
        const int shares = h2 ? (n * WH3::NB < SHARES3H ? n * WH3::NB : SHARES3H) : (n * WG3::NB < SHARES3 ? n * WG3::NB : SHARES3);
        if (h2) hipLaunchKernelGGL((k_t_wgrad_h2<64, 128, 20, 32, 10>), dim3(WH3::TYPES, shares), dim3(640), WH3::LDS_BYTES, w, t->a2, t->z3, t->part, n);
        else hipLaunchKernelGGL((k_t_wgrad<64, 128, 20, 32, 64, 5, 10>), dim3(WG3::TYPES, shares), dim3(512), WG3::LDS_BYTES, w, t->a2, t->z3, t->part, n);
        hipLaunchKernelGGL(k_t_wgrad_reduce, dim3((25 * 64 * 128 + 255) / 256), dim3(256), 0, w, t->part, shares, 64, 128, 32, G + o[T_C3W]);
        if (h2) hipLaunchKernelGGL((k_t_conv5_h2<128, 64, 20, 10, 32, 64>), dim3(n * H3B::BPC), dim3(512), H3B::LDS_BYTES, s, t->z3, t->wh3b, (const float*)nullptr, t->da2, t->wh_scale + 1, (double*)nullptr);
        else {
            hipLaunchKernelGGL(k_t_repack_bwd, dim3((25 * 128 * 64 + 255) / 256), dim3(256), 0, s, P + o[T_C3W], 64, 128, 32, 64, 32, t->wb3);
            hipLaunchKernelGGL((k_conv5<128, 64, 20, 10, 32, CONV_EPI_RAW, 64>), dim3(n * G3B::BPC), dim3(512), G3B::LDS_BYTES, s, t->z3, t->wb3, (const float*)nullptr, t->da2);
        }
    }
    // block 2
    const int nb2 = bn_backward<64>(t, s, 1, t->da2, t->z2, n, 40, T_G2, T_BE2, T_C2B, k2, scale);
    fork(4);
    bias_finalize(t, w, 1, nb2, 64, T_C2B);
    {
        const int shares = h2 ? (n * WH2::NB < SHARES2H ? n * WH2::NB : SHARES2H) : (n * WG2::NB < SHARES2 ? n * WG2::NB : SHARES2);
        if (h2) hipLaunchKernelGGL((k_t_wgrad_h2<16, 64, 40, 16, 10>), dim3(WH2::TYPES, shares), dim3(640), WH2::LDS_BYTES, w, t->a1, t->z2, t->part, n);
        else hipLaunchKernelGGL((k_t_wgrad<16, 64, 40, 16, 64, 3, 5>), dim3(WG2::TYPES, shares), dim3(512), WG2::LDS_BYTES, w, t->a1, t->z2, t->part, n);
        hipLaunchKernelGGL(k_t_wgrad_reduce, dim3((25 * 16 * 64 + 255) / 256), dim3(256), 0, w, t->part, shares, 16, 64, 16, G + o[T_C2W]);
        if (h2) hipLaunchKernelGGL((k_t_conv5_h2<64, 32, 40, 20, 16, 16>), dim3(n * H2B::BPC), dim3(512), H2B::LDS_BYTES, s, t->z2, t->wh2b, (const float*)nullptr, t->da1, t->wh_scale, (double*)nullptr);
        else {
            hipLaunchKernelGGL(k_t_repack_bwd, dim3((25 * 64 * 16 + 255) / 256), dim3(256), 0, s, P + o[T_C2W], 16, 64, 16, 16, 16, t->wb2);
            hipLaunchKernelGGL((k_conv5_n16<64, 40, 20, 16>), dim3(n * G2B::BPC), dim3(512), G2B::LDS_BYTES, s, t->z2, t->wb2, t->da1);
        }
    }
    // block 1
    const int nb1 = bn_backward<16>(t, s, 0, t->da1, t->z1, n, 80, T_G1, T_BE1, T_C1B, k1, scale);
    fork(7);
    bias_finalize(t, w, 0, nb1, 16, T_C1B);
    if (forked) (void)hipEventRecord(t->ev[5], w);                          // the side stream's last piece of this step
    if (t->CH == 1) launch_wgrad1<1>(t, s, x, n); else launch_wgrad1<3>(t, s, x, n);
    if (forked) (void)hipStreamWaitEvent(es, t->ev[5], 0);                  // join: every gradient is in G
    // ---- optimizer
    t->step += 1;
    {
        const double b1 = t->p.beta1, b2 = t->p.beta2;
        const double bc1 = 1.0 - std::pow(b1, (double)t->step), bc2 = 1.0 - std::pow(b2, (double)t->step);
        AdamSkip skip;
        const int bufs[6] = {T_RM1, T_RV1, T_RM2, T_RV2, T_RM3, T_RV3};
        for (int k = 0; k < 6; ++k) { skip.lo[k] = (uint32_t)o[bufs[k]]; skip.hi[k] = (uint32_t)(o[bufs[k]] + t->cnt[bufs[k]]); }
        hipLaunchKernelGGL(k_t_adam, dim3((unsigned)((t->total + 255) / 256)), dim3(256), 0, s, P, G, t->M, t->V, (uint32_t)t->total, skip, (float)(1.0 - b1), (float)b2,
                           (float)(1.0 - b2), (float)((double)t->p.lr / bc1), (float)std::sqrt(bc2), t->p.eps, t->bad_target);
    }
    TH_CHECK_HIP(hipGetLastError());
    if (h_loss || h_correct) {
        float two[2];
        TH_CHECK_HIP(hipMemcpyAsync(two, t->out2, sizeof(two), hipMemcpyDeviceToHost, s));
        TH_CHECK_HIP(hipStreamSynchronize(s));
        if (h_loss) *h_loss = two[0];
        if (h_correct) *h_correct = (int32_t)two[1];
        return check_targets(t);
    }
    return TREXHIP_OK;
}

}  // namespace trexhip

struct trexhip_trainer { trexhip::Trainer* t; };

extern "C" {

using namespace trexhip;

size_t trexhip_weight_blob_bytes(int32_t classes, int32_t channels) {
    size_t n = 0;
    for (int k = 0; k < T_COUNT; ++k) n += tensor_count(k, classes, channels);
    return 32 + 4 * n;
}

int trexhip_trainer_create(trexhip_ctx* ctx, const void* blob, size_t bytes, const trexhip_train_params* p, trexhip_trainer** out) {
    if (!ctx || !blob || !p || !out) { set_error("trexhip_trainer_create: null argument"); return TREXHIP_E_INVALID; }
    *out = nullptr;
    if (bytes < 32) { set_error("trexhip_trainer_create: blob too small"); return TREXHIP_E_INVALID; }
    int32_t hdr[8];
    std::memcpy(hdr, blob, 32);
    if (hdr[0] != 0x57585254 || hdr[1] != 1) { set_error("trexhip_trainer_create: bad magic/version"); return TREXHIP_E_INVALID; }
    const int classes = hdr[2], CH = hdr[5];
    if (hdr[3] != 80 || hdr[4] != 80) { set_error("trexhip_trainer_create: only individual_image_size 80x80 is supported"); return TREXHIP_E_UNSUPPORTED; }
    if (CH != 1 && CH != 3) { set_error("trexhip_trainer_create: channels must be 1 or 3"); return TREXHIP_E_UNSUPPORTED; }
    if (classes < 1 || classes > 1024) { set_error("trexhip_trainer_create: classes must be 1..1024"); return TREXHIP_E_INVALID; }
    if (bytes != trexhip_weight_blob_bytes(classes, CH)) { set_error("trexhip_trainer_create: blob size does not match its header"); return TREXHIP_E_INVALID; }
    if (p->max_batch < 1 || p->max_batch > 4096) { set_error("trexhip_trainer_create: max_batch must be 1..4096"); return TREXHIP_E_INVALID; }
    if (p->precision != 0 && p->precision != 1) { set_error("trexhip_trainer_create: precision must be 0 (fp16 two-piece split convolutions) or 1 (exact fp32 MFMA)"); return TREXHIP_E_INVALID; }
    if (!(p->lr > 0.f) || !(p->beta1 >= 0.f && p->beta1 < 1.f) || !(p->beta2 >= 0.f && p->beta2 < 1.f) || !(p->eps > 0.f) ||
        !(p->dropout >= 0.f && p->dropout < 1.f) || !(p->bn_momentum >= 0.f && p->bn_momentum <= 1.f)) {
        set_error("trexhip_trainer_create: lr > 0, 0 <= beta < 1, eps > 0, 0 <= dropout < 1, 0 <= bn_momentum <= 1 are required");
        return TREXHIP_E_INVALID;
    }
    if (hipSetDevice(ctx->p.device) != hipSuccess) { set_error("trexhip_trainer_create: hipSetDevice failed"); return TREXHIP_E_DEVICE; }
    Trainer* t = new Trainer();
    t->ctx = ctx; t->classes = classes; t->CH = CH; t->max_n = p->max_batch; t->p = *p;
    size_t at = 0;
    for (int k = 0; k < T_COUNT; ++k) {
        t->off[k] = at; t->cnt[k] = tensor_count(k, classes, CH);
        at += (t->cnt[k] + 63) / 64 * 64;
    }
    t->off[T_COUNT] = at; t->total = at;
    const size_t n = (size_t)t->max_n;
    int rc = TREXHIP_OK;
#define TRY(x) do { if (rc == TREXHIP_OK) rc = (x); } while (0)
    TRY(dev_alloc(t, &t->P, at)); TRY(dev_alloc(t, &t->G, at)); TRY(dev_alloc(t, &t->M, at)); TRY(dev_alloc(t, &t->V, at));
    TRY(dev_alloc(t, &t->z1, n * 6400 * 16)); TRY(dev_alloc(t, &t->a1, n * 1600 * 16)); TRY(dev_alloc(t, &t->z2, n * 1600 * 64));
    TRY(dev_alloc(t, &t->a2, n * 400 * 64)); TRY(dev_alloc(t, &t->z3, n * 400 * 128)); TRY(dev_alloc(t, &t->a3, n * 100 * 128));
    TRY(dev_alloc(t, &t->da3, n * 100 * 128)); TRY(dev_alloc(t, &t->da2, n * 400 * 64)); TRY(dev_alloc(t, &t->da1, n * 1600 * 16));
    TRY(dev_alloc(t, &t->wb3, (size_t)4 * 25 * 32 * 64)); TRY(dev_alloc(t, &t->wb2, (size_t)2 * 25 * 32 * 32));
    TRY(dev_alloc(t, &t->wh2f, (size_t)2 * 2 * 25 * 64)); TRY(dev_alloc(t, &t->wh2b, (size_t)2 * 8 * 25 * 32));
    TRY(dev_alloc(t, &t->wh3f, (size_t)2 * 8 * 25 * 128)); TRY(dev_alloc(t, &t->wh3b, (size_t)2 * 16 * 25 * 64));
    TRY(dev_alloc(t, &t->wh_scale, 2));
    t->part_floats = std::max(std::max(std::max((size_t)SHARES3 * 25 * 64 * 128, (size_t)SHARES2 * 25 * 16 * 64), std::max((size_t)SHARES3H * 25 * 64 * 128, (size_t)SHARES2H * 25 * 16 * 64)),
                              n * 10 * CH * 400);
    TRY(dev_alloc(t, &t->part, t->part_floats)); TRY(dev_alloc(t, &t->part1, n * 10 * CH * 400)); TRY(dev_alloc(t, &t->red_bias, (size_t)3 * BWD_BLOCKS * 128));
    TRY(dev_alloc(t, &t->stat, (size_t)3 * 512));
    TRY(dev_alloc(t, &t->hpart, n * 100 * 100)); TRY(dev_alloc(t, &t->xhat, n * 100)); TRY(dev_alloc(t, &t->hd, n * 100));
    TRY(dev_alloc(t, &t->dl, n * classes)); TRY(dev_alloc(t, &t->dy, n * 100)); TRY(dev_alloc(t, &t->dh, n * 100));
    TRY(dev_alloc(t, &t->loss, n)); TRY(dev_alloc(t, &t->correct, n)); TRY(dev_alloc(t, &t->out2, 2)); TRY(dev_alloc(t, &t->bad_target, 2));
    TRY(dev_alloc(t, &t->red, std::max((size_t)BWD_BLOCKS * 2 * 128, n * 640)));   // conv1 leaves 20 n x 2 x 16 partials, conv2 / conv3 2 n x 2 x C
    TRY(dev_alloc(t, &t->keep, n * 308));
    TRY(dev_alloc(t, &t->x_stage, n * 6400 * CH)); TRY(dev_alloc(t, &t->y_stage, n)); TRY(dev_alloc(t, &t->keep_stage, n * 308));
#undef TRY
    if (rc == TREXHIP_OK && hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking) != hipSuccess) { set_error("trexhip_trainer_create: no second stream"); rc = TREXHIP_E_DEVICE; }
    for (hipEvent_t& e : t->ev)
        if (rc == TREXHIP_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { set_error("trexhip_trainer_create: no event"); rc = TREXHIP_E_DEVICE; }
    if (rc != TREXHIP_OK) { trainer_free(t); return rc; }
    std::vector<float> host(at, 0.f);
    const float* src = reinterpret_cast<const float*>(static_cast<const char*>(blob) + 32);
    for (int k = 0; k < T_COUNT; ++k) {
        for (size_t i = 0; i < t->cnt[k]; ++i) host[t->off[k] + to_internal(k, i, CH)] = src[i];
        src += t->cnt[k];
    }
    bool ok = hipMemcpy(t->P, host.data(), at * 4, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(t->bad_target, 0, 8) == hipSuccess;
    ok = ok && hipMemset(t->G, 0, at * 4) == hipSuccess && hipMemset(t->M, 0, at * 4) == hipSuccess && hipMemset(t->V, 0, at * 4) == hipSuccess;
    if (!ok) { trainer_free(t); set_error("trexhip_trainer_create: upload failed"); return TREXHIP_E_DEVICE; }
    trexhip_trainer* h = new trexhip_trainer{t};
    *out = h;
    return TREXHIP_OK;
}

void trexhip_trainer_destroy(trexhip_trainer* h) {
    if (!h) return;
    if (h->t) { (void)hipSetDevice(h->t->ctx->p.device); (void)hipStreamSynchronize(h->t->ctx->stream); trainer_free(h->t); }
    delete h;
}

int trexhip_trainer_set_lr(trexhip_trainer* h, float lr) {
    if (!h || !(lr > 0.f)) { set_error("trexhip_trainer_set_lr: a trainer and lr > 0 are required"); return TREXHIP_E_INVALID; }
    h->t->p.lr = lr;
    return TREXHIP_OK;
}

int64_t trexhip_trainer_steps(trexhip_trainer* h) { return h ? h->t->step : -1; }

int trexhip_train_step_device(trexhip_trainer* h, const float* d_inputs, const int32_t* d_targets, int32_t n, const uint8_t* d_keep_masks, float* loss,
                              int32_t* correct) {
    if (!h || !d_inputs || !d_targets) { set_error("trexhip_train_step_device: null argument"); return TREXHIP_E_INVALID; }
    if (n < 1 || n > h->t->max_n) { set_error("trexhip_train_step_device: n must be 1..max_batch"); return TREXHIP_E_INVALID; }
    return trainer_step(h->t, d_inputs, d_targets, n, d_keep_masks, loss, correct);
}

int trexhip_train_step(trexhip_trainer* h, const float* inputs, const int32_t* targets, int32_t n, const uint8_t* keep_masks, float* loss, int32_t* correct) {
    if (!h || !inputs || !targets) { set_error("trexhip_train_step: null argument"); return TREXHIP_E_INVALID; }
    Trainer* t = h->t;
    if (n < 1 || n > t->max_n) { set_error("trexhip_train_step: n must be 1..max_batch"); return TREXHIP_E_INVALID; }
    for (int i = 0; i < n; ++i)
        if (targets[i] < 0 || targets[i] >= t->classes) { set_error("trexhip_train_step: target class out of range"); return TREXHIP_E_INVALID; }   // visual_recognition_torch.py:1112
    TH_CHECK_HIP(hipSetDevice(t->ctx->p.device));
    hipStream_t s = t->ctx->stream;
    TH_CHECK_HIP(hipMemcpyAsync(t->x_stage, inputs, (size_t)n * 6400 * t->CH * sizeof(float), hipMemcpyHostToDevice, s));
    TH_CHECK_HIP(hipMemcpyAsync(t->y_stage, targets, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (keep_masks) TH_CHECK_HIP(hipMemcpyAsync(t->keep_stage, keep_masks, (size_t)n * 308, hipMemcpyHostToDevice, s));
    const int rc = trainer_step(t, t->x_stage, t->y_stage, n, keep_masks ? t->keep_stage : nullptr, loss, correct);
    if (rc != TREXHIP_OK) return rc;
    TH_CHECK_HIP(hipStreamSynchronize(s));        // the caller's buffers may go away
    return TREXHIP_OK;
}

int trexhip_train_eval_device(trexhip_trainer* h, const float* d_inputs, const int32_t* d_targets, int32_t n, float* loss, int32_t* correct) {
    if (!h || !d_inputs || !d_targets) { set_error("trexhip_train_eval_device: null argument"); return TREXHIP_E_INVALID; }
    if (n < 1 || n > h->t->max_n) { set_error("trexhip_train_eval_device: n must be 1..max_batch"); return TREXHIP_E_INVALID; }
    return trainer_eval(h->t, d_inputs, d_targets, n, loss, correct);
}

int trexhip_train_eval(trexhip_trainer* h, const float* inputs, const int32_t* targets, int32_t n, float* loss, int32_t* correct) {
    if (!h || !inputs || !targets) { set_error("trexhip_train_eval: null argument"); return TREXHIP_E_INVALID; }
    Trainer* t = h->t;
    if (n < 1 || n > t->max_n) { set_error("trexhip_train_eval: n must be 1..max_batch"); return TREXHIP_E_INVALID; }
    for (int i = 0; i < n; ++i)
        if (targets[i] < 0 || targets[i] >= t->classes) { set_error("trexhip_train_eval: target class out of range"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(t->ctx->p.device));
    hipStream_t s = t->ctx->stream;
    TH_CHECK_HIP(hipMemcpyAsync(t->x_stage, inputs, (size_t)n * 6400 * t->CH * sizeof(float), hipMemcpyHostToDevice, s));
    TH_CHECK_HIP(hipMemcpyAsync(t->y_stage, targets, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    return trainer_eval(t, t->x_stage, t->y_stage, n, loss, correct);
}

int trexhip_trainer_read(trexhip_trainer* h, int32_t tensor, int32_t kind, float* out, size_t count) {
    if (!h || !out || tensor < 0 || tensor >= T_COUNT || kind < 0 || kind > 3) { set_error("trexhip_trainer_read: bad argument"); return TREXHIP_E_INVALID; }
    Trainer* t = h->t;
    if (count != t->cnt[tensor]) { set_error("trexhip_trainer_read: count does not match the tensor"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(t->ctx->p.device));
    TH_CHECK_HIP(hipStreamSynchronize(t->ctx->stream));
    { const int rc = check_targets(t); if (rc != TREXHIP_OK) return rc; }
    const float* src = kind == 0 ? t->P : kind == 1 ? t->G : kind == 2 ? t->M : t->V;
    std::vector<float> tmp(count);
    TH_CHECK_HIP(hipMemcpy(tmp.data(), src + t->off[tensor], count * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < count; ++i) out[i] = tmp[to_internal(tensor, i, t->CH)];
    return TREXHIP_OK;
}

int trexhip_trainer_export(trexhip_trainer* h, void* blob, size_t capacity, size_t* bytes) {
    if (!h || !blob) { set_error("trexhip_trainer_export: null argument"); return TREXHIP_E_INVALID; }
    Trainer* t = h->t;
    const size_t need = trexhip_weight_blob_bytes(t->classes, t->CH);
    if (bytes) *bytes = need;
    if (capacity < need) { set_error("trexhip_trainer_export: buffer too small (trexhip_weight_blob_bytes)"); return TREXHIP_E_CAPACITY; }
    const int32_t hdr[8] = {0x57585254, 1, t->classes, 80, 80, t->CH, 0, 0};
    std::memcpy(blob, hdr, 32);
    float* dst = reinterpret_cast<float*>(static_cast<char*>(blob) + 32);
    for (int k = 0; k < T_COUNT; ++k) {
        const int rc = trexhip_trainer_read(h, k, 0, dst, t->cnt[k]);
        if (rc != TREXHIP_OK) return rc;
        dst += t->cnt[k];
    }
    return TREXHIP_OK;
}

}  // extern "C"
