// cnn_wpre.h -- the identity network's default chain (round 3): every Winograd-domain operand image is written by the PRODUCING layer.
// Included by cnn.hip inside namespace trexhip, after the helpers it uses (wino_bt, split2h_pair, mfma16, buf_load16, WinoGeom...).
//
// Network: visual_identification_network_torch.py:184-258 (V118_3, eval mode).  Same arithmetic as k_conv5_wino / k_conv5_wino1
// (F(4,5) Winograd along x, direct along y, fp16 two-piece split of both operands, three piece products, fp32 accumulate):
//   k_conv1_wpre  u8 crop -> conv1 (matrix cores) -> bias, ReLU, pool -> B^T d along x -> fp16 pieces -> V2
//   k_conv2_wpre2 V2 -> conv2 (40 position GEMMs) -> A^T, bias, ReLU, pool -> B^T d -> fp16 pieces -> V3
//   k_conv5_wpre  V3 -> conv3 -> A^T, bias, ReLU, pool -> act3 (fp32, NHWC) -> fc1 -> head
// (since round 4 the first two run as ONE kernel for 1-channel crops -- cnn_fused12.h: k_conv12_wpre produces V2's rows straight into conv2's LDS row
// ring; k_conv1_wpre + k_conv2_wpre2 below serve 3-channel crops and TREXHIP_CONV_GEOM bit 28)
// so the consumers' staging is a plain 16-byte copy HBM -> VGPR -> LDS (no transform, no split, no 4-byte LDS scatter inside the
// matrix-bound tap loops), and fp32 activations of conv1 / conv2 never travel through HBM.
//   V2 [q2 = crop*40 + y][piece 2][group 2][pg 4][tx 10][16 ci] halves   (positions of group 0: 0,1,2,7; group 1: 3,4,5,6)  5120 B per row
//   V3 [q3 = crop*20 + y][chunk 4][piece 2][position 8][tx 5][16 ci] halves                                                 10240 B per row
// The fp16 range guard moves to the producers: an activation >= 4368 (|B^T d| <= 15 max|d| < 65520) or a NaN raises the flag and the
// host-side guard re-runs the layer stack with the bf16 kernels from the crops.

static constexpr int V2_ROWB = 5120, V3_ROWB = 10240;
// V3 since round 5 (k_conv5_wpair, cnn_conv3p.h): a row's 10240 bytes are [position pair 4][piece 2][chunk 4][position of the pair 2][tx 5][16 ci],
// pairs in the order (1,2) (3,4) (5,6) (0,7) -- conv3 walks a pass pair by pair (all 64 input channels of two positions = one 1280-byte plane
// per row and piece), so that the pairs the output transform combines are final one after the other.  -DTREXHIP_V3_OLD: [chunk][piece][position 8]
// [tx][16 ci] for k_conv5_wpre.  The producers' stores of position p, chunk c, piece pc, tile tx: v3_off(p, pc) + c * V3_CHUNKB + tx * 32.
#ifdef TREXHIP_V3_OLD
static constexpr bool V3_PAIR = false;
static constexpr int V3_CHUNKB = 2560;
__host__ __device__ constexpr int v3_off(const int p, const int piece) { return piece * 1280 + p * 160; }
#else
static constexpr bool V3_PAIR = true;
static constexpr int V3_CHUNKB = 320;
__host__ __device__ constexpr int v3_pair_of(const int p) { return p == 0 || p == 7 ? 3 : (p - 1) / 2; }
__host__ __device__ constexpr int v3_half_of(const int p) { return p == 0 ? 0 : p == 7 ? 1 : (p - 1) % 2; }
__host__ __device__ constexpr int v3_off(const int p, const int piece) { return v3_pair_of(p) * 2560 + piece * 1280 + v3_half_of(p) * 160; }
#endif

// one LDS-DMA instruction: 64 lanes x 16 bytes from the lanes' global addresses (wave-uniform 64-bit base in SGPRs + a 32-bit lane offset: one
// address register instead of two) to LDS [lds_addr, lds_addr + 1024).  Raw, so that the compiler's wait-count pass does not know of it: it
// cannot tell the staging buffer from the one being read and would drain the vector-memory queue (the weight fragments in flight included) in
// front of every LDS read of the tap loop.  M0 is the compiler's: handed back as found.
__device__ __forceinline__ unsigned long long wave_uniform64(const unsigned long long v) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ void wpre_dma16(const unsigned long long sbase, const uint32_t voff, const uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

// 8 consecutive pixels x 8 channels (lo = channels 0..3, hi = 4..7 of each pixel) -> per position the two fp16 pieces of the 8 channels
__device__ __forceinline__ void wino_pack8(const float4* __restrict__ lo, const float4* __restrict__ hi, uint4* __restrict__ o1, uint4* __restrict__ o2) {
    uint32_t w1[8][4], w2[8][4];
#define WP_PAIR(cp_, src_, ca_, cb_)                                                                                              \
    do {                                                                                                                          \
        float ua_[8], ub_[8];                                                                                                     \
        wino_bt(src_[0].ca_, src_[1].ca_, src_[2].ca_, src_[3].ca_, src_[4].ca_, src_[5].ca_, src_[6].ca_, src_[7].ca_, ua_);     \
        wino_bt(src_[0].cb_, src_[1].cb_, src_[2].cb_, src_[3].cb_, src_[4].cb_, src_[5].cb_, src_[6].cb_, src_[7].cb_, ub_);     \
        _Pragma("unroll") for (int p_ = 0; p_ < 8; ++p_) split2h_pair(ua_[p_], ub_[p_], w1[p_][cp_], w2[p_][cp_]);                \
    } while (0)
    WP_PAIR(0, lo, x, y);
    WP_PAIR(1, lo, z, w);
    WP_PAIR(2, hi, x, y);
    WP_PAIR(3, hi, z, w);
#undef WP_PAIR
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        o1[p] = make_uint4(w1[p][0], w1[p][1], w1[p][2], w1[p][3]);
        o2[p] = make_uint4(w2[p][0], w2[p][1], w2[p][2], w2[p][3]);
    }
}

// ------------------------------------------------------------------------------------------------
// conv1 (1 or 3 input channels) with the V2 epilogue.  The matrix-core part is k_conv1_mfma / k_conv1_mfma3's (same fragments, same
// products, same order); the crop is worked in 5 bands of 8 pooled rows = 20 M-tiles: a band's activations go to LDS as fp32, then 160
// threads take (row, tile of 4 pixels, channel octet) items: 8 pixels x 8 channels -> B^T d -> pieces -> 16 x 16-byte stores.
// ------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void k_conv1_wpre(const uint8_t* __restrict__ crops /*[N][80][80][CH]*/, const uint4* __restrict__ wtab,
                                                    const float* __restrict__ bias, uint8_t* __restrict__ v2, const float inv_scale,
                                                    uint32_t* __restrict__ overflow) {
    constexpr int S = 80, PH = 84, PITCH = 88, PLANE = PH * PITCH;
    constexpr int PBP = 20;                                              // floats per pooled pixel in the band buffer (16 + 4: the four q groups of a store hit two bank sets)
    __shared__ __attribute__((aligned(16))) _Float16 img[CH * PLANE];
    __shared__ __attribute__((aligned(16))) float pb[320 * PBP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int crop = blockIdx.x;
    for (int i = tid; i < CH * PLANE * 2 / 16; i += 256) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if constexpr (CH == 1) {
        const uint8_t* src = crops + (size_t)crop * S * S;
        for (int i = tid; i < S * S / 16; i += 256) {
            const int y = i / (S / 16), c16 = (i - y * (S / 16)) * 16;
            const uint4 v = reinterpret_cast<const uint4*>(src)[i];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            uint32_t* d = reinterpret_cast<uint32_t*>(img + (y + 2) * PITCH + 2 + c16);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t b0 = (w4[k >> 1] >> (16 * (k & 1))) & 0xffu, b1 = (w4[k >> 1] >> (16 * (k & 1) + 8)) & 0xffu;
                d[k] = pack_h2((_Float16)(float)b0, (_Float16)(float)b1);
            }
        }
    } else {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(crops + (size_t)crop * S * S * 3);
        for (int i = tid; i < S * S / 4; i += 256) {
            const int y = i / (S / 4), x = (i - y * (S / 4)) * 4;
            const uint32_t w0 = src[3 * i], w1 = src[3 * i + 1], w2 = src[3 * i + 2];
            const uint32_t by[12] = {w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff, w0 >> 24, w1 & 0xff, (w1 >> 8) & 0xff, (w1 >> 16) & 0xff, w1 >> 24,
                                     w2 & 0xff, (w2 >> 8) & 0xff, (w2 >> 16) & 0xff, w2 >> 24};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                uint32_t* d = reinterpret_cast<uint32_t*>(img + c * PLANE + (y + 2) * PITCH + 2 + x);
#pragma unroll
                for (int k = 0; k < 2; ++k) d[k] = pack_h2((_Float16)(float)by[(2 * k) * 3 + c], (_Float16)(float)by[(2 * k + 1) * 3 + c]);
            }
        }
    }
    constexpr int NF = CH == 1 ? 16 : 32;
    uint4 bf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) bf[f] = wtab[f * 64 + lane];
    __syncthreads();
    const int r = lane & 15, q = lane >> 4;
    const int co = r;
    const float bz = bias[co];
    bool ovf = false;
    uint8_t* vc = v2 + (size_t)crop * 40 * V2_ROWB;
    for (int band = 0; band < 5; ++band) {
        for (int ti = 0; ti < 5; ++ti) {
            const int tl = wave + 4 * ti, tile = band * 20 + tl;
            const int wdx = tile * 8 + (r >> 1);
            const int yp = wdx / 20, x4 = (wdx - yp * 20) * 4;
            const int row = 2 * yp + (r & 1);
            f32x4 acc[4];
            if constexpr (CH == 1) {
                const _Float16* p1 = img + (row + q) * PITCH + x4;
                const _Float16* p2 = img + (row + 4) * PITCH + x4;
                uint4 a1u, a2u;
                { const uint2 l2 = *reinterpret_cast<const uint2*>(p1), h2 = *reinterpret_cast<const uint2*>(p1 + 4); a1u = make_uint4(l2.x, l2.y, h2.x, h2.y); }
                { const uint2 l2 = *reinterpret_cast<const uint2*>(p2), h2 = *reinterpret_cast<const uint2*>(p2 + 4); a2u = make_uint4(l2.x, l2.y, h2.x, h2.y); }
                const f16x8_c1 a1 = __builtin_bit_cast(f16x8_c1, a1u), a2 = __builtin_bit_cast(f16x8_c1, a2u);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 1]), c, 0, 0, 0);   // low pieces first
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 3]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 2]), c, 0, 0, 0);
                    acc[s] = c;
                }
            } else {
                f16x8_c1 a[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const _Float16* pp = m < 3 ? img + m * PLANE + (row + q) * PITCH + x4 : img + (q < 3 ? q : 0) * PLANE + (row + 4) * PITCH + x4;
                    const uint2 l2 = *reinterpret_cast<const uint2*>(pp), h2 = *reinterpret_cast<const uint2*>(pp + 4);
                    a[m] = __builtin_bit_cast(f16x8_c1, make_uint4(l2.x, l2.y, h2.x, h2.y));
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 4; ++m) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], __builtin_bit_cast(f16x8_c1, bf[(s * 4 + m) * 2 + 1]), c, 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], __builtin_bit_cast(f16x8_c1, bf[(s * 4 + m) * 2 + 0]), c, 0, 0, 0);
                    acc[s] = c;
                }
            }
            // lane (co, q): pooled pixels 4q .. 4q+3 of this tile, channel co
            float* pw = pb + (tl * 16 + 4 * q) * PBP + co;
#pragma unroll
            for (int pos = 0; pos < 2; ++pos) {
                const float m0 = fmaxf(fmaxf(acc[0][2 * pos], acc[0][2 * pos + 1]), fmaxf(acc[1][2 * pos], acc[1][2 * pos + 1]));
                const float m1 = fmaxf(fmaxf(acc[2][2 * pos], acc[2][2 * pos + 1]), fmaxf(acc[3][2 * pos], acc[3][2 * pos + 1]));
                const float v0 = fmaxf(m0 * inv_scale + bz, 0.f), v1 = fmaxf(m1 * inv_scale + bz, 0.f);
                ovf |= !(v0 < 4368.0f) | !(v1 < 4368.0f);
                pw[(2 * pos) * PBP] = v0;
                pw[(2 * pos + 1) * PBP] = v1;
            }
        }
        __syncthreads();
        if (tid < 160) {
            const int row = tid / 20, rem = tid - row * 20, tx = rem >> 1, oct = rem & 1;
            float4 lo[8], hi[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int x = 4 * tx - 2 + k;
                lo[k] = make_float4(0.f, 0.f, 0.f, 0.f); hi[k] = lo[k];
                if (x >= 0 && x < 40) {
                    const float4* s4 = reinterpret_cast<const float4*>(pb + (row * 40 + x) * PBP + oct * 8);
                    lo[k] = s4[0]; hi[k] = s4[1];
                }
            }
            uint4 o1[8], o2[8];
            wino_pack8(lo, hi, o1, o2);
            uint8_t* d = vc + (size_t)(band * 8 + row) * V2_ROWB + tx * 32 + oct * 16;
#pragma unroll
            for (int pi = 0; pi < 8; ++pi) {
                const int p = pi < 3 ? pi : (pi == 3 ? 7 : pi - 1);      // stored order 0,1,2,7 | 3,4,5,6: the position groups of k_conv2_wpre2
                *reinterpret_cast<uint4*>(d + pi * 320) = o1[p];
                *reinterpret_cast<uint4*>(d + 2560 + pi * 320) = o2[p];
            }
        }
        __syncthreads();
    }
    if (__any(ovf) && lane == 0) atomicOr(overflow, 3u);      // bit 1: this kernel does not know the crop (k_guard_plan: every crop)
}

// ------------------------------------------------------------------------------------------------
// k_conv2_wpre2: conv2 on V2 -> V3 with TWO workgroups per CU, so that one workgroup's epilogue (A^T, pool, B^T, split: about as many
// VALU cycles as the tap loop has matrix cycles -- conv2 has a single 16-channel chunk) runs under the other's MFMAs.
//   * one M-tile per wave (128 accumulator registers, <= 256 registers per lane), 4 waves = 2 co-tiles x 2 M-groups, a pass = 3 row
//     pairs = 60 tiles in 64 M-slots; 10 input rows x 2 pieces x 2 position groups = 50 KB + 15 KB epilogue buffer per workgroup;
//   * no staging inside the tap loops at all: an HBM miss in the vector memory queue holds back every weight fragment issued behind it
//     (loads return in order), measured 1.0-1.2 ms per 25600 crops.  Both position groups of the NEXT pass are fetched at the start of
//     the epilogue by LDS-DMA (global_load_lds_dwordx4: no registers, no LDS store instructions), wave w = (group, piece) plane w,
//     and have landed by the epilogue's first barrier;
//   * LDS rows are linear (a DMA instruction writes 64 x 16 contiguous bytes); bank conflicts of the A reads are kept down by rotating
//     each row's 16-byte units within their 256-byte blocks by rot(slot) = (slot & 1) + 4 ((slot >> 1) & 3): applied to the SOURCE
//     address of the DMA and to the read address alike.
// ------------------------------------------------------------------------------------------------
struct W2bGeom {
    static constexpr int CO = 64, S = 40, TPP = 20, RPP = 3, NR = 2 * RPP + 4;
    static constexpr int ROWL = 1280;                                   // one (row, piece, group): 4 positions x 10 tiles x 32 B, in V2 and in LDS
    static constexpr int PLANE = (NR + 1) * ROWL, BUF = 2 * PLANE;      // slot 0 = the zero row; BUF = the two pieces of one position group
    static constexpr int PBUF_OFF = 2 * BUF, PBUF = RPP * 20 * 64 * 4;
    static constexpr int LDS_BYTES = PBUF_OFF + PBUF;
    static constexpr int BV = 2 * 2 * CO;
    static_assert(2 * (LDS_BYTES + 64) <= 160 * 1024, "two workgroups per CU");
};
__device__ __forceinline__ int w2b_rot(const int slot) { return (slot & 1) + ((slot & 6) << 1); }

template <int DBG = 0, int STAGGER = 5, int AD = 1, int BD = 3>      // DBG (dev builds): 1 no staging, 2 no epilogue, 4 no weight loads, 8 no A reads, 32 no V3 transform; STAGGER: x 1024 cycles; AD / BD: taps of lead of the A / weight fragments
__global__ __launch_bounds__(256, 2) void k_conv2_wpre2(const uint8_t* __restrict__ v2, const uint4* __restrict__ wp /*[5][8][2][2][64] x 16 B*/,
                                                        const float* __restrict__ bias, uint8_t* __restrict__ v3, const float out_scale,
                                                        uint32_t* __restrict__ overflow, const int n_crops, uint32_t* __restrict__ pass_ctr) {
    using G = W2bGeom;
    constexpr int CO = 64, S = 40;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsb[];
    __shared__ int s_next_pass;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = wave & 1, mg = wave >> 1;
    const int total_pairs = n_crops * (S / 2);
    const int n_pass = (total_pairs + G::RPP - 1) / G::RPP;
    constexpr int PK = 4;                                                // consecutive passes per ticket: their halo rows are L2 hits
    int pass = blockIdx.x * PK;
    if (pass >= n_pass) return;
    for (int i = tid; i < 4 * (G::ROWL / 16); i += 256) {                // the zero rows of the four planes
        const int pl = i / (G::ROWL / 16), o = i - pl * (G::ROWL / 16);
        *reinterpret_cast<uint4*>(ldsb + pl * G::PLANE + o * 16) = make_uint4(0, 0, 0, 0);
    }
#define W2B_ROWS(pass_, qmin_, nrows_)                                                                                           \
    do {                                                                                                                         \
        const int gp0_ = (pass_) * G::RPP;                                                                                       \
        int gpl_ = gp0_ + G::RPP - 1;                                                                                            \
        gpl_ = gpl_ < total_pairs ? gpl_ : total_pairs - 1;                                                                      \
        const int y0_ = (2 * gp0_) % S, yl_ = (2 * gpl_) % S + 1;                                                                \
        qmin_ = 2 * gp0_ - (y0_ >= 2 ? 2 : 0);                                                                                   \
        nrows_ = 2 * gpl_ + 1 + (yl_ + 2 <= S - 1 ? 2 : 0) - qmin_ + 1;                                                          \
    } while (0)
    // rows [lo_, hi_) (even bounds) of plane (group = wave >> 1, piece = wave & 1), HBM -> LDS.  The row slots are a RING: batch row q lives
    // in slot q % NR + 1, so the 4 halo rows that consecutive passes share stay where they are and only the 6 new rows of a pass are fetched
    // (the DMA bytes and the HBM reads of the layer fall by 40 %).  Two rows = 2560 contiguous bytes = 2.5 DMA instructions.
#define W2B_DMA(lo_, hi_)                                                                                                        \
    do {                                                                                                                         \
        const uint8_t* src_ = v2 + (wave & 1) * 2560 + (wave >> 1) * 1280;                                                       \
        uint8_t* dst_ = ldsb + (wave >> 1) * G::BUF + (wave & 1) * G::PLANE;                                                     \
        _Pragma("unroll 1") for (int q_ = (lo_); q_ < (hi_); q_ += 2) {                                                          \
            const int slot0_ = q_ % G::NR + 1;                                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                                   \
                const int o_ = i_ * 1024 + lane * 16;                                                                            \
                if (o_ < 2 * G::ROWL) {                                                                                          \
                    const int r_ = o_ >= G::ROWL ? 1 : 0;                                                                        \
                    const int wl_ = (o_ - r_ * G::ROWL) >> 4;                                                                    \
                    const int w_ = (wl_ & ~15) | ((wl_ - w2b_rot(slot0_ + r_)) & 15);                                            \
                    int row_ = q_ + r_;                                                                                          \
                    row_ = row_ < total_rows ? row_ : total_rows - 1;                                                            \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + (size_t)row_ * V2_ROWB + w_ * 16), \
                                                     (__attribute__((address_space(3))) void*)(dst_ + slot0_ * G::ROWL + i_ * 1024), 16, 0, 0); \
                }                                                                                                                \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    const int total_rows = n_crops * S;

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (uint32_t)(40 * G::BV * 16));
    const int boff = (h * CO + n * 32 + j) * 16;
    const int co = n * 32 + j;
    const float bz = bias[co];
    int qmin, nrows;
    W2B_ROWS(pass, qmin, nrows);
    if (!(DBG & 1)) W2B_DMA(qmin, qmin + nrows);
    int res_hi = qmin + nrows;                                           // rows [this pass's qmin, res_hi) are resident
    if (tid == 0) s_next_pass = ((int)atomicAdd(pass_ctr, 1u) + (int)gridDim.x) * PK;
    __syncthreads();
#define W2_POS(tau_) (((tau_) / 20) == 0 ? ((tau_) % 4 == 3 ? 7 : (tau_) % 4) : 3 + (tau_) % 4)
#define W2_BOFF(tau_) (((((tau_) % 20) / 4) * 8 + W2_POS(tau_)) * G::BV * 16)
    uint4 bq[8][2];
#pragma unroll
    for (int t = 0; t < BD; ++t) { bq[t][0] = buf_load16(wrs, boff, W2_BOFF(t)); bq[t][1] = buf_load16(wrs, boff, W2_BOFF(t) + 2 * CO * 16); }
    bool ovf = false;
    // the two workgroups of a CU start together and would keep step -- both in their tap loops, then both in their epilogues -- with
    // nothing to overlap: the second half of the grid starts half a pass late
    if (STAGGER > 0 && blockIdx.x >= gridDim.x / 2) {
#pragma unroll 1
        for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
    }
    for (;;) {
        // A-operand byte offsets of this lane's tile: per kernel row the row slot (out-of-crop rows -> the zero row), per position of a
        // group the rotated unit
        int aoff[5][4];
        {
            int s = mg * 32 + j;
            s = s < G::RPP * G::TPP ? s : G::RPP * G::TPP - 1;
            const int rp = s / G::TPP, r2 = s - rp * G::TPP;
            int gp = pass * G::RPP + rp;
            gp = gp < total_pairs ? gp : total_pairs - 1;
            const int tx = r2 >> 1, qo = 2 * gp + (r2 & 1), y = qo % S;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int iy = y + ky - 2;
                const int slot = (iy >= 0 && iy < S) ? (qo + ky - 2) % G::NR + 1 : 0;
                const int rot = w2b_rot(slot);
#pragma unroll
                for (int pg = 0; pg < 4; ++pg) {
                    const int w = pg * 20 + tx * 2 + h;
                    aoff[ky][pg] = slot * G::ROWL + ((w & ~15) | ((w + rot) & 15)) * 16;
                }
            }
        }
        f32x16 acc[8];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // the tap loop outranks the other workgroup's epilogue on the shared issue port: MFMA and VALU instructions are arbitrated by
        // priority, then age, and an older wave in its (VALU-dense) epilogue would leave a younger wave's MFMAs only the leftover slots
        if (!(DBG & 64)) __builtin_amdgcn_s_setprio(3);
        // operand fetches are issued AD taps (A fragments, LDS) and BD taps (weight fragments, L2) ahead.  Measured: (1, 3), (2, 4), (2, 6) and
        // (3, 7) all take 3.83-3.95 ms per 25600 crops -- the waves' waits (SQ_WAIT_ANY 52 %) are the pass's barriers and its DMA, not these
        uint4 af[AD + 1][2];                                              // ring over taps: [tap % (AD + 1)][piece]
#define W2B_AREAD(dst_, tau_)                                                                                                    \
        do {                                                                                                                     \
            const uint8_t* an_ = ldsb + ((tau_) / 20) * G::BUF + aoff[((tau_) % 20) / 4][(tau_) % 4];                            \
            dst_[0] = *reinterpret_cast<const uint4*>(an_);                                                                      \
            dst_[1] = *reinterpret_cast<const uint4*>(an_ + G::PLANE);                                                           \
        } while (0)
#pragma unroll
        for (int t = 0; t < AD; ++t) W2B_AREAD(af[t], t);
#pragma clang loop unroll(full)
        for (int tau = 0; tau < 40; ++tau) {
            const int tl = tau % 20;
            if (!(DBG & 8) && tau + AD < 40) W2B_AREAD(af[(tau + AD) % (AD + 1)], tau + AD);
            if (!(DBG & 4)) {
                const int wt = W2_BOFF((tau + BD) % 40);
                bq[(tau + BD) % 8][0] = buf_load16(wrs, boff, wt);
                bq[(tau + BD) % 8][1] = buf_load16(wrs, boff, wt + 2 * CO * 16);
            }
            const int p = W2_POS(tau);
            const f16x8 b1 = __builtin_bit_cast(f16x8, bq[tau % 8][0]), b2 = __builtin_bit_cast(f16x8, bq[tau % 8][1]);
            const f16x8 a1 = __builtin_bit_cast(f16x8, af[tau % (AD + 1)][0]), a2 = __builtin_bit_cast(f16x8, af[tau % (AD + 1)][1]);
            acc[p] = mfma16(a2, b1, tl < 4 ? zero16 : acc[p]);            // kernel row 0 starts the accumulator
            acc[p] = mfma16(a1, b2, acc[p]);
            acc[p] = mfma16(a1, b1, acc[p]);
            // nothing moves across a tap: left alone, the scheduler sinks every operand fetch down to its use (to save registers) and the
            // wave then waits out the full LDS / L2 latency in front of each MFMA
            if (!(DBG & 512)) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // the A fragments of tap + AD ...
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);       // ... and the weight fragments of tap + BD go first,
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // then the three MFMAs with the address arithmetic in between
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef W2B_AREAD
        if (!(DBG & 64)) __builtin_amdgcn_s_setprio(0);
        __syncthreads();                                                  // every wave is done with the operand planes
        const bool draw = pass % PK == PK - 1;                            // the last pass of a ticket moves on to the next ticket
        const int next_pass = draw ? s_next_pass : pass + 1;              // (written at the end of an earlier epilogue, or in the prologue)
        const bool have_next = next_pass < n_pass;
        int qmin_n = qmin, nrows_n = nrows;
        uint32_t ticket = 0;
        // the ticket after the next one; its value is needed at the end of the epilogue only.  Raw instruction: atomicAdd() goes through the
        // compiler's wave-reduction form, which waits for the returned value (and with it for every weight fragment in flight) on the spot
        if (draw && tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(pass_ctr), "v"(1u) : "memory");
        if (have_next) {
            W2B_ROWS(next_pass, qmin_n, nrows_n);
            // the next pass of the same ticket: rows below res_hi are already there (its qmin is not below this pass's)
            const int lo_new = (next_pass == pass + 1 && res_hi > qmin_n && !(DBG & 256)) ? res_hi : qmin_n;
            if (!(DBG & 1)) W2B_DMA(lo_new, qmin_n + nrows_n);
            res_hi = qmin_n + nrows_n;
        }
        if (DBG & 2) {
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" :: "a"(acc[p]));
        }
        // epilogue 1: Y = A^T M, pool, bias, ReLU -> the pass's 3 x 20 x 64 activations as fp32 in LDS
        float* pbuf = reinterpret_cast<float*>(ldsb + G::PBUF_OFF);
        if (!(DBG & 2)) {
            f32x16 y0, y1, y2, y3;
            {
                const f32x16 e1 = acc[1] + acc[2], o1 = acc[1] - acc[2];
                y0 = acc[0] + e1; y1 = o1; y2 = e1; y3 = o1 + acc[7];
            }
            {
                const f32x16 e2 = acc[3] + acc[4], o2 = acc[3] - acc[4];
                y0 += e2; y1 += 2.f * o2; y2 += 4.f * e2; y3 += 8.f * o2;
            }
            {
                const f32x16 e3 = acc[5] + acc[6], o3 = acc[5] - acc[6];
                y0 += e3; y1 += 0.5f * o3; y2 += 0.25f * e3; y3 += 0.125f * o3;
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 2 * rr;
                const int s = mg * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // even: rows y, y+1 of one tile column
                const float v0 = fmaxf(fmaxf(y0[r], y1[r]), fmaxf(y0[r + 1], y1[r + 1]));
                const float v1 = fmaxf(fmaxf(y2[r], y3[r]), fmaxf(y2[r + 1], y3[r + 1]));
                if (s < G::RPP * G::TPP) {
                    const int rp = s / G::TPP, tx = (s - rp * G::TPP) >> 1;
                    const float a0 = fmaxf(v0 * out_scale + bz, 0.f), a1 = fmaxf(v1 * out_scale + bz, 0.f);
                    ovf |= !(a0 < 4368.0f) | !(a1 < 4368.0f);
                    float* o = pbuf + (rp * 20 + 2 * tx) * 64 + co;
                    o[0] = a0;
                    o[64] = a1;
                }
            }
        }
        if (DBG & 128) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            // raw barrier: the activations in LDS are what the second half needs; the DMA keeps flying until the end of the epilogue
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // epilogue 2: (pooled row, conv3 tile, channel quad) items -> V3
        if (!(DBG & (2 | 32)) && tid < 240) {
            const int rp = tid / 80, rem = tid - rp * 80, tx = rem >> 4, quad = rem & 15;
            const int gp = pass * G::RPP + rp;                            // = q3: pooled row of the batch
            if (gp < total_pairs) {
                // the reads are invisible to the compiler's wait-count pass on purpose: it cannot tell pbuf from the operand planes the
                // LDS-DMA above is still filling and would drain that DMA (s_waitcnt vmcnt(0)) in front of every one of them
                typedef float f32x4n __attribute__((ext_vector_type(4)));
                f32x4n d[8];
                const uint32_t pa = (uint32_t)(uintptr_t)(pbuf + rp * 20 * 64 + quad * 4);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int x = 4 * tx - 2 + k;
                    x = x < 0 ? 0 : (x > 19 ? 19 : x);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(d[k]) : "v"(pa + (uint32_t)x * 256u));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = 4 * tx - 2 + k;
                    if (x < 0 || x >= 20) d[k] = f32x4n{0.f, 0.f, 0.f, 0.f};
                }
                float ua[8], ub[8], uc[8], ud[8];
                wino_bt(d[0].x, d[1].x, d[2].x, d[3].x, d[4].x, d[5].x, d[6].x, d[7].x, ua);
                wino_bt(d[0].y, d[1].y, d[2].y, d[3].y, d[4].y, d[5].y, d[6].y, d[7].y, ub);
                wino_bt(d[0].z, d[1].z, d[2].z, d[3].z, d[4].z, d[5].z, d[6].z, d[7].z, uc);
                wino_bt(d[0].w, d[1].w, d[2].w, d[3].w, d[4].w, d[5].w, d[6].w, d[7].w, ud);
                uint8_t* dst = v3 + (size_t)gp * V3_ROWB + (quad >> 2) * V3_CHUNKB + tx * 32 + (quad & 3) * 8;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    uint32_t l0, l1, m0, m1;
                    split2h_pair(ua[p], ub[p], l0, m0);
                    split2h_pair(uc[p], ud[p], l1, m1);
                    *reinterpret_cast<uint2*>(dst + v3_off(p, 0)) = make_uint2(l0, l1);
                    *reinterpret_cast<uint2*>(dst + v3_off(p, 1)) = make_uint2(m0, m1);
                }
            }
        }
        if (!have_next) break;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");    // this wave's DMA has landed (its V3 stores have reached L2, its ticket is back) ...
        if (draw && tid == 0) s_next_pass = ((int)ticket + (int)gridDim.x) * PK;   // read behind a later pass's first barrier
        __syncthreads();                                                  // ... and behind the barrier everybody's: the next pass's planes are complete
        pass = next_pass; qmin = qmin_n; nrows = nrows_n;
    }
#undef W2B_ROWS
#undef W2B_DMA
#undef W2_POS
#undef W2_BOFF
    if (__any(ovf) && lane == 0) atomicOr(overflow, 3u);      // bit 1: this kernel does not know the crop (k_guard_plan: every crop)
}

// ------------------------------------------------------------------------------------------------
// conv3 (64 -> 128 channels, 20x20) on V3: k_conv5_wino<64,128,20,2>'s tap loop and epilogue, with the staging of a 16-channel chunk
// reduced to 13 x (16-byte load, 16-byte LDS store) per thread in two batches (7 loads at tap 0, 6 at tap 19; the stores follow one
// per tap 12 / 11 taps later).  LDS layout unchanged: [buffer][piece][row slot][position][tx][16 ci], a (row, piece) is 1280
// contiguous bytes both in V3 and in LDS.
// ------------------------------------------------------------------------------------------------
template <int DBG = 0, int BD = 7, int PK = 4, int STG = 1, int DT0 = 2>      // PK: consecutive passes per workgroup and ticket (their halo rows are then L2 hits); STG: 1 = staging by LDS-DMA from tap DT0 on, 0 = through registers
__global__ __launch_bounds__(256) void k_conv5_wpre(const uint8_t* __restrict__ v3, const uint4* __restrict__ wp /*[4][5][8][2][2][128] x 16 B*/,
                                                    const float* __restrict__ bias, float* __restrict__ out, const float out_scale,
                                                    const int n_crops, uint32_t* __restrict__ pass_ctr,
                                                    const int n_big /* tickets of PK passes; the passes behind them go out one by one, so that the workgroups finish together */) {
    constexpr int CI = 64, CO = 128, S = 20, TPW = 2;
    using G = WinoGeom<CI, CO, S, TPW>;
    static_assert(G::NTHR == 256 && G::RP0 == 1280 && G::NCH * 2 * G::RP0 == V3_ROWB, "geometry");
    constexpr int NU = G::NR * 2 * 80, NIT = (NU + 255) / 256;          // 16-byte units per chunk, per thread
    static_assert(NIT == 13, "one batch of 13 units");
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsb[];
    __shared__ int s_next_pass;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = wave % G::NT, mg = wave / G::NT;
    const int total_tiles = n_crops * G::TPC;
    const int n_pass = (total_tiles + G::MB - 1) / G::MB;
    const int big_end = n_big * PK;
    auto ticket_first = [&](const int t) { return t < n_big ? t * PK : big_end + (t - n_big); };
    int pass = ticket_first((int)blockIdx.x);
    if (pass >= n_pass) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)ldsb;
    // Staging of a chunk by LDS-DMA (no registers, no LDS store instructions -- a 16-byte ds_write costs the whole CU ~30 cycles): the NR padded
    // rows of a piece plane are one linear LDS range of NR * RP bytes = NI instructions of 1 KB; lane l of instruction ii covers byte
    // ii * 1024 + 16 l of it = (row, unit) by one division -- units in a row's pad fetch the row's last unit, lanes past the range stay off --
    // and supplies the offset of its own unit.  Wave w issues the instructions w, w + 4, ...: at most NDW per wave and chunk, one per tap.
    constexpr int NI = (G::NR * G::RP + 1023) / 1024, NDW = (2 * NI + 3) / 4;
#define W3_BASE(cc_, qmin_) wave_uniform64(reinterpret_cast<unsigned long long>(v3) + (unsigned long long)(qmin_) * V3_ROWB + (unsigned)((cc_) * 2560))
#define W3_DMA(k_, sbase_, nrows_, buf_)                                                                                         \
    do {                                                                                                                         \
        const int i_ = wave + 4 * (k_);                                                                                          \
        if (i_ < 2 * NI) {                                                                                                       \
            const int pc_ = i_ >= NI ? 1 : 0, ii_ = i_ - pc_ * NI;                                                               \
            const int o_ = ii_ * 1024 + lane * 16;                                                                               \
            if (o_ < G::NR * G::RP) {                                                                                            \
                int row_ = o_ / G::RP, w_ = o_ - row_ * G::RP;                                                                   \
                w_ = w_ < G::RP0 ? w_ : G::RP0 - 16;                                                                             \
                row_ = row_ < (nrows_) ? row_ : (nrows_) - 1;                                                                    \
                wpre_dma16(sbase_, (uint32_t)(row_ * V3_ROWB + pc_ * 1280 + w_),                                                 \
                           lds0 + (uint32_t)((buf_) * G::BUF + pc_ * G::PLANE + G::RP + ii_ * 1024));                            \
            }                                                                                                                    \
        }                                                                                                                        \
    } while (0)
    for (int i = tid; i < 4 * (G::RP / 16); i += G::NTHR) {
        const int pl = i / (G::RP / 16), o = i - pl * (G::RP / 16);
        *reinterpret_cast<uint4*>(ldsb + pl * G::PLANE + o * 16) = make_uint4(0, 0, 0, 0);
    }
    uint4 sreg[13] = {};
    __amdgpu_buffer_rsrc_t srs = make_rsrc(v3, 0);
    // (the LDS address of a unit is recomputed at its store: a register held across the 40 taps is one the weight ring cannot have, and a spilled
    // one comes back through a scratch load that drains the whole VMEM queue)
#define W3_UNIT(k_, nrows_)                                                                                                      \
        int u_ = tid + (k_) * 256;                                                                                               \
        asm volatile("" : "+v"(u_));             /* not loop-invariant for the compiler: no hoisting, nothing to keep or spill */ \
        const int rp_ = u_ / 80, w_ = u_ - rp_ * 80;                                                                             \
        int row_ = rp_ >> 1;                                                                                                     \
        const int pc_ = rp_ & 1;                                                                                                 \
        row_ = row_ < (nrows_) ? row_ : (nrows_) - 1;
#define W3_L(k_, j_, cc_, qmin_, nrows_)                                                                                         \
    do {                                                                                                                         \
        W3_UNIT(k_, nrows_)                                                                                                      \
        srs = make_rsrc(v3 + (size_t)((DBG & 16) ? 0 : (qmin_)) * V3_ROWB, (uint32_t)(G::NR * V3_ROWB));                         \
        sreg[j_] = buf_load16(srs, row_ * V3_ROWB + pc_ * 1280 + w_ * 16, (cc_) * 2560);                                         \
    } while (0)
#define W3_S(k_, j_, nrows_, base_)                                                                                              \
    do {                                                                                                                         \
        W3_UNIT(k_, nrows_)                                                                                                      \
        *reinterpret_cast<uint4*>((base_) + (row_ + 1) * G::RP + pc_ * G::PLANE + w_ * 16) = sreg[j_];                           \
    } while (0)

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (uint32_t)(G::NCH * 40 * G::BV * 16));
    const int boff = (h * CO + n * 32 + j) * 16;
    const int co = n * 32 + j;
    const float bz = bias[co];
    int qmin, nrows;
    wino_pass_rows<G, S>(pass, total_tiles, qmin, nrows);
    if (!(DBG & 1)) {
        if constexpr (STG) {
#pragma unroll
            for (int k = 0; k < NDW; ++k) W3_DMA(k, W3_BASE(0, qmin), nrows, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int k = 0; k < 13; ++k) W3_L(k, k, 0, qmin, nrows);
#pragma unroll
            for (int k = 0; k < 13; ++k) W3_S(k, k, nrows, ldsb);
        }
    }
    __syncthreads();
    uint4 bq[8][2];
#pragma unroll
    for (int t = 0; t < BD; ++t) {
        bq[t][0] = buf_load16(wrs, boff, t * G::BV * 16);
        bq[t][1] = buf_load16(wrs, boff, t * G::BV * 16 + 2 * CO * 16);
    }
    int bufsel = 0;
    for (;;) {
        int aoff[TPW][5];
        const int T0 = pass * G::MB + mg * TPW * 32;
#pragma unroll
        for (int m = 0; m < TPW; ++m) {
            int T = T0 + m * 32 + j;
            if (T > total_tiles - 1) T = total_tiles - 1;
            const int gp = T / G::TPP, r2 = T - gp * G::TPP;
            const int tx = r2 >> 1, qo = 2 * gp + (r2 & 1), y = qo % S;
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int iy = y + ky - 2;
                aoff[m][ky] = ((iy >= 0 && iy < S) ? (qo + ky - 2 - qmin + 1) * G::RP : 0) + tx * 32 + h * 16;
            }
        }
        // (not zeroed: the first product of every accumulator -- taps 0..7 of the pass's first chunk -- takes the constant 0 as its C operand;
        // 256 v_accvgpr_write per pass and wave otherwise)
        f32x16 acc[TPW][8];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // The output transform A^T (4 outputs from the 8 positions) in three position groups; y3 takes position 7 last (it is the last to arrive:
        // position p is final after tap 32 + p of the last chunk).  Tile 0's first two groups are WRITTEN under taps 34-39 of the last chunk.
        // What the compiler makes of that: the arithmetic itself sinks back into the guarded store blocks behind the loop, but tile 0's 128
        // accumulator reads stay under the last taps, and with one wave per SIMD everything behind the last tap is exposed: 4.52 -> 4.32 ms.
        // (Pinning the arithmetic under the taps as well -- an empty asm on the partial sums -- needs 48-64 more live registers: spills, 7 ms.)
        f32x16 ya0, ya1, ya2, ya3;
        auto ep1 = [&](const int m, const int ra, const int rb, f32x16& y0, f32x16& y1, f32x16& y2, f32x16& y3) {
#pragma unroll
            for (int r = ra; r < rb; ++r) {
                const float e1 = acc[m][1][r] + acc[m][2][r], o1 = acc[m][1][r] - acc[m][2][r];
                y0[r] = acc[m][0][r] + e1; y1[r] = o1; y2[r] = e1; y3[r] = o1;
            }
        };
        auto ep2 = [&](const int m, const int ra, const int rb, f32x16& y0, f32x16& y1, f32x16& y2, f32x16& y3) {
#pragma unroll
            for (int r = ra; r < rb; ++r) {
                const float e2 = acc[m][3][r] + acc[m][4][r], o2 = acc[m][3][r] - acc[m][4][r];
                y0[r] += e2; y1[r] += 2.f * o2; y2[r] += 4.f * e2; y3[r] += 8.f * o2;
            }
        };
        auto ep3 = [&](const int m, f32x16& y0, f32x16& y1, f32x16& y2, f32x16& y3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e3 = acc[m][5][r] + acc[m][6][r], o3 = acc[m][5][r] - acc[m][6][r];
                y0[r] += e3; y1[r] += 0.5f * o3; y2[r] += 0.25f * e3; y3[r] += 0.125f * o3;
                y3[r] += acc[m][7][r];
            }
        };
        int next_pass = pass + 1;
        const bool draw = pass >= big_end || pass % PK == PK - 1;          // the last pass of a ticket draws the next one
        if (draw && tid == 0) s_next_pass = ticket_first((int)atomicAdd(pass_ctr, 1u) + (int)gridDim.x);   // read by everyone after the first chunk's barrier
        bool have_next = false;
        int qmin_n = qmin, nrows_n = nrows;
#pragma clang loop unroll(full)
        for (int cc = 0; cc < G::NCH; ++cc) {
            const bool last_c = cc == G::NCH - 1;
            if (last_c) {
                if (draw) next_pass = s_next_pass;
                have_next = next_pass < n_pass;
                if (have_next) wino_pass_rows<G, S>(next_pass, total_tiles, qmin_n, nrows_n);
            }
            const uint8_t* pbase = ldsb + bufsel * G::BUF;
            uint8_t* nbase = ldsb + (bufsel ^ 1) * G::BUF;
            // staged under this chunk: the next chunk of this pass, or the first chunk of the next pass (without one: this pass's first
            // chunk once more, into the buffer nobody reads again)
            const int scc = last_c ? 0 : cc + 1;
            const int sqmin = last_c ? qmin_n : qmin, snrows = last_c ? nrows_n : nrows;
            const int wc = cc * 40 * G::BV * 16, wn = scc * 40 * G::BV * 16;
            const unsigned long long sbase = STG ? W3_BASE(scc, sqmin) : 0ull;
            uint4 af[2][TPW][2];
#pragma unroll
            for (int m = 0; m < TPW; ++m) {
                af[0][m][0] = *reinterpret_cast<const uint4*>(pbase + aoff[m][0]);
                af[0][m][1] = *reinterpret_cast<const uint4*>(pbase + aoff[m][0] + G::PLANE);
            }
#pragma clang loop unroll(full)
            for (int t = 0; t < 40; ++t) {
                const int cur = t & 1, nxt = cur ^ 1;
                if (!(DBG & 8) && t + 1 < 40) {
                    const uint8_t* an = pbase + ((t + 1) % 8) * G::PS;
#pragma unroll
                    for (int m = 0; m < TPW; ++m) {
                        af[nxt][m][0] = *reinterpret_cast<const uint4*>(an + aoff[m][(t + 1) / 8]);
                        af[nxt][m][1] = *reinterpret_cast<const uint4*>(an + aoff[m][(t + 1) / 8] + G::PLANE);
                    }
                }
                if (!(DBG & 4)) {
                    const int wt = t + BD < 40 ? wc + (t + BD) * G::BV * 16 : wn + (t + BD - 40) * G::BV * 16;
                    bq[(t + BD) % 8][0] = buf_load16(wrs, boff, wt);
                    bq[(t + BD) % 8][1] = buf_load16(wrs, boff, wt + 2 * CO * 16);
                }
                if (!(DBG & 1)) {
                    if constexpr (STG) {
                        if (t >= DT0 && t < DT0 + NDW) W3_DMA(t - DT0, sbase, snrows, bufsel ^ 1);
                    } else {
                        // ONE batch of loads per chunk: every batch of HBM misses holds back the weight fragments queued behind it once
                        if (t == 0 && !(DBG & 128)) { _Pragma("unroll") for (int k = 0; k < 13; ++k) W3_L(k, k, scc, sqmin, snrows); }
                        if (t >= 16 && t < 29 && !(DBG & 64)) W3_S(t - 16, t - 16, snrows, nbase);
                    }
                }
                const int p = t % 8;
                const f16x8 b1 = __builtin_bit_cast(f16x8, bq[t % 8][0]);
                const f16x8 b2 = __builtin_bit_cast(f16x8, bq[t % 8][1]);
                f16x8 a1[TPW], a2[TPW];
#pragma unroll
                for (int m = 0; m < TPW; ++m) { a1[m] = __builtin_bit_cast(f16x8, af[cur][m][0]); a2[m] = __builtin_bit_cast(f16x8, af[cur][m][1]); }
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[m][p] = mfma16(a2[m], b1, (cc == 0 && t < 8) ? zero16 : acc[m][p]);
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[m][p] = mfma16(a1[m], b2, acc[m][p]);
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[m][p] = mfma16(a1[m], b1, acc[m][p]);
                if (cc == G::NCH - 1 && !(DBG & 2)) {
                    if (t == 34) ep1(0, 0, 5, ya0, ya1, ya2, ya3);
                    if (t == 35) ep1(0, 5, 11, ya0, ya1, ya2, ya3);
                    if (t == 36) ep1(0, 11, 16, ya0, ya1, ya2, ya3);
                    if (t == 37) ep2(0, 0, 5, ya0, ya1, ya2, ya3);
                    if (t == 38) ep2(0, 5, 11, ya0, ya1, ya2, ya3);
                    if (t == 39) ep2(0, 11, 16, ya0, ya1, ya2, ya3);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * TPW, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 10, 0);
#pragma unroll
                for (int g = 0; g < 3 * TPW; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x206, 8, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's part of the next chunk has landed: loads return in order, so everything older than the 2 x BD weight fragments in flight
            // (the next chunk's first taps) is complete -- the DMA instructions were issued before them.  (vmcnt(0) would drain those fragments too:
            // one L2 round trip per chunk)
            if constexpr (STG) { if constexpr (BD == 7 && !(DBG & 256)) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
            bufsel ^= 1;
        }
#pragma unroll
        for (int m = 0; m < ((DBG & 2) ? 0 : TPW); ++m) {
            f32x16 y0, y1, y2, y3;
            if (m == 0) { y0 = ya0; y1 = ya1; y2 = ya2; y3 = ya3; }
            else {
                ep1(m, 0, 16, y0, y1, y2, y3);
                __builtin_amdgcn_sched_barrier(0);
                ep2(m, 0, 16, y0, y1, y2, y3);
                __builtin_amdgcn_sched_barrier(0);
            }
            ep3(m, y0, y1, y2, y3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = 2 * rr;
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int T = T0 + m * 32 + i;
                const float v0 = fmaxf(fmaxf(y0[r], y1[r]), fmaxf(y0[r + 1], y1[r + 1]));
                const float v1 = fmaxf(fmaxf(y2[r], y3[r]), fmaxf(y2[r + 1], y3[r + 1]));
                if (T < total_tiles) {
                    const int gp = T / G::TPP, tx = (T - gp * G::TPP) >> 1;
                    float* o = out + ((size_t)gp * (S / 2) + 2 * tx) * CO + co;
                    __builtin_nontemporal_store(fmaxf(v0 * out_scale + bz, 0.f), o);
                    __builtin_nontemporal_store(fmaxf(v1 * out_scale + bz, 0.f), o + CO);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DBG & 2) {
#pragma unroll
            for (int m = 0; m < TPW; ++m)
#pragma unroll
                for (int p = 0; p < 8; ++p) asm volatile("" :: "a"(acc[m][p]));
        }
        if (!have_next) break;
        pass = next_pass; qmin = qmin_n; nrows = nrows_n;
    }
#undef W3_L
#undef W3_UNIT
#undef W3_S
#undef W3_DMA
#undef W3_BASE
}
