// cnn_fused12.h -- conv1 inside conv2 (round 4): the V2 operand image never travels through HBM.
// Included by cnn.hip inside namespace trexhip, after cnn_wpre.h (W2bGeom, w2b_rot, the helpers).
//
// k_conv2_wpre2 fetched the rows of V2 (conv1's output as the Winograd-domain fp16 operand image of conv2: 204800 B per crop, written by
// k_conv1_wpre at the HBM write rate, read back 1.16 x) into a ring of LDS row slots by LDS-DMA.  Here the workgroup PRODUCES the rows it
// is about to need straight into those slots from the u8 crop (6400 B per crop, L2-resident):
//   P0  the 6 crop rows under each new V2 row -> fp16, zero-padded, LDS (36 rows of 88 halves for a chunk of 6 V2 rows)
//   P1  conv1 on the matrix cores, k_conv1_wpre's tiles and products in its order (8 windows x 2 image rows per 16 x 16 x 32 tile, two
//       fp16 weight pieces, low pieces first), bias, ReLU, 2x2 max-pool -> fp32 [pooled pixel][16 channels] in LDS (the buffer of
//       conv2's own epilogue, free at this point)
//   P2  (row, conv2 tile, channel quad) items: 8 pooled pixels x 4 channels -> B^T d -> two fp16 pieces -> the rotated 16-byte units of
//       the four operand planes, exactly where the DMA put them
// behind the epilogue of the pass, under the other workgroup's tap loop (two workgroups per CU).  Same arithmetic, same rounding, same
// order as k_conv1_wpre + k_conv2_wpre2: V3 and everything downstream are bit-identical to the two-kernel chain (tests/test_cnn_gpu.py).
// HBM traffic of the layer pair: 5.4 + 11.3 GB -> 5.4 GB per 25600 crops (V3 out + the crops in).

struct W12Geom {
    using G = W2bGeom;
    static constexpr int CHUNK = 6;                                       // V2 rows produced at a time (what a pass inside a ticket needs)
    static constexpr int IMG_PITCH = 88, IMG_ROWS = CHUNK * 6;            // halves per padded crop row; 6 crop rows under each V2 row
    static constexpr int IMG_OFF = G::LDS_BYTES, IMG_BYTES = IMG_ROWS * IMG_PITCH * 2;
    static constexpr int RAW_OFF = IMG_OFF + IMG_BYTES, RAW_BYTES = 3072;   // the u8 crop rows of the next chunk, fetched by LDS-DMA under the epilogue
    static constexpr int LDS_BYTES = RAW_OFF + RAW_BYTES;
    static_assert(IMG_ROWS * 80 <= RAW_BYTES, "raw crop rows");
    static_assert(CHUNK * 40 * 16 * 4 <= G::PBUF, "the pooled activations of a chunk use conv2's epilogue buffer");
    static_assert(2 * (LDS_BYTES + 64) <= 160 * 1024, "two workgroups per CU");
};

template <int DBG = 0, int PRIO = 0x30, int STAGGER = 5, int AD = 1, int BD = 3>      // PRIO: s_setprio of (the vector phases, the tap loop) as hex digits      // DBG (dev builds): 1 no crop loads, 2 no conv1 MFMAs, 4 no P2 transform, 8 no production at all, 16 no epilogue, 32 no tap loop, 64 half the weight fragments (the second piece = a copy of the first: wrong results, same matrix work), 128 phase stamps
__global__ __launch_bounds__(256, 2) void k_conv12_wpre(const uint8_t* __restrict__ crops /*[N][80][80]*/, const uint4* __restrict__ w1tab /*[16][64]*/,
                                                        const float* __restrict__ bias1, const float inv_scale1,
                                                        const uint4* __restrict__ wp /*[5][8][2][2][64] x 16 B*/, const float* __restrict__ bias,
                                                        uint8_t* __restrict__ v3, const float out_scale, uint32_t* __restrict__ overflow,
                                                        const int n_crops, uint32_t* __restrict__ pass_ctr, const int PK /* consecutive passes per ticket: the first one produces 10 rows, the others 6 */,
                                                        uint8_t* __restrict__ crop_flags /* per-crop range flags (may be null) */,
                                                        unsigned long long* __restrict__ dbg_stamps = nullptr /* DBG & 128: cycles per phase of workgroups 0 and gridDim.x / 2 */) {
    using G = W2bGeom;
    using F = W12Geom;
    constexpr int CO = 64, S = 40;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsb[];
    __shared__ int s_next_pass;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int n = wave & 1, mg = wave >> 1;
    const int total_pairs = n_crops * (S / 2);
    const int total_rows = n_crops * S;
    const int n_pass = (total_pairs + G::RPP - 1) / G::RPP;
    int pass = blockIdx.x * PK;
    if (pass >= n_pass) return;
    __builtin_amdgcn_s_setprio((PRIO >> 4) & 0xf);
    for (int i = tid; i < 4 * (G::ROWL / 16); i += 256) {                // the zero rows of the four planes
        const int pl = i / (G::ROWL / 16), o = i - pl * (G::ROWL / 16);
        *reinterpret_cast<uint4*>(ldsb + pl * G::PLANE + o * 16) = make_uint4(0, 0, 0, 0);
    }
    _Float16* img = reinterpret_cast<_Float16*>(ldsb + F::IMG_OFF);
    float* pbuf = reinterpret_cast<float*>(ldsb + G::PBUF_OFF);
    for (int i = tid; i < F::IMG_BYTES / 16; i += 256) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);   // the x padding stays zero
    bool ovf = false;
    // the range guard is per crop: what was raised since the last call belongs to units [u_lo, u_hi] (V2 rows: 40 per crop, pooled rows: 20 per
    // crop), at most two crops; flagging a neighbour too only means that it is re-run in the wider arithmetic as well
    auto flag_crops = [&](const int u_lo, const int u_hi, const int per_crop) {
        if (__any(ovf)) {
            if (lane == 0) { if (crop_flags) { crop_flags[u_lo / per_crop] = 1; crop_flags[u_hi / per_crop] = 1; } atomicOr(overflow, 1u); }
            ovf = false;
        }
    };
    // DBG & 128 (dev builds): thread 0 of two workgroups sums the cycles between the phase boundaries of its passes (tools/f12_stamps.py)
    unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
    bool st_on = false;                                                  // switched on behind the prologue
#define F12_STAMP(i_) do { if ((DBG & 128) && st_on) { const unsigned long long t_ = __builtin_readcyclecounter(); st_sum[i_] += t_ - st_last; st_last = t_; } } while (0)
#define W2B_ROWS(pass_, qmin_, nrows_)                                                                                           \
    do {                                                                                                                         \
        const int gp0_ = (pass_) * G::RPP;                                                                                       \
        int gpl_ = gp0_ + G::RPP - 1;                                                                                            \
        gpl_ = gpl_ < total_pairs ? gpl_ : total_pairs - 1;                                                                      \
        const int y0_ = (2 * gp0_) % S, yl_ = (2 * gpl_) % S + 1;                                                                \
        qmin_ = 2 * gp0_ - (y0_ >= 2 ? 2 : 0);                                                                                   \
        nrows_ = 2 * gpl_ + 1 + (yl_ + 2 <= S - 1 ? 2 : 0) - qmin_ + 1;                                                          \
    } while (0)

    // V2 rows [lo, hi) of the batch (q = crop * 40 + y) -> their ring slots (q % NR + 1) in the four operand planes.  Every barrier inside
    // is workgroup-uniform (lo, hi are).  On entry nobody reads the slots being replaced, pbuf or img any more; on exit the rows are complete.
    // the crop-row unit (16 pixels) item `it` of a chunk that starts at V2 row c0 stands for: item = (V2 row v, crop row k of its six, unit u)
#define W12_ITEM(it_, c0_)                                                                                                       \
                const int vk = (it_) / 5, u = (it_) - vk * 5;                                                                    \
                const int v = vk / 6, k = vk - v * 6;                                                                            \
                int q = (c0_) + v;                                                                                               \
                q = q < total_rows ? q : total_rows - 1;                                                                         \
                const int crop = q / S, y = q - crop * S;                                                                        \
                const int iy = 2 * y - 2 + k;
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(ldsb + F::RAW_OFF);
    const unsigned long long crops_u = wave_uniform64(reinterpret_cast<unsigned long long>(crops));
    // the first chunk of rows [lo, hi): its crop rows HBM / L2 -> raw by LDS-DMA (no registers; issued at the start of the epilogue, landed by
    // the time the production phase starts).  Waves 0..2, one instruction each; rows outside the crop fetch row 0 (P0 writes zeros for them)
    auto prefetch = [&](const int lo, const int hi) {
        const int nr = hi - lo < F::CHUNK ? hi - lo : F::CHUNK;
        const int it = wave * 64 + lane;
        if (wave < 3 && it < nr * 30 && !(DBG & (1 | 8))) {
            W12_ITEM(it, lo)
            const int iyc = iy < 0 ? 0 : (iy > 79 ? 79 : iy);
            wpre_dma16(crops_u, (uint32_t)((crop * 80 + iyc) * 80 + u * 16), raw_lds + (uint32_t)(wave * 1024));
        }
    };
    // Production of the V2 rows [c0, c0 + nr) (nr <= CHUNK) in three phases with a barrier between them:
    //   p0  their crop rows (from raw when the LDS-DMA prefetch fetched them, else straight from the crop) -> fp16 -> img; also issues the
    //       loads of conv1's four base weight fragments (shift 0: kernel rows 0..3 | row 4, piece hi | lo), used by p1
    //   p1  conv1 tiles on the matrix cores -> pooled activations in pbuf
    //   p2  B^T d, fp16 pieces -> the ring slots (q % NR + 1) of the four operand planes
    auto p0 = [&](const int c0, const int nr, const bool from_raw, uint4 (&bf0)[4]) {
#pragma unroll
        for (int f = 0; f < 4; ++f) bf0[f] = w1tab[f * 64 + lane];
        {
            // P0: crop rows -> img: one 16-byte unit (from raw, or straight from the crop), 16 halves
            for (int it = tid; it < nr * 30; it += 256) {
                W12_ITEM(it, c0)
                uint4 px = make_uint4(0, 0, 0, 0);
                if (!(DBG & 1) && iy >= 0 && iy < 80) {
                    if (from_raw) px = *reinterpret_cast<const uint4*>(ldsb + F::RAW_OFF + it * 16);
                    else px = *reinterpret_cast<const uint4*>(crops + ((size_t)crop * 80 + iy) * 80 + u * 16);
                }
                const uint32_t w4[4] = {px.x, px.y, px.z, px.w};
                uint32_t* d = reinterpret_cast<uint32_t*>(img + vk * F::IMG_PITCH + 2 + u * 16);       // (4-byte aligned: 2 halves of left padding)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t b0 = (w4[e >> 1] >> (16 * (e & 1))) & 0xffu, b1 = (w4[e >> 1] >> (16 * (e & 1) + 8)) & 0xffu;
                    d[e] = pack_h2((_Float16)(float)b0, (_Float16)(float)b1);
                }
            }
        }
    };
    auto p1 = [&](const int nr, const uint4 (&bf0)[4]) {
        {
            // P1: conv1 tiles (8 windows of 4 outputs x 2 image rows; 20 windows per V2 row), wave w takes tiles w, w + 4, ...
            {
                const int r = lane & 15, q4 = lane >> 4;
                const float bz = bias1[r];
                // fragment (s, mf, piece) = the base fragment (mf, piece) moved up by s window slots (16 s bits; the slots it leaves are zero
                // weights, the three it pushes out were zero): [shift s][mfma 0: ky 0..3 | 1: ky 4][piece hi | lo] as the host table has them
                uint4 bf[16];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const uint4 b = bf0[f];
                    bf[f] = b;
                    bf[4 + f] = make_uint4(b.x << 16, __builtin_amdgcn_alignbit(b.y, b.x, 16), __builtin_amdgcn_alignbit(b.z, b.y, 16), __builtin_amdgcn_alignbit(b.w, b.z, 16));
                    bf[8 + f] = make_uint4(0u, b.x, b.y, b.z);
                    bf[12 + f] = make_uint4(0u, b.x << 16, __builtin_amdgcn_alignbit(b.y, b.x, 16), __builtin_amdgcn_alignbit(b.z, b.y, 16));
                }
                const int n_win = nr * 20, n_tiles = (n_win + 7) >> 3;
                for (int tile = wave; tile < n_tiles; tile += 4) {
                    int wdx = tile * 8 + (r >> 1);
                    wdx = wdx < n_win ? wdx : n_win - 1;
                    const int v = wdx / 20, x4 = (wdx - v * 20) * 4;
                    const int row = v * 6 + (r & 1);
                    const _Float16* p1 = img + (row + q4) * F::IMG_PITCH + x4;
                    const _Float16* p2 = img + (row + 4) * F::IMG_PITCH + x4;
                    uint4 a1u, a2u;
                    { const uint2 l2 = *reinterpret_cast<const uint2*>(p1), h2 = *reinterpret_cast<const uint2*>(p1 + 4); a1u = make_uint4(l2.x, l2.y, h2.x, h2.y); }
                    { const uint2 l2 = *reinterpret_cast<const uint2*>(p2), h2 = *reinterpret_cast<const uint2*>(p2 + 4); a2u = make_uint4(l2.x, l2.y, h2.x, h2.y); }
                    const f16x8_c1 a1 = __builtin_bit_cast(f16x8_c1, a1u), a2 = __builtin_bit_cast(f16x8_c1, a2u);
                    f32x4 acc[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
                        if (DBG & 2) { acc[s] = c; continue; }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 1]), c, 0, 0, 0);   // low pieces first
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 3]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8_c1, bf[s * 4 + 2]), c, 0, 0, 0);
                        acc[s] = c;
                    }
                    // lane (co = r, q4): accumulator rows 4 q4 .. 4 q4 + 3 = windows 2 q4, 2 q4 + 1 x the two image rows: pooled pixels
                    // 2 (window) and 2 (window) + 1 of the V2 row, channel co.  pbuf [slot of the pooled pixel][16 channels]; slot = the pixel
                    // index with its two low bit pairs swapped, so that the four q4 groups of a store (pixels 4 apart) fill 256 contiguous bytes
                    // and P2's lanes (tiles 4 pixels apart, channel quads) read contiguously as well
#pragma unroll
                    for (int pos = 0; pos < 2; ++pos) {
                        const int w = tile * 8 + 2 * q4 + pos;
                        const float m0 = fmaxf(fmaxf(acc[0][2 * pos], acc[0][2 * pos + 1]), fmaxf(acc[1][2 * pos], acc[1][2 * pos + 1]));
                        const float m1 = fmaxf(fmaxf(acc[2][2 * pos], acc[2][2 * pos + 1]), fmaxf(acc[3][2 * pos], acc[3][2 * pos + 1]));
                        const float v0 = fmaxf(m0 * inv_scale1 + bz, 0.f), v1 = fmaxf(m1 * inv_scale1 + bz, 0.f);
                        if (w < n_win) {
                            ovf |= !(v0 < 4368.0f) | !(v1 < 4368.0f);
                            const int px = 2 * w;                                    // = v * 40 + x: 20 windows of 2 pooled pixels per row
                            float* o = pbuf + ((px & ~15) | ((px & 3) << 2) | ((px >> 2) & 3)) * 16 + r;
                            o[0] = v0;
                            o[64] = v1;                                              // px + 1: bit 0 of the pixel is bit 2 of the slot
                        }
                    }
                }
            }
        }
    };
    auto p2 = [&](const int c0, const int nr) {
        {
            // P2: (V2 row v, conv2 tile tx, channel quad): 8 pooled pixels x 4 channels -> B^T d -> pieces -> the planes
            if (!(DBG & 4) && tid < nr * 40) {
                const int v = tid / 40, rem = tid - v * 40, tx = rem >> 2, quad = rem & 3;
                const int q = c0 + v;
                const int slot = q % G::NR + 1, rot = w2b_rot(slot);
                float4 d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = 4 * tx - 2 + k;
                    d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (x >= 0 && x < 40) {
                        const int px = v * 40 + x;
                        d[k] = *reinterpret_cast<const float4*>(pbuf + ((px & ~15) | ((px & 3) << 2) | ((px >> 2) & 3)) * 16 + quad * 4);
                    }
                }
                float ua[8], ub[8], uc[8], ud[8];
                wino_bt(d[0].x, d[1].x, d[2].x, d[3].x, d[4].x, d[5].x, d[6].x, d[7].x, ua);
                wino_bt(d[0].y, d[1].y, d[2].y, d[3].y, d[4].y, d[5].y, d[6].y, d[7].y, ub);
                wino_bt(d[0].z, d[1].z, d[2].z, d[3].z, d[4].z, d[5].z, d[6].z, d[7].z, uc);
                wino_bt(d[0].w, d[1].w, d[2].w, d[3].w, d[4].w, d[5].w, d[6].w, d[7].w, ud);
                uint8_t* rowb = ldsb + slot * G::ROWL + (quad & 1) * 8;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int pi = p < 3 ? p : (p == 7 ? 3 : p + 1);             // stored order 0,1,2,7 | 3,4,5,6: group = pi >> 2, place pi & 3
                    const int w = (pi & 3) * 20 + tx * 2 + (quad >> 1);
                    uint8_t* dst = rowb + (pi >> 2) * G::BUF + ((w & ~15) | ((w + rot) & 15)) * 16;
                    uint32_t l0, l1, m0, m1;
                    split2h_pair(ua[p], ub[p], l0, m0);
                    split2h_pair(uc[p], ud[p], l1, m1);
                    *reinterpret_cast<uint2*>(dst) = make_uint2(l0, l1);
                    *reinterpret_cast<uint2*>(dst + G::PLANE) = make_uint2(m0, m1);
                }
            }
        }
    };
    // rows [lo, hi) chunk by chunk; the first chunk's p0 has already run when first_done.  Ends with a barrier: the rows are complete
    auto produce = [&](const int lo, const int hi, const bool first_done, uint4 (&bf0)[4]) {
        for (int c0 = (DBG & 8) ? hi : lo; c0 < hi; c0 += F::CHUNK) {
            const int nr = hi - c0 < F::CHUNK ? hi - c0 : F::CHUNK;
            if (!(first_done && c0 == lo)) p0(c0, nr, false, bf0);
            F12_STAMP(5);
            __syncthreads();
            F12_STAMP(6);
            p1(nr, bf0);
            flag_crops(c0 < total_rows ? c0 : total_rows - 1, c0 + nr - 1 < total_rows ? c0 + nr - 1 : total_rows - 1, S);
            F12_STAMP(7);
            __syncthreads();
            F12_STAMP(8);
            p2(c0, nr);
            F12_STAMP(9);
            __syncthreads();
            F12_STAMP(10);
        }
    };

    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (uint32_t)(40 * G::BV * 16));
    const int boff = (h * CO + n * 32 + j) * 16;
    const int co = n * 32 + j;
    const float bz = bias[co];
    int qmin, nrows;
    W2B_ROWS(pass, qmin, nrows);
    __syncthreads();
    {
        uint4 bf0[4];
        produce(qmin, qmin + nrows, false, bf0);
    }
    int res_hi = qmin + nrows;                                           // rows [this pass's qmin, res_hi) are resident
    if (tid == 0) s_next_pass = ((int)atomicAdd(pass_ctr, 1u) + (int)gridDim.x) * PK;
    __syncthreads();
#define W2_POS(tau_) (((tau_) / 20) == 0 ? ((tau_) % 4 == 3 ? 7 : (tau_) % 4) : 3 + (tau_) % 4)
#define W2_BOFF(tau_) (((((tau_) % 20) / 4) * 8 + W2_POS(tau_)) * G::BV * 16)
    // the two workgroups of a CU start together and would keep step: the second half of the grid starts half a pass late
    if (STAGGER > 0 && blockIdx.x >= gridDim.x / 2) {
#pragma unroll 1
        for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if ((DBG & 128) && dbg_stamps && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) { st_on = true; st_last = __builtin_readcyclecounter(); }
    for (;;) {
        uint4 bq[8][2];
#pragma unroll
        for (int t = 0; t < BD; ++t) { bq[t][0] = buf_load16(wrs, boff, W2_BOFF(t)); bq[t][1] = (DBG & 64) ? bq[t][0] : buf_load16(wrs, boff, W2_BOFF(t) + 2 * CO * 16); }
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
        // The VECTOR phases outrank the tap loop (round 5, tools/ubench_simd.hip): a wave whose next instruction is an MFMA that waits for its
        // accumulator holds the SIMD's issue port, and at equal (or higher) priority the other workgroup's wave in its vector phase gets next
        // to nothing (960 v_fma beside 120 dependent MFMAs: 8486 cycles at priorities 0 / 0 or 3 / 0, 5444 with the vector wave above -- the
        // MFMA wave loses 1 %).  Rounds 3-4 had it the other way round (tap loop 3, everything else 0): the phases added up instead of overlapping.
        __builtin_amdgcn_s_setprio(PRIO & 0xf);
        uint4 af[AD + 1][2];
#define W2B_AREAD(dst_, tau_)                                                                                                    \
        do {                                                                                                                     \
            const uint8_t* an_ = ldsb + ((tau_) / 20) * G::BUF + aoff[((tau_) % 20) / 4][(tau_) % 4];                            \
            dst_[0] = *reinterpret_cast<const uint4*>(an_);                                                                      \
            dst_[1] = *reinterpret_cast<const uint4*>(an_ + G::PLANE);                                                           \
        } while (0)
#pragma unroll
        for (int t = 0; t < AD; ++t) W2B_AREAD(af[t], t);
#pragma clang loop unroll(full)
        for (int tau = 0; tau < ((DBG & 32) ? 0 : 40); ++tau) {
            const int tl = tau % 20;
            if (tau + AD < 40) W2B_AREAD(af[(tau + AD) % (AD + 1)], tau + AD);
            if (tau + BD < 40) {
                const int wt = W2_BOFF(tau + BD);
                bq[(tau + BD) % 8][0] = buf_load16(wrs, boff, wt);
                bq[(tau + BD) % 8][1] = (DBG & 64) ? bq[(tau + BD) % 8][0] : buf_load16(wrs, boff, wt + 2 * CO * 16);
            }
            const int p = W2_POS(tau);
            const f16x8 b1 = __builtin_bit_cast(f16x8, bq[tau % 8][0]), b2 = __builtin_bit_cast(f16x8, bq[tau % 8][1]);
            const f16x8 a1 = __builtin_bit_cast(f16x8, af[tau % (AD + 1)][0]), a2 = __builtin_bit_cast(f16x8, af[tau % (AD + 1)][1]);
            acc[p] = mfma16(a2, b1, tl < 4 ? zero16 : acc[p]);            // kernel row 0 starts the accumulator
            acc[p] = mfma16(a1, b2, acc[p]);
            acc[p] = mfma16(a1, b1, acc[p]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef W2B_AREAD
        __builtin_amdgcn_s_setprio((PRIO >> 4) & 0xf);
        F12_STAMP(0);
        __syncthreads();                                                  // every wave is done with the operand planes
        F12_STAMP(1);
        const bool draw = pass % PK == PK - 1;                            // the last pass of a ticket moves on to the next ticket
        const int next_pass = draw ? s_next_pass : pass + 1;
        const bool have_next = next_pass < n_pass;
        int qmin_n = qmin, nrows_n = nrows;
        uint32_t ticket = 0;
        if (draw && tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(pass_ctr), "v"(1u) : "memory");
        int lo_new = 0;
        if (have_next) {
            W2B_ROWS(next_pass, qmin_n, nrows_n);
            // the next pass of the same ticket: rows below res_hi are already there (its qmin is not below this pass's)
            lo_new = (next_pass == pass + 1 && res_hi > qmin_n) ? res_hi : qmin_n;
            prefetch(lo_new, qmin_n + nrows_n);
        }
        if (DBG & 32) { _Pragma("unroll") for (int p = 0; p < 8; ++p) acc[p] = zero16; }
        // epilogue 1: Y = A^T M, pool, bias, ReLU -> the pass's 3 x 20 x 64 activations as fp32 in LDS
        if (!(DBG & 16)) {
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
        { const int g0 = pass * G::RPP, g1 = g0 + G::RPP - 1; flag_crops(g0, g1 < total_pairs ? g1 : total_pairs - 1, S / 2); }
        F12_STAMP(2);
        __syncthreads();                                                  // the activations are in pbuf
        F12_STAMP(3);
        uint4 bf0[4];
        const int hi_new = qmin_n + nrows_n;
        // epilogue 2: (pooled row, conv3 tile, channel quad) items -> V3
        if (!(DBG & 16) && tid < 240) {
            const int rp = tid / 80, rem = tid - rp * 80, tx = rem >> 4, quad = rem & 15;
            const int gp = pass * G::RPP + rp;                            // = q3: pooled row of the batch
            if (gp < total_pairs) {
                float4 d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = 4 * tx - 2 + k;
                    d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (x >= 0 && x < 20) d[k] = *reinterpret_cast<const float4*>(pbuf + (rp * 20 + x) * 64 + quad * 4);
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
        F12_STAMP(4);
        if (!have_next) break;
        // the first chunk of the next pass's rows starts here: every lane converts the crop-row unit its OWN LDS-DMA fetched (item = thread
        // index in prefetch and in p0 alike), so no barrier stands between the fetch and the conversion -- only the wave's own counter: the 16
        // V3 stores of epilogue 2 were issued behind the DMA and may stay in flight
        asm volatile("s_waitcnt vmcnt(16)" : "+v"(ticket) :: "memory");
        if (!(DBG & 8)) p0(lo_new, hi_new - lo_new < F::CHUNK ? hi_new - lo_new : F::CHUNK, true, bf0);
        produce(lo_new, hi_new, true, bf0);                               // (starts with the barrier behind epilogue 2 / p0, ends with one: the planes are complete)
        res_hi = qmin_n + nrows_n;
        if (draw) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");
            if (tid == 0) s_next_pass = ((int)ticket + (int)gridDim.x) * PK;       // read behind a later pass's first barrier
        }
        pass = next_pass; qmin = qmin_n; nrows = nrows_n;
        if ((DBG & 128) && st_on) st_sum[11] += 1;
    }
    if ((DBG & 128) && st_on) { _Pragma("unroll") for (int i = 0; i < 12; ++i) dbg_stamps[(blockIdx.x ? 12 : 0) + i] = st_sum[i]; }
#undef F12_STAMP
#undef W2B_ROWS
#undef W12_ITEM
#undef W2_POS
#undef W2_BOFF
    if (__any(ovf) && lane == 0) atomicOr(overflow, 3u);      // (nothing is left here behind the last flag_crops; if it ever is, it is unattributed: every crop)
}
