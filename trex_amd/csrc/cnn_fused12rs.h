// cnn_fused12rs.h -- conv1 inside conv2 with the work split by ROLE between the waves of one workgroup (round 5).
// Included by cnn.hip inside namespace trexhip, after cnn_fused12.h (W2bGeom, w2b_rot, wino_bt, split2h_pair, the helpers).
//
// What round 5 measured about k_conv12_wpre (profiles/r05_f12_stamps.txt, profiles/r05_ubench_simd.txt, profiles/r05_ubench_blocks.txt):
//   * a wave issues one instruction per ~5 cycles whatever its kind (256 independent v_add / v_fma: 5.2-5.4 cycles each), and a pass of that kernel
//     is ~2500 instructions per wave, one dependent chain tap loop -> output transform -> V3 transform -> crop rows -> conv1 -> V2 transform;
//   * ONE workgroup alone on a CU needs 22.9 k cycles per pass, two per CU 29.5 k each: the kernel's time is the length of that chain, not
//     the occupancy of any pipe (matrix pipe 0.41, vector issue about 0.5).
// So the chain is cut in two and the halves run side by side on every SIMD:
//   consumer waves 0..3   tap loop (conv2's 40 position GEMMs on the matrix cores) -> output transform, pool, bias, ReLU -> pbufE
//   producer waves 4..7   V3 transform of the PREVIOUS pass (pbufE -> B^T d -> fp16 pieces -> HBM), then the V2 rows of the NEXT pass:
//                         crop rows -> fp16 (P0), conv1 on the matrix cores + pool (P1), B^T d + pieces -> the LDS row ring (P2)
// One workgroup of 8 waves per CU (a consumer and a producer wave on each SIMD), 256 registers each; every wave executes the same
// sequence of s_barrier per round (three: after taps 0-19 | P0, after taps 20-39 | P1, after the output transform | P2), so the roles cannot
// drift apart.  The arithmetic of every phase is k_conv12_wpre's, instruction for instruction: V3 and the probabilities stay bit-identical to
// the two-kernel chain (tests/test_cnn_gpu.py::test_fused_equals_two_kernel_chain).
//
// Measured and not kept (round 5, profiles/r05_f12_two_ahead.txt): the producer TWO passes ahead -- V2 transform into registers beside taps 20-39,
// conv1 beside the output transform, where the matrix pipe idles; contiguous passes per workgroup instead of tickets.  Bit-identical, and taps
// 20-39 do fall from 3660 to 2580 cycles without conv1's MFMAs beside them, but conv1 + pool is one wave's serial chain of ~350 instructions and
// 64 MFMAs (~3900 cycles wherever it runs): behind the 1770-cycle output transform it is exposed, the round grows from 10.3 k to 11.7 k cycles
// (3.63 -> 3.68 ms).  Both roles are single instruction streams of ~8300-8500 cycles per round; the stages only decide how they line up.
//
// LDS: the four operand planes (ring of NR = 10 row slots + the zero row, as before: P2 writes the ring in the stage where no tap reads it),
// pbufE (consumer -> producer), pbufP (conv1's pooled activations of a chunk), the padded fp16 crop rows of a chunk: 93 KB.

// max of three / two floats as ONE instruction each (the compiler's fmaxf puts a canonicalising v_max v, v, v in front of operands it does not
// know to be canonical); exact, so results do not change.  ONLY for operands written by ordinary VALU instructions: an MFMA result read by
// inline asm is not padded by the hazard recognizer (MI355X guide 5.7) -- round 5 learnt that the hard way
__device__ __forceinline__ float rs_max3(const float a, const float b, const float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float rs_max3z(const float a, const float b) { float r; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }      // max(a, b, 0)
__device__ __forceinline__ float rs_max(const float a, const float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Order of the 20 taps of a stage (ORD 1, round 5): position PAIR major -- i = tap % 20: pair i / 10, kernel row (i % 10) / 2, position of the pair
// i % 2 -- with stage 1 = positions (1,2) then (0,7), stage 2 = (3,4) then (5,6): the pairs the output transform A^T combines are final 10 taps
// before their stage ends, and their part of A^T runs under the taps that follow (the tap loop of stage 2 is bound by the matrix pipe, the
// output transform behind it by the vector issue of two waves).  Per accumulator the order of the sums is unchanged (kernel row 0..4).
// ORD 0: kernel row major (i / 4, i % 4), the order of k_conv12_wpre and k_conv2_wpre2.  pg = the position's place in its stored group
// (group 0 holds positions 0,1,2,7; group 1 holds 3,4,5,6).
__host__ __device__ constexpr int rs_ky(const int ord, const int tau) { return ord ? ((tau % 20) % 10) / 2 : (tau % 20) / 4; }
__host__ __device__ constexpr int rs_pg(const int ord, const int tau) {
    const int i = tau % 20;
    if (!ord) return i % 4;
    return tau / 20 == 0 ? (i / 10 == 0 ? 1 + i % 2 : (i % 2 ? 3 : 0)) : 2 * (i / 10) + i % 2;
}
__host__ __device__ constexpr int rs_pos(const int ord, const int tau) { return tau / 20 == 0 ? (rs_pg(ord, tau) == 3 ? 7 : rs_pg(ord, tau)) : 3 + rs_pg(ord, tau); }

struct W12RGeom {
    using G = W2bGeom;
    static constexpr int CHUNK = 6;
    static constexpr int IMG_PITCH = 88, IMG_ROWS = CHUNK * 6;
    // pbufE: the pass's 3 x 20 x 64 pooled activations of conv2 (fp32), rows padded by two zero pixels on either side: the V3 transform's 8-pixel
    // windows read their halo as plain loads (no clamps, no selects); pbufP: conv1's pooled activations of a chunk, rows of 40 pixels in a pitch of
    // 48 (4 zero pixels left and right), 16 channels
    static constexpr int PEP = 24, PPP = 48;
    static constexpr int PBE_OFF = G::PBUF_OFF, PBE_BYTES = G::RPP * PEP * 64 * 4;
    static constexpr int PBP_OFF = PBE_OFF + PBE_BYTES, PBP_BYTES = CHUNK * PPP * 16 * 4;
    static constexpr int IMG_OFF = PBP_OFF + PBP_BYTES, IMG_BYTES = IMG_ROWS * IMG_PITCH * 2;
    static constexpr int CTL_OFF = IMG_OFF + IMG_BYTES;                 // the next ticket's first pass
    // the weight fragments of NRES of the 40 taps stay in LDS for the whole kernel (4 KB per tap: 2 pieces x 2 k-octets x 64 output channels x
    // 16 B): the consumer waves' weight stream asks for 84 B/clk of the CU's 64 B/clk vector-memory path, and the producers' V3 stores ride on it too
    static constexpr int WRES_OFF = CTL_OFF + 16, WTAP = 4096;
    static constexpr int NRES = (160 * 1024 - WRES_OFF) / WTAP;
    static constexpr int LDS_BYTES = WRES_OFF + NRES * WTAP;
    static_assert(LDS_BYTES <= 160 * 1024 && CTL_OFF % 16 == 0 && NRES >= 8 && NRES <= 40, "one workgroup per CU");
    // the FIRST taps of a pass are the resident ones: they run beside the producers' V3 stores and crop loads (stage 1), the later taps beside
    // conv1, which touches no vector memory (stage 2) -- spread evenly over the loop the same 15 taps bought 10 % of stage 1, not 40
    static constexpr bool resident(const int tau, const int pat = 0) { return pat == 0 ? tau < NRES : (pat == 1 ? (tau < 10 || (tau >= 20 && tau < 25)) : (tau < 7 || (tau >= 20 && tau < 28))); }
    static constexpr int res_slot(const int tau, const int pat = 0) { return pat == 0 ? tau : (pat == 1 ? (tau < 10 ? tau : tau - 10) : (tau < 7 ? tau : tau - 13)); }
};

template <int DBG = 0, int BD = 3, int TSPLIT = 20, int PRIO = 0x201, int PAIR = 0, int RESPAT = 0, int ORD = 1>      // PRIO 0x201: since the taps run in pair order the producers' V2 transform is the longer half of stage 3 and outranks the output transform (0x202, equal priorities: 3.65 ms, 0x201 / 0x200 / 0x301 / 0x311: 3.58-3.60); PAIR: two taps at a time, consecutive MFMAs on different accumulators; TSPLIT: taps in front of the first barrier of a round; PRIO: s_setprio of (producer, tap loop, output transform) as hex digits
__global__ __launch_bounds__(512) void k_conv12_rs(const uint8_t* __restrict__ crops /*[N][80][80]*/, const uint4* __restrict__ w1tab /*[16][64]*/,
                                                   const float* __restrict__ bias1, const float inv_scale1,
                                                   const uint4* __restrict__ wp /*[5][8][2][2][64] x 16 B*/, const float* __restrict__ bias,
                                                   uint8_t* __restrict__ v3, const float out_scale, uint32_t* __restrict__ overflow,
                                                   const int n_crops, uint32_t* __restrict__ pass_ctr, const int PK /* consecutive passes per ticket */,
                                                   uint8_t* __restrict__ crop_flags /* per-crop range flags (may be null) */,
                                                   unsigned long long* __restrict__ dbg_stamps = nullptr /* DBG & 128 (dev): cycles per stage of waves 0 and 4 of workgroup 0 */) {
    using G = W2bGeom;
    using F = W12RGeom;
    constexpr int CO = 64, S = 40;
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;                                     // wave-uniform
    const int rt = tid & 255, rw = wave & 3;                             // thread / wave index within the role
    const int total_pairs = n_crops * (S / 2);
    const int total_rows = n_crops * S;
    const int n_pass = (total_pairs + G::RPP - 1) / G::RPP;
    int pass = blockIdx.x * PK;
    if (pass >= n_pass) return;
    // the next ticket's first pass: written by thread 0, read by every wave behind a barrier (a plain LDS word: through a volatile pointer into
    // the dynamic array it became a FLAT load with a vmcnt(0) behind it, and everything computed from it vector arithmetic)
    int* s_next = reinterpret_cast<int*>(ldsb + F::CTL_OFF);
    int tleft = PK;                                                      // passes left in this ticket, this one included (tickets are PK consecutive passes)
    for (int i = tid; i < 4 * (G::ROWL / 16); i += 512) {                // the zero rows of the four planes
        const int pl = i / (G::ROWL / 16), o = i - pl * (G::ROWL / 16);
        *reinterpret_cast<uint4*>(ldsb + pl * G::PLANE + o * 16) = make_uint4(0, 0, 0, 0);
    }
    _Float16* img = reinterpret_cast<_Float16*>(ldsb + F::IMG_OFF);
    float* pbe = reinterpret_cast<float*>(ldsb + F::PBE_OFF);
    float* pbp = reinterpret_cast<float*>(ldsb + F::PBP_OFF);
    for (int i = tid; i < F::IMG_BYTES / 16; i += 512) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);   // the x padding stays zero
    for (int i = tid; i < (F::PBE_BYTES + F::PBP_BYTES) / 16; i += 512) reinterpret_cast<uint4*>(ldsb + F::PBE_OFF)[i] = make_uint4(0, 0, 0, 0);      // the halo pixels stay zero
    // resident weight fragments: tap tau -> slot res_slot(tau); a slot holds the tap's [piece][k-octet h][co] x 16 B exactly as the weight image does
    for (int i = tid; i < 40 * (F::WTAP / 16); i += 512) {
        const int tau = i / (F::WTAP / 16), u = i - tau * (F::WTAP / 16);
        const int slot = F::res_slot(tau, RESPAT);
        if (F::resident(tau, RESPAT)) reinterpret_cast<uint4*>(ldsb + F::WRES_OFF + slot * F::WTAP)[u] = wp[(size_t)(rs_ky(ORD, tau) * 8 + rs_pos(ORD, tau)) * G::BV + u];
    }
    if (tid == 0) *s_next = ((int)atomicAdd(pass_ctr, 1u) + (int)gridDim.x) * PK;
    float ovfm = 0.f;                                                    // the largest activation seen (all are >= 0 behind their ReLU): the fp16 range guard
    // the range guard is per crop: what was raised since the last call belongs to units [u_lo, u_hi] (V2 rows: 40 per crop, pooled rows: 20 per crop)
    auto flag_crops = [&](const int u_lo, const int u_hi, const int per_crop) {
        if (__any(!(ovfm < 4368.0f))) {
            if (lane == 0) { if (crop_flags) { crop_flags[u_lo / per_crop] = 1; crop_flags[u_hi / per_crop] = 1; } atomicOr(overflow, 1u); }
            ovfm = 0.f;
        }
    };
    // DBG & 128 (dev builds): lane 0 of waves 0 (consumer) and 4 (producer) of workgroup 0 sum the cycles between the stage boundaries
    unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
    const bool st_on = (DBG & 128) && dbg_stamps && blockIdx.x == 0 && (tid == 0 || tid == 256);
#define RS_STAMP(i_) do { if ((DBG & 128) && st_on) { const unsigned long long t_ = __builtin_readcyclecounter(); st_sum[i_] += t_ - st_last; st_last = t_; } } while (0)
    // every wave: LDS writes done, then the workgroup barrier.  Both roles execute the SAME number of these per round.
#define RS_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RS_ROWS(pass_, qmin_, nrows_)                                                                                            \
    do {                                                                                                                         \
        const int gp0_ = (pass_) * G::RPP;                                                                                       \
        int gpl_ = gp0_ + G::RPP - 1;                                                                                            \
        gpl_ = gpl_ < total_pairs ? gpl_ : total_pairs - 1;                                                                      \
        const int y0_ = (2 * gp0_) % S, yl_ = (2 * gpl_) % S + 1;                                                                \
        qmin_ = 2 * gp0_ - (y0_ >= 2 ? 2 : 0);                                                                                   \
        nrows_ = 2 * gpl_ + 1 + (yl_ + 2 <= S - 1 ? 2 : 0) - qmin_ + 1;                                                          \
    } while (0)
    // the rows the NEXT pass needs that are not in the ring yet: [lo, hi); the next pass itself (the last pass of a ticket moves on to the
    // ticket drawn one round earlier).  Computed identically by every wave.
#define RS_NEXT(pass_, res_hi_, next_, have_, lo_, hi_)                                                                          \
    do {                                                                                                                         \
        const bool draw_ = tleft == 1;                                                                                           \
        next_ = draw_ ? __builtin_amdgcn_readfirstlane(*s_next) : (pass_) + 1;                                                   \
        have_ = next_ < n_pass;                                                                                                  \
        lo_ = hi_ = 0;                                                                                                           \
        if (have_) {                                                                                                             \
            int qn_, nn_;                                                                                                        \
            RS_ROWS(next_, qn_, nn_);                                                                                            \
            lo_ = (next_ == (pass_) + 1 && (res_hi_) > qn_) ? (res_hi_) : qn_;                                                   \
            hi_ = qn_ + nn_;                                                                                                     \
        }                                                                                                                        \
    } while (0)
    __syncthreads();

    int qmin, nrows;
    RS_ROWS(pass, qmin, nrows);
    if (producer) {
        // ------------------------------------------------------------------------------------------------------------------------
        // PRODUCER: V3 transform of the previous pass, V2 rows of the next one
        // ------------------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio((PRIO >> 8) & 3);
        // the crop-row unit (16 pixels) item `it` of a chunk that starts at V2 row c0 stands for: item = (V2 row v, crop row k of its six, unit u)
#define RS_ITEM(it_, c0_)                                                                                                        \
                const int vk = (it_) / 5, u = (it_) - vk * 5;                                                                    \
                const int v = vk / 6, k = vk - v * 6;                                                                            \
                int q = (c0_) + v;                                                                                               \
                q = q < total_rows ? q : total_rows - 1;                                                                         \
                const int crop = q / S, y = q - crop * S;                                                                        \
                const int iy = 2 * y - 2 + k;
        // P0 in two halves: the loads (issued in front of the V3 transform, whose instructions hide their latency), then the conversion
        auto p0_load = [&](const int c0, const int nr, uint4& px) {
            px = make_uint4(0, 0, 0, 0);
            if (rt < nr * 30) {
                RS_ITEM(rt, c0)
                if (!(DBG & 1) && iy >= 0 && iy < 80) px = *reinterpret_cast<const uint4*>(crops + ((size_t)crop * 80 + iy) * 80 + u * 16);
            }
        };
        auto p0_store = [&](const int nr, const uint4 px) {
            if (rt < nr * 30) {
                const int vk = rt / 5, u = rt - vk * 5;
                const uint32_t w4[4] = {px.x, px.y, px.z, px.w};
                uint32_t* d = reinterpret_cast<uint32_t*>(img + vk * F::IMG_PITCH + 2 + u * 16);       // (4-byte aligned: 2 halves of left padding)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t b0 = (w4[e >> 1] >> (16 * (e & 1))) & 0xffu, b1 = (w4[e >> 1] >> (16 * (e & 1) + 8)) & 0xffu;
                    d[e] = pack_h2((_Float16)(float)b0, (_Float16)(float)b1);
                }
            }
        };
        // P1: conv1 tiles (8 windows of 4 outputs x 2 image rows; 20 windows per V2 row), role wave w takes tiles w, w + 4, ...  k_conv1_wpre's
        // tiles and products in its order (two fp16 weight pieces, low pieces first), bias, ReLU, 2x2 max-pool -> pbufP
        const float bz1 = bias1[lane & 15];
        // conv1's sixteen weight fragments stay in registers for the whole kernel (a producer wave holds no accumulators): fragment (s, mf, piece) =
        // the base fragment (mf, piece) moved up by s window slots (16 s bits; the slots it leaves are zero weights, the three it pushes out were zero)
        uint4 bf[16];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint4 b = w1tab[f * 64 + lane];
            bf[f] = b;
            bf[4 + f] = make_uint4(b.x << 16, __builtin_amdgcn_alignbit(b.y, b.x, 16), __builtin_amdgcn_alignbit(b.z, b.y, 16), __builtin_amdgcn_alignbit(b.w, b.z, 16));
            bf[8 + f] = make_uint4(0u, b.x, b.y, b.z);
            bf[12 + f] = make_uint4(0u, b.x << 16, __builtin_amdgcn_alignbit(b.y, b.x, 16), __builtin_amdgcn_alignbit(b.z, b.y, 16));
        }
        auto p1 = [&](const int nr) {
            const int r = lane & 15, q4 = lane >> 4;
            const int n_win = nr * 20, n_tiles = (n_win + 7) >> 3;
            // all (up to four) tiles of the wave in one step (t0, t0 + 4, t0 + 8, t0 + 12): every LDS read first, then the sixteen product chains, then
            // the pooling -- the vector part of the stage then runs beside the consumers' taps instead of between two bursts of matrix work
            constexpr int NT1 = 4;
            for (int t0 = rw; t0 < n_tiles; t0 += 4 * NT1) {
                f16x8_c1 a1[NT1], a2[NT1];
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    const int tile = t0 + 4 * u < n_tiles ? t0 + 4 * u : t0;
                    int wdx = tile * 8 + (r >> 1);
                    wdx = wdx < n_win ? wdx : n_win - 1;
                    const int v = wdx / 20, x4 = (wdx - v * 20) * 4;
                    const int row = v * 6 + (r & 1);
                    const _Float16* p1a = img + (row + q4) * F::IMG_PITCH + x4;
                    const _Float16* p2a = img + (row + 4) * F::IMG_PITCH + x4;
                    const uint2 l1 = *reinterpret_cast<const uint2*>(p1a), h1 = *reinterpret_cast<const uint2*>(p1a + 4);
                    const uint2 l2 = *reinterpret_cast<const uint2*>(p2a), h2 = *reinterpret_cast<const uint2*>(p2a + 4);
                    a1[u] = __builtin_bit_cast(f16x8_c1, make_uint4(l1.x, l1.y, h1.x, h1.y));
                    a2[u] = __builtin_bit_cast(f16x8_c1, make_uint4(l2.x, l2.y, h2.x, h2.y));
                }
                f32x4 acc[NT1][4];
#pragma unroll
                for (int u = 0; u < NT1; ++u)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[u][s] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(DBG & 2)) {
                    // per chain the order of k_conv1_wpre: low pieces first (fragments 1, 3), then the high ones (0, 2)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int u = 0; u < NT1; ++u)
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16((m & 1) ? a2[u] : a1[u], __builtin_bit_cast(f16x8_c1, bf[s * 4 + (m == 0 ? 1 : m == 1 ? 3 : m == 2 ? 0 : 2)]), acc[u][s], 0, 0, 0);
                }
                // lane (co = r, q4): accumulator rows 4 q4 .. 4 q4 + 3 = windows 2 q4, 2 q4 + 1 x the two image rows: pooled pixels 2 (window) and
                // 2 (window) + 1 of the V2 row, channel co.  pbufP [slot of the pooled pixel][16 channels]; slot = the pixel index with its two low
                // bit pairs swapped (the four q4 groups of a store fill 256 contiguous bytes, P2's lanes read contiguously as well)
#pragma unroll
                for (int u = 0; u < NT1; ++u) {
                    const int tile = t0 + 4 * u;
#pragma unroll
                    for (int pos = 0; pos < 2; ++pos) {
                        const int w = tile * 8 + 2 * q4 + pos;
                        // bias + ReLU per element first (v_fma_f32 written by the compiler: it pads the MFMA -> VALU read), then the pool as two
                        // v_max3: relu(max(x) s + b) = max(relu(x s + b)) exactly (s > 0, rounding is monotone), 6 instructions instead of 9.
                        // (the asm helpers only ever see VALU results: an MFMA result read by inline asm is not padded by the hazard recognizer)
                        const float t0 = acc[u][0][2 * pos] * inv_scale1 + bz1, t1 = acc[u][0][2 * pos + 1] * inv_scale1 + bz1;
                        const float t2 = acc[u][1][2 * pos] * inv_scale1 + bz1, t3 = acc[u][1][2 * pos + 1] * inv_scale1 + bz1;
                        const float t4 = acc[u][2][2 * pos] * inv_scale1 + bz1, t5 = acc[u][2][2 * pos + 1] * inv_scale1 + bz1;
                        const float t6 = acc[u][3][2 * pos] * inv_scale1 + bz1, t7 = acc[u][3][2 * pos + 1] * inv_scale1 + bz1;
                        const float v0 = rs_max3z(rs_max3(t0, t1, t2), t3), v1 = rs_max3z(rs_max3(t4, t5, t6), t7);
                        if (tile < n_tiles && w < n_win) {
                            ovfm = rs_max3(ovfm, v0, v1);
                            const int vr = w / 20, px = vr * F::PPP + 4 + 2 * (w - vr * 20);     // row of the chunk, padded pixel index (even)
                            float* o = pbp + ((px & ~15) | ((px & 3) << 2) | ((px >> 2) & 3)) * 16 + r;
                            o[0] = v0;
                            o[64] = v1;                                          // px + 1: bit 0 of the pixel is bit 2 of the slot
                        }
                    }
                }
            }
        };
        // P2: (V2 row v, conv2 tile tx, channel quad): 8 pooled pixels x 4 channels -> B^T d -> pieces -> the ring slots (q % NR + 1) of the planes
        auto p2 = [&](const int c0, const int nr) {
            if (!(DBG & 4) && rt < nr * 40) {
                const int v = rt / 40, rem = rt - v * 40, tx = rem >> 2, quad = rem & 3;
                const int q = c0 + v;
                const int slot = q % G::NR + 1, rot = w2b_rot(slot);
                float4 d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int px = v * F::PPP + 4 * tx + 2 + k;                  // x = 4 tx - 2 + k in the padded row: the halo pixels are zeros
                    d[k] = *reinterpret_cast<const float4*>(pbp + ((px & ~15) | ((px & 3) << 2) | ((px >> 2) & 3)) * 16 + quad * 4);
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
        };
        // E2: the V3 transform of pass `ep`: (pooled row, conv3 tile, channel quad) items of pbufE -> B^T d -> pieces -> V3
        auto e2 = [&](const int ep) {
            if (!(DBG & 16) && rt < 240) {
                const int rp = rt / 80, rem = rt - rp * 80, tx = rem >> 4, quad = rem & 15;
                const int gp = ep * G::RPP + rp;                            // = q3: pooled row of the batch
                if (gp < total_pairs) {
                    float4 d[8];
                    const float* src = pbe + (rp * F::PEP + 4 * tx) * 64 + quad * 4;      // x = 4 tx - 2 in the padded row: the halo pixels are zeros
#pragma unroll
                    for (int k = 0; k < 8; ++k) d[k] = *reinterpret_cast<const float4*>(src + k * 64);
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
        };
        // prologue: every row of the first pass, chunk by chunk (three barriers each; the consumers wait at them)
        for (int c0 = qmin; c0 < qmin + nrows; c0 += F::CHUNK) {
            const int nr = qmin + nrows - c0 < F::CHUNK ? qmin + nrows - c0 : F::CHUNK;
            uint4 px;
            p0_load(c0, nr, px);
            p0_store(nr, px);
            RS_BAR();
            p1(nr);
            flag_crops(c0, c0 + nr - 1, S);
            RS_BAR();
            p2(c0, nr);
            RS_BAR();
        }
        int res_hi = qmin + nrows;
        int prev = -1;
        if ((DBG & 128) && st_on) st_last = __builtin_readcyclecounter();
        for (;;) {
            int next_pass, lo, hi;
            bool have_next;
            RS_NEXT(pass, res_hi, next_pass, have_next, lo, hi);
            // S1 (beside taps 0-19): the crop rows of the first chunk are requested, the V3 transform of the previous pass runs under their flight
            const int nr0 = hi - lo < F::CHUNK ? hi - lo : F::CHUNK;
            uint4 px;
            if (!(DBG & 8)) p0_load(lo, nr0, px);
            if (prev >= 0) e2(prev);
            if (!(DBG & 8)) p0_store(nr0, px);
            RS_STAMP(0);
            RS_BAR();
            RS_STAMP(1);
            // S2 (beside taps 20-39): conv1
            if (!(DBG & 8)) { p1(nr0); if (nr0 > 0) flag_crops(lo, lo + nr0 - 1 < total_rows ? lo + nr0 - 1 : total_rows - 1, S); }
            RS_STAMP(2);
            RS_BAR();
            RS_STAMP(3);
            // S3 (beside the output transform; nobody reads the ring): the rows go to their slots
            if (!(DBG & 8)) p2(lo, nr0);
            RS_STAMP(4);
            RS_BAR();
            RS_STAMP(5);
            // a ticket's first pass needs 10 rows: the second chunk in three more stages (the consumers wait)
            for (int c0 = lo + F::CHUNK; c0 < hi; c0 += F::CHUNK) {
                const int nr = hi - c0 < F::CHUNK ? hi - c0 : F::CHUNK;
                if (!(DBG & 8)) { p0_load(c0, nr, px); p0_store(nr, px); }
                RS_BAR();
                if (!(DBG & 8)) { p1(nr); flag_crops(c0, c0 + nr - 1 < total_rows ? c0 + nr - 1 : total_rows - 1, S); }
                RS_BAR();
                if (!(DBG & 8)) p2(c0, nr);
                RS_BAR();
            }
            RS_STAMP(6);
            if ((DBG & 128) && st_on) st_sum[7] += 1;
            prev = pass;
            if (!have_next) break;
            res_hi = hi;
            tleft = tleft == 1 ? PK : tleft - 1;
            pass = next_pass;
        }
        e2(prev);                                                         // the last pass's activations (behind its third barrier)
#undef RS_ITEM
    } else {
        // ------------------------------------------------------------------------------------------------------------------------
        // CONSUMER: conv2's tap loop and output transform
        // ------------------------------------------------------------------------------------------------------------------------
        const int j = lane & 31, h = lane >> 5;
        const int n = rw & 1, mg = rw >> 1;
        const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (uint32_t)(40 * G::BV * 16));
        const int boff = (h * CO + n * 32 + j) * 16;
        const int co = n * 32 + j;
        const float bz = bias[co];
        for (int c0 = qmin; c0 < qmin + nrows; c0 += F::CHUNK) { RS_BAR(); RS_BAR(); RS_BAR(); }      // the producers' prologue
        int res_hi = qmin + nrows;
#define W2_POS(tau_) rs_pos(ORD, tau_)
#define W2_BOFF(tau_) ((rs_ky(ORD, tau_) * 8 + W2_POS(tau_)) * G::BV * 16)
        // the weight fragments of taps 0 .. BD - 1 are the same for every pass: the last taps of a pass fetch them for the next one
        uint4 bq[8][2];
        const uint8_t* wres = ldsb + F::WRES_OFF + boff;
#define RS_WFETCH(dst_, tau_)                                                                                                    \
        do {                                                                                                                     \
            if (F::resident(tau_, RESPAT)) {                                                                                     \
                dst_[0] = *reinterpret_cast<const uint4*>(wres + F::res_slot(tau_, RESPAT) * F::WTAP);                           \
                dst_[1] = *reinterpret_cast<const uint4*>(wres + F::res_slot(tau_, RESPAT) * F::WTAP + 2 * CO * 16);             \
            } else {                                                                                                             \
                dst_[0] = buf_load16(wrs, boff, W2_BOFF(tau_));                                                                  \
                dst_[1] = (DBG & 64) ? dst_[0] : buf_load16(wrs, boff, W2_BOFF(tau_) + 2 * CO * 16);                             \
            }                                                                                                                    \
        } while (0)
#pragma unroll
        for (int t = 0; t < BD; ++t) RS_WFETCH(bq[t], t);
        // A-operand byte offsets of this lane's tile for pass `ps_`: per kernel row the row slot (out-of-crop rows -> the zero row), per position of a
        // group the rotated unit.  Computed for the NEXT pass behind the output transform, off the path to the first tap
        // A-operand byte offsets of this lane's tile: per kernel row the row slot of the pass (out-of-crop rows -> the zero row) and its rotation --
        // computed for the NEXT pass behind the output transform -- and per position of a group the rotated unit, put together inside taps 0-19,
        // one tap ahead of its first use (taps 20-39 use the same twenty offsets for the second position group)
        int aoff[5][4], srow[5], srot[5];
        int wh[4], wl[4];
        {
            int s_ = mg * 32 + j;
            s_ = s_ < G::RPP * G::TPP ? s_ : G::RPP * G::TPP - 1;
            const int r2_ = s_ % G::TPP, tx_ = r2_ >> 1;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) { const int w_ = pg * 20 + tx_ * 2 + h; wh[pg] = (w_ & ~15) * 16; wl[pg] = w_ & 15; }
        }
#define RS_AOFF(ps_)                                                                                                             \
        do {                                                                                                                     \
            int s_ = mg * 32 + j;                                                                                                \
            s_ = s_ < G::RPP * G::TPP ? s_ : G::RPP * G::TPP - 1;                                                                \
            const int rp_ = s_ / G::TPP, r2_ = s_ - rp_ * G::TPP;                                                                \
            int gp_ = (ps_) * G::RPP + rp_;                                                                                      \
            gp_ = gp_ < total_pairs ? gp_ : total_pairs - 1;                                                                     \
            const int qo_ = 2 * gp_ + (r2_ & 1), y_ = qo_ % S;                                                                   \
            _Pragma("unroll") for (int ky = 0; ky < 5; ++ky) {                                                                   \
                const int iy_ = y_ + ky - 2;                                                                                     \
                const int slot_ = (iy_ >= 0 && iy_ < S) ? (qo_ + ky - 2) % G::NR + 1 : 0;                                        \
                srow[ky] = slot_ * G::ROWL; srot[ky] = w2b_rot(slot_);                                                           \
            }                                                                                                                    \
        } while (0)
#define RS_AUNIT(tau_) do { if ((tau_) < 20) aoff[rs_ky(ORD, tau_)][rs_pg(ORD, tau_)] = srow[rs_ky(ORD, tau_)] + (wh[rs_pg(ORD, tau_)] | (((wl[rs_pg(ORD, tau_)] + srot[rs_ky(ORD, tau_)]) & 15) << 4)); } while (0)
        RS_AOFF(pass);
        if ((DBG & 128) && st_on) st_last = __builtin_readcyclecounter();
        for (;;) {
            int next_pass, lo, hi;
            bool have_next;
            RS_NEXT(pass, res_hi, next_pass, have_next, lo, hi);
            const bool draw = tleft == 1;
            uint32_t ticket = 0;
            // the ticket after the next one (raw instruction: atomicAdd() waits for the returned value on the spot); stored behind the second barrier
            if (draw && tid == 0) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(pass_ctr), "v"(1u) : "memory");
            f32x16 acc[8];
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_s_setprio((PRIO >> 4) & 3);
            uint4 af[PAIR ? 4 : 2][2];
            // ORD 1: the part of A^T whose positions are final, two accumulator rows per tap, in place (an empty volatile asm keeps each result at
            // its tap; the operations and their order are the output transform's: results stay bit-identical).  After it acc[0] = y0, acc[2] = y1,
            // acc[1] = y2, acc[7] = y3 up to the pair (5,6), which stage 3 adds.
            auto rs_fold = [&](const int tau) {
#define RS_PIN(x_) asm volatile("" : "+v"(x_))
                if (tau >= 11 && tau < 19) {                               // positions 1, 2 (final after tap 9): e1 -> acc[1], o1 -> acc[2]
#pragma unroll
                    for (int r = 2 * (tau - 11); r < 2 * (tau - 11) + 2; ++r) {
                        float e = acc[1][r] + acc[2][r], o = acc[1][r] - acc[2][r];
                        RS_PIN(e); RS_PIN(o); acc[1][r] = e; acc[2][r] = o;
                    }
                }
                if (tau >= 21 && tau < 29) {                               // positions 0, 7 (final after tap 19): y0 = a0 + e1, y3 = o1 + a7
#pragma unroll
                    for (int r = 2 * (tau - 21); r < 2 * (tau - 21) + 2; ++r) {
                        float a = acc[0][r] + acc[1][r], d = acc[2][r] + acc[7][r];
                        RS_PIN(a); RS_PIN(d); acc[0][r] = a; acc[7][r] = d;
                    }
                }
                if (tau >= 31 && tau < 39) {                               // positions 3, 4 (final after tap 29)
#pragma unroll
                    for (int r = 2 * (tau - 31); r < 2 * (tau - 31) + 2; ++r) {
                        const float e2 = acc[3][r] + acc[4][r], o2 = acc[3][r] - acc[4][r];
                        float a = acc[0][r] + e2, b = acc[2][r] + 2.f * o2, c = acc[1][r] + 4.f * e2, d = acc[7][r] + 8.f * o2;
                        RS_PIN(a); RS_PIN(b); RS_PIN(c); RS_PIN(d); acc[0][r] = a; acc[2][r] = b; acc[1][r] = c; acc[7][r] = d;
                    }
                }
#undef RS_PIN
            };
#define RS_AREAD(dst_, tau_)                                                                                                     \
            do {                                                                                                                 \
                const uint8_t* an_ = ldsb + ((tau_) / 20) * G::BUF + aoff[rs_ky(ORD, tau_)][rs_pg(ORD, tau_)];                   \
                dst_[0] = *reinterpret_cast<const uint4*>(an_);                                                                  \
                dst_[1] = *reinterpret_cast<const uint4*>(an_ + G::PLANE);                                                       \
            } while (0)
#define RS_TAP(tau)                                                                                                              \
            do {                                                                                                                 \
                const int tl = (tau) % 20;                                                                                       \
                if ((tau) + 1 < 40) { RS_AUNIT((tau) + 1); RS_AREAD(af[((tau) + 1) % 2], (tau) + 1); }                           \
                RS_WFETCH(bq[((tau) + BD) % 8], ((tau) + BD) % 40);                                                              \
                const int p = W2_POS(tau);                                                                                       \
                const f16x8 b1 = __builtin_bit_cast(f16x8, bq[(tau) % 8][0]), b2 = __builtin_bit_cast(f16x8, bq[(tau) % 8][1]); \
                const f16x8 a1 = __builtin_bit_cast(f16x8, af[(tau) % 2][0]), a2 = __builtin_bit_cast(f16x8, af[(tau) % 2][1]); \
                acc[p] = mfma16(a2, b1, rs_ky(ORD, tau) == 0 ? zero16 : acc[p]);      /* kernel row 0 starts the accumulator */   \
                acc[p] = mfma16(a1, b2, acc[p]);                                                                                 \
                acc[p] = mfma16(a1, b1, acc[p]);                                                                                 \
                if constexpr (ORD == 1) { if (!(DBG & 16)) rs_fold(tau); }                                                        \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                                               \
                _Pragma("unroll") for (int g = 0; g < 3; ++g) {                                                                  \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
                    __builtin_amdgcn_sched_group_barrier(0x006, ORD ? 6 : 4, 0);                                                 \
                }                                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
            } while (0)
            // two taps (tau, tau + 1: two positions of one kernel row) at a time, their three products interleaved: consecutive MFMAs go to different
            // accumulators.  A wave whose next MFMA waits for its own accumulator holds the SIMD's issue port (profiles/r05_ubench_simd.txt: beside
            // dependent MFMAs a vector wave of equal priority got 1 % of the port, beside independent ones 84 %) -- and here the producer wave outranks
            // the tap loop, so every port conflict lands on a dependent chain of the consumer
#define RS_TAP2(tau)                                                                                                             \
            do {                                                                                                                 \
                const int tl = (tau) % 20;                                                                                       \
                if ((tau) + 2 < 40) {                                                                                            \
                    RS_AUNIT((tau) + 2); RS_AREAD(af[((tau) + 2) % 4], (tau) + 2);                                               \
                    RS_AUNIT((tau) + 3); RS_AREAD(af[((tau) + 3) % 4], (tau) + 3);                                               \
                }                                                                                                                \
                RS_WFETCH(bq[((tau) + BD) % 8], ((tau) + BD) % 40);                                                              \
                RS_WFETCH(bq[((tau) + BD + 1) % 8], ((tau) + BD + 1) % 40);                                                      \
                const int p = W2_POS(tau), q = W2_POS((tau) + 1);                                                                \
                const f16x8 b1 = __builtin_bit_cast(f16x8, bq[(tau) % 8][0]), b2 = __builtin_bit_cast(f16x8, bq[(tau) % 8][1]); \
                const f16x8 c1 = __builtin_bit_cast(f16x8, bq[((tau) + 1) % 8][0]), c2 = __builtin_bit_cast(f16x8, bq[((tau) + 1) % 8][1]); \
                const f16x8 a1 = __builtin_bit_cast(f16x8, af[(tau) % 4][0]), a2 = __builtin_bit_cast(f16x8, af[(tau) % 4][1]); \
                const f16x8 d1 = __builtin_bit_cast(f16x8, af[((tau) + 1) % 4][0]), d2 = __builtin_bit_cast(f16x8, af[((tau) + 1) % 4][1]); \
                acc[p] = mfma16(a2, b1, rs_ky(ORD, tau) == 0 ? zero16 : acc[p]);                                                 \
                acc[q] = mfma16(d2, c1, rs_ky(ORD, (tau) + 1) == 0 ? zero16 : acc[q]);                                           \
                acc[p] = mfma16(a1, b2, acc[p]);                                                                                 \
                acc[q] = mfma16(d1, c2, acc[q]);                                                                                 \
                acc[p] = mfma16(a1, b1, acc[p]);                                                                                 \
                acc[q] = mfma16(d1, c1, acc[q]);                                                                                 \
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);        /* the A fragments of the next pair (+ resident weight fragments) */ \
                __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);        /* weight fragments from L2 */                          \
                _Pragma("unroll") for (int g = 0; g < 6; ++g) {                                                                  \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);                                                           \
                }                                                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
            } while (0)
            RS_AUNIT(0);
            RS_AREAD(af[0], 0);
            if constexpr (PAIR) { RS_AUNIT(1); RS_AREAD(af[1], 1); }
            if constexpr (PAIR) {
                static_assert(!PAIR || (TSPLIT % 2 == 0 && BD % 2 == 0 && BD <= 4), "pairs of taps");
#pragma clang loop unroll(full)
                for (int tau = 0; tau < ((DBG & 32) ? 0 : TSPLIT); tau += 2) RS_TAP2(tau);
            } else {
#pragma clang loop unroll(full)
            for (int tau = 0; tau < ((DBG & 32) ? 0 : TSPLIT); ++tau) RS_TAP(tau);
            }
            RS_STAMP(0);
            RS_BAR();                                                     // S1 | S2
            RS_STAMP(1);
            if constexpr (PAIR) {
#pragma clang loop unroll(full)
                for (int tau = TSPLIT; tau < ((DBG & 32) ? 0 : 40); tau += 2) RS_TAP2(tau);
            } else {
#pragma clang loop unroll(full)
            for (int tau = TSPLIT; tau < ((DBG & 32) ? 0 : 40); ++tau) RS_TAP(tau);
            }
            if (DBG & 32) { _Pragma("unroll") for (int p = 0; p < 8; ++p) acc[p] = zero16; }
            if (draw) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket) :: "memory");
                if (tid == 0) *s_next = ((int)ticket + (int)gridDim.x) * PK;       // read at the start of a later round
            }
            RS_STAMP(2);
            RS_BAR();                                                     // S2 | S3: the producers have read pbufE (in S1), it may be written
            RS_STAMP(3);
            __builtin_amdgcn_s_setprio(PRIO & 3);
            // output transform: Y = A^T M, pool, bias, ReLU -> the pass's 3 x 20 x 64 activations as fp32 in pbufE
            if (!(DBG & 16)) {
                f32x16 y0, y1, y2, y3;
                if constexpr (ORD == 1) { y0 = acc[0]; y1 = acc[2]; y2 = acc[1]; y3 = acc[7]; }
                else {
                    {
                        const f32x16 e1 = acc[1] + acc[2], o1 = acc[1] - acc[2];
                        y0 = acc[0] + e1; y1 = o1; y2 = e1; y3 = o1 + acc[7];
                    }
                    {
                        const f32x16 e2 = acc[3] + acc[4], o2 = acc[3] - acc[4];
                        y0 += e2; y1 += 2.f * o2; y2 += 4.f * e2; y3 += 8.f * o2;
                    }
                }
                {
                    const f32x16 e3 = acc[5] + acc[6], o3 = acc[5] - acc[6];
                    y0 += e3; y1 += 0.5f * o3; y2 += 0.25f * e3; y3 += 0.125f * o3;
                }
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 2 * rr;
                    const int s = mg * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // even: rows y, y+1 of one tile column
                    const float v0 = rs_max(rs_max3(y0[r], y1[r], y0[r + 1]), y1[r + 1]);
                    const float v1 = rs_max(rs_max3(y2[r], y3[r], y2[r + 1]), y3[r + 1]);
                    if (s < G::RPP * G::TPP) {
                        const int rp = s / G::TPP, tx = (s - rp * G::TPP) >> 1;
                        const float a0 = fmaxf(v0 * out_scale + bz, 0.f), a1 = fmaxf(v1 * out_scale + bz, 0.f);
                        ovfm = rs_max3(ovfm, a0, a1);
                        float* o = pbe + (rp * F::PEP + 2 + 2 * tx) * 64 + co;
                        o[0] = a0;
                        o[64] = a1;
                    }
                }
            }
            { const int g0 = pass * G::RPP, g1 = g0 + G::RPP - 1; flag_crops(g0, g1 < total_pairs ? g1 : total_pairs - 1, S / 2); }
            if (have_next) RS_AOFF(next_pass);
            RS_STAMP(4);
            RS_BAR();                                                     // S3 | the next round
            RS_STAMP(5);
            for (int c0 = lo + F::CHUNK; c0 < hi; c0 += F::CHUNK) { RS_BAR(); RS_BAR(); RS_BAR(); }    // the second chunk of a ticket's first pass
            RS_STAMP(6);
            if ((DBG & 128) && st_on) st_sum[7] += 1;
            if (!have_next) break;
            res_hi = hi;
            tleft = tleft == 1 ? PK : tleft - 1;
            pass = next_pass;
        }
#undef RS_TAP
#undef RS_TAP2
#undef RS_WFETCH
#undef RS_AOFF
#undef RS_AUNIT
#undef RS_AREAD
#undef W2_POS
#undef W2_BOFF
    }
    if ((DBG & 128) && st_on) { _Pragma("unroll") for (int i = 0; i < 8; ++i) dbg_stamps[(tid ? 8 : 0) + i] = st_sum[i]; }
#undef RS_STAMP
#undef RS_NEXT
#undef RS_ROWS
#undef RS_BAR
    if (__any(!(ovfm < 4368.0f)) && lane == 0) atomicOr(overflow, 3u);      // (nothing is left here behind the last flag_crops; if it ever is, it is unattributed: every crop)
}
