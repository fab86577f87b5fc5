// cnn_conv3p.h -- conv3 of the identity network's default chain, walked PAIR BY PAIR (round 5).  Included by cnn.hip after cnn_wpre.h.
//
// Network: visual_identification_network_torch.py:184-258 (V118_3, eval mode): conv3 = 5x5 'same', 64 -> 128 channels on the 20x20 map, then
// ReLU and the 2x2 max-pool.  Arithmetic as in k_conv5_wpre (cnn_wpre.h): F(4,5) Winograd along x -- 8 position products per kernel row give 4
// neighbouring outputs --, direct along y, both operands as two fp16 pieces, three piece products per term, fp32 accumulation; per position the
// order of the sums is the same (16-channel chunk, then kernel row), so the position sums are bit-identical to k_conv5_wpre's.
//
// What is different is the ORDER OF THE POSITIONS.  k_conv5_wpre walks a pass chunk by chunk with all 8 positions of a chunk inside one tap
// group: every position is final only at the very end of the pass, 16 accumulator tuples (all 256 accumulator registers) are live throughout,
// and the output transform A^T, the pool and the stores -- ~1000 vector instructions per pass -- wait behind the last tap with nothing to run
// beside them (one wave per SIMD): 19 % of a pass.  Here a pass is 4 UNITS of 40 taps, one per position pair in the order (1,2) (3,4) (5,6) (0,7),
// each over all 64 input channels (V3 is laid out for that, cnn_wpre.h).  A^T combines exactly these pairs:
//     e_k = M_a + M_b, o_k = M_a - M_b;   y0 = e1 + e2 + e3 + M0,  y1 = o1 + 2 o2 + o3 / 2,  y2 = e1 + 4 e2 + e3 / 4,  y3 = o1 + 8 o2 + o3 / 8 + M7
// so the pair of unit g is folded into the four partial outputs UNDER THE TAPS OF UNIT g + 1 (1-2 vector instructions per MFMA: free, a wave's
// own vector instructions between its MFMAs cost nothing up to ~5 per MFMA, profiles/r05_ubench_simd.txt), and the last unit's pair, the pool,
// bias, ReLU and the stores run under unit 0 of the wave's NEXT pass.  Two accumulator banks (a unit writes one while the other is folded) =
// 128 accumulator registers instead of 256; the partial outputs of both tiles live in 128 vector registers.  Nothing of the epilogue is left
// behind the last tap; the next pass's A offsets are computed under the last unit as well.
// (y0 takes position 0 last instead of first: the only change in the order of a sum against k_conv5_wpre; probabilities differ by <= 6e-7.)
//
// Measured (25600 crops, A/B against k_conv5_wpre on the same boxes, tools/r05_conv3.sh): 4.42 -> 4.21, 4.57 -> 4.38 ms; matrix pipe 0.64 ->
// 0.69 busy (profiles/r05_pmc_conv_pair.txt) at a clock that FALLS as the pipe fills (1.56 -> 1.50 GHz: the kernel is power-limited; with its
// operand loads switched off the same 960 MFMAs per pass run at 2.6 GHz).  A first version with the same tap order was SLOWER than
// k_conv5_wpre (4.50 against 4.39): 47 instructions per tap instead of 28 -- spilled scalars, hoisted address arithmetic, lane masks around the
// DMA -- and a wave issues one instruction per ~4.5 cycles, so a tap of 6 MFMAs (192 cycles) has room for ~40.  Weight fragments 3 / 5 / 7 taps
// ahead: 4.52 / 4.39 / 4.39; A fragments 2 taps ahead: no change.

template <int DBG = 0, int BD = 7, int AD = 1, int PK = 4, int DT0 = 2>      // DBG (dev builds): 1 no staging, 2 no folding / tail, 4 no weight loads, 8 no A reads; BD: taps of lead of the weight fragments; PK: passes per ticket
__global__ __launch_bounds__(256) void k_conv5_wpair(const uint8_t* __restrict__ v3, const uint4* __restrict__ wp /*[4][5][8][2][2][128] x 16 B*/,
                                                     const float* __restrict__ bias, float* __restrict__ out, const float out_scale,
                                                     const int n_crops, uint32_t* __restrict__ pass_ctr,
                                                     const int n_big /* tickets of PK passes; the passes behind them go out one by one */) {
    constexpr int CI = 64, CO = 128, S = 20, TPW = 2;
    using G = WinoGeom<CI, CO, S, TPW>;
    static_assert(V3_PAIR, "k_conv5_wpair reads the pair-major V3");
    static_assert(G::NTHR == 256 && G::RP0 == 1280 && G::NCH * 2 * G::RP0 == V3_ROWB && G::NT == 4 && G::WM == 1, "geometry");
    static_assert(BD >= 1 && BD <= 7, "weight ring of 8 taps");
    extern __shared__ __attribute__((aligned(16))) uint8_t ldsb[];
    __shared__ int s_next_pass;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int total_tiles = n_crops * G::TPC;
    const int n_pass = (total_tiles + G::MB - 1) / G::MB;
    const int big_end = n_big * PK;
    auto ticket_first = [&](const int t) { return t < n_big ? t * PK : big_end + (t - n_big); };
    int pass = ticket_first((int)blockIdx.x);
    if (pass >= n_pass) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)ldsb;
    // Staging of a unit by LDS-DMA (no registers, no LDS store instructions).  The two piece planes of a buffer are ONE linear LDS range from row
    // slot 1 of piece 0 to the last row of piece 1 (piece 1's zero row in between) = NI instructions of 1 KB; lane l of instruction i covers
    // byte o = 1024 i + 16 l of it and fetches its unit of (row, piece) from the unit's rows through a BUFFER resource over exactly the pass's
    // rows: what is out of range -- rows the pass does not have, the lanes over piece 1's zero row, the lanes past the last row (they land on
    // the next buffer's zero row, or in the kilobyte of slack behind the second buffer) -- is given an offset outside the resource and comes
    // back as ZEROS, which is what a zero row holds.  No clamps, no lane masks, no branches: per instruction one scalar add (M0), the lane's
    // precomputed offset, the load.  Wave w issues the instructions w, w + 4, ...: NDW per wave and unit, one per tap.
    constexpr int RNG = (2 * G::NR + 1) * G::RP, NI = (RNG + 1023) / 1024, NDW = (NI + 3) / 4;
    static_assert(DT0 + NDW <= 40 - BD, "the DMA instructions are older than the last 2 BD weight loads of a unit");
    static_assert(NI * 1024 - RNG <= 1024, "slack behind the second buffer");
    uint32_t dvo[NDW];
#pragma unroll
    for (int k = 0; k < NDW; ++k) {
        const int o = (wave + 4 * k) * 1024 + lane * 16;
        const int pc = o >= (G::NR + 1) * G::RP ? 1 : 0, q = o - pc * (G::NR + 1) * G::RP;
        const int row = q / G::RP, w = q - row * G::RP;
        dvo[k] = (q < G::NR * G::RP && o < RNG) ? (uint32_t)(row * V3_ROWB + pc * 1280 + (w < G::RP0 ? w : G::RP0 - 16)) : 0x80000000u;
    }
    const uint32_t mbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + G::RP + wave * 1024));
    auto stage_rsrc = [&](const int g, const int qmin_, const int nrows_) {
        const unsigned long long base = wave_uniform64(reinterpret_cast<unsigned long long>(v3) + (unsigned long long)qmin_ * V3_ROWB + (unsigned)(g * 2560));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, __builtin_amdgcn_readfirstlane(nrows_ * V3_ROWB - g * 2560), 0x00020000);
    };
    // (M0 is written and not restored: nothing else in this kernel reads it -- checked in the ISA)
#define P3_DMA(k_, srs_, buf_)                                                                                                   \
    do {                                                                                                                         \
        if ((k_) < NDW - 1 || wave + 4 * (NDW - 1) < NI)                                                                         \
            asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds"                          \
                         :: "s"(mbase), "n"((buf_) * G::BUF + (k_) * 4096), "v"(dvo[k_]), "s"(srs_) : "memory", "scc");           \
    } while (0)
    for (int i = tid; i < 4 * (G::RP / 16); i += G::NTHR) {          // row slot 0 of the four planes = the zero row
        const int pl = i / (G::RP / 16), o = i - pl * (G::RP / 16);
        *reinterpret_cast<uint4*>(ldsb + pl * G::PLANE + o * 16) = make_uint4(0, 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(wp, (uint32_t)(G::NCH * 40 * G::BV * 16));
    const int boff = (h * CO + wave * 32 + j) * 16;
    const int co = wave * 32 + j;
    const float bz = bias[co];
    const int ooff = (4 * h * CO + co) * 4;
    int qmin, nrows;
    wino_pass_rows<G, S>(pass, total_tiles, qmin, nrows);
    if (!(DBG & 1)) {
        const __amdgpu_buffer_rsrc_t srs0 = stage_rsrc(0, qmin, nrows);
#pragma unroll
        for (int k = 0; k < NDW; ++k) P3_DMA(k, srs0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // tap tt of unit g: chunk tt / 10, kernel row (tt % 10) / 2, position PAIRS[g][tt & 1]; its weights [chunk][ky][position] x BV x 16 bytes
    auto w_off = [](const int g, const int tt) {
        const int hp = tt & 1, p = g == 3 ? (hp ? 7 : 0) : 2 * g + 1 + hp;
        return ((tt / 10) * 40 + ((tt % 10) / 2) * 8 + p) * G::BV * 16;
    };
    uint4 bq[8][2];
#pragma unroll
    for (int t = 0; t < BD; ++t) {
        bq[t][0] = buf_load16(wrs, boff, w_off(0, t));
        bq[t][1] = buf_load16(wrs, boff, w_off(0, t) + 2 * CO * 16);
    }
    // A-operand byte offsets of this lane's two tiles, per kernel row: the row slot (out-of-crop rows -> the zero row) + the tile's 32 bytes
    // (per tile: the offset of its own input row and of the zero row, and its y; per kernel row a compare and a select)
    auto a_base = [&](const int pass_, const int qmin_, const int m, int& inr, int& zr, int& yy) {
        int T = pass_ * G::MB + m * 32 + j;
        if (T > total_tiles - 1) T = total_tiles - 1;
        const int gp = T / G::TPP, r2 = T - gp * G::TPP;
        const int tx = r2 >> 1, qo = 2 * gp + (r2 & 1);
        yy = qo % S; zr = tx * 32 + h * 16; inr = (qo - qmin_ + 1) * G::RP + zr;
    };
    auto a_offset = [&](const int inr, const int zr, const int yy, const int ky) {
        const int iy = yy + ky - 2;
        return (iy >= 0 && iy < S) ? inr + (ky - 2) * G::RP : zr;
    };
    int aoff[TPW][5];
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
        int inr, zr, yy;
        a_base(pass, qmin, m, inr, zr, yy);
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) aoff[m][ky] = a_offset(inr, zr, yy, ky);
    }
    int an_in[TPW] = {}, an_z[TPW] = {}, an_y[TPW] = {};
    // two accumulator banks x two tiles x the two positions of a pair; the partial outputs y0..y3 of both tiles.  All of it lives across
    // passes (the tail of a pass runs inside the next one); the first pass's "previous pass" stores into an empty buffer resource (dropped)
    // one accumulator register read where it is written (the compiler's own copies move a whole 16-register tuple to vector registers at
    // the first use: 32 registers the kernel does not have).  The MFMAs that wrote the tuple are a barrier and two taps back.
    auto P3_ACC = [](const float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[2][TPW][2];
    float y[TPW][4][16];
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k >> 1][m][k & 1] = zero16;
#pragma unroll
            for (int r = 0; r < 16; ++r) y[m][k][r] = 0.f;
        }
    }
    // (an empty volatile asm on a result keeps its arithmetic at the tap it is written under: left alone, the compiler sinks all of it to the
    // first use of the value -- the tail, one pass later -- and the registers of three units' accumulators with it)
#define P3_PIN(x_) asm volatile("" : "+v"(x_))
    __amdgpu_buffer_rsrc_t ors_prev = make_rsrc(out, 0);
    // pool (rows r, r + 1 x outputs 0,1 / 2,3), bias, ReLU, store.  Accumulator rows r, r + 1 (r even) of lane half h are the tiles i, i + 1 = the
    // two rows of a row pair at one tx, and the tile number of the even one IS the pooled output's index: T = 10 gp + 2 tx (TPP = 10 is even, so
    // T & 1 = the row's parity).  No division: the address is the lane's constant offset + a compile-time one, and tiles past the end of the
    // batch fall outside the pass's buffer resource (dropped by the hardware).
    auto tail = [&](const __amdgpu_buffer_rsrc_t rs, const int m, const int rr) {
        const int r = 2 * rr, i = (r & 3) + 8 * (r >> 2);
        const float a0 = y[m][0][r] + P3_ACC(acc[1][m][0][r]), a1 = y[m][0][r + 1] + P3_ACC(acc[1][m][0][r + 1]);
        const float d0 = y[m][3][r] + P3_ACC(acc[1][m][1][r]), d1 = y[m][3][r + 1] + P3_ACC(acc[1][m][1][r + 1]);
        const float v0 = fmaxf(fmaxf(a0, y[m][1][r]), fmaxf(a1, y[m][1][r + 1]));
        const float v1 = fmaxf(fmaxf(y[m][2][r], d0), fmaxf(y[m][2][r + 1], d1));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(v0 * out_scale + bz, 0.f)), rs, ooff, (m * 32 + i) * CO * 4, 2 /* nt */);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(v1 * out_scale + bz, 0.f)), rs, ooff, (m * 32 + i + 1) * CO * 4, 2);
    };
    // the pair of unit g (bank g & 1) folded into the partial outputs, one accumulator row of one tile at a time
    auto fold = [&](const int g, const int m, const int r) {
        const float a = P3_ACC(acc[g & 1][m][0][r]), b = P3_ACC(acc[g & 1][m][1][r]);
        const float e = a + b, o = a - b;
        if (g == 0) { y[m][2][r] = e; y[m][1][r] = o; P3_PIN(y[m][2][r]); P3_PIN(y[m][1][r]); }
        if (g == 1) {
            const float e1 = y[m][2][r], o1 = y[m][1][r];
            y[m][0][r] = e1 + e; y[m][1][r] = o1 + 2.f * o; y[m][2][r] = e1 + 4.f * e; y[m][3][r] = o1 + 8.f * o;
        }
        if (g == 2) { y[m][0][r] += e; y[m][1][r] += 0.5f * o; y[m][2][r] += 0.25f * e; y[m][3][r] += 0.125f * o; }
        if (g >= 1) { P3_PIN(y[m][0][r]); P3_PIN(y[m][1][r]); P3_PIN(y[m][2][r]); P3_PIN(y[m][3][r]); }
    };
    for (;;) {
        const int tleft = total_tiles - pass * G::MB;
        const __amdgpu_buffer_rsrc_t ors = make_rsrc(out + (size_t)(pass * G::MB) * CO, (uint32_t)((tleft < G::MB ? tleft : G::MB) * CO * 4));
        int next_pass = pass + 1;
        const bool draw = pass >= big_end || pass % PK == PK - 1;          // the last pass of a ticket draws the next one
        if (draw && tid == 0) s_next_pass = ticket_first((int)atomicAdd(pass_ctr, 1u) + (int)gridDim.x);   // read by everyone after the first unit's barrier
        bool have_next = false;
        int qmin_n = qmin, nrows_n = nrows;
#pragma clang loop unroll(full)
        for (int g = 0; g < 4; ++g) {
            const bool last_u = g == 3;
            if (last_u) {
                if (draw) next_pass = s_next_pass;
                have_next = next_pass < n_pass;
                if (have_next) wino_pass_rows<G, S>(next_pass, total_tiles, qmin_n, nrows_n);
            }
            const uint8_t* pbase = ldsb + (g & 1) * G::BUF;
            // staged under this unit: the next unit of this pass, or unit 0 of the next pass (without one: this pass's once more, into the buffer
            // nobody reads again)
            const int sg = last_u ? 0 : g + 1;
            const int sqmin = last_u ? qmin_n : qmin, snrows = last_u ? nrows_n : nrows;
            const __amdgpu_buffer_rsrc_t srs = stage_rsrc(sg, sqmin, snrows);
            // A fragments AD taps ahead (tap tt: position slot (tt / 10) * 2 + (tt & 1), kernel row (tt % 10) / 2)
            uint4 af[AD + 1][TPW][2];
#pragma unroll
            for (int a = 0; a < AD; ++a)
#pragma unroll
                for (int m = 0; m < TPW; ++m) {
                    af[a][m][0] = *reinterpret_cast<const uint4*>(pbase + (a & 1) * G::PS + aoff[m][a / 2]);
                    af[a][m][1] = *reinterpret_cast<const uint4*>(pbase + (a & 1) * G::PS + aoff[m][a / 2] + G::PLANE);
                }
#pragma clang loop unroll(full)
            for (int tt = 0; tt < 40; ++tt) {
                const int cur = tt % (AD + 1), nxt = (tt + AD) % (AD + 1);
                if (!(DBG & 8) && tt + AD < 40) {
                    const uint8_t* an = pbase + (((tt + AD) / 10) * 2 + ((tt + AD) & 1)) * G::PS;
#pragma unroll
                    for (int m = 0; m < TPW; ++m) {
                        af[nxt][m][0] = *reinterpret_cast<const uint4*>(an + aoff[m][((tt + AD) % 10) / 2]);
                        af[nxt][m][1] = *reinterpret_cast<const uint4*>(an + aoff[m][((tt + AD) % 10) / 2] + G::PLANE);
                    }
                }
                if (!(DBG & 4)) {
                    const int wt = tt + BD < 40 ? w_off(g, tt + BD) : w_off(sg, tt + BD - 40);
                    bq[(tt + BD) % 8][0] = buf_load16(wrs, boff, wt);
                    bq[(tt + BD) % 8][1] = buf_load16(wrs, boff, wt + 2 * CO * 16);
                }
                if (!(DBG & 1) && tt >= DT0 && tt < DT0 + NDW) P3_DMA(tt - DT0, srs, (g & 1) ^ 1);
                const int hp = tt & 1;
                const f16x8 b1 = __builtin_bit_cast(f16x8, bq[tt % 8][0]);
                const f16x8 b2 = __builtin_bit_cast(f16x8, bq[tt % 8][1]);
                f16x8 a1[TPW], a2[TPW];
#pragma unroll
                for (int m = 0; m < TPW; ++m) { a1[m] = __builtin_bit_cast(f16x8, af[cur][m][0]); a2[m] = __builtin_bit_cast(f16x8, af[cur][m][1]); }
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[g & 1][m][hp] = mfma16(a2[m], b1, tt < 2 ? zero16 : acc[g & 1][m][hp]);
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[g & 1][m][hp] = mfma16(a1[m], b2, acc[g & 1][m][hp]);
#pragma unroll
                for (int m = 0; m < TPW; ++m) acc[g & 1][m][hp] = mfma16(a1[m], b1, acc[g & 1][m][hp]);
                if (!(DBG & 2)) {
                    if (g == 0) {
                        // the previous pass's tail (its positions 0 and 7 are in bank 1, which this pass first writes in unit 1)
                        if (tt >= 2 && tt < 34 && !(tt & 1)) tail(ors_prev, (tt - 2) / 16, ((tt - 2) / 2) % 8);
                    } else if (tt >= 2 && tt < 34) {
                        fold(g - 1, (tt - 2) / 16, (tt - 2) % 16);
                    }
                    // the next pass's A offsets, in place: kernel row ky was last read for tap 31 + 2 ky of this unit, one tap ahead
                    if (last_u && have_next && (tt == 28 || tt == 30)) a_base(next_pass, qmin_n, (tt - 28) / 2, an_in[(tt - 28) / 2], an_z[(tt - 28) / 2], an_y[(tt - 28) / 2]);
                    if (last_u && have_next && (tt == 39 || (tt >= 32 && tt < 39 && !(tt & 1)))) {
                        const int ky = tt == 39 ? 4 : (tt - 32) / 2;
#pragma unroll
                        for (int m = 0; m < TPW; ++m) aoff[m][ky] = a_offset(an_in[m], an_z[m], an_y[m], ky);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * TPW, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 10, 0);
#pragma unroll
                for (int k = 0; k < 3 * TPW; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x246, 8, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // this wave's part of the next unit has landed: loads return in order, so everything older than the 2 x BD weight fragments in
            // flight is complete -- the DMA instructions were issued before them (stores in between only make the wait stricter)
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DBG & 4) ? 0 : 2 * BD) : "memory");
            __syncthreads();
        }
        ors_prev = ors;
        if (!have_next) break;
        pass = next_pass; qmin = qmin_n; nrows = nrows_n;
    }
    if (!(DBG & 2)) {          // the last pass's tail has no next pass to run under
        // (fold of unit 2 ran under unit 3; unit 3's pair is positions 0 and 7, taken by the tail itself)
#pragma unroll
        for (int b = 0; b < 16; ++b) tail(ors_prev, b / 8, b % 8);
    }
#undef P3_DMA
#undef P3_PIN
}
