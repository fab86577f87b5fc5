// dev: what ONE SIMD of gfx950 overlaps.  A 512-thread workgroup per CU puts two waves on every SIMD (waves w and w + 4 -- checked through HW_ID);
// waves 0..3 run role A, waves 4..7 role B, each a fixed amount of work between two s_memtime reads.  Printed per configuration: cycles per
// iteration of each role alone and side by side.  Every instruction of the timed loops is written as volatile asm (nothing moves, nothing is dropped).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_simd.hip -o tools/ubench_simd.bin && tools/ubench_simd.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA32(acc_, a_, b_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define MFMA16(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define MFMA4(acc_, a_, b_) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define FMA(x_, a_, b_) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x_) : "v"(a_), "v"(b_))
#define PKFMA(x_, a_, b_) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x_) : "v"(a_), "v"(b_))
#define DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

struct Params { int modeA, modeB, prioA, prioB, iters; };

__device__ __forceinline__ void setprio(int p) {
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
}

// role bodies: return a value that depends on everything computed
// M1: conv2's tap loop shape: 40 taps, three DEPENDENT 32x32x16 products per tap on accumulator tap % 8
template <int FILL>      // FILL independent v_fma_f32 behind every MFMA (same wave)
__device__ float body_mfma_dep(int iters, float seed) {
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = seed;
    f16x8 a1, a2, b1, b2;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = (_Float16)(seed * 0.001f); a2[k] = (_Float16)(seed * 0.002f); b1[k] = (_Float16)0.5f; b2[k] = (_Float16)0.25f; }
    float f[8] = {seed, seed + 1, seed + 2, seed + 3, seed + 4, seed + 5, seed + 6, seed + 7};
    const float ca = 0.999f, cb = 0.001f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tau = 0; tau < 40; ++tau) {
            const int p = tau % 8;
            MFMA32(acc[p], a2, b1);
#pragma unroll
            for (int k = 0; k < FILL; ++k) FMA(f[k % 8], ca, cb);
            MFMA32(acc[p], a1, b2);
#pragma unroll
            for (int k = 0; k < FILL; ++k) FMA(f[(k + 3) % 8], ca, cb);
            MFMA32(acc[p], a1, b1);
#pragma unroll
            for (int k = 0; k < FILL; ++k) FMA(f[(k + 5) % 8], ca, cb);
        }
    }
    DRAIN();
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][15];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k];
    return s;
}
// M2: the same 120 products per iteration, but consecutive MFMAs go to DIFFERENT accumulators (two positions interleaved)
__device__ float body_mfma_indep(int iters, float seed) {
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = seed;
    f16x8 a1, a2, b1, b2;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = (_Float16)(seed * 0.001f); a2[k] = (_Float16)(seed * 0.002f); b1[k] = (_Float16)0.5f; b2[k] = (_Float16)0.25f; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tau = 0; tau < 20; ++tau) {
            const int p = (2 * tau) % 8, q = p + 1;
            MFMA32(acc[p], a2, b1); MFMA32(acc[q], a2, b1);
            MFMA32(acc[p], a1, b2); MFMA32(acc[q], a1, b2);
            MFMA32(acc[p], a1, b1); MFMA32(acc[q], a1, b1);
        }
    }
    DRAIN();
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) s += acc[p][0] + acc[p][15];
    return s;
}
// V: 960 v_fma_f32 per iteration in CH independent chains
template <int CH>
__device__ float body_valu(int iters, float seed) {
    float f[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) f[k] = seed + k;
    const float ca = 0.999f, cb = 0.001f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 960; ++i) FMA(f[i % CH], ca, cb);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) s += f[k];
    return s;
}
// PK: 480 v_pk_fma_f32 per iteration (the same 960 x 64 fmas), 8 chains
__device__ float body_pk(int iters, float seed) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { f[k][0] = seed + k; f[k][1] = seed - k; }
    f32x2 ca = {0.999f, 0.998f}, cb = {0.001f, 0.002f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 480; ++i) PKFMA(f[i % 8], ca, cb);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k][0] + f[k][1];
    return s;
}
// T: the output transform A^T as 4x4x1 16-block fp32 MFMAs: 16 independent chains of 8 products (one chain per accumulator register of a tile)
__device__ float body_mfma4(int iters, float seed) {
    f32x4 y[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] = f32x4{seed, seed, seed, seed};
    float a = 0.5f, m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = seed * 0.01f + k;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) MFMA4(y[r], a, m[p]);       // 256 per iteration, consecutive ones independent
    }
    DRAIN();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += y[k][0] + y[k][3];
    return s;
}
// M16: 240 16x16x32 products per iteration (conv1's shape: chains of four on one accumulator)
__device__ float body_mfma16(int iters, float seed) {
    f32x4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = f32x4{seed, seed, seed, seed};
    f16x8 a1, b1;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a1[k] = (_Float16)(seed * 0.001f); b1[k] = (_Float16)0.5f; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 60; ++i) { MFMA16(acc[i % 4], a1, b1); MFMA16(acc[i % 4], a1, b1); MFMA16(acc[i % 4], a1, b1); MFMA16(acc[i % 4], a1, b1); }
    }
    DRAIN();
    return acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
// L: LDS traffic of a transform phase: per iteration 64 x (ds_read_b128 x 2, 12 v_fma, ds_write_b64), waits as the compiler would place them
__device__ float body_lds(int iters, float seed, float* lds) {
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = seed + k;
    const float ca = 0.999f, cb = 0.001f;
    const unsigned base = (unsigned)(uintptr_t)lds + (threadIdx.x & 255) * 16;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            f32x4 d0, d1;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d0), "=&v"(d1) : "v"(base));
#pragma unroll
            for (int k = 0; k < 12; ++k) FMA(f[k % 8], ca, cb);
            f[0] += d0[0] + d1[3];
            { typedef float f32x2 __attribute__((ext_vector_type(2))); const f32x2 w2 = {f[0], f[1]}; asm volatile("ds_write_b64 %0, %1 offset:8192" :: "v"(base), "v"(w2) : "memory"); }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += f[k];
    return s;
}

__device__ float run_mode(int mode, int iters, float seed, float* lds) {
    switch (mode) {
        case 1: return body_mfma_dep<0>(iters, seed);
        case 2: return body_mfma_indep(iters, seed);
        case 3: return body_valu<8>(iters, seed);
        case 4: return body_valu<1>(iters, seed);
        case 5: return body_valu<2>(iters, seed);
        case 6: return body_pk(iters, seed);
        case 7: return body_mfma4(iters, seed);
        case 8: return body_mfma16(iters, seed);
        case 9: return body_lds(iters, seed, lds);
        case 10: return body_mfma_dep<1>(iters, seed);
        case 11: return body_mfma_dep<2>(iters, seed);
        case 12: return body_mfma_dep<3>(iters, seed);
        case 13: return body_mfma_dep<4>(iters, seed);
        case 14: return body_mfma_dep<5>(iters, seed);
        case 15: return body_mfma_dep<6>(iters, seed);
        case 16: return body_mfma_dep<7>(iters, seed);
        case 17: return body_valu<4>(iters, seed);
        default: return 0.f;
    }
}

__global__ __launch_bounds__(512) void k_bench(Params p, unsigned long long* out /*[grid][8][4]*/, float* sink) {
    __shared__ float lds[8192];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.5f;
    __syncthreads();
    const int role_b = wave >= 4;
    const int mode = role_b ? p.modeB : p.modeA;
    setprio(role_b ? p.prioB : p.prioA);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const float s = run_mode(mode, p.iters, (float)(threadIdx.x & 7) + 1.f, lds);
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_setprio(0);
    if ((threadIdx.x & 63) == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 4;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = hwid; o[3] = mode;
    }
    if (s == 12345.678f) sink[0] = s;
}

static const char* NAME[] = {"-", "mfma32 dep", "mfma32 indep", "valu x8", "valu x1", "valu x2", "pk_fma x8", "mfma4x4", "mfma16", "lds+valu", "mfma32+1v", "mfma32+2v", "mfma32+3v", "mfma32+4v", "mfma32+5v", "mfma32+6v", "mfma32+7v", "valu x4"};

int main() {
    const int grid = 256, iters = 200;
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, grid * 8 * 4 * 8); hipMalloc(&d_sink, 4);
    std::vector<unsigned long long> h(grid * 8 * 4);
    auto run = [&](int a, int b, int pa, int pb) {
        Params p{a, b, pa, pb, iters};
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_bench, dim3(grid), dim3(512), 0, 0, p, d_out, d_sink); }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        double ca = 0, cb = 0, ra = 0, rb = 0;
        for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) {
            const unsigned long long* o = &h[((size_t)g * 8 + w) * 4];
            if (w < 4) { ca += (double)o[0]; ra += (double)o[1]; } else { cb += (double)o[0]; rb += (double)o[1]; }
        }
        ca /= grid * 4.0 * iters; cb /= grid * 4.0 * iters; ra /= grid * 4.0 * iters; rb /= grid * 4.0 * iters;
        printf("A %-13s prio %d | B %-13s prio %d : A %8.1f cyc/iter (%6.2f us@100MHz-ticks %7.1f) | B %8.1f cyc/iter (ticks %7.1f)\n", NAME[a], pa, NAME[b], pb, ca, ra / 100.0, ra, cb, rb);
    };
    // which waves share a SIMD?
    { Params p{3, 3, 0, 0, 1}; hipLaunchKernelGGL(k_bench, dim3(grid), dim3(512), 0, 0, p, d_out, d_sink); hipDeviceSynchronize();
      hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      for (int g = 0; g < 2; ++g) { printf("wg %d HW_ID simd/wave-slot of waves 0..7:", g); for (int w = 0; w < 8; ++w) { unsigned id = (unsigned)h[((size_t)g * 8 + w) * 4 + 2]; printf(" s%u.w%u(cu%u)", (id >> 4) & 3, id & 15, (id >> 8) & 15); } printf("\n"); } }
    printf("# alone\n");
    for (int m : {1, 2, 3, 4, 5, 17, 6, 7, 8, 9}) run(m, 0, 0, 0);
    printf("# fillers inside one wave (120 mfma + 120 k valu per iteration)\n");
    for (int m : {10, 11, 12, 13, 14, 15, 16}) run(m, 0, 0, 0);
    printf("# two waves of one SIMD\n");
    run(1, 1, 0, 0); run(2, 2, 0, 0); run(3, 3, 0, 0); run(4, 4, 0, 0); run(5, 5, 0, 0); run(9, 9, 0, 0);
    for (int b : {3, 4, 5, 17, 6, 9, 7, 8}) { run(1, b, 0, 0); run(1, b, 3, 0); run(1, b, 0, 3); }
    for (int b : {3, 4, 9}) { run(2, b, 0, 0); run(2, b, 3, 0); }
    for (int a : {12, 14}) for (int b : {3, 4}) run(a, b, 0, 0);
    run(7, 3, 0, 0); run(8, 3, 0, 0); run(7, 1, 0, 0);
    return 0;
}
