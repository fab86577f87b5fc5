// dev microbenchmark: how fast can one wave-per-row streaming pass read frame+background?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// variant 0: plain read, sum reduce (frame only)
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, size_t n16, uint32_t* out) {
    size_t i = blockIdx.x * 256ull + threadIdx.x; const size_t stride = (size_t)gridDim.x * 256ull;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// variant 1: frame + bg, sad reject, per row task (grid-stride over rows), no run extraction
template<int NCH, int ORDER, int PREFETCH>
__global__ __launch_bounds__(256) void k_sad(const uint8_t* __restrict__ frames, const uint8_t* __restrict__ bg, int W, int H, int B, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t ntask = (uint32_t)B * H, nwave = gridDim.x * 4u;
    uint32_t task = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t cnt = 0;
    for (; task < ntask; task += nwave) {
        const uint32_t f = ORDER == 0 ? task % (uint32_t)B : task / (uint32_t)H;
        const uint32_t y = ORDER == 0 ? task / (uint32_t)B : task % (uint32_t)H;
        const uint8_t* fp = frames + ((size_t)f * H + y) * W;
        const uint8_t* bp = bg + (size_t)y * W;
        uint4 a[NCH], b[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) { const int x = ch * 1024 + lane * 16; a[ch] = *(const uint4*)(fp + x); b[ch] = *(const uint4*)(bp + x); }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            uint32_t s0 = __builtin_amdgcn_sad_u8(a[ch].x, b[ch].x, 0u), s1 = __builtin_amdgcn_sad_u8(a[ch].y, b[ch].y, 0u);
            uint32_t s2 = __builtin_amdgcn_sad_u8(a[ch].z, b[ch].z, 0u), s3 = __builtin_amdgcn_sad_u8(a[ch].w, b[ch].w, 0u);
            cnt += (max(max(s0, s1), max(s2, s3)) >= 16u);
        }
    }
    if (__any(cnt == 0xffffffffu)) out[0] = cnt;
}
int main(int argc, char** argv) {
    const int W = 2048, H = 2048, B = 64;
    uint8_t *fr, *bg; uint32_t* out;
    CK(hipMalloc(&fr, (size_t)B * W * H)); CK(hipMalloc(&bg, (size_t)W * H)); CK(hipMalloc(&out, 64));
    CK(hipMemset(fr, 120, (size_t)B * W * H)); CK(hipMemset(bg, 121, (size_t)W * H));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-40s %8.1f us  %8.1f GB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
    };
    const size_t n16 = (size_t)B * W * H / 16;
    for (int blocks : {1024, 2048, 4096, 16384}) {
        char nm[64]; snprintf(nm, 64, "read frames only, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const uint4*)fr, n16, out); }, (double)B * W * H);
    }
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "sad order0 (frame fastest), %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((k_sad<2, 0, 0>), dim3(blocks), dim3(256), 0, 0, fr, bg, W, H, B, out); }, 2.0 * B * W * H);
        snprintf(nm, 64, "sad order1 (row fastest), %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL((k_sad<2, 1, 0>), dim3(blocks), dim3(256), 0, 0, fr, bg, W, H, B, out); }, 2.0 * B * W * H);
    }
    return 0;
}
