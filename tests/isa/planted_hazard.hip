// Test fixture of tests/test_isa_hazards.py, never part of the library: a matrix instruction whose destination is read by an inline-asm
// instruction right behind it.  hipcc does not pad inline asm (cdna_hip_programming.md 5.7), so the object carries the hazard the
// checker (tools/isa_hazards.py) must report.  -DPADDED puts the wait states inside the string: then it must stay silent.
#include <hip/hip_runtime.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
extern "C" __global__ void planted(const h8* a, const h8* b, float* out) {
    f16v acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    float v;
#ifdef PADDED
    asm volatile("s_nop 11\n\tv_max3_f32 %0, %1, %2, %3" : "=v"(v) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#else
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(v) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#endif
    out[threadIdx.x] = v;
}
