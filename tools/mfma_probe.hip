// Probe: raw results of the gfx950 bf16 MFMA instructions on given operands, for
// offline analysis of their internal summation order (tools/mfma_probe.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

// A[p][16][32] bf16, Bt[p][16][32] bf16 (Bt[n][k]), Cm[p][16][16] fp32 -> D[p][16][16]
__global__ void probe16(const uint16_t* A, const uint16_t* Bt, const float* Cm, float* D) {
    const int p = blockIdx.x, l = threadIdx.x;
    union { uint4 u; bf16x8_t v; } a, b;
    a.u = *(const uint4*)(A + ((size_t)p * 16 + (l & 15)) * 32 + 8 * (l >> 4));
    b.u = *(const uint4*)(Bt + ((size_t)p * 16 + (l & 15)) * 32 + 8 * (l >> 4));
    f32x4_t c;
    for (int r = 0; r < 4; ++r) c[r] = Cm[((size_t)p * 16 + 4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((size_t)p * 16 + 4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
// A[p][32][16], Bt[p][32][16], Cm[p][32][32] -> D[p][32][32]
__global__ void probe32(const uint16_t* A, const uint16_t* Bt, const float* Cm, float* D) {
    const int p = blockIdx.x, l = threadIdx.x;
    union { uint4 u; bf16x8_t v; } a, b;
    a.u = *(const uint4*)(A + ((size_t)p * 32 + (l & 31)) * 16 + 8 * (l >> 5));
    b.u = *(const uint4*)(Bt + ((size_t)p * 32 + (l & 31)) * 16 + 8 * (l >> 5));
    f32x16_t c;
    for (int r = 0; r < 16; ++r) c[r] = Cm[((size_t)p * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((size_t)p * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
extern "C" int run_probe16(const void* A, const void* Bt, const void* C, void* D, int np) {
    hipLaunchKernelGGL(probe16, dim3(np), dim3(64), 0, 0, (const uint16_t*)A, (const uint16_t*)Bt, (const float*)C, (float*)D);
    return (int)hipDeviceSynchronize();
}
extern "C" int run_probe32(const void* A, const void* Bt, const void* C, void* D, int np) {
    hipLaunchKernelGGL(probe32, dim3(np), dim3(64), 0, 0, (const uint16_t*)A, (const uint16_t*)Bt, (const float*)C, (float*)D);
    return (int)hipDeviceSynchronize();
}
