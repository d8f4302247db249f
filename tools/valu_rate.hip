// Issue cost of the vector instructions the dropout epilogue is made of (gfx950): cycles per wave-instruction with 1, 2
// and 4 waves on a SIMD.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define KERNEL(NAME, ASM)                                                                        \
    __global__ void NAME(unsigned long long* out, unsigned a, unsigned b) {                      \
        unsigned v0 = threadIdx.x + a, v1 = b, v2 = a * 3, v3 = b + 7;                           \
        unsigned long long w = ((unsigned long long)a << 32) | b;                                \
        float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 2.f;                                        \
        (void)w; (void)f0; (void)f1; (void)f2; (void)f3;                                         \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        for (int i = 0; i < 64; ++i) { REP16(ASM) }                                              \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                         \
        if (v0 + v1 + v2 + v3 + (unsigned)w + (unsigned)f0 + (unsigned)f1 + (unsigned)f2 + (unsigned)f3 == 0x12345) out[0] = 1;     \
    }
KERNEL(k_add_f32, asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3" : "+v"(f0), "+v"(f2) : "v"(f1), "v"(f3));)
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3" : "+v"(f0), "+v"(f2) : "v"(f1), "v"(f3));)
KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0\n v_mad_u64_u32 %0, vcc, %3, %2, 0" : "+v"(w) : "v"(v0), "v"(v1), "v"(v2));)
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(v0), "+v"(v2) : "v"(v1), "v"(v3));)
KERNEL(k_mulhi, asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3" : "+v"(v0), "+v"(v2) : "v"(v1), "v"(v3));)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3" : "+v"(v0), "+v"(v2) : "v"(v1), "v"(v3));)
KERNEL(k_cmp_sdwa_cnd, asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:DWORD\n v_cndmask_b32 %2, 0, %3, vcc" : "+v"(v0), "+v"(v1), "+v"(v2) : "v"(v3) : "vcc");)
KERNEL(k_cvt_pk, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %2, %1" : "+v"(v0) : "v"(f1), "v"(f3), "v"(v2));)
KERNEL(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1" : "+v"(w) : "v"(w));)
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3" : "+v"(f0), "+v"(f2) : "v"(f1), "v"(f3));)
KERNEL(k_lshl_and, asm volatile("v_lshlrev_b32 %0, 16, %1\n v_and_b32 %2, 0xffff0000, %1" : "+v"(v0), "+v"(v1), "+v"(v2));)

template <typename K> void run(const char* name, K k) {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    for (int wps : {1, 2, 4}) {                       // waves per SIMD: blocks of 256 threads (1 wave per SIMD each), wps blocks per CU
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, 3u, 5u);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(256 * wps), dim3(256), 0, 0, d, 3u, 5u);
        hipDeviceSynchronize();
        unsigned long long h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 64; ++i) s += (double)h[i];
        printf("%-16s %d wave(s)/SIMD: %6.2f cycles per instruction per wave (%5.2f per SIMD issue)\n", name, wps, s / 64 / (64 * 16 * 2), s / 64 / (64 * 16 * 2) / wps);
    }
    hipFree(d);
}
int main() {
    run("v_add_f32", k_add_f32); run("v_max_f32", k_max_f32); run("v_mul_f32", k_mul_f32); run("v_xor_b32", k_xor);
    run("v_lshl/v_and", k_lshl_and); run("v_mad_u64_u32", k_mad64); run("v_mul_lo_u32", k_mullo); run("v_mul_hi_u32", k_mulhi);
    run("cmp_sdwa+cndmask", k_cmp_sdwa_cnd); run("v_cvt_pk_bf16", k_cvt_pk); run("v_pk_mul_f32", k_pk_mul);
    return 0;
}
