// How fast can ONE wave issue vector instructions on gfx950, as a function of the independent chains it interleaves?
// hipcc --offload-arch=gfx950 -O3 tools/valu_ilp.hip -o /tmp/valu_ilp && /tmp/valu_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int CH> __global__ void k_chain(unsigned long long* out, float a, float b) {
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = a + i + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if constexpr (CH == 1) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(b));) }
        if constexpr (CH == 2) { REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(f[0]), "+v"(f[1]) : "v"(b));) }
        if constexpr (CH == 4) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(b));) }
        if constexpr (CH == 8) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(b));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 0x12345) out[0] = 1;
}
template <int CH> void run() {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    for (int wps : {1, 2, 4}) {
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k_chain<CH>, dim3(256 * wps), dim3(256), 0, 0, d, 3.f, 5.f); hipDeviceSynchronize(); }
        unsigned long long h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 64; ++i) s += (double)h[i];
        const double per = s / 64 / (64.0 * 16 * CH);
        printf("v_add_f32 x %d chain(s), %d wave(s)/SIMD: %6.2f ticks per instruction per wave (%5.2f per SIMD issue)\n", CH, wps, per, per / wps);
    }
    hipFree(d);
}
__global__ void k_clock(unsigned long long* out) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    unsigned long long c0 = clock64();
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = c1 - c0; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("s_memtime ticks %llu, wall_clock64 ticks (100 MHz) %llu, clock64 %llu -> s_memtime runs at %.1f MHz\n", h[0], h[1], h[2], 100.0 * h[0] / h[1]);
    run<1>(); run<2>(); run<4>(); run<8>();
    return 0;
}
