// VALU issue rate on gfx950: N independent v_add_u32 per wave, 1..8 waves per SIMD, loop vs straight-line.
#include <hip/hip_runtime.h>
#include <cstdio>
#define A4 asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c));
#define A16 A4 A4 A4 A4
#define A64 A16 A16 A16 A16
#define A256 A64 A64 A64 A64
#define A1024 A256 A256 A256 A256
__global__ __launch_bounds__(256) void k_loop(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { A64 }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_straight(unsigned* out, unsigned c) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    A1024 A1024 A1024 A1024 A1024 A1024 A1024 A1024
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
        for (int mode = 0; mode < 2; mode++) {
            const int blocks = 256 * wg_per_cu;   // 4 waves each: wg_per_cu waves per SIMD
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(256), 0, 0, d, 5u, 128);
                else hipLaunchKernelGGL(k_straight, dim3(blocks), dim3(256), 0, 0, d, 5u);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = 8192.0 * wg_per_cu;   // 8192 VALU per wave
            printf("%s, %d waves/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
                   mode ? "straight-line (64 KB of code)" : "loop of 64", wg_per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    }
    return 0;
}
