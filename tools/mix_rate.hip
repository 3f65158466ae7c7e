#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
// dependent, groups of 4: t = x ^ c ; a = bcnt(t) + a
__global__ __launch_bounds__(256) void k_dep4(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
        "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) out[0] = a0;
}
// independent, groups of 4: t = x ^ c ; a = bcnt(x) + a
__global__ __launch_bounds__(256) void k_ind4(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
        "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ t0 ^ t1 ^ t2 ^ t3) == 0x12345u) out[0] = a0;
}
// independent, no accumulate: a = bcnt(x) + 0 (no chains at all)
__global__ __launch_bounds__(256) void k_ind4n(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
        "v_bcnt_u32_b32 %0, %8, 0\n v_bcnt_u32_b32 %1, %9, 0\n v_bcnt_u32_b32 %2, %10, 0\n v_bcnt_u32_b32 %3, %11, 0"
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ t0 ^ t1 ^ t2 ^ t3) == 0x12345u) out[0] = a0;
}
// two half-rate classes mixed, independent: perm + bcnt
__global__ __launch_bounds__(256) void k_perm_bcnt(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_perm_b32 %4, %8, %12, %12\n v_perm_b32 %5, %9, %12, %12\n v_perm_b32 %6, %10, %12, %12\n v_perm_b32 %7, %11, %12, %12\n"
        "v_bcnt_u32_b32 %0, %8, 0\n v_bcnt_u32_b32 %1, %9, 0\n v_bcnt_u32_b32 %2, %10, 0\n v_bcnt_u32_b32 %3, %11, 0"
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ t0 ^ t1 ^ t2 ^ t3) == 0x12345u) out[0] = a0;
}
// two full-rate ops mixed: xor + add
__global__ __launch_bounds__(256) void k_xor_add(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
        "v_add_u32 %0, %8, %12\n v_add_u32 %1, %9, %12\n v_add_u32 %2, %10, %12\n v_add_u32 %3, %11, %12"
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ t0 ^ t1 ^ t2 ^ t3) == 0x12345u) out[0] = a0;
}
// dependent xor -> add (full-rate pair with RAW at distance 4)
__global__ __launch_bounds__(256) void k_xor_add_dep(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0=0,t1=0,t2=0,t3=0,a0=0,a1=0,a2=0,a3=0;
    for (int i = 0; i < iters; i++) { R4(R4(asm volatile(
        "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
        "v_add_u32 %0, %4, %0\n v_add_u32 %1, %5, %1\n v_add_u32 %2, %6, %2\n v_add_u32 %3, %7, %3"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));)) }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ t0 ^ t1 ^ t2 ^ t3) == 0x12345u) out[0] = a0;
}
template <typename K> void run(const char* name, K kern, unsigned* d, int wg_per_cu) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * wg_per_cu, iters = 1024;
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 5u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double instr_per_simd = (double)wg_per_cu * iters * 128;
    printf("%-22s %d waves/SIMD: %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wg_per_cu, best * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { unsigned* d; (void)hipMalloc(&d, 64);
    for (int w : {8, 2, 1}) {
    run("xor->bcnt dep4", k_dep4, d, w); run("xor|bcnt ind4", k_ind4, d, w); run("xor|bcnt ind4 noacc", k_ind4n, d, w);
    run("perm|bcnt", k_perm_bcnt, d, w); run("xor|add", k_xor_add, d, w); run("xor->add dep", k_xor_add_dep, d, w); }
    return 0; }
