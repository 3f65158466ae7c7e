// Issue rate of the double-precision instructions the fundamental-matrix scoring is made of, for 1, 2, 4 and 8
// wavefronts per SIMD (four independent dependency chains per lane).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
#define KERNEL(NAME, TEXT)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(double* out, double c, float cf, int iters) {                     \
        double x0 = threadIdx.x + 1.5, x1 = 1.25, x2 = 2.5, x3 = 3.75;                                            \
        for (int i = 0; i < iters; i++) { R16(asm volatile(TEXT : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c), "v"(cf) : "vcc");) } \
        if (x0 + x1 + x2 + x3 == 0.12345) out[0] = x0;                                                            \
    }
KERNEL(k_add, "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4")
KERNEL(k_mul, "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4")
KERNEL(k_fma, "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4")
KERNEL(k_rcp, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
KERNEL(k_div_scale, "v_div_scale_f64 %0, vcc, %0, %4, %0\n v_div_scale_f64 %1, vcc, %1, %4, %1\n v_div_scale_f64 %2, vcc, %2, %4, %2\n v_div_scale_f64 %3, vcc, %3, %4, %3")
KERNEL(k_div_fmas, "v_div_fmas_f64 %0, %0, %4, %4\n v_div_fmas_f64 %1, %1, %4, %4\n v_div_fmas_f64 %2, %2, %4, %4\n v_div_fmas_f64 %3, %3, %4, %4")
KERNEL(k_div_fixup, "v_div_fixup_f64 %0, %0, %4, %4\n v_div_fixup_f64 %1, %1, %4, %4\n v_div_fixup_f64 %2, %2, %4, %4\n v_div_fixup_f64 %3, %3, %4, %4")
KERNEL(k_cvt, "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %5\n v_cvt_f64_f32 %3, %5")
KERNEL(k_cmp, "v_cmp_lt_f64 vcc, %0, %4\n v_cmp_lt_f64 vcc, %1, %4\n v_cmp_lt_f64 vcc, %2, %4\n v_cmp_lt_f64 vcc, %3, %4")
KERNEL(k_add1, "v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %0, %0, %4")
KERNEL(k_mul1, "v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %0, %0, %4")
template <typename K> void run(const char* name, K kern, double* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-22s", name);
    for (int waves = 1; waves <= 8; waves *= 2) {   // wavefronts per SIMD
        const int blocks = 256 * waves, iters = 512;
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1.5f, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double instr_per_simd = (double)waves * iters * 64;
        printf("  %d w/SIMD: %5.2f", waves, best * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("   cycles per wave-instruction per SIMD at 2.4 GHz\n");
}
int main() { double* d; (void)hipMalloc(&d, 64);
    run("add_f64", k_add, d); run("mul_f64", k_mul, d); run("fma_f64", k_fma, d); run("rcp_f64", k_rcp, d);
    run("div_scale_f64", k_div_scale, d); run("div_fmas_f64", k_div_fmas, d); run("div_fixup_f64", k_div_fixup, d);
    run("cvt_f64_f32", k_cvt, d); run("cmp_lt_f64", k_cmp, d);
    run("add_f64 one chain", k_add1, d); run("mul_f64 one chain", k_mul1, d);
    return 0; }
