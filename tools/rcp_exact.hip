// Exhaustive check over all 2^32 float bit patterns: which cheap sequences reproduce the correctly rounded 1.0f / x
// that the compiler's IEEE division gives?  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float cand_a(float x) {   // one correction step
    const float r = rcp_hw(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ float cand_b(float x) {   // two correction steps
    float r = rcp_hw(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(r, e, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ float cand_c(float x) {   // one step + the hardware's special-case fix-up
    return __builtin_amdgcn_div_fixupf(cand_a(x), x, 1.0f);
}
struct Stat { unsigned long long bad; unsigned min_abs, max_abs; };
__global__ void k_check(Stat* st, unsigned lo_bits, unsigned hi_bits) {
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned bits = (unsigned)i;
        const unsigned ab = bits & 0x7fffffffu;
        if (ab < lo_bits || ab > hi_bits) continue;   // the guarded range only
        const float x = __uint_as_float(bits);
        const float ref = 1.0f / x;
        const float c[3] = {cand_a(x), cand_b(x), cand_c(x)};
        for (int k = 0; k < 3; k++) {
            const bool same = __float_as_uint(c[k]) == __float_as_uint(ref) || (c[k] != c[k] && ref != ref);
            if (!same) {
                atomicAdd(&st[k].bad, 1ull);
                atomicMin(&st[k].min_abs, ab);
                atomicMax(&st[k].max_abs, ab);
            }
        }
    }
}
int main() {
    Stat* d; (void)hipMalloc(&d, 3 * sizeof(Stat));
    const unsigned ranges[3][2] = {{0u, 0x7fffffffu}, {0x00800000u, 0x7f7fffffu}, {0x03800000u, 0x7b800000u}};
    const char* names[3] = {"all bit patterns", "normal numbers", "2^-120 <= |x| <= 2^120"};
    for (int r = 0; r < 3; r++) {
        Stat h[3] = {{0, 0xffffffffu, 0}, {0, 0xffffffffu, 0}, {0, 0xffffffffu, 0}};
        (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, ranges[r][0], ranges[r][1]);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const char* cn[3] = {"rcp + 1 step", "rcp + 2 steps", "rcp + 1 step + div_fixup"};
        for (int k = 0; k < 3; k++)
            printf("%-24s %-26s mismatches %llu  (|x| bits %08x .. %08x)\n", names[r], cn[k], h[k].bad, h[k].min_abs, h[k].max_abs);
    }
    return 0;
}
