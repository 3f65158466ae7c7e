// Issue-rate probes (no counterpart in the reference): how many wave-instructions per second this GPU sustains for
// the instruction mixes that bound the matching and scoring kernels, measured in the same process and at the same
// clocks as the kernels they are compared with.  Every lane runs four independent dependency chains of the probed
// instructions, eight wavefronts per SIMD, so the figure is the vector pipes' issue rate, not a latency.
#include "common.h"

using namespace orbx;

namespace {

#define ORBX_REP4(X) X X X X
#define ORBX_REP16(X) ORBX_REP4(X) ORBX_REP4(X) ORBX_REP4(X) ORBX_REP4(X)

// Instructions per loop trip: 64 in k_probe_add_u32 (16 x 4), 128 in the two-instruction mixes (16 x 8).
__global__ __launch_bounds__(256) void k_probe_add_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) {
        ORBX_REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c));)
    }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}

// the Hamming inner step: xor of a query word with a train word, population count added to an accumulator; the four
// accumulators are the only dependency chains (as in the matcher, where every xor has fresh operands)
__global__ __launch_bounds__(256) void k_probe_xor_bcnt(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, t0 = 0, t1 = 0, t2 = 0, t3 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int i = 0; i < iters; i++) {
        ORBX_REP4(ORBX_REP4(asm volatile(
            "v_xor_b32 %4, %8, %12\n v_xor_b32 %5, %9, %12\n v_xor_b32 %6, %10, %12\n v_xor_b32 %7, %11, %12\n"
            "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)
            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(c));))
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) out[0] = a0;
}

// un-fused multiply and add in double (the fundamental-matrix scorer's arithmetic)
__global__ __launch_bounds__(256) void k_probe_f64(double* out, double c, int iters) {
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) {
        ORBX_REP4(ORBX_REP4(asm volatile(
            "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
            "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c));))
    }
    if (x0 + x1 + x2 + x3 == 12345.0) out[0] = x0;
}

// the same in float (the homography scorer's arithmetic)
__global__ __launch_bounds__(256) void k_probe_f32(float* out, float c, int iters) {
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) {
        ORBX_REP4(ORBX_REP4(asm volatile(
            "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
            "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c));))
    }
    if (x0 + x1 + x2 + x3 == 12345.0f) out[0] = x0;
}

}  // namespace

extern "C" orbx_status orbx_probe_issue_rate(int device, int kind, double* wave_instructions_per_second) {
    if (!wave_instructions_per_second || kind < 0 || kind > 3) return ORBX_BAD_ARG;
    ORBX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ORBX_HIP(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8, iters = 2048;  // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
    DevBuf<double> d_out;
    ORBX_TRY(d_out.ensure(8));
    hipEvent_t e0, e1;
    ORBX_HIP(hipEventCreate(&e0));
    ORBX_HIP(hipEventCreate(&e1));
    float best_ms = 0.f;
    for (int rep = 0; rep < 3; rep++) {
        ORBX_HIP(hipEventRecord(e0, 0));
        switch (kind) {
            case 0: hipLaunchKernelGGL(k_probe_add_u32, dim3(blocks), dim3(256), 0, 0, (unsigned*)d_out.p, 5u, iters); break;
            case 1: hipLaunchKernelGGL(k_probe_xor_bcnt, dim3(blocks), dim3(256), 0, 0, (unsigned*)d_out.p, 5u, iters); break;
            case 2: hipLaunchKernelGGL(k_probe_f64, dim3(blocks), dim3(256), 0, 0, d_out.p, 1.0000001, iters); break;
            default: hipLaunchKernelGGL(k_probe_f32, dim3(blocks), dim3(256), 0, 0, (float*)d_out.p, 1.0000001f, iters); break;
        }
        ORBX_HIP(hipEventRecord(e1, 0));
        ORBX_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        ORBX_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0 || ms < best_ms) best_ms = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    d_out.release();
    const double wave_instr = (double)blocks * 4 * iters * (kind == 0 ? 64 : 128);
    *wave_instructions_per_second = wave_instr / (best_ms * 1e-3);
    return ORBX_OK;
}
