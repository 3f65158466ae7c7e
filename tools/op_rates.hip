#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(X) X X X X
#define R16(X) R4(X) R4(X) R4(X) R4(X)
__global__ __launch_bounds__(256) void k_min_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_max_i32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_sub_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_and(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_or(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_lshlrev(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_lshrrev(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_mov(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_alignbyte(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_alignbyte_b32 %0, %0, %4, 1\n v_alignbyte_b32 %1, %1, %4, 1\n v_alignbyte_b32 %2, %2, %4, 1\n v_alignbyte_b32 %3, %3, %4, 1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_bfe_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_and_or(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_lshl_or(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_lshl_or_b32 %0, %0, 8, %4\n v_lshl_or_b32 %1, %1, 8, %4\n v_lshl_or_b32 %2, %2, 8, %4\n v_lshl_or_b32 %3, %3, 8, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_lshl_add(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_min_u16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_min_u16 %0, %0, %4\n v_min_u16 %1, %1, %4\n v_min_u16 %2, %2, %4\n v_min_u16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_max_i16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_max_i16 %0, %0, %4\n v_max_i16 %1, %1, %4\n v_max_i16 %2, %2, %4\n v_max_i16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_add_u16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_add_u16 %0, %0, %4\n v_add_u16 %1, %1, %4\n v_add_u16 %2, %2, %4\n v_add_u16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_sub_u16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_sub_u16 %0, %0, %4\n v_sub_u16 %1, %1, %4\n v_sub_u16 %2, %2, %4\n v_sub_u16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_pk_sub_i16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_pk_max_i16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_pk_add_u16(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_min3_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_min3_u32 %0, %0, %4, %4\n v_min3_u32 %1, %1, %4, %4\n v_min3_u32 %2, %2, %4, %4\n v_min3_u32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_med3_i32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_med3_i32 %0, %0, %4, %4\n v_med3_i32 %1, %1, %4, %4\n v_med3_i32 %2, %2, %4, %4\n v_med3_i32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_cndmask(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_cmp_gt(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_mad_u32_u24(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mad_u32_u24 %0, %0, %4, %4\n v_mad_u32_u24 %1, %1, %4, %4\n v_mad_u32_u24 %2, %2, %4, %4\n v_mad_u32_u24 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_mul_lo_u32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_fma_f32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_add_f32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_mul_f32(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_mbcnt_lo(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_lo_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_lo_u32_b32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_sad_u8_sdwa_min(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_min_u16_sdwa %0, %0, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_min_u16_sdwa %1, %1, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_min_u16_sdwa %2, %2, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n v_min_u16_sdwa %3, %3, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_add_sdwa(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}
__global__ __launch_bounds__(256) void k_min_u32_dpp(unsigned* out, unsigned c, int iters) {
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_min_u32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2, %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) }
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0;
}

// Three-operand and packed floating-point minima / maxima (pixel values are exact in f16 and f32).
#define RATE_KERNEL3(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned c, int iters) { \
    unsigned x0 = threadIdx.x | 0x64006400u, x1 = 0x64016401u, x2 = 0x64026402u, x3 = 0x64036403u; \
    for (int i = 0; i < iters; i++) { R16(asm volatile(OP " %0, %0, %4, %4\n " OP " %1, %1, %4, %4\n " OP " %2, %2, %4, %4\n " OP " %3, %3, %4, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) } \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0; \
}
#define RATE_KERNEL2(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned c, int iters) { \
    unsigned x0 = threadIdx.x | 0x64006400u, x1 = 0x64016401u, x2 = 0x64026402u, x3 = 0x64036403u; \
    for (int i = 0; i < iters; i++) { R16(asm volatile(OP " %0, %0, %4\n " OP " %1, %1, %4\n " OP " %2, %2, %4\n " OP " %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) } \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0; \
}
#define RATE_KERNEL1(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned c, int iters) { \
    unsigned x0 = threadIdx.x | 0x64006400u, x1 = 0x64016401u, x2 = 0x64026402u, x3 = 0x64036403u; \
    for (int i = 0; i < iters; i++) { R16(asm volatile(OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) } \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0; \
}
RATE_KERNEL3(k_pk_minimum3_f16, "v_pk_minimum3_f16")
RATE_KERNEL3(k_pk_maximum3_f16, "v_pk_maximum3_f16")
RATE_KERNEL2(k_pk_min_f16, "v_pk_min_f16")
RATE_KERNEL2(k_pk_max_f16, "v_pk_max_f16")
RATE_KERNEL2(k_pk_add_f16, "v_pk_add_f16")
RATE_KERNEL3(k_min3_f16, "v_min3_f16")
RATE_KERNEL3(k_min3_f32, "v_min3_f32")
RATE_KERNEL3(k_max3_f32, "v_max3_f32")
RATE_KERNEL3(k_minimum3_f32, "v_minimum3_f32")
RATE_KERNEL3(k_min3_i16, "v_min3_i16")
RATE_KERNEL3(k_min3_u16, "v_min3_u16")
RATE_KERNEL3(k_max3_i16, "v_max3_i16")
RATE_KERNEL2(k_min_f32, "v_min_f32")
RATE_KERNEL2(k_max_f32, "v_max_f32")
RATE_KERNEL2(k_max_f16, "v_max_f16")
RATE_KERNEL1(k_cvt_f32_ubyte0, "v_cvt_f32_ubyte0")
RATE_KERNEL1(k_cvt_f32_ubyte3, "v_cvt_f32_ubyte3")
RATE_KERNEL2(k_pk_min_u16, "v_pk_min_u16")
RATE_KERNEL2(k_pk_sub_u16_clamp, "v_pk_sub_u16")
RATE_KERNEL3(k_perm, "v_perm_b32")
RATE_KERNEL3(k_bfi, "v_bfi_b32")
RATE_KERNEL3(k_or3, "v_or3_b32")
RATE_KERNEL3(k_xad, "v_xad_u32")
RATE_KERNEL3(k_add3, "v_add3_u32")
#define RATE_KERNEL_DPP(NAME, CTRL) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned c, int iters) { \
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3; \
    for (int i = 0; i < iters; i++) { R16(asm volatile("v_mov_b32_dpp %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(c) : "vcc");) } \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345u) out[0] = x0; \
}
RATE_KERNEL_DPP(k_mov_dpp_row_shr, "row_shr:1")
RATE_KERNEL_DPP(k_mov_dpp_wave_shr, "wave_shr:1")
RATE_KERNEL_DPP(k_mov_dpp_wave_shl, "wave_shl:1")
RATE_KERNEL_DPP(k_mov_dpp_quad, "quad_perm:[1,0,3,2]")
RATE_KERNEL_DPP(k_mov_dpp_bcast15, "row_bcast:15")
#define RATE_KERNEL_PK64(NAME, OP) \
__global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned c, int iters) { \
    unsigned long long x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, cc = ((unsigned long long)c << 32) | c; \
    for (int i = 0; i < iters; i++) { R16(asm volatile(OP " %0, %0, %4\n " OP " %1, %1, %4\n " OP " %2, %2, %4\n " OP " %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(cc) : "vcc");) } \
    if ((x0 ^ x1 ^ x2 ^ x3) == 0x12345ull) out[0] = (unsigned)x0; \
}
RATE_KERNEL_PK64(k_pk_mul_f32, "v_pk_mul_f32")
RATE_KERNEL_PK64(k_pk_add_f32, "v_pk_add_f32")
RATE_KERNEL1(k_rndne_f32, "v_rndne_f32")
RATE_KERNEL1(k_cvt_i32_f32, "v_cvt_i32_f32")
RATE_KERNEL1(k_cvt_f32_i32, "v_cvt_f32_i32")
template <typename K> void run(const char* name, K kern, unsigned* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 1024;
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 5u, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double instr_per_simd = 8.0 * iters * 64;
    printf("%-18s %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, best * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() { unsigned* d; (void)hipMalloc(&d, 64);
    run("min_u32", k_min_u32, d);
    run("max_i32", k_max_i32, d);
    run("sub_u32", k_sub_u32, d);
    run("and", k_and, d);
    run("or", k_or, d);
    run("lshlrev", k_lshlrev, d);
    run("lshrrev", k_lshrrev, d);
    run("mov", k_mov, d);
    run("alignbyte", k_alignbyte, d);
    run("bfe_u32", k_bfe_u32, d);
    run("and_or", k_and_or, d);
    run("lshl_or", k_lshl_or, d);
    run("lshl_add", k_lshl_add, d);
    run("min_u16", k_min_u16, d);
    run("max_i16", k_max_i16, d);
    run("add_u16", k_add_u16, d);
    run("sub_u16", k_sub_u16, d);
    run("pk_sub_i16", k_pk_sub_i16, d);
    run("pk_max_i16", k_pk_max_i16, d);
    run("pk_add_u16", k_pk_add_u16, d);
    run("min3_u32", k_min3_u32, d);
    run("med3_i32", k_med3_i32, d);
    run("cndmask", k_cndmask, d);
    run("cmp_gt", k_cmp_gt, d);
    run("mad_u32_u24", k_mad_u32_u24, d);
    run("mul_lo_u32", k_mul_lo_u32, d);
    run("fma_f32", k_fma_f32, d);
    run("add_f32", k_add_f32, d);
    run("mul_f32", k_mul_f32, d);
    run("mbcnt_lo", k_mbcnt_lo, d);
    run("sad_u8_sdwa_min", k_sad_u8_sdwa_min, d);
    run("add_sdwa", k_add_sdwa, d);
    run("min_u32_dpp", k_min_u32_dpp, d);
    run("pk_minimum3_f16", k_pk_minimum3_f16, d);
    run("pk_maximum3_f16", k_pk_maximum3_f16, d);
    run("pk_min_f16", k_pk_min_f16, d);
    run("pk_max_f16", k_pk_max_f16, d);
    run("pk_add_f16", k_pk_add_f16, d);
    run("min3_f16", k_min3_f16, d);
    run("min3_f32", k_min3_f32, d);
    run("max3_f32", k_max3_f32, d);
    run("minimum3_f32", k_minimum3_f32, d);
    run("min3_i16", k_min3_i16, d);
    run("min3_u16", k_min3_u16, d);
    run("max3_i16", k_max3_i16, d);
    run("min_f32", k_min_f32, d);
    run("max_f32", k_max_f32, d);
    run("max_f16", k_max_f16, d);
    run("cvt_f32_ubyte0", k_cvt_f32_ubyte0, d);
    run("cvt_f32_ubyte3", k_cvt_f32_ubyte3, d);
    run("pk_min_u16", k_pk_min_u16, d);
    run("pk_sub_u16", k_pk_sub_u16_clamp, d);
    run("perm", k_perm, d);
    run("bfi", k_bfi, d);
    run("or3", k_or3, d);
    run("xad", k_xad, d);
    run("add3", k_add3, d);
    run("mov_dpp row_shr:1", k_mov_dpp_row_shr, d);
    run("mov_dpp wave_shr:1", k_mov_dpp_wave_shr, d);
    run("mov_dpp wave_shl:1", k_mov_dpp_wave_shl, d);
    run("mov_dpp quad_perm", k_mov_dpp_quad, d);
    run("mov_dpp row_bcast:15", k_mov_dpp_bcast15, d);
    run("pk_mul_f32", k_pk_mul_f32, d);
    run("pk_add_f32", k_pk_add_f32, d);
    run("rndne_f32", k_rndne_f32, d);
    run("cvt_i32_f32", k_cvt_i32_f32, d);
    run("cvt_f32_i32", k_cvt_f32_i32, d);
    return 0; }