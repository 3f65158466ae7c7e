// Streaming-copy rates for the access shapes the pyramid kernels use: per lane a 12-byte load at offset -4 and a
// 4-byte store (blur / resize), an 8-byte or a 16-byte load and store (gray).  Rows are walked like the blur does:
// a lane keeps its column and steps down `rows` rows of pitch `pitch`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W>  // bytes per lane and row
__global__ __launch_bounds__(256) void k_copy(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                              int pitch, int rows_per_strip, int strips_x) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int strip = blockIdx.x * 4 + wv;
    const int sx = strip % strips_x, sy = strip / strips_x;
    const size_t frame = (size_t)blockIdx.y * pitch * 512;
    const unsigned col = sx * 64 * W + lane * W;
    const unsigned row0 = sy * rows_per_strip;
    const unsigned char* s = src + frame;
    unsigned char* d = dst + frame;
#pragma unroll 8
    for (int r = 0; r < rows_per_strip; r++) {
        const unsigned o = (row0 + r) * pitch + col;
        if (W == 4) {
            struct R3 { unsigned a, b, c; } v;
            __builtin_memcpy(&v, s + o + 60, 12);   // the dword and its two neighbours (offset keeps it in range)
            *(unsigned*)(d + o + 64) = v.a + v.b + v.c;
        } else if (W == 8) {
            uint2 v = *(const uint2*)(s + o + 64);
            *(uint2*)(d + o + 64) = v;
        } else {
            uint4 v = *(const uint4*)(s + o + 64);
            *(uint4*)(d + o + 64) = v;
        }
    }
}

template <int W>
int run(const unsigned char* src, unsigned char* dst, int frames) {
    const int pitch = 2048, rows = 512, rps = 32;
    const int strips_x = (pitch - 128) / (64 * W), strips = strips_x * (rows / rps);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy<W>, dim3(strips / 4, frames), dim3(256), 0, 0, src, dst, pitch, rps, strips_x);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 2.0 * (double)strips_x * 64 * W * rows * frames;
        if (rep == 2) printf("%2d bytes per lane: %.3f ms, %.2f TB/s (read + write)\n", W, ms, bytes / ms / 1e9);
    }
    return 0;
}

int main() {
    const int frames = 792;
    const size_t n = (size_t)frames * 2048 * 512 + 4096;
    unsigned char *src, *dst;
    CK(hipMalloc(&src, n)); CK(hipMalloc(&dst, n));
    CK(hipMemset(src, 1, n)); CK(hipMemset(dst, 0, n));
    if (run<4>(src, dst, frames) || run<8>(src, dst, frames) || run<16>(src, dst, frames)) return 1;
    return 0;
}
