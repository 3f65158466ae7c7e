// Internal helpers shared by the orbx translation units (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "orbx.h"

namespace orbx {

void set_last_error(const char* what, hipError_t e, const char* file, int line);
void set_last_error_text(const char* text);

#define ORBX_HIP(expr)                                                   \
    do {                                                                 \
        hipError_t e__ = (expr);                                         \
        if (e__ != hipSuccess) {                                         \
            ::orbx::set_last_error(#expr, e__, __FILE__, __LINE__);      \
            return ORBX_DEVICE;                                          \
        }                                                                \
    } while (0)

#define ORBX_TRY(expr)                        \
    do {                                      \
        orbx_status s__ = (expr);             \
        if (s__ != ORBX_OK) return s__;       \
    } while (0)

// A handle owns one stream and one set of buffers and serves one call at a time.  Every entry point passes this
// gate first: a second thread that arrives while a call is running gets ORBX_BUSY instead of racing on the buffers.
// The thread that holds the gate may re-enter (entry points call each other).
struct CallGate {
    std::atomic<uintptr_t> owner{0};
    int depth = 0;
};
class CallGuard {
public:
    explicit CallGuard(CallGate& g) : gate_(g) {
        static thread_local char tag;
        const uintptr_t me = (uintptr_t)&tag;
        uintptr_t free_slot = 0;
        entered_ = gate_.owner.load(std::memory_order_relaxed) == me ||
                   gate_.owner.compare_exchange_strong(free_slot, me, std::memory_order_acquire);
        if (entered_) gate_.depth++;
    }
    ~CallGuard() {
        if (entered_ && --gate_.depth == 0) gate_.owner.store(0, std::memory_order_release);
    }
    bool entered() const { return entered_; }
private:
    CallGate& gate_;
    bool entered_;
};
#define ORBX_ENTER(h)                                                                                          \
    if (!(h)) return ORBX_BAD_ARG;                                                                             \
    ::orbx::CallGuard call_guard__((h)->gate);                                                                 \
    if (!call_guard__.entered()) {                                                                             \
        ::orbx::set_last_error_text("the handle is serving a call of another thread (one call at a time per "  \
                                    "handle: give every thread its own)");                                     \
        return ORBX_BUSY;                                                                                      \
    }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel on a device, shared by every handle of the
// process: it must only ever be RAISED (a handle with a small request must not lower what another handle's launches
// rely on).  Remembers the largest request per (device, kernel) and sets the attribute only above it.
orbx_status ensure_dynamic_lds(const void* kernel, size_t bytes);

// Device allocation that remembers its size; freed by the owning handle.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    orbx_status ensure(size_t count) {
        if (count <= n) return ORBX_OK;
        if (p) { ORBX_HIP(hipFree(p)); p = nullptr; n = 0; }
        ORBX_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return ORBX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Pinned host staging buffer.
template <typename T>
struct PinBuf {
    T* p = nullptr;
    size_t n = 0;
    orbx_status ensure(size_t count) {
        if (count <= n) return ORBX_OK;
        if (p) { ORBX_HIP(hipHostFree(p)); p = nullptr; n = 0; }
        ORBX_HIP(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
        n = count;
        return ORBX_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
};

// A fixed set of named event pairs; record(i, true/false) brackets the launches of stage i on one stream.
struct StageTimer {
    static const int kMax = 12;
    hipEvent_t start[kMax] = {}, stop[kMax] = {};
    bool used[kMax] = {};
    bool ready = false;
    orbx_status init() {
        for (int i = 0; i < kMax; i++) {
            ORBX_HIP(hipEventCreate(&start[i]));
            ORBX_HIP(hipEventCreate(&stop[i]));
        }
        ready = true;
        return ORBX_OK;
    }
    void destroy() {
        if (!ready) return;
        for (int i = 0; i < kMax; i++) {
            (void)hipEventDestroy(start[i]);
            (void)hipEventDestroy(stop[i]);
        }
        ready = false;
    }
    void clear() { std::memset(used, 0, sizeof(used)); }
    void begin(int i, hipStream_t s) { (void)hipEventRecord(start[i], s); used[i] = true; }
    void end(int i, hipStream_t s) { (void)hipEventRecord(stop[i], s); }
    float ms(int i) const {
        if (!used[i]) return 0.f;
        float v = 0.f;
        if (hipEventElapsedTime(&v, start[i], stop[i]) != hipSuccess) return 0.f;
        return v;
    }
};

inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace orbx
