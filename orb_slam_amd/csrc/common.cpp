// Error reporting shared by the orbx translation units.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

namespace orbx {
static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    g_last_error = buf;
}
void set_last_error_text(const char* text) { g_last_error = text; }
const std::string& last_error() { return g_last_error; }

orbx_status ensure_dynamic_lds(const void* kernel, size_t bytes) {
    static std::mutex m;
    static std::map<std::pair<int, const void*>, size_t> granted;
    int device = 0;
    ORBX_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(m);
    size_t& have = granted[std::make_pair(device, kernel)];
    if (bytes <= have) return ORBX_OK;
    ORBX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return ORBX_OK;
}
}  // namespace orbx

extern "C" {

const char* orbx_status_string(orbx_status s) {
    switch (s) {
        case ORBX_OK: return "ok";
        case ORBX_BAD_ARG: return "bad argument";
        case ORBX_CAPACITY: return "capacity exceeded";
        case ORBX_DEVICE: return "device error";
        case ORBX_UNSUPPORTED: return "unsupported";
        case ORBX_BUSY: return "handle busy";
    }
    return "unknown";
}

const char* orbx_last_error(void) { return orbx::last_error().c_str(); }

int orbx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
