// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace clibd {

// thread-local last-error text (defined in capi.hip)
char* last_error_buf();
constexpr int kErrBufLen = 256;

inline int set_error(int code, const char* msg) {
    snprintf(last_error_buf(), kErrBufLen, "%s", msg);
    return code;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Launch-time errors only (no synchronisation): configuration errors surface here, device faults do not.
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(last_error_buf(), kErrBufLen, "%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

// grid size for a grid-stride elementwise kernel: ceil(n / per_block), capped
inline unsigned grid_for(size_t n, int per_block = 256, unsigned cap = 4096) {
    size_t b = (n + per_block - 1) / per_block;
    if (b > cap) b = cap;
    return b < 1 ? 1u : (unsigned)b;
}

}  // namespace clibd
