// Shared helpers for the libmakani_amd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

namespace mk {

void set_error(const std::string& msg);

#define MK_REQUIRE(cond, msg)                                                     \
    do {                                                                          \
        if (!(cond)) {                                                            \
            ::mk::set_error(std::string(__func__) + ": " + (msg));                \
            return 1;                                                             \
        }                                                                         \
    } while (0)

#define MK_LAUNCH_CHECK()                                                         \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            ::mk::set_error(std::string(__func__) + ": " + hipGetErrorString(e__)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace mk
