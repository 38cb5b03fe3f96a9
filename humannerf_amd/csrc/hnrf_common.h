// Shared helpers for libhnrf (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/hnrf.h"

namespace hnrf {

void set_error(const char* fmt, ...);

#define HNRF_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            hnrf::set_error(__VA_ARGS__);        \
            return (code);                       \
        }                                        \
    } while (0)

static inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return HNRF_E_LAUNCH;
    }
    return HNRF_OK;
}

constexpr int kWave = 64;

}  // namespace hnrf
