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

// Opt a kernel into > 64 KiB of dynamic LDS, once per device of this process (`done`: one bit per device id;
// the attribute is per device, and a process may drive more than one).
static inline int reserve_lds(const void* fn, int bytes, unsigned long long& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return HNRF_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS", what, bytes);
        return HNRF_E_LAUNCH;
    }
    done |= bit;
    return HNRF_OK;
}

// HNRF_MLP_F16X3 back end (hnrf_mlp_f16.hip)
size_t canonical16_bytes();
size_t nonrigid16_bytes();
size_t canonical16_status_offset();
size_t nonrigid16_status_offset();
int canonical16_pack(const float* const* w, const float* const* b, void* packed, hipStream_t st);
int nonrigid16_pack(const float* const* w, const float* const* b, const float* cond, void* packed, hipStream_t st);
int canonical16_fwd(const float* xyz, const void* packed, int64_t P, float* raw, const int* idx, const int* count,
                    bool guard, hipStream_t st);
int canonical16_fwd_train(const float* xyz, const void* packed, int64_t P, float* raw, float* pe_out, float* acts,
                          uint32_t* relu_bits, int half, hipStream_t st);
int nonrigid16_fwd(const float* x_skel, const float* hann_w, const void* packed, int64_t P, float* xyz,
                   float* offsets, const int* idx, const int* count, bool guard, hipStream_t st);

int nonrigid16_fwd_train(const float* x_skel, const float* hann_w, const void* packed, int64_t P, float* xyz,
                         float* offsets, float* pe_out, float* acts, uint32_t* relu_bits, int half, hipStream_t st);

size_t canonical16_bwd_bytes();
int canonical16_bwd_pack(const float* const* w, void* packed, hipStream_t st);
int canonical16_bwd(const float* xyz, const float* d_raw, const uint32_t* relu_bits, const void* packed, int64_t P,
                    const float* d_raw_amax, float* dZ, float* d_xyz, float* dz_amax, int half, hipStream_t st);

size_t nonrigid16_bwd_bytes();
int nonrigid16_bwd_pack(const float* const* w, void* packed, hipStream_t st);
int nonrigid16_bwd(const float* x_skel, const float* hann_w, const float* d_xyz, const uint32_t* relu_bits,
                   const void* packed, int64_t P, const float* d_xyz_amax, float* dZ, float* d_x_skel, float* dz_amax,
                   int half, hipStream_t st);

}  // namespace hnrf
