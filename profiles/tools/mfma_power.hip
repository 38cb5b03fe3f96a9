// Scratch: what does the matrix pipe ALONE cost in power?  Pure v_mfma_f32_32x32x16_f16 stream (3 independent
// accumulators, operands = random f16 data rotating through 8 register sets, no LDS / VMEM in the loop), looped for
// ~6 s so that rocm-smi can be polled next to it (profiles/tools/power_probe.py does the same for the real kernel).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(const h16x8* __restrict__ src, float* out, int iters) {
    h16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 256 * i) % 4096]; b[i] = src[(threadIdx.x * 7 + 64 * i + 13) % 4096]; }
    f32x16 acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[u], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + 3) & 7], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(u + 5) & 7], b[u], acc3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 6.0;
    // operand kinds (the power limit makes the sustained rate a measure of energy per MFMA):
    //  0 random in [-2, 2)   1 the same, half of the values zero (post-ReLU)   2 non-negative   3 5-bit mantissas
    //  4 tiny (1e-5 scale: f16 subnormals, like un-scaled low parts)   5 all zero
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<_Float16> h(4096 * 8);
    for (auto& v : h) {
        float f = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
        if (mode == 1 && (rand() & 1)) f = 0.f;
        if (mode == 2) f = f < 0 ? -f : f;
        if (mode == 4) f *= 1e-5f;
        if (mode == 5) f = 0.f;
        v = (_Float16)f;
        if (mode == 3) { unsigned short u; memcpy(&u, &v, 2); u &= 0xFFE0; memcpy(&v, &u, 2); }
    }
    h16x8* src; float* out;
    hipMalloc((void**)&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int blocks = 1024, iters = 20000;
    hipMalloc((void**)&out, blocks * 256 * 4);
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, out, iters);
        hipDeviceSynchronize();
        ++launches;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double mfma = (double)launches * blocks * 4 * iters * 24;
    printf("mode %d: ", mode);
    printf("pure MFMA: %.2f s, %.1f TFLOP/s f16 (%.1f %% of 2516 nominal)\n", el, mfma * 2 * 16384 / el / 1e12,
           100 * mfma * 2 * 16384 / el / 1e12 / 2516);
    return 0;
}
