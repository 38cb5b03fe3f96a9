// Scratch micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 in the accumulate pattern of the MLP kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    h16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 2, 2, 3, 3, 4, 4}, c = {2, 2, 2, 2, 1, 1, 1, 1};
    a[0] = (_Float16)threadIdx.x;
    f32x16 acc1 = {0}, acc2 = {0}, acc3 = {0};
    long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {          // kernel pattern: acc1, acc2, acc2
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, c, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, acc2, 0, 0, 0);
            } else if (MODE == 1) {   // three independent accumulators
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, c, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, acc3, 0, 0, 0);
            } else {                  // single chain
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, c, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, b, acc1, 0, 0, 0);
            }
        }
    }
    long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 48.0f);
}

template <int MODE>
void run(const char* name, int blocks) {
    float* d;
    hipMalloc(&d, blocks * 256 * 4);
    int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    double mf = (double)blocks * 4 * iters * 48;
    printf("%-28s blocks %4d: %.3f ms, %.1f ns-clock cycles/MFMA(memtime), %.1f TFLOP/s f16, per-SIMD %.2f ns/MFMA\n", name, blocks, ms, cyc,
           mf * 2 * 16384 / ms / 1e9, ms * 1e6 / (mf / (blocks * 4 < 1024 ? blocks * 4 : 1024)));
    hipFree(d);
}
int main() {
    for (int blocks : {256, 1024}) {
        run<0>("pattern acc1,acc2,acc2", blocks);
        run<1>("3 independent accs", blocks);
        run<2>("single chain", blocks);
    }
    return 0;
}
