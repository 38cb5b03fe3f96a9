// Scratch: how much other work fits between the MFMAs of ONE wave per SIMD before the matrix pipe starves, and does it
// matter whether consecutive MFMAs accumulate into the same registers (one dependency chain, what the single-accumulator
// forward kernels issue) or alternate between two accumulators?
//   mfma_issue              -> table: cycles per v_mfma_f32_32x32x16_f16 (32 = pipe-bound) for 1 / 2 chains x 0..12 VALU
//                              instructions (v_fma_f32 on private registers) per MFMA, and with one LDS-DMA piece
//                              (global_load_lds_dwordx4, 1 KiB) per 3 MFMAs as in the kernels' k-steps
// One workgroup of 4 waves per CU (96 KiB of LDS requested), order pinned with volatile asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int CH, int NV, int DMA, int NS = 0, int ND = 0>
__global__ __launch_bounds__(256) void k(const h16x8* __restrict__ src, const char* __restrict__ img, float* out,
                                         unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 256 * i) % 4096]; b[i] = src[(threadIdx.x * 7 + 64 * i + 13) % 4096]; }
    f32x16 acc[2] = {{0}, {0}};
    float v[12];
    h16x8 frag[4];
    for (int j = 0; j < 12; ++j) v[j] = 1.0f + j + threadIdx.x;
    const float c = 0.999f, d = 0.001f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
    const unsigned voff = (threadIdx.x & 63) * 16;
    const char* g = img + (blockIdx.x & 7) * 65536 + wave * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            f32x16& A = acc[CH == 1 ? 0 : (u & 1)];
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(A) : "v"(a[u & 3]), "v"(b[(u + 1) & 3]));
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
#pragma unroll
            for (int j = 0; j < NS; ++j) { unsigned t; asm volatile("s_add_u32 %0, %1, 0x1000" : "=s"(t) : "s"(lds) : "scc"); }
            if (ND && u % 3 != 2) {                           // two LDS fragment reads per three MFMAs, consumed a k-step later
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[u & 3]) : "v"(voff), "n"(0));
            }
            if (ND && u % 3 == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            if (DMA == 2 && u % 3 == 2) {                      // grouped form: one instruction per piece
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(g), "{m0}"(lds), "n"(1024) : "memory");
            }
            if (DMA == 1 && u % 3 == 2) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(g + (u / 3) * 4096), "s"(lds + (u / 3) * 4096) : "memory");
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    for (int j = 0; j < 12; ++j) s += v[j];
    if (ND) for (int j = 0; j < 4; ++j) s += (float)frag[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CH, int NV, int DMA, int NS = 0, int ND = 0>
static double run(const h16x8* src, const char* img, float* out, unsigned long long* cyc) {
    const int blocks = 256, iters = 2000, lds = 96 * 1024;
    hipFuncSetAttribute((const void*)k<CH, NV, DMA, NS, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<CH, NV, DMA, NS, ND>), dim3(blocks), dim3(256), lds, 0, src, img, out, cyc, iters);
    hipLaunchKernelGGL((k<CH, NV, DMA, NS, ND>), dim3(blocks), dim3(256), lds, 0, src, img, out, cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return -1; }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : h) s += (double)x;
    return s / h.size() / ((double)iters * 12);
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.0f);
    h16x8* src; float* out; unsigned long long* cyc; char* img;
    hipMalloc((void**)&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc((void**)&out, 256 * 256 * 4); hipMalloc((void**)&cyc, 256 * 4 * 8);
    hipMalloc((void**)&img, 8 * 65536 + 65536); hipMemset(img, 0, 8 * 65536 + 65536);
    printf("cycles per MFMA (32 = the pipe's own time), one wave per SIMD\n");
    printf("VALU per MFMA:            0      2      4      6      8     12\n");
#define ROW(CH, DMA, label)                                                                                             \
    printf("%s %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", label, run<CH, 0, DMA>(src, img, out, cyc),                      \
           run<CH, 2, DMA>(src, img, out, cyc), run<CH, 4, DMA>(src, img, out, cyc), run<CH, 6, DMA>(src, img, out, cyc), \
           run<CH, 8, DMA>(src, img, out, cyc), run<CH, 12, DMA>(src, img, out, cyc));
    ROW(1, 0, "1 chain              ")
    ROW(2, 0, "2 chains             ")
    ROW(1, 1, "1 chain  + DMA piece ")
    ROW(2, 1, "2 chains + DMA piece ")
    ROW(1, 2, "1 chain + grouped DMA")
#define ROW2(NS, ND, label)                                                                                            \
    printf("%s %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n", label, run<1, 0, 0, NS, ND>(src, img, out, cyc),                \
           run<1, 2, 0, NS, ND>(src, img, out, cyc), run<1, 4, 0, NS, ND>(src, img, out, cyc), run<1, 6, 0, NS, ND>(src, img, out, cyc), \
           run<1, 8, 0, NS, ND>(src, img, out, cyc), run<1, 12, 0, NS, ND>(src, img, out, cyc));
    ROW2(2, 0, "1 chain + 2 SALU/MFMA")
    ROW2(4, 0, "1 chain + 4 SALU/MFMA")
    ROW2(0, 1, "1 chain + LDS reads  ")
    return 0;
}
