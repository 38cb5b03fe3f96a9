// Scratch: where does the immediate offset of global_load_lds_dwordx4 apply -- to the global address only, or to the LDS
// address as well?  (LDS-DMA: LDS_ADDR = M0 base + inst_offset + 16 lane, per the CDNA3 ISA's wording for the MUBUF form.)
// One wave: global words g[i] = i; piece issued with M0 = 4096, offset:1024; LDS pre-filled with -1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* g, int* out) {
    __shared__ __attribute__((aligned(16))) int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = -1;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) int*)lds) + 4096;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(g), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<int> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    int *g, *out;
    hipMalloc((void**)&g, 8192 * 4); hipMemcpy(g, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipMalloc((void**)&out, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, out);
    hipDeviceSynchronize();
    std::vector<int> o(4096);
    hipMemcpy(o.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; ++i) if (o[i] != -1) { if (first < 0) first = i; last = i; }
    printf("LDS words written: %d..%d (M0 base = word 1024; offset:1024 bytes = 256 words)\n", first, last);
    if (first >= 0) printf("first written word holds global word %d (offset applied to the global side: 256)\n", o[first]);
    return 0;
}
