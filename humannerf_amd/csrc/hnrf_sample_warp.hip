// K1: z-sampling along the ray + inverse-LBS warp into the canonical space.
//
// One lane per sample.  For each of the B bones: bone-local position
// pos = R_b x + T_b, trilinear lookup of the bone's 32^3 skinning-weight channel
// at pos (align_corners, zero padding -- the F.grid_sample semantics of the
// reference, network.py:409-413), then x_skel = sum_b w_b pos_b / max(sum w, 1e-4).
// The reference computes the 24 bone-local positions twice and launches 24
// grid_sample kernels (network.py:407-425); here everything stays in registers.
//
// Memory: the weight volume (24 x 32^3 fp32 = 3.1 MB) is L2-resident; the 8
// corner gathers per bone are the dominant traffic (768 B of cache traffic per
// sample).  HBM traffic per sample is 4 B (z) + 12 B (x_skel) + 4 B (mask) out.
#include "hnrf_common.h"

namespace hnrf {

// 8-byte gather that only promises 4-byte alignment
struct __attribute__((packed, aligned(4))) f32x2u { float x, y; };

// torch.linspace(0, 1, S)[s] in fp32 (start + step*i below the midpoint,
// end - step*(S-1-i) above it), then the reference's lerp (network.py:457-458).
__device__ __forceinline__ float z_at(float nr, float fr, int s, int S) {
#pragma clang fp contract(off)
    const float step = 1.0f / (float)(S - 1);
    const float t = (s < S / 2) ? step * (float)s : 1.0f - step * (float)(S - 1 - s);
    return nr * (1.0f - t) + fr * t;
}

// BT: the bone count as a compile-time constant (24 = the SMPL skeleton of every config of the reference; 0 = read B at
// run time).  With BT the four-bone trips lose their per-bone `b < B` branches.
template <bool WRITE_BMW, int BT>
__global__ __launch_bounds__(256) void sample_warp_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ near, const float* __restrict__ far,
    const float* __restrict__ t_rand,
    const float* __restrict__ Rs, const float* __restrict__ Ts,
    const float* __restrict__ vol, const float* __restrict__ bbox_min,
    const float* __restrict__ bbox_scale,
    int64_t P, int S, int B, int G,
    float* __restrict__ z_vals, float* __restrict__ x_skel,
    float* __restrict__ fg_mask, float* __restrict__ bmw) {
    // the diagnostic per-bone weights of a block's 256 samples are one contiguous run of 256 B floats: staged in LDS as
    // [quad of bones][sample] and written back as whole 1-KiB wavefront stores (as they leave the bone loop -- 16 bytes
    // per lane at a 96-byte stride -- every store instruction touched 64 different 128-byte lines: 1.28 GB of write
    // requests per chunk for 0.40 GB of weights, profiles/r03_frame_traffic.csv)
    // Arithmetic: every operation of the reference's tensor expressions is rounded on its own (no compiler-chosen fma:
    // with contraction left to hipcc, two instances of this template could differ in the last bit of a grid coordinate,
    // which the division by a small weight sum turns into 1e-4 of x_skel); the one place that IS a sum of products in
    // the reference (the 3x3 transform: a matmul) use explicit fmaf.
#pragma clang fp contract(off)
    __shared__ float4 stage[WRITE_BMW ? 6 * 256 : 1];
    const int64_t p_raw = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in_range = p_raw < P;
    const int64_t p = in_range ? p_raw : P - 1;        // (lanes past the end recompute the last sample and store nothing)
    const int64_t r = p / S;
    const int s = (int)(p - r * S);

    const float nr = near[r], fr = far[r];
    float z = z_at(nr, fr, s, S);
    if (t_rand != nullptr) {   // _stratified_sampling, network.py:463-471
        const float zm = (s > 0) ? z_at(nr, fr, s - 1, S) : z;
        const float zp = (s < S - 1) ? z_at(nr, fr, s + 1, S) : z;
        const float lower = (s > 0) ? 0.5f * (z + zm) : z;
        const float upper = (s < S - 1) ? 0.5f * (zp + z) : z;
        z = lower + (upper - lower) * t_rand[p];
    }
    const float px = rays_o[r * 3 + 0] + rays_d[r * 3 + 0] * z;
    const float py = rays_o[r * 3 + 1] + rays_d[r * 3 + 1] * z;
    const float pz = rays_o[r * 3 + 2] + rays_d[r * 3 + 2] * z;

    const float bmx = bbox_min[0], bmy = bbox_min[1], bmz = bbox_min[2];
    const float bsx = bbox_scale[0], bsy = bbox_scale[1], bsz = bbox_scale[2];
    const float gm1 = (float)(G - 1);
    const int GG = G * G;

    float wsum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    // four bones per trip (their 16 gathers in flight together); the diagnostic per-bone weights leave as one 16-byte
    // store per trip instead of four 4-byte stores at a 96-byte stride between lanes
    const bool bmw4 = WRITE_BMW && BT == 24;           // staged form (the default skeleton); any other B: plain stores
    const int Bn = BT ? BT : B;
    for (int b0 = 0; b0 < Bn; b0 += 4) {
    float w4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = b0 + j;
        if (!BT && b >= B) break;
        const float* Rb = Rs + b * 9;   // wave-uniform address: scalar loads
        const float* Tb = Ts + b * 3;
        const float qx = fmaf(Rb[2], pz, fmaf(Rb[1], py, Rb[0] * px)) + Tb[0];
        const float qy = fmaf(Rb[5], pz, fmaf(Rb[4], py, Rb[3] * px)) + Tb[1];
        const float qz = fmaf(Rb[8], pz, fmaf(Rb[7], py, Rb[6] * px)) + Tb[2];
        // normalised grid coordinate, then grid_sample's align_corners un-normalise
        const float ix = (((qx - bmx) * bsx - 1.0f) + 1.0f) * 0.5f * gm1;
        const float iy = (((qy - bmy) * bsy - 1.0f) + 1.0f) * 0.5f * gm1;
        const float iz = (((qz - bmz) * bsz - 1.0f) + 1.0f) * 0.5f * gm1;
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const float wx1 = ix - fx0, wy1 = iy - fy0, wz1 = iz - fz0;
        const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy, wz0 = (fz0 + 1.0f) - iz;
        // clamp before the int conversion: far-away points must not overflow
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), gm1 + 1.0f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), gm1 + 1.0f);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), gm1 + 1.0f);
        const bool vx0 = (x0 >= 0) & (x0 < G), vx1 = (x0 + 1 >= 0) & (x0 + 1 < G);
        const bool vy0 = (y0 >= 0) & (y0 < G), vy1 = (y0 + 1 >= 0) & (y0 + 1 < G);
        const bool vz0 = (z0 >= 0) & (z0 < G), vz1 = (z0 + 1 >= 0) & (z0 + 1 < G);
        // x-neighbours are adjacent in memory: one 8-byte gather at clamp(x0, 0, G-2) serves both corners
        // (whichever of the two is out of range is masked below, so only the in-range one has to be right)
        const int xb = min(max(x0, 0), G - 2);
        const bool lo_is_a = (x0 == xb), hi_is_b = (x0 == xb);   // x0 == G-1 -> corner 0 is .y; x0 == -1 -> corner 1 is .x
        const int cy0 = min(max(y0, 0), G - 1), cy1 = min(max(y0 + 1, 0), G - 1);
        const int cz0 = min(max(z0, 0), G - 1), cz1 = min(max(z0 + 1, 0), G - 1);
        // wave-uniform base (SGPR pair) + one unsigned element offset per gather: with the lane's x folded into the
        // pointer every gather paid a sign extension and a 64-bit add chain of its own (this kernel is VALU-bound)
        const float* vb = vol + (size_t)b * G * GG;
        const unsigned rz0 = (unsigned)(cz0 * GG + xb), rz1 = (unsigned)(cz1 * GG + xb);
        const unsigned ry0 = (unsigned)(cy0 * G), ry1 = (unsigned)(cy1 * G);
        const f32x2u p00 = *(const f32x2u*)(vb + (rz0 + ry0));
        const f32x2u p01 = *(const f32x2u*)(vb + (rz0 + ry1));
        const f32x2u p10 = *(const f32x2u*)(vb + (rz1 + ry0));
        const f32x2u p11 = *(const f32x2u*)(vb + (rz1 + ry1));
        const float v000 = lo_is_a ? p00.x : p00.y, v001 = hi_is_b ? p00.y : p00.x;
        const float v010 = lo_is_a ? p01.x : p01.y, v011 = hi_is_b ? p01.y : p01.x;
        const float v100 = lo_is_a ? p10.x : p10.y, v101 = hi_is_b ? p10.y : p10.x;
        const float v110 = lo_is_a ? p11.x : p11.y, v111 = hi_is_b ? p11.y : p11.x;
        // zero padding through the six 1-D weights instead of eight per-corner selects: an out-of-range corner gets the
        // product value * (0 * . * .) = 0, and adding that zero leaves the sum's bits as skipping the corner did (the
        // value gathered at the clamped address is a finite entry of the volume); in-range corners: the same products
        const float ux0 = vx0 ? wx0 : 0.f, ux1 = vx1 ? wx1 : 0.f;
        const float uy0 = vy0 ? wy0 : 0.f, uy1 = vy1 ? wy1 : 0.f;
        const float uz0 = vz0 ? wz0 : 0.f, uz1 = vz1 ? wz1 : 0.f;
        float w = 0.f;
        w += v000 * (ux0 * uy0 * uz0);
        w += v001 * (ux1 * uy0 * uz0);
        w += v010 * (ux0 * uy1 * uz0);
        w += v011 * (ux1 * uy1 * uz0);
        w += v100 * (ux0 * uy0 * uz1);
        w += v101 * (ux1 * uy0 * uz1);
        w += v110 * (ux0 * uy1 * uz1);
        w += v111 * (ux1 * uy1 * uz1);
        wsum += w;
        ax += w * qx;                   // (product rounded, then summed: torch.sum over the stacked w_b * pos_b, network.py:437-441)
        ay += w * qy;
        az += w * qz;
        w4[j] = w;
        if (WRITE_BMW && !bmw4 && in_range) bmw[p * B + b] = w;
    }
    if (bmw4) stage[(b0 >> 2) * 256 + threadIdx.x] = make_float4(w4[0], w4[1], w4[2], w4[3]);
    }
    if (in_range) {
        const float den = fmaxf(wsum, 0.0001f);
        z_vals[p] = z;
        x_skel[p * 3 + 0] = ax / den;
        x_skel[p * 3 + 1] = ay / den;
        x_skel[p * 3 + 2] = az / den;
        fg_mask[p] = wsum;
    }
    if (bmw4) {                                        // (block-uniform: no thread has left the kernel)
        __syncthreads();
        const int64_t base = (int64_t)blockIdx.x * 256;
        const int nq = (int)((P - base < 256 ? P - base : 256) * 6);       // float4s of this block's run
        float4* out = reinterpret_cast<float4*>(bmw + base * 24);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int o = k * 256 + threadIdx.x;       // linear float4 of the run = sample o / 6, quad o % 6
            if (o < nq) out[o] = stage[(o % 6) * 256 + o / 6];
        }
    }
}

// Stream compaction of the samples worth evaluating: idx[0..count) = indices p with fg_mask[p] >= eps.
// A sample with fg_mask < eps has alpha < eps (alpha = (1-exp(..)) * fg_mask), so dropping it changes a
// ray's rgb / alpha / depth by at most ~2 S eps (own weight + its effect on later transmittances);
// eps = 0 keeps everything.  Order: block-contiguous runs, blocks in arrival order (irrelevant to the MLPs).
__global__ __launch_bounds__(256) void compact_kernel(const float* __restrict__ fg_mask, float eps, int64_t P,
                                                      int* __restrict__ idx, int* __restrict__ count) {
    __shared__ int wave_tot[4];
    __shared__ int block_base;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool keep = p < P && fg_mask[p] >= eps;
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) block_base = atomicAdd(count, wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]);
    __syncthreads();
    int off = block_base + before;
    for (int w = 0; w < wave; ++w) off += wave_tot[w];
    if (keep) idx[off] = (int)p;
}

}  // namespace hnrf

extern "C" int hnrf_compact_samples(const float* fg_mask, float eps, int64_t P, int* idx, int* count, void* stream) {
    using namespace hnrf;
    HNRF_REQUIRE(fg_mask && idx && count, HNRF_E_ARG, "hnrf_compact_samples: null pointer");
    HNRF_REQUIRE(P >= 0 && P < 2147483647LL, HNRF_E_ARG, "hnrf_compact_samples: bad P");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(count, 0, sizeof(int), st) != hipSuccess) {
        set_error("hnrf_compact_samples: memset failed");
        return HNRF_E_LAUNCH;
    }
    if (P == 0) return HNRF_OK;
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, fg_mask, eps, P, idx, count);
    return check_launch("hnrf_compact_samples");
}

extern "C" int hnrf_sample_warp_fwd(const float* rays_o, const float* rays_d,
                                    const float* near, const float* far, const float* t_rand,
                                    const float* motion_Rs, const float* motion_Ts,
                                    const float* vol, const float* bbox_min, const float* bbox_scale,
                                    int64_t R, int S, int B, int G,
                                    float* z_vals, float* x_skel, float* fg_mask, float* bmw,
                                    void* stream) {
    using namespace hnrf;
    HNRF_REQUIRE(rays_o && rays_d && near && far && motion_Rs && motion_Ts && vol && bbox_min && bbox_scale,
                 HNRF_E_ARG, "hnrf_sample_warp_fwd: null input pointer");
    HNRF_REQUIRE(z_vals && x_skel && fg_mask, HNRF_E_ARG, "hnrf_sample_warp_fwd: null output pointer");
    HNRF_REQUIRE(R >= 0 && S >= 2 && B >= 1 && G >= 2 && G <= 1024, HNRF_E_ARG,
                 "hnrf_sample_warp_fwd: bad dims R=%lld S=%d B=%d G=%d", (long long)R, S, B, G);
    HNRF_REQUIRE(!bmw || B != 24 || ((uintptr_t)bmw & 15) == 0, HNRF_E_ARG,
                 "hnrf_sample_warp_fwd: with 24 bones the per-bone weight output must be 16-byte aligned (written in 16-byte pieces)");
    if (R == 0) return HNRF_OK;
    const int64_t P = R * (int64_t)S;
    const int64_t blocks = (P + 255) / 256;
    HNRF_REQUIRE(blocks < (int64_t)2147483647, HNRF_E_ARG, "hnrf_sample_warp_fwd: too many samples");
    hipStream_t st = (hipStream_t)stream;
#define HNRF_K1(W, BT_)                                                                                               \
    hipLaunchKernelGGL((sample_warp_kernel<W, BT_>), dim3((unsigned)blocks), dim3(256), 0, st, rays_o, rays_d, near, far, \
                       t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, P, S, B, G, z_vals, x_skel, fg_mask, bmw)
    if (bmw) { if (B == 24) HNRF_K1(true, 24); else HNRF_K1(true, 0); }
    else { if (B == 24) HNRF_K1(false, 24); else HNRF_K1(false, 0); }
#undef HNRF_K1
    return check_launch("hnrf_sample_warp_fwd");
}
