// Ray generation + bbox intersection + compaction: the step in front of the hot path
// (SURVEY.md section 8(f) rank 1).  Restates get_rays_from_KRT (core/utils/camera_util.py:132-159)
// and rays_intersect_3d_bbox (camera_util.py:162-208) as the reference's datasets call them
// (core/data/human_nerf/freeview.py:220-230) with float32 cameras:
//   rays_o = -R^T T;  pixel (i, j): cam = [i, j, 1] Kinv^T;  world = (cam - T) R;  rays_d = world - rays_o
//   (all float32, direction NOT normalised);  |d_c| < 1e-5 -> 1e-5 (the reference clamps in place, so the
//   clamped direction is what the renderer sees);  slab test in float64 against the bbox padded by 1 cm:
//   the 6 plane hits p = o + t d that lie inside the box (eps 1e-6) are counted, a ray is kept iff exactly
//   two do, near/far = min/max of |p - o| / |d|_f32, cast to float32.
// The numpy version walks H*W pixels on the host per frame (tens of ms at 512x512 -- a quarter of the
// render time of the frame at this renderer's speed) and uploads 32 B per ray; here the kept rays are
// produced in HBM in pixel order (the order Network.forward and the image unpack rely on).
//
// Three launches: (1) hit flag per pixel + per-block counts, (2) exclusive scan of the block counts
// (one block), (3) recompute + write at block offset + rank (ballot / popcount).  HBM-bound: 1 B (mask)
// + 32 B per kept ray written, nothing read but 30 scalars.
#include "hnrf_common.h"

namespace hnrf {

struct Cam {
    float kinv[9], R[9], T[3];
    double lo[3], hi[3];   // padded bbox
};

struct Ray {
    float o[3], d[3], near, far;
    bool hit;
};

__device__ __forceinline__ Ray make_ray(const Cam& c, int i, int j) {
    Ray r;
    const float fi = (float)i, fj = (float)j;
    float cam[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) cam[a] = fmaf(1.0f, c.kinv[3 * a + 2], fmaf(fj, c.kinv[3 * a + 1], fi * c.kinv[3 * a]));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = -fmaf(c.R[6 + k], c.T[2], fmaf(c.R[3 + k], c.T[1], c.R[k] * c.T[0]));
        const float w = fmaf(cam[2] - c.T[2], c.R[6 + k], fmaf(cam[1] - c.T[1], c.R[3 + k], (cam[0] - c.T[0]) * c.R[k]));
        float d = w - r.o[k];
        if (fabsf(d) < 1e-5f) d = 1e-5f;
        r.d[k] = d;
    }
    // the six plane hits in the reference's order: min_x, min_y, min_z, max_x, max_y, max_z
    const double o[3] = {(double)r.o[0], (double)r.o[1], (double)r.o[2]};
    const double d[3] = {(double)r.d[0], (double)r.d[1], (double)r.d[2]};
    int nhit = 0;
    double dist[2] = {0.0, 0.0};
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const int ax = f % 3;
        const double bound = f < 3 ? c.lo[ax] : c.hi[ax];
        const double t = (bound - o[ax]) / d[ax];
        const double p0 = t * d[0] + o[0], p1 = t * d[1] + o[1], p2 = t * d[2] + o[2];
        const bool in = p0 >= c.lo[0] - 1e-6 && p0 <= c.hi[0] + 1e-6 && p1 >= c.lo[1] - 1e-6 && p1 <= c.hi[1] + 1e-6 &&
                        p2 >= c.lo[2] - 1e-6 && p2 <= c.hi[2] + 1e-6;
        if (in) {
            const double e0 = p0 - o[0], e1 = p1 - o[1], e2 = p2 - o[2];
            if (nhit < 2) dist[nhit] = sqrt(e0 * e0 + e1 * e1 + e2 * e2);
            ++nhit;
        }
    }
    r.hit = nhit == 2;
    const float nrm = sqrtf(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);   // float32 norm, like numpy's
    const double d0 = dist[0] / (double)nrm, d1 = dist[1] / (double)nrm;
    r.near = (float)fmin(d0, d1);
    r.far = (float)fmax(d0, d1);
    return r;
}

__device__ __forceinline__ Cam load_cam(const float* kinv, const float* R, const float* T, const float* bmin,
                                        const float* bmax) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 9; ++i) { c.kinv[i] = kinv[i]; c.R[i] = R[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c.T[i] = T[i];
        c.lo[i] = (double)bmin[i] + -0.01;
        c.hi[i] = (double)bmax[i] + 0.01;
    }
    return c;
}

__global__ __launch_bounds__(256) void raygen_mask_kernel(const float* kinv, const float* R, const float* T,
                                                          const float* bmin, const float* bmax, int H, int W,
                                                          uint8_t* __restrict__ ray_mask, int* __restrict__ blk_cnt) {
    __shared__ int wave_tot[4];
    const Cam c = load_cam(kinv, R, T, bmin, bmax);
    const int p = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (p < H * W) {
        hit = make_ray(c, p % W, p / W).hit;
        ray_mask[p] = hit ? 1 : 0;
    }
    const unsigned long long bal = __ballot(hit);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
}

// exclusive scan of n block counts in place; total -> *count.  One block of 1024 threads.
__global__ __launch_bounds__(1024) void raygen_scan_kernel(int* __restrict__ blk, int n, int* __restrict__ count) {
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(n, b0 + per);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += blk[b];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {              // Hillis-Steele inclusive scan
        const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int b = b0; b < b1; ++b) {
        const int v = blk[b];
        blk[b] = run;
        run += v;
    }
    if (threadIdx.x == 1023) *count = part[1023];
}

__global__ __launch_bounds__(256) void raygen_emit_kernel(const float* kinv, const float* R, const float* T,
                                                          const float* bmin, const float* bmax, int H, int W,
                                                          const int* __restrict__ blk_off, float* __restrict__ rays_o,
                                                          float* __restrict__ rays_d, float* __restrict__ near,
                                                          float* __restrict__ far) {
    __shared__ int wave_tot[4];
    const Cam c = load_cam(kinv, R, T, bmin, bmax);
    const int p = blockIdx.x * 256 + threadIdx.x;
    Ray r;
    r.hit = false;
    if (p < H * W) r = make_ray(c, p % W, p / W);
    const unsigned long long bal = __ballot(r.hit);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    if (!r.hit) return;
    int idx = blk_off[blockIdx.x] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) idx += wave_tot[w];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays_o[(int64_t)idx * 3 + k] = r.o[k];
        rays_d[(int64_t)idx * 3 + k] = r.d[k];
    }
    near[idx] = r.near;
    far[idx] = r.far;
}

}  // namespace hnrf

using namespace hnrf;

extern "C" size_t hnrf_gen_rays_workspace_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    return (size_t)(((int64_t)H * W + 255) / 256) * sizeof(int);
}

extern "C" int hnrf_gen_rays(const float* Kinv, const float* R, const float* T, const float* bbox_min,
                             const float* bbox_max, int H, int W, float* rays_o, float* rays_d, float* near,
                             float* far, uint8_t* ray_mask, int* count, void* workspace, size_t workspace_bytes,
                             void* stream) {
    HNRF_REQUIRE(Kinv && R && T && bbox_min && bbox_max && rays_o && rays_d && near && far && ray_mask && count &&
                     workspace,
                 HNRF_E_ARG, "hnrf_gen_rays: null pointer");
    HNRF_REQUIRE(H > 0 && W > 0 && (int64_t)H * W < 2147483647LL, HNRF_E_ARG, "hnrf_gen_rays: bad image size %dx%d", H, W);
    HNRF_REQUIRE(workspace_bytes >= hnrf_gen_rays_workspace_bytes(H, W), HNRF_E_ARG, "hnrf_gen_rays: workspace too small");
    const int nblk = (int)(((int64_t)H * W + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    int* blk = (int*)workspace;
    hipLaunchKernelGGL(raygen_mask_kernel, dim3(nblk), dim3(256), 0, st, Kinv, R, T, bbox_min, bbox_max, H, W, ray_mask, blk);
    hipLaunchKernelGGL(raygen_scan_kernel, dim3(1), dim3(1024), 0, st, blk, nblk, count);
    hipLaunchKernelGGL(raygen_emit_kernel, dim3(nblk), dim3(256), 0, st, Kinv, R, T, bbox_min, bbox_max, H, W, blk, rays_o,
                       rays_d, near, far);
    return check_launch("hnrf_gen_rays");
}
