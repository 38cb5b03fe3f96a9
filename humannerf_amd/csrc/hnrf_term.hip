// Opt-in early ray termination for the lean path (rgb / alpha / depth only).
//
// The reference evaluates both MLPs on every sample of every ray and only then composites
// (network.py:474-602).  Front to back, a sample behind transmittance T contributes at most T to
// the ray's colour, opacity and depth weight, so once T falls under `term_eps` the rest of the ray can be
// skipped with |d rgb|, |d alpha| <= term_eps -- the standard volume-rendering cut, not the reference
// arithmetic (term_eps = 0 never takes this path; the default path evaluates everything).
//
// The samples are walked in depth slabs of `SS` samples: per slab (1) the samples of still-alive rays that also
// pass the foreground-likelihood cut are compacted (ballot / popcount, count stays on the device), (2) K2 and K3
// run in their sparse forms on that list, (3) the slab is composited into per-ray running sums and the running
// transmittance.  2 rays per wavefront (32 samples each), segmented product scan.
#include "hnrf_common.h"

namespace hnrf {

struct RayState {          // per ray, 6 floats: T, sum w r, sum w g, sum w b, sum w, sum w z
    float v[6];
};

__global__ void term_init_kernel(float* __restrict__ st, int64_t R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * 6) st[i] = (i % 6 == 0) ? 1.0f : 0.0f;
}

// idx[0..count) = global sample indices r*S + s of slab [s0, s0+ss) whose ray is alive and whose fg_mask passes
__global__ __launch_bounds__(256) void compact_slab_kernel(const float* __restrict__ fg_mask,
                                                           const float* __restrict__ st, float cull_eps,
                                                           float term_eps, int64_t R, int S, int s0, int ss,
                                                           int* __restrict__ idx, int* __restrict__ count) {
    __shared__ int wave_tot[4];
    __shared__ int block_base;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    int64_t p = 0;
    if (i < R * ss) {
        const int64_t r = i / ss;
        p = r * S + s0 + (int)(i - r * ss);
        keep = st[r * 6] >= term_eps && fg_mask[p] >= cull_eps;
    }
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

// One slab of <= 32 samples of one ray per half-wave: alpha from raw where the sample was evaluated (same
// predicate as compact_slab_kernel, on the state BEFORE this slab), running transmittance, running sums.
__global__ __launch_bounds__(256) void composite_slab_kernel(const float4* __restrict__ raw,
                                                             const float* __restrict__ fg_mask,
                                                             const float* __restrict__ z_vals,
                                                             const float* __restrict__ rays_d, float cull_eps,
                                                             float term_eps, int64_t R, int S, int s0, int ss,
                                                             float* __restrict__ st) {
    const int l32 = threadIdx.x & 31;
    const int64_t ray = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (ray >= R) return;                                     // whole half-waves; shuffles below stay inside a half
    float* s_ = st + ray * 6;
    const float T_in = s_[0];
    const bool alive = T_in >= term_eps;
    const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const int s = s0 + l32;
    const bool in = l32 < ss;
    const int64_t p = ray * S + (in ? s : s0);
    const float z = z_vals[p];
    const float zn = (in && s + 1 < S) ? z_vals[p + 1] : z;
    const float mk = in ? fg_mask[p] : 0.f;
    const bool live = in && alive && mk >= cull_eps;
    float al = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
    if (live) {
        const float4 rw = raw[p];
        const float dist = ((s >= S - 1) ? 1e10f : (zn - z)) * dnorm;
        al = (1.0f - expf(-fmaxf(rw.w, 0.f) * dist)) * mk;
        cr = 1.0f / (1.0f + expf(-rw.x));
        cg = 1.0f / (1.0f + expf(-rw.y));
        cb = 1.0f / (1.0f + expf(-rw.z));
    }
    const float f = in ? (1.0f - al + 1e-10f) : 1.0f;
    float incl = f;                                           // inclusive product scan inside the half-wave
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const float o = __shfl_up(incl, off, 32);
        if (l32 >= off) incl *= o;
    }
    float Tex = __shfl_up(incl, 1, 32);
    if (l32 == 0) Tex = 1.f;
    const float w = al * (T_in * Tex);
    float sr = w * cr, sg = w * cg, sb = w * cb, sa = w, sd = w * z;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        sr += __shfl_xor(sr, off, 32);
        sg += __shfl_xor(sg, off, 32);
        sb += __shfl_xor(sb, off, 32);
        sa += __shfl_xor(sa, off, 32);
        sd += __shfl_xor(sd, off, 32);
    }
    const float Tall = __shfl(incl, 31, 32);
    if (l32 == 0) {
        s_[0] = T_in * Tall;
        s_[1] += sr;
        s_[2] += sg;
        s_[3] += sb;
        s_[4] += sa;
        s_[5] += sd;
    }
}

__global__ void term_finish_kernel(const float* __restrict__ st, const float* __restrict__ bgcolor, int64_t R,
                                   float* __restrict__ rgb, float* __restrict__ alpha, float* __restrict__ depth) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* s = st + r * 6;
    const float k = 1.0f - s[4];
    rgb[r * 3 + 0] = s[1] + k * bgcolor[0] / 255.f;
    rgb[r * 3 + 1] = s[2] + k * bgcolor[1] / 255.f;
    rgb[r * 3 + 2] = s[3] + k * bgcolor[2] / 255.f;
    alpha[r] = s[4];
    depth[r] = s[5];
}

__global__ void term_add_count_kernel(const int* __restrict__ c, int* __restrict__ e) { *e += *c; }

static inline size_t align256t(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace hnrf

using namespace hnrf;

extern "C" int hnrf_nonrigid_fwd_sparse(const float* x_skel, const float* hann_w, const void* packed, int mode,
                                        int64_t P, const int* idx, const int* count, float* xyz, float* offsets,
                                        void* stream);
extern "C" int hnrf_canonical_fwd_sparse(const float* xyz, const void* packed, int mode, int64_t P, const int* idx,
                                         const int* count, float* raw, void* stream);

// workspace: the carve of hnrf_render_rays_fwd + the per-ray state
extern "C" size_t hnrf_render_term_workspace_bytes(int64_t R, int S) {
    if (R < 0 || S < 0) return 0;
    return hnrf_render_workspace_bytes(R, S) + align256t((size_t)R * 6 * sizeof(float)) + 256;
}

extern "C" int hnrf_render_rays_term_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                                         const float* t_rand, const float* motion_Rs, const float* motion_Ts,
                                         const float* vol, const float* bbox_min, const float* bbox_scale,
                                         const float* hann_w, const void* nr_packed, const void* cnl_packed,
                                         const float* bgcolor, int mode, float cull_eps, float term_eps, int64_t R,
                                         int S, int B, int G, void* workspace, size_t workspace_bytes, float* rgb,
                                         float* alpha, float* depth, int* evaluated, void* stream) {
    HNRF_REQUIRE(workspace && cnl_packed && rgb && alpha && depth, HNRF_E_ARG, "hnrf_render_rays_term_fwd: null pointer");
    HNRF_REQUIRE(((uintptr_t)workspace & 255) == 0, HNRF_E_ARG, "hnrf_render_rays_term_fwd: workspace must be 256-byte aligned");
    HNRF_REQUIRE(workspace_bytes >= hnrf_render_term_workspace_bytes(R, S), HNRF_E_WORKSPACE,
                 "hnrf_render_rays_term_fwd: workspace %zu < %zu bytes", workspace_bytes,
                 hnrf_render_term_workspace_bytes(R, S));
    HNRF_REQUIRE(term_eps > 0.f && term_eps < 1.f && cull_eps >= 0.f, HNRF_E_ARG,
                 "hnrf_render_rays_term_fwd: need 0 < term_eps < 1 and cull_eps >= 0");
    HNRF_REQUIRE(nr_packed == nullptr || hann_w != nullptr, HNRF_E_ARG, "hnrf_render_rays_term_fwd: hann_w missing");
    HNRF_REQUIRE(R >= 0 && S >= 2 && (int64_t)R * 32 < 2147483647LL, HNRF_E_ARG, "hnrf_render_rays_term_fwd: bad dims");
    if (R == 0) return HNRF_OK;
    const size_t P = (size_t)R * (size_t)S;
    char* w = (char*)workspace;
    float* z_vals = (float*)w;  w += align256t(P * 4);
    float* mask = (float*)w;    w += align256t(P * 4);
    float* x_skel = (float*)w;  w += align256t(P * 12);
    float* xyz = (float*)w;     w += align256t(P * 12);
    float* raw = (float*)w;     w += align256t(P * 16);
    int* idx = (int*)w;         w += align256t(P * 4);
    int* count = (int*)w;       w += 256;
    float* st = (float*)w;
    hipStream_t stq = (hipStream_t)stream;
    int rc = hnrf_sample_warp_fwd(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                                  R, S, B, G, z_vals, x_skel, mask, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(term_init_kernel, dim3((unsigned)((R * 6 + 255) / 256)), dim3(256), 0, stq, st, R);
    if (evaluated && hipMemsetAsync(evaluated, 0, sizeof(int), stq) != hipSuccess) {
        set_error("hnrf_render_rays_term_fwd: memset failed");
        return HNRF_E_LAUNCH;
    }
    constexpr int SS = 32;
    for (int s0 = 0; s0 < S; s0 += SS) {
        const int ss = S - s0 < SS ? S - s0 : SS;
        const int64_t cap = R * (int64_t)ss;
        if (hipMemsetAsync(count, 0, sizeof(int), stq) != hipSuccess) {
            set_error("hnrf_render_rays_term_fwd: memset failed");
            return HNRF_E_LAUNCH;
        }
        hipLaunchKernelGGL(compact_slab_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stq, mask, st, cull_eps,
                           term_eps, R, S, s0, ss, idx, count);
        const float* cnl_in = x_skel;
        if (nr_packed) {
            rc = hnrf_nonrigid_fwd_sparse(x_skel, hann_w, nr_packed, mode, cap, idx, count, xyz, nullptr, stream);
            if (rc) return rc;
            cnl_in = xyz;
        }
        rc = hnrf_canonical_fwd_sparse(cnl_in, cnl_packed, mode, cap, idx, count, raw, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(composite_slab_kernel, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, stq, (const float4*)raw,
                           mask, z_vals, rays_d, cull_eps, term_eps, R, S, s0, ss, st);
        if (evaluated)     // running total of evaluated samples (diagnostic), kept on the device
            hipLaunchKernelGGL(term_add_count_kernel, dim3(1), dim3(1), 0, stq, count, evaluated);
    }
    hipLaunchKernelGGL(term_finish_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, stq, st, bgcolor, R, rgb,
                       alpha, depth);
    return check_launch("hnrf_render_rays_term_fwd");
}
