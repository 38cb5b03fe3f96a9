// K4: front-to-back alpha compositing of one ray per wavefront.
//
// Restates Network._raw2outputs (network.py:355-388): delta_i = (z_{i+1}-z_i)*|d|
// (last = 1e10*|d|), alpha = (1-exp(-relu(sigma)*delta))*mask,
// T_i = prod_{j<i}(1-alpha_j+1e-10), w = alpha*T, rgb = sum w*sigmoid(raw_rgb) +
// (1-sum w)*bg/255, depth = sum w*z, acc = sum w, argmax-w gathers.
//
// Layout: 64 lanes x SPL consecutive samples each (SPL = ceil(S/64); 2 at S=128),
// so a lane reads SPL*16 contiguous bytes of `raw` and the wave reads the whole
// 2 KB ray row coalesced.  The transmittance is a wave-level exclusive product
// scan (6 shuffle steps) instead of torch.cumprod's serial pass.
// HBM bytes per sample: 16 (raw) + 4 (mask) + 4 (z) in; per ray 20 out (+ the
// optional per-sample diagnostics).
#include "hnrf_common.h"

namespace hnrf {

template <int SPL>
__global__ __launch_bounds__(256) void composite_kernel(
    const float4* __restrict__ raw, const float* __restrict__ fg_mask, const float* __restrict__ z_vals,
    const float* __restrict__ rays_d, const float* __restrict__ xyz, const float* __restrict__ bgcolor,
    int64_t R, int S, float cull_eps,
    float* __restrict__ rgb_out, float* __restrict__ alpha_out, float* __restrict__ depth_out,
    float* __restrict__ weights_out, float* __restrict__ rgb_on_rays,
    float* __restrict__ cnl_xyz, float* __restrict__ cnl_rgb, float* __restrict__ cnl_weight) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;   // wave-uniform

    const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const int64_t base = ray * S;

    float zv[SPL], al[SPL], cr[SPL], cg[SPL], cb[SPL];
    float4 rw[SPL];
    float mk[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const bool ok = s < S;
        const int sc = ok ? s : S - 1;
        rw[i] = raw[base + sc];
        mk[i] = ok ? fg_mask[base + sc] : 0.f;
        zv[i] = z_vals[base + sc];
    }
    // z of the next lane's first sample (for the last local delta)
    const float znext_lane = __shfl_down(zv[0], 1, 64);
    float tl = 1.f;   // product of this lane's (1 - alpha + 1e-10)
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const float zn = (i + 1 < SPL) ? zv[(i + 1 < SPL) ? i + 1 : i] : znext_lane;
        float dist = (s >= S - 1) ? 1e10f : (zn - zv[i]);
        dist *= dnorm;
        const float sig = fmaxf(rw[i].w, 0.f);
        // culled samples (fg_mask < cull_eps) were never evaluated: their raw is undefined -> weight 0
        const bool live = !(mk[i] < cull_eps);
        al[i] = live ? (1.0f - expf(-sig * dist)) * mk[i] : 0.f;
        cr[i] = live ? 1.0f / (1.0f + expf(-rw[i].x)) : 0.f;
        cg[i] = live ? 1.0f / (1.0f + expf(-rw[i].y)) : 0.f;
        cb[i] = live ? 1.0f / (1.0f + expf(-rw[i].z)) : 0.f;
        if (s < S) tl *= (1.0f - al[i] + 1e-10f);
    }
    // inclusive product scan over lanes, then shift to exclusive
    float incl = tl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= o;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.f;

    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    float wbest = -1.f;
    int ibest = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        if (s < S) {
            const float w = al[i] * T;
            T *= (1.0f - al[i] + 1e-10f);
            sr += w * cr[i];
            sg += w * cg[i];
            sb += w * cb[i];
            sd += w * zv[i];
            sa += w;
            if (w > wbest) { wbest = w; ibest = s; }
            if (weights_out) weights_out[base + s] = w;
            if (rgb_on_rays) {
                rgb_on_rays[(base + s) * 3 + 0] = cr[i];
                rgb_on_rays[(base + s) * 3 + 1] = cg[i];
                rgb_on_rays[(base + s) * 3 + 2] = cb[i];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        sr += __shfl_xor(sr, off, 64);
        sg += __shfl_xor(sg, off, 64);
        sb += __shfl_xor(sb, off, 64);
        sd += __shfl_xor(sd, off, 64);
        sa += __shfl_xor(sa, off, 64);
        const float wo = __shfl_xor(wbest, off, 64);
        const int io = __shfl_xor(ibest, off, 64);
        if (wo > wbest || (wo == wbest && io < ibest)) { wbest = wo; ibest = io; }
    }
    if (lane == 0) {
        const float k = 1.0f - sa;
        rgb_out[ray * 3 + 0] = sr + k * bgcolor[0] / 255.f;
        rgb_out[ray * 3 + 1] = sg + k * bgcolor[1] / 255.f;
        rgb_out[ray * 3 + 2] = sb + k * bgcolor[2] / 255.f;
        alpha_out[ray] = sa;
        depth_out[ray] = sd;
        if (cnl_weight) cnl_weight[ray] = wbest;
    }
    if (cnl_xyz || cnl_rgb) {
        // the lane that owns sample ibest publishes the gathered values
        const int owner = ibest / SPL;
        if (lane == owner) {
            const int i = ibest - owner * SPL;
            float r_ = cr[0], g_ = cg[0], b_ = cb[0];
#pragma unroll
            for (int k = 1; k < SPL; ++k)
                if (i == k) { r_ = cr[k]; g_ = cg[k]; b_ = cb[k]; }
            if (cnl_rgb) {
                cnl_rgb[ray * 3 + 0] = r_;
                cnl_rgb[ray * 3 + 1] = g_;
                cnl_rgb[ray * 3 + 2] = b_;
            }
            if (cnl_xyz) {
                cnl_xyz[ray * 3 + 0] = xyz[(base + ibest) * 3 + 0];
                cnl_xyz[ray * 3 + 1] = xyz[(base + ibest) * 3 + 1];
                cnl_xyz[ray * 3 + 2] = xyz[(base + ibest) * 3 + 2];
            }
        }
    }
}

}  // namespace hnrf

extern "C" int hnrf_composite_fwd(const float* raw, const float* fg_mask, const float* z_vals,
                                  const float* rays_d, const float* xyz, const float* bgcolor,
                                  int64_t R, int S, float cull_eps,
                                  float* rgb, float* alpha, float* depth,
                                  float* weights, float* rgb_on_rays,
                                  float* cnl_xyz, float* cnl_rgb, float* cnl_weight,
                                  void* stream) {
    using namespace hnrf;
    HNRF_REQUIRE(raw && fg_mask && z_vals && rays_d && bgcolor, HNRF_E_ARG, "hnrf_composite_fwd: null input pointer");
    HNRF_REQUIRE(rgb && alpha && depth, HNRF_E_ARG, "hnrf_composite_fwd: null output pointer");
    HNRF_REQUIRE(!(cnl_xyz && !xyz), HNRF_E_ARG, "hnrf_composite_fwd: cnl_xyz requested without xyz");
    HNRF_REQUIRE(R >= 0 && S >= 2, HNRF_E_ARG, "hnrf_composite_fwd: bad dims R=%lld S=%d", (long long)R, S);
    HNRF_REQUIRE(S <= 512, HNRF_E_UNSUPPORTED, "hnrf_composite_fwd: S=%d > 512 samples per ray not built", S);
    HNRF_REQUIRE(((uintptr_t)raw & 15) == 0, HNRF_E_ARG, "hnrf_composite_fwd: raw must be 16-byte aligned");
    if (R == 0) return HNRF_OK;
    const int64_t blocks = (R + 3) / 4;
    HNRF_REQUIRE(blocks < (int64_t)2147483647, HNRF_E_ARG, "hnrf_composite_fwd: too many rays");
    hipStream_t st = (hipStream_t)stream;
    const int spl = (S + 63) / 64;
#define HNRF_LAUNCH_COMPOSITE(N)                                                                                  \
    hipLaunchKernelGGL(composite_kernel<N>, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)raw, fg_mask, \
                       z_vals, rays_d, xyz, bgcolor, R, S, cull_eps, rgb, alpha, depth, weights, rgb_on_rays, cnl_xyz,      \
                       cnl_rgb, cnl_weight)
    if (spl <= 1) HNRF_LAUNCH_COMPOSITE(1);
    else if (spl <= 2) HNRF_LAUNCH_COMPOSITE(2);
    else if (spl <= 4) HNRF_LAUNCH_COMPOSITE(4);
    else HNRF_LAUNCH_COMPOSITE(8);
#undef HNRF_LAUNCH_COMPOSITE
    return check_launch("hnrf_composite_fwd");
}
