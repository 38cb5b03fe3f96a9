// Error channel, version, and the whole-path entry point (K1 -> K2 -> K3 -> K4).
#include <stdarg.h>
#include <string.h>

#include "hnrf_common.h"

namespace hnrf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace hnrf

using namespace hnrf;

extern "C" int hnrf_abi_version(void) { return 8; }
extern "C" const char* hnrf_last_error(void) { return g_err; }

// workspace carve: z_vals[P] | mask[P] | x_skel[3P] | xyz[3P] | raw[4P] | idx[P] | count
extern "C" size_t hnrf_render_workspace_bytes(int64_t R, int S) {
    if (R < 0 || S < 0) return 0;
    const size_t P = (size_t)R * (size_t)S;
    return align256(P * 4) * 2 + align256(P * 12) * 2 + align256(P * 16) + align256(P * 4) + 256;
}

extern "C" int hnrf_render_rays_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                                    const float* t_rand, const float* motion_Rs, const float* motion_Ts,
                                    const float* vol, const float* bbox_min, const float* bbox_scale,
                                    const float* hann_w, const void* nr_packed, const void* cnl_packed,
                                    const float* bgcolor, int mode, float cull_eps, int64_t R, int S, int B, int G,
                                    void* workspace, size_t workspace_bytes, float* rgb, float* alpha, float* depth,
                                    void* ev_mlp_start, void* ev_mlp_stop, void* stream) {
    HNRF_REQUIRE(workspace && cnl_packed, HNRF_E_ARG, "hnrf_render_rays_fwd: null workspace / canonical weights");
    HNRF_REQUIRE(((uintptr_t)workspace & 255) == 0, HNRF_E_ARG, "hnrf_render_rays_fwd: workspace must be 256-byte aligned");
    HNRF_REQUIRE(workspace_bytes >= hnrf_render_workspace_bytes(R, S), HNRF_E_WORKSPACE,
                 "hnrf_render_rays_fwd: workspace %zu < %zu bytes", workspace_bytes, hnrf_render_workspace_bytes(R, S));
    HNRF_REQUIRE(nr_packed == nullptr || hann_w != nullptr, HNRF_E_ARG, "hnrf_render_rays_fwd: hann_w missing");
    const size_t P = (size_t)R * (size_t)S;
    char* w = (char*)workspace;
    float* z_vals = (float*)w;  w += align256(P * 4);
    float* mask = (float*)w;    w += align256(P * 4);
    float* x_skel = (float*)w;  w += align256(P * 12);
    float* xyz = (float*)w;     w += align256(P * 12);
    float* raw = (float*)w;     w += align256(P * 16);
    int* idx = (int*)w;         w += align256(P * 4);
    int* count = (int*)w;
    const bool cull = cull_eps > 0.f;
    int rc = hnrf_sample_warp_fwd(rays_o, rays_d, near, far, t_rand, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale,
                                  R, S, B, G, z_vals, x_skel, mask, nullptr, stream);
    if (rc) return rc;
    if (cull) {
        rc = hnrf_compact_samples(mask, cull_eps, (int64_t)P, idx, count, stream);
        if (rc) return rc;
    }
    const int* ci = cull ? idx : nullptr;
    const int* cc = cull ? count : nullptr;
    const float* cnl_in = x_skel;
    if (nr_packed) {
        rc = hnrf_nonrigid_fwd_sparse(x_skel, hann_w, nr_packed, mode, (int64_t)P, ci, cc, xyz, nullptr, stream);
        if (rc) return rc;
        cnl_in = xyz;
    }
    if (ev_mlp_start) (void)hipEventRecord((hipEvent_t)ev_mlp_start, (hipStream_t)stream);
    rc = hnrf_canonical_fwd_sparse(cnl_in, cnl_packed, mode, (int64_t)P, ci, cc, raw, stream);
    if (ev_mlp_stop) (void)hipEventRecord((hipEvent_t)ev_mlp_stop, (hipStream_t)stream);
    if (rc) return rc;
    return hnrf_composite_fwd(raw, mask, z_vals, rays_d, nullptr, bgcolor, R, S, cull ? cull_eps : 0.f, rgb, alpha,
                              depth, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
