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

extern "C" int hnrf_abi_version(void) { return 13; }
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

// ---------------------------------------------------------------------------------------------------------------
// Whole frame: Network._batchify_rays (network.py:330-352) over _render_rays -- every ray chunk through K1..K4, the
// per-ray / per-sample results written straight into whole-frame buffers.  With a side stream the LBS warp (K1, an
// L2-gather kernel without LDS) of chunk i+1 runs while the matrix-bound MLP kernels of chunk i own the CUs: K1 leaves
// the critical path (3 % of a 512x512x128 frame).  Two workspaces alternate; the caller provides the events.
extern "C" size_t hnrf_render_frame_workspace_bytes(int64_t chunk, int S) { return 2 * hnrf_render_workspace_bytes(chunk, S); }

extern "C" int hnrf_render_frame_fwd(const float* rays_o, const float* rays_d, const float* near, const float* far,
                                     const float* t_rand, const float* motion_Rs, const float* motion_Ts,
                                     const float* vol, const float* bbox_min, const float* bbox_scale,
                                     const float* hann_w, const void* nr_packed, const void* cnl_packed,
                                     const float* bgcolor, int mode, float cull_eps, int64_t N, int S, int B, int G,
                                     int64_t chunk, void* workspace, size_t workspace_bytes, float* rgb, float* alpha,
                                     float* depth, float* weights_on_rays, float* rgb_on_rays, float* cnl_xyz,
                                     float* cnl_rgb, float* cnl_weight, float* xyz_on_rays, float* bmw, float* offsets,
                                     void* side_stream, void* const* events, void* const* mlp_events, void* stream) {
    HNRF_REQUIRE(workspace && cnl_packed && rgb && alpha && depth, HNRF_E_ARG, "hnrf_render_frame_fwd: null pointer");
    HNRF_REQUIRE(N >= 0 && chunk >= 1 && S >= 2, HNRF_E_ARG, "hnrf_render_frame_fwd: bad dims");
    HNRF_REQUIRE(((uintptr_t)workspace & 255) == 0, HNRF_E_ARG, "hnrf_render_frame_fwd: workspace must be 256-byte aligned");
    const int64_t cr = chunk < N ? chunk : (N > 0 ? N : 1);
    const size_t ws_one = hnrf_render_workspace_bytes(cr, S);
    HNRF_REQUIRE(workspace_bytes >= 2 * ws_one, HNRF_E_WORKSPACE, "hnrf_render_frame_fwd: workspace %zu < %zu bytes",
                 workspace_bytes, 2 * ws_one);
    HNRF_REQUIRE(nr_packed == nullptr || hann_w != nullptr, HNRF_E_ARG, "hnrf_render_frame_fwd: hann_w missing");
    const bool diag = weights_on_rays != nullptr;
    HNRF_REQUIRE(!diag || (rgb_on_rays && cnl_xyz && cnl_rgb && cnl_weight && xyz_on_rays && bmw && offsets), HNRF_E_ARG,
                 "hnrf_render_frame_fwd: the eight diagnostic outputs go together");
    HNRF_REQUIRE(!diag || cull_eps == 0.f, HNRF_E_UNSUPPORTED, "hnrf_render_frame_fwd: sample culling exists in the lean form only");
    HNRF_REQUIRE(side_stream == nullptr || events != nullptr, HNRF_E_ARG, "hnrf_render_frame_fwd: a side stream needs 5 events");
    if (N == 0) return HNRF_OK;
    hipStream_t st = (hipStream_t)stream, sd = side_stream ? (hipStream_t)side_stream : st;
    const bool two = side_stream != nullptr && side_stream != stream;
    const int64_t nchunk = (N + chunk - 1) / chunk;
    const bool cull = cull_eps > 0.f;
    struct Carve { float *z, *mask, *x_skel, *xyz, *raw; int *idx, *count; };
    auto carve = [&](int slot, int64_t R) {
        const size_t P = (size_t)R * (size_t)S;
        char* w = (char*)workspace + (size_t)slot * ws_one;
        Carve c;
        c.z = (float*)w;       w += align256(P * 4);
        c.mask = (float*)w;    w += align256(P * 4);
        c.x_skel = (float*)w;  w += align256(P * 12);
        c.xyz = (float*)w;     w += align256(P * 12);
        c.raw = (float*)w;     w += align256(P * 16);
        c.idx = (int*)w;       w += align256(P * 4);
        c.count = (int*)w;
        return c;
    };
    auto warp = [&](int64_t i) {                                  // K1 of chunk i on the side stream
        const int64_t r0 = i * chunk, R = (N - r0 < chunk) ? N - r0 : chunk;
        const Carve c = carve((int)(i & 1), R);
        return hnrf_sample_warp_fwd(rays_o + 3 * r0, rays_d + 3 * r0, near + r0, far + r0, t_rand ? t_rand + r0 * S : nullptr,
                                    motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, R, S, B, G, c.z, c.x_skel, c.mask,
                                    bmw ? bmw + r0 * S * B : nullptr, sd);
    };
#define HNRF_HIP(call)                                                          \
    do {                                                                        \
        if ((call) != hipSuccess) {                                             \
            set_error("hnrf_render_frame_fwd: %s failed", #call);               \
            return HNRF_E_LAUNCH;                                               \
        }                                                                       \
    } while (0)
    hipEvent_t ev_in = nullptr, ev_k1[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    if (two) {
        ev_in = (hipEvent_t)events[0];
        ev_k1[0] = (hipEvent_t)events[1]; ev_k1[1] = (hipEvent_t)events[2];
        ev_done[0] = (hipEvent_t)events[3]; ev_done[1] = (hipEvent_t)events[4];
        HNRF_HIP(hipEventRecord(ev_in, st));                      // the side stream starts behind everything queued so far
        HNRF_HIP(hipStreamWaitEvent(sd, ev_in, 0));
    }
    int rc = warp(0);
    if (rc) return rc;
    if (two) HNRF_HIP(hipEventRecord(ev_k1[0], sd));
    for (int64_t i = 0; i < nchunk; ++i) {
        const int64_t r0 = i * chunk, R = (N - r0 < chunk) ? N - r0 : chunk;
        const size_t P = (size_t)R * (size_t)S;
        const Carve c = carve((int)(i & 1), R);
        if (i + 1 < nchunk) {                                     // next chunk's K1: its workspace was last read by chunk i-1
            if (two && i >= 1) HNRF_HIP(hipStreamWaitEvent(sd, ev_done[(i + 1) & 1], 0));
            if (two) {
                if ((rc = warp(i + 1))) return rc;
                HNRF_HIP(hipEventRecord(ev_k1[(i + 1) & 1], sd));
            }
        }
        if (two) HNRF_HIP(hipStreamWaitEvent(st, ev_k1[i & 1], 0));
        if (cull && (rc = hnrf_compact_samples(c.mask, cull_eps, (int64_t)P, c.idx, c.count, st))) return rc;
        const int* ci = cull ? c.idx : nullptr;
        const int* cc = cull ? c.count : nullptr;
        float* xyz = diag ? xyz_on_rays + r0 * S * 3 : c.xyz;
        const float* cnl_in = c.x_skel;
        // f16-range guard (hnrf.h): every chunk, none, or the one chunk the caller's rotating index names
        int cmode = mode & (HNRF_MLP_ARITH_MASK | HNRF_MLP_NO_RANGE_GUARD);
        if ((mode & HNRF_MLP_GUARD_ONE_CHUNK) && (int64_t)((unsigned)mode >> 16) % nchunk != i) cmode |= HNRF_MLP_NO_RANGE_GUARD;
        if (nr_packed) {
            if ((rc = hnrf_nonrigid_fwd_sparse(c.x_skel, hann_w, nr_packed, cmode, (int64_t)P, ci, cc, xyz,
                                               diag ? offsets + r0 * S * 3 : nullptr, st))) return rc;
            cnl_in = xyz;
        } else if (diag) {                                        // network.py:276-277: xyz = x_skel, offsets = 0
            HNRF_HIP(hipMemcpyAsync(xyz, c.x_skel, P * 12, hipMemcpyDeviceToDevice, st));
            HNRF_HIP(hipMemsetAsync(offsets + r0 * S * 3, 0, P * 12, st));
        }
        if (mlp_events) (void)hipEventRecord((hipEvent_t)mlp_events[2 * i], st);
        rc = hnrf_canonical_fwd_sparse(cnl_in, cnl_packed, cmode, (int64_t)P, ci, cc, c.raw, st);
        if (mlp_events) (void)hipEventRecord((hipEvent_t)mlp_events[2 * i + 1], st);
        if (rc) return rc;
        if ((rc = hnrf_composite_fwd(c.raw, c.mask, c.z, rays_d + 3 * r0, diag ? cnl_in : nullptr, bgcolor, R, S,
                                     cull ? cull_eps : 0.f, rgb + 3 * r0, alpha + r0, depth + r0,
                                     diag ? weights_on_rays + r0 * S : nullptr, diag ? rgb_on_rays + r0 * S * 3 : nullptr,
                                     diag ? cnl_xyz + 3 * r0 : nullptr, diag ? cnl_rgb + 3 * r0 : nullptr,
                                     diag ? cnl_weight + r0 : nullptr, st))) return rc;
        if (two) HNRF_HIP(hipEventRecord(ev_done[i & 1], st));
        if (!two && i + 1 < nchunk && (rc = warp(i + 1))) return rc;   // single stream: plain sequence
    }
#undef HNRF_HIP
    return HNRF_OK;
}
