// A host with no Python and no torch in the process: drives one ray chunk through the C ABI of libhnrf.so
// (include/hnrf.h) the way INTEGRATION.md section 2 says a C/C++ caller does -- hipMalloc'd buffers, one stream,
// packed weight images, hnrf_render_rays_fwd -- and dumps rgb/alpha/depth for tests/test_gpu_cabi_host.py, which
// regenerates the same inputs (the LCG below) and compares with the Python binding's result.
//   usage: cabi_host R S mode out.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/hnrf.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define HN(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, hnrf_last_error()); return 3; } } while (0)

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static float lcg() {                      // uniform in [0, 1), 24 bits: reproduced bit-for-bit in the test
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((lcg_state >> 40) & 0xFFFFFF) / 16777216.0f;
}
static std::vector<float> uni(size_t n, float lo, float hi) {
    std::vector<float> v(n);
    for (auto& x : v) x = lo + (hi - lo) * lcg();
    return v;
}
template <class T> static T* up(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: cabi_host R S mode out.bin\n"); return 1; }
    const int64_t R = atoll(argv[1]);
    const int S = atoi(argv[2]), mode = atoi(argv[3]);
    const int B = 24, G = 32;
    if (hnrf_abi_version() < 7) { fprintf(stderr, "old libhnrf\n"); return 1; }
    hipStream_t st;
    CK(hipStreamCreate(&st));

    // inputs, in this order (the test draws them in the same order)
    auto rays_o = uni(R * 3, -0.2f, 0.2f); for (int64_t r = 0; r < R; ++r) rays_o[r * 3 + 2] -= 3.0f;
    auto rays_d = uni(R * 3, -0.25f, 0.25f); for (int64_t r = 0; r < R; ++r) rays_d[r * 3 + 2] = 1.0f;
    auto near = uni(R, 2.0f, 2.3f), far = uni(R, 3.6f, 4.0f);
    std::vector<float> Rs(B * 9), Ts = uni(B * 3, -0.1f, 0.1f);
    for (int b = 0; b < B; ++b) {                                  // small rotations about z + scale ~1
        const float a = 0.3f * (lcg() - 0.5f), c = std::cos(a), s = std::sin(a);
        const float m[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) Rs[b * 9 + k] = m[k];
    }
    auto vol = uni((size_t)(B + 1) * G * G * G, 0.0f, 0.08f);
    const std::vector<float> bmin = {-1.2f, -1.2f, -1.2f}, bscale = {2.0f / 2.4f, 2.0f / 2.4f, 2.0f / 2.4f};
    const std::vector<float> hann(6, 1.0f), bg = {255.f, 128.f, 0.f};
    auto cond = uni(69, -0.3f, 0.3f);
    const int nr_out[7] = {128, 128, 128, 128, 128, 128, 3}, nr_in[7] = {105, 128, 128, 128, 164, 128, 128};
    const int cn_out[9] = {256, 256, 256, 256, 256, 256, 256, 256, 4}, cn_in[9] = {63, 256, 256, 256, 256, 319, 256, 256, 256};
    std::vector<const float*> nw(7), nb(7), cw(9), cb(9);
    for (int l = 0; l < 7; ++l) {
        const float a = std::sqrt(6.0f / (nr_in[l] + nr_out[l])) * 1.4f;
        nw[l] = up(uni((size_t)nr_out[l] * nr_in[l], -a, a));
        nb[l] = up(uni(nr_out[l], -0.05f, 0.05f));
    }
    for (int l = 0; l < 9; ++l) {
        const float a = std::sqrt(6.0f / (cn_in[l] + cn_out[l])) * 1.4f;
        cw[l] = up(uni((size_t)cn_out[l] * cn_in[l], -a, a));
        cb[l] = up(uni(cn_out[l], -0.05f, 0.05f));
    }
    float *d_ro = up(rays_o), *d_rd = up(rays_d), *d_near = up(near), *d_far = up(far), *d_Rs = up(Rs), *d_Ts = up(Ts),
          *d_vol = up(vol), *d_bmin = up(bmin), *d_bscale = up(bscale), *d_hann = up(hann), *d_bg = up(bg), *d_cond = up(cond);

    void *nr_packed, *cn_packed, *ws;
    CK(hipMalloc(&nr_packed, hnrf_nonrigid_packed_bytes(mode)));
    CK(hipMalloc(&cn_packed, hnrf_canonical_packed_bytes(mode)));
    const size_t ws_bytes = hnrf_render_workspace_bytes(R, S);
    CK(hipMalloc(&ws, ws_bytes));
    HN(hnrf_nonrigid_pack(nw.data(), nb.data(), d_cond, mode, nr_packed, st));
    HN(hnrf_canonical_pack(cw.data(), cb.data(), mode, cn_packed, st));
    float *rgb, *alpha, *depth;
    CK(hipMalloc((void**)&rgb, R * 3 * sizeof(float)));
    CK(hipMalloc((void**)&alpha, R * sizeof(float)));
    CK(hipMalloc((void**)&depth, R * sizeof(float)));
    HN(hnrf_render_rays_fwd(d_ro, d_rd, d_near, d_far, nullptr, d_Rs, d_Ts, d_vol, d_bmin, d_bscale, d_hann, nr_packed,
                            cn_packed, d_bg, mode, 0.0f, R, S, B, G, ws, ws_bytes, rgb, alpha, depth, nullptr, nullptr, st));
    // the error channel: a bad argument comes back as a code + message, nothing throws or exits
    if (hnrf_render_rays_fwd(d_ro, d_rd, d_near, d_far, nullptr, d_Rs, d_Ts, d_vol, d_bmin, d_bscale, d_hann, nr_packed,
                             cn_packed, d_bg, mode, 0.0f, R, S, B, G, ws, 16, rgb, alpha, depth, nullptr, nullptr, st) == 0) {
        fprintf(stderr, "undersized workspace was accepted\n");
        return 4;
    }
    CK(hipStreamSynchronize(st));
    std::vector<float> out(R * 5);
    CK(hipMemcpy(out.data(), rgb, R * 3 * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data() + R * 3, alpha, R * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(out.data() + R * 4, depth, R * sizeof(float), hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(out.data(), sizeof(float), out.size(), f) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[4]); return 5; }
    fclose(f);
    printf("cabi_host: R=%lld S=%d mode=%d ok, last error after the refused call: %s\n", (long long)R, S, mode, hnrf_last_error());
    return 0;
}
