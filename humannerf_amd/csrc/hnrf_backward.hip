// Backward kernels of the per-sample stages around the MLPs (training):
//   K4' composite backward   -- d(rgb, alpha, depth) -> d raw, d fg_mask
//   PE' positional-encoding backward -- d PE -> d xyz
//   K1' LBS-warp backward    -- d x_skel, d fg_mask -> d volume (atomics), d motion_Rs, d motion_Ts
// They restate what torch.autograd derives for the reference's forward ops
// (network.py:355-388, fourier.py / hannw_fourier.py embed, network.py:392-444 incl. the
// grid_sample gradient w.r.t. both the volume and the sampling position).  The MLP
// backward itself (dX chains, dW = dZ^T X) is in hnrf_mlp_bwd.hip / hnrf_mlp_f16.hip.
#include "hnrf_common.h"

namespace hnrf {

// ----------------------------------------------------------------------------- K4'
// One wave per ray, SPL samples per lane (same layout as composite_kernel).
//   w_i = a_i T_i,  T_i = prod_{j<i} t_j,  t_j = 1 - a_j + 1e-10
//   out = sum w_i c_i + (1 - sum w_i) bg,  depth = sum w_i z_i,  acc = sum w_i
//   dL/dw_i = g_rgb.(c_i - bg) + g_acc + g_depth z_i
//   dL/da_i = T_i dL/dw_i - (sum_{j>i} dL/dw_j w_j) / t_i
template <int SPL>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float4* __restrict__ raw, const float* __restrict__ fg_mask, const float* __restrict__ z_vals,
    const float* __restrict__ rays_d, const float* __restrict__ bgcolor, const float* __restrict__ g_rgb,
    const float* __restrict__ g_alpha, const float* __restrict__ g_depth, int64_t R, int S,
    float4* __restrict__ d_raw, float* __restrict__ d_mask) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float dx = rays_d[ray * 3 + 0], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
    const int64_t base = ray * S;
    const float bg0 = bgcolor[0] / 255.f, bg1 = bgcolor[1] / 255.f, bg2 = bgcolor[2] / 255.f;
    const float gr = g_rgb[ray * 3 + 0], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    const float ga = g_alpha ? g_alpha[ray] : 0.f, gd = g_depth ? g_depth[ray] : 0.f;

    float zv[SPL], al[SPL], ee[SPL], dl[SPL], cr[SPL], cg[SPL], cb[SPL], mk[SPL];
    float4 rw[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const int sc = s < S ? s : S - 1;
        rw[i] = raw[base + sc];
        mk[i] = s < S ? fg_mask[base + sc] : 0.f;
        zv[i] = z_vals[base + sc];
    }
    const float znext_lane = __shfl_down(zv[0], 1, 64);
    float tl = 1.f;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        const float zn = (i + 1 < SPL) ? zv[(i + 1 < SPL) ? i + 1 : i] : znext_lane;
        float dist = (s >= S - 1) ? 1e10f : (zn - zv[i]);
        dist *= dnorm;
        dl[i] = dist;
        const float sig = fmaxf(rw[i].w, 0.f);
        ee[i] = expf(-sig * dist);
        al[i] = (1.0f - ee[i]) * mk[i];
        cr[i] = 1.0f / (1.0f + expf(-rw[i].x));
        cg[i] = 1.0f / (1.0f + expf(-rw[i].y));
        cb[i] = 1.0f / (1.0f + expf(-rw[i].z));
        if (s < S) tl *= (1.0f - al[i] + 1e-10f);
    }
    float incl = tl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= o;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.f;
    // per-sample T, w, dL/dw and the lane-local sum of dL/dw * w
    float Ti[SPL], gw[SPL], wv[SPL];
    float loc = 0.f;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane * SPL + i;
        Ti[i] = T;
        wv[i] = al[i] * T;
        T *= (1.0f - al[i] + 1e-10f);
        gw[i] = gr * (cr[i] - bg0) + gg * (cg[i] - bg1) + gb * (cb[i] - bg2) + ga + gd * zv[i];
        if (s >= S) { gw[i] = 0.f; wv[i] = 0.f; }
        loc += gw[i] * wv[i];
    }
    // exclusive suffix sum over lanes of `loc`
    float suf = loc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_down(suf, off, 64);
        if (lane + off < 64) suf += o;
    }
    float after = suf - loc;   // sum over later lanes
#pragma unroll
    for (int i = SPL - 1; i >= 0; --i) {
        const int s = lane * SPL + i;
        if (s < S) {
            const float t = 1.0f - al[i] + 1e-10f;
            const float da = Ti[i] * gw[i] - after / t;
            after += gw[i] * wv[i];
            const float dsig = (rw[i].w > 0.f) ? da * mk[i] * dl[i] * ee[i] : 0.f;
            float4 o;
            o.x = wv[i] * gr * cr[i] * (1.0f - cr[i]);
            o.y = wv[i] * gg * cg[i] * (1.0f - cg[i]);
            o.z = wv[i] * gb * cb[i] * (1.0f - cb[i]);
            o.w = dsig;
            d_raw[base + s] = o;
            d_mask[base + s] = da * (1.0f - ee[i]);
        }
    }
}

// ----------------------------------------------------------------------------- PE'
// d xyz[p][ax] (+)= [include_input] g[p][ax] + sum_k w_k 2^k (cos(2^k x) g_sin - sin(2^k x) g_cos)
// g layout = the embedders' output order; NB bands; hann == nullptr -> all weights 1.
__global__ void pe_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ hann,
                              int64_t P, int NB, int include_input, int accumulate, float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * 3) return;
    const int64_t p = i / 3;
    const int ax = (int)(i - p * 3);
    const int C = (include_input ? 3 : 0) + 6 * NB;
    const float* gp = g + p * C;
    const float xv = x[i];
    float acc = include_input ? gp[ax] : 0.f;
    const int o0 = include_input ? 3 : 0;
    for (int k = 0; k < NB; ++k) {
        const float f = (float)(1 << k);
        float sv, cv;
        sincosf(xv * f, &sv, &cv);
        const float w = hann ? hann[k] : 1.0f;
        acc += w * f * (cv * gp[o0 + 6 * k + ax] - sv * gp[o0 + 6 * k + 3 + ax]);
    }
    dx[i] = accumulate ? dx[i] + acc : acc;
}

// ----------------------------------------------------------------------------- K1'
// grid = (sample blocks, bones).  x_skel = A / den, A = sum_b w_b pos_b, den = max(mask, 1e-4),
// mask = sum_b w_b (both saved by the forward), so bone b only needs its own w_b, pos_b:
//   dL/dw_b   = gA . pos_b + g_ws,   gA = g_x / den,
//   g_ws      = g_mask + [mask >= 1e-4] * (-(g_x . x_skel) / den)
//   dL/dpos_b = w_b gA + dL/dw_b * grad_pos trilinear(vol_b)        (grid_sample's grid gradient)
//   dL/dvol_b[corner] += dL/dw_b * corner weight                     (global float atomics)
//   dL/dR_b += dL/dpos_b (x) x,   dL/dT_b += dL/dpos_b               (block reduction, 12 atomics)
// NTH threads per block.  The LDS form (one 128-KiB grid = one block per CU) runs 16 waves per block (see the launcher).
// lane l <- lane l + N of the same 16-lane row (0 beyond the row)
template <int N>
__device__ __forceinline__ int dpp_row_shl(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v) { return __int_as_float(dpp_row_shl<N>(__float_as_int(v))); }

template <bool LDS_VOL, int NTH>
__global__ __launch_bounds__(NTH) void sample_warp_bwd_kernel(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ z_vals,
    const float* __restrict__ Rs, const float* __restrict__ Ts, const float* __restrict__ vol,
    const float* __restrict__ bbox_min, const float* __restrict__ bbox_scale, const float* __restrict__ x_skel,
    const float* __restrict__ fg_mask, const float* __restrict__ g_x, const float* __restrict__ g_mask, int64_t P,
    int S, int G, float* __restrict__ d_vol, float* __restrict__ d_Rs, float* __restrict__ d_Ts) {
    const int b = blockIdx.y;
    const float* Rb = Rs + b * 9;
    const float* Tb = Ts + b * 3;
    const float R0 = Rb[0], R1 = Rb[1], R2 = Rb[2], R3 = Rb[3], R4 = Rb[4], R5 = Rb[5], R6 = Rb[6], R7 = Rb[7],
                R8 = Rb[8];
    const float T0 = Tb[0], T1 = Tb[1], T2 = Tb[2];
    const float bmx = bbox_min[0], bmy = bbox_min[1], bmz = bbox_min[2];
    const float bsx = bbox_scale[0], bsy = bbox_scale[1], bsz = bbox_scale[2];
    const float gm1 = (float)(G - 1);
    const int GG = G * G;
    const float* vb = vol + (size_t)b * G * GG;
    float* dvb = d_vol + (size_t)b * G * GG;
    // LDS_VOL: this block accumulates its bone's whole G^3 gradient grid in LDS (128 KiB at G = 32;
    // samples of one ray chunk hammer the same few voxels, global float atomics on one line run
    // ~14x below their spread rate) and flushes the touched voxels once at the end.
    extern __shared__ float lvol[];
    if (LDS_VOL) {
        for (int i = threadIdx.x; i < G * GG; i += NTH) lvol[i] = 0.f;
        __syncthreads();
    }

    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;

    // One block per CU (the 128-KiB grid fills the LDS): the loop is bound by the NUMBER of instructions per (sample,
    // bone) pair, not by latency (4 samples per trip with all loads batched: no change).
    // Hence: branch-free corners (per-axis validity folded into the corner weights, clamped addresses), three index
    // products per sample instead of one per corner, one reciprocal instead of three divisions, 32-bit sample / ray
    // arithmetic (P < 2^31 is checked by the launcher).  900 -> ~350 instructions per pair: 0.74 -> 0.44 ms with the
    // atomics idle (all-zero gradient).  With real gradients the 8 ds_add_f32 per pair dominate (0.87 ms): the 64 lanes
    // of a wave are consecutive samples of one ray, ~4.5 of them per voxel, and same-address LDS atomics serialise.
    // LDS float atomics cost ~3.4 cycles per ACTIVE lane and more when lanes share an address (measured: 0.43 ms of
    // 0.87; with a quarter of the lanes active they vanish).  The lanes of a wave are consecutive samples of a ray, ~4.5
    // per voxel cell: runs of lanes in the same cell are folded first (segments inside aligned groups of 8 lanes, three
    // DPP doubling steps shared by the 8 corners) and only the head of a segment issues the atomic.  Every lane runs
    // every trip (dead lanes carry zeros) so that the cross-lane reads see defined data.
    const unsigned stride = gridDim.x * NTH;
    const int lane = threadIdx.x & 63;
    for (unsigned pw = blockIdx.x * NTH + (threadIdx.x & ~63u); pw < (unsigned)P; pw += stride) {
        const bool live = pw + lane < (unsigned)P;
        const unsigned p = live ? pw + lane : pw;
        const unsigned r = p / (unsigned)S;
        const float z = z_vals[p];
        const float px = rays_o[r * 3 + 0] + rays_d[r * 3 + 0] * z;
        const float py = rays_o[r * 3 + 1] + rays_d[r * 3 + 1] * z;
        const float pz = rays_o[r * 3 + 2] + rays_d[r * 3 + 2] * z;
        const float m = fg_mask[p];
        const float inv_den = 1.0f / fmaxf(m, 0.0001f);
        const float* gxp = g_x + (size_t)p * 3;
        const float* xsp = x_skel + (size_t)p * 3;
        const float gx0 = gxp[0], gx1 = gxp[1], gx2 = gxp[2];
        const float gA0 = gx0 * inv_den, gA1 = gx1 * inv_den, gA2 = gx2 * inv_den;
        float gws = g_mask[p];
        if (m >= 0.0001f) gws -= (gx0 * xsp[0] + gx1 * xsp[1] + gx2 * xsp[2]) * inv_den;
        const float alive = live ? 1.f : 0.f;

        const float qx = R0 * px + R1 * py + R2 * pz + T0;
        const float qy = R3 * px + R4 * py + R5 * pz + T1;
        const float qz = R6 * px + R7 * py + R8 * pz + T2;
        const float ix = (((qx - bmx) * bsx - 1.0f) + 1.0f) * 0.5f * gm1;
        const float iy = (((qy - bmy) * bsy - 1.0f) + 1.0f) * 0.5f * gm1;
        const float iz = (((qz - bmz) * bsz - 1.0f) + 1.0f) * 0.5f * gm1;
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fminf(fmaxf(fx0, -2.0f), gm1 + 1.0f);
        const int y0 = (int)fminf(fmaxf(fy0, -2.0f), gm1 + 1.0f);
        const int z0 = (int)fminf(fmaxf(fz0, -2.0f), gm1 + 1.0f);
        // per axis: weight of the lower / upper corner (zero when that corner lies outside the grid: grid_sample's zero
        // padding), its derivative sign folded in later, and its clamped coordinate
        const bool vx0 = x0 >= 0 && x0 < G, vx1 = x0 + 1 >= 0 && x0 + 1 < G;
        const bool vy0 = y0 >= 0 && y0 < G, vy1 = y0 + 1 >= 0 && y0 + 1 < G;
        const bool vz0 = z0 >= 0 && z0 < G, vz1 = z0 + 1 >= 0 && z0 + 1 < G;
        const float wxa[2] = {vx0 ? (fx0 + 1.0f) - ix : 0.f, vx1 ? ix - fx0 : 0.f};
        const float wya[2] = {vy0 ? (fy0 + 1.0f) - iy : 0.f, vy1 ? iy - fy0 : 0.f};
        const float wza[2] = {vz0 ? (fz0 + 1.0f) - iz : 0.f, vz1 ? iz - fz0 : 0.f};
        const float sxa[2] = {vx0 ? -1.f : 0.f, vx1 ? 1.f : 0.f};     // d weight / d ix of a valid corner
        const float sya[2] = {vy0 ? -1.f : 0.f, vy1 ? 1.f : 0.f};
        const float sza[2] = {vz0 ? -1.f : 0.f, vz1 ? 1.f : 0.f};
        const int xc[2] = {min(max(x0, 0), G - 1), min(max(x0 + 1, 0), G - 1)};
        const int yo[2] = {min(max(y0, 0), G - 1) * G, min(max(y0 + 1, 0), G - 1) * G};
        const int zo[2] = {min(max(z0, 0), G - 1) * GG, min(max(z0 + 1, 0), G - 1) * GG};
        float w = 0.f, dwx = 0.f, dwy = 0.f, dwz = 0.f;   // value and d/d(ix,iy,iz)
        const float dLdw_base = (gA0 * qx + gA1 * qy + gA2 * qz + gws) * alive;
        // segments: lanes l, l+1, .. of one aligned group of 8 with the same cell (x0, y0, z0 are within [-2, G + 1])
        const int key = live ? ((z0 + 2) << 22) | ((y0 + 2) << 11) | (x0 + 2) : -1 - lane;
        const int key_next = dpp_row_shl<1>(key), key_prev = __builtin_amdgcn_update_dpp(0, key, 0x110 + 1, 0xF, 0xF, true);
        const bool same1 = key_next == key && (lane & 7) < 7;
        const int s1_next = dpp_row_shl<1>((int)same1);
        const bool same2 = same1 && s1_next != 0 && (lane & 7) < 6;
        const int s2_next = dpp_row_shl<2>((int)same2);
        const bool same4 = same2 && s2_next != 0 && (lane & 7) < 4;
        const bool head = (lane & 7) == 0 || key_prev != key;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = vb[zo[c >> 2] + yo[(c >> 1) & 1] + xc[c & 1]];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ox = c & 1, oy = (c >> 1) & 1, oz = c >> 2;
            const float wyz = wya[oy] * wza[oz], wxz = wxa[ox] * wza[oz], wxy = wxa[ox] * wya[oy];
            const float www = wxa[ox] * wyz;                      // 0 for a corner outside the grid
            w += v[c] * www;
            dwx += v[c] * sxa[ox] * wyz;
            dwy += v[c] * sya[oy] * wxz;
            dwz += v[c] * sza[oz] * wxy;
            float contrib = dLdw_base * www;
            // (the cross-lane read first, the select after: inside the conditional only the lanes that take it would
            // execute the DPP move, and a DPP read from a lane that is masked off returns 0)
            const float t1 = dpp_row_shl<1>(contrib);
            contrib += same1 ? t1 : 0.f;
            const float t2 = dpp_row_shl<2>(contrib);
            contrib += same2 ? t2 : 0.f;
            const float t4 = dpp_row_shl<4>(contrib);
            contrib += same4 ? t4 : 0.f;
            if (head && contrib != 0.f) {
                const int idx = zo[oz] + yo[oy] + xc[ox];
                if (LDS_VOL) atomicAdd(lvol + idx, contrib);
                else atomicAdd(dvb + idx, contrib);
            }
        }
        // d pos = w gA + dL/dw * grad_pos(w);  d ix / d qx = bsx * 0.5 * (G-1)
        const float dqx = (w * gA0 + dLdw_base * dwx * (bsx * 0.5f * gm1)) * alive;
        const float dqy = (w * gA1 + dLdw_base * dwy * (bsy * 0.5f * gm1)) * alive;
        const float dqz = (w * gA2 + dLdw_base * dwz * (bsz * 0.5f * gm1)) * alive;
        acc[0] += dqx * px; acc[1] += dqx * py; acc[2] += dqx * pz;
        acc[3] += dqy * px; acc[4] += dqy * py; acc[5] += dqy * pz;
        acc[6] += dqz * px; acc[7] += dqz * py; acc[8] += dqz * pz;
        acc[9] += dqx; acc[10] += dqy; acc[11] += dqz;
    }
    if (LDS_VOL) {
        __syncthreads();
        for (int i = threadIdx.x; i < G * GG; i += NTH) {
            const float v = lvol[i];
            if (v != 0.f) atomicAdd(dvb + i, v);
        }
    }
    // block reduction: wave butterflies, then one partial per wave through LDS
    __shared__ float red[NTH / 64][12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NTH / 64; ++wv) v += red[wv][threadIdx.x];
        if (threadIdx.x < 9) atomicAdd(d_Rs + b * 9 + threadIdx.x, v);
        else atomicAdd(d_Ts + b * 3 + (threadIdx.x - 9), v);
    }
}

}  // namespace hnrf

using namespace hnrf;

extern "C" int hnrf_composite_bwd(const float* raw, const float* fg_mask, const float* z_vals, const float* rays_d,
                                  const float* bgcolor, const float* g_rgb, const float* g_alpha,
                                  const float* g_depth, int64_t R, int S, float* d_raw, float* d_mask,
                                  void* stream) {
    HNRF_REQUIRE(raw && fg_mask && z_vals && rays_d && bgcolor && g_rgb && d_raw && d_mask, HNRF_E_ARG,
                 "hnrf_composite_bwd: null pointer");
    HNRF_REQUIRE(R >= 0 && S >= 2, HNRF_E_ARG, "hnrf_composite_bwd: bad dims");
    HNRF_REQUIRE(S <= 512, HNRF_E_UNSUPPORTED, "hnrf_composite_bwd: S=%d > 512 not built", S);
    HNRF_REQUIRE((((uintptr_t)raw | (uintptr_t)d_raw) & 15) == 0, HNRF_E_ARG, "hnrf_composite_bwd: raw/d_raw alignment");
    if (R == 0) return HNRF_OK;
    const int64_t blocks = (R + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    const int spl = (S + 63) / 64;
#define HNRF_LAUNCH_CB(N)                                                                                          \
    hipLaunchKernelGGL(composite_bwd_kernel<N>, dim3((unsigned)blocks), dim3(256), 0, st, (const float4*)raw,     \
                       fg_mask, z_vals, rays_d, bgcolor, g_rgb, g_alpha, g_depth, R, S, (float4*)d_raw, d_mask)
    if (spl <= 1) HNRF_LAUNCH_CB(1);
    else if (spl <= 2) HNRF_LAUNCH_CB(2);
    else if (spl <= 4) HNRF_LAUNCH_CB(4);
    else HNRF_LAUNCH_CB(8);
#undef HNRF_LAUNCH_CB
    return check_launch("hnrf_composite_bwd");
}

extern "C" int hnrf_pe_bwd(const float* x, const float* g, const float* hann_w, int64_t P, int n_bands,
                           int include_input, int accumulate, float* dx, void* stream) {
    HNRF_REQUIRE(x && g && dx, HNRF_E_ARG, "hnrf_pe_bwd: null pointer");
    HNRF_REQUIRE(P >= 0 && n_bands >= 1 && n_bands <= 16, HNRF_E_ARG, "hnrf_pe_bwd: bad dims");
    if (P == 0) return HNRF_OK;
    const int64_t n = P * 3;
    hipLaunchKernelGGL(pe_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, g,
                       hann_w, P, n_bands, include_input, accumulate, dx);
    return check_launch("hnrf_pe_bwd");
}

extern "C" int hnrf_sample_warp_bwd(const float* rays_o, const float* rays_d, const float* z_vals,
                                    const float* motion_Rs, const float* motion_Ts, const float* vol,
                                    const float* bbox_min, const float* bbox_scale, const float* x_skel,
                                    const float* fg_mask, const float* g_x_skel, const float* g_mask, int64_t R,
                                    int S, int B, int G, float* d_vol, float* d_Rs, float* d_Ts, void* stream) {
    HNRF_REQUIRE(rays_o && rays_d && z_vals && motion_Rs && motion_Ts && vol && bbox_min && bbox_scale && x_skel &&
                     fg_mask && g_x_skel && g_mask && d_vol && d_Rs && d_Ts,
                 HNRF_E_ARG, "hnrf_sample_warp_bwd: null pointer");
    HNRF_REQUIRE(R >= 0 && S >= 2 && B >= 1 && B <= 65535 && G >= 2 && G <= 1024, HNRF_E_ARG,
                 "hnrf_sample_warp_bwd: bad dims");
    HNRF_REQUIRE(R * (int64_t)S < (int64_t)1 << 31, HNRF_E_UNSUPPORTED, "hnrf_sample_warp_bwd: %lld samples (32-bit sample index)",
                 (long long)(R * (int64_t)S));
    hipStream_t st = (hipStream_t)stream;
    // d_vol covers the B bone channels only; the caller owns the (zero) background-channel gradient
    if (hipMemsetAsync(d_vol, 0, (size_t)B * G * G * G * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(d_Rs, 0, (size_t)B * 9 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(d_Ts, 0, (size_t)B * 3 * sizeof(float), st) != hipSuccess) {
        set_error("hnrf_sample_warp_bwd: memset failed");
        return HNRF_E_LAUNCH;
    }
    if (R == 0) return HNRF_OK;
    const int64_t P = R * (int64_t)S;
    int64_t blocks = (P + 255) / 256;
    const size_t lds = (size_t)G * G * G * sizeof(float);
    if (lds + 1024 <= 160 * 1024 && P >= 65536) {       // (+ the block reduction's 16 x 12 floats of static LDS)
        // one 128-KiB LDS grid per block -> 1 block per CU; ~2 waves of blocks over the chip
        int64_t bx = 512 / B;                        // blocks per bone
        if (bx < 1) bx = 1;
        if (bx > blocks) bx = blocks;
        // 16 waves per block (four per SIMD): with ONE wave per SIMD every one of the ~350 instructions per (sample, bone)
        // pair costs ~5 cycles of issue (profiles/r03_mfma_issue.txt); 0.529 ms (256 threads) -> 0.357 (512) -> 0.295 (1024).
        // (Before the atomics were folded, more waves lost to LDS-atomic contention: 0.90 -> 1.00 ms, round 2.)
        static const int nth = getenv("HNRF_K1B_THREADS") ? atoi(getenv("HNRF_K1B_THREADS")) : 1024;
#define HNRF_K1B(NTH_)                                                                                                   \
    do {                                                                                                                 \
        static unsigned long long lds_done = 0;                                                                          \
        if (int rc = reserve_lds((const void*)sample_warp_bwd_kernel<true, NTH_>, 150 * 1024, lds_done, "hnrf_sample_warp_bwd")) \
            return rc;                                                                                                   \
        hipLaunchKernelGGL((sample_warp_bwd_kernel<true, NTH_>), dim3((unsigned)bx, (unsigned)B), dim3(NTH_), lds, st, rays_o, \
                           rays_d, z_vals, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, x_skel, fg_mask, g_x_skel,  \
                           g_mask, P, S, G, d_vol, d_Rs, d_Ts);                                                          \
    } while (0)
        if (nth == 512) HNRF_K1B(512); else if (nth == 1024) HNRF_K1B(1024); else HNRF_K1B(256);
#undef HNRF_K1B
    } else {
        if (blocks > 1024) blocks = 1024;     // grid-stride: few, long blocks keep the 12-value reduction cheap
        hipLaunchKernelGGL((sample_warp_bwd_kernel<false, 256>), dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, st,
                           rays_o, rays_d, z_vals, motion_Rs, motion_Ts, vol, bbox_min, bbox_scale, x_skel, fg_mask,
                           g_x_skel, g_mask, P, S, G, d_vol, d_Rs, d_Ts);
    }
    return check_launch("hnrf_sample_warp_bwd");
}
