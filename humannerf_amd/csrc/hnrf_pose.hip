// Per-frame kinematics of the path: MotionBasisComputer.forward (core/utils/network_util.py:125-156) and its backward
// as ONE single-wave kernel each.
//
//   G_i = [R_i | T_i ; 0 0 0 1],  A_0 = G_0,  A_i = A_parent(i) G_i  (SMPL_PARENT, network_util.py:91-94),
//   F_i = C_i A_i^-1  (C = cnl_gtfms),  outputs F_i[:3,:3], F_i[:3,3].
//
// In PyTorch this is 23 dependent 4x4 matmuls, a batched LU inverse and their autograd twins: ~120 kernels of 2-4 us
// per training step that do 10 KFLOP between them (0.5 ms of a 12.5 ms step, and the same launches in front of every
// rendered frame).  Here lane 0 walks the chain, lanes 0..23 invert their A_i (Gauss-Jordan with partial pivoting --
// torch.inverse is a general LU inverse too) and everything runs in fp64 internally: the results are the correctly
// rounded fp32 values of the exact formulas, i.e. at least as close to the reference as its own fp32 evaluation.
// Backward: dInv_i = C_i^T gF_i,  dA_i = -A_i^-T dInv_i A_i^-T,  then down the tree in reverse bone order
//   dG_i = A_p^T dA_i,  dA_p += dA_i G_i^T.
//
// With ``rvec`` the pose refinement in front of it is folded in as well (BodyPoseRefiner's Rodrigues step,
// network_util.py:57-83, and network.py:677-688): R_i <- R_i Rodrigues(rvec_{i-1}) for the 23 non-root bones,
//   theta = sqrt(1e-5 + |r|^2), n = r / theta, Rodrigues = (1 - cos) n n^T + cos I + sin [n]x,
// ~150 more elementwise launches of a few hundred bytes each (forward + autograd) per training step otherwise.
#include "hnrf_common.h"

namespace hnrf {

constexpr int POSE_B = 24;
__constant__ int kSmplParent[POSE_B] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};

__device__ inline void mm4(const double* a, const double* b, double* c) {          // c = a b (row-major 4x4)
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}
__device__ inline void mm4_tn(const double* a, const double* b, double* c) {       // c = a^T b
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[4 * k + i] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}
__device__ inline void mm4_nt(const double* a, const double* b, double* c) {       // c = a b^T
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * j + k];
            c[4 * i + j] = s;
        }
}

// Gauss-Jordan inverse with partial pivoting; a singular matrix yields inf / nan like torch.inverse's unchecked form
__device__ inline void inv4(const double* a, double* inv) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = a[4 * i + j];
            m[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (fabs(m[r][col]) > fabs(m[piv][col])) piv = r;
        if (piv != col)
            for (int j = 0; j < 8; ++j) {
                const double t = m[col][j];
                m[col][j] = m[piv][j];
                m[piv][j] = t;
            }
        const double d = 1.0 / m[col][col];
        for (int j = 0; j < 8; ++j) m[col][j] *= d;
        for (int r = 0; r < 4; ++r)
            if (r != col) {
                const double f = m[r][col];
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[col][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[4 * i + j] = m[i][4 + j];
}

// Rodrigues(rvec) with the reference's regularised angle; n is NOT a unit vector (|n|^2 = |r|^2 / (1e-5 + |r|^2))
__device__ inline void rodrigues3(const float* rvec, double* R, double* n, double* theta, double* cs) {
    const double rx = (double)rvec[0], ry = (double)rvec[1], rz = (double)rvec[2];
    const double th = sqrt(1e-5 + rx * rx + ry * ry + rz * rz);
    const double x = rx / th, y = ry / th, z = rz / th, c = cos(th), sn = sin(th), oc = 1.0 - c;
    R[0] = x * x + (1.0 - x * x) * c; R[1] = x * y * oc - z * sn;       R[2] = x * z * oc + y * sn;
    R[3] = x * y * oc + z * sn;       R[4] = y * y + (1.0 - y * y) * c; R[5] = y * z * oc - x * sn;
    R[6] = x * z * oc - y * sn;       R[7] = y * z * oc + x * sn;       R[8] = z * z + (1.0 - z * z) * c;
    n[0] = x; n[1] = y; n[2] = z;
    *theta = th;
    cs[0] = c; cs[1] = sn;
}

// G_i = [R_i (Rodrigues(rvec_{i-1}) for i >= 1 when rvec is given) | T_i ; 0 0 0 1]
__device__ inline void load_G(const float* Rs, const float* Ts, const float* rvec, int i, double* g) {
    double R[9];
    for (int k = 0; k < 9; ++k) R[k] = (double)Rs[9 * i + k];
    if (rvec != nullptr && i >= 1) {
        double D[9], n[3], th, cs[2], M[9];
        rodrigues3(rvec + 3 * (i - 1), D, n, &th, cs);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M[3 * r + c] = R[3 * r] * D[c] + R[3 * r + 1] * D[3 + c] + R[3 * r + 2] * D[6 + c];
        for (int k = 0; k < 9; ++k) R[k] = M[k];
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) g[4 * r + c] = R[3 * r + c];
        g[4 * r + 3] = (double)Ts[3 * i + r];
    }
    g[12] = g[13] = g[14] = 0.0;
    g[15] = 1.0;
}

// saved: [A (24 x 16) | A^-1 (24 x 16)] doubles for the backward
__global__ __launch_bounds__(64) void motion_basis_fwd_kernel(const float* __restrict__ rvec,
                                                              const float* __restrict__ dst_Rs,
                                                              const float* __restrict__ dst_Ts,
                                                              const float* __restrict__ cnl_gtfms,
                                                              float* __restrict__ Rs, float* __restrict__ Ts,
                                                              double* __restrict__ saved) {
    __shared__ double G[POSE_B][16];
    __shared__ double A[POSE_B][16];
    const int i = threadIdx.x;
    if (i < POSE_B) load_G(dst_Rs, dst_Ts, rvec, i, G[i]);
    __syncthreads();
    if (i == 0) {
        for (int k = 0; k < 16; ++k) A[0][k] = G[0][k];
        for (int b = 1; b < POSE_B; ++b) mm4(A[kSmplParent[b]], G[b], A[b]);
    }
    __syncthreads();
    if (i < POSE_B) {
        double inv[16], c[16], f[16];
        inv4(A[i], inv);
        for (int k = 0; k < 16; ++k) c[k] = (double)cnl_gtfms[16 * i + k];
        mm4(c, inv, f);
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) Rs[9 * i + 3 * r + cc] = (float)f[4 * r + cc];
            Ts[3 * i + r] = (float)f[4 * r + 3];
        }
        if (saved != nullptr)
            for (int k = 0; k < 16; ++k) {
                saved[16 * i + k] = A[i][k];
                saved[16 * (POSE_B + i) + k] = inv[k];
            }
    }
}

__global__ __launch_bounds__(64) void motion_basis_bwd_kernel(const float* __restrict__ g_Rs, const float* __restrict__ g_Ts,
                                                              const float* __restrict__ rvec,
                                                              const float* __restrict__ dst_Rs,
                                                              const float* __restrict__ dst_Ts,
                                                              const float* __restrict__ cnl_gtfms,
                                                              const double* __restrict__ saved,
                                                              float* __restrict__ d_rvec,
                                                              float* __restrict__ d_dst_Rs, float* __restrict__ d_dst_Ts) {
    __shared__ double dA[POSE_B][16];
    __shared__ double G[POSE_B][16];
    __shared__ double dG[POSE_B][16];
    const int i = threadIdx.x;
    if (i < POSE_B) {
        double gf[16], c[16], dinv[16], t[16];
        load_G(dst_Rs, dst_Ts, rvec, i, G[i]);
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) gf[4 * r + cc] = (double)g_Rs[9 * i + 3 * r + cc];
            gf[4 * r + 3] = (double)g_Ts[3 * i + r];
        }
        gf[12] = gf[13] = gf[14] = gf[15] = 0.0;
        for (int k = 0; k < 16; ++k) c[k] = (double)cnl_gtfms[16 * i + k];
        mm4_tn(c, gf, dinv);                                     // dL/dA^-1 = C^T gF
        const double* inv = saved + 16 * (POSE_B + i);
        mm4_tn(inv, dinv, t);                                    // A^-T dInv
        mm4_nt(t, inv, dinv);                                    // ... A^-T
        for (int k = 0; k < 16; ++k) dA[i][k] = -dinv[k];
    }
    __syncthreads();
    if (i == 0) {
        double t[16];
        for (int b = POSE_B - 1; b >= 1; --b) {                  // children come after their parents
            const int p = kSmplParent[b];
            mm4_tn(saved + 16 * p, dA[b], dG[b]);                // dG_b = A_p^T dA_b
            mm4_nt(dA[b], G[b], t);                              // dA_p += dA_b G_b^T
            for (int k = 0; k < 16; ++k) dA[p][k] += t[k];
        }
        for (int k = 0; k < 16; ++k) dG[0][k] = dA[0][k];        // A_0 = G_0
    }
    __syncthreads();
    if (i < POSE_B) {
        double gR[9];                                            // gradient at the (refined) rotation of G_i
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) gR[3 * r + cc] = dG[i][4 * r + cc];
            d_dst_Ts[3 * i + r] = (float)dG[i][4 * r + 3];
        }
        if (rvec == nullptr || i == 0) {
            for (int k = 0; k < 9; ++k) d_dst_Rs[9 * i + k] = (float)gR[k];
        } else {
            double D[9], n[3], th, cs[2], R0[9], gD[9];
            rodrigues3(rvec + 3 * (i - 1), D, n, &th, cs);
            for (int k = 0; k < 9; ++k) R0[k] = (double)dst_Rs[9 * i + k];
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) {
                    // R = R0 D:  dR0 = gR D^T,  dD = R0^T gR
                    d_dst_Rs[9 * i + 3 * r + cc] =
                        (float)(gR[3 * r] * D[3 * cc] + gR[3 * r + 1] * D[3 * cc + 1] + gR[3 * r + 2] * D[3 * cc + 2]);
                    gD[3 * r + cc] = R0[r] * gR[cc] + R0[3 + r] * gR[3 + cc] + R0[6 + r] * gR[6 + cc];
                }
            // D = (1 - c) n n^T + c I + s [n]x
            const double c = cs[0], sn = cs[1], oc = 1.0 - c, x = n[0], y = n[1], z = n[2];
            double gn_[3], gtn[3];                               // gD n, gD^T n
            for (int r = 0; r < 3; ++r) {
                gn_[r] = gD[3 * r] * x + gD[3 * r + 1] * y + gD[3 * r + 2] * z;
                gtn[r] = gD[r] * x + gD[3 + r] * y + gD[6 + r] * z;
            }
            const double g_c = gD[0] + gD[4] + gD[8] - (x * gn_[0] + y * gn_[1] + z * gn_[2]);
            const double ax = gD[7] - gD[5], ay = gD[2] - gD[6], az = gD[3] - gD[1];   // axial part of gD
            const double g_s = x * ax + y * ay + z * az;
            const double g_n[3] = {oc * (gn_[0] + gtn[0]) + sn * ax, oc * (gn_[1] + gtn[1]) + sn * ay,
                                   oc * (gn_[2] + gtn[2]) + sn * az};
            const double g_th = -sn * g_c + c * g_s;
            const double rx = x * th, ry = y * th, rz = z * th;
            const double k = (g_th - (g_n[0] * rx + g_n[1] * ry + g_n[2] * rz) / (th * th)) / th;
            d_rvec[3 * (i - 1) + 0] = (float)(g_n[0] / th + k * rx);
            d_rvec[3 * (i - 1) + 1] = (float)(g_n[1] / th + k * ry);
            d_rvec[3 * (i - 1) + 2] = (float)(g_n[2] / th + k * rz);
        }
    }
}

// ---- the pose refiner's MLP (BodyPoseRefiner.block_mlps, pose_decoders/mlp_delta_body_pose.py:14-41) -------------------
// 69 -> 256 -> 256 -> 256 -> 256 -> 69 on ONE vector per frame: in PyTorch 5 GEMV launches + bias / ReLU kernels forward
// and ~25 launches backward (0.6 ms of a training step for 0.5 MFLOP).  One small launch per layer each way (a
// single-workgroup version of the whole MLP was tried first: latency-bound, 0.5 ms -- no better than PyTorch): forward
// one wave per output row; backward one launch per layer on a (column block, row block) grid that writes dW and the
// row-block partial sums of W^T dz, which the next launch folds together with relu' while it loads its dz
// (fixed order: deterministic).
constexpr int PMLP_MAX_LAYERS = 9;
constexpr int PMLP_MAX_WIDTH = 256;
constexpr int PMLP_RB = 32;                                       // rows per block of the backward grid
constexpr int PMLP_MAX_PARTS = PMLP_MAX_WIDTH / PMLP_RB;

// out[r] = act(b[r] + W[r, :] . x): one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void pose_layer_fwd_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                             const float* __restrict__ x, int ni, int no, int relu,
                                                             float* __restrict__ out, float* __restrict__ out2) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= no) return;
    float acc = 0.f;
    for (int k = lane; k < ni; k += 64) acc = fmaf(W[(size_t)r * ni + k], x[k], acc);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        float v = acc + b[r];
        if (relu) v = fmaxf(v, 0.f);
        out[r] = v;
        if (out2 != nullptr) out2[r] = v;
    }
}

// One layer of the backward.  dz[j] = relu'(h_out[j]) * sum_p part_in[p][j]  (h_out null: no relu', top layer).
// Block (cb, rb): columns 64 cb .. +63, rows 32 rb .. +31; thread (c, q) takes rows q, q + 4, ... of the row block.
//   dW[j][k] = dz[j] h_in[k];  db[j] = dz[j] (cb == 0);  part_out[rb][k] = sum_{j in row block} W[j][k] dz[j].
__global__ __launch_bounds__(256) void pose_layer_bwd_kernel(const float* __restrict__ W, const float* __restrict__ part_in,
                                                             int nparts_in, const float* __restrict__ h_out,
                                                             const float* __restrict__ h_in, int ni, int no,
                                                             float* __restrict__ dW, float* __restrict__ db,
                                                             float* __restrict__ part_out) {
    __shared__ float dz[PMLP_RB];
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c, j0 = blockIdx.y * PMLP_RB;
    if (threadIdx.x < PMLP_RB) {
        const int j = j0 + threadIdx.x;
        float g = 0.f;
        if (j < no) {
            for (int p = 0; p < nparts_in; ++p) g += part_in[p * PMLP_MAX_WIDTH + j];
            if (h_out != nullptr && !(h_out[j] > 0.f)) g = 0.f;
            if (blockIdx.x == 0) db[j] = g;
        }
        dz[threadIdx.x] = g;
    }
    __syncthreads();
    float acc = 0.f;
    if (k < ni) {
        const float hk = h_in[k];
#pragma unroll
        for (int i = 0; i < PMLP_RB / 4; ++i) {
            const int jj = q + 4 * i, j = j0 + jj;
            if (j < no) {
                const float g = dz[jj];
                acc = fmaf(W[(size_t)j * ni + k], g, acc);
                dW[(size_t)j * ni + k] = g * hk;
            }
        }
    }
    red[q][c] = acc;
    __syncthreads();
    if (q == 0 && k < ni) part_out[blockIdx.y * PMLP_MAX_WIDTH + k] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

}  // namespace hnrf

using namespace hnrf;

static int pose_mlp_args(const char* who, const float* const* W, const float* const* b, const int* dims, int layers) {
    HNRF_REQUIRE(W && b && dims, HNRF_E_ARG, "%s: null pointer", who);
    HNRF_REQUIRE(layers >= 1 && layers <= PMLP_MAX_LAYERS, HNRF_E_UNSUPPORTED, "%s: %d layers (built: 1..%d)", who, layers,
                 PMLP_MAX_LAYERS);
    for (int l = 0; l <= layers; ++l)
        HNRF_REQUIRE(dims[l] >= 1 && dims[l] <= PMLP_MAX_WIDTH, HNRF_E_UNSUPPORTED, "%s: width %d (built: 1..%d)", who, dims[l],
                     PMLP_MAX_WIDTH);
    for (int l = 0; l < layers; ++l) HNRF_REQUIRE(W[l] && b[l], HNRF_E_ARG, "%s: null layer %d", who, l);
    return HNRF_OK;
}

// saved = [hidden activations: (layers - 1) x 256 | backward scratch: 2 x 8 x 256 partial sums] floats
extern "C" size_t hnrf_pose_mlp_saved_bytes(int layers) {
    return layers >= 1 && layers <= PMLP_MAX_LAYERS
               ? (size_t)((layers - 1) + 2 * PMLP_MAX_PARTS) * PMLP_MAX_WIDTH * sizeof(float)
               : 0;
}

extern "C" int hnrf_pose_mlp_fwd(const float* x, const float* const* W, const float* const* b, const int* dims, int layers,
                                 float* out, float* saved, void* stream) {
    if (int rc = pose_mlp_args("hnrf_pose_mlp_fwd", W, b, dims, layers)) return rc;
    HNRF_REQUIRE(x && out && (saved || layers == 1), HNRF_E_ARG, "hnrf_pose_mlp_fwd: null pointer");
    const float* in = x;
    for (int l = 0; l < layers; ++l) {
        const bool hidden = l + 1 < layers;
        float* dst = hidden ? saved + (size_t)l * PMLP_MAX_WIDTH : out;
        hipLaunchKernelGGL(pose_layer_fwd_kernel, dim3((dims[l + 1] + 3) / 4), dim3(256), 0, (hipStream_t)stream, W[l], b[l], in,
                           dims[l], dims[l + 1], hidden ? 1 : 0, dst, (float*)nullptr);
        in = dst;
    }
    return check_launch("hnrf_pose_mlp_fwd");
}

extern "C" int hnrf_pose_mlp_bwd(const float* g_out, const float* x, const float* const* W, const float* const* b,
                                 const int* dims, int layers, float* saved, float* const* dW, float* const* db,
                                 float* d_x_parts, void* stream) {
    if (int rc = pose_mlp_args("hnrf_pose_mlp_bwd", W, b, dims, layers)) return rc;
    HNRF_REQUIRE(g_out && x && dW && db && saved, HNRF_E_ARG, "hnrf_pose_mlp_bwd: null pointer");
    for (int l = 0; l < layers; ++l)
        HNRF_REQUIRE(dW[l] && db[l], HNRF_E_ARG, "hnrf_pose_mlp_bwd: null gradient buffer of layer %d", l);
    float* scratch = saved + (size_t)(layers - 1) * PMLP_MAX_WIDTH;
    const float* part_in = g_out;
    int nparts = 1;
    for (int l = layers - 1; l >= 0; --l) {
        const int ni = dims[l], no = dims[l + 1];
        const float* h_out = l + 1 < layers ? saved + (size_t)l * PMLP_MAX_WIDTH : nullptr;
        const float* h_in = l == 0 ? x : saved + (size_t)(l - 1) * PMLP_MAX_WIDTH;
        // layer 0's partial sums of W^T dz are d_x (folded by the caller if it wants them); ping-pong otherwise
        float* part_out = l == 0 && d_x_parts != nullptr ? d_x_parts
                                                         : scratch + (size_t)(l & 1) * PMLP_MAX_PARTS * PMLP_MAX_WIDTH;
        const dim3 grid((ni + 63) / 64, (no + PMLP_RB - 1) / PMLP_RB);
        hipLaunchKernelGGL(pose_layer_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, W[l], part_in, nparts, h_out, h_in,
                           ni, no, dW[l], db[l], part_out);
        part_in = part_out;
        nparts = (int)grid.y;
    }
    return check_launch("hnrf_pose_mlp_bwd");
}

extern "C" size_t hnrf_motion_basis_saved_bytes(void) { return (size_t)2 * POSE_B * 16 * sizeof(double); }

extern "C" int hnrf_motion_basis_fwd(const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int B, float* Rs,
                                     float* Ts, void* saved, void* stream) {
    HNRF_REQUIRE(dst_Rs && dst_Ts && cnl_gtfms && Rs && Ts, HNRF_E_ARG, "hnrf_motion_basis_fwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_motion_basis_fwd: %d bones (the SMPL tree has 24)", B);
    HNRF_REQUIRE(((uintptr_t)saved & 7) == 0, HNRF_E_ARG, "hnrf_motion_basis_fwd: saved must be 8-byte aligned");
    hipLaunchKernelGGL(motion_basis_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)nullptr, dst_Rs,
                       dst_Ts, cnl_gtfms, Rs, Ts, (double*)saved);
    return check_launch("hnrf_motion_basis_fwd");
}

extern "C" int hnrf_motion_basis_bwd(const float* g_Rs, const float* g_Ts, const float* dst_Rs, const float* dst_Ts,
                                     const float* cnl_gtfms, int B, const void* saved, float* d_dst_Rs, float* d_dst_Ts,
                                     void* stream) {
    HNRF_REQUIRE(g_Rs && g_Ts && dst_Rs && dst_Ts && cnl_gtfms && saved && d_dst_Rs && d_dst_Ts, HNRF_E_ARG,
                 "hnrf_motion_basis_bwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_motion_basis_bwd: %d bones (the SMPL tree has 24)", B);
    hipLaunchKernelGGL(motion_basis_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_Rs, g_Ts, (const float*)nullptr,
                       dst_Rs, dst_Ts, cnl_gtfms, (const double*)saved, (float*)nullptr, d_dst_Rs, d_dst_Ts);
    return check_launch("hnrf_motion_basis_bwd");
}

extern "C" int hnrf_refined_motion_basis_fwd(const float* rvec, const float* dst_Rs, const float* dst_Ts,
                                             const float* cnl_gtfms, int B, float* Rs, float* Ts, void* saved, void* stream) {
    HNRF_REQUIRE(rvec && dst_Rs && dst_Ts && cnl_gtfms && Rs && Ts, HNRF_E_ARG, "hnrf_refined_motion_basis_fwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_refined_motion_basis_fwd: %d bones (the SMPL tree has 24)", B);
    HNRF_REQUIRE(((uintptr_t)saved & 7) == 0, HNRF_E_ARG, "hnrf_refined_motion_basis_fwd: saved must be 8-byte aligned");
    hipLaunchKernelGGL(motion_basis_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rvec, dst_Rs, dst_Ts, cnl_gtfms, Rs,
                       Ts, (double*)saved);
    return check_launch("hnrf_refined_motion_basis_fwd");
}

extern "C" int hnrf_refined_motion_basis_bwd(const float* g_Rs, const float* g_Ts, const float* rvec, const float* dst_Rs,
                                             const float* dst_Ts, const float* cnl_gtfms, int B, const void* saved,
                                             float* d_rvec, float* d_dst_Rs, float* d_dst_Ts, void* stream) {
    HNRF_REQUIRE(g_Rs && g_Ts && rvec && dst_Rs && dst_Ts && cnl_gtfms && saved && d_rvec && d_dst_Rs && d_dst_Ts, HNRF_E_ARG,
                 "hnrf_refined_motion_basis_bwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_refined_motion_basis_bwd: %d bones (the SMPL tree has 24)", B);
    hipLaunchKernelGGL(motion_basis_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_Rs, g_Ts, rvec, dst_Rs, dst_Ts,
                       cnl_gtfms, (const double*)saved, d_rvec, d_dst_Rs, d_dst_Ts);
    return check_launch("hnrf_refined_motion_basis_bwd");
}
