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

__device__ inline void load_G(const float* Rs, const float* Ts, int i, double* g) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) g[4 * r + c] = (double)Rs[9 * i + 3 * r + c];
        g[4 * r + 3] = (double)Ts[3 * i + r];
    }
    g[12] = g[13] = g[14] = 0.0;
    g[15] = 1.0;
}

// saved: [A (24 x 16) | A^-1 (24 x 16)] doubles for the backward
__global__ __launch_bounds__(64) void motion_basis_fwd_kernel(const float* __restrict__ dst_Rs,
                                                              const float* __restrict__ dst_Ts,
                                                              const float* __restrict__ cnl_gtfms,
                                                              float* __restrict__ Rs, float* __restrict__ Ts,
                                                              double* __restrict__ saved) {
    __shared__ double A[POSE_B][16];
    const int i = threadIdx.x;
    if (i == 0) {
        double g[16];
        load_G(dst_Rs, dst_Ts, 0, A[0]);
        for (int b = 1; b < POSE_B; ++b) {
            load_G(dst_Rs, dst_Ts, b, g);
            mm4(A[kSmplParent[b]], g, A[b]);
        }
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
                                                              const float* __restrict__ dst_Rs,
                                                              const float* __restrict__ dst_Ts,
                                                              const float* __restrict__ cnl_gtfms,
                                                              const double* __restrict__ saved,
                                                              float* __restrict__ d_dst_Rs, float* __restrict__ d_dst_Ts) {
    __shared__ double dA[POSE_B][16];
    const int i = threadIdx.x;
    if (i < POSE_B) {
        double gf[16], c[16], dinv[16], t[16];
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
        double g[16], dg[16], t[16];
        for (int b = POSE_B - 1; b >= 1; --b) {                  // children come after their parents
            const int p = kSmplParent[b];
            load_G(dst_Rs, dst_Ts, b, g);
            mm4_tn(saved + 16 * p, dA[b], dg);                   // dG_b = A_p^T dA_b
            mm4_nt(dA[b], g, t);                                 // dA_p += dA_b G_b^T
            for (int k = 0; k < 16; ++k) dA[p][k] += t[k];
            for (int r = 0; r < 3; ++r) {
                for (int cc = 0; cc < 3; ++cc) d_dst_Rs[9 * b + 3 * r + cc] = (float)dg[4 * r + cc];
                d_dst_Ts[3 * b + r] = (float)dg[4 * r + 3];
            }
        }
        for (int r = 0; r < 3; ++r) {                            // A_0 = G_0
            for (int cc = 0; cc < 3; ++cc) d_dst_Rs[3 * r + cc] = (float)dA[0][4 * r + cc];
            d_dst_Ts[r] = (float)dA[0][4 * r + 3];
        }
    }
}

}  // namespace hnrf

using namespace hnrf;

extern "C" size_t hnrf_motion_basis_saved_bytes(void) { return (size_t)2 * POSE_B * 16 * sizeof(double); }

extern "C" int hnrf_motion_basis_fwd(const float* dst_Rs, const float* dst_Ts, const float* cnl_gtfms, int B, float* Rs,
                                     float* Ts, void* saved, void* stream) {
    HNRF_REQUIRE(dst_Rs && dst_Ts && cnl_gtfms && Rs && Ts, HNRF_E_ARG, "hnrf_motion_basis_fwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_motion_basis_fwd: %d bones (the SMPL tree has 24)", B);
    HNRF_REQUIRE(((uintptr_t)saved & 7) == 0, HNRF_E_ARG, "hnrf_motion_basis_fwd: saved must be 8-byte aligned");
    hipLaunchKernelGGL(motion_basis_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst_Rs, dst_Ts, cnl_gtfms, Rs, Ts,
                       (double*)saved);
    return check_launch("hnrf_motion_basis_fwd");
}

extern "C" int hnrf_motion_basis_bwd(const float* g_Rs, const float* g_Ts, const float* dst_Rs, const float* dst_Ts,
                                     const float* cnl_gtfms, int B, const void* saved, float* d_dst_Rs, float* d_dst_Ts,
                                     void* stream) {
    HNRF_REQUIRE(g_Rs && g_Ts && dst_Rs && dst_Ts && cnl_gtfms && saved && d_dst_Rs && d_dst_Ts, HNRF_E_ARG,
                 "hnrf_motion_basis_bwd: null pointer");
    HNRF_REQUIRE(B == POSE_B, HNRF_E_UNSUPPORTED, "hnrf_motion_basis_bwd: %d bones (the SMPL tree has 24)", B);
    hipLaunchKernelGGL(motion_basis_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_Rs, g_Ts, dst_Rs, dst_Ts,
                       cnl_gtfms, (const double*)saved, d_dst_Rs, d_dst_Ts);
    return check_launch("hnrf_motion_basis_bwd");
}
