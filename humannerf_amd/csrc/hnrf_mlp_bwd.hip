// Backward of the two MLPs: weight gradients.
//
// dW[o][i] = sum_s dZ[s][o] X[s][i]   (nn.Linear autograd; the reference gets it from
// loss.backward(), core/train/trainers/human_nerf/trainer.py:139-170), db[o] = sum_s dZ[s][o].
//
// A GEMM whose contraction runs over ~10^6 samples and whose result is one 256x256 (or
// smaller) matrix: the library's answer is a split-K GEMM at ~55 TFLOP/s.  Here one
// workgroup per CU keeps the WHOLE output in the accumulators of its 4 waves (wave w: rows
// [64w, 64w+64) x all columns = 16 tiles of 32x32 = 256 accumulator registers) and streams
// its slice of the samples straight from the row-major [P, width] matrices the forward
// saved -- every byte of dZ and X is read from HBM exactly once.
//
// Operand trick: v_mfma_f32_32x32x2_f32 wants A[row c][k = h] and B[k = h][col c] from lane
// (c, h).  The order of the contraction is free, and so is the order of the rows inside a
// tile: a lane loads 2 (4) CONSECUTIVE columns of its sample's dZ (X) row with one 8 (16)
// byte load and uses them as row (column) c of 2 (4) different tiles, i.e. tile q holds
// o = 2 c + q (i = 4 c + q).  The store undoes the interleave.
//
// Roofline: MFMA-bound for 256x256 (103 GFLOP per 786 k samples; 1.6 GB of HBM reads
// = 0.25 ms at 6.5 TB/s vs 0.66 ms at the 157 TFLOP/s fp32-MFMA peak).
#include "hnrf_common.h"

namespace hnrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) f32x2p { float x, y; };

constexpr int DW_DEPTH = 16;       // k-steps (2 samples each) of loads in flight per wave
constexpr int DW_SPLIT = 256;      // sample slices: one resident workgroup per CU (its accumulators fill the register file)

// OT: 32-row tiles per wave (n_out = 128 OT).  IB: 128-column blocks of X (n_in = 128 IB);
// IB == 0: n_in <= 64, arbitrary row stride, masked 4-byte loads (the positional encodings).
template <int OT, int IB>
__global__ __launch_bounds__(256) void mlp_dw_kernel(const float* __restrict__ dZ, int64_t ldz,
                                                     const float* __restrict__ X, int64_t ldx, int n_in,
                                                     int64_t P, int64_t per_wg, float* __restrict__ part,
                                                     float* __restrict__ dbpart) {
    constexpr int IT = IB ? 4 * IB : 2;
    constexpr int NOW = 128 * OT, NIP = 32 * IT;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;      // s0 < P by construction of the grid

    f32x16 acc[OT][IT];
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int b = 0; b < IT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float bsum[OT];
#pragma unroll
    for (int a = 0; a < OT; ++a) bsum[a] = 0.f;

    const float* ap = dZ + 32 * OT * w + OT * c;
    const float* bp = X + (IB ? 4 * c : 2 * c);
    const bool b0ok = IB || 2 * c < n_in, b1ok = IB || 2 * c + 1 < n_in;

    float ar[DW_DEPTH][OT];
    float br[DW_DEPTH][IT];
    // rows past the slice are clamped here and zeroed where they are consumed (masking the loaded value
    // here would put a wait for THIS load in front of the MFMAs)
    auto fetch = [&](int slot, int64_t ks) {
        const int64_t s = s0 + 2 * ks + h;
        const int64_t sc = s < s1 ? s : s1 - 1;
        if (OT == 2) {
            const f32x2p v = *reinterpret_cast<const f32x2p*>(ap + sc * ldz);
            ar[slot][0] = v.x;
            ar[slot][OT - 1] = v.y;
        } else {
            const float v = ap[sc * ldz];
            ar[slot][0] = v;
        }
        if (IB) {
#pragma unroll
            for (int b = 0; b < (IB ? IB : 1); ++b) {
                const f32x4v v = *reinterpret_cast<const f32x4v*>(bp + sc * ldx + 128 * b);
                br[slot][4 * b + 0] = v.x;
                br[slot][(4 * b + 1) % IT] = v.y;
                br[slot][(4 * b + 2) % IT] = v.z;
                br[slot][(4 * b + 3) % IT] = v.w;
            }
        } else {
            br[slot][0] = b0ok ? bp[sc * ldx] : 0.f;
            br[slot][1] = b1ok ? bp[sc * ldx + 1] : 0.f;
        }
    };

    const int64_t nks = (per_wg + 1) / 2;                      // per_wg is a multiple of 2 DW_DEPTH
#pragma unroll
    for (int d = 0; d < DW_DEPTH; ++d) fetch(d, d);
    for (int64_t k0 = 0; k0 < nks; k0 += DW_DEPTH) {
#pragma unroll
        for (int d = 0; d < DW_DEPTH; ++d) {
            float a[OT], b[IT];
#pragma unroll
            for (int q = 0; q < OT; ++q) a[q] = (s0 + 2 * (k0 + d) + h < s1) ? ar[d][q] : 0.f;
#pragma unroll
            for (int q = 0; q < IT; ++q) b[q] = br[d][q];
            fetch(d, k0 + d + DW_DEPTH);
            // keep the loads HERE (DW_DEPTH k-steps ahead of their use): left alone, hipcc batches all MFMAs of
            // the unrolled body first and parks the loads at its end, in front of a vmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qa = 0; qa < OT; ++qa) {
                bsum[qa] += a[qa];
#pragma unroll
                for (int qb = 0; qb < IT; ++qb)
                    acc[qa][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[qa], b[qb], acc[qa][qb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // partial result of this slice: part[blockIdx][o][i], o = 32 OT w + OT ra + qa, i = 4 cb + qb (+128 blk)
    float* out = part + (int64_t)blockIdx.x * NOW * NIP;
#pragma unroll
    for (int qa = 0; qa < OT; ++qa)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ra = (r & 3) + 8 * (r >> 2) + 4 * h;
            float* row = out + (int64_t)(32 * OT * w + OT * ra + qa) * NIP;
            if (IB) {
#pragma unroll
                for (int b = 0; b < (IB ? IB : 1); ++b)
                    *reinterpret_cast<f32x4v*>(row + 128 * b + 4 * c) =
                        f32x4v{acc[qa][4 * b][r], acc[qa][(4 * b + 1) % IT][r], acc[qa][(4 * b + 2) % IT][r],
                               acc[qa][(4 * b + 3) % IT][r]};
            } else {
                row[2 * c] = acc[qa][0][r];
                row[2 * c + 1] = acc[qa][1][r];
            }
        }
    if (dbpart != nullptr) {
#pragma unroll
        for (int qa = 0; qa < OT; ++qa) {
            const float t = bsum[qa] + __shfl_xor(bsum[qa], 32, 64);
            if (h == 0) dbpart[(int64_t)blockIdx.x * NOW + 32 * OT * w + OT * c + qa] = t;
        }
    }
}

// dW[o][i] = sum over slices; db likewise.  Block = 64 consecutive elements x 4 slice groups (group g takes
// slices g, g+4, ..), combined through LDS in a fixed order: the result does not depend on scheduling.
__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(const float* __restrict__ part,
                                                            const float* __restrict__ dbpart, int nsplit, int now,
                                                            int nip, int n_out, int n_in, float* __restrict__ dW,
                                                            int64_t ldw, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int g = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int nelem = now * nip;
    const bool is_db = (int)blockIdx.x * 64 >= nelem;          // trailing blocks reduce the bias partials
    const int e = is_db ? (int)blockIdx.x * 64 - nelem + t : (int)blockIdx.x * 64 + t;
    const float* src = is_db ? dbpart : part;
    const int64_t stride = is_db ? now : (int64_t)nelem;
    const bool ok = is_db ? e < now : true;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (ok) {
        int k = g;
        for (; k + 12 < nsplit; k += 16) {
            s0 += src[(int64_t)k * stride + e];
            s1 += src[(int64_t)(k + 4) * stride + e];
            s2 += src[(int64_t)(k + 8) * stride + e];
            s3 += src[(int64_t)(k + 12) * stride + e];
        }
        for (; k < nsplit; k += 4) s0 += src[(int64_t)k * stride + e];
    }
    red[g][t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && ok) {
        const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (is_db) {
            if (e < n_out) db[e] = v;
        } else {
            const int o = e / nip, i = e - o * nip;
            if (o < n_out && i < n_in) dW[(int64_t)o * ldw + i] = v;
        }
    }
}

struct DwPlan {
    int now, nip;          // padded output tile
    int64_t per_wg;
    int nsplit;
};

static bool dw_plan(int64_t P, int n_out, int n_in, DwPlan& pl) {
    if (n_out == 256 || n_out == 128) pl.now = n_out; else return false;
    if (n_in == 256 || n_in == 128) pl.nip = n_in; else if (n_in >= 1 && n_in <= 64) pl.nip = 64; else return false;
    const int64_t unit = 2 * DW_DEPTH;
    int64_t per = (P + DW_SPLIT - 1) / DW_SPLIT;
    per = (per + unit - 1) / unit * unit;
    if (per < 4 * unit) per = 4 * unit;
    pl.per_wg = per;
    pl.nsplit = (int)((P + per - 1) / per);
    return true;
}

}  // namespace hnrf

using namespace hnrf;

extern "C" size_t hnrf_mlp_dw_workspace_bytes(int64_t P, int n_out, int n_in) {
    DwPlan pl;
    if (P <= 0 || !dw_plan(P, n_out, n_in, pl)) return 0;
    return ((size_t)pl.nsplit * pl.now * pl.nip + (size_t)pl.nsplit * pl.now) * sizeof(float);
}

extern "C" int hnrf_mlp_dw(const float* dZ, int64_t ldz, const float* X, int64_t ldx, int64_t P, int n_out, int n_in,
                           float* dW, int64_t ldw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    HNRF_REQUIRE(dZ && X && dW && workspace, HNRF_E_ARG, "hnrf_mlp_dw: null pointer");
    DwPlan pl;
    HNRF_REQUIRE(P > 0 && dw_plan(P, n_out, n_in, pl), HNRF_E_UNSUPPORTED,
                 "hnrf_mlp_dw: shape P=%lld n_out=%d n_in=%d not built (n_out 128|256, n_in 128|256|<=64)",
                 (long long)P, n_out, n_in);
    HNRF_REQUIRE(ldz >= n_out && ldx >= n_in && ldw >= n_in, HNRF_E_ARG, "hnrf_mlp_dw: row stride below width");
    HNRF_REQUIRE(workspace_bytes >= hnrf_mlp_dw_workspace_bytes(P, n_out, n_in), HNRF_E_ARG,
                 "hnrf_mlp_dw: workspace too small");
    if (n_in > 64)
        HNRF_REQUIRE(((uintptr_t)X & 15) == 0 && ldx % 4 == 0, HNRF_E_ARG,
                     "hnrf_mlp_dw: X must be 16-byte aligned with a row stride that is a multiple of 4");
    float* part = (float*)workspace;
    float* dbpart = part + (size_t)pl.nsplit * pl.now * pl.nip;
    hipStream_t st = (hipStream_t)stream;
#define HNRF_DW(OT, IB)                                                                                        \
    hipLaunchKernelGGL((mlp_dw_kernel<OT, IB>), dim3(pl.nsplit), dim3(256), 0, st, dZ, ldz, X, ldx, n_in, P, \
                       pl.per_wg, part, db ? dbpart : nullptr)
    if (n_out == 256) {
        if (n_in == 256) HNRF_DW(2, 2); else if (n_in == 128) HNRF_DW(2, 1); else HNRF_DW(2, 0);
    } else {
        if (n_in == 256) HNRF_DW(1, 2); else if (n_in == 128) HNRF_DW(1, 1); else HNRF_DW(1, 0);
    }
#undef HNRF_DW
    int rc = check_launch("hnrf_mlp_dw");
    if (rc) return rc;
    const int nblk = pl.now * pl.nip / 64 + (db ? (pl.now + 63) / 64 : 0);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(nblk), dim3(256), 0, st, part, dbpart, pl.nsplit, pl.now,
                       pl.nip, n_out, n_in, dW, ldw, db);
    return check_launch("hnrf_mlp_dw (reduce)");
}
