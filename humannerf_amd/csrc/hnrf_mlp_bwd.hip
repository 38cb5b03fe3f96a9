// Backward of the two MLPs: weight gradients (hnrf_mlp_dw) and the dX chain (hnrf_*_bwd, below).
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
#include "hnrf_sincos.h"
#include "hnrf_mlp_layout.h"

namespace hnrf {

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

// ---------------------------------------------------------------------------- dW, split-f16
// Same decomposition (one workgroup per CU owns the whole output, samples streamed once), but the products run
// on the f16 matrix pipe at fp32-class accuracy: every operand v is split hi = f16(v), lo = f16(v - hi) and
//   a b ~= ah bh + ah bl + al bh      (three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator).
// What matters for a sum over ~10^6 samples is the ABSOLUTE error of each operand: dZ is scaled by a power of
// two so that its largest magnitude lands in [2^7, 2^8) (amax comes from the chain kernel), which puts the
// residual lo in f16's normal range for every element that matters and bounds the operand error by
// max(2^-22 |v|, 2^-25 * 2^-8 amax); X (activations, |x| <= 65504 by the forward's clamp) needs no scale.
// The MFMA wants 8 consecutive SAMPLES of one column per lane, the matrices are [sample][column]: the wave's own
// dZ columns are gathered by 8 row loads per lane (two columns each, tile interleave as above); X is shared by
// the 4 waves, so each thread converts 1/256 of the 16 x 256 block of a k-step and the fragments meet in LDS
// (double-buffered, one raw s_barrier per k-step -- __syncthreads() would drain the global prefetch ring).
// Per k-step (16 samples) a wave issues 48 MFMAs (1536 cycles) against 12 loads and ~100 VALU: matrix-bound,
// and at 0.3-0.4 ms per 256x256 layer it runs into the 1.6 GB / layer of HBM reads.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int DW16_DEPTH = 4;      // k-steps of raw global loads in flight (also the unroll of the main loop)

__device__ __forceinline__ void split_pair(float v0, float v1, _Float16& h0, _Float16& h1, _Float16& l0, _Float16& l1) {
    const f32x2v p = {__builtin_amdgcn_fmed3f(v0, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v1, -65504.f, 65504.f)};
    const h16x2 hh = __builtin_convertvector(p, h16x2);
    const f32x2v r = {fmaf((float)hh[0], -1.0f, p[0]), fmaf((float)hh[1], -1.0f, p[1])};   // v_fma_mix_f32, exact
    const h16x2 ll = __builtin_convertvector(r, h16x2);
    h0 = hh[0]; h1 = hh[1]; l0 = ll[0]; l1 = ll[1];
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// OT: 32-row tiles of dZ columns per wave (n_out = 128 OT); IB: 128-column blocks of X (n_in = 128 IB).
template <int OT, int IB>
__global__ __launch_bounds__(256) void mlp_dw16_kernel(const float* __restrict__ dZ, int64_t ldz,
                                                       const float* __restrict__ X, int64_t ldx, int64_t P,
                                                       int64_t per_wg, const float* __restrict__ dz_amax,
                                                       int n_amax, float* __restrict__ part,
                                                       float* __restrict__ dbpart) {
    constexpr int IT = 4 * IB, NOW = 128 * OT, NIP = 32 * IT;
    constexpr int RPT = 2 * IB;                  // X rows per thread and k-step (256 threads cover 16 x 128 IB)
    constexpr int NQ = 32 * IB;                  // column quads
    __shared__ h16x8 bfrag[2][IT * 2 * 64];      // [buffer][(tile, hi|lo)][lane]
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5, w = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;
    // power-of-two scale of dZ: largest magnitude -> [2^7, 2^8)
    int ex = 0;
    float amax = 0.f;
    for (int i = 0; i < n_amax; ++i) amax = fmaxf(amax, dz_amax[i]);   // uniform address: scalar loads
    if (amax > 0.f) (void)frexpf(amax, &ex);
    const float scale = ldexpf(1.0f, 8 - ex), descale = ldexpf(1.0f, ex - 8);

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

    // this thread's share of X: column quad `quad`, rows RPT*part .. of each 16-row k-step
    const int quad = tid % NQ, part_i = tid / NQ;
    const int xr0 = RPT * part_i;
    const float* ap = dZ + 32 * OT * w + OT * c;
    const float* bp = X + 4 * quad;
    // LDS destination of the thread's half-fragments: tile 4 (quad >> 5) + q, lane (xr0 >> 3) * 32 + (quad & 31)
    const int frag_lane = (xr0 >> 3) * 32 + (quad & 31);

    float araw[DW16_DEPTH][8][OT];
    float braw[DW16_DEPTH][RPT][4];
    auto fetch = [&](int slot, int64_t ks) {
        const int64_t sa = s0 + 16 * ks + 8 * h;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t s = sa + j < s1 ? sa + j : s1 - 1;
            if (OT == 2) {
                const f32x2p v = *reinterpret_cast<const f32x2p*>(ap + s * ldz);
                araw[slot][j][0] = v.x;
                araw[slot][j][OT - 1] = v.y;
            } else {
                araw[slot][j][0] = ap[s * ldz];
            }
        }
        const int64_t sb = s0 + 16 * ks + xr0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int64_t s = sb + r < s1 ? sb + r : s1 - 1;
            const f32x4v v = *reinterpret_cast<const f32x4v*>(bp + s * ldx);
            braw[slot][r][0] = v.x; braw[slot][r][1] = v.y; braw[slot][r][2] = v.z; braw[slot][r][3] = v.w;
        }
    };
    // X rows of k-step ks (ring slot) -> f16 hi / lo half-fragments in LDS buffer buf
    auto stage_b = [&](int slot, int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            _Float16 hi[RPT], lo[RPT];
#pragma unroll
            for (int r = 0; r < RPT; r += 2) split_pair(braw[slot][r][q], braw[slot][r + 1][q], hi[r], hi[r + 1], lo[r], lo[r + 1]);
            const int tile = 4 * (quad >> 5) + q;
            typedef _Float16 hvec __attribute__((ext_vector_type(RPT)));
            hvec vh, vl;
#pragma unroll
            for (int r = 0; r < RPT; ++r) { vh[r] = hi[r]; vl[r] = lo[r]; }
            *reinterpret_cast<hvec*>(reinterpret_cast<_Float16*>(&bfrag[buf][(tile * 2 + 0) * 64 + frag_lane]) + (xr0 & 7)) = vh;
            *reinterpret_cast<hvec*>(reinterpret_cast<_Float16*>(&bfrag[buf][(tile * 2 + 1) * 64 + frag_lane]) + (xr0 & 7)) = vl;
        }
    };

    const int64_t nks = per_wg / 16;                           // per_wg is a multiple of 16 DW16_DEPTH
#pragma unroll
    for (int d = 0; d < DW16_DEPTH; ++d) fetch(d, d);
    stage_b(0, 0);
    lds_barrier();
    for (int64_t k0 = 0; k0 < nks; k0 += DW16_DEPTH) {
#pragma unroll
        for (int d = 0; d < DW16_DEPTH; ++d) {
            // this wave's dZ columns of k-step k0 + d: scale, zero the rows past the slice, split
            h16x8 ah[OT], al[OT];
#pragma unroll
            for (int q = 0; q < OT; ++q) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = s0 + 16 * (k0 + d) + 8 * h + j < s1;
                    v[j] = ok ? araw[d][j][q] : 0.f;
                    bsum[q] += v[j];
                    v[j] *= scale;
                }
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    _Float16 h0, h1, l0, l1;
                    split_pair(v[j], v[j + 1], h0, h1, l0, l1);
                    ah[q][j] = h0; ah[q][j + 1] = h1; al[q][j] = l0; al[q][j + 1] = l1;
                }
            }
            stage_b((d + 1) % DW16_DEPTH, (d + 1) & 1);        // X of the NEXT k-step -> the other LDS buffer
            fetch(d, k0 + d + DW16_DEPTH);                      // refill this ring slot
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const h16x8 bh = bfrag[d & 1][(it * 2 + 0) * 64 + lane];
                const h16x8 bl = bfrag[d & 1][(it * 2 + 1) * 64 + lane];
#pragma unroll
                for (int qa = 0; qa < OT; ++qa) acc[qa][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[qa], bh, acc[qa][it], 0, 0, 0);
#pragma unroll
                for (int qa = 0; qa < OT; ++qa) acc[qa][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[qa], bl, acc[qa][it], 0, 0, 0);
#pragma unroll
                for (int qa = 0; qa < OT; ++qa) acc[qa][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[qa], bh, acc[qa][it], 0, 0, 0);
            }
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float* out = part + (int64_t)blockIdx.x * NOW * NIP;
#pragma unroll
    for (int qa = 0; qa < OT; ++qa)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ra = (r & 3) + 8 * (r >> 2) + 4 * h;
            float* row = out + (int64_t)(32 * OT * w + OT * ra + qa) * NIP;
#pragma unroll
            for (int b = 0; b < IB; ++b)
                *reinterpret_cast<f32x4v*>(row + 128 * b + 4 * c) =
                    f32x4v{acc[qa][4 * b][r] * descale, acc[qa][4 * b + 1][r] * descale, acc[qa][4 * b + 2][r] * descale,
                           acc[qa][4 * b + 3][r] * descale};
        }
    if (dbpart != nullptr) {
#pragma unroll
        for (int qa = 0; qa < OT; ++qa) {
            const float t = bsum[qa] + __shfl_xor(bsum[qa], 32, 64);
            if (h == 0) dbpart[(int64_t)blockIdx.x * NOW + 32 * OT * w + OT * c + qa] = t;
        }
    }
}

// Head layers (n_out <= 4: sigma/rgb, xyz offset): no matrix work, one pass over X.  Thread i owns column i of
// its slice; dY rows arrive through the scalar cache.  Partials in the layout of mlp_dw_kernel (NOW = 4).
template <int NI>
__global__ __launch_bounds__(NI) void mlp_dw_head_kernel(const float* __restrict__ dY, int64_t ldy, int n_out,
                                                          const float* __restrict__ X, int64_t ldx, int64_t P,
                                                          int64_t per_wg, float* __restrict__ part,
                                                          float* __restrict__ dbpart) {
    const int i = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b = 0.f;
    const float* xp = X + i;
    auto step = [&](int64_t s, float x) {
        const float* g = dY + s * ldy;                        // block-uniform address
        const float g0 = g[0], g1 = n_out > 1 ? g[1] : 0.f, g2 = n_out > 2 ? g[2] : 0.f, g3 = n_out > 3 ? g[3] : 0.f;
        a0 = fmaf(g0, x, a0);
        a1 = fmaf(g1, x, a1);
        a2 = fmaf(g2, x, a2);
        a3 = fmaf(g3, x, a3);
        b += i == 0 ? g0 : (i == 1 ? g1 : (i == 2 ? g2 : g3));
    };
    int64_t s = s0;
    for (; s + 8 <= s1; s += 8) {                             // 8 independent row loads in flight per thread
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = xp[(s + u) * ldx];
#pragma unroll
        for (int u = 0; u < 8; ++u) step(s + u, x[u]);
    }
    for (; s < s1; ++s) step(s, xp[s * ldx]);
    float* out = part + (int64_t)blockIdx.x * 4 * NI;
    out[i] = a0;
    out[NI + i] = a1;
    out[2 * NI + i] = a2;
    out[3 * NI + i] = a3;
    if (dbpart != nullptr && i < 4) dbpart[(int64_t)blockIdx.x * 4 + i] = b;
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

static bool dw_plan(int64_t P, int n_out, int n_in, DwPlan& pl, bool f16 = false) {
    if (n_out == 256 || n_out == 128) pl.now = n_out; else if (n_out >= 1 && n_out <= 4) pl.now = 4; else return false;
    if (n_in == 256 || n_in == 128) pl.nip = n_in;
    else if (n_in >= 1 && n_in <= 64 && n_out > 4) pl.nip = 64;
    else return false;
    const int64_t unit = f16 ? 16 * DW16_DEPTH : 2 * DW_DEPTH;
    int64_t per = (P + DW_SPLIT - 1) / DW_SPLIT;
    per = (per + unit - 1) / unit * unit;
    if (per < 4 * unit) per = 4 * unit;
    if (pl.now == 4 && per > 512) per = 512;                   // bandwidth-bound head pass: many small slices per CU
    pl.per_wg = per;
    pl.nsplit = (int)((P + per - 1) / per);
    return true;
}


// ---------------------------------------------------------------------------- dW from half-precision operands
// Training with cfg.amd.train_dw_mode = 'f16' stores the two operands of every weight gradient -- the activations
// (training forward) and dZ (backward chain, in its power-of-two scaled domain) -- as f16: half the HBM bytes of the
// step's largest buffers, written once and read once.  dW = sum over ~10^5..10^6 samples of dZ[s][o] X[s][i] with
// each operand rounded to 11 bits (round to nearest, unbiased): the rounding errors of different samples are
// independent, so the relative error of the SUM falls with 1/sqrt(samples) (measured against fp64 by
// tests/test_gpu_grad.py).  Accumulation stays fp32.
//
// Both matrices are row-major [sample][column] and the MFMA contracts over samples, i.e. both operands want 8
// consecutive SAMPLES of one column per lane.  gfx950 has the instruction for exactly that: rows travel HBM ->
// registers -> LDS as they are (16-byte chunks, fully coalesced) and ds_read_b64_tr_b16 hands every lane 4 rows of its
// own column -- a transposed read, no gather loads, no VALU.  LDS image: 32-sample stages, 128-column blocks of 256-byte
// rows, 16-byte chunk index XORed with (row & 3) << 2 so that the 4 rows of a transposed read fall into 4 different
// bank quarters (64 columns: 128-byte rows, XOR ((row >> 1) & 1) << 2).  One workgroup per CU owns the whole
// output like the kernels above; two stages of global loads are in flight per thread (64 KiB per CU).
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short vs16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) vs16x4 lds_vs16x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int C>
__device__ __forceinline__ int dwh_off(int row, int col) {          // byte offset of (row, col) in a 32 x C stage image
    if (C >= 128) {
        const int c = col & 127;
        return (col >> 7) * (32 * 256) + row * 256 + 16 * ((c >> 3) ^ ((row & 3) << 2)) + 2 * (c & 7);
    }
    return row * 128 + 16 * ((col >> 3) ^ (((row >> 1) & 1) << 2)) + 2 * (col & 7);
}

// 8 consecutive samples (rows kr + 8 h .. + 7 of the stage) of column f0 + (lane & 31): one MFMA A / B operand
__device__ __forceinline__ h16x8 dwh_frag2(unsigned addr_lo, unsigned addr_hi) {
    const vs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_vs16x4*)(size_t)addr_lo);
    const vs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_vs16x4*)(size_t)addr_hi);
    union { vs16x4 v[2]; h16x8 h; } u;
    u.v[0] = lo;
    u.v[1] = hi;
    return u.h;
}

// OT: 32-row tiles of dZ columns per wave (n_out = 128 OT); IT: 32-column tiles of X (n_in padded to 32 IT: 2, 4, 8).
// ZBLK / XBLK: the matrix arrives in the BLOCKED layout the training kernels write (hnrf_mlp_f16.hip, save_pair_bh):
// 32-sample blocks of [32-feature tile][k = 8 groups of 4 features][32 sample slots][4 halves], sample c in slot
// c ^ 4 k.  A stage is then ONE contiguous block, copied to LDS as it is (linear 16-byte chunks), and the slot swizzle
// is what makes the transposed reads conflict-free: a read takes 4 samples x 8 groups per lane half, whose 8-byte
// granules (k 32 + (c ^ 4 k)) 8 cover the 64 banks once.  Rows past P are zero in a blocked dZ (the chain kernels
// store zeros there), so padded blocks need no masking.
template <int OT, int IT, bool ZBLK, bool XBLK>
__global__ __launch_bounds__(256) void mlp_dwh_kernel(const _Float16* __restrict__ dZ, int64_t ldz,
                                                      const _Float16* __restrict__ X, int64_t ldx, int64_t P,
                                                      int64_t per_wg, const float* __restrict__ dz_scale,
                                                      float* __restrict__ part, float* __restrict__ dbpart) {
    constexpr int NOW = 128 * OT, NIP = 32 * IT;
    constexpr int ZB = 32 * NOW * 2, XB = 32 * NIP * 2;          // bytes per stage
    constexpr int ZC = NOW / 64, XC = NIP / 64;                   // 16-byte chunks per thread and stage
    constexpr int ZROW = NOW >= 128 ? 256 : 128, XROW = NIP >= 128 ? 256 : 128;
    static_assert(!XBLK || NIP >= 128, "blocked X: hidden layers only");
    __shared__ __attribute__((aligned(16))) char lds[2][ZB + XB];
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&lds[0][0];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;

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

    // transposed-read addresses of this lane.  Row-major image: one per operand tile (the row part (kr + 4 j) rowbytes
    // and the 128-column block are immediates).  Blocked image: one per (k-step, j) -- the tile is an immediate.
    const int g = lane >> 4, pq = lane & 15, q = pq >> 2, pp = pq & 3;
    int aoff[OT], boff[IT < 4 ? IT : 4], bpos[2][2];
#pragma unroll
    for (int a = 0; a < OT; ++a) aoff[a] = dwh_off<NOW>(8 * (g >> 1) + q, (32 * OT * w + 32 * a) + 16 * (g & 1) + 4 * pp);
#pragma unroll
    for (int b = 0; b < (IT < 4 ? IT : 4); ++b) boff[b] = ZB + dwh_off<NIP>(8 * (g >> 1) + q, 32 * b + 16 * (g & 1) + 4 * pp);
    {
        const int k = pp + 4 * (g & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) bpos[ks][j] = (k * 32 + ((8 * (g >> 1) + q + 16 * ks + 4 * j) ^ (4 * k))) * 8;
    }

    u32x4 rz[2][ZC], rx[2][XC];
    auto fetch = [&](int slot, int64_t stage) {
        const int64_t sb = s0 + 32 * stage;                         // s0 and per_wg are multiples of 32: sb is a block start
#pragma unroll
        for (int i = 0; i < ZC; ++i) {
            const int id = tid + 256 * i;
            if (ZBLK) {
                rz[slot][i] = sb < s1 ? *reinterpret_cast<const u32x4*>(dZ + (sb >> 5) * (int64_t)(NOW * 32) + 8 * id) : u32x4{0u, 0u, 0u, 0u};
            } else {
                const int row = id / (NOW / 8), ch = id % (NOW / 8);
                rz[slot][i] = sb + row < s1 ? *reinterpret_cast<const u32x4*>(dZ + (sb + row) * ldz + 8 * ch) : u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int id = tid + 256 * i;
            if (XBLK) {
                rx[slot][i] = sb < s1 ? *reinterpret_cast<const u32x4*>(X + (sb >> 5) * (int64_t)(NIP * 32) + 8 * id) : u32x4{0u, 0u, 0u, 0u};
            } else {
                const int row = id / (NIP / 8), ch = id % (NIP / 8);
                rx[slot][i] = sb + row < s1 ? *reinterpret_cast<const u32x4*>(X + (sb + row) * ldx + 8 * ch) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    auto stage_in = [&](int slot, int buf) {
#pragma unroll
        for (int i = 0; i < ZC; ++i) {
            const int id = tid + 256 * i, row = id / (NOW / 8), ch = id % (NOW / 8);
            *reinterpret_cast<u32x4*>(&lds[buf][ZBLK ? 16 * id : dwh_off<NOW>(row, 8 * ch)]) = rz[slot][i];
        }
#pragma unroll
        for (int i = 0; i < XC; ++i) {
            const int id = tid + 256 * i, row = id / (NIP / 8), ch = id % (NIP / 8);
            *reinterpret_cast<u32x4*>(&lds[buf][ZB + (XBLK ? 16 * id : dwh_off<NIP>(row, 8 * ch))]) = rx[slot][i];
        }
    };
    auto compute = [&](int buf) {
        const unsigned lb = lbase + buf * (ZB + XB);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 af[OT];
#pragma unroll
            for (int a = 0; a < OT; ++a) {
                if (ZBLK) af[a] = dwh_frag2(lb + (OT * w + a) * 2048 + bpos[ks][0], lb + (OT * w + a) * 2048 + bpos[ks][1]);
                else af[a] = dwh_frag2(lb + aoff[a] + 16 * ks * ZROW, lb + aoff[a] + 16 * ks * ZROW + 4 * ZROW);
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    bsum[a] = __builtin_amdgcn_fdot2(h16x2{af[a][j], af[a][j + 1]}, h16x2{(_Float16)1.0f, (_Float16)1.0f}, bsum[a], false);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                h16x8 bf;
                if (XBLK) bf = dwh_frag2(lb + ZB + it * 2048 + bpos[ks][0], lb + ZB + it * 2048 + bpos[ks][1]);
                else {
                    const unsigned ba = lb + boff[it & 3] + (it >> 2) * (32 * 256) + 16 * ks * XROW;
                    bf = dwh_frag2(ba, ba + 4 * XROW);
                }
#pragma unroll
                for (int a = 0; a < OT; ++a) acc[a][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf, acc[a][it], 0, 0, 0);
            }
        }
    };

    const int64_t ns = per_wg / 32;                                // per_wg is a multiple of 64
    fetch(0, 0);
    fetch(1, 1);
    stage_in(0, 0);
    __syncthreads();
    for (int64_t s = 0; s < ns; s += 2) {
        fetch(0, s + 2);                 // slot 0 (stage s) already sits in LDS buffer 0
        compute(0);
        stage_in(1, 1);                  // stage s + 1 -> buffer 1 (last read before the previous barrier)
        __syncthreads();
        fetch(1, s + 3);
        compute(1);
        stage_in(0, 0);                  // stage s + 2 -> buffer 0
        __syncthreads();
    }

    const float inv = dz_scale ? 1.0f / dz_scale[0] : 1.0f;
    const int c = lane & 31, h = lane >> 5;
    float* out = part + (int64_t)blockIdx.x * NOW * NIP;
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = 32 * OT * w + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int it = 0; it < IT; ++it) out[(int64_t)o * NIP + 32 * it + c] = acc[a][it][r] * inv;
        }
    if (dbpart != nullptr) {
#pragma unroll
        for (int a = 0; a < OT; ++a) {
            const float t = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (h == 0) dbpart[(int64_t)blockIdx.x * NOW + 32 * OT * w + 32 * a + c] = t * inv;
        }
    }
}

// The same for BOTH operands in the blocked layout (the hidden layers of the training step: 12 of its 18 weight-gradient
// launches), fed by LDS-DMA instead of through registers.  A stage of the blocked layout is one contiguous run that is
// copied to LDS as it stands -- exactly what `global_load_lds_dwordx4` does -- so the staging registers (64 VGPRs), the
// ds_write instructions and their waits go, and with them the limit of two stages in flight: four LDS buffers (128 KiB
// for the 256 x 256 layers), three stages under way while the fourth is multiplied.  Each wave copies a contiguous 4-KiB
// (2-KiB) run of the dZ part and one of the X part per stage: one source base and one M0 per run, the pieces through the
// instruction's immediate offset (which applies to both addresses, profiles/tools/dma_offset.hip).  Stage order, and with it
// the summation order, is that of mlp_dwh_kernel: the results are bit-identical.
template <int R>
__device__ __forceinline__ void dwh_dma_piece(const char* gbase, unsigned voff, unsigned lds_addr) {
    // a scalar write of M0 needs one wait state before an LDS-DMA instruction reads it (gfx9 hazard), and hipcc, which
    // places the write, cannot see into the asm to insert it: hence the s_nop inside the statement.  In EVERY piece, not
    // only a run's first: where the pieces of a run sit in different basic blocks (runtime piece counts at layer
    // boundaries) the compiler writes M0 again in front of later pieces (the same value, so a stale read would be
    // harmless -- but that is an argument about today's code generation, not a guarantee)
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(gbase), "{m0}"(lds_addr), "n"(R * 1024) : "memory");
}
template <int NP>
__device__ __forceinline__ void dwh_dma_run(const char* gbase, unsigned voff, unsigned lds_addr) {
    static_assert(NP == 2 || NP == 4, "pieces per wave and operand");
    dwh_dma_piece<0>(gbase, voff, lds_addr);
    dwh_dma_piece<1>(gbase, voff, lds_addr);
    if (NP == 4) {
        dwh_dma_piece<2>(gbase, voff, lds_addr);
        dwh_dma_piece<3>(gbase, voff, lds_addr);
    }
}
__device__ __forceinline__ void dwh_wait_keep(int keep) {               // at most `keep` DMA pieces of this wave still in flight
#define HNRF_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (keep) {
        HNRF_VM(4) HNRF_VM(8) HNRF_VM(12) HNRF_VM(16) HNRF_VM(20) HNRF_VM(24) HNRF_VM(28) HNRF_VM(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef HNRF_VM
}

// LDS buffers = stages under way + 1.  Four: five (256 x 256: all 160 KiB) / eight (128 x 128) measured no faster --
// 5.6-5.7 TB/s of operand reads + partial-sum writes is what this access pattern gets from the HBM (guide: 6.0-6.3 for a
// pure stream)
constexpr int dwh_dma_bufs(int stage_bytes) { return 160 * 1024 / stage_bytes < 4 ? 160 * 1024 / stage_bytes : 4; }

template <int OT, int IT>
__global__ __launch_bounds__(256) void mlp_dwh_dma_kernel(const _Float16* __restrict__ dZ, const _Float16* __restrict__ X,
                                                          int64_t P, int64_t per_wg, const float* __restrict__ dz_scale,
                                                          float* __restrict__ part, float* __restrict__ dbpart) {
    constexpr int NOW = 128 * OT, NIP = 32 * IT;
    constexpr int ZB = 32 * NOW * 2, XB = 32 * NIP * 2, SB = ZB + XB;   // bytes per stage (32 samples)
    constexpr int ZP = ZB / 4096, XP = XB / 4096, PW = ZP + XP;         // 1-KiB pieces per wave and stage
    constexpr int NBUF = dwh_dma_bufs(SB), AHEAD = NBUF - 1;            // stages under way beside the one being multiplied
    static_assert(NIP >= 128, "blocked X: hidden layers only");
    static_assert((AHEAD - 1) * PW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char dwh_lds[];
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)dwh_lds);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;                    // multiple of 32: a block start
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;
    const int64_t ns = (s1 - s0 + 31) / 32;                             // stages of this slice (the last block may be ragged:
                                                                        // rows past P are zero in a blocked dZ)
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

    const int g = lane >> 4, pq = lane & 15, q = pq >> 2, pp = pq & 3;
    int bpos[2][2];
    {
        const int k = pp + 4 * (g & 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) bpos[ks][j] = (k * 32 + ((8 * (g >> 1) + q + 16 * ks + 4 * j) ^ (4 * k))) * 8;
    }
    const unsigned voff = lane * 16;
    auto issue = [&](int64_t stage) {                                   // wave-uniform
        const int64_t blk = (s0 >> 5) + stage;
        const unsigned dst = lbase + (unsigned)(stage % NBUF) * SB;
        dwh_dma_run<ZP>(reinterpret_cast<const char*>(dZ + blk * (int64_t)(NOW * 32)) + w * (ZP * 1024), voff, dst + w * (ZP * 1024));
        dwh_dma_run<XP>(reinterpret_cast<const char*>(X + blk * (int64_t)(NIP * 32)) + w * (XP * 1024), voff, dst + ZB + w * (XP * 1024));
    };
    auto compute = [&](int buf) {
        const unsigned lb = lbase + buf * SB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h16x8 af[OT];
#pragma unroll
            for (int a = 0; a < OT; ++a) {
                af[a] = dwh_frag2(lb + (OT * w + a) * 2048 + bpos[ks][0], lb + (OT * w + a) * 2048 + bpos[ks][1]);
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    bsum[a] = __builtin_amdgcn_fdot2(h16x2{af[a][j], af[a][j + 1]}, h16x2{(_Float16)1.0f, (_Float16)1.0f}, bsum[a], false);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const h16x8 bf = dwh_frag2(lb + ZB + it * 2048 + bpos[ks][0], lb + ZB + it * 2048 + bpos[ks][1]);
#pragma unroll
                for (int a = 0; a < OT; ++a) acc[a][it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf, acc[a][it], 0, 0, 0);
            }
        }
    };

#pragma unroll
    for (int d = 0; d < AHEAD; ++d)
        if (d < ns) issue(d);
    for (int64_t s = 0; s < ns; ++s) {
        const int64_t younger = ns - 1 - s < AHEAD - 1 ? ns - 1 - s : AHEAD - 1;   // stages issued after stage s
        dwh_wait_keep((int)younger * PW);                               // this wave's pieces of stage s have landed ...
        __syncthreads();                                                // ... and everybody's; everybody is done with stage s - 1
        if (s + AHEAD < ns) issue(s + AHEAD);                           // into the buffer of stage s - 1
        compute((int)(s % NBUF));
    }

    const float inv = dz_scale ? 1.0f / dz_scale[0] : 1.0f;
    const int c = lane & 31, h = lane >> 5;
    float* out = part + (int64_t)blockIdx.x * NOW * NIP;
#pragma unroll
    for (int a = 0; a < OT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = 32 * OT * w + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int it = 0; it < IT; ++it) out[(int64_t)o * NIP + 32 * it + c] = acc[a][it][r] * inv;
        }
    if (dbpart != nullptr) {
#pragma unroll
        for (int a = 0; a < OT; ++a) {
            const float t = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (h == 0) dbpart[(int64_t)blockIdx.x * NOW + 32 * OT * w + 32 * a + c] = t * inv;
        }
    }
}

// head layers with f16 activations: mlp_dw_head_kernel reading X as halves (XBLK: blocked layout, see mlp_dwh_kernel)
template <int NI, bool XBLK>
__global__ __launch_bounds__(NI) void mlp_dwh_head_kernel(const float* __restrict__ dY, int64_t ldy, int n_out,
                                                           const _Float16* __restrict__ X, int64_t ldx, int64_t P,
                                                           int64_t per_wg, float* __restrict__ part,
                                                           float* __restrict__ dbpart) {
    const int i = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b = 0.f;
    const int kx = ((i & 31) >> 3) * 2 + ((i >> 2) & 1);             // blocked: feature group of column i, slot swizzle 4 kx
    const _Float16* xp = XBLK ? X + (i >> 5) * 1024 + kx * 128 + (i & 3) : X + i;
    auto at = [&](int64_t s) { return XBLK ? (float)xp[(s >> 5) * (int64_t)(NI * 32) + (((int)(s & 31)) ^ (4 * kx)) * 4] : (float)xp[s * ldx]; };
    auto step = [&](int64_t s, float x) {
        const float* g = dY + s * ldy;
        const float g0 = g[0], g1 = n_out > 1 ? g[1] : 0.f, g2 = n_out > 2 ? g[2] : 0.f, g3 = n_out > 3 ? g[3] : 0.f;
        a0 = fmaf(g0, x, a0);
        a1 = fmaf(g1, x, a1);
        a2 = fmaf(g2, x, a2);
        a3 = fmaf(g3, x, a3);
        b += i == 0 ? g0 : (i == 1 ? g1 : (i == 2 ? g2 : g3));
    };
    int64_t s = s0;
    for (; s + 8 <= s1; s += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = at(s + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) step(s + u, x[u]);
    }
    for (; s < s1; ++s) step(s, at(s));
    float* out = part + (int64_t)blockIdx.x * 4 * NI;
    out[i] = a0;
    out[NI + i] = a1;
    out[2 * NI + i] = a2;
    out[3 * NI + i] = a3;
    if (dbpart != nullptr && i < 4) dbpart[(int64_t)blockIdx.x * 4 + i] = b;
}

// Head layers, blocked f16 activations, coalesced: the 256 threads of a block take the 256 8-byte granules of one
// 32-feature tile of a 32-sample block (thread = (feature group k, sample slot): 2 KiB contiguous per step), every
// thread keeps 4 features x 4 outputs per tile for ITS sample slot, and the 32 slots of a group are summed by wave
// shuffles once at the end.  Partials in the layout of mlp_dw_head_kernel.
template <int NI>
__global__ __launch_bounds__(256, 2) void mlp_dwh_head_blk_kernel(const float* __restrict__ dY, int64_t ldy, int n_out,
                                                                  const _Float16* __restrict__ X, int64_t P, int64_t per_wg,
                                                                  float* __restrict__ part, float* __restrict__ dbpart) {
    // Pure streaming: 64 B of X and 16 B of dY per thread and trip.  Round 3: the kernel ran one wave per SIMD (the 128
    // accumulators + the compiler's free hand up to 512 registers) with the loads of a trip issued and awaited inside it --
    // 16 KB in flight per CU, 2.4 TB/s.  Now two workgroups per CU (register cap of launch_bounds) and the next trip's
    // loads issued before this trip's FMAs.
    constexpr int NTILE = NI / 32;
    const int tid = threadIdx.x, k = tid >> 5, slot = tid & 31, c = slot ^ (4 * k);
    const int64_t s0 = (int64_t)blockIdx.x * per_wg;                 // multiple of 32
    const int64_t s1 = s0 + per_wg < P ? s0 + per_wg : P;
    float acc[NTILE][4][4];
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[t][j][o] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    h16x4 xn[NTILE];
    float gn[4] = {0.f, 0.f, 0.f, 0.f};
    // whole 32-sample blocks: unconditional loads (a per-lane `s < s1` around them made the compiler wait for dY inside
    // the trip); the one ragged block at the end of the last slice is done separately below
    auto fetch = [&](int64_t sb) {
        const float* gp = dY + (sb + c) * ldy;
#pragma unroll
        for (int o = 0; o < 4; ++o) gn[o] = gp[o < n_out ? o : 0];       // (masked where consumed: a select here waits for the load)
        const _Float16* xb = X + (sb >> 5) * (int64_t)(NI * 32) + tid * 4;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) xn[t] = *reinterpret_cast<const h16x4*>(xb + t * 1024);
    };
    auto accumulate = [&](const float (&g)[4], const h16x4 (&xv)[NTILE]) {
#pragma unroll
        for (int o = 0; o < 4; ++o) bsum[o] += g[o];
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = (float)xv[t][j];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[t][j][o] = fmaf(g[o], x, acc[t][j][o]);
            }
    };
    const int64_t sfull = s1 > s0 ? s0 + ((s1 - s0) & ~(int64_t)31) : s0;
    if (s0 < sfull) fetch(s0);
    for (int64_t sb = s0; sb < sfull; sb += 32) {
        float g[4];
        h16x4 xv[NTILE];
#pragma unroll
        for (int o = 0; o < 4; ++o) g[o] = o < n_out ? gn[o] : 0.f;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) xv[t] = xn[t];
        if (sb + 32 < sfull) fetch(sb + 32);
        accumulate(g, xv);
    }
    if (sfull < s1) {
        const int64_t s = sfull + c;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (s < s1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) g[o] = o < n_out ? dY[s * ldy + o] : 0.f;
        }
        h16x4 xv[NTILE];
        const _Float16* xb = X + (sfull >> 5) * (int64_t)(NI * 32) + tid * 4;
#pragma unroll
        for (int t = 0; t < NTILE; ++t) xv[t] = *reinterpret_cast<const h16x4*>(xb + t * 1024);
        accumulate(g, xv);
    }
    // sum over the 32 sample slots of each feature group (a half wave)
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float v = acc[t][j][o];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
                acc[t][j][o] = v;
            }
    float* out = part + (int64_t)blockIdx.x * 4 * NI;
    if (slot == 0) {
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 32 * t + 8 * (k >> 1) + 4 * (k & 1) + j;
#pragma unroll
                for (int o = 0; o < 4; ++o) out[o * NI + f] = acc[t][j][o];
            }
    }
    if (dbpart != nullptr && k == 0) {                               // feature group 0 saw every sample once
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float v = bsum[o];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (slot == 0) dbpart[(int64_t)blockIdx.x * 4 + o] = v;
        }
    }
}

static bool dwh_plan(int64_t P, int n_out, int n_in, DwPlan& pl) {
    if (n_out == 256 || n_out == 128) pl.now = n_out; else if (n_out >= 1 && n_out <= 4) pl.now = 4; else return false;
    if (n_in == 256 || n_in == 128) pl.nip = n_in;
    else if (n_in >= 1 && n_in <= 64 && n_out > 4) pl.nip = 64;
    else return false;
    int64_t per = (P + DW_SPLIT - 1) / DW_SPLIT;
    per = (per + 63) / 64 * 64;
    if (per < 128) per = 128;
    if (pl.now == 4) {                       // head kernels: 512 workgroups = two per CU, all resident at once
        per = ((P + 511) / 512 + 31) / 32 * 32;
        if (per < 128) per = 128;
    }
    pl.per_wg = per;
    pl.nsplit = (int)((P + per - 1) / per);
    return true;
}

// =========================================================================== dX chain
// Backward through all layers of one MLP for 32 samples per wave, register-resident like the
// forward (hnrf_mlp.hip): dH_{l-1}^T [in-features x samples] = W_l^T . dZ_l^T, then
// dZ_{l-1} = dH_{l-1} * [act_{l-1} > 0] with act read from the matrices the training forward
// saved.  The MFMA result of one stage is, verbatim, the B operand of the next; the weight image
// holds W_l^T in A-operand order with the K order of that register layout (hid_feat), and the
// rows of the two positional-encoding blocks (layer 0 and the skip layer) in the order of the
// forward's PE K-steps, so that d PE lands on the lane that can recompute sin/cos of its own
// sample: the PE backward is fused (d_xyz leaves the kernel, d PE never exists in memory).
// Every dZ_l is written once ([L][P][width], the layout of the saved activations) for
// hnrf_mlp_dw.  Per sample: 8 KB of activations read + 8 KB of dZ written (canonical).

// HNRF_AMAX_SLOTS (hnrf.h): per-layer |dZ| maxima are spread over this many atomic targets

struct PackBwd {
    const float* W;        // nn.Linear weight (n_out, n_in) of the forward layer
    int n_out, n_in;
    int NT, NG;            // output tiles (rows = forward in-features) / K groups (K = forward out-features)
    int row_kind;          // PE_NONE: row rho <-> in-feature col0 + rho;  else: rows in PE K-step order
    int col0;              // first forward in-feature column of this block
    int head;              // 1: K = the n_out (<= 4) head outputs on lane half 0 of steps 0..3
    int64_t w_off;         // float offset in the image
};

__global__ void pack_bwd_kernel(PackBwd d, float* __restrict__ packed) {
    const int64_t n = (int64_t)d.NT * d.NG * 256;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    const int64_t gg = i >> 8;
    const int g = (int)(gg % d.NG);
    const int t = (int)(gg / d.NG);
    const int h = lane >> 5;
    const int j = 4 * g + e;                                   // K-step
    const int o = d.head ? ((h == 0 && j < d.n_out) ? j : -1) : hid_feat(j, h);
    const int rho = 32 * t + (lane & 31);                      // output row of this stage
    int col;
    if (d.row_kind == PE_NONE) {
        col = d.col0 + rho;
    } else {
        // register (tile tt, r) on half hh holds row 32 tt + 8 (r >> 2) + 4 hh + (r & 3): make it PE step 16 tt + r
        const int tt = rho >> 5, w = rho & 31;
        const int r = 4 * (w >> 3) + (w & 3), hh = (w >> 2) & 1;
        const int c = pe_col(d.row_kind, 16 * tt + r, hh);
        col = c < 0 ? -1 : d.col0 + c;
    }
    float v = 0.f;
    if (o >= 0 && o < d.n_out && col >= 0 && col < d.n_in) v = d.W[(int64_t)o * d.n_in + col];
    packed[d.w_off + i] = v;
}

// One backward stage.  b: dZ of the layer above in register layout; out: NT tiles of W^T b.
// mbits != nullptr: multiply by relu' -- this lane's NT/2 words of the sign mask the training forward wrote
// (bit 16 (t & 1) + r of word t >> 1 = [register r of tile t was positive]); 16 bytes per lane and layer instead
// of re-reading the activations: a load that waits on HBM holds back, in the in-order vmcnt queue, the L2-resident
// weight stream issued behind it, so the stage has exactly one such load, at its start.
// dz != nullptr: store the masked result (this lane's row of the dZ matrix).
template <int PH, int NT, int NG, int NB, int NO>
__device__ __forceinline__ void bwd_stage(const float4* __restrict__& wptr, float4 (&ring)[PF], const float (&b)[NB],
                                          float (&out)[NO], const uint32_t* __restrict__ mbits,
                                          float* __restrict__ dz, int h, float* __restrict__ amax_slot = nullptr) {
    static_assert(NB >= NG * 4 && NO >= NT * 16, "operand arrays too small");
    uint32_t mw[(NT + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NT + 1) / 2; ++i) mw[i] = mbits != nullptr ? mbits[i] : 0xffffffffu;
    float stage_max = 0.f;                       // largest |dZ| of this stage (scale of the split-f16 dW kernel)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot = (PH + t * NG + g) % PF;   // PH: groups consumed before this stage, mod PF
            const float4 w = ring[slot];
            ring[slot] = wptr[PF * 64];
            wptr += 64;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b[4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b[4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b[4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b[4 * g + 3], acc, 0, 0, 0);
            // pin the weight load PF groups ahead of its use: at the 256-VGPR limit hipcc otherwise sinks it
            // to the consuming MFMA (load; s_waitcnt vmcnt(0); mfma) to shorten the live range
            __builtin_amdgcn_sched_barrier(0);
        }
        const uint32_t m = mw[t >> 1] >> (16 * (t & 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) out[t * 16 + r] = (m >> r) & 1u ? acc[r] : 0.f;
        if (amax_slot != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage_max = fmaxf(stage_max, fabsf(out[t * 16 + r]));
        }
        if (dz != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(dz + 32 * t + 8 * q + 4 * h) =
                    make_float4(out[t * 16 + 4 * q], out[t * 16 + 4 * q + 1], out[t * 16 + 4 * q + 2],
                                out[t * 16 + 4 * q + 3]);
        }
    }
    if (amax_slot != nullptr) {                  // one atomic per wave and layer, spread over 64 slots per layer
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) stage_max = fmaxf(stage_max, __shfl_xor(stage_max, off, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax_slot), __float_as_uint(stage_max));
    }
}

// image layouts (floats); every block is [tile][group][lane][4]
constexpr int64_t CB_HEAD = 0;                                    // W8^T: 8 tiles x 1 group
constexpr int64_t CB_MID = 8 * 32 * 256;                          // one 256x256 block
constexpr int64_t CB_L7 = CB_HEAD + 8 * 1 * 256;                  // W7^T, W6^T
constexpr int64_t CB_L5H = CB_L7 + 2 * CB_MID;                    // W5^T hidden rows
constexpr int64_t CB_L5P = CB_L5H + CB_MID;                       // W5^T PE rows: 2 tiles x 32 groups
constexpr int64_t CB_L4 = CB_L5P + 2 * 32 * 256;                  // W4^T .. W1^T
constexpr int64_t CB_L0P = CB_L4 + 4 * CB_MID;                    // W0^T PE rows
constexpr int64_t CB_END = CB_L0P + 2 * 32 * 256;
constexpr int64_t CB_FLOATS = CB_END + PF * 256;                  // prefetch over-run pad
constexpr int64_t NB_HEAD = 0;                                    // W6^T: 4 tiles x 1 group
constexpr int64_t NB_MID = 4 * 16 * 256;
constexpr int64_t NB_L5 = NB_HEAD + 4 * 1 * 256;                  // W5^T
constexpr int64_t NB_L4H = NB_L5 + NB_MID;                        // W4^T hidden rows
constexpr int64_t NB_L4P = NB_L4H + NB_MID;                       // W4^T PE rows: 2 tiles x 16 groups
constexpr int64_t NB_L3 = NB_L4P + 2 * 16 * 256;                  // W3^T .. W1^T
constexpr int64_t NB_L0P = NB_L3 + 3 * NB_MID;                    // W0^T PE rows
constexpr int64_t NB_END = NB_L0P + 2 * 16 * 256;
constexpr int64_t NB_FLOATS = NB_END + PF * 256;

__device__ __forceinline__ void ring_fill_b(const float4* __restrict__ wptr, float4 (&ring)[PF]) {
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wptr[i * 64];
}

// Canonical MLP.  relu_bits [8][P][2][4] (sign masks of the saved activations), d_raw [P,4]; writes
// dZ [8][P][256], d_xyz [P,3].
__global__ __launch_bounds__(256) void canonical_bwd_kernel(const float* __restrict__ xyz,
                                                            const float4* __restrict__ d_raw,
                                                            const uint32_t* __restrict__ relu_bits,
                                                            const float* __restrict__ packed, int64_t P,
                                                            float* __restrict__ dZ, float* __restrict__ d_xyz,
                                                            float* __restrict__ dz_amax) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (lane & 31);
    const int64_t sample = slot < P ? slot : P - 1;            // clamped lanes compute, but never store
    // dz_amax [8][HNRF_AMAX_SLOTS]: clamped lanes repeat sample P-1, which cannot raise a maximum
    float* am = dz_amax ? dz_amax + 7 * HNRF_AMAX_SLOTS + (blockIdx.x % HNRF_AMAX_SLOTS) : nullptr;
    const bool live = slot < P;
    const int64_t stride = P * 256;

    const float4 g = d_raw[sample];
    const float in0[4] = {h ? 0.f : g.x, h ? 0.f : g.y, h ? 0.f : g.z, h ? 0.f : g.w};
    const float4* wptr = reinterpret_cast<const float4*>(packed) + lane;
    float4 ring[PF];
    ring_fill_b(wptr, ring);

    const int64_t bstride = P * 8;
    const uint32_t* act = relu_bits + 7 * bstride + sample * 8 + h * 4;
    float* dz = dZ + 7 * stride + sample * 256;
    float hA[128], hB[128], dpe[32], dpe0[32];
    constexpr int C_PH = 8 % PF;                               // after the 8-group head every stage is a multiple of 64 groups
    static_assert(64 % PF == 0, "stages must keep the ring phase");
    bwd_stage<0, 8, 1>(wptr, ring, in0, hA, act, live ? dz : nullptr, h, am);                  // dZ7
#pragma unroll 1
    for (int l = 7; l >= 6; --l) {                                                      // dZ6, dZ5
        act -= bstride;
        dz -= stride;
        if (am) am -= HNRF_AMAX_SLOTS;
        bwd_stage<C_PH, 8, 32>(wptr, ring, hA, hB, act, live ? dz : nullptr, h, am);
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = hB[i];
    }
    act -= bstride;
    dz -= stride;
    if (am) am -= HNRF_AMAX_SLOTS;
    bwd_stage<C_PH, 8, 32>(wptr, ring, hA, hB, act, live ? dz : nullptr, h, am);                  // skip layer: dZ4 ...
    bwd_stage<C_PH, 2, 32>(wptr, ring, hA, dpe, nullptr, nullptr, h);                         // ... and its d PE
#pragma unroll
    for (int i = 0; i < 128; ++i) hA[i] = hB[i];
#pragma unroll 1
    for (int l = 4; l >= 1; --l) {                                                      // dZ3 .. dZ0
        act -= bstride;
        dz -= stride;
        if (am) am -= HNRF_AMAX_SLOTS;
        bwd_stage<C_PH, 8, 32>(wptr, ring, hA, hB, act, live ? dz : nullptr, h, am);
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = hB[i];
    }
    bwd_stage<C_PH, 2, 32>(wptr, ring, hA, dpe0, nullptr, nullptr, h);                        // layer 0's d PE
#pragma unroll
    for (int j = 0; j < 32; ++j) dpe[j] += dpe0[j];

    // PE backward on the lane's own sample (fourier.py): step j = 3 k + axis, half 0 = sin, half 1 = cos
    const float x[3] = {xyz[sample * 3 + 0], xyz[sample * 3 + 1], xyz[sample * 3 + 2]};
    float dx[3] = {0.f, 0.f, 0.f};
    {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int j = 0; j < 30; ++j) {
            float sv, cv;
            ph[j % 3].next(sv, cv);
            const float f = (float)(1 << (j / 3));
            dx[j % 3] += f * (h ? -sv : cv) * dpe[j];
        }
    }
    dx[h ? 1 : 0] += dpe[30];
    if (h == 0) dx[2] += dpe[31];
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] += __shfl_xor(dx[a], 32, 64);
    if (h == 0 && live) {
        d_xyz[sample * 3 + 0] = dx[0];
        d_xyz[sample * 3 + 1] = dx[1];
        d_xyz[sample * 3 + 2] = dx[2];
    }
}

// Non-rigid MLP (xyz = x_skel + offset(x_skel)).  relu_bits [6][P][2][2], d_xyz [P,3]; writes dZ [6][P][128] and
// d_x_skel = d_xyz + (d offset / d x_skel)^T d_xyz.
__global__ __launch_bounds__(256) void nonrigid_bwd_kernel(const float* __restrict__ x_skel,
                                                           const float* __restrict__ hann_w,
                                                           const float* __restrict__ d_xyz,
                                                           const uint32_t* __restrict__ relu_bits,
                                                           const float* __restrict__ packed, int64_t P,
                                                           float* __restrict__ dZ, float* __restrict__ d_x_skel,
                                                           float* __restrict__ dz_amax) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (lane & 31);
    const int64_t sample = slot < P ? slot : P - 1;
    float* am = dz_amax ? dz_amax + 5 * HNRF_AMAX_SLOTS + (blockIdx.x % HNRF_AMAX_SLOTS) : nullptr;
    const bool live = slot < P;
    const int64_t stride = P * 128;

    const float g[3] = {d_xyz[sample * 3 + 0], d_xyz[sample * 3 + 1], d_xyz[sample * 3 + 2]};
    const float in0[4] = {h ? 0.f : g[0], h ? 0.f : g[1], h ? 0.f : g[2], 0.f};
    const float4* wptr = reinterpret_cast<const float4*>(packed) + lane;
    float4 ring[PF];
    ring_fill_b(wptr, ring);

    const int64_t bstride = P * 4;
    const uint32_t* act = relu_bits + 5 * bstride + sample * 4 + h * 2;
    float* dz = dZ + 5 * stride + sample * 128;
    float hA[64], hB[64], dpe[32], dpe0[32];
    constexpr int N_PH = 4 % PF;                               // 4-group head, then multiples of 32 groups
    static_assert(32 % PF == 0, "stages must keep the ring phase");
    bwd_stage<0, 4, 1>(wptr, ring, in0, hA, act, live ? dz : nullptr, h, am);                  // dZ5
    act -= bstride;
    dz -= stride;
    if (am) am -= HNRF_AMAX_SLOTS;
    bwd_stage<N_PH, 4, 16>(wptr, ring, hA, hB, act, live ? dz : nullptr, h, am);                  // dZ4 (the skip layer's)
    act -= bstride;
    dz -= stride;
    if (am) am -= HNRF_AMAX_SLOTS;
    bwd_stage<N_PH, 4, 16>(wptr, ring, hB, hA, act, live ? dz : nullptr, h, am);                  // skip [h | PE]: dZ3 ...
    bwd_stage<N_PH, 2, 16>(wptr, ring, hB, dpe, nullptr, nullptr, h);                         // ... and its d PE
#pragma unroll 1
    for (int l = 3; l >= 1; --l) {                                                      // dZ2 .. dZ0
        act -= bstride;
        dz -= stride;
        if (am) am -= HNRF_AMAX_SLOTS;
        bwd_stage<N_PH, 4, 16>(wptr, ring, hA, hB, act, live ? dz : nullptr, h, am);
#pragma unroll
        for (int i = 0; i < 64; ++i) hA[i] = hB[i];
    }
    bwd_stage<N_PH, 2, 16>(wptr, ring, hA, dpe0, nullptr, nullptr, h);
#pragma unroll
    for (int j = 0; j < 18; ++j) dpe[j] += dpe0[j];

    const float x[3] = {x_skel[sample * 3 + 0], x_skel[sample * 3 + 1], x_skel[sample * 3 + 2]};
    float dx[3] = {0.f, 0.f, 0.f};
    {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            float sv, cv;
            ph[j % 3].next(sv, cv);
            const float f = hann_w[j / 3] * (float)(1 << (j / 3));
            dx[j % 3] += f * (h ? -sv : cv) * dpe[j];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] += __shfl_xor(dx[a], 32, 64);
    if (h == 0 && live) {
        d_x_skel[sample * 3 + 0] = g[0] + dx[0];
        d_x_skel[sample * 3 + 1] = g[1] + dx[1];
        d_x_skel[sample * 3 + 2] = g[2] + dx[2];
    }
}

static int launch_pack_bwd(const PackBwd& d, float* packed, hipStream_t st) {
    const int64_t n = (int64_t)d.NT * d.NG * 256;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, packed);
    return check_launch("hnrf pack (backward)");
}

}  // namespace hnrf

using namespace hnrf;

extern "C" size_t hnrf_mlp_dw_workspace_bytes(int64_t P, int n_out, int n_in) {
    DwPlan pl;
    if (P <= 0 || !dw_plan(P, n_out, n_in, pl)) return 0;
    return ((size_t)pl.nsplit * pl.now * pl.nip + (size_t)pl.nsplit * pl.now) * sizeof(float);
}

extern "C" int hnrf_mlp_dw(const float* dZ, int64_t ldz, const float* X, int64_t ldx, int64_t P, int n_out, int n_in,
                           int mode, const float* dz_amax, int n_amax, float* dW, int64_t ldw, float* db,
                           void* workspace, size_t workspace_bytes, void* stream) {
    HNRF_REQUIRE(dZ && X && dW && workspace, HNRF_E_ARG, "hnrf_mlp_dw: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED, "hnrf_mlp_dw: mode %d not built", mode);
    // split-f16 is built for the matrix-shaped layers; positional-encoding blocks and heads stay on the fp32 kernels
    const bool f16 = mode == HNRF_MLP_F16X3 && (n_out == 128 || n_out == 256) && (n_in == 128 || n_in == 256);
    HNRF_REQUIRE(!f16 || (dz_amax && n_amax >= 1 && n_amax <= 4096), HNRF_E_ARG,
                 "hnrf_mlp_dw: HNRF_MLP_F16X3 needs dz_amax (n_amax device floats whose maximum bounds |dZ|)");
    DwPlan pl;
    HNRF_REQUIRE(P > 0 && dw_plan(P, n_out, n_in, pl, f16), HNRF_E_UNSUPPORTED,
                 "hnrf_mlp_dw: shape P=%lld n_out=%d n_in=%d not built (n_out 128|256 with n_in 128|256|<=64; n_out <= 4 with n_in 128|256)",
                 (long long)P, n_out, n_in);
    HNRF_REQUIRE(ldz >= n_out && ldx >= n_in && ldw >= n_in, HNRF_E_ARG, "hnrf_mlp_dw: row stride below width");
    HNRF_REQUIRE(workspace_bytes >= hnrf_mlp_dw_workspace_bytes(P, n_out, n_in), HNRF_E_ARG,
                 "hnrf_mlp_dw: workspace too small");
    if (n_in > 64 && n_out > 4)
        HNRF_REQUIRE(((uintptr_t)X & 15) == 0 && ldx % 4 == 0, HNRF_E_ARG,
                     "hnrf_mlp_dw: X must be 16-byte aligned with a row stride that is a multiple of 4");
    float* part = (float*)workspace;
    float* dbpart = part + (size_t)pl.nsplit * pl.now * pl.nip;
    hipStream_t st = (hipStream_t)stream;
#define HNRF_DW(OT, IB)                                                                                        \
    hipLaunchKernelGGL((mlp_dw_kernel<OT, IB>), dim3(pl.nsplit), dim3(256), 0, st, dZ, ldz, X, ldx, n_in, P, \
                       pl.per_wg, part, db ? dbpart : nullptr)
#define HNRF_DW16(OT, IB)                                                                                       \
    hipLaunchKernelGGL((mlp_dw16_kernel<OT, IB>), dim3(pl.nsplit), dim3(256), 0, st, dZ, ldz, X, ldx, P, pl.per_wg, \
                       dz_amax, n_amax, part, db ? dbpart : nullptr)
    if (f16) {
        if (n_out == 256) { if (n_in == 256) HNRF_DW16(2, 2); else HNRF_DW16(2, 1); }
        else { if (n_in == 256) HNRF_DW16(1, 2); else HNRF_DW16(1, 1); }
    } else if (n_out <= 4) {
        if (n_in == 256)
            hipLaunchKernelGGL(mlp_dw_head_kernel<256>, dim3(pl.nsplit), dim3(256), 0, st, dZ, ldz, n_out, X, ldx, P,
                               pl.per_wg, part, db ? dbpart : nullptr);
        else
            hipLaunchKernelGGL(mlp_dw_head_kernel<128>, dim3(pl.nsplit), dim3(128), 0, st, dZ, ldz, n_out, X, ldx, P,
                               pl.per_wg, part, db ? dbpart : nullptr);
    } else if (n_out == 256) {
        if (n_in == 256) HNRF_DW(2, 2); else if (n_in == 128) HNRF_DW(2, 1); else HNRF_DW(2, 0);
    } else {
        if (n_in == 256) HNRF_DW(1, 2); else if (n_in == 128) HNRF_DW(1, 1); else HNRF_DW(1, 0);
    }
#undef HNRF_DW
#undef HNRF_DW16
    int rc = check_launch("hnrf_mlp_dw");
    if (rc) return rc;
    const int nblk = pl.now * pl.nip / 64 + (db ? (pl.now + 63) / 64 : 0);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(nblk), dim3(256), 0, st, part, dbpart, pl.nsplit, pl.now,
                       pl.nip, n_out, n_in, dW, ldw, db);
    return check_launch("hnrf_mlp_dw (reduce)");
}

extern "C" size_t hnrf_mlp_dw_h_workspace_bytes(int64_t P, int n_out, int n_in) {
    DwPlan pl;
    if (P <= 0 || !dwh_plan(P, n_out, n_in, pl)) return 0;
    return ((size_t)pl.nsplit * pl.now * pl.nip + (size_t)pl.nsplit * pl.now) * sizeof(float);
}

extern "C" int hnrf_mlp_dw_h(const void* dZ, int64_t ldz, const void* X, int64_t ldx, int64_t P, int n_out, int n_in,
                             int layout, const float* dz_scale, float* dW, int64_t ldw, float* db, void* workspace,
                             size_t workspace_bytes, void* stream) {
    HNRF_REQUIRE(dZ && X && dW && workspace, HNRF_E_ARG, "hnrf_mlp_dw_h: null pointer");
    HNRF_REQUIRE((layout & ~(HNRF_DWH_DZ_BLOCKED | HNRF_DWH_X_BLOCKED)) == 0, HNRF_E_ARG, "hnrf_mlp_dw_h: bad layout %d", layout);
    const bool zb = layout & HNRF_DWH_DZ_BLOCKED, xb = layout & HNRF_DWH_X_BLOCKED;
    DwPlan pl;
    HNRF_REQUIRE(P > 0 && dwh_plan(P, n_out, n_in, pl), HNRF_E_UNSUPPORTED,
                 "hnrf_mlp_dw_h: shape P=%lld n_out=%d n_in=%d not built (n_out 128|256 with n_in 128|256|<=64; n_out <= 4 "
                 "with n_in 128|256)", (long long)P, n_out, n_in);
    HNRF_REQUIRE(workspace_bytes >= hnrf_mlp_dw_h_workspace_bytes(P, n_out, n_in), HNRF_E_ARG,
                 "hnrf_mlp_dw_h: workspace too small");
    float* part = (float*)workspace;
    float* dbpart = part + (size_t)pl.nsplit * pl.now * pl.nip;
    hipStream_t st = (hipStream_t)stream;
    if (n_out <= 4) {
        // dZ is the fp32 [P, n_out] gradient at the head's output; X the f16 activations of the last hidden layer
        HNRF_REQUIRE(!zb && ldz >= n_out && (xb || ldx >= n_in) && ldw >= n_in, HNRF_E_ARG, "hnrf_mlp_dw_h: bad head operands");
#define HNRF_DWHH(NI, XB)                                                                                              \
    hipLaunchKernelGGL((mlp_dwh_head_kernel<NI, XB>), dim3(pl.nsplit), dim3(NI), 0, st, (const float*)dZ, ldz, n_out, \
                       (const _Float16*)X, ldx, P, pl.per_wg, part, db ? dbpart : nullptr)
        if (xb) {
            if (n_in == 256)
                hipLaunchKernelGGL(mlp_dwh_head_blk_kernel<256>, dim3(pl.nsplit), dim3(256), 0, st, (const float*)dZ, ldz, n_out,
                                   (const _Float16*)X, P, pl.per_wg, part, db ? dbpart : nullptr);
            else
                hipLaunchKernelGGL(mlp_dwh_head_blk_kernel<128>, dim3(pl.nsplit), dim3(256), 0, st, (const float*)dZ, ldz, n_out,
                                   (const _Float16*)X, P, pl.per_wg, part, db ? dbpart : nullptr);
        } else if (n_in == 256) HNRF_DWHH(256, false);
        else HNRF_DWHH(128, false);
#undef HNRF_DWHH
    } else {
        HNRF_REQUIRE((zb || ldz >= pl.now) && (xb || ldx >= pl.nip) && ldw >= n_in, HNRF_E_ARG,
                     "hnrf_mlp_dw_h: row stride below the padded width (X rows must hold %d halves)", pl.nip);
        HNRF_REQUIRE((((uintptr_t)dZ | (uintptr_t)X) & 15) == 0 && (zb || ldz % 8 == 0) && (xb || ldx % 8 == 0), HNRF_E_ARG,
                     "hnrf_mlp_dw_h: dZ and X must be 16-byte aligned with row strides that are multiples of 8 halves");
        HNRF_REQUIRE(!xb || pl.nip >= 128, HNRF_E_UNSUPPORTED, "hnrf_mlp_dw_h: blocked X needs n_in 128 | 256");
#define HNRF_DWH1(OT, IT, ZB_, XB_)                                                                                    \
    hipLaunchKernelGGL((mlp_dwh_kernel<OT, IT, ZB_, XB_>), dim3(pl.nsplit), dim3(256), 0, st, (const _Float16*)dZ, ldz, \
                       (const _Float16*)X, ldx, P, pl.per_wg, dz_scale, part, db ? dbpart : nullptr)
#define HNRF_DWH_DMA(OT, IT)                                                                                         \
    do {                                                                                                             \
        constexpr int lds = dwh_dma_bufs(32 * 128 * OT * 2 + 32 * 32 * IT * 2) * (32 * 128 * OT * 2 + 32 * 32 * IT * 2);  \
        static unsigned long long done = 0;                                                                          \
        if (int rc_ = reserve_lds((const void*)mlp_dwh_dma_kernel<OT, IT>, lds, done, "hnrf_mlp_dw_h")) return rc_;   \
        hipLaunchKernelGGL((mlp_dwh_dma_kernel<OT, IT>), dim3(pl.nsplit), dim3(256), lds, st, (const _Float16*)dZ,    \
                           (const _Float16*)X, P, pl.per_wg, dz_scale, part, db ? dbpart : nullptr);                  \
    } while (0)
#define HNRF_DWH(OT, IT)                                                       \
    do {                                                                       \
        if (zb && xb) { if (IT >= 4) { if (use_dma) HNRF_DWH_DMA(OT, (IT >= 4 ? IT : 4)); else HNRF_DWH1(OT, (IT >= 4 ? IT : 4), true, true); } } \
        else if (zb) HNRF_DWH1(OT, IT, true, false);                           \
        else HNRF_DWH1(OT, IT, false, false);                                  \
    } while (0)
        // (HNRF_DWH_NO_DMA in the environment: the register-staged form, for A/B runs)
        static const bool use_dma = getenv("HNRF_DWH_NO_DMA") == nullptr;
        HNRF_REQUIRE(zb || !xb, HNRF_E_UNSUPPORTED, "hnrf_mlp_dw_h: blocked X with row-major dZ is not built");
        if (n_out == 256) { if (pl.nip == 256) HNRF_DWH(2, 8); else if (pl.nip == 128) HNRF_DWH(2, 4); else HNRF_DWH(2, 2); }
        else { if (pl.nip == 256) HNRF_DWH(1, 8); else if (pl.nip == 128) HNRF_DWH(1, 4); else HNRF_DWH(1, 2); }
#undef HNRF_DWH
#undef HNRF_DWH_DMA
#undef HNRF_DWH1
    }
    int rc = check_launch("hnrf_mlp_dw_h");
    if (rc) return rc;
    const int nblk = pl.now * pl.nip / 64 + (db ? (pl.now + 63) / 64 : 0);
    hipLaunchKernelGGL(mlp_dw_reduce_kernel, dim3(nblk), dim3(256), 0, st, part, dbpart, pl.nsplit, pl.now,
                       pl.nip, n_out, n_in, dW, ldw, db);
    return check_launch("hnrf_mlp_dw_h (reduce)");
}

extern "C" size_t hnrf_canonical_bwd_packed_bytes(int mode) {
    if (mode == HNRF_MLP_F16X3) return canonical16_bwd_bytes();
    return mode == HNRF_MLP_F32 ? (size_t)CB_FLOATS * sizeof(float) : 0;
}
extern "C" size_t hnrf_nonrigid_bwd_packed_bytes(int mode) {
    if (mode == HNRF_MLP_F16X3) return nonrigid16_bwd_bytes();
    return mode == HNRF_MLP_F32 ? (size_t)NB_FLOATS * sizeof(float) : 0;
}

extern "C" int hnrf_canonical_bwd_pack(const float* const* weights, int mode, void* packed, void* stream) {
    HNRF_REQUIRE(weights && packed, HNRF_E_ARG, "hnrf_canonical_bwd_pack: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_canonical_bwd_pack: mode %d not built", mode);
    for (int i = 0; i < 9; ++i) HNRF_REQUIRE(weights[i], HNRF_E_ARG, "hnrf_canonical_bwd_pack: null layer %d", i);
    hipStream_t st = (hipStream_t)stream;
    if (mode == HNRF_MLP_F16X3) return canonical16_bwd_pack(weights, packed, st);
    float* out = (float*)packed;
    if (hipMemsetAsync(out + CB_END, 0, PF * 256 * sizeof(float), st) != hipSuccess) {
        set_error("hnrf_canonical_bwd_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    int rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[8], 4, 256, 8, 1, PE_NONE, 0, 1, CB_HEAD}, out, st))) return rc;
    for (int l = 7; l >= 6; --l)
        if ((rc = launch_pack_bwd(PackBwd{weights[l], 256, 256, 8, 32, PE_NONE, 0, 0, CB_L7 + (7 - l) * CB_MID}, out, st)))
            return rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[5], 256, 319, 8, 32, PE_NONE, 63, 0, CB_L5H}, out, st))) return rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[5], 256, 319, 2, 32, PE_CANONICAL, 0, 0, CB_L5P}, out, st))) return rc;
    for (int l = 4; l >= 1; --l)
        if ((rc = launch_pack_bwd(PackBwd{weights[l], 256, 256, 8, 32, PE_NONE, 0, 0, CB_L4 + (4 - l) * CB_MID}, out, st)))
            return rc;
    return launch_pack_bwd(PackBwd{weights[0], 256, 63, 2, 32, PE_CANONICAL, 0, 0, CB_L0P}, out, st);
}

extern "C" int hnrf_nonrigid_bwd_pack(const float* const* weights, int mode, void* packed, void* stream) {
    HNRF_REQUIRE(weights && packed, HNRF_E_ARG, "hnrf_nonrigid_bwd_pack: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_nonrigid_bwd_pack: mode %d not built", mode);
    for (int i = 0; i < 7; ++i) HNRF_REQUIRE(weights[i], HNRF_E_ARG, "hnrf_nonrigid_bwd_pack: null layer %d", i);
    hipStream_t st = (hipStream_t)stream;
    if (mode == HNRF_MLP_F16X3) return nonrigid16_bwd_pack(weights, packed, st);
    float* out = (float*)packed;
    if (hipMemsetAsync(out + NB_END, 0, PF * 256 * sizeof(float), st) != hipSuccess) {
        set_error("hnrf_nonrigid_bwd_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    int rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[6], 3, 128, 4, 1, PE_NONE, 0, 1, NB_HEAD}, out, st))) return rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[5], 128, 128, 4, 16, PE_NONE, 0, 0, NB_L5}, out, st))) return rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[4], 128, 164, 4, 16, PE_NONE, 0, 0, NB_L4H}, out, st))) return rc;
    if ((rc = launch_pack_bwd(PackBwd{weights[4], 128, 164, 2, 16, PE_NONRIGID, 128, 0, NB_L4P}, out, st))) return rc;
    for (int l = 3; l >= 1; --l)
        if ((rc = launch_pack_bwd(PackBwd{weights[l], 128, 128, 4, 16, PE_NONE, 0, 0, NB_L3 + (3 - l) * NB_MID}, out, st)))
            return rc;
    return launch_pack_bwd(PackBwd{weights[0], 128, 105, 2, 16, PE_NONRIGID, 69, 0, NB_L0P}, out, st);
}

extern "C" int hnrf_canonical_bwd(const float* xyz, const float* d_raw, const uint32_t* relu_bits, const void* packed,
                                  int mode, const float* d_raw_amax, int64_t P, float* dZ, float* d_xyz,
                                  float* dz_amax, void* stream) {
    HNRF_REQUIRE(xyz && d_raw && relu_bits && packed && dZ && d_xyz, HNRF_E_ARG, "hnrf_canonical_bwd: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3 || mode == HNRF_MLP_F16X3_H, HNRF_E_UNSUPPORTED,
                 "hnrf_canonical_bwd: mode %d not built", mode);
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || d_raw_amax, HNRF_E_ARG,
                 "hnrf_canonical_bwd: HNRF_MLP_F16X3 needs d_raw_amax (device scalar >= max |d_raw|)");
    HNRF_REQUIRE(mode != HNRF_MLP_F16X3_H || dz_amax, HNRF_E_ARG, "hnrf_canonical_bwd: HNRF_MLP_F16X3_H needs dz_amax ([8] scales out)");
    HNRF_REQUIRE(P >= 0, HNRF_E_ARG, "hnrf_canonical_bwd: bad P");
    HNRF_REQUIRE((((uintptr_t)d_raw | (uintptr_t)relu_bits | (uintptr_t)dZ | (uintptr_t)packed) & 15) == 0, HNRF_E_ARG,
                 "hnrf_canonical_bwd: d_raw, relu_bits, dZ, packed must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    const int64_t blocks = (P + 127) / 128;
    HNRF_REQUIRE(blocks < 2147483647LL, HNRF_E_ARG, "hnrf_canonical_bwd: too many samples");
    if (mode == HNRF_MLP_F16X3_H)
        return canonical16_bwd(xyz, d_raw, relu_bits, packed, P, d_raw_amax, dZ, d_xyz, dz_amax, 1, (hipStream_t)stream);
    if (dz_amax && hipMemsetAsync(dz_amax, 0, 8 * HNRF_AMAX_SLOTS * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        set_error("hnrf_canonical_bwd: memset failed");
        return HNRF_E_LAUNCH;
    }
    if (mode == HNRF_MLP_F16X3)
        return canonical16_bwd(xyz, d_raw, relu_bits, packed, P, d_raw_amax, dZ, d_xyz, dz_amax, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(canonical_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xyz,
                       (const float4*)d_raw, relu_bits, (const float*)packed, P, dZ, d_xyz, dz_amax);
    return check_launch("hnrf_canonical_bwd");
}

extern "C" int hnrf_nonrigid_bwd(const float* x_skel, const float* hann_w, const float* d_xyz,
                                 const uint32_t* relu_bits, const void* packed, int mode,
                                 const float* d_xyz_amax, int64_t P, float* dZ, float* d_x_skel, float* dz_amax,
                                 void* stream) {
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3 || mode == HNRF_MLP_F16X3_H, HNRF_E_UNSUPPORTED, "hnrf_nonrigid_bwd: mode %d not built", mode);
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || d_xyz_amax, HNRF_E_ARG,
                 "hnrf_nonrigid_bwd: HNRF_MLP_F16X3 needs d_xyz_amax (device scalar >= max |d_xyz|)");
    HNRF_REQUIRE(mode != HNRF_MLP_F16X3_H || dz_amax, HNRF_E_ARG, "hnrf_nonrigid_bwd: HNRF_MLP_F16X3_H needs dz_amax ([6] scales out)");
    HNRF_REQUIRE(x_skel && hann_w && d_xyz && relu_bits && packed && dZ && d_x_skel, HNRF_E_ARG,
                 "hnrf_nonrigid_bwd: null pointer");
    HNRF_REQUIRE(P >= 0, HNRF_E_ARG, "hnrf_nonrigid_bwd: bad P");
    HNRF_REQUIRE((((uintptr_t)relu_bits | (uintptr_t)dZ | (uintptr_t)packed) & 15) == 0, HNRF_E_ARG,
                 "hnrf_nonrigid_bwd: relu_bits, dZ, packed must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    const int64_t blocks = (P + 127) / 128;
    HNRF_REQUIRE(blocks < 2147483647LL, HNRF_E_ARG, "hnrf_nonrigid_bwd: too many samples");
    if (mode == HNRF_MLP_F16X3_H)
        return nonrigid16_bwd(x_skel, hann_w, d_xyz, relu_bits, packed, P, d_xyz_amax, dZ, d_x_skel, dz_amax, 1,
                              (hipStream_t)stream);
    if (dz_amax && hipMemsetAsync(dz_amax, 0, 6 * HNRF_AMAX_SLOTS * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        set_error("hnrf_nonrigid_bwd: memset failed");
        return HNRF_E_LAUNCH;
    }
    if (mode == HNRF_MLP_F16X3)
        return nonrigid16_bwd(x_skel, hann_w, d_xyz, relu_bits, packed, P, d_xyz_amax, dZ, d_x_skel, dz_amax, 0,
                              (hipStream_t)stream);
    hipLaunchKernelGGL(nonrigid_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_skel, hann_w,
                       d_xyz, relu_bits, (const float*)packed, P, dZ, d_x_skel, dz_amax);
    return check_launch("hnrf_nonrigid_bwd");
}
