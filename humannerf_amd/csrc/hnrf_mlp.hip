// K2 / K3: the two per-sample MLPs, fused end to end (positional encoding ->
// all layers -> head) with the activations of 32 samples living in ONE wave's
// registers for the whole network.
//
// Formulation (transposed GEMM):   H_{l+1}^T [features x samples] = W_l . H_l^T
//   A operand = W_l (rows = output features), B operand = H_l^T (cols = samples).
// v_mfma_f32_32x32x2_f32 leaves D[row][col] with col = lane&31 (the sample) and
// rows (r&3) + 8*(r>>2) + 4*(lane>>5) in its 16 registers.  A 32x32x2 B operand is
// ONE f32 per lane: lane (c, h) supplies B[k = h][col = c].  Because the order of
// the K summation is free, the 16 D registers of tile t on lane half h can be fed
// back VERBATIM as the B operands of 16 K-steps of the next layer -- step
// j = 16 t + r then contracts feature hid_feat(j, 0) (lower lane half) and
// hid_feat(j, 1) (upper half) -- provided the weight image was packed in exactly
// that K order.  No LDS round trip, no cross-lane movement, no barrier: a wave
// never talks to another wave.  Activations: 128 VGPR/AGPR (256 features x 32
// samples / 64 lanes) in, 128 out.
//
// Weights: packed once per parameter update into the A-operand order
// [layer][tile][group of 4 K-steps][lane][4] so a wave fetches its next four A
// operands with one coalesced global_load_dwordx4 (1 KiB per wave), prefetched
// PF groups (>= 1 K cycles) ahead.  The image (2.0 MB canonical, 0.4 MB non-rigid)
// is shared by every wave on the chip and stays resident in each XCD's 4 MB L2;
// per-CU demand is 16 B/clk (4 waves x 256 B per 64-cycle MFMA), ~1/4 of what
// the L2 delivers.
//
// Roofline: MFMA-bound.  9312 MFMAs (7808 canonical + 1504 non-rigid) of 64
// cycles per 32 samples = 596 K SIMD-cycles; 592 384 algorithmic MAC per sample
// (99.4 % of the issued MACs are algorithmic; the rest is K/N padding).
#include "hnrf_common.h"
#include "hnrf_sincos.h"
#include "hnrf_mlp_layout.h"

namespace hnrf {

// ------------------------------------------------------------------ packing
struct PackLayer {
    const float* W;      // nn.Linear weight (n_out, n_in)
    const float* b;      // (n_out)
    int n_out, n_in;
    int NT;              // output tiles of 32 rows
    int NGA, NGB;        // groups (4 K-steps) of the PE part and of the hidden part
    int pe_kind;         // column map of the PE part
    int a_col0, b_col0;  // first column of the PE part / of the hidden part in W
    int a_first;         // 1: PE part precedes the hidden part in the K order
    int fold_cols;       // >0: bias += W[:, :fold_cols] . cond   (non-rigid layer 0)
    int64_t w_off;       // float offset of this layer in the packed image
    int64_t b_off;       // float offset of this layer's bias block
};

__global__ void pack_layer_kernel(PackLayer d, const float* __restrict__ cond, float* __restrict__ packed) {
    const int NG = d.NGA + d.NGB;
    const int64_t nw = (int64_t)d.NT * NG * 256;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) {
        const int e = (int)(i & 3);
        const int lane = (int)((i >> 2) & 63);
        const int64_t gg = i >> 8;
        const int g = (int)(gg % NG);
        const int t = (int)(gg / NG);
        const int h = lane >> 5;
        const int row = 32 * t + (lane & 31);
        int col = -1;
        const int ga = d.a_first ? g : g - d.NGB;   // group index inside the PE part
        const int gb = d.a_first ? g - d.NGA : g;   // group index inside the hidden part
        if (ga >= 0 && ga < d.NGA) {
            const int c = pe_col(d.pe_kind, 4 * ga + e, h);
            col = c < 0 ? -1 : d.a_col0 + c;
        } else if (gb >= 0 && gb < d.NGB) {
            col = d.b_col0 + hid_feat(4 * gb + e, h);
        }
        float v = 0.f;
        if (row < d.n_out && col >= 0 && col < d.n_in) v = d.W[(int64_t)row * d.n_in + col];
        packed[d.w_off + i] = v;
    }
    const int nb = d.NT * 32;
    if (i < nb) {
        const int r = (int)(i & 15);
        const int h = (int)((i >> 4) & 1);
        const int t = (int)(i >> 5);
        const int row = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
        float v = 0.f;
        if (row < d.n_out) {
            v = d.b[row];
            for (int c = 0; c < d.fold_cols; ++c) v += d.W[(int64_t)row * d.n_in + c] * cond[c];
        }
        packed[d.b_off + i] = v;
    }
}

// ------------------------------------------------------------------ layouts
// canonical: PE63(+1 pad) -> 256 x8 (skip [PE | h] into layer 5) -> 4
constexpr int64_t CNL_W_L0 = 0;                                   // 8 tiles x 8 groups
constexpr int64_t CNL_W_MID = 8 * 32 * 256;                       // 65536 floats per 256x256 layer
constexpr int64_t CNL_W_L1 = CNL_W_L0 + 8 * 8 * 256;              // layers 1..4
constexpr int64_t CNL_W_L5 = CNL_W_L1 + 4 * CNL_W_MID;            // 8 tiles x 40 groups
constexpr int64_t CNL_W_L6 = CNL_W_L5 + 8 * 40 * 256;             // layers 6..7
constexpr int64_t CNL_W_OUT = CNL_W_L6 + 2 * CNL_W_MID;           // 1 tile x 32 groups
constexpr int64_t CNL_W_END = CNL_W_OUT + 32 * 256;
constexpr int64_t CNL_B_OFF = CNL_W_END + PF * 256;               // prefetch over-run pad
constexpr int64_t CNL_FLOATS = CNL_B_OFF + 8 * 256 + 32;
// non-rigid: PE36(+4 pad) [cond folded into bias] -> 128 x6 (skip [h | PE] into layer 4) -> 3
constexpr int64_t NR_W_L0 = 0;                                    // 4 tiles x 5 groups
constexpr int64_t NR_W_MID = 4 * 16 * 256;                        // 16384
constexpr int64_t NR_W_L1 = NR_W_L0 + 4 * 5 * 256;                // layers 1..3
constexpr int64_t NR_W_L4 = NR_W_L1 + 3 * NR_W_MID;               // 4 tiles x 21 groups
constexpr int64_t NR_W_L5 = NR_W_L4 + 4 * 21 * 256;
constexpr int64_t NR_W_OUT = NR_W_L5 + NR_W_MID;                  // 1 tile x 16 groups
constexpr int64_t NR_W_END = NR_W_OUT + 16 * 256;
constexpr int64_t NR_B_OFF = NR_W_END + PF * 256;
constexpr int64_t NR_FLOATS = NR_B_OFF + 6 * 128 + 32;

// ------------------------------------------------------------------ compute
// One layer for this wave's 32 samples.  K order = [a (NGA groups) | b (NGB groups)]
// when A_FIRST else [b | a].  `wptr` is this lane's cursor into the packed image
// (float4 units, already offset by lane); `ring` holds the next PF groups.
// `save` (training): this lane's row of the layer's [P, 32 NT] activation matrix, or nullptr;
// lane half h writes features 32 t + 8 q + 4 h + (0..3) as one float4 per (t, q).
// `bits` (training): this lane's NT/2 words of the layer's ReLU sign mask, bit 16 (t & 1) + r of word t >> 1
// = [register r of tile t is positive] -- what the backward chain needs instead of re-reading the activations.
template <int PH, int NT, int NGA, int NGB, bool A_FIRST, bool RELU, int NA, int NB, int NO>
__device__ __forceinline__ void mlp_layer(const float4* __restrict__& wptr, float4 (&ring)[PF],
                                          const float* __restrict__ bias, const float (&a)[NA],
                                          const float (&b)[NB], float (&out)[NO], float* save = nullptr,
                                          int h = 0, uint32_t* bits = nullptr) {
    static_assert(NA >= NGA * 4 && NB >= NGB * 4 && NO >= NT * 16, "operand arrays too small");
    constexpr int NG = NGA + NGB;
    uint32_t bw[(NT + 1) / 2];
#pragma unroll
    for (int i = 0; i < (NT + 1) / 2; ++i) bw[i] = 0u;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4* bp = reinterpret_cast<const float4*>(bias + t * 32);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int slot = (PH + t * NG + g) % PF;   // PH: groups consumed before this layer, mod PF
            const float4 w = ring[slot];
            ring[slot] = wptr[PF * 64];
            wptr += 64;
            const bool in_a = A_FIRST ? (g < NGA) : (g >= NGB);
            // operand index of this group's first K-step (clamped in the untaken arm)
            const int ia = in_a ? 4 * (A_FIRST ? g : g - NGB) : 0;
            const int ib = in_a ? 0 : 4 * (A_FIRST ? g - NGA : g);
            const float o0 = in_a ? a[ia + 0 < NA ? ia + 0 : 0] : b[ib + 0 < NB ? ib + 0 : 0];
            const float o1 = in_a ? a[ia + 1 < NA ? ia + 1 : 0] : b[ib + 1 < NB ? ib + 1 : 0];
            const float o2 = in_a ? a[ia + 2 < NA ? ia + 2 : 0] : b[ib + 2 < NB ? ib + 2 : 0];
            const float o3 = in_a ? a[ia + 3 < NA ? ia + 3 : 0] : b[ib + 3 < NB ? ib + 3 : 0];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, o0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, o1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, o2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, o3, acc, 0, 0, 0);
            // pin the weight load PF groups ahead of its use (at the 256-VGPR limit hipcc otherwise sinks some of
            // them to the consuming MFMA: load; s_waitcnt vmcnt(0); mfma)
            __builtin_amdgcn_sched_barrier(0);
        }
        const float bs[16] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                              b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + bs[r];
            out[t * 16 + r] = RELU ? fmaxf(v, 0.f) : v;
        }
        if (save != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(save + 32 * t + 8 * q + 4 * h) =
                    make_float4(out[t * 16 + 4 * q], out[t * 16 + 4 * q + 1], out[t * 16 + 4 * q + 2],
                                out[t * 16 + 4 * q + 3]);
            uint32_t m = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) m |= (out[t * 16 + r] > 0.f ? 1u : 0u) << r;
            bw[t >> 1] |= m << (16 * (t & 1));
        }
    }
    if (save != nullptr && bits != nullptr) {
#pragma unroll
        for (int i = 0; i < (NT + 1) / 2; ++i) bits[i] = bw[i];
    }
}

__device__ __forceinline__ void ring_fill(const float4* __restrict__ wptr, float4 (&ring)[PF]) {
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wptr[i * 64];
}

// K3.  grid = ceil(P / 128) workgroups of 4 independent waves x 32 samples.
// SAVE (training forward): also writes the positional encoding pe_out [P,63] (reference column
// order) and the 8 post-ReLU activation matrices acts [8][P][256] that the backward GEMMs read.
template <bool SAVE>
__global__ __launch_bounds__(256) void canonical_f32_kernel(const float* __restrict__ xyz,
                                                            const float* __restrict__ packed, int64_t P,
                                                            float4* __restrict__ raw, float* __restrict__ pe_out,
                                                            float* __restrict__ acts, uint32_t* __restrict__ relu_bits,
                                                            const int* __restrict__ idx,
                                                            const int* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    if (idx != nullptr) {                     // sparse launch (hnrf_compact_samples): waves are independent
        P = *count;
        if (((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 >= P) return;
    }
    const int64_t slot = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (lane & 31);
    const int64_t sclamp = slot < P ? slot : P - 1;
    const int64_t sidx = idx ? (int64_t)idx[sclamp] : sclamp;
    const int64_t sample = slot < P ? sidx : P;
    const float x[3] = {xyz[sidx * 3 + 0], xyz[sidx * 3 + 1], xyz[sidx * 3 + 2]};

    // positional encoding in B-operand order (see pe_col): 32 K-steps
    float pe[32];
    {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int j = 0; j < 30; ++j) {
            float sv, cv;
            ph[j % 3].next(sv, cv);
            pe[j] = h ? cv : sv;
        }
    }
    pe[30] = h ? x[1] : x[0];
    pe[31] = h ? 0.f : x[2];
    float* save = nullptr;                       // this lane's activation row (advances one layer per step)
    uint32_t* bits = nullptr;                    // this lane's 4 words of the sign mask [8][P][2][4]
    if (SAVE && slot < P) {
        save = acts + sample * 256;
        bits = relu_bits + sample * 8 + h * 4;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int col = pe_col(PE_CANONICAL, j, h);
            if (col >= 0) pe_out[sample * 63 + col] = pe[j];
        }
    }
    const int64_t act_stride = P * 256, bit_stride = P * 8;

    const float4* wptr = reinterpret_cast<const float4*>(packed) + lane;
    const float* bias = packed + CNL_B_OFF + h * 16;
    float4 ring[PF];
    ring_fill(wptr, ring);

    float hA[128], hB[128];
    const float none[1] = {0.f};
    // ring phase (groups consumed so far, mod PF) entering layers 1..5 and 6..8; 256-group layers keep it
    constexpr int C_PH1 = (8 * 8) % PF, C_PH6 = (8 * 8 + 8 * 40) % PF;
    static_assert((8 * 32) % PF == 0, "looped layers must keep the ring phase");
    mlp_layer<0, 8, 8, 0, true, true>(wptr, ring, bias, pe, none, hA, save, h, bits);
    bias += 256;
    if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        mlp_layer<C_PH1, 8, 0, 32, true, true>(wptr, ring, bias, none, hA, hB, save, h, bits);
        bias += 256;
        if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = hB[i];
    }
    mlp_layer<C_PH1, 8, 8, 32, true, true>(wptr, ring, bias, pe, hA, hB, save, h, bits);   // skip: [PE | h]
    bias += 256;
    if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll
    for (int i = 0; i < 128; ++i) hA[i] = hB[i];
#pragma unroll 1
    for (int l = 6; l <= 7; ++l) {
        mlp_layer<C_PH6, 8, 0, 32, true, true>(wptr, ring, bias, none, hA, hB, save, h, bits);
        bias += 256;
        if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll
        for (int i = 0; i < 128; ++i) hA[i] = hB[i];
    }
    float o[16];
    mlp_layer<C_PH6, 1, 0, 32, true, false>(wptr, ring, bias, none, hA, o);
    // rows 0..3 of the head tile live in registers 0..3 of the lower lane half
    if (h == 0 && slot < P) raw[sample] = make_float4(o[0], o[1], o[2], o[3]);
}

// K2.  Same structure, width 128.  SAVE: pe_out [P,36], acts [6][P][128].
template <bool SAVE>
__global__ __launch_bounds__(256) void nonrigid_f32_kernel(const float* __restrict__ x_skel,
                                                           const float* __restrict__ hann_w,
                                                           const float* __restrict__ packed, int64_t P,
                                                           float* __restrict__ xyz, float* __restrict__ offsets,
                                                           float* __restrict__ pe_out, float* __restrict__ acts,
                                                           uint32_t* __restrict__ relu_bits,
                                                           const int* __restrict__ idx, const int* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    if (idx != nullptr) {
        P = *count;
        if (((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 >= P) return;
    }
    const int64_t slot = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (lane & 31);
    const int64_t sclamp = slot < P ? slot : P - 1;
    const int64_t sidx = idx ? (int64_t)idx[sclamp] : sclamp;
    const int64_t sample = slot < P ? sidx : P;
    const float x[3] = {x_skel[sidx * 3 + 0], x_skel[sidx * 3 + 1], x_skel[sidx * 3 + 2]};

    float pe[20];
    {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            float sv, cv;
            ph[j % 3].next(sv, cv);
            pe[j] = hann_w[j / 3] * (h ? cv : sv);
        }
    }
    pe[18] = 0.f;
    pe[19] = 0.f;
    float* save = nullptr;
    uint32_t* bits = nullptr;                    // sign mask [6][P][2][2]
    if (SAVE && slot < P) {
        save = acts + sample * 128;
        bits = relu_bits + sample * 4 + h * 2;
#pragma unroll
        for (int j = 0; j < 18; ++j) pe_out[sample * 36 + pe_col(PE_NONRIGID, j, h)] = pe[j];
    }
    const int64_t act_stride = P * 128, bit_stride = P * 4;

    const float4* wptr = reinterpret_cast<const float4*>(packed) + lane;
    const float* bias = packed + NR_B_OFF + h * 16;
    float4 ring[PF];
    ring_fill(wptr, ring);

    float hA[64], hB[64];
    const float none[1] = {0.f};
    constexpr int N_PH1 = (4 * 5) % PF, N_PH5 = (4 * 5 + 4 * 21) % PF;
    static_assert((4 * 16) % PF == 0, "looped layers must keep the ring phase");
    mlp_layer<0, 4, 5, 0, true, true>(wptr, ring, bias, pe, none, hA, save, h, bits);
    bias += 128;
    if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        mlp_layer<N_PH1, 4, 0, 16, true, true>(wptr, ring, bias, none, hA, hB, save, h, bits);
        bias += 128;
        if (save) { save += act_stride; bits += bit_stride; }
#pragma unroll
        for (int i = 0; i < 64; ++i) hA[i] = hB[i];
    }
    mlp_layer<N_PH1, 4, 5, 16, false, true>(wptr, ring, bias, pe, hA, hB, save, h, bits);   // skip: [h | PE]
    bias += 128;
    if (save) { save += act_stride; bits += bit_stride; }
    mlp_layer<N_PH5, 4, 0, 16, true, true>(wptr, ring, bias, none, hB, hA, save, h, bits);
    bias += 128;
    float o[16];
    mlp_layer<N_PH5, 1, 0, 16, true, false>(wptr, ring, bias, none, hA, o);
    if (h == 0 && slot < P) {
        xyz[sample * 3 + 0] = x[0] + o[0];
        xyz[sample * 3 + 1] = x[1] + o[1];
        xyz[sample * 3 + 2] = x[2] + o[2];
        if (offsets) {
            offsets[sample * 3 + 0] = o[0];
            offsets[sample * 3 + 1] = o[1];
            offsets[sample * 3 + 2] = o[2];
        }
    }
}

static int launch_pack(const PackLayer& d, const float* cond, float* packed, hipStream_t st) {
    const int64_t n = (int64_t)d.NT * (d.NGA + d.NGB) * 256;
    hipLaunchKernelGGL(pack_layer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, cond, packed);
    return check_launch("hnrf pack");
}

}  // namespace hnrf

using namespace hnrf;

extern "C" size_t hnrf_canonical_packed_bytes(int mode) {
    if (mode == HNRF_MLP_F16X3) return canonical16_bytes();
    return mode == HNRF_MLP_F32 ? (size_t)CNL_FLOATS * sizeof(float) : 0;
}
extern "C" size_t hnrf_nonrigid_packed_bytes(int mode) {
    if (mode == HNRF_MLP_F16X3) return nonrigid16_bytes();
    return mode == HNRF_MLP_F32 ? (size_t)NR_FLOATS * sizeof(float) : 0;
}

// byte offset of the image's status word (HNRF_STATUS_*), 0 = this arithmetic has none (fp32 MFMA cannot leave its range)
extern "C" size_t hnrf_canonical_status_offset(int mode) { return mode == HNRF_MLP_F16X3 ? canonical16_status_offset() : 0; }
extern "C" size_t hnrf_nonrigid_status_offset(int mode) { return mode == HNRF_MLP_F16X3 ? nonrigid16_status_offset() : 0; }

extern "C" int hnrf_canonical_pack(const float* const* weights, const float* const* biases, int mode, void* packed,
                                   void* stream) {
    HNRF_REQUIRE(weights && biases && packed, HNRF_E_ARG, "hnrf_canonical_pack: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_canonical_pack: mode %d not built", mode);
    for (int i = 0; i < 9; ++i)
        HNRF_REQUIRE(weights[i] && biases[i], HNRF_E_ARG, "hnrf_canonical_pack: null layer %d", i);
    hipStream_t st = (hipStream_t)stream;
    if (mode == HNRF_MLP_F16X3) return canonical16_pack(weights, biases, packed, st);
    float* out = (float*)packed;
    int rc;
    // zero the prefetch over-run pad
    if (hipMemsetAsync(out + CNL_W_END, 0, PF * 256 * sizeof(float), st) != hipSuccess) {
        set_error("hnrf_canonical_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    PackLayer d{};
    d.a_first = 1;
    d.fold_cols = 0;
    // layer 0: PE63 -> 256
    d = PackLayer{weights[0], biases[0], 256, 63, 8, 8, 0, PE_CANONICAL, 0, 0, 1, 0, CNL_W_L0, CNL_B_OFF};
    if ((rc = launch_pack(d, nullptr, out, st))) return rc;
    for (int l = 1; l <= 4; ++l) {
        d = PackLayer{weights[l], biases[l], 256, 256, 8, 0, 32, PE_NONE, 0, 0, 1, 0,
                      CNL_W_L1 + (l - 1) * CNL_W_MID, CNL_B_OFF + l * 256};
        if ((rc = launch_pack(d, nullptr, out, st))) return rc;
    }
    // layer 5: [PE63 | h] -> 256  (mlp_rgb_sigma.py:163-165: h = cat([pos_embed, h]))
    d = PackLayer{weights[5], biases[5], 256, 319, 8, 8, 32, PE_CANONICAL, 0, 63, 1, 0, CNL_W_L5, CNL_B_OFF + 5 * 256};
    if ((rc = launch_pack(d, nullptr, out, st))) return rc;
    for (int l = 6; l <= 7; ++l) {
        d = PackLayer{weights[l], biases[l], 256, 256, 8, 0, 32, PE_NONE, 0, 0, 1, 0,
                      CNL_W_L6 + (l - 6) * CNL_W_MID, CNL_B_OFF + l * 256};
        if ((rc = launch_pack(d, nullptr, out, st))) return rc;
    }
    d = PackLayer{weights[8], biases[8], 4, 256, 1, 0, 32, PE_NONE, 0, 0, 1, 0, CNL_W_OUT, CNL_B_OFF + 8 * 256};
    return launch_pack(d, nullptr, out, st);
}

extern "C" int hnrf_nonrigid_pack(const float* const* weights, const float* const* biases, const float* cond,
                                  int mode, void* packed, void* stream) {
    HNRF_REQUIRE(weights && biases && cond && packed, HNRF_E_ARG, "hnrf_nonrigid_pack: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_nonrigid_pack: mode %d not built", mode);
    for (int i = 0; i < 7; ++i)
        HNRF_REQUIRE(weights[i] && biases[i], HNRF_E_ARG, "hnrf_nonrigid_pack: null layer %d", i);
    hipStream_t st = (hipStream_t)stream;
    if (mode == HNRF_MLP_F16X3) return nonrigid16_pack(weights, biases, cond, packed, st);
    float* out = (float*)packed;
    int rc;
    if (hipMemsetAsync(out + NR_W_END, 0, PF * 256 * sizeof(float), st) != hipSuccess) {
        set_error("hnrf_nonrigid_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    // layer 0: [cond69 | PE36] -> 128; the cond columns are folded into the bias
    PackLayer d{weights[0], biases[0], 128, 105, 4, 5, 0, PE_NONRIGID, 69, 0, 1, 69, NR_W_L0, NR_B_OFF};
    if ((rc = launch_pack(d, cond, out, st))) return rc;
    for (int l = 1; l <= 3; ++l) {
        d = PackLayer{weights[l], biases[l], 128, 128, 4, 0, 16, PE_NONE, 0, 0, 1, 0,
                      NR_W_L1 + (l - 1) * NR_W_MID, NR_B_OFF + l * 128};
        if ((rc = launch_pack(d, cond, out, st))) return rc;
    }
    // layer 4: [h | PE36] -> 128  (mlp_offset.py:81-82: h = cat([h, pos_embed]))
    d = PackLayer{weights[4], biases[4], 128, 164, 4, 5, 16, PE_NONRIGID, 128, 0, 0, 0, NR_W_L4, NR_B_OFF + 4 * 128};
    if ((rc = launch_pack(d, cond, out, st))) return rc;
    d = PackLayer{weights[5], biases[5], 128, 128, 4, 0, 16, PE_NONE, 0, 0, 1, 0, NR_W_L5, NR_B_OFF + 5 * 128};
    if ((rc = launch_pack(d, cond, out, st))) return rc;
    d = PackLayer{weights[6], biases[6], 3, 128, 1, 0, 16, PE_NONE, 0, 0, 1, 0, NR_W_OUT, NR_B_OFF + 6 * 128};
    return launch_pack(d, cond, out, st);
}

extern "C" int hnrf_canonical_fwd_sparse(const float* xyz, const void* packed, int mode, int64_t P, const int* idx,
                                         const int* count, float* raw, void* stream);
extern "C" int hnrf_canonical_fwd(const float* xyz, const void* packed, int mode, int64_t P, float* raw,
                                  void* stream) {
    return hnrf_canonical_fwd_sparse(xyz, packed, mode, P, nullptr, nullptr, raw, stream);
}
extern "C" int hnrf_canonical_fwd_sparse(const float* xyz, const void* packed, int mode, int64_t P, const int* idx,
                                         const int* count, float* raw, void* stream) {
    const bool guard = !(mode & HNRF_MLP_NO_RANGE_GUARD);         // (HNRF_MLP_GUARD_ONE_CHUNK concerns hnrf_render_frame_fwd only)
    mode &= HNRF_MLP_ARITH_MASK;
    HNRF_REQUIRE(xyz && packed && raw, HNRF_E_ARG, "hnrf_canonical_fwd: null pointer");
    HNRF_REQUIRE((idx == nullptr) == (count == nullptr), HNRF_E_ARG, "hnrf_canonical_fwd: idx and count go together");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_canonical_fwd: mode %d not built", mode);
    HNRF_REQUIRE(P >= 0 && (P + 127) / 128 < 2147483647LL, HNRF_E_ARG, "hnrf_canonical_fwd: bad P=%lld", (long long)P);
    HNRF_REQUIRE((((uintptr_t)packed | (uintptr_t)raw) & 15) == 0, HNRF_E_ARG,
                 "hnrf_canonical_fwd: packed/raw must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    if (mode == HNRF_MLP_F16X3) return canonical16_fwd(xyz, packed, P, raw, idx, count, guard, (hipStream_t)stream);
    hipLaunchKernelGGL(canonical_f32_kernel<false>, dim3((unsigned)((P + 127) / 128)), dim3(256), 0,
                       (hipStream_t)stream, xyz, (const float*)packed, P, (float4*)raw, nullptr, nullptr, nullptr, idx, count);
    return check_launch("hnrf_canonical_fwd");
}

extern "C" int hnrf_canonical_fwd_train(const float* xyz, const void* packed, int mode, int64_t P, float* raw,
                                        float* pe_out, float* acts, uint32_t* relu_bits, void* stream) {
    HNRF_REQUIRE(xyz && packed && raw && pe_out && acts && relu_bits, HNRF_E_ARG,
                 "hnrf_canonical_fwd_train: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3 || mode == HNRF_MLP_F16X3_H, HNRF_E_UNSUPPORTED,
                 "hnrf_canonical_fwd_train: mode %d not built", mode);
    HNRF_REQUIRE(P >= 0 && (P + 127) / 128 < 2147483647LL, HNRF_E_ARG, "hnrf_canonical_fwd_train: bad P");
    HNRF_REQUIRE((((uintptr_t)packed | (uintptr_t)raw | (uintptr_t)acts) & 15) == 0, HNRF_E_ARG,
                 "hnrf_canonical_fwd_train: packed/raw/acts must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    if (mode != HNRF_MLP_F32)
        return canonical16_fwd_train(xyz, packed, P, raw, pe_out, acts, relu_bits, mode == HNRF_MLP_F16X3_H, (hipStream_t)stream);
    hipLaunchKernelGGL(canonical_f32_kernel<true>, dim3((unsigned)((P + 127) / 128)), dim3(256), 0,
                       (hipStream_t)stream, xyz, (const float*)packed, P, (float4*)raw, pe_out, acts, relu_bits, nullptr, nullptr);
    return check_launch("hnrf_canonical_fwd_train");
}

extern "C" int hnrf_nonrigid_fwd_sparse(const float* x_skel, const float* hann_w, const void* packed, int mode,
                                        int64_t P, const int* idx, const int* count, float* xyz, float* offsets,
                                        void* stream);
extern "C" int hnrf_nonrigid_fwd(const float* x_skel, const float* hann_w, const void* packed, int mode, int64_t P,
                                 float* xyz, float* offsets, void* stream) {
    return hnrf_nonrigid_fwd_sparse(x_skel, hann_w, packed, mode, P, nullptr, nullptr, xyz, offsets, stream);
}
extern "C" int hnrf_nonrigid_fwd_sparse(const float* x_skel, const float* hann_w, const void* packed, int mode,
                                        int64_t P, const int* idx, const int* count, float* xyz, float* offsets,
                                        void* stream) {
    const bool guard = !(mode & HNRF_MLP_NO_RANGE_GUARD);
    mode &= HNRF_MLP_ARITH_MASK;
    HNRF_REQUIRE(x_skel && hann_w && packed && xyz, HNRF_E_ARG, "hnrf_nonrigid_fwd: null pointer");
    HNRF_REQUIRE((idx == nullptr) == (count == nullptr), HNRF_E_ARG, "hnrf_nonrigid_fwd: idx and count go together");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3, HNRF_E_UNSUPPORTED,
                 "hnrf_nonrigid_fwd: mode %d not built", mode);
    HNRF_REQUIRE(P >= 0 && (P + 127) / 128 < 2147483647LL, HNRF_E_ARG, "hnrf_nonrigid_fwd: bad P=%lld", (long long)P);
    HNRF_REQUIRE(((uintptr_t)packed & 15) == 0, HNRF_E_ARG, "hnrf_nonrigid_fwd: packed must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    if (mode == HNRF_MLP_F16X3)
        return nonrigid16_fwd(x_skel, hann_w, packed, P, xyz, offsets, idx, count, guard, (hipStream_t)stream);
    hipLaunchKernelGGL(nonrigid_f32_kernel<false>, dim3((unsigned)((P + 127) / 128)), dim3(256), 0,
                       (hipStream_t)stream, x_skel, hann_w, (const float*)packed, P, xyz, offsets, nullptr, nullptr, nullptr, idx, count);
    return check_launch("hnrf_nonrigid_fwd");
}

extern "C" int hnrf_nonrigid_fwd_train(const float* x_skel, const float* hann_w, const void* packed, int mode,
                                       int64_t P, float* xyz, float* offsets, float* pe_out, float* acts,
                                       uint32_t* relu_bits, void* stream) {
    HNRF_REQUIRE(x_skel && hann_w && packed && xyz && pe_out && acts && relu_bits, HNRF_E_ARG,
                 "hnrf_nonrigid_fwd_train: null pointer");
    HNRF_REQUIRE(mode == HNRF_MLP_F32 || mode == HNRF_MLP_F16X3 || mode == HNRF_MLP_F16X3_H, HNRF_E_UNSUPPORTED,
                 "hnrf_nonrigid_fwd_train: mode %d not built", mode);
    HNRF_REQUIRE(P >= 0 && (P + 127) / 128 < 2147483647LL, HNRF_E_ARG, "hnrf_nonrigid_fwd_train: bad P");
    HNRF_REQUIRE((((uintptr_t)packed | (uintptr_t)acts) & 15) == 0, HNRF_E_ARG,
                 "hnrf_nonrigid_fwd_train: packed/acts must be 16-byte aligned");
    if (P == 0) return HNRF_OK;
    if (mode != HNRF_MLP_F32)
        return nonrigid16_fwd_train(x_skel, hann_w, packed, P, xyz, offsets, pe_out, acts, relu_bits,
                                    mode == HNRF_MLP_F16X3_H, (hipStream_t)stream);
    hipLaunchKernelGGL(nonrigid_f32_kernel<true>, dim3((unsigned)((P + 127) / 128)), dim3(256), 0,
                       (hipStream_t)stream, x_skel, hann_w, (const float*)packed, P, xyz, offsets, pe_out, acts, relu_bits, nullptr, nullptr);
    return check_launch("hnrf_nonrigid_fwd_train");
}
