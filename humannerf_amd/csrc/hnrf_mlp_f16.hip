// K2 / K3 in HNRF_MLP_F16X3 mode: fp32-equivalent GEMMs on the f16 matrix cores.
//
// Every fp32 operand v is split into two f16 numbers,  v = hi + lo * 2^-11  with
//   hi = f16(v),  lo = f16((v - hi) * 2^11)        (both round-to-nearest; the 2^11
// keeps `lo` a NORMAL f16 with all 11 bits whenever hi is normal), and a product is
//   w*x ~= wh*xh + 2^-11 (wh*xl + wl*xh)             (dropped: wl*xl ~ 2^-22 |w x|)
// = three v_mfma_f32_32x32x16_f16 into two fp32 accumulators (unit scale / 2^-11
// scale).  Per-product relative error ~3*2^-22 = 7e-7 (fp32 fma chain: 2^-24 per
// product plus 2^-24 per accumulation step, which dominates for K = 256), at 3/16 of
// the cost of the f32-input MFMA.  |v| must stay below 65504 (f16 range); activations
// are clamped there.
//
// Structure: same transposed GEMM as the f32 kernels (activations of 32 samples stay
// in one wave's registers, sample on the lane; the D tile of 32x32x16 feeds the next
// layer's B operand after an in-register hi/lo split, K order permuted in the weight
// image).  What changes is the weight path: 3 MFMAs (96 cycles) consume 2 KiB of A
// operand per wave, 85 B/clk/CU for 4 waves -- too much for L1, so the workgroup's 4
// waves share each 32-row weight slab through LDS: slab n+1 is fetched with
// global_load_lds_dwordx4 (LDS-DMA, no VGPR staging) into the other half of a double
// buffer while slab n feeds the MFMAs through conflict-free ds_read_b128 (the image in
// HBM/L2 is already in LDS order: [k-step][hi|lo][lane][8 halves], + one 1 KiB block
// holding the 32 fp32 biases).  One barrier per slab.
#include "hnrf_common.h"

namespace hnrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

constexpr int SLAB_MAX = 41 * 1024;        // canonical skip layer: (4 + 16) k-steps x 2 KiB + bias block
constexpr float LO_SCALE = 2048.0f;        // 2^11
constexpr float LO_INV = 1.0f / 2048.0f;

enum { PE16_NONE = 0, PE16_CANONICAL = 1, PE16_NONRIGID = 2 };

// hidden feature contracted by element j of k-step ks on lane half h
__host__ __device__ inline int hid_feat16(int ks, int j, int h) {
    return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
}
// PE argument index a = 8 ks + j (see pe_col in hnrf_mlp.hip for the column maps)
__device__ inline int pe_col16(int kind, int a, int h) {
    if (kind == PE16_CANONICAL) {
        if (a < 30) return 3 + 6 * (a / 3) + 3 * h + (a % 3);
        if (a == 30) return h;
        return h == 0 ? 2 : -1;
    }
    if (a < 18) return 6 * (a / 3) + 3 * h + (a % 3);
    return -1;
}

struct PackLayer16 {
    const float* W;
    const float* b;
    int n_out, n_in;
    int NT, NKA, NKB;          // tiles; k-steps (16 features each) of the PE part / hidden part
    int pe_kind, a_col0, b_col0, fold_cols;
    int64_t off;               // byte offset of the layer's first slab
};

// one thread per f16 PAIR of the image; slab = (2 NK + 1) KiB
__global__ void pack_layer16_kernel(PackLayer16 d, const float* __restrict__ cond, char* __restrict__ packed) {
    const int NK = d.NKA + d.NKB;
    const int slab_bytes = (2 * NK + 1) * 1024;
    const int64_t n = (int64_t)d.NT * (slab_bytes / 4);           // 4-byte units
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i / (slab_bytes / 4));
    const int u = (int)(i % (slab_bytes / 4));                    // 4-byte unit inside the slab
    char* dst = packed + d.off + (int64_t)t * slab_bytes + (int64_t)u * 4;
    if (u >= 2 * NK * 256) {                                       // bias block: 32 floats then zeros
        const int k = u - 2 * NK * 256;
        float v = 0.f;
        const int row = 32 * t + k;
        if (k < 32 && row < d.n_out) {
            v = d.b[row];
            for (int c = 0; c < d.fold_cols; ++c) v += d.W[(int64_t)row * d.n_in + c] * cond[c];
        }
        *reinterpret_cast<float*>(dst) = v;
        return;
    }
    const int blk = u >> 8;                 // 1 KiB block = (k-step, part)
    const int ks = blk >> 1, part = blk & 1;
    const int lane = (u >> 2) & 63;
    const int j0 = (u & 3) * 2;             // two consecutive elements j0, j0+1
    const int h = lane >> 5;
    const int row = 32 * t + (lane & 31);
    _Float16 outv[2];
    for (int e = 0; e < 2; ++e) {
        const int j = j0 + e;
        int col = -1;
        if (ks < d.NKA) {
            const int c = pe_col16(d.pe_kind, 8 * ks + j, h);
            col = c < 0 ? -1 : d.a_col0 + c;
        } else {
            col = d.b_col0 + hid_feat16(ks - d.NKA, j, h);
        }
        float w = 0.f;
        if (row < d.n_out && col >= 0 && col < d.n_in) w = d.W[(int64_t)row * d.n_in + col];
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)((w - (float)hi) * LO_SCALE);
        outv[e] = part ? lo : hi;
    }
    *reinterpret_cast<h16x2*>(dst) = h16x2{outv[0], outv[1]};
}

// ---- layouts (bytes) ------------------------------------------------------
constexpr int64_t KB = 1024;
constexpr int CNL16_NB_L0 = 2 * 4 + 1, CNL16_NB_MID = 2 * 16 + 1, CNL16_NB_L5 = 2 * 20 + 1;
constexpr int64_t CNL16_L0 = 0;
constexpr int64_t CNL16_L1 = CNL16_L0 + 8 * CNL16_NB_L0 * KB;
constexpr int64_t CNL16_L5 = CNL16_L1 + 4 * 8 * CNL16_NB_MID * KB;
constexpr int64_t CNL16_L6 = CNL16_L5 + 8 * CNL16_NB_L5 * KB;
constexpr int64_t CNL16_OUT = CNL16_L6 + 2 * 8 * CNL16_NB_MID * KB;
constexpr int64_t CNL16_BYTES = CNL16_OUT + CNL16_NB_MID * KB;
constexpr int NR16_NB_L0 = 2 * 3 + 1, NR16_NB_MID = 2 * 8 + 1, NR16_NB_L4 = 2 * 11 + 1;
constexpr int64_t NR16_L0 = 0;
constexpr int64_t NR16_L1 = NR16_L0 + 4 * NR16_NB_L0 * KB;
constexpr int64_t NR16_L4 = NR16_L1 + 3 * 4 * NR16_NB_MID * KB;
constexpr int64_t NR16_L5 = NR16_L4 + 4 * NR16_NB_L4 * KB;
constexpr int64_t NR16_OUT = NR16_L5 + 4 * NR16_NB_MID * KB;
constexpr int64_t NR16_BYTES = NR16_OUT + NR16_NB_MID * KB;

// ---- device helpers ---------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// LDS-DMA of `nblocks` 1-KiB blocks of a slab, spread over the 4 waves.
__device__ __forceinline__ void slab_issue(const char* gsrc, char* lds_dst, int nblocks, int wave, int lane) {
    for (int b = wave; b < nblocks; b += 4) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(gsrc + b * 1024 + lane * 16), (lds_void*)(lds_dst + b * 1024), 16,
                                         0, 0);
    }
}

// hi/lo split of 8 fp32 values into two f16x8 B-operand fragments
__device__ __forceinline__ void split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 p = {v[2 * i], v[2 * i + 1]};
        const h16x2 hh = __builtin_convertvector(p, h16x2);
        const f32x2 back = __builtin_convertvector(hh, f32x2);
        const f32x2 rem = (p - back) * LO_SCALE;
        const h16x2 ll = __builtin_convertvector(rem, h16x2);
        hi[2 * i] = hh[0];
        hi[2 * i + 1] = hh[1];
        lo[2 * i] = ll[0];
        lo[2 * i + 1] = ll[1];
    }
}

// One layer.  K order = [a (NKA k-steps) | b (NKB k-steps)].  `g` = cursor to this
// layer's first slab (wave-uniform), already resident in LDS buffer 0 (layers always
// start on an even slab).  next_nb_last = blocks of the slab that follows this layer.
template <int NT, int NKA, int NKB, bool RELU, int NA, int NB, int NO>
__device__ __forceinline__ void layer16(const char*& g, char* smem, int next_nb_last, int wave, int lane,
                                        const h16x8 (&ah)[NA], const h16x8 (&al)[NA], const h16x8 (&bh)[NB],
                                        const h16x8 (&bl)[NB], h16x8 (&oh)[NO], h16x8 (&ol)[NO], float (&last)[16]) {
    static_assert(NA >= (NKA > 0 ? NKA : 1) && NB >= (NKB > 0 ? NKB : 1) && NO >= 2 * NT, "operand arrays too small");
    static_assert(NT == 1 || (NT % 2) == 0, "layers must keep the double-buffer parity");
    constexpr int NK = NKA + NKB;
    constexpr int NBLK = 2 * NK + 1;
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        char* cur = smem + (t & 1) * SLAB_MAX;
        char* nxt = smem + ((t + 1) & 1) * SLAB_MAX;
        const char* gnext = g + NBLK * 1024;
        slab_issue(gnext, nxt, (t == NT - 1) ? next_nb_last : NBLK, wave, lane);

        const float4* bp = reinterpret_cast<const float4*>(cur + 2 * NK * 1024);
        const float4 b0 = bp[0 + h], b1 = bp[2 + h], b2 = bp[4 + h], b3 = bp[6 + h];
        f32x16 acc1 = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const h16x8 wh = *reinterpret_cast<const h16x8*>(cur + (2 * ks) * 1024 + lane * 16);
            const h16x8 wl = *reinterpret_cast<const h16x8*>(cur + (2 * ks + 1) * 1024 + lane * 16);
            const bool in_a = ks < NKA;
            const int ia = in_a ? ks : 0, ib = in_a ? 0 : ks - NKA;
            const h16x8 xh = in_a ? ah[ia < NA ? ia : 0] : bh[ib < NB ? ib : 0];
            const h16x8 xl = in_a ? al[ia < NA ? ia : 0] : bl[ib < NB ? ib : 0];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc2, 0, 0, 0);
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = acc1[r] + acc2[r] * LO_INV;
            if (RELU) x = fminf(fmaxf(x, 0.f), 65504.f);
            v[r] = x;
        }
        if (NT == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) last[r] = v[r];
        } else {
            const float v0[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
            const float v1[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
            split8(v0, oh[2 * t], ol[2 * t]);
            split8(v1, oh[2 * t + 1], ol[2 * t + 1]);
        }
        __syncthreads();   // slab t+1 has landed (vmcnt(0)) and every wave is done reading slab t
        g = gnext;
    }
}

// K3, f16x3.  grid = ceil(P / 128) workgroups of 4 waves x 32 samples; dynamic LDS 2 x SLAB_MAX.
__global__ __launch_bounds__(256) void canonical_f16x3_kernel(const float* __restrict__ xyz,
                                                              const char* __restrict__ packed, int64_t P,
                                                              float4* __restrict__ raw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int64_t sample = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const int64_t sidx = sample < P ? sample : P - 1;

    const char* g = packed;
    slab_issue(g, smem, CNL16_NB_L0, wave, lane);      // slab 0 flies while the PE is computed

    const float x[3] = {xyz[sidx * 3 + 0], xyz[sidx * 3 + 1], xyz[sidx * 3 + 2]};
    h16x8 ph[4], pl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int a = 8 * ks + j;
            if (a < 30) {
                float sv, cv;
                sincosf(x[a % 3] * (float)(1 << (a / 3)), &sv, &cv);
                v[j] = h ? cv : sv;
            } else if (a == 30) {
                v[j] = h ? x[1] : x[0];
            } else {
                v[j] = h ? 0.f : x[2];
            }
        }
        split8(v, ph[ks], pl[ks]);
    }
    __syncthreads();

    h16x8 hAh[16], hAl[16], hBh[16], hBl[16];
    const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const h16x8 none[1] = {zero8};
    float last[16];
    layer16<8, 4, 0, true>(g, smem, CNL16_NB_MID, wave, lane, ph, pl, none, none, hAh, hAl, last);
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        layer16<8, 0, 16, true>(g, smem, l == 4 ? CNL16_NB_L5 : CNL16_NB_MID, wave, lane, none, none, hAh, hAl, hBh,
                                hBl, last);
#pragma unroll
        for (int i = 0; i < 16; ++i) { hAh[i] = hBh[i]; hAl[i] = hBl[i]; }
    }
    layer16<8, 4, 16, true>(g, smem, CNL16_NB_MID, wave, lane, ph, pl, hAh, hAl, hBh, hBl, last);
#pragma unroll
    for (int i = 0; i < 16; ++i) { hAh[i] = hBh[i]; hAl[i] = hBl[i]; }
#pragma unroll 1
    for (int l = 6; l <= 7; ++l) {
        layer16<8, 0, 16, true>(g, smem, CNL16_NB_MID, wave, lane, none, none, hAh, hAl, hBh, hBl, last);
#pragma unroll
        for (int i = 0; i < 16; ++i) { hAh[i] = hBh[i]; hAl[i] = hBl[i]; }
    }
    h16x8 dh[2], dl[2];
    layer16<1, 0, 16, false>(g, smem, 0, wave, lane, none, none, hAh, hAl, dh, dl, last);
    if (h == 0 && sample < P) raw[sample] = make_float4(last[0], last[1], last[2], last[3]);
}

// K2, f16x3 (width 128: 4 tiles, 8 k-steps).
__global__ __launch_bounds__(256) void nonrigid_f16x3_kernel(const float* __restrict__ x_skel,
                                                             const float* __restrict__ hann_w,
                                                             const char* __restrict__ packed, int64_t P,
                                                             float* __restrict__ xyz, float* __restrict__ offsets) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    const int64_t sample = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const int64_t sidx = sample < P ? sample : P - 1;

    const char* g = packed;
    slab_issue(g, smem, NR16_NB_L0, wave, lane);

    const float x[3] = {x_skel[sidx * 3 + 0], x_skel[sidx * 3 + 1], x_skel[sidx * 3 + 2]};
    h16x8 ph[3], pl[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int a = 8 * ks + j;
            if (a < 18) {
                float sv, cv;
                sincosf(x[a % 3] * (float)(1 << (a / 3)), &sv, &cv);
                v[j] = hann_w[a / 3] * (h ? cv : sv);
            } else {
                v[j] = 0.f;
            }
        }
        split8(v, ph[ks], pl[ks]);
    }
    __syncthreads();

    h16x8 hAh[8], hAl[8], hBh[8], hBl[8];
    const h16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const h16x8 none[1] = {zero8};
    float last[16];
    layer16<4, 3, 0, true>(g, smem, NR16_NB_MID, wave, lane, ph, pl, none, none, hAh, hAl, last);
#pragma unroll 1
    for (int l = 1; l <= 3; ++l) {
        layer16<4, 0, 8, true>(g, smem, l == 3 ? NR16_NB_L4 : NR16_NB_MID, wave, lane, none, none, hAh, hAl, hBh, hBl,
                               last);
#pragma unroll
        for (int i = 0; i < 8; ++i) { hAh[i] = hBh[i]; hAl[i] = hBl[i]; }
    }
    layer16<4, 3, 8, true>(g, smem, NR16_NB_MID, wave, lane, ph, pl, hAh, hAl, hBh, hBl, last);
    layer16<4, 0, 8, true>(g, smem, NR16_NB_MID, wave, lane, none, none, hBh, hBl, hAh, hAl, last);
    h16x8 dh[2], dl[2];
    layer16<1, 0, 8, false>(g, smem, 0, wave, lane, none, none, hAh, hAl, dh, dl, last);
    if (h == 0 && sample < P) {
        xyz[sample * 3 + 0] = x[0] + last[0];
        xyz[sample * 3 + 1] = x[1] + last[1];
        xyz[sample * 3 + 2] = x[2] + last[2];
        if (offsets) {
            offsets[sample * 3 + 0] = last[0];
            offsets[sample * 3 + 1] = last[1];
            offsets[sample * 3 + 2] = last[2];
        }
    }
}

static int launch_pack16(const PackLayer16& d, const float* cond, char* packed, hipStream_t st) {
    const int64_t n = (int64_t)d.NT * ((2 * (d.NKA + d.NKB) + 1) * 256);
    hipLaunchKernelGGL(pack_layer16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, cond, packed);
    return check_launch("hnrf pack (f16x3)");
}

size_t canonical16_bytes() { return (size_t)CNL16_BYTES; }
size_t nonrigid16_bytes() { return (size_t)NR16_BYTES; }

int canonical16_pack(const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    int rc;
    PackLayer16 d{w[0], b[0], 256, 63, 8, 4, 0, PE16_CANONICAL, 0, 0, 0, CNL16_L0};
    if ((rc = launch_pack16(d, nullptr, out, st))) return rc;
    for (int l = 1; l <= 4; ++l) {
        d = PackLayer16{w[l], b[l], 256, 256, 8, 0, 16, PE16_NONE, 0, 0, 0, CNL16_L1 + (l - 1) * 8 * CNL16_NB_MID * KB};
        if ((rc = launch_pack16(d, nullptr, out, st))) return rc;
    }
    d = PackLayer16{w[5], b[5], 256, 319, 8, 4, 16, PE16_CANONICAL, 0, 63, 0, CNL16_L5};
    if ((rc = launch_pack16(d, nullptr, out, st))) return rc;
    for (int l = 6; l <= 7; ++l) {
        d = PackLayer16{w[l], b[l], 256, 256, 8, 0, 16, PE16_NONE, 0, 0, 0, CNL16_L6 + (l - 6) * 8 * CNL16_NB_MID * KB};
        if ((rc = launch_pack16(d, nullptr, out, st))) return rc;
    }
    d = PackLayer16{w[8], b[8], 4, 256, 1, 0, 16, PE16_NONE, 0, 0, 0, CNL16_OUT};
    return launch_pack16(d, nullptr, out, st);
}

int nonrigid16_pack(const float* const* w, const float* const* b, const float* cond, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    int rc;
    PackLayer16 d{w[0], b[0], 128, 105, 4, 3, 0, PE16_NONRIGID, 69, 0, 69, NR16_L0};
    if ((rc = launch_pack16(d, cond, out, st))) return rc;
    for (int l = 1; l <= 3; ++l) {
        d = PackLayer16{w[l], b[l], 128, 128, 4, 0, 8, PE16_NONE, 0, 0, 0, NR16_L1 + (l - 1) * 4 * NR16_NB_MID * KB};
        if ((rc = launch_pack16(d, cond, out, st))) return rc;
    }
    d = PackLayer16{w[4], b[4], 128, 164, 4, 3, 8, PE16_NONRIGID, 128, 0, 0, NR16_L4};   // W4 cols: [h(128) | PE36]
    if ((rc = launch_pack16(d, cond, out, st))) return rc;
    d = PackLayer16{w[5], b[5], 128, 128, 4, 0, 8, PE16_NONE, 0, 0, 0, NR16_L5};
    if ((rc = launch_pack16(d, cond, out, st))) return rc;
    d = PackLayer16{w[6], b[6], 3, 128, 1, 0, 8, PE16_NONE, 0, 0, 0, NR16_OUT};
    return launch_pack16(d, cond, out, st);
}

int canonical16_fwd(const float* xyz, const void* packed, int64_t P, float* raw, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)canonical_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * SLAB_MAX);
        attr_set = true;
    }
    hipLaunchKernelGGL(canonical_f16x3_kernel, dim3((unsigned)((P + 127) / 128)), dim3(256), 2 * SLAB_MAX, st, xyz,
                       (const char*)packed, P, (float4*)raw);
    return check_launch("hnrf_canonical_fwd (f16x3)");
}

int nonrigid16_fwd(const float* x_skel, const float* hann_w, const void* packed, int64_t P, float* xyz,
                   float* offsets, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)nonrigid_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * SLAB_MAX);
        attr_set = true;
    }
    hipLaunchKernelGGL(nonrigid_f16x3_kernel, dim3((unsigned)((P + 127) / 128)), dim3(256), 2 * SLAB_MAX, st, x_skel,
                       hann_w, (const char*)packed, P, xyz, offsets);
    return check_launch("hnrf_nonrigid_fwd (f16x3)");
}

}  // namespace hnrf
