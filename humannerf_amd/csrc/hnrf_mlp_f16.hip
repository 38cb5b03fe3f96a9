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
// waves share each 32-row weight slab through LDS.  Slabs travel L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4, no VGPR staging) into a 3-deep ring: the DMA for slab n+2
// is issued at the top of tile n and retired with a COUNTED s_waitcnt vmcnt(N) + raw
// s_barrier at the end of tile n+1's predecessor, so ~2 tiles (>3000 cycles) of DMA
// latency are covered.  The DMA is issued from inline asm: for the builtin hipcc emits
// s_waitcnt vmcnt(0) before the next ds_read, which serialises DMA and MFMAs (measured:
// MFMA busy 43 %, 38 % of wave time in waits).  Slab n feeds the MFMAs through
// conflict-free ds_read_b128 (the image in HBM/L2 is already in LDS order:
// [k-step][hi|lo][lane][8 halves]).  All biases sit in a small LDS table loaded once.
#include "hnrf_common.h"
#include "hnrf_sincos.h"

namespace hnrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

constexpr int RING = 3;                    // slabs resident in LDS
constexpr float LO_SCALE = 2048.0f;        // 2^11
constexpr float LO_INV = 1.0f / 2048.0f;
// Forward images (inference and activation-saving kernels): the weights' low parts are stored UN-lifted, lo = f16(w - hi),
// like the activations' (epi_pair_u), so that all three products of a k-step -- wh.xh, wh.xl, wl.xh -- go into ONE fp32
// accumulator: no second accumulator (32 VGPRs), no `acc1 + acc2 / 2^11` in the epilogue (2 of its 8 VALU per value
// pair), and the third MFMA multiplies tiny operands (the matrix pipe's energy depends on its operands' bits, and these
// kernels run at the package power limit).  The price: w - hi ~ 2^-12 |w| sits in f16's subnormal range for |w| < 0.25,
// so a weight is represented to max(2^-25, 2^-23 |w|) instead of 2^-23 |w| -- an ABSOLUTE floor of 3e-8 against layers
// whose largest weights are 0.1 .. 0.3; a dot product over 256 activations of order 0.5 sees 1.4e-7 of it, below the
// 6e-7 of fp32 accumulation itself (measured: tests/test_gpu_parity.py -s).  Layers with abnormally small weights are
// lifted as a whole by an exact power of two (layer_exponent), as before.  The dX chains keep the lifted form.
// -DHNRF_WLO_LIFTED builds the round-2 scheme (two accumulators) for A/B through HNRF_LIB_PATH.
#ifdef HNRF_WLO_LIFTED
constexpr bool WLO_UNLIFTED = false;
#else
constexpr bool WLO_UNLIFTED = true;
#endif

enum { PE16_NONE = 0, PE16_CANONICAL = 1, PE16_NONRIGID = 2 };

// what layer16 stores besides feeding the next layer (template parameter SAVE)
enum { SV_NONE = 0,
       SV_ACT = 1,      // training forward: fp32 activations + sign masks
       SV_DZ = 2,       // backward chain: relu' from the sign mask, fp32 dZ (un-scaled)
       SV_PE = 3,       // backward chain, positional-encoding stage: fp32 values to registers
       SV_ACT_H = 4,    // like SV_ACT with f16 activations
       SV_DZ_H = 5 };   // like SV_DZ with f16 dZ in the chain's power-of-two scaled domain
__host__ __device__ constexpr bool sv_fwd(int s) { return s == SV_ACT || s == SV_ACT_H; }
__host__ __device__ constexpr bool sv_bwd(int s) { return s == SV_DZ || s == SV_DZ_H; }
__host__ __device__ constexpr bool sv_has_bias(int s) { return s == SV_NONE || sv_fwd(s); }

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
    int64_t bias_off;          // byte offset of the layer's first 128-B bias record
    int head_scale;            // 1 (head layers): weights stored x 2^k, 2^-k left in float 8 of the bias record
};

// Power-of-two exponent that brings a layer with abnormally small or large weights back to the usual magnitude
// (max |w| in [1/8, 1/4)); 0 for anything between 2^-9 and 2^3.  The split v = hi + lo 2^-11 has an ABSOLUTE floor
// (f16 subnormals): a layer of +-1e-5 -- the reference initialises the last layer of the non-rigid MLP and of the pose
// decoder like that (mlp_offset.py:60-66) -- would keep ~17 of its 22 bits; scaled by 2^14 it keeps all of them, and
// the scale is exact and folded back where the values leave the kernel.
__host__ __device__ inline int layer_exponent(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
    int ex = 0;
    (void)frexpf(amax, &ex);
    return (ex < -8 || ex > 3) ? -2 - ex : 0;
}

// one thread per f16 PAIR of the weight image (slab = 2 NK KiB) + the layer's bias records; blockIdx.y = layer: the
// whole image of an MLP is ONE launch (16 + 18 launches of ~6 us per training step when every layer had its own)
struct PackSet16 {
    PackLayer16 d[9];
};
__global__ void pack_layer16_kernel(PackSet16 set, const float* __restrict__ cond, char* __restrict__ packed) {
    const PackLayer16& d = set.d[blockIdx.y];
    const int NK = d.NKA + d.NKB;
    const int slab_units = 2 * NK * 256;                          // 4-byte units per slab
    const int64_t n = (int64_t)d.NT * slab_units;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((int64_t)blockIdx.x * blockDim.x >= (n > d.NT * 32 ? n : (int64_t)d.NT * 32)) return;   // (grid sized for the largest layer)
    int kexp = 0;
    if (d.head_scale) {                                            // (<= 4 x 256 weights: every block finds the maximum itself)
        __shared__ float red[256];
        float m = 0.f;
        for (int e = threadIdx.x; e < d.n_out * d.n_in; e += blockDim.x) m = fmaxf(m, fabsf(d.W[e]));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
            __syncthreads();
        }
        kexp = layer_exponent(red[0]);
    }
    if (i < (int64_t)d.NT * 32) {                                  // bias: 32 fp32 per tile
        const int row = (int)i;
        float v = 0.f;
        if (row < d.n_out) {
            v = d.b[row];
            for (int c = 0; c < d.fold_cols; ++c) v += d.W[(int64_t)row * d.n_in + c] * cond[c];
        }
        if (d.head_scale && i == 8) v = ldexpf(1.0f, -kexp);
        *reinterpret_cast<float*>(packed + d.bias_off + i * 4) = v;
    }
    if (i >= n) return;
    const int t = (int)(i / slab_units);
    const int u = (int)(i % slab_units);
    char* dst = packed + d.off + ((int64_t)t * slab_units + u) * 4;
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
        if (row < d.n_out && col >= 0 && col < d.n_in) w = ldexpf(d.W[(int64_t)row * d.n_in + col], kexp);
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = WLO_UNLIFTED ? (_Float16)(w - (float)hi) : (_Float16)((w - (float)hi) * LO_SCALE);
        outv[e] = part ? lo : hi;
    }
    *reinterpret_cast<h16x2*>(dst) = h16x2{outv[0], outv[1]};
}

// ---- layouts (bytes) ------------------------------------------------------
constexpr int64_t KB = 1024;
// canonical: k-steps  L0 4 (PE64) | mid 16 | L5 4+16 | out 16 ; blocks per tile = 2 x k-steps
constexpr int CNL16_NB_L0 = 8, CNL16_NB_MID = 32, CNL16_NB_L5 = 40;
constexpr int64_t CNL16_L0 = 0;
constexpr int64_t CNL16_L1 = CNL16_L0 + 8 * CNL16_NB_L0 * KB;
constexpr int64_t CNL16_L5 = CNL16_L1 + 4 * 8 * CNL16_NB_MID * KB;
constexpr int64_t CNL16_L6 = CNL16_L5 + 8 * CNL16_NB_L5 * KB;
constexpr int64_t CNL16_OUT = CNL16_L6 + 2 * 8 * CNL16_NB_MID * KB;
constexpr int64_t CNL16_BIAS = CNL16_OUT + CNL16_NB_MID * KB;          // 8 layers x 8 tiles x 128 B, then the head's
constexpr int CNL16_BIAS_LDS = 8 * 1024;                               // part of the table kept in LDS
constexpr int64_t CNL16_BYTES = CNL16_BIAS + 9 * 1024;
constexpr int CNL16_SLAB = 40 * 1024;
// non-rigid: k-steps  L0 4 (PE36 padded to 64) | mid 8 | L4 4+8 | out 8
constexpr int NR16_NB_L0 = 8, NR16_NB_MID = 16, NR16_NB_L4 = 24;
constexpr int64_t NR16_L0 = 0;
constexpr int64_t NR16_L1 = NR16_L0 + 4 * NR16_NB_L0 * KB;
constexpr int64_t NR16_L4 = NR16_L1 + 3 * 4 * NR16_NB_MID * KB;
constexpr int64_t NR16_L5 = NR16_L4 + 4 * NR16_NB_L4 * KB;
constexpr int64_t NR16_OUT = NR16_L5 + 4 * NR16_NB_MID * KB;
constexpr int64_t NR16_BIAS = NR16_OUT + NR16_NB_MID * KB;            // 6 layers x 4 tiles x 128 B, then the head's
constexpr int NR16_BIAS_LDS = 3 * 1024;
constexpr int64_t NR16_BYTES = NR16_BIAS + 4 * 1024;
constexpr int NR16_SLAB = 32 * 1024;                                   // L0 travels as ONE slab of 4 tiles
// Status word of an image (uint32, in the unused tail of the head's bias record; zeroed by every pack).
constexpr int64_t CNL16_STATUS = CNL16_BIAS + 8 * 1024 + 512;
constexpr int64_t NR16_STATUS = NR16_BIAS + 3 * 1024 + 512;
constexpr int PE_STASH = 32 * 1024;        // per workgroup: 4 waves x 4 k-steps x (hi|lo) x 1 KiB

// ---- device helpers ---------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) h16x8 lds_h16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// LDS accesses by 32-bit byte address (keeps address arithmetic in one VGPR, ds_*_b128 + imm offset)
__device__ __forceinline__ h16x8 lds_ld8(unsigned addr) { return *reinterpret_cast<const lds_h16x8*>((size_t)addr); }
__device__ __forceinline__ f32x4 lds_ld4f(unsigned addr) { return *reinterpret_cast<const lds_f32x4*>((size_t)addr); }
__device__ __forceinline__ void lds_st8(unsigned addr, h16x8 v) { *reinterpret_cast<lds_h16x8*>((size_t)addr) = v; }

// Lane id recomputed where it is needed (2 VALU).  Volatile on purpose: addresses derived
// from a lane id hoisted to kernel entry get spilled, and the reload's compiler-inserted
// s_waitcnt vmcnt(0) would drain the hand-counted DMA queue on every tile.
__device__ __forceinline__ unsigned lane_now() {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at gbase + voff(l) land at LDS byte lds_addr + 16 l.
// Inline asm so that hipcc does not count it (its own bookkeeping would put
// s_waitcnt vmcnt(0) in front of the next ds_read); retired by wait_dma_keep().
__device__ __forceinline__ void dma_piece(const char* gbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(gbase), "s"(lds_addr)
        : "memory");
}

// The same inside the k-loops, where every instruction of the one wave per SIMD costs ~5 cycles of issue next to the
// MFMAs' 32 (profiles/tools/mfma_issue.hip: 8 + 5 n cycles per MFMA with n other instructions of ANY kind, scalar ones
// included): the form above is 8 instructions per piece (m0 saved / set / restored, s_nop, two-word source add, LDS
// address add), 1.3 per MFMA of the canonical kernel.  The instruction's immediate offset applies to BOTH addresses
// (LDS address = M0 + offset + 16 lane, measured: profiles/tools/dma_offset.hip), so four consecutive 1-KiB pieces
// share one source base and one M0 value (offsets 0 / 1024 / 2048 / 3072); M0 is handed to the compiler as an operand
// ("{m0}": it materialises the value and the hazard wait itself, and nothing else in these kernels uses M0).
// Wave w therefore moves the CONTIGUOUS pieces [w n, (w + 1) n) of a slab (n = pieces per wave), not w, w+4, ...
template <int R>
__device__ __forceinline__ void dma_piece_g(const char* gbase, unsigned voff, unsigned lds_addr) {
    // a scalar write of M0 needs one wait state before an LDS-DMA instruction reads it (gfx9 hazard), and hipcc, which
    // places the write, cannot see into the asm to insert it: hence the s_nop inside the statement.  In EVERY piece, not
    // only a run's first: where the pieces of a run sit in different basic blocks (runtime piece counts at layer
    // boundaries) the compiler writes M0 again in front of later pieces (the same value, so a stale read would be
    // harmless -- but that is an argument about today's code generation, not a guarantee)
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" : : "v"(voff), "s"(gbase), "{m0}"(lds_addr), "n"(R * 1024) : "memory");
}
// piece i (compile-time after unrolling) of this wave's run: base addresses of the wave's run in, group of four out
__device__ __forceinline__ void dma_run_piece(const char* run_src, unsigned voff, unsigned run_dst, int i) {
    const char* g = run_src + (i >> 2) * 4096;
    const unsigned d = run_dst + (i >> 2) * 4096;
    switch (i & 3) {
        case 0: dma_piece_g<0>(g, voff, d); break;
        case 1: dma_piece_g<1>(g, voff, d); break;
        case 2: dma_piece_g<2>(g, voff, d); break;
        default: dma_piece_g<3>(g, voff, d); break;
    }
}

// DMA of nblocks 1-KiB blocks: wave w moves blocks w, w+4, ...  (nblocks % 4 == 0
// wherever a counted wait follows, so that every wave has the same count in flight)
__device__ __forceinline__ void slab_issue(const char* gsrc, unsigned lds_addr, int nblocks, int wave, int nw = 4) {
    const unsigned voff = lane_now() * 16;
    for (int b = wave; b < nblocks; b += nw) dma_piece(gsrc + b * 1024, voff, lds_addr + b * 1024);
}

// wait until at most `keep` of this wave's DMA pieces are still in flight
__device__ __forceinline__ void wait_dma_keep(int keep) {
#define HNRF_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (keep) {
        // (even counts: four waves per workgroup; the odd ones: eight, 2 or 3 pieces per wave and slab.  Training variants:
        // the 4 activation stores per finished tile sit in the same in-order queue)
        HNRF_VMCNT_CASE(1) HNRF_VMCNT_CASE(2) HNRF_VMCNT_CASE(3) HNRF_VMCNT_CASE(4) HNRF_VMCNT_CASE(5) HNRF_VMCNT_CASE(6)
        HNRF_VMCNT_CASE(7) HNRF_VMCNT_CASE(8) HNRF_VMCNT_CASE(9) HNRF_VMCNT_CASE(10) HNRF_VMCNT_CASE(11) HNRF_VMCNT_CASE(12)
        HNRF_VMCNT_CASE(13) HNRF_VMCNT_CASE(14) HNRF_VMCNT_CASE(15) HNRF_VMCNT_CASE(16) HNRF_VMCNT_CASE(18) HNRF_VMCNT_CASE(20)
        HNRF_VMCNT_CASE(22) HNRF_VMCNT_CASE(24) HNRF_VMCNT_CASE(26) HNRF_VMCNT_CASE(28) HNRF_VMCNT_CASE(30)
        HNRF_VMCNT_CASE(32) HNRF_VMCNT_CASE(34) HNRF_VMCNT_CASE(36) HNRF_VMCNT_CASE(38) HNRF_VMCNT_CASE(40)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef HNRF_VMCNT_CASE
}

__device__ __forceinline__ void tile_sync(int keep) {
    wait_dma_keep(keep);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// hi/lo split of 8 fp32 values into two f16x8 B-operand fragments
__device__ __forceinline__ void split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 p = {v[2 * i], v[2 * i + 1]};
        const h16x2 hh = __builtin_convertvector(p, h16x2);
        const f32x2 rem = {fmaf((float)hh[0], -LO_SCALE, p[0] * LO_SCALE), fmaf((float)hh[1], -LO_SCALE, p[1] * LO_SCALE)};
        const h16x2 ll = __builtin_convertvector(rem, h16x2);
        hi[2 * i] = hh[0];
        hi[2 * i + 1] = hh[1];
        lo[2 * i] = ll[0];
        lo[2 * i + 1] = ll[1];
    }
}

// Per-workgroup weight pipeline state: ALL wave-uniform (SGPRs); lane-dependent
// addresses are re-derived per tile from lane_now().
// LDS map: [bias table | PE stash (4 waves x 8 KiB) | RING x slab]
struct Pipe {
    const char* gi;      // next slab to ISSUE (two slabs ahead of the one being consumed)
    unsigned lds_base;   // LDS byte address of the dynamic segment
    unsigned ring_off, slab_bytes, pe_off;
    unsigned ph;         // ring slot of the slab being consumed
    unsigned bias_off;   // byte offset of the current tile's bias record
    int wave;
    unsigned long long sat;   // lanes whose activations came near the f16 clamp (sat_check_frag)
    bool guard;               // compile-time constant of the kernel instance (template parameter GUARD)
#ifdef HNRF_STAMP
    unsigned long long t_last, sum_k, sum_b;   // diagnostic build only: cycles in k-loops / between them
#endif
};

// Epilogue of one PAIR of accumulator registers (2i, 2i+1) of a finished tile: combine the
// two accumulators, ReLU + f16-range clamp, hi/lo split, and drop the two halves into the
// next layer's B-operand fragment (k-step 2t + (i>>2), elements 2(i&3), 2(i&3)+1).
template <bool RELU>
__device__ __forceinline__ void epi_pair(const f32x16& a1, const f32x16& a2, int i, h16x8& hi, h16x8& lo) {
    // scalar f32 ops on purpose: beside MFMAs the packed forms (v_pk_add/mul_f32) cost more issue time
    float x0 = fmaf(a2[2 * i], LO_INV, a1[2 * i]);
    float x1 = fmaf(a2[2 * i + 1], LO_INV, a1[2 * i + 1]);
    if (RELU) {
        x0 = __builtin_amdgcn_fmed3f(x0, 0.f, 65504.f);
        x1 = __builtin_amdgcn_fmed3f(x1, 0.f, 65504.f);
    }
    const h16x2 hh = __builtin_convertvector(f32x2{x0, x1}, h16x2);
    // (x - hi) * 2^11, exact; written so that hipcc folds the f16 -> f32 conversion into v_fma_mix_f32
    const float r0 = fmaf((float)hh[0], -LO_SCALE, x0 * LO_SCALE);
    const float r1 = fmaf((float)hh[1], -LO_SCALE, x1 * LO_SCALE);
    const h16x2 ll = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    const int e = 2 * (i & 3);
    hi[e] = hh[0];
    hi[e + 1] = hh[1];
    lo[e] = ll[0];
    lo[e + 1] = ll[1];
}

// Un-scaled low part (two-group K2 kernel): lo = f16(x - hi) without the 2^11 lift, one VALU less per value.  lo then
// drops into f16 subnormals for |x| < 2^-3 and x is represented to max(2^-25, 2^-22 |x|) instead of 2^-22 |x|: an
// ABSOLUTE floor of 3e-8 on activations of order 0.1 .. 1 -- what a dot product over them sees is the same.  The
// products wh . xl then carry no 2^11 and go into the FIRST accumulator; wl keeps its lift (weights are packed
// offline and 2^-12 |w| would sit far down in the subnormals).
// f16-range guard of the inference kernels.  The split v = hi + lo holds below the f16 range; the epilogue clamps
// post-ReLU activations at 65504, and a checkpoint whose hidden activations get there would render a wrong image without
// any error.  GUARDED kernel instances (template parameter GUARD, Pipe::guard) therefore look at every finished
// B-operand fragment (every fourth pair): its 8 non-negative halves are folded with three v_pk_max_u16 and compared --
// as integers, which order like the values and put +inf / NaN on top -- against SAT_HALF; the verdict lands in an SGPR
// pair (no VGPR lives on: these kernels sit at the 256 + 256 register limit, one more live VGPR sent the two-group
// kernel to scratch), and the kernel raises HNRF_STATUS_F16_RANGE in the image's status word at its end.
// Cost, A/B on one box: 5 VALU per 8 values = +2.5 % canonical-kernel time, -3.0 % rays/s on the headline frame -- which
// is why the callers can ask for unguarded instances (HNRF_MLP_NO_RANGE_GUARD / HNRF_MLP_GUARD_ONE_CHUNK, hnrf.h).
// Two things that did not work: (1) the hardware's own sticky overflow bit -- without the clamp an overflowing
// v_cvt_pk_f16_f32 gives +inf, and TRAPSTS.EXCP[3] read with s_getreg_b32 at the kernel's end would cost nothing: on
// gfx950 the bit stays 0 with traps disabled (measured; the poisoned values then die in the next layer's ReLU, whose
// v_max_f32 drops NaN); (2) a v_max3_f32 running maximum: 4 VALU per 8 values but one more live VGPR (see above).
// The inline asm needs its "scc" clobber: s_or_b64 writes SCC, and without it hipcc scheduled the block between an
// s_add_u32 / s_addc_u32 pair -- a hit then added a carry, i.e. 4 GiB, to a weight pointer (memory fault).
constexpr unsigned SAT_HALF = 0x7B53u;      // f16(60000)
__device__ __forceinline__ void sat_check_frag(const h16x8& hi, unsigned long long& sflag) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    const u16x8 v = __builtin_bit_cast(u16x8, hi);               // (v_pk_max_u16: the float form adds a canonicalising
    const u16x2 a = {v[0], v[1]}, b = {v[2], v[3]}, c = {v[4], v[5]}, d = {v[6], v[7]};      // v_pk_max per operand)
    const u16x2 m = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
    const unsigned u = __builtin_bit_cast(unsigned, m);
    asm volatile("v_cmp_le_u32 vcc, %2, %1\n\ts_or_b64 %0, %0, vcc\n\tv_cmp_le_u16 vcc, %3, %1\n\ts_or_b64 %0, %0, vcc"
                 : "+s"(sflag) : "v"(u), "s"(SAT_HALF << 16), "s"(SAT_HALF) : "vcc", "scc");
}
__device__ __forceinline__ void raise_f16_range(const char* packed, int64_t off, unsigned long long sflag) {
    // a plain store, not an atomic OR: the word has this one bit, every writer writes the same value
    if (sflag != 0ull && (threadIdx.x & 63) == 0)
        __hip_atomic_store(reinterpret_cast<unsigned*>(const_cast<char*>(packed) + off), (unsigned)HNRF_STATUS_F16_RANGE,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool RELU>
__device__ __forceinline__ void epi_pair_u(const f32x16& a1, const f32x16& a2, int i, h16x8& hi, h16x8& lo,
                                           unsigned long long& sflag, bool guard) {
    float x0 = WLO_UNLIFTED ? a1[2 * i] : fmaf(a2[2 * i], LO_INV, a1[2 * i]);
    float x1 = WLO_UNLIFTED ? a1[2 * i + 1] : fmaf(a2[2 * i + 1], LO_INV, a1[2 * i + 1]);
    if (RELU) {
        x0 = __builtin_amdgcn_fmed3f(x0, 0.f, 65504.f);
        x1 = __builtin_amdgcn_fmed3f(x1, 0.f, 65504.f);
    }
    const h16x2 hh = __builtin_convertvector(f32x2{x0, x1}, h16x2);
    // x - hi in ONE v_fma_mix_f32 (f16 operand converted inside): with a literal -1 hipcc rewrites the fma as
    // v_cvt_f32_f16 + v_sub_f32, so the factor is made opaque
    float m1 = -1.0f;
    asm("" : "+v"(m1));
    const float r0 = fmaf((float)hh[0], m1, x0);
    const float r1 = fmaf((float)hh[1], m1, x1);
    const h16x2 ll = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    const int e = 2 * (i & 3);
    hi[e] = hh[0];
    hi[e + 1] = hh[1];
    lo[e] = ll[0];
    lo[e + 1] = ll[1];
    if (RELU && guard && (i & 3) == 3) sat_check_frag(hi, sflag);
}
__device__ __forceinline__ void split8_u(const float (&v)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 p = {v[2 * i], v[2 * i + 1]};
        const h16x2 hh = __builtin_convertvector(p, h16x2);
        const h16x2 ll = __builtin_convertvector(f32x2{p[0] - (float)hh[0], p[1] - (float)hh[1]}, h16x2);
        hi[2 * i] = hh[0];
        hi[2 * i + 1] = hh[1];
        lo[2 * i] = ll[0];
        lo[2 * i + 1] = ll[1];
    }
}

// epi_pair that also hands back the two activations (training variant stores them)
template <bool RELU, bool ULO = false>
__device__ __forceinline__ h16x2 epi_pair_x(const f32x16& a1, const f32x16& a2, int i, h16x8& hi, h16x8& lo, float& x0o,
                                            float& x1o) {
    float x0 = (ULO && WLO_UNLIFTED) ? a1[2 * i] : fmaf(a2[2 * i], LO_INV, a1[2 * i]);
    float x1 = (ULO && WLO_UNLIFTED) ? a1[2 * i + 1] : fmaf(a2[2 * i + 1], LO_INV, a1[2 * i + 1]);
    if (RELU) {
        x0 = __builtin_amdgcn_fmed3f(x0, 0.f, 65504.f);
        x1 = __builtin_amdgcn_fmed3f(x1, 0.f, 65504.f);
    }
    const h16x2 hh = __builtin_convertvector(f32x2{x0, x1}, h16x2);
    float r0, r1;
    if (ULO) {                  // un-scaled low part, see epi_pair_u
        float m1 = -1.0f;
        asm("" : "+v"(m1));
        r0 = fmaf((float)hh[0], m1, x0);
        r1 = fmaf((float)hh[1], m1, x1);
    } else {                    // (x - hi) * 2^11, exact; written so that hipcc folds the f16 -> f32 conversion into v_fma_mix_f32
        r0 = fmaf((float)hh[0], -LO_SCALE, x0 * LO_SCALE);
        r1 = fmaf((float)hh[1], -LO_SCALE, x1 * LO_SCALE);
    }
    const h16x2 ll = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    const int e = 2 * (i & 3);
    hi[e] = hh[0];
    hi[e + 1] = hh[1];
    lo[e] = ll[0];
    lo[e + 1] = ll[1];
    x0o = x0;
    x1o = x1;
    return hh;
}

// Training variant (SAVE): where this lane's activations go.  row = the lane's row of the layer's [P, width]
// fp32 activation matrix (+ 4 h floats: the lane half's column offset), bits = its words of the sign mask.
// EVERY lane stores (lanes past P carry a copy of sample P-1 and rewrite its values): the number of VMEM
// operations per tile must be the same for all waves, the hand-counted vmcnt waits include them.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

struct SaveCtx {
    float* row;
    _Float16* rowh;            // SV_ACT_H / SV_DZ_H: the wave's block of the blocked f16 matrix (+ 128 h), see save_pair_bh
    int cx;                    // c ^ 4 h
    uint32_t* bits;
    float sv0, sv1;            // even pair of the current float4
    // backward chain (layer16 SAVE = 2 / 3): sign-mask words of the stage, 1 / scale of the gradient, running
    // maximum of the |dZ| stored by the stage, fp32 outputs of a positional-encoding stage
    uint32_t mask[4];
    float descale, amax;
    float fout[32];
};

// Backward epilogue of one accumulator pair: x = (acc1 + acc2 / 2^11) * relu'(mask bit), split for the next stage;
// hands back the two (still scaled) values.
__device__ __forceinline__ void epi_pair_m(const f32x16& a1, const f32x16& a2, int i, uint32_t mbits, h16x8& hi,
                                           h16x8& lo, float& x0o, float& x1o) {
    float x0 = fmaf(a2[2 * i], LO_INV, a1[2 * i]);
    float x1 = fmaf(a2[2 * i + 1], LO_INV, a1[2 * i + 1]);
    x0 = (mbits >> (2 * i)) & 1u ? __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f) : 0.f;
    x1 = (mbits >> (2 * i + 1)) & 1u ? __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f) : 0.f;
    const h16x2 hh = __builtin_convertvector(f32x2{x0, x1}, h16x2);
    // (x - hi) * 2^11, exact; written so that hipcc folds the f16 -> f32 conversion into v_fma_mix_f32
    const float r0 = fmaf((float)hh[0], -LO_SCALE, x0 * LO_SCALE);
    const float r1 = fmaf((float)hh[1], -LO_SCALE, x1 * LO_SCALE);
    const h16x2 ll = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    const int e = 2 * (i & 3);
    hi[e] = hh[0];
    hi[e + 1] = hh[1];
    lo[e] = ll[0];
    lo[e + 1] = ll[1];
    x0o = x0;
    x1o = x1;
}

// dZ store of the backward chain: un-scaled fp32, 4 values per store like save_pair; tracks max |dZ|
__device__ __forceinline__ void save_pair_b(SaveCtx& sc, int t, int i, float x0, float x1) {
    x0 *= sc.descale;
    x1 *= sc.descale;
    sc.amax = fmaxf(sc.amax, fmaxf(fabsf(x0), fabsf(x1)));
    if (i & 1) {
        *reinterpret_cast<f32x4*>(sc.row + 32 * t + 8 * (i >> 1)) = f32x4{sc.sv0, sc.sv1, x0, x1};
    } else {
        sc.sv0 = x0;
        sc.sv1 = x1;
    }
}

template <int NW>
__device__ __forceinline__ void save_pair(SaveCtx& sc, uint32_t (&bw)[NW], int t, int i, float x0, float x1) {
    bw[t >> 1] |= ((x0 > 0.f ? 1u : 0u) | (x1 > 0.f ? 2u : 0u)) << (16 * (t & 1) + 2 * i);
    if (i & 1) {
        *reinterpret_cast<f32x4*>(sc.row + 32 * t + 8 * (i >> 1)) = f32x4{sc.sv0, sc.sv1, x0, x1};
    } else {
        sc.sv0 = x0;
        sc.sv1 = x1;
    }
}

// f16 forms (SV_ACT_H / SV_DZ_H): 4 halves (8 bytes) per store, same number of stores per tile as the fp32 forms (the
// hand-counted vmcnt waits do not change).  What is stored is EXACTLY the hi part of the split the next layer consumes
// (hi = f16(x) is the f16 rounding of the value), so the store takes its two dwords straight out of the B-operand
// fragment: no conversion, no extra VALU.  dZ is stored as it stands in the chain (scaled by a power of two that the
// weight-gradient kernel divides out), activations un-scaled.
// BLOCKED layout (HNRF_DWH_*_BLOCKED in hnrf_mlp_dw_h): the matrix of one layer is a sequence of 32-sample blocks, a
// block a sequence of 32-feature tiles of 2 KiB laid out [k = 2 g + h: 8 groups of 4 features][32 sample slots][4 halves]
// with sample c in slot c ^ 4 k -- the order in which a wave holds it (lane (c, h), registers 4 g .. 4 g + 3), so that
// every store instruction writes 512 contiguous bytes (row-major [sample][feature] would scatter each instruction over
// 32 rows: 32 partial cache-line writes), and the slot swizzle makes the image conflict-free for the weight-gradient
// kernel's transposed LDS reads as it stands (it is copied to LDS linearly).
// SaveCtx::rowh = block base + 128 h (halves), SaveCtx::cx = c ^ 4 h.
// forward form: the two dwords come straight out of the finished B-operand fragment (i odd: pairs i - 1, i)
__device__ __forceinline__ void store_frag_h(_Float16* rowh, int cx, int t, int i, const h16x8& frag) {
    const int g = i >> 1, e = 2 * (i & 3);
    *reinterpret_cast<h16x4*>(rowh + 1024 * t + 256 * g + ((cx ^ (8 * g)) << 2)) = h16x4{frag[e - 2], frag[e - 1], frag[e], frag[e + 1]};
}
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_pair_h(_Float16* rowh, int cx, unsigned& svu, int t, int i, h16x2 hh) {     // hh: the hi parts of pair i
    const unsigned u = __builtin_bit_cast(unsigned, hh);
    if (i & 1) {
        const int g = i >> 1;
        *reinterpret_cast<u32x2s*>(rowh + 1024 * t + 256 * g + ((cx ^ (8 * g)) << 2)) = u32x2s{svu, u};
    } else {
        svu = u;
    }
}

// sign mask of the f16 forms, one PAIR of values per push: the pair's hi parts sit packed in one register (the B-operand
// fragment the next layer consumes), so [h != 0] for both is one v_pk_min_u16 against 1 and acc = 2 acc + bit for both one
// v_pk_mad_u16 -- 2 VALU per pair where v_cmp + v_addc per value cost 4 (round 3: the masks were 1 072 of the 1 494 VALU
// instructions the activation-saving forward had over the inference form).  The low half of a word collects the pairs'
// first values, the high half their second values; a word takes the 16 pairs of two tiles (even tile first), push
// n = 8 (tile & 1) + pair ends in bit 15 - n of its half.
// [f16(x) != 0] instead of [x > 0]: the two differ for 0 < x <= 2^-25 only (such a value rounds to a zero hi part), i.e.
// for pre-activations within 3e-8 of the kink -- two orders of magnitude rarer than the sign flips the forward's own
// last-bit rounding causes (6e-7 relative), and exact zeros (dead units, all-zero inputs) give 0 as torch does.
__device__ __forceinline__ void push_pair_bits(uint32_t& acc, h16x2 hh) {
    const unsigned u = __builtin_bit_cast(unsigned, hh);
    unsigned tmp;
    asm("v_pk_min_u16 %1, %2, %3\n\tv_pk_mad_u16 %0, %0, %4, %1" : "+v"(acc), "=&v"(tmp) : "v"(u), "s"(0x00010001u), "s"(0x00020002u));
}

// Backward epilogue of the f16 form: relu' from that mask (the whole word; n = 8 (tile & 1) + pair as pushed): a 1-bit
// signed field extract gives 0 / -1, one AND applies it (2 VALU per value).
__device__ __forceinline__ h16x2 epi_pair_mh(const f32x16& a1, const f32x16& a2, int i, uint32_t mword, int n, h16x8& hi, h16x8& lo) {
    float x0 = fmaf(a2[2 * i], LO_INV, a1[2 * i]);
    float x1 = fmaf(a2[2 * i + 1], LO_INV, a1[2 * i + 1]);
    x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
    x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
    x0 = __uint_as_float(__float_as_uint(x0) & (uint32_t)__builtin_amdgcn_sbfe((int)mword, 15 - n, 1));
    x1 = __uint_as_float(__float_as_uint(x1) & (uint32_t)__builtin_amdgcn_sbfe((int)mword, 31 - n, 1));
    const h16x2 hh = __builtin_convertvector(f32x2{x0, x1}, h16x2);
    const float r0 = fmaf((float)hh[0], -LO_SCALE, x0 * LO_SCALE);
    const float r1 = fmaf((float)hh[1], -LO_SCALE, x1 * LO_SCALE);
    const h16x2 ll = __builtin_convertvector(f32x2{r0, r1}, h16x2);
    const int e = 2 * (i & 3);
    hi[e] = hh[0];
    hi[e + 1] = hh[1];
    lo[e] = ll[0];
    lo[e + 1] = ll[1];
    return hh;
}

// One layer.  K order = [PE part (NKA k-steps, B operand from the LDS stash) | hidden part
// (NKB k-steps, B operand from registers)].  The layer's NT tiles travel TPS per slab.
// nb1 / nb2: blocks of the first / second slab that follow this layer in the image (0 = none).
//
// Schedule (pinned with sched_barrier, hipcc otherwise sinks every LDS read to its use and
// parks all epilogues at the end of the layer):
//   * LDS -> register operand queue PFK k-steps deep (LDS latency under load is above the
//     96 cycles of one k-step's MFMAs);
//   * the VALU epilogue of tile t-1 (8 pair-units) is spread over the k-steps of tile t,
//     so it runs in the shadow of the MFMAs; only the last tile's epilogue is exposed.
//   * PEND_IN / DEFER (inference): the epilogue of a layer's LAST tile has no next tile of its own layer to hide
//     behind; with DEFER it is left pending in (pend1, pend2) and the next layer (PEND_IN) runs it inside its first
//     tile, writing the last two fragments of its own input just before the k-steps that consume them.
template <int NT, int TPS, int NKA, int NKB, bool RELU, int SAVE = 0, bool PEND_IN = false, bool DEFER = false,
          bool ULO = false, int NW = 4, int NB, int NO>
__device__ __forceinline__ void layer16(Pipe& p, int nb1, int nb2, h16x8 (&bh)[NB], h16x8 (&bl)[NB],
                                        h16x8 (&oh)[NO], h16x8 (&ol)[NO], float (&last)[16], SaveCtx* sc = nullptr,
                                        f32x16* pend1 = nullptr, f32x16* pend2 = nullptr) {
    static_assert(!(PEND_IN || DEFER) || SAVE == SV_NONE, "deferred epilogues exist in the inference form only");
    static_assert(!PEND_IN || NKB >= 2, "a pending tile fills the last two hidden fragments");
    static_assert(NB >= (NKB > 0 ? NKB : 1) && NO >= 2 * NT && NT % TPS == 0, "bad layer shape");
    constexpr int NK = NKA + NKB;
    constexpr int NBLK = 2 * NK;             // blocks per tile
    constexpr int NS = NT / TPS;             // slabs of this layer
    constexpr int PFK = NK < 4 ? NK : 4;
    static_assert((NBLK * TPS) % NW == 0, "every wave must issue the same number of DMA pieces per slab");
    static_assert(!ULO || SAVE == SV_NONE || sv_fwd(SAVE), "un-scaled activation low parts exist in the forward kernels only");
    f32x16 pacc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 pacc2 = pacc1;                    // accumulators of the previous tile (epilogue pending)
    f32x16 nbias = pacc1;                    // bias of the NEXT tile: the table is resident in LDS, so it is read
                                             // during the current tile instead of on the critical path behind the barrier
    uint32_t bw[(NT + 1) / 2];               // SAVE: sign mask of this layer's outputs (this lane half)
    // f16 forms: locals, not SaveCtx fields (the struct is reached through a pointer: fields written here would
    // live in scratch memory)
    _Float16* const rowh = (SAVE == SV_ACT_H || SAVE == SV_DZ_H) ? sc->rowh : nullptr;
    const int cxh = (SAVE == SV_ACT_H || SAVE == SV_DZ_H) ? sc->cx : 0;
    unsigned svu = 0u;
#pragma unroll
    for (int i = 0; i < (NT + 1) / 2; ++i) bw[i] = 0u;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // slab n+2 -> ring slot (ph + 2) % RING (last read during step n-1: fenced by the previous barrier)
        const int nissue = (s + 2 < NS) ? NBLK * TPS : (s + 2 == NS ? nb1 : nb2);
        const unsigned slot2 = p.ph == 0 ? 2 : p.ph - 1;
        // its 1-KiB pieces are issued one per k-step INSIDE this slab's MFMA stream (an LDS-DMA issue
        // costs ~100-150 cycles of the wave's issue slot: 8-10 of them in front of the tile idle the
        // matrix pipe for ~1000 cycles per tile -- measured with cycle stamps)
        const int dcnt = nissue / NW;                      // pieces per wave: this wave moves the run [wave dcnt, (wave + 1) dcnt)
        const char* dsrc = p.gi + p.wave * dcnt * 1024;
        const unsigned ddst = p.lds_base + p.ring_off + slot2 * p.slab_bytes + p.wave * dcnt * 1024;
        p.gi += nissue * 1024;
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = s * TPS + tt;
            const unsigned lane = lane_now();
            unsigned cur = p.lds_base + p.ring_off + p.ph * p.slab_bytes + tt * (NBLK * 1024) + lane * 16;
            unsigned pe = p.lds_base + p.pe_off + p.wave * (PE_STASH / 4) + lane * 16;
            // opaque to constant propagation: where the ring slot is known at compile time hipcc otherwise builds
            // every operand address as (absolute constant + lane offset) with a v_add per ds_read instead of using
            // the instruction's immediate offset on one base register
            asm volatile("" : "+v"(cur), "+v"(pe));
            f32x16 acc1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc2 = acc1;
            if (NT > 1 && sv_has_bias(SAVE)) {             // (the backward stages have no bias)
                if (t == 0) {
                    const unsigned bp = p.lds_base + p.bias_off + tt * 128 + (lane >> 5) * 16;
                    const f32x4 b0 = lds_ld4f(bp), b1 = lds_ld4f(bp + 32), b2 = lds_ld4f(bp + 64), b3 = lds_ld4f(bp + 96);
                    acc1 = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                                  b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
                } else {
                    acc1 = nbias;                          // read during the previous tile (see below)
                }
            }
            h16x8 qwh[PFK], qwl[PFK], qxh[PFK], qxl[PFK];
#pragma unroll
            for (int i = 0; i < PFK; ++i) {
                qwh[i] = lds_ld8(cur + (2 * i) * 1024);
                qwl[i] = lds_ld8(cur + (2 * i + 1) * 1024);
                if (i < NKA) {
                    qxh[i] = lds_ld8(pe + (2 * i) * 1024);
                    qxl[i] = lds_ld8(pe + (2 * i + 1) * 1024);
                }
            }
#ifdef HNRF_STAMP
            { const unsigned long long t0 = __builtin_readcyclecounter(); p.sum_b += t0 - p.t_last; p.t_last = t0; }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int q = ks % PFK;
                const h16x8 wh = qwh[q], wl = qwl[q];
                h16x8 xh, xl;
                if (ks < NKA) {
                    xh = qxh[q];
                    xl = qxl[q];
                } else {
                    const int ib = ks - NKA < NB ? ks - NKA : 0;
                    xh = bh[ib];
                    xl = bl[ib];
                }
                if (ks + PFK < NK) {
                    // issue order = reverse of the order of first use (wh and xh by the k-step's first MFMA, xl by its
                    // second, wl by its third): the compiler's s_waitcnt in front of the first MFMA then covers the
                    // whole refill and the two further waits per k-step disappear (every instruction of the one wave
                    // per SIMD costs issue time, s_waitcnt included)
                    qwl[q] = lds_ld8(cur + (2 * (ks + PFK) + 1) * 1024);
                    if (ks + PFK < NKA) qxl[q] = lds_ld8(pe + (2 * (ks + PFK) + 1) * 1024);
                    qwh[q] = lds_ld8(cur + (2 * (ks + PFK)) * 1024);
                    if (ks + PFK < NKA) qxh[q] = lds_ld8(pe + (2 * (ks + PFK)) * 1024);
                }
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc1, 0, 0, 0);
                if (ULO) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc1, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc2, 0, 0, 0);
                if (ULO && WLO_UNLIFTED) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc1, 0, 0, 0);
                else acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc2, 0, 0, 0);
                if (NT > 1 && sv_has_bias(SAVE) && ks == NK - 1 && t + 1 < NT) {
                    const unsigned bn = p.lds_base + p.bias_off + (tt + 1) * 128 + (lane >> 5) * 16;   // (tt + 1 == TPS: next slab's first)
                    const f32x4 b0 = lds_ld4f(bn), b1 = lds_ld4f(bn + 32), b2 = lds_ld4f(bn + 64), b3 = lds_ld4f(bn + 96);
                    nbias = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                                   b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
                }
                {   // DMA piece gk of slab n+2 at slab-local k-step gk = tt NK + ks (from the slab's very first k-step:
                    // every k-step of head start shortens the vmcnt wait at the end of the next slab)
                    constexpr int MAXP = (TPS * NK - 1) < 10 ? (TPS * NK - 1) : 10;
                    const int i = tt * NK + ks;
                    if (i >= 0 && i < MAXP) {
                        if (s + 2 < NS) {                      // piece count known at compile time
                            if (i < NBLK * TPS / NW) dma_run_piece(dsrc, lane * 16, ddst, i);
                        } else if (i < dcnt) {                 // wave-uniform branch (layer-boundary slabs)
                            dma_run_piece(dsrc, lane * 16, ddst, i);
                        }
                    }
                }
                if (PEND_IN && t == 0) {   // the previous layer's last tile: fragments NKB-2, NKB-1 of this layer's input,
                                           // complete before the k-steps that read them (the last one at NK - 1)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((i * (NK - 1)) / 8 == ks) {
                            epi_pair_u<true>(*pend1, *pend2, i, bh[NKB - 2 + (i >> 2)], bl[NKB - 2 + (i >> 2)], p.sat, p.guard);
                            asm volatile("" : "+v"(bh[NKB - 2 + (i >> 2)]), "+v"(bl[NKB - 2 + (i >> 2)]));
                        }
                }
                if (t > 0) {      // pending epilogue of tile t-1: pair i runs at k-step (i * NK) / 8
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if ((i * NK) / 8 == ks) {
                            if constexpr (SAVE == SV_ACT) {
                                float x0, x1;
                                epi_pair_x<RELU, ULO>(pacc1, pacc2, i, oh[2 * (t - 1) + (i >> 2)], ol[2 * (t - 1) + (i >> 2)],
                                                 x0, x1);
                                save_pair(*sc, bw, t - 1, i, x0, x1);
                            } else if constexpr (SAVE == SV_ACT_H) {
                                float x0, x1;
                                const h16x2 hh = epi_pair_x<RELU, ULO>(pacc1, pacc2, i, oh[2 * (t - 1) + (i >> 2)], ol[2 * (t - 1) + (i >> 2)], x0, x1);
                                push_pair_bits(bw[(t - 1) >> 1], hh);
                                if (i & 1) store_frag_h(rowh, cxh, t - 1, i, oh[2 * (t - 1) + (i >> 2)]);
                            } else if constexpr (SAVE == SV_DZ) {
                                float x0, x1;
                                epi_pair_m(pacc1, pacc2, i, sc->mask[(t - 1) >> 1] >> (16 * ((t - 1) & 1)),
                                           oh[2 * (t - 1) + (i >> 2)], ol[2 * (t - 1) + (i >> 2)], x0, x1);
                                save_pair_b(*sc, t - 1, i, x0, x1);
                            } else if constexpr (SAVE == SV_DZ_H) {
                                const h16x2 hh = epi_pair_mh(pacc1, pacc2, i, sc->mask[(t - 1) >> 1], 8 * ((t - 1) & 1) + i,
                                                             oh[2 * (t - 1) + (i >> 2)], ol[2 * (t - 1) + (i >> 2)]);
                                store_pair_h(rowh, cxh, svu, t - 1, i, hh);
                            } else if constexpr (SAVE == SV_PE) {
                                sc->fout[16 * (t - 1) + 2 * i] = fmaf(pacc2[2 * i], LO_INV, pacc1[2 * i]);
                                sc->fout[16 * (t - 1) + 2 * i + 1] = fmaf(pacc2[2 * i + 1], LO_INV, pacc1[2 * i + 1]);
                            } else
                            epi_pair_u<RELU>(pacc1, pacc2, i, oh[2 * (t - 1) + (i >> 2)], ol[2 * (t - 1) + (i >> 2)], p.sat, p.guard);
                            // pin the result here: without a use in this block hipcc sinks the whole
                            // epilogue to the first consumer (the next layer), out of the MFMA shadow
                            if constexpr (SAVE != SV_PE)
                                asm volatile("" : "+v"(oh[2 * (t - 1) + (i >> 2)]), "+v"(ol[2 * (t - 1) + (i >> 2)]));
                        }
                }
                // issue order inside the k-step: keep the matrix pipe fed -- one MFMA, then <= 6 of the
                // pending VALU / LDS fillers (PMC: 6.4 non-MFMA instructions per MFMA; clumped behind the
                // third MFMA they left the pipe idle for a third of every k-step)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef HNRF_STAMP
            { const unsigned long long t1 = __builtin_readcyclecounter(); p.sum_k += t1 - p.t_last; p.t_last = t1; }
#endif
            pacc1 = acc1;
            pacc2 = acc2;
        }
        // slab n+1 landed (only the pieces of slab n+2 may still fly); every wave done with slab n.
        // SAVE: the activation stores issued during this slab (4 per tile that had a pending epilogue) are younger
        // than every piece of slab n+1 as well, so they may stay in flight too
        const int n_st = (sv_fwd(SAVE) || sv_bwd(SAVE)) ? 4 * (s == 0 ? TPS - 1 : TPS) : 0;
        tile_sync(nissue / NW + n_st);
        p.ph = p.ph == RING - 1 ? 0 : p.ph + 1;
        p.bias_off += TPS * 128;
    }
    if (NT == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) last[r] = (ULO && WLO_UNLIFTED) ? pacc1[r] : pacc1[r] + pacc2[r] * LO_INV;
    } else if (DEFER) {
        *pend1 = pacc1;
        *pend2 = pacc2;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (SAVE == SV_ACT) {
                float x0, x1;
                epi_pair_x<RELU, ULO>(pacc1, pacc2, i, oh[2 * (NT - 1) + (i >> 2)], ol[2 * (NT - 1) + (i >> 2)], x0, x1);
                save_pair(*sc, bw, NT - 1, i, x0, x1);
            } else if constexpr (SAVE == SV_ACT_H) {
                float x0, x1;
                const h16x2 hh = epi_pair_x<RELU, ULO>(pacc1, pacc2, i, oh[2 * (NT - 1) + (i >> 2)], ol[2 * (NT - 1) + (i >> 2)], x0, x1);
                push_pair_bits(bw[(NT - 1) >> 1], hh);
                if (i & 1) store_frag_h(rowh, cxh, NT - 1, i, oh[2 * (NT - 1) + (i >> 2)]);
            } else if constexpr (SAVE == SV_DZ) {
                float x0, x1;
                epi_pair_m(pacc1, pacc2, i, sc->mask[(NT - 1) >> 1] >> (16 * ((NT - 1) & 1)), oh[2 * (NT - 1) + (i >> 2)],
                           ol[2 * (NT - 1) + (i >> 2)], x0, x1);
                save_pair_b(*sc, NT - 1, i, x0, x1);
            } else if constexpr (SAVE == SV_DZ_H) {
                const h16x2 hh = epi_pair_mh(pacc1, pacc2, i, sc->mask[(NT - 1) >> 1], 8 * ((NT - 1) & 1) + i,
                                             oh[2 * (NT - 1) + (i >> 2)], ol[2 * (NT - 1) + (i >> 2)]);
                store_pair_h(rowh, cxh, svu, NT - 1, i, hh);
            } else if constexpr (SAVE == SV_PE) {
                sc->fout[16 * (NT - 1) + 2 * i] = fmaf(pacc2[2 * i], LO_INV, pacc1[2 * i]);
                sc->fout[16 * (NT - 1) + 2 * i + 1] = fmaf(pacc2[2 * i + 1], LO_INV, pacc1[2 * i + 1]);
            } else
            epi_pair_u<RELU>(pacc1, pacc2, i, oh[2 * (NT - 1) + (i >> 2)], ol[2 * (NT - 1) + (i >> 2)], p.sat, p.guard);
        }
        if constexpr (sv_fwd(SAVE)) {
#pragma unroll
            for (int i = 0; i < (NT + 1) / 2; ++i) sc->bits[i] = bw[i];
        }
    }
}

// Two sample groups per wave (64 samples): every weight fragment read from LDS feeds six MFMAs instead of three
// and the per-tile fixed costs (barrier, operand-queue priming, DMA issue) are paid once for twice the work.
// Only the 128-wide non-rigid MLP has the registers for it (2 x 128 fragment registers per group); inference form
// only.  bh / bl / oh / ol and `last` carry a leading group index; the PE stash holds group g at
// wave * stash_per_wave + g * stash_per_wave / 2.
template <int NT, int TPS, int NKA, int NKB, bool RELU, int NB, int NO>
__device__ __forceinline__ void layer16x2(Pipe& p, unsigned stash_per_wave, int nb1, int nb2,
                                          const h16x8 (&bh)[2][NB], const h16x8 (&bl)[2][NB], h16x8 (&oh)[2][NO],
                                          h16x8 (&ol)[2][NO], float (&last)[2][16]) {
    static_assert(NB >= (NKB > 0 ? NKB : 1) && NO >= 2 * NT && NT % TPS == 0, "bad layer shape");
    constexpr int NK = NKA + NKB;
    constexpr int NBLK = 2 * NK;
    constexpr int NS = NT / TPS;
    constexpr int PFK = NK < 4 ? NK : 4;
    static_assert((NBLK * TPS) % 4 == 0, "every wave must issue the same number of DMA pieces per slab");
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 pacc1[2] = {zero, zero}, pacc2[2] = {zero, zero};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int nissue = (s + 2 < NS) ? NBLK * TPS : (s + 2 == NS ? nb1 : nb2);
        const unsigned slot2 = p.ph == 0 ? 2 : p.ph - 1;
        const int dcnt = nissue / 4;                       // pieces per wave: this wave moves the run [wave dcnt, (wave + 1) dcnt)
        const char* dsrc = p.gi + p.wave * dcnt * 1024;
        const unsigned ddst = p.lds_base + p.ring_off + slot2 * p.slab_bytes + p.wave * dcnt * 1024;
        p.gi += nissue * 1024;
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            const int t = s * TPS + tt;
            const unsigned lane = lane_now();
            unsigned cur = p.lds_base + p.ring_off + p.ph * p.slab_bytes + tt * (NBLK * 1024) + lane * 16;
            unsigned pe = p.lds_base + p.pe_off + p.wave * stash_per_wave + lane * 16;
            asm volatile("" : "+v"(cur), "+v"(pe));        // see layer16
            f32x16 acc1[2] = {zero, zero}, acc2[2] = {zero, zero};
            if (NT > 1) {
                const unsigned bp = p.lds_base + p.bias_off + tt * 128 + (lane >> 5) * 16;
                const f32x4 b0 = lds_ld4f(bp), b1 = lds_ld4f(bp + 32), b2 = lds_ld4f(bp + 64), b3 = lds_ld4f(bp + 96);
                acc1[0] = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w,
                                 b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
                acc1[1] = acc1[0];
            }
            h16x8 qwh[PFK], qwl[PFK], qxh[2][PFK], qxl[2][PFK];
#pragma unroll
            for (int i = 0; i < PFK; ++i) {
                qwh[i] = lds_ld8(cur + (2 * i) * 1024);
                qwl[i] = lds_ld8(cur + (2 * i + 1) * 1024);
                if (i < NKA) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        qxh[g][i] = lds_ld8(pe + g * (stash_per_wave / 2) + (2 * i) * 1024);
                        qxl[g][i] = lds_ld8(pe + g * (stash_per_wave / 2) + (2 * i + 1) * 1024);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const int q = ks % PFK;
                const h16x8 wh = qwh[q], wl = qwl[q];
                h16x8 xh[2], xl[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (ks < NKA) {
                        xh[g] = qxh[g][q];
                        xl[g] = qxl[g][q];
                    } else {
                        const int ib = ks - NKA < NB ? ks - NKA : 0;
                        xh[g] = bh[g][ib];
                        xl[g] = bl[g][ib];
                    }
                }
                if (ks + PFK < NK) {
                    // (issue order = reverse of the order of first use, see layer16)
                    qwl[q] = lds_ld8(cur + (2 * (ks + PFK) + 1) * 1024);
                    if (ks + PFK < NKA) {
#pragma unroll
                        for (int g = 1; g >= 0; --g) qxl[g][q] = lds_ld8(pe + g * (stash_per_wave / 2) + (2 * (ks + PFK) + 1) * 1024);
                    }
                    qwh[q] = lds_ld8(cur + (2 * (ks + PFK)) * 1024);
                    if (ks + PFK < NKA) {
#pragma unroll
                        for (int g = 1; g >= 0; --g) qxh[g][q] = lds_ld8(pe + g * (stash_per_wave / 2) + (2 * (ks + PFK)) * 1024);
                    }
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[g], acc1[g], 0, 0, 0);
                    acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[g], acc1[g], 0, 0, 0);     // (xl un-scaled: epi_pair_u)
                    if (WLO_UNLIFTED) acc1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[g], acc1[g], 0, 0, 0);
                    else acc2[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[g], acc2[g], 0, 0, 0);
                }
                {
                    constexpr int MAXP = (TPS * NK - 1) < 10 ? (TPS * NK - 1) : 10;
                    const int i = tt * NK + ks;
                    if (i >= 0 && i < MAXP) {
                        if (s + 2 < NS) {
                            if (i < NBLK * TPS / 4) dma_run_piece(dsrc, lane * 16, ddst, i);
                        } else if (i < dcnt) {
                            dma_run_piece(dsrc, lane * 16, ddst, i);
                        }
                    }
                }
                if (t > 0) {      // pending epilogue of tile t-1: 16 pair-units (2 groups x 8), unit j at k-step (j NK) / 16
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if ((j * NK) / 16 == ks) {
                            const int g = j >> 3, i = j & 7;
                            epi_pair_u<RELU>(pacc1[g], pacc2[g], i, oh[g][2 * (t - 1) + (i >> 2)], ol[g][2 * (t - 1) + (i >> 2)], p.sat, p.guard);
                            asm volatile("" : "+v"(oh[g][2 * (t - 1) + (i >> 2)]), "+v"(ol[g][2 * (t - 1) + (i >> 2)]));
                        }
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) { pacc1[g] = acc1[g]; pacc2[g] = acc2[g]; }
        }
        tile_sync(nissue / 4);
        p.ph = p.ph == RING - 1 ? 0 : p.ph + 1;
        p.bias_off += TPS * 128;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (NT == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) last[g][r] = WLO_UNLIFTED ? pacc1[g][r] : pacc1[g][r] + pacc2[g][r] * LO_INV;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                epi_pair_u<RELU>(pacc1[g], pacc2[g], i, oh[g][2 * (NT - 1) + (i >> 2)], ol[g][2 * (NT - 1) + (i >> 2)], p.sat, p.guard);
        }
    }
}

// Starts the pipeline: bias table + the first two slabs are put in flight.
__device__ __forceinline__ Pipe pipe_start(const char* packed, int64_t bias_img_off, int bias_bytes, int slab_bytes,
                                           int nb0, int nb1, char* smem, int stash_bytes = PE_STASH, int nw = 4) {
    Pipe p;
    p.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    p.lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char*)smem);
    p.pe_off = bias_bytes;
    p.ring_off = bias_bytes + stash_bytes;
    p.slab_bytes = slab_bytes;
    p.ph = 0;
    p.bias_off = 0;
    p.sat = 0ull;
    p.guard = false;
#ifdef HNRF_STAMP
    p.t_last = __builtin_readcyclecounter();
    p.sum_k = p.sum_b = 0;
#endif
    slab_issue(packed + bias_img_off, p.lds_base, bias_bytes / 1024, p.wave, nw);
    slab_issue(packed, p.lds_base + p.ring_off, nb0, p.wave, nw);
    slab_issue(packed + nb0 * 1024, p.lds_base + p.ring_off + slab_bytes, nb1, p.wave, nw);
    p.gi = packed + (nb0 + nb1) * 1024;
    return p;
}

template <bool ULO = false>
__device__ __forceinline__ void stash_pe(const Pipe& p, int ks, const float (&v)[8]) {
    h16x8 hi, lo;
    if (ULO) split8_u(v, hi, lo);
    else split8(v, hi, lo);
    const unsigned pe = p.lds_base + p.pe_off + p.wave * (PE_STASH / 4) + lane_now() * 16;
    lds_st8(pe + (2 * ks) * 1024, hi);
    lds_st8(pe + (2 * ks + 1) * 1024, lo);
}

// K3, f16x3.  grid = ceil(P / 128) workgroups of 4 waves x 32 samples; LDS = 160 KiB.
// SAVE (training forward): also writes pe_out [P,63], acts [8][P][256] (fp32 post-ReLU values, as combined in the
// epilogue) and relu_bits [8][P][8] like canonical_f32_kernel<true>; no sparse form.  SAVE == SV_ACT_H: acts and
// pe_out are f16 matrices ([8][P][256], [P][64] with column 63 zero) for hnrf_mlp_dw_h.
template <int SAVE, bool GUARD = false>
__global__ __launch_bounds__(256) void canonical_f16x3_kernel(const float* __restrict__ xyz,
                                                              const char* __restrict__ packed, int64_t P,
                                                              float4* __restrict__ raw, const int* __restrict__ idx,
                                                              const int* __restrict__ count,
                                                              float* __restrict__ pe_out, float* __restrict__ acts,
                                                              uint32_t* __restrict__ relu_bits) {
    constexpr bool UL = true;                                    // un-scaled activation low parts (epi_pair_u): all forward forms
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // sparse launch: only the `*count` samples listed in idx are evaluated (hnrf_compact_samples)
    if (idx != nullptr) {
        P = *count;
        if ((int64_t)blockIdx.x * 128 >= P) return;           // whole workgroup, before any DMA / barrier
    }
    Pipe p = pipe_start(packed, CNL16_BIAS, CNL16_BIAS_LDS, CNL16_SLAB, 4 * CNL16_NB_L0, 4 * CNL16_NB_L0, smem);
    p.guard = GUARD && SAVE == SV_NONE;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * 4 + p.wave) * 32 + (lane & 31);
    const int64_t sclamp = slot < P ? slot : P - 1;
    const int64_t sidx = idx ? (int64_t)idx[sclamp] : sclamp;
    const int64_t sample = slot < P ? sidx : P;               // P = "do not store"

    // positional encoding -> LDS stash, while the first slabs fly
    const float x[3] = {xyz[sidx * 3 + 0], xyz[sidx * 3 + 1], xyz[sidx * 3 + 2]};
    float pev[32];                                // argument a = 3 k + axis; a = 30, 31: raw x
    pe_sincos_shared<10>(x, h, pev);
    pev[30] = h ? x[1] : x[0];
    pev[31] = h ? 0.f : x[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float v[8] = {pev[8 * ks], pev[8 * ks + 1], pev[8 * ks + 2], pev[8 * ks + 3],
                            pev[8 * ks + 4], pev[8 * ks + 5], pev[8 * ks + 6], pev[8 * ks + 7]};
        stash_pe<UL>(p, ks, v);
    }
    SaveCtx sc;
    if constexpr (SAVE) {
        // lanes past P hold a copy of sample P-1 and (re)write its rows: same values, and every wave issues the
        // same number of stores (see SaveCtx)
        sc.row = acts + sidx * 256 + 4 * h;
        sc.rowh = reinterpret_cast<_Float16*>(acts) + (slot >> 5) * (256 * 32) + h * 128;                        // blocked
        sc.cx = (lane & 31) ^ (4 * h);
        sc.bits = relu_bits + sidx * 8 + 4 * h;
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            const int col = pe_col16(PE16_CANONICAL, a, h);
            if constexpr (SAVE == SV_ACT_H) reinterpret_cast<_Float16*>(pe_out)[sidx * 64 + (col >= 0 ? col : 63)] = (_Float16)(col >= 0 ? pev[a] : 0.f);
            else if (col >= 0) pe_out[sidx * 63 + col] = pev[a];
        }
    }
    const int64_t act_stride = P * 256, bit_stride = P * 8;
    const int64_t acth_stride = (int64_t)gridDim.x * 128 * 256;     // blocked f16 layers are padded to whole workgroups
    tile_sync(0);

    h16x8 hA_h[16], hA_l[16], hB_h[16], hB_l[16];
    float last[16];
    f32x16 pend1, pend2;                          // inference: last-tile epilogues travel into the next layer
    constexpr bool DF = SAVE == SV_NONE;
    layer16<8, 4, 4, 0, true, SAVE, false, DF, UL>(p, CNL16_NB_MID, CNL16_NB_MID, hB_h, hB_l, hA_h, hA_l, last, &sc, &pend1, &pend2);   // (hB unused: NKB = 0)
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        const int nb = l == 4 ? CNL16_NB_L5 : CNL16_NB_MID;
        layer16<8, 1, 0, 16, true, SAVE, DF, DF, UL>(p, nb, nb, hA_h, hA_l, hB_h, hB_l, last, &sc, &pend1, &pend2);
        if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
#pragma unroll
        for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
    }
    layer16<8, 1, 4, 16, true, SAVE, DF, DF, UL>(p, CNL16_NB_MID, CNL16_NB_MID, hA_h, hA_l, hB_h, hB_l, last, &sc, &pend1, &pend2);   // skip layer
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
#pragma unroll 1
    for (int l = 6; l <= 7; ++l) {
        layer16<8, 1, 0, 16, true, SAVE, DF, DF, UL>(p, CNL16_NB_MID, l == 7 ? 0 : CNL16_NB_MID, hA_h, hA_l, hB_h, hB_l, last, &sc, &pend1, &pend2);
        if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
#pragma unroll
        for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
    }
    h16x8 dh[2], dl[2];
    layer16<1, 1, 0, 16, false, 0, DF, false, UL>(p, 0, 0, hA_h, hA_l, dh, dl, last, nullptr, &pend1, &pend2);
    // head bias and the head's power-of-two descale (pack_layer16_kernel, head_scale): scalar loads
    const float* ob = reinterpret_cast<const float*>(packed + CNL16_BIAS + CNL16_BIAS_LDS);
    const float hs = ob[8];
    if (h == 0 && slot < P)
        raw[sample] = make_float4(fmaf(last[0], hs, ob[0]), fmaf(last[1], hs, ob[1]), fmaf(last[2], hs, ob[2]), fmaf(last[3], hs, ob[3]));
    if constexpr (SAVE == SV_NONE && GUARD) raise_f16_range(packed, CNL16_STATUS, p.sat);
#ifdef HNRF_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 4096) {   // stamps leave through a buffer nothing else reads
        const unsigned long long te = __builtin_readcyclecounter();
        unsigned long long* dbg = (unsigned long long*)(const_cast<char*>(packed) + CNL16_BYTES);
        dbg[blockIdx.x * 4 + 0] = p.sum_k; dbg[blockIdx.x * 4 + 1] = p.sum_b + (te - p.t_last);
    }
#endif
}

// K2, f16x3 (width 128: 4 tiles, 8 k-steps).  SAVE: pe_out [P,36], acts [6][P][128], relu_bits [6][P][4];
// SV_ACT_H: f16 acts and an f16 pe_out [P][64] (columns 36..63 zero).
// NW = 8 (training forward, P a multiple of 256): EIGHT waves share the weight ring -- two per SIMD.  With one wave per
// SIMD every instruction of this issue-bound kernel (6.5 other instructions per MFMA) costs ~5 cycles next to the MFMAs'
// 32 (profiles/r03_mfma_issue.txt); a second wave fills those slots.  The registers allow it (239 VGPRs, no AGPRs); the
// LDS does with the two-group kernel's layout: 3 KiB bias + 8 x 8 KiB PE stash + 3 x 24 KiB slabs = 139 KiB, one tile per
// slab except layer 0's two.  Same image, same per-wave arithmetic and store addresses: bit-identical results.
constexpr int NR16W8_SLAB = 24 * 1024;
constexpr int NR16W8_STASH = 64 * 1024;

template <int SAVE, int NW = 4>
__global__ __launch_bounds__(64 * NW) void nonrigid_f16x3_kernel(const float* __restrict__ x_skel,
                                                             const float* __restrict__ hann_w,
                                                             const char* __restrict__ packed, int64_t P,
                                                             float* __restrict__ xyz, float* __restrict__ offsets,
                                                             const int* __restrict__ idx,
                                                             const int* __restrict__ count,
                                                             float* __restrict__ pe_out, float* __restrict__ acts,
                                                             uint32_t* __restrict__ relu_bits) {
    constexpr bool UL = true;                                    // un-scaled activation low parts (epi_pair_u): all forward forms
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (idx != nullptr) {
        P = *count;
        if ((int64_t)blockIdx.x * (32 * NW) >= P) return;
    }
    static_assert(NW == 4 || NW == 8, "waves per workgroup");
    constexpr bool W8 = NW == 8;
    Pipe p = W8 ? pipe_start(packed, NR16_BIAS, NR16_BIAS_LDS, NR16W8_SLAB, 2 * NR16_NB_L0, 2 * NR16_NB_L0, smem, NR16W8_STASH, 8)
                : pipe_start(packed, NR16_BIAS, NR16_BIAS_LDS, NR16_SLAB, 4 * NR16_NB_L0, 2 * NR16_NB_MID, smem);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * NW + p.wave) * 32 + (lane & 31);
    const int64_t sclamp = slot < P ? slot : P - 1;
    const int64_t sidx = idx ? (int64_t)idx[sclamp] : sclamp;
    const int64_t sample = slot < P ? sidx : P;

    const float x[3] = {x_skel[sidx * 3 + 0], x_skel[sidx * 3 + 1], x_skel[sidx * 3 + 2]};
    float pev[32];
    pe_sincos_shared<6>(x, h, pev);
#pragma unroll
    for (int a = 0; a < 18; ++a) pev[a] *= hann_w[a / 3];
#pragma unroll
    for (int a = 18; a < 32; ++a) pev[a] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const float v[8] = {pev[8 * ks], pev[8 * ks + 1], pev[8 * ks + 2], pev[8 * ks + 3],
                            pev[8 * ks + 4], pev[8 * ks + 5], pev[8 * ks + 6], pev[8 * ks + 7]};
        stash_pe<UL>(p, ks, v);
    }
    SaveCtx sc;
    if constexpr (SAVE) {
        sc.row = acts + sidx * 128 + 4 * h;
        sc.rowh = reinterpret_cast<_Float16*>(acts) + (slot >> 5) * (128 * 32) + h * 128;                        // blocked
        sc.cx = (lane & 31) ^ (4 * h);
        sc.bits = relu_bits + sidx * 4 + 2 * h;
        if constexpr (SAVE == SV_ACT_H) {
            _Float16* ph = reinterpret_cast<_Float16*>(pe_out) + sidx * 64;
#pragma unroll
            for (int a = 0; a < 18; ++a) ph[pe_col16(PE16_NONRIGID, a, h)] = (_Float16)pev[a];
#pragma unroll
            for (int a = 18; a < 32; ++a) ph[36 + 2 * (a - 18) + h] = (_Float16)0.f;
        } else {
#pragma unroll
            for (int a = 0; a < 18; ++a) pe_out[sidx * 36 + pe_col16(PE16_NONRIGID, a, h)] = pev[a];
        }
    }
    const int64_t act_stride = P * 128, bit_stride = P * 4;
    const int64_t acth_stride = (int64_t)gridDim.x * (32 * NW) * 128;       // blocked f16 layers are padded to whole workgroups
    tile_sync(0);

    h16x8 hA_h[8], hA_l[8], hB_h[8], hB_l[8];
    float last[16];
    h16x8 dh[2], dl[2];
    if constexpr (W8) {
        constexpr int MID = NR16_NB_MID, L4B = NR16_NB_L4;      // blocks per tile: 16 / 24
#define HNRF_NEXT_LAYER if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
        layer16<4, 2, 4, 0, true, SAVE, false, false, UL, 8>(p, MID, MID, hB_h, hB_l, hA_h, hA_l, last, &sc);        // L0
        HNRF_NEXT_LAYER
        layer16<4, 1, 0, 8, true, SAVE, false, false, UL, 8>(p, MID, MID, hA_h, hA_l, hB_h, hB_l, last, &sc);        // L1
        HNRF_NEXT_LAYER
        layer16<4, 1, 0, 8, true, SAVE, false, false, UL, 8>(p, MID, MID, hB_h, hB_l, hA_h, hA_l, last, &sc);        // L2
        HNRF_NEXT_LAYER
        layer16<4, 1, 0, 8, true, SAVE, false, false, UL, 8>(p, L4B, L4B, hA_h, hA_l, hB_h, hB_l, last, &sc);        // L3
        HNRF_NEXT_LAYER
        layer16<4, 1, 4, 8, true, SAVE, false, false, UL, 8>(p, MID, MID, hB_h, hB_l, hA_h, hA_l, last, &sc);        // skip layer
        HNRF_NEXT_LAYER
        layer16<4, 1, 0, 8, true, SAVE, false, false, UL, 8>(p, MID, 0, hA_h, hA_l, hB_h, hB_l, last, &sc);
#undef HNRF_NEXT_LAYER
        layer16<1, 1, 0, 8, false, 0, false, false, UL, 8>(p, 0, 0, hB_h, hB_l, dh, dl, last);
    } else {
    layer16<4, 4, 4, 0, true, SAVE, false, false, UL>(p, 0 /*already in flight*/, 2 * NR16_NB_MID, hB_h, hB_l, hA_h, hA_l, last, &sc);
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
    // 128-wide tiles have only 8 k-steps: two tiles per slab halve the barriers per MFMA.  The layers alternate
    // between the two fragment arrays (no loop with a copy-back: this kernel's time follows its instruction count)
    layer16<4, 2, 0, 8, true, SAVE, false, false, UL>(p, 2 * NR16_NB_MID, 2 * NR16_NB_MID, hA_h, hA_l, hB_h, hB_l, last, &sc);     // L1
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
    layer16<4, 2, 0, 8, true, SAVE, false, false, UL>(p, 2 * NR16_NB_MID, 2 * NR16_NB_MID, hB_h, hB_l, hA_h, hA_l, last, &sc);     // L2
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
    layer16<4, 2, 0, 8, true, SAVE, false, false, UL>(p, NR16_NB_L4, NR16_NB_L4, hA_h, hA_l, hB_h, hB_l, last, &sc);               // L3
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
    layer16<4, 1, 4, 8, true, SAVE, false, false, UL>(p, 2 * NR16_NB_MID, 2 * NR16_NB_MID, hB_h, hB_l, hA_h, hA_l, last, &sc);    // skip layer
    if constexpr (SAVE) { sc.row += act_stride; sc.rowh += acth_stride; sc.bits += bit_stride; }
    layer16<4, 2, 0, 8, true, SAVE, false, false, UL>(p, NR16_NB_MID, 0, hA_h, hA_l, hB_h, hB_l, last, &sc);
    layer16<1, 1, 0, 8, false, 0, false, false, UL>(p, 0, 0, hB_h, hB_l, dh, dl, last);
    }
    const float* ob = reinterpret_cast<const float*>(packed + NR16_BIAS + NR16_BIAS_LDS);
    const float hs = ob[8];                                     // head descale (pack_layer16_kernel, head_scale)
    if (h == 0 && slot < P) {
        const float o0 = fmaf(last[0], hs, ob[0]), o1 = fmaf(last[1], hs, ob[1]), o2 = fmaf(last[2], hs, ob[2]);
        xyz[sample * 3 + 0] = x[0] + o0;
        xyz[sample * 3 + 1] = x[1] + o1;
        xyz[sample * 3 + 2] = x[2] + o2;
        if (offsets) {
            offsets[sample * 3 + 0] = o0;
            offsets[sample * 3 + 1] = o1;
            offsets[sample * 3 + 2] = o2;
        }
    }
}

// K2, f16x3, two sample groups per wave (inference): grid = ceil(P / 256) workgroups of 4 waves x 64 samples.
// LDS = 3 KiB bias + 64 KiB PE stash (4 waves x 2 groups x 8 KiB) + 3 x 24 KiB slabs = 139 KiB; the weight image is
// the one nonrigid16_pack writes (tiles are consecutive; only the grouping into slabs differs: L0 2 tiles per slab,
// one tile per slab elsewhere).
constexpr int NR16X2_SLAB = 24 * 1024;
constexpr int NR16X2_STASH = 64 * 1024;

template <bool GUARD>
__global__ __launch_bounds__(256) void nonrigid_f16x3_x2_kernel(const float* __restrict__ x_skel,
                                                                const float* __restrict__ hann_w,
                                                                const char* __restrict__ packed, int64_t P,
                                                                float* __restrict__ xyz, float* __restrict__ offsets,
                                                                const int* __restrict__ idx,
                                                                const int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (idx != nullptr) {
        P = *count;
        if ((int64_t)blockIdx.x * 256 >= P) return;
    }
    // pipe_start lays the LDS out as [bias | PE_STASH | ring]: the stash of this kernel is twice as large
    Pipe p = pipe_start(packed, NR16_BIAS, NR16_BIAS_LDS, NR16X2_SLAB, 2 * NR16_NB_L0, 2 * NR16_NB_L0, smem,
                        NR16X2_STASH);
    p.guard = GUARD;
    constexpr unsigned SPW = NR16X2_STASH / 4;                  // stash bytes per wave
    const int lane = threadIdx.x & 63, h = lane >> 5;
    int64_t sidx[2];
    bool live[2];
    float x[2][3];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int64_t slot = ((int64_t)blockIdx.x * 4 + p.wave) * 64 + g * 32 + (lane & 31);
        const int64_t sclamp = slot < P ? slot : P - 1;
        sidx[g] = idx ? (int64_t)idx[sclamp] : sclamp;
        live[g] = slot < P;
#pragma unroll
        for (int a = 0; a < 3; ++a) x[g][a] = x_skel[sidx[g] * 3 + a];
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float pev[32];
        pe_sincos_shared<6>(x[g], h, pev);
#pragma unroll
        for (int a = 0; a < 18; ++a) pev[a] *= hann_w[a / 3];
#pragma unroll
        for (int a = 18; a < 32; ++a) pev[a] = 0.f;
        const unsigned pe = p.lds_base + p.pe_off + p.wave * SPW + g * (SPW / 2) + lane_now() * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float v[8] = {pev[8 * ks], pev[8 * ks + 1], pev[8 * ks + 2], pev[8 * ks + 3],
                                pev[8 * ks + 4], pev[8 * ks + 5], pev[8 * ks + 6], pev[8 * ks + 7]};
            h16x8 hi, lo;
            split8_u(v, hi, lo);
            lds_st8(pe + (2 * ks) * 1024, hi);
            lds_st8(pe + (2 * ks + 1) * 1024, lo);
        }
    }
    tile_sync(0);

    h16x8 hA_h[2][8], hA_l[2][8], hB_h[2][8], hB_l[2][8];
    float last[2][16];
    constexpr int MID = NR16_NB_MID, L4B = NR16_NB_L4;          // blocks per tile: 16 / 24
    layer16x2<4, 2, 4, 0, true>(p, SPW, MID, MID, hB_h, hB_l, hA_h, hA_l, last);            // L0 (hB unused: NKB = 0)
    // the layers alternate between the two fragment arrays: a loop over the three equal mid layers would have to copy
    // its output back into its input array, 128 registers per layer and group pair, most of them AGPR <-> VGPR moves --
    // a third of the kernel's non-MFMA instructions, and this kernel's time follows its instruction count
    layer16x2<4, 1, 0, 8, true>(p, SPW, MID, MID, hA_h, hA_l, hB_h, hB_l, last);             // L1
    layer16x2<4, 1, 0, 8, true>(p, SPW, MID, MID, hB_h, hB_l, hA_h, hA_l, last);             // L2
    layer16x2<4, 1, 0, 8, true>(p, SPW, L4B, L4B, hA_h, hA_l, hB_h, hB_l, last);             // L3
    layer16x2<4, 1, 4, 8, true>(p, SPW, MID, MID, hB_h, hB_l, hA_h, hA_l, last);             // skip layer
    layer16x2<4, 1, 0, 8, true>(p, SPW, MID, 0, hA_h, hA_l, hB_h, hB_l, last);
    h16x8 dh[2][2], dl[2][2];
    layer16x2<1, 1, 0, 8, false>(p, SPW, 0, 0, hB_h, hB_l, dh, dl, last);
    const float* ob = reinterpret_cast<const float*>(packed + NR16_BIAS + NR16_BIAS_LDS);
    const float hs = ob[8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (h == 0 && live[g]) {
            const float o0 = fmaf(last[g][0], hs, ob[0]), o1 = fmaf(last[g][1], hs, ob[1]), o2 = fmaf(last[g][2], hs, ob[2]);
            xyz[sidx[g] * 3 + 0] = x[g][0] + o0;
            xyz[sidx[g] * 3 + 1] = x[g][1] + o1;
            xyz[sidx[g] * 3 + 2] = x[g][2] + o2;
            if (offsets) {
                offsets[sidx[g] * 3 + 0] = o0;
                offsets[sidx[g] * 3 + 1] = o1;
                offsets[sidx[g] * 3 + 2] = o2;
            }
        }
    }
    if constexpr (GUARD) raise_f16_range(packed, NR16_STATUS, p.sat);
}

// ============================================================================ dX chain, split-f16
// The backward chain of the canonical MLP (hnrf_mlp_bwd.hip has the fp32-MFMA version and the derivation) on the
// same LDS-DMA weight pipeline as the forward: stage m multiplies W_{m+1}^T (transposed image, K order = the
// register layout of the previous stage's output, hid_feat16) with dZ_{m+1}, applies relu' from the sign mask,
// stores dZ_m (fp32, un-scaled) and hands the split result to the next stage.  Gradients are scaled by a power of
// two so that max |d_raw| lands in [2, 4): 14 binary orders of head-room for growth through the layers, and
// everything within 2^-16 of the running maximum keeps the full 22-bit split.  The two positional-encoding stages
// (rows in PE K-step order) leave fp32 values for the fused PE backward.
struct PackBwd16 {
    const float* W;        // forward nn.Linear weight (n_out, n_in)
    int n_out, n_in;
    int NT, NK;            // tiles of 32 rows (forward in-features) / k-steps of 16 (forward out-features)
    int row_kind;          // PE16_NONE: row rho <-> column col0 + rho; else rows in PE argument order
    int col0;
    int head;              // 1: K = the n_out (<= 4) head outputs, elements 0..3 of lane half 0 of k-step 0
    int64_t off;           // byte offset of the stage's first slab
    const float* amax;     // device: max |W| of this layer; the image is stored x 2^layer_exponent(amax)
};

// max |W_l| of every layer of an MLP: grid (layers, 16 slices), float maxima combined with an integer atomicMax (non-negative
// floats order like their bit patterns); the table is zeroed before the launch.  layer_exponent() of an entry is the
// power of two that layer's image is stored with.
struct LayerSet {
    const float* w[9];
    int n[9];
};
__global__ __launch_bounds__(256) void layer_amax_kernel(LayerSet ls, float* __restrict__ out) {
    __shared__ float red[256];
    const float* W = ls.w[blockIdx.x];
    const int n = ls.n[blockIdx.x];
    float m = 0.f;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < n; e += 256 * gridDim.y) m = fmaxf(m, fabsf(W[e]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(out) + blockIdx.x, __float_as_uint(red[0]));
}

struct PackBwdSet16 {
    PackBwd16 d[10];
};
__global__ void pack_bwd16_kernel(PackBwdSet16 set, char* __restrict__ packed) {
    const PackBwd16& d = set.d[blockIdx.y];                       // blockIdx.y = stage: one launch per MLP image
    const int slab_units = 2 * d.NK * 256;
    const int64_t n = (int64_t)d.NT * slab_units;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i / slab_units);
    const int u = (int)(i % slab_units);
    const int blk = u >> 8;
    const int ks = blk >> 1, part = blk & 1;
    const int lane = (u >> 2) & 63;
    const int j0 = (u & 3) * 2;
    const int h = lane >> 5;
    const int rho = 32 * t + (lane & 31);
    const int kexp = layer_exponent(*d.amax);
    int col;
    if (d.row_kind == PE16_NONE) {
        col = d.col0 + rho;
    } else {       // register (tile tt, r) on half hh holds row 32 tt + 8 (r >> 2) + 4 hh + (r & 3): PE argument 16 tt + r
        const int tt = rho >> 5, w = rho & 31;
        const int r = 4 * (w >> 3) + (w & 3), hh = (w >> 2) & 1;
        const int c = pe_col16(d.row_kind, 16 * tt + r, hh);
        col = c < 0 ? -1 : d.col0 + c;
    }
    _Float16 outv[2];
    for (int e = 0; e < 2; ++e) {
        const int j = j0 + e;
        const int o = d.head ? ((ks == 0 && h == 0 && j < d.n_out) ? j : -1) : hid_feat16(ks, j, h);
        float w = 0.f;
        if (o >= 0 && o < d.n_out && col >= 0 && col < d.n_in) w = ldexpf(d.W[(int64_t)o * d.n_in + col], kexp);
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)((w - (float)hi) * LO_SCALE);      // (the dX chains keep the lifted form)
        outv[e] = part ? lo : hi;
    }
    *reinterpret_cast<h16x2*>(packed + d.off + ((int64_t)t * slab_units + u) * 4) = h16x2{outv[0], outv[1]};
}

// image: head (8 tiles x 2 blocks) | W7^T W6^T W5^T(hidden rows) (8 x 32 each) | W5^T PE rows (2 x 32) |
//        W4^T .. W1^T (8 x 32 each) | W0^T PE rows (2 x 32)
constexpr int64_t CB16_HEAD = 0;
constexpr int64_t CB16_FULL = 8 * 32 * KB;
constexpr int64_t CB16_PE = 2 * 32 * KB;
constexpr int64_t CB16_L7 = CB16_HEAD + 16 * KB;
constexpr int64_t CB16_L5P = CB16_L7 + 3 * CB16_FULL;
constexpr int64_t CB16_L4 = CB16_L5P + CB16_PE;
constexpr int64_t CB16_L0P = CB16_L4 + 4 * CB16_FULL;
constexpr int64_t CB16_BYTES = CB16_L0P + CB16_PE;            // followed by float amax[9] (layer_amax_kernel), 256 bytes

__device__ __forceinline__ void chain_amax(SaveCtx& sc, float* __restrict__ slot) {
    float m = sc.amax;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (slot != nullptr && lane_now() == 0) atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(m));
    sc.amax = 0.f;
}

// HALF: dZ is an f16 matrix holding the chain's SCALED values (dz_scale[l] = the power of two dZ_l was multiplied by, for
// hnrf_mlp_dw_h); otherwise fp32, un-scaled, with the per-layer maxima in dz_amax.
template <bool HALF>
__global__ __launch_bounds__(256) void canonical_bwd16_kernel(const float* __restrict__ xyz,
                                                              const float4* __restrict__ d_raw,
                                                              const uint32_t* __restrict__ relu_bits,
                                                              const char* __restrict__ packed, int64_t P,
                                                              const float* __restrict__ d_raw_amax,
                                                              float* __restrict__ dZ, float* __restrict__ d_xyz,
                                                              float* __restrict__ dz_amax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SV = HALF ? SV_DZ_H : SV_DZ;
    // slabs: head (16 blocks), then 32-block slabs; the first THREE are put in flight here (the head stage has
    // only 8 k-steps to issue DMA pieces from, one short of a 32-block slab)
    Pipe p = pipe_start(packed, 0, CNL16_BIAS_LDS, CNL16_SLAB, 16, 32, smem);
    slab_issue(p.gi, p.lds_base + p.ring_off + 2 * CNL16_SLAB, 32, p.wave);
    p.gi += 32 * 1024;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * 4 + p.wave) * 32 + (lane & 31);
    const int64_t sample = slot < P ? slot : P - 1;            // lanes past P repeat sample P-1 (same values stored)
    const int64_t stride = P * 256, bstride = P * 8;
    const float* wmax = reinterpret_cast<const float*>(packed + CB16_BYTES);
    auto kt = [&](int l) { return layer_exponent(wmax[l]); };          // weights of layer l are stored x 2^kt(l)

    // non-finite incoming gradient (an overflowed loss): the scale would be undefined -- gradients of zero scale come
    // out as NaN everywhere instead of silently wrong (and the caller's finite check sees them)
    int ex = 0;
    const float am_in = *d_raw_amax;
    if (am_in > 0.f) (void)frexpf(am_in, &ex);
    const float scale = (am_in == am_in && am_in < 3.0e38f) ? ldexpf(1.0f, 2 - ex) : __builtin_nanf("");
    // HALF: lanes past P must leave zeros in the blocked dZ (the weight-gradient kernel reads whole blocks): a zero
    // incoming gradient does that for every stage
    const float4 g = (HALF && slot >= P) ? float4{0.f, 0.f, 0.f, 0.f} : d_raw[sample];
    {
        const float v[8] = {h ? 0.f : g.x * scale, h ? 0.f : g.y * scale, h ? 0.f : g.z * scale, h ? 0.f : g.w * scale,
                            0.f, 0.f, 0.f, 0.f};
        stash_pe(p, 0, v);
    }
    SaveCtx sc;
    float S = scale * ldexpf(1.0f, kt(8));                      // scale of the stage being computed (dZ7 = W8^T d_raw)
    sc.descale = 1.0f / S;
    sc.amax = 0.f;
    sc.row = dZ + 7 * stride + sample * 256 + 4 * h;
    const int64_t strideh = (int64_t)gridDim.x * 128 * 256;     // blocked f16 layers, padded to whole workgroups
    sc.rowh = reinterpret_cast<_Float16*>(dZ) + 7 * strideh + (slot >> 5) * (256 * 32) + h * 128;
    sc.cx = (lane & 31) ^ (4 * h);
    const uint32_t* mrow = relu_bits + 7 * bstride + sample * 8 + 4 * h;
    float* am = dz_amax ? dz_amax + 7 * HNRF_AMAX_SLOTS + (blockIdx.x % HNRF_AMAX_SLOTS) : nullptr;
    float* sout = (HALF && dz_amax && blockIdx.x == 0 && threadIdx.x == 0) ? dz_amax + 7 : nullptr;   // HALF: dz_amax = scale[8]
    if (sout) *sout = S;
    auto next_stage = [&](int w_next) {                          // after a stored stage: step a layer down; the next stage
        if constexpr (!HALF) chain_amax(sc, am);                 // multiplies by the weights of layer w_next
        sc.row -= stride;
        sc.rowh -= strideh;
        mrow -= bstride;
        if (am) am -= HNRF_AMAX_SLOTS;
        S *= ldexpf(1.0f, kt(w_next));
        sc.descale = 1.0f / S;
        if (sout && w_next > 0) { --sout; *sout = S; }           // (w_next = 0: the scale of layer 0's d PE, not a dZ)
    };
    auto load_mask = [&]() {
        const uint4 m = *reinterpret_cast<const uint4*>(mrow);
        sc.mask[0] = m.x; sc.mask[1] = m.y; sc.mask[2] = m.z; sc.mask[3] = m.w;
    };
    tile_sync(0);

    h16x8 hA_h[16], hA_l[16], hB_h[16], hB_l[16];
    h16x8 dh[4], dl[4];
    float last[16];
    load_mask();
    layer16<8, 8, 1, 0, false, SV>(p, 0, 0, hB_h, hB_l, hA_h, hA_l, last, &sc);                // dZ7 (hB unused: NKB = 0)
    next_stage(7);
#pragma unroll 1
    for (int m = 6; m >= 5; --m) {                                                             // dZ6, dZ5
        load_mask();
        layer16<8, 1, 0, 16, false, SV>(p, 32, 32, hA_h, hA_l, hB_h, hB_l, last, &sc);
        next_stage(m);
#pragma unroll
        for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
    }
    load_mask();
    layer16<8, 1, 0, 16, false, SV>(p, 32, 32, hA_h, hA_l, hB_h, hB_l, last, &sc);            // skip layer: dZ4 ...
    const float inv_skip = sc.descale;                                                         // (W5^T: hidden and PE rows share kt[5])
    next_stage(4);
    layer16<2, 1, 0, 16, false, SV_PE>(p, 32, 32, hA_h, hA_l, dh, dl, last, &sc);             // ... and its d PE
    // PE backward on the lane's own sample (argument a = 3 k + axis, half 0 = sin, half 1 = cos), folded into d_xyz right
    // here: carrying the 32 d PE values through the four stages below would cost 32 registers of a kernel that has none
    const float x[3] = {xyz[sample * 3 + 0], xyz[sample * 3 + 1], xyz[sample * 3 + 2]};
    float dx[3] = {0.f, 0.f, 0.f};
    auto pe_backward = [&](float sc_inv) {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int a = 0; a < 30; ++a) {
            float sv, cv;
            ph[a % 3].next(sv, cv);
            const float f = (float)(1 << (a / 3)) * sc_inv;
            dx[a % 3] += f * (h ? -sv : cv) * sc.fout[a];
        }
        dx[h ? 1 : 0] += sc.fout[30] * sc_inv;
        if (h == 0) dx[2] += sc.fout[31] * sc_inv;
    };
    pe_backward(inv_skip);
#pragma unroll
    for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
#pragma unroll 1
    for (int m = 3; m >= 0; --m) {                                                             // dZ3 .. dZ0
        load_mask();
        layer16<8, 1, 0, 16, false, SV>(p, 32, 32, hA_h, hA_l, hB_h, hB_l, last, &sc);
        next_stage(m);                                                                         // (m = 0: scale of layer 0's d PE)
#pragma unroll
        for (int i = 0; i < 16; ++i) { hA_h[i] = hB_h[i]; hA_l[i] = hB_l[i]; }
    }
    layer16<2, 1, 0, 16, false, SV_PE>(p, 0, 0, hA_h, hA_l, dh, dl, last, &sc);               // layer 0's d PE
    pe_backward(sc.descale);
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] += __shfl_xor(dx[a], 32, 64);
    if (h == 0 && slot < P) {
        d_xyz[sample * 3 + 0] = dx[0];
        d_xyz[sample * 3 + 1] = dx[1];
        d_xyz[sample * 3 + 2] = dx[2];
    }
}

// image of the non-rigid chain: head (4 tiles x 2 blocks) | W5^T (4 x 16) | W4^T hidden rows (4 x 16) |
//   W4^T PE rows (2 x 16) | W3^T W2^T W1^T (4 x 16 each) | W0^T PE rows (2 x 16); slabs of 32 blocks after the head
constexpr int64_t NB16_HEAD = 0;
constexpr int64_t NB16_FULL = 4 * 16 * KB;
constexpr int64_t NB16_PE = 2 * 16 * KB;
constexpr int64_t NB16_L5 = NB16_HEAD + 8 * KB;
constexpr int64_t NB16_L4P = NB16_L5 + 2 * NB16_FULL;
constexpr int64_t NB16_L3 = NB16_L4P + NB16_PE;
constexpr int64_t NB16_L0P = NB16_L3 + 3 * NB16_FULL;
constexpr int64_t NB16_BYTES = NB16_L0P + NB16_PE;            // followed by float amax[7], 256 bytes

// Non-rigid MLP, split-f16 (xyz = x_skel + offset): d_x_skel = d_xyz + J_offset^T d_xyz, dZ [6][P][128].
// HALF: see canonical_bwd16_kernel.
// NW = 8: eight waves per workgroup as in nonrigid_f16x3_kernel (one tile per slab, 16-KiB ring slots; P a multiple of 256)
template <bool HALF, int NW = 4>
__global__ __launch_bounds__(64 * NW) void nonrigid_bwd16_kernel(const float* __restrict__ x_skel,
                                                             const float* __restrict__ hann_w,
                                                             const float* __restrict__ d_xyz,
                                                             const uint32_t* __restrict__ relu_bits,
                                                             const char* __restrict__ packed, int64_t P,
                                                             const float* __restrict__ d_xyz_amax,
                                                             float* __restrict__ dZ, float* __restrict__ d_x_skel,
                                                             float* __restrict__ dz_amax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SV = HALF ? SV_DZ_H : SV_DZ;
    constexpr bool W8 = NW == 8;
    constexpr int SLAB = W8 ? 16 * 1024 : NR16_SLAB, SB = W8 ? 16 : 32;         // ring slot bytes; blocks per mid-layer slab
    constexpr int TP = W8 ? 1 : 2;                                               // tiles per slab of the 128-wide stages
    Pipe p = pipe_start(packed, 0, NR16_BIAS_LDS, SLAB, 8, SB, smem, W8 ? NR16W8_STASH : PE_STASH, NW);
    slab_issue(p.gi, p.lds_base + p.ring_off + 2 * SLAB, SB, p.wave, NW);       // third slab: see canonical_bwd16_kernel
    p.gi += SB * 1024;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t slot = ((int64_t)blockIdx.x * NW + p.wave) * 32 + (lane & 31);
    const int64_t sample = slot < P ? slot : P - 1;
    const int64_t stride = P * 128, bstride = P * 4;
    const float* wmax = reinterpret_cast<const float*>(packed + NB16_BYTES);
    auto kt = [&](int l) { return layer_exponent(wmax[l]); };

    int ex = 0;
    const float am_in = *d_xyz_amax;
    if (am_in > 0.f) (void)frexpf(am_in, &ex);
    const float scale = (am_in == am_in && am_in < 3.0e38f) ? ldexpf(1.0f, 2 - ex) : __builtin_nanf("");
    const bool dead = HALF && slot >= P;                        // see canonical_bwd16_kernel
    const float g[3] = {dead ? 0.f : d_xyz[sample * 3 + 0], dead ? 0.f : d_xyz[sample * 3 + 1], dead ? 0.f : d_xyz[sample * 3 + 2]};
    {
        const float v[8] = {h ? 0.f : g[0] * scale, h ? 0.f : g[1] * scale, h ? 0.f : g[2] * scale, 0.f, 0.f, 0.f, 0.f, 0.f};
        stash_pe(p, 0, v);
    }
    SaveCtx sc;
    float S = scale * ldexpf(1.0f, kt(6));                       // dZ5 = W6^T d_xyz
    sc.descale = 1.0f / S;
    sc.amax = 0.f;
    sc.mask[2] = sc.mask[3] = 0u;
    sc.row = dZ + 5 * stride + sample * 128 + 4 * h;
    const int64_t strideh = (int64_t)gridDim.x * (32 * NW) * 128;
    sc.rowh = reinterpret_cast<_Float16*>(dZ) + 5 * strideh + (slot >> 5) * (128 * 32) + h * 128;
    sc.cx = (lane & 31) ^ (4 * h);
    const uint32_t* mrow = relu_bits + 5 * bstride + sample * 4 + 2 * h;
    float* am = dz_amax ? dz_amax + 5 * HNRF_AMAX_SLOTS + (blockIdx.x % HNRF_AMAX_SLOTS) : nullptr;
    float* sout = (HALF && dz_amax && blockIdx.x == 0 && threadIdx.x == 0) ? dz_amax + 5 : nullptr;   // HALF: dz_amax = scale[6]
    if (sout) *sout = S;
    auto next_stage = [&](int w_next) {
        if constexpr (!HALF) chain_amax(sc, am);
        sc.row -= stride;
        sc.rowh -= strideh;
        mrow -= bstride;
        if (am) am -= HNRF_AMAX_SLOTS;
        S *= ldexpf(1.0f, kt(w_next));
        sc.descale = 1.0f / S;
        if (sout && w_next > 0) { --sout; *sout = S; }
    };
    auto load_mask = [&]() {
        const uint2 m = *reinterpret_cast<const uint2*>(mrow);
        sc.mask[0] = m.x; sc.mask[1] = m.y;
    };
    tile_sync(0);

    h16x8 hA_h[8], hA_l[8], hB_h[8], hB_l[8];
    h16x8 dh[4], dl[4];
    float last[16];
    load_mask();
    layer16<4, 4, 1, 0, false, SV, false, false, false, NW>(p, 0, 0, hB_h, hB_l, hA_h, hA_l, last, &sc);   // dZ5
    next_stage(5);
    load_mask();
    layer16<4, TP, 0, 8, false, SV, false, false, false, NW>(p, SB, SB, hA_h, hA_l, hB_h, hB_l, last, &sc);             // dZ4 (the skip layer's)
    next_stage(4);
    load_mask();
    layer16<4, TP, 0, 8, false, SV, false, false, false, NW>(p, SB, SB, hB_h, hB_l, hA_h, hA_l, last, &sc);             // skip [h | PE]: dZ3 ...
    const float inv_skip = sc.descale;                                                         // (W4^T: hidden and PE rows share kt[4])
    next_stage(3);
    layer16<2, TP, 0, 8, false, SV_PE, false, false, false, NW>(p, SB, SB, hB_h, hB_l, dh, dl, last, &sc);              // ... and its d PE
    float dpe[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) dpe[j] = sc.fout[j] * inv_skip;
    load_mask();
    layer16<4, TP, 0, 8, false, SV, false, false, false, NW>(p, SB, SB, hA_h, hA_l, hB_h, hB_l, last, &sc);             // dZ2
    next_stage(2);
    load_mask();
    layer16<4, TP, 0, 8, false, SV, false, false, false, NW>(p, SB, SB, hB_h, hB_l, hA_h, hA_l, last, &sc);             // dZ1
    next_stage(1);
    load_mask();
    layer16<4, TP, 0, 8, false, SV, false, false, false, NW>(p, SB, W8 ? SB : 0, hA_h, hA_l, hB_h, hB_l, last, &sc);   // dZ0
    next_stage(0);                                                                             // (scale of layer 0's d PE)
    layer16<2, TP, 0, 8, false, SV_PE, false, false, false, NW>(p, 0, 0, hB_h, hB_l, dh, dl, last, &sc);   // layer 0's d PE
#pragma unroll
    for (int j = 0; j < 18; ++j) dpe[j] = fmaf(sc.fout[j], sc.descale, dpe[j]);

    const float x[3] = {x_skel[sample * 3 + 0], x_skel[sample * 3 + 1], x_skel[sample * 3 + 2]};
    float dx[3] = {0.f, 0.f, 0.f};
    {
        OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
#pragma unroll
        for (int a = 0; a < 18; ++a) {
            float sv, cv;
            ph[a % 3].next(sv, cv);
            const float f = hann_w[a / 3] * (float)(1 << (a / 3));
            dx[a % 3] += f * (h ? -sv : cv) * dpe[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) dx[a] += __shfl_xor(dx[a], 32, 64);
    if (h == 0 && slot < P) {
        d_x_skel[sample * 3 + 0] = g[0] + dx[0];
        d_x_skel[sample * 3 + 1] = g[1] + dx[1];
        d_x_skel[sample * 3 + 2] = g[2] + dx[2];
    }
}

static int launch_pack16(const PackLayer16* ds, int count, const float* cond, char* packed, hipStream_t st) {
    PackSet16 set{};
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        set.d[i] = ds[i];
        const int64_t n = (int64_t)ds[i].NT * (2 * (ds[i].NKA + ds[i].NKB) * 256);
        nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(pack_layer16_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)count), dim3(256), 0, st, set, cond,
                       packed);
    return check_launch("hnrf pack (f16x3)");
}

size_t canonical16_bytes() { return (size_t)CNL16_BYTES; }
size_t nonrigid16_bytes() { return (size_t)NR16_BYTES; }
size_t canonical16_status_offset() { return (size_t)CNL16_STATUS; }
size_t nonrigid16_status_offset() { return (size_t)NR16_STATUS; }

int canonical16_pack(const float* const* w, const float* const* b, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    int rc;
    if (hipMemsetAsync(out + CNL16_BIAS, 0, 9 * 1024, st) != hipSuccess) {
        set_error("hnrf_canonical_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    (void)rc;
    PackLayer16 d[9];
    d[0] = PackLayer16{w[0], b[0], 256, 63, 8, 4, 0, PE16_CANONICAL, 0, 0, 0, CNL16_L0, CNL16_BIAS, 0};
    for (int l = 1; l <= 4; ++l)
        d[l] = PackLayer16{w[l], b[l], 256, 256, 8, 0, 16, PE16_NONE, 0, 0, 0,
                           CNL16_L1 + (l - 1) * 8 * CNL16_NB_MID * KB, CNL16_BIAS + l * 1024, 0};
    d[5] = PackLayer16{w[5], b[5], 256, 319, 8, 4, 16, PE16_CANONICAL, 0, 63, 0, CNL16_L5, CNL16_BIAS + 5 * 1024, 0};
    for (int l = 6; l <= 7; ++l)
        d[l] = PackLayer16{w[l], b[l], 256, 256, 8, 0, 16, PE16_NONE, 0, 0, 0,
                           CNL16_L6 + (l - 6) * 8 * CNL16_NB_MID * KB, CNL16_BIAS + l * 1024, 0};
    d[8] = PackLayer16{w[8], b[8], 4, 256, 1, 0, 16, PE16_NONE, 0, 0, 0, CNL16_OUT, CNL16_BIAS + 8 * 1024, 1};
    return launch_pack16(d, 9, nullptr, out, st);
}

int nonrigid16_pack(const float* const* w, const float* const* b, const float* cond, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    int rc;
    if (hipMemsetAsync(out + NR16_BIAS, 0, 4 * 1024, st) != hipSuccess) {
        set_error("hnrf_nonrigid_pack: memset failed");
        return HNRF_E_LAUNCH;
    }
    (void)rc;
    PackLayer16 d[7];
    d[0] = PackLayer16{w[0], b[0], 128, 105, 4, 4, 0, PE16_NONRIGID, 69, 0, 69, NR16_L0, NR16_BIAS, 0};
    for (int l = 1; l <= 3; ++l)
        d[l] = PackLayer16{w[l], b[l], 128, 128, 4, 0, 8, PE16_NONE, 0, 0, 0,
                           NR16_L1 + (l - 1) * 4 * NR16_NB_MID * KB, NR16_BIAS + l * 512, 0};
    // W4 columns: [h(128) | PE36] (mlp_offset.py:81-82); our K order is [PE | h]
    d[4] = PackLayer16{w[4], b[4], 128, 164, 4, 4, 8, PE16_NONRIGID, 128, 0, 0, NR16_L4, NR16_BIAS + 4 * 512, 0};
    d[5] = PackLayer16{w[5], b[5], 128, 128, 4, 0, 8, PE16_NONE, 0, 0, 0, NR16_L5, NR16_BIAS + 5 * 512, 0};
    d[6] = PackLayer16{w[6], b[6], 3, 128, 1, 0, 8, PE16_NONE, 0, 0, 0, NR16_OUT, NR16_BIAS + 6 * 512, 1};
    return launch_pack16(d, 7, cond, out, st);
}

int canonical16_fwd(const float* xyz, const void* packed, int64_t P, float* raw, const int* idx, const int* count,
                    bool guard, hipStream_t st) {
    constexpr int lds = CNL16_BIAS_LDS + PE_STASH + RING * CNL16_SLAB;
    static unsigned long long lds_done = 0, lds_done_g = 0;
    const dim3 grid((unsigned)((P + 127) / 128));
    if (guard) {
        if (int rc = reserve_lds((const void*)canonical_f16x3_kernel<SV_NONE, true>, lds, lds_done_g, "hnrf_canonical_fwd (f16x3)")) return rc;
        hipLaunchKernelGGL((canonical_f16x3_kernel<SV_NONE, true>), grid, dim3(256), lds, st, xyz, (const char*)packed, P,
                           (float4*)raw, idx, count, nullptr, nullptr, nullptr);
    } else {
        if (int rc = reserve_lds((const void*)canonical_f16x3_kernel<SV_NONE, false>, lds, lds_done, "hnrf_canonical_fwd (f16x3)")) return rc;
        hipLaunchKernelGGL((canonical_f16x3_kernel<SV_NONE, false>), grid, dim3(256), lds, st, xyz, (const char*)packed, P,
                           (float4*)raw, idx, count, nullptr, nullptr, nullptr);
    }
    return check_launch("hnrf_canonical_fwd (f16x3)");
}

int canonical16_fwd_train(const float* xyz, const void* packed, int64_t P, float* raw, float* pe_out, float* acts,
                          uint32_t* relu_bits, int half, hipStream_t st) {
    constexpr int lds = CNL16_BIAS_LDS + PE_STASH + RING * CNL16_SLAB;
    static unsigned long long lds_done = 0, lds_done_h = 0;
    const dim3 grid((unsigned)((P + 127) / 128));
    if (half) {
        if (int rc = reserve_lds((const void*)canonical_f16x3_kernel<SV_ACT_H>, lds, lds_done_h, "hnrf_canonical_fwd_train (f16x3, f16 operands)")) return rc;
        hipLaunchKernelGGL(canonical_f16x3_kernel<SV_ACT_H>, grid, dim3(256), lds, st, xyz, (const char*)packed, P,
                           (float4*)raw, nullptr, nullptr, pe_out, acts, relu_bits);
    } else {
        if (int rc = reserve_lds((const void*)canonical_f16x3_kernel<SV_ACT>, lds, lds_done, "hnrf_canonical_fwd_train (f16x3)")) return rc;
        hipLaunchKernelGGL(canonical_f16x3_kernel<SV_ACT>, grid, dim3(256), lds, st, xyz, (const char*)packed, P,
                           (float4*)raw, nullptr, nullptr, pe_out, acts, relu_bits);
    }
    return check_launch("hnrf_canonical_fwd_train (f16x3)");
}

int nonrigid16_fwd(const float* x_skel, const float* hann_w, const void* packed, int64_t P, float* xyz,
                   float* offsets, const int* idx, const int* count, bool guard, hipStream_t st) {
    constexpr int lds2 = NR16_BIAS_LDS + NR16X2_STASH + RING * NR16X2_SLAB;
    static unsigned long long lds2_done = 0, lds2_done_g = 0;
    const dim3 grid((unsigned)((P + 255) / 256));
    if (guard) {
        if (int rc = reserve_lds((const void*)nonrigid_f16x3_x2_kernel<true>, lds2, lds2_done_g, "hnrf_nonrigid_fwd (f16x3)")) return rc;
        hipLaunchKernelGGL(nonrigid_f16x3_x2_kernel<true>, grid, dim3(256), lds2, st, x_skel, hann_w, (const char*)packed, P, xyz,
                           offsets, idx, count);
    } else {
        if (int rc = reserve_lds((const void*)nonrigid_f16x3_x2_kernel<false>, lds2, lds2_done, "hnrf_nonrigid_fwd (f16x3)")) return rc;
        hipLaunchKernelGGL(nonrigid_f16x3_x2_kernel<false>, grid, dim3(256), lds2, st, x_skel, hann_w, (const char*)packed, P, xyz,
                           offsets, idx, count);
    }
    return check_launch("hnrf_nonrigid_fwd (f16x3)");
}

int nonrigid16_fwd_train(const float* x_skel, const float* hann_w, const void* packed, int64_t P, float* xyz,
                         float* offsets, float* pe_out, float* acts, uint32_t* relu_bits, int half, hipStream_t st) {
    constexpr int lds = NR16_BIAS_LDS + PE_STASH + RING * NR16_SLAB;
    static unsigned long long lds_done = 0, lds_done_h = 0;
    const dim3 grid((unsigned)((P + 127) / 128));
    // eight waves per workgroup where the sample count allows it (whole 256-sample workgroups: the blocked activation
    // layers are padded to whole workgroups, and the chain / weight-gradient kernels pad to 128); HNRF_K2T_W4: the
    // four-wave form, for A/B runs
    static const bool w8_ok = getenv("HNRF_K2T_W4") == nullptr;
    if (half && w8_ok && P % 256 == 0) {
        constexpr int lds8 = NR16_BIAS_LDS + NR16W8_STASH + RING * NR16W8_SLAB;
        static unsigned long long lds_done_8 = 0;
        if (int rc = reserve_lds((const void*)nonrigid_f16x3_kernel<SV_ACT_H, 8>, lds8, lds_done_8, "hnrf_nonrigid_fwd_train (f16x3, f16 operands, 8 waves)")) return rc;
        hipLaunchKernelGGL((nonrigid_f16x3_kernel<SV_ACT_H, 8>), dim3((unsigned)(P / 256)), dim3(512), lds8, st, x_skel, hann_w,
                           (const char*)packed, P, xyz, offsets, nullptr, nullptr, pe_out, acts, relu_bits);
    } else if (half) {
        if (int rc = reserve_lds((const void*)nonrigid_f16x3_kernel<SV_ACT_H>, lds, lds_done_h, "hnrf_nonrigid_fwd_train (f16x3, f16 operands)")) return rc;
        hipLaunchKernelGGL(nonrigid_f16x3_kernel<SV_ACT_H>, grid, dim3(256), lds, st, x_skel, hann_w, (const char*)packed, P,
                           xyz, offsets, nullptr, nullptr, pe_out, acts, relu_bits);
    } else {
        if (int rc = reserve_lds((const void*)nonrigid_f16x3_kernel<SV_ACT>, lds, lds_done, "hnrf_nonrigid_fwd_train (f16x3)")) return rc;
        hipLaunchKernelGGL(nonrigid_f16x3_kernel<SV_ACT>, grid, dim3(256), lds, st, x_skel, hann_w, (const char*)packed, P,
                           xyz, offsets, nullptr, nullptr, pe_out, acts, relu_bits);
    }
    return check_launch("hnrf_nonrigid_fwd_train (f16x3)");
}

size_t canonical16_bwd_bytes() { return (size_t)CB16_BYTES + 256; }

static int launch_layer_amax(const float* const* w, const int* n, int count, float* out, hipStream_t st) {
    LayerSet ls;
    for (int i = 0; i < 9; ++i) { ls.w[i] = i < count ? w[i] : nullptr; ls.n[i] = i < count ? n[i] : 0; }
    if (hipMemsetAsync(out, 0, 64, st) != hipSuccess) {
        set_error("hnrf pack (layer maxima): memset failed");
        return HNRF_E_LAUNCH;
    }
    hipLaunchKernelGGL(layer_amax_kernel, dim3(count, 16), dim3(256), 0, st, ls, out);
    return check_launch("hnrf pack (layer maxima)");
}

static int launch_pack_bwd16(const PackBwd16* ds, int count, char* out, hipStream_t st) {
    PackBwdSet16 set{};
    int64_t nmax = 0;
    for (int i = 0; i < count; ++i) {
        set.d[i] = ds[i];
        const int64_t n = (int64_t)ds[i].NT * 2 * ds[i].NK * 256;
        nmax = n > nmax ? n : nmax;
    }
    hipLaunchKernelGGL(pack_bwd16_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)count), dim3(256), 0, st, set, out);
    return check_launch("hnrf pack (backward, f16x3)");
}

int canonical16_bwd_pack(const float* const* w, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    float* kexp = reinterpret_cast<float*>(out + CB16_BYTES);
    const int sizes[9] = {256 * 63, 256 * 256, 256 * 256, 256 * 256, 256 * 256, 256 * 319, 256 * 256, 256 * 256, 4 * 256};
    int rc;
    if ((rc = launch_layer_amax(w, sizes, 9, kexp, st))) return rc;
    PackBwd16 d[10];
    int n = 0;
    d[n++] = PackBwd16{w[8], 4, 256, 8, 1, PE16_NONE, 0, 1, CB16_HEAD, kexp + 8};
    for (int l = 7; l >= 6; --l) d[n++] = PackBwd16{w[l], 256, 256, 8, 16, PE16_NONE, 0, 0, CB16_L7 + (7 - l) * CB16_FULL, kexp + l};
    d[n++] = PackBwd16{w[5], 256, 319, 8, 16, PE16_NONE, 63, 0, CB16_L7 + 2 * CB16_FULL, kexp + 5};
    d[n++] = PackBwd16{w[5], 256, 319, 2, 16, PE16_CANONICAL, 0, 0, CB16_L5P, kexp + 5};
    for (int l = 4; l >= 1; --l) d[n++] = PackBwd16{w[l], 256, 256, 8, 16, PE16_NONE, 0, 0, CB16_L4 + (4 - l) * CB16_FULL, kexp + l};
    d[n++] = PackBwd16{w[0], 256, 63, 2, 16, PE16_CANONICAL, 0, 0, CB16_L0P, kexp};
    return launch_pack_bwd16(d, n, out, st);
}

// half != 0: dZ is an f16 matrix in the chain's scaled domain and dz_amax receives the [8] scales instead of maxima
int canonical16_bwd(const float* xyz, const float* d_raw, const uint32_t* relu_bits, const void* packed, int64_t P,
                    const float* d_raw_amax, float* dZ, float* d_xyz, float* dz_amax, int half, hipStream_t st) {
    constexpr int lds = CNL16_BIAS_LDS + PE_STASH + RING * CNL16_SLAB;
    static unsigned long long lds_done = 0, lds_done_h = 0;
    const dim3 grid((unsigned)((P + 127) / 128));
    if (half) {
        if (int rc = reserve_lds((const void*)canonical_bwd16_kernel<true>, lds, lds_done_h, "hnrf_canonical_bwd (f16x3, f16 operands)")) return rc;
        hipLaunchKernelGGL(canonical_bwd16_kernel<true>, grid, dim3(256), lds, st, xyz, (const float4*)d_raw, relu_bits,
                           (const char*)packed, P, d_raw_amax, dZ, d_xyz, dz_amax);
    } else {
        if (int rc = reserve_lds((const void*)canonical_bwd16_kernel<false>, lds, lds_done, "hnrf_canonical_bwd (f16x3)")) return rc;
        hipLaunchKernelGGL(canonical_bwd16_kernel<false>, grid, dim3(256), lds, st, xyz, (const float4*)d_raw, relu_bits,
                           (const char*)packed, P, d_raw_amax, dZ, d_xyz, dz_amax);
    }
    return check_launch("hnrf_canonical_bwd (f16x3)");
}

size_t nonrigid16_bwd_bytes() { return (size_t)NB16_BYTES + 256; }

int nonrigid16_bwd_pack(const float* const* w, void* packed, hipStream_t st) {
    char* out = (char*)packed;
    float* kexp = reinterpret_cast<float*>(out + NB16_BYTES);
    const int sizes[7] = {128 * 105, 128 * 128, 128 * 128, 128 * 128, 128 * 164, 128 * 128, 3 * 128};
    int rc;
    if ((rc = launch_layer_amax(w, sizes, 7, kexp, st))) return rc;
    PackBwd16 d[10];
    int n = 0;
    d[n++] = PackBwd16{w[6], 3, 128, 4, 1, PE16_NONE, 0, 1, NB16_HEAD, kexp + 6};
    d[n++] = PackBwd16{w[5], 128, 128, 4, 8, PE16_NONE, 0, 0, NB16_L5, kexp + 5};
    d[n++] = PackBwd16{w[4], 128, 164, 4, 8, PE16_NONE, 0, 0, NB16_L5 + NB16_FULL, kexp + 4};
    d[n++] = PackBwd16{w[4], 128, 164, 2, 8, PE16_NONRIGID, 128, 0, NB16_L4P, kexp + 4};
    for (int l = 3; l >= 1; --l) d[n++] = PackBwd16{w[l], 128, 128, 4, 8, PE16_NONE, 0, 0, NB16_L3 + (3 - l) * NB16_FULL, kexp + l};
    d[n++] = PackBwd16{w[0], 128, 105, 2, 8, PE16_NONRIGID, 69, 0, NB16_L0P, kexp};
    return launch_pack_bwd16(d, n, out, st);
}

int nonrigid16_bwd(const float* x_skel, const float* hann_w, const float* d_xyz, const uint32_t* relu_bits,
                   const void* packed, int64_t P, const float* d_xyz_amax, float* dZ, float* d_x_skel, float* dz_amax,
                   int half, hipStream_t st) {
    constexpr int lds = NR16_BIAS_LDS + PE_STASH + RING * NR16_SLAB;
    static unsigned long long lds_done = 0, lds_done_h = 0;
    const dim3 grid((unsigned)((P + 127) / 128));
    // (an eight-wave instance as in nonrigid16_fwd_train -- nonrigid_bwd16_kernel<true, 8> -- was built and measured: the chain
    // needs 292 registers per lane, at 256 it spills 260 bytes and runs 0.659 instead of 0.605 ms; not dispatched)
    if (half) {
        if (int rc = reserve_lds((const void*)nonrigid_bwd16_kernel<true>, lds, lds_done_h, "hnrf_nonrigid_bwd (f16x3, f16 operands)")) return rc;
        hipLaunchKernelGGL(nonrigid_bwd16_kernel<true>, grid, dim3(256), lds, st, x_skel, hann_w, d_xyz, relu_bits,
                           (const char*)packed, P, d_xyz_amax, dZ, d_x_skel, dz_amax);
    } else {
        if (int rc = reserve_lds((const void*)nonrigid_bwd16_kernel<false>, lds, lds_done, "hnrf_nonrigid_bwd (f16x3)")) return rc;
        hipLaunchKernelGGL(nonrigid_bwd16_kernel<false>, grid, dim3(256), lds, st, x_skel, hann_w, d_xyz, relu_bits,
                           (const char*)packed, P, d_xyz_amax, dZ, d_x_skel, dz_amax);
    }
    return check_launch("hnrf_nonrigid_bwd (f16x3)");
}

}  // namespace hnrf
