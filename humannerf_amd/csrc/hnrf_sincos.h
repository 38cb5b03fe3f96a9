// Accurate sin/cos of the positional-encoding arguments x * 2^k, k = 0..NB-1.
//
// The reference evaluates sin(x * 2^k) on the fp32 product (exact, it is a power-of-two
// scaling), with arguments up to hundreds of radians (fourier.py:21-24), so a fast
// __sinf is out and libm's sincosf pays a Payne-Hanek reduction per call (~150
// instructions; 30 calls per sample were 20-30 % of the f16x3 MLP kernel).  Here the
// reduction is done ONCE per axis in fp64 and carried across the octaves exactly:
//   u_0 = x / (2 pi)  (fp64),  f_k = u_k - rint(u_k) in [-1/2, 1/2] revolutions,
//   u_{k+1} = 2 f_k  (exact doubling, exact subtraction),
// then sin / cos (2 pi f_k) from fp32 Taylor polynomials on |t| <= 1/8 revolution after
// a quadrant split.  Phase error: 2^k |u_0| 2^-53 revolutions (< 1e-12); polynomial +
// Horner rounding ~1-2 ulp -- the same class as libm's 1-ulp sincosf.
#pragma once
#include <hip/hip_runtime.h>

namespace hnrf {

// sin / cos of 2*pi*r, r in revolutions with |r| <= 0.5
__device__ __forceinline__ void sincos_rev(float r, float& s, float& c) {
    const float q = rintf(r * 4.0f);              // nearest quarter turn: -2 .. 2
    const float t = fmaf(q, -0.25f, r);           // exact, |t| <= 1/8
    const float t2 = t * t;
    float sn = fmaf(t2, -15.094642576822990f, 42.058693944897658f);   // (2pi)^11/11!, (2pi)^9/9!
    sn = fmaf(t2, sn, -76.705859753061385f);
    sn = fmaf(t2, sn, 81.605249276075054f);
    sn = fmaf(t2, sn, -41.341702240399755f);
    sn = fmaf(t2, sn, 6.2831853071795865f);
    sn *= t;
    float cs = fmaf(t2, -26.426256783374397f, 60.244641371876660f);   // (2pi)^10/10!, (2pi)^8/8!
    cs = fmaf(t2, cs, -85.456817206693725f);
    cs = fmaf(t2, cs, 64.939394022668291f);
    cs = fmaf(t2, cs, -19.739208802178717f);
    cs = fmaf(t2, cs, 1.0f);
    const int qi = (int)q & 3;                    // rotation by qi quarter turns
    const float s1 = (qi & 1) ? cs : sn;
    const float c1 = (qi & 1) ? sn : cs;
    s = (qi == 2 || qi == 3) ? -s1 : s1;
    c = (qi == 1 || qi == 2) ? -c1 : c1;
}

// State of one axis: call next() once per octave, in order k = 0, 1, 2, ...
struct OctavePhase {
    double u;
    __device__ __forceinline__ explicit OctavePhase(float x) : u((double)x * 0.15915494309189535) {}
    __device__ __forceinline__ void next(float& s, float& c) {
        const double f = u - rint(u);
        u = f + f;
        sincos_rev((float)f, s, c);
    }
    // the same octave sequence taken two at a time (4 f is as exact as 2 (2 f)): skip() moves one octave on
    // without evaluating, next2() evaluates this octave and moves two on
    __device__ __forceinline__ void skip() {
        const double f = u - rint(u);
        u = f + f;
    }
    __device__ __forceinline__ void next2(float& s, float& c) {
        const double f = u - rint(u);
        u = 4.0 * f;
        sincos_rev((float)f, s, c);
    }
};

// Value held by the partner lane (lane ^ 32) -- the other half of the same sample in the MLP kernels' layout.
__device__ __forceinline__ float swap32(float v, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(h ? r[0] : r[1]);
}

// Positional-encoding values of one sample in the MLP kernels' B-operand order: out[3 k + axis] = sin (lane half 0)
// or cos (lane half 1) of 2^k x[axis], k < NB (even).  The two lanes of a sample share the work: half h evaluates
// the octaves of parity h (sin AND cos come out of one evaluation), keeps the one it needs and hands the other to
// its partner with one v_permlane32_swap -- 3 NB / 2 evaluations per lane instead of 3 NB, bit-identical values.
template <int NB>
__device__ __forceinline__ void pe_sincos_shared(const float (&x)[3], int h, float* out) {
    static_assert(NB % 2 == 0, "octaves are taken in pairs");
    OctavePhase ph[3] = {OctavePhase(x[0]), OctavePhase(x[1]), OctavePhase(x[2])};
    if (h) {
#pragma unroll
        for (int a = 0; a < 3; ++a) ph[a].skip();
    }
#pragma unroll
    for (int k2 = 0; k2 < NB / 2; ++k2)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            float sv, cv;
            ph[ax].next2(sv, cv);
            const float own = h ? cv : sv;
            const float recv = swap32(h ? sv : cv, h);
            out[6 * k2 + ax] = h ? recv : own;          // octave 2 k2
            out[6 * k2 + 3 + ax] = h ? own : recv;      // octave 2 k2 + 1
        }
}

}  // namespace hnrf
