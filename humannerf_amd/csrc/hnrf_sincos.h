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
};

}  // namespace hnrf
