// K-order / register-layout conventions shared by the fp32-MFMA MLP kernels (forward: hnrf_mlp.hip,
// backward chain: hnrf_mlp_bwd.hip).
#pragma once
#include "hnrf_common.h"

namespace hnrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Weight prefetch distance in groups (4 K-steps = 256 MFMA cycles each).  The weight stream shares the in-order
// vmcnt queue with the training kernels' activation stores, whose acknowledgements take longer than an L2 hit.
constexpr int PF = 16;

// Feature of the previous layer's output contracted at K-step j on lane half h.
__host__ __device__ inline int hid_feat(int j, int h) {
    const int t = j >> 4, r = j & 15;
    return 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
}

enum { PE_NONE = 0, PE_CANONICAL = 1, PE_NONRIGID = 2 };

// Column of W contracted by PE K-step j on lane half h (-1: zero padding).
// canonical (embedders/fourier.py): [x(3) | sin(2^k x)(3) cos(2^k x)(3)]_k=0..9
//   steps 0..29 = (band k, axis): h=0 sin, h=1 cos; step 30 = (x0, x1); 31 = (x2, 0).
// non-rigid (embedders/hannw_fourier.py): [w_k sin(2^k x)(3) w_k cos(2^k x)(3)]_k=0..5
//   steps 0..17 = (band k, axis): h=0 sin, h=1 cos; steps 18.. = padding.
__device__ inline int pe_col(int kind, int j, int h) {
    if (kind == PE_CANONICAL) {
        if (j < 30) return 3 + 6 * (j / 3) + 3 * h + (j % 3);
        if (j == 30) return h;
        return h == 0 ? 2 : -1;
    }
    if (j < 18) return 6 * (j / 3) + 3 * h + (j % 3);
    return -1;
}

}  // namespace hnrf
