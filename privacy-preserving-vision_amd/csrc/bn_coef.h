// Train-mode BatchNorm coefficients from folded partial sums -- ONE operation sequence (explicit fused multiply-adds, no compiler
// contraction choices), shared by every kernel that derives them itself (bn_act_fold_wg_kernel of trunk_ops.hip, the BNIN halo kernel of
// conv_halo.hip): two kernels given the same sums produce the same bits.  Semantics: torch.nn.BatchNorm2d in train mode behind
// Image_Caption/train.py:245 (biased variance for the normalisation, unbiased for the running estimate, eps inside the square root).
#pragma once
#include <hip/hip_runtime.h>

namespace ppv {

struct BnCoef { float sc, sh, mean, invstd, var_unbiased; };

// s, q: sum and sum of squares over `1 / inv_count` samples (f64 totals of the f32 partial rows, added in row order)
__device__ __forceinline__ BnCoef bn_coef_pinned(double s, double q, double inv_count, double unbias, float gamma, float beta, float eps) {
    const double mean = s * inv_count;
    double var = __builtin_fma(-mean, mean, q * inv_count);
    if (var < 0) var = 0;
    const float ve = (float)(var + (double)eps);
    // no f64 division / square root (hundreds of instructions each): the f32 reciprocal square root refined by one Newton step
    float invstd = __builtin_amdgcn_rsqf(ve);
    const float h = (0.5f * ve) * invstd;
    invstd = invstd * __builtin_fmaf(-h, invstd, 1.5f);
    BnCoef r;
    r.sc = gamma * invstd;
    r.sh = __builtin_fmaf(-(float)mean, r.sc, beta);
    r.mean = (float)mean;
    r.invstd = invstd;
    r.var_unbiased = (float)(var * unbias);
    return r;
}

// y = relu(x * sc + sh) on one value, the arithmetic of every BatchNorm + ReLU apply
__device__ __forceinline__ float bn_relu_pinned(float x, float sc, float sh) { return fmaxf(__builtin_fmaf(x, sc, sh), 0.f); }

}  // namespace ppv
