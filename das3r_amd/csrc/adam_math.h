// adam_math.h — the one Adam update every kernel of the library applies (adam.hip, pretransform.hip): torch's single-tensor arithmetic,
// m += (g - m)(1 - b1); v = v b2 + (1 - b2) g g; p -= step_size m / (sqrt(v) / sqrt(bc2) + eps) with step_size = lr / bc1.
#pragma once

namespace das3r {

__device__ __forceinline__ void adam_update(float &p, float &m, float &v, const float g, const float beta1, const float beta2, const float eps,
                                            const float step_size, const float bc2_sqrt) {
    // (explicit roundings: every kernel that inlines this takes the same step bit for bit, whatever the compiler would contract around it)
    m = __fmaf_rn(g - m, 1.0f - beta1, m);
    v = __fmaf_rn(__fmul_rn(1.0f - beta2, g), g, __fmul_rn(v, beta2));
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
    p = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p);
}

}  // namespace das3r
