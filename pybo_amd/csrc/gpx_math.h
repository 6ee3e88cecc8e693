// gpx_math.h -- the covariance functions, shared by every kernel that evaluates one (Gram build, cross-Gram,
// gradient path) so that K, K* and dK*/dx come from the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gpx.h"

namespace gpx {

// exp(x) for x <= 0 -- the only arguments a stationary covariance produces.  Cody-Waite reduction
// x = k ln2 + r (|r| <= ln2/2; ln2 split so that k*ln2_hi is exact), Taylor polynomial of degree 13 in Horner
// form (remainder r^14/14! < 4e-18 relative), scaling by v_ldexp_f64 (which also produces the denormal /
// zero tail); NaN propagates.  Measured against a 50-digit reference: <= 0.79 ulp, the same as the library's.
// Why not the library exp: the cross-Gram is VALU-issue-bound (PMC: SQ_INSTS_VALU x 4 cycles = its run time,
// 73 instructions per K* entry) and two thirds of the library call's instruction slots were v_mov_b32 pairs
// re-materialising its 64-bit literals in VGPRs for every evaluation.  The coefficients here live in
// constant memory, so they arrive by scalar loads and feed v_fma_f64 directly as its one SGPR operand.
__constant__ double kExpC[16] = {
    1.60590438368216133e-10, 2.08767569878681002e-09, 2.50521083854417202e-08, 2.75573192239858883e-07,
    2.75573192239858925e-06, 2.48015873015873016e-05, 1.98412698412698413e-04, 1.38888888888888894e-03,
    8.33333333333333322e-03, 4.16666666666666644e-02, 1.66666666666666657e-01, 0.5,
    1.44269504088896340736,           // [12] 1/ln2
    -6.93147180369123816490e-01,      // [13] -ln2_hi (21 trailing zero bits)
    -1.90821492927058770002e-10,      // [14] -ln2_lo
    -746.0};                          // [15] below this exp() is 0 in fp64
// 1.5 * 2^52: fma(x, 1/ln2, kExpMagic) is that constant plus the integer nearest x / ln2 (one rounding), the integer
// itself sits in the low mantissa bits -- the low dword IS k in two's complement for |k| < 2^31 -- and subtracting the
// constant again returns it as a double: v_rndne_f64 and v_cvt_i32_f64 (0.7 of an FMA's issue rate) are not needed.
__constant__ double kExpMagic = 6755399441055744.0;

__device__ __forceinline__ double exp_nonpos(double x) {
    x = (x < kExpC[15]) ? kExpC[15] : x;                        // NaN passes through
    const double t = fma(x, kExpC[12], kExpMagic);
    const double k = t - kExpMagic;
    double r = fma(k, kExpC[13], x);
    r = fma(k, kExpC[14], r);
    double p = kExpC[0];
#pragma unroll
    for (int i = 1; i < 12; ++i) p = fma(p, r, kExpC[i]);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, __double2loint(t));                         // (t NaN: p is NaN too)
}

// sqrt of a squared distance: x >= 0 (a sum of squares) or NaN.  v_rsq_f64 seed (5e-8) + one coupled Goldschmidt step +
// two residual corrections -- the library's own iteration without its range scaling (4 instructions shorter): arguments
// below 1e-280 (distances below 1e-140 length scales) give 0, which is what their covariance rounds to anyway.
__device__ __forceinline__ double sqrt_r2(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, x);
    g = fma(e, h, g);
    e = fma(-g, g, x);
    g = fma(e, h, g);
    // 0 -> 0, NaN -> NaN; +inf (an overflowed scaled distance: very small length scales) -> +inf like the library's sqrt
    // -- rsq(inf) = 0 makes g = inf * 0 = NaN, which would put NaN instead of 0 into every Matern covariance
    return (x > 1.0e-280) ? ((x < 1.0e300) ? g : x) : x * 0.0;
}

// covariance as a function of the squared scaled distance r2 = sum_k ((x_k - z_k)/ell_k)^2
__device__ __forceinline__ double kern_eval(int kid, double r2, double rho) {
    switch (kid) {
        case GPX_KERN_SE_ARD:
            return rho * exp_nonpos(-0.5 * r2);
        case GPX_KERN_MATERN52: {
            const double s = 2.23606797749978969641 * sqrt_r2(r2);
            return rho * (1.0 + s + (5.0 / 3.0) * r2) * exp_nonpos(-s);
        }
        case GPX_KERN_MATERN32: {
            const double s = 1.73205080756887729353 * sqrt_r2(r2);
            return rho * (1.0 + s) * exp_nonpos(-s);
        }
        default:
            return rho * exp_nonpos(-sqrt_r2(r2));
    }
}

}  // namespace gpx
