// gpx_math.h -- the covariance functions, shared by every kernel that evaluates one (Gram build, cross-Gram,
// gradient path) so that K, K* and dK*/dx come from the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/gpx.h"

namespace gpx {

// covariance as a function of the squared scaled distance r2 = sum_k ((x_k - z_k)/ell_k)^2
__device__ __forceinline__ double kern_eval(int kid, double r2, double rho) {
    switch (kid) {
        case GPX_KERN_SE_ARD:
            return rho * exp(-0.5 * r2);
        case GPX_KERN_MATERN52: {
            const double s = 2.23606797749978969641 * sqrt(r2);
            return rho * (1.0 + s + (5.0 / 3.0) * r2) * exp(-s);
        }
        case GPX_KERN_MATERN32: {
            const double s = 1.73205080756887729353 * sqrt(r2);
            return rho * (1.0 + s) * exp(-s);
        }
        default:
            return rho * exp(-sqrt(r2));
    }
}

}  // namespace gpx
