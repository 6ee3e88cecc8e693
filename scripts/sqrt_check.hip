// gpx::sqrt_r2 against the library sqrt (correctly rounded) on 2^20 arguments spread over 1e-300 .. 1e300, plus specials.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/sqrt_check.hip -o scripts/sqrt_check.bin && scripts/sqrt_check.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../pybo_amd/csrc/gpx_math.h"

__global__ void k(const double* x, double* mine, double* lib, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mine[i] = gpx::sqrt_r2(x[i]); lib[i] = sqrt(x[i]); }
}

int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        x[i] = (i % 2 == 0) ? 50.0 * u : pow(10.0, 560.0 * u - 279.0);
    }
    x[0] = 0.0; x[1] = 1e-300; x[2] = 1.0; x[3] = 4.0; x[4] = nan(""); x[5] = 2.2250738585072014e-308; x[6] = 1e300;
    double *dx, *da, *db;
    if (hipMalloc(&dx, n * 8) || hipMalloc(&da, n * 8) || hipMalloc(&db, n * 8)) return 1;
    if (hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice)) return 1;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    if (hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost) || hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost)) return 1;
    long long differ = 0, worst = 0;
    for (int i = 7; i < n; ++i) {
        long long ia, ib;
        memcpy(&ia, &a[i], 8); memcpy(&ib, &b[i], 8);
        const long long dlt = ia > ib ? ia - ib : ib - ia;
        if (dlt) ++differ;
        if (dlt > worst) worst = dlt;
    }
    printf("sqrt_r2 vs library sqrt on %d arguments: %lld differ, worst %lld ulp\n", n - 7, differ, worst);
    printf("specials: sqrt_r2(0) = %g, (1e-300) = %g, (1) = %.17g, (4) = %.17g, (nan) = %g, (min normal) = %g, (1e300) = %.17g [lib %.17g]\n",
           a[0], a[1], a[2], a[3], a[4], a[5], a[6], b[6]);
    return worst > 1;
}
