// Accuracy of gpx::exp_nonpos against the library exp on the device; the values are dumped so that
// scripts/exp_check.py can compare both with a 50-digit mpmath reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp_check.hip -o scripts/exp_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../pybo_amd/csrc/gpx_math.h"

__global__ void k(const double* x, double* mine, double* lib, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mine[i] = gpx::exp_nonpos(x[i]); lib[i] = exp(x[i]); }
}

int main(int argc, char** argv) {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        // a third each: [-2, 0], [-50, 0], [-760, 0] (down to the denormal / zero tail)
        x[i] = (i % 3 == 0) ? -2.0 * u : (i % 3 == 1) ? -50.0 * u : -760.0 * u;
    }
    x[0] = 0.0; x[1] = -0.0; x[2] = -745.2; x[3] = -1e300; x[4] = -708.4; x[5] = -0.34657359027997264;
    x[6] = nan("");
    double *dx, *da, *db;
    if (hipMalloc(&dx, n * 8) || hipMalloc(&da, n * 8) || hipMalloc(&db, n * 8)) return 1;
    if (hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice)) return 1;
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    if (hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost) || hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost))
        return 1;
    FILE* f = fopen(argc > 1 ? argv[1] : "exp_check.out", "wb");
    fwrite(x.data(), 8, n, f); fwrite(a.data(), 8, n, f); fwrite(b.data(), 8, n, f);
    fclose(f);
    printf("wrote %d values\n", n);
    return 0;
}
