// Micro-benchmark of the diagonal-block kernels of the Cholesky chain (one workgroup each): the MFMA-blocked
// k_potrf16 and the block inverse k_trtri_diag128, with wall-clock phase stamps, the residual |R^T R - A| / |A| and
// the accuracy of v_rsq_f64 with 0 / 1 / 2 Newton steps.  (Round 1's register-resident pivot-pair kernel, removed in
// round 2, measured 65.2 us per launch in the same harness; k_potrf16 32.8 us.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pybo_amd/csrc scripts/potrf_bench.hip -o scripts/potrf_bench.bin
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../pybo_amd/csrc/kernels_fit.hip"
namespace gpx {      // (kernels_fit.hip calls into the task-graph translation unit: not linked here)
bool launch_cholesky_tg(gpx_handle*) { return false; }
int tg_abort_code(gpx_handle*) { return 0; }
}

using namespace gpx;

__global__ void k_rsq_err(const double* x, int n, double* out) {
    // out[0..2]: max relative error of v_rsq_f64 raw / after one / after two Newton steps
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = x[i], ex = 1.0 / sqrt(v);
        double y = __builtin_amdgcn_rsq(v);
        e0 = fmax(e0, fabs(y - ex) / ex);
        y = fma(y, fma(-0.5 * v * y, y, 0.5), y);
        e1 = fmax(e1, fabs(y - ex) / ex);
        y = fma(y, fma(-0.5 * v * y, y, 0.5), y);
        e2 = fmax(e2, fabs(y - ex) / ex);
    }
    atomicMax((unsigned long long*)&out[0], __double_as_longlong(e0));
    atomicMax((unsigned long long*)&out[1], __double_as_longlong(e1));
    atomicMax((unsigned long long*)&out[2], __double_as_longlong(e2));
}

int main() {
    printf("built with GPX_PF_NR = %d\n", GPX_PF_NR);
    {
        const int n = 1 << 16;
        std::vector<double> x(n);
        unsigned q = 777;
        for (auto& v : x) { q = q * 1664525u + 1013904223u; v = exp(((q >> 8) & 0xffffff) / 16777216.0 * 40.0 - 20.0); }
        double *dx, *de;
        hipMalloc(&dx, n * 8); hipMalloc(&de, 64); hipMemset(de, 0, 64);
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_rsq_err, dim3(1), dim3(256), 0, 0, dx, n, de);
        double e[3];
        hipMemcpy(e, de, 24, hipMemcpyDeviceToHost);
        printf("v_rsq_f64 max relative error vs 1/sqrt (fp64 library): raw %.3g, +1 Newton %.3g, +2 Newton %.3g\n", e[0], e[1], e[2]);
    }
    const int64_t Np = 1024;             // block 3 of an 8-block matrix (strides as in a real fit)
    const int p = 3;
    std::vector<double> A((size_t)Np * Np, 0.0);
    // SPD diagonal block: A = B B^T / 128 + I
    std::vector<double> B(128 * 128);
    unsigned s = 12345;
    for (auto& b : B) { s = s * 1664525u + 1013904223u; b = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
            double acc = (i == j) ? 1.0 : 0.0;
            for (int k = 0; k < 128; ++k) acc += B[i * 128 + k] * B[j * 128 + k] / 128.0;
            A[(size_t)(p * 128 + i) * Np + p * 128 + j] = acc;
        }
    double *dS, *dR, *dT, *dU;
    int* dflag;
    long long* ddbg;
    size_t bytes = (size_t)Np * Np * 8;
    hipMalloc(&dS, bytes); hipMalloc(&dR, bytes); hipMalloc(&dT, bytes); hipMalloc(&dU, bytes);
    hipMalloc(&dflag, 64); hipMalloc(&ddbg, 64 * 8);
    hipMemset(dflag, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    auto resid = [&](const char* name) {
        std::vector<double> R((size_t)Np * Np), T((size_t)Np * Np);
        hipMemcpy(R.data(), dR, bytes, hipMemcpyDeviceToHost);
        hipMemcpy(T.data(), dT, bytes, hipMemcpyDeviceToHost);
        double num = 0, den = 0, inv = 0;
        for (int i = 0; i < 128; ++i)
            for (int j = 0; j < 128; ++j) {
                double acc = 0, ti = 0;
                for (int k = 0; k < 128; ++k) {
                    acc += R[(size_t)(p * 128 + k) * Np + p * 128 + i] * R[(size_t)(p * 128 + k) * Np + p * 128 + j];
                    ti += T[(size_t)(p * 128 + i) * Np + p * 128 + k] * R[(size_t)(p * 128 + j) * Np + p * 128 + k];   // T R^T
                }
                const double a = A[(size_t)(p * 128 + i) * Np + p * 128 + j];
                num += (acc - a) * (acc - a); den += a * a;
                inv += (ti - (i == j)) * (ti - (i == j));
            }
        printf("%-28s |R^T R - A|/|A| = %.2e   |T R^T - I|_F = %.2e\n", name, sqrt(num / den), sqrt(inv));
    };
    const int reps = 200;
    for (int variant = 1; variant < 3; ++variant) {
        hipMemcpy(dS, A.data(), bytes, hipMemcpyHostToDevice);
        hipMemset(dR, 0, bytes); hipMemset(dT, 0, bytes); hipMemset(dU, 0, bytes);
        if (variant == 2)       // the block inverse completes what k_potrf16 leaves behind
            hipLaunchKernelGGL(k_potrf16<false>, dim3(1), dim3(256), 0, 0, dS, dR, dT, dU, Np, p, dflag, (long long*)nullptr, (int64_t)0);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int r = 0; r < reps; ++r) {
            if (variant == 1) hipLaunchKernelGGL(k_potrf16<true>, dim3(1), dim3(256), 0, 0, dS, dR, dT, dU, Np, p, dflag, ddbg, (int64_t)0);
            if (variant == 2) hipLaunchKernelGGL(k_trtri_diag128, dim3(1), dim3(256), 0, 0, dR, dT, dU, Np, p, dflag);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const char* names[] = {"", "k_potrf16", "k_trtri_diag128"};
        printf("%-28s %.2f us per launch (back to back, %d launches)\n", names[variant], ms * 1e3 / reps, reps);
        if (variant == 1) {
            long long st[16];
            hipMemcpy(st, ddbg, sizeof st, hipMemcpyDeviceToHost);
            printf("  k_potrf16 phases (us, wall clock %d kHz): load %.2f |", rate, (st[1] - st[0]) * 1e3 / rate);
            for (int i = 2; i <= 9; ++i) printf(" step%d %.2f", i - 2, (st[i] - st[i - 1]) * 1e3 / rate);
            printf(" | store %.2f | total %.2f\n", (st[10] - st[9]) * 1e3 / rate, (st[10] - st[0]) * 1e3 / rate);
        }
        resid(names[variant]);
    }
    return 0;
}
