// cabi_bench.cpp -- the hot path driven from native code through the C-ABI alone (no Python, no PyTorch): what a Rust / C++
// host such as the reference's src/expressions.rs would do.  Builds with `make -C examples` (hipcc only for the HIP runtime
// headers / libamdhip64; this file contains no device code).
//
//   ./cabi_bench [groups=10000] [rows=1000] [features=8] [steps=50]
//
// Generates BASELINE configs[1]-shaped columns on the host (x ~ N(0,1), y = sum x + 0.1 N(0,1), tests/test_ols.py:22-51), uploads
// them once, runs pols_least_squares (OLS, predictions + coefficients, f32) `steps` times on a private stream, checks group 0
// against a normal-equation solve in double precision, and prints one JSON line with the rate and the kernel-level bandwidth.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../include/pols_mi355x.h"
#include "../include/pols_mi355x_debug.h"   // pols_timing_*, pols_last_kernel_name: measurement aids

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define POLS_OK_(x) do { int r_ = (x); if (r_ != POLS_OK) { std::fprintf(stderr, "%s: %d %s\n", #x, r_, pols_last_error()); return 3; } } while (0)

int main(int argc, char **argv) {
    const int64_t G = argc > 1 ? std::atoll(argv[1]) : 10000, n = argc > 2 ? std::atoll(argv[2]) : 1000;
    const int k = argc > 3 ? std::atoi(argv[3]) : 8, steps = argc > 4 ? std::atoi(argv[4]) : 50;
    const int64_t N = G * n;
    std::mt19937_64 rng(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<std::vector<float>> cols(k, std::vector<float>(N));
    std::vector<float> y(N);
    for (int64_t i = 0; i < N; ++i) {
        float s = 0.f;
        for (int j = 0; j < k; ++j) { cols[j][i] = nd(rng); s += cols[j][i]; }
        y[i] = s + 0.1f * nd(rng);
    }
    std::vector<int64_t> offs(G + 1);
    for (int64_t g = 0; g <= G; ++g) offs[g] = g * n;

    pols_ctx *ctx = nullptr;
    POLS_OK_(pols_create(0, &ctx));
    POLS_OK_(pols_use_private_stream(ctx));
    std::vector<const void *> d_cols(k);
    float *d_y = nullptr, *d_pred = nullptr, *d_coef = nullptr;
    int32_t *d_status = nullptr;
    for (int j = 0; j < k; ++j) {
        void *p = nullptr;
        HIP_OK(hipMalloc(&p, sizeof(float) * N));
        HIP_OK(hipMemcpy(p, cols[j].data(), sizeof(float) * N, hipMemcpyHostToDevice));
        d_cols[j] = p;
    }
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_y), sizeof(float) * N));
    HIP_OK(hipMemcpy(d_y, y.data(), sizeof(float) * N, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_pred), sizeof(float) * N));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_coef), sizeof(float) * G * k));
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_status), sizeof(int32_t) * G));

    pols_batch b{};
    b.dtype = POLS_F32; b.mem = POLS_MEM_DEVICE; b.n_rows = N; b.n_groups = G; b.group_offsets = offs.data();
    b.n_features = k; b.y = d_y; b.x_cols = d_cols.data();
    pols_ols_params p;
    pols_ols_params_default(&p);
    pols_out o{};
    o.coef = d_coef; o.pred = d_pred; o.status = d_status;

    for (int i = 0; i < 10; ++i) POLS_OK_(pols_least_squares(ctx, &b, &p, &o));
    POLS_OK_(pols_synchronize(ctx));
    POLS_OK_(pols_timing_enable(ctx, 4));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) POLS_OK_(pols_least_squares(ctx, &b, &p, &o));
    POLS_OK_(pols_synchronize(ctx));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<float> ms(steps);
    const int nt = pols_timing_collect(ctx, ms.data(), steps);
    double kms = 0.0;
    for (int i = 0; i < nt; ++i) kms += ms[i];
    kms = nt > 0 ? kms / nt : NAN;

    // ---- check group 0 against the normal equations in double precision
    std::vector<float> coef0(k), pred0(n);
    std::vector<int32_t> st(G);
    HIP_OK(hipMemcpy(coef0.data(), d_coef, sizeof(float) * k, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pred0.data(), d_pred, sizeof(float) * n, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(st.data(), d_status, sizeof(int32_t) * G, hipMemcpyDeviceToHost));
    std::vector<double> A(k * k, 0.0), rhs(k, 0.0), beta(k, 0.0);
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < k; ++i) {
            rhs[i] += (double)cols[i][r] * y[r];
            for (int j = 0; j < k; ++j) A[i * k + j] += (double)cols[i][r] * cols[j][r];
        }
    for (int j = 0; j < k; ++j) {                                   // Gaussian elimination (A is SPD)
        for (int i = j + 1; i < k; ++i) {
            const double f = A[i * k + j] / A[j * k + j];
            for (int c = j; c < k; ++c) A[i * k + c] -= f * A[j * k + c];
            rhs[i] -= f * rhs[j];
        }
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = rhs[i];
        for (int c = i + 1; c < k; ++c) s -= A[i * k + c] * beta[c];
        beta[i] = s / A[i * k + i];
    }
    double max_dc = 0.0, max_dp = 0.0;
    for (int j = 0; j < k; ++j) max_dc = std::fmax(max_dc, std::fabs(coef0[j] - beta[j]));
    for (int64_t r = 0; r < n; ++r) {
        double pr = 0.0;
        for (int j = 0; j < k; ++j) pr += cols[j][r] * beta[j];
        max_dp = std::fmax(max_dp, std::fabs(pred0[r] - pr));
    }
    int64_t flagged = 0;
    for (int64_t g = 0; g < G; ++g) flagged += st[g] != 0;
    const double alg_bytes = 4.0 * n * (k + 2) * G;
    std::printf("{\"harness\": \"native C-ABI\", \"groups\": %lld, \"rows\": %lld, \"features\": %d, \"steps\": %d, "
                "\"regressions_per_s\": %.6g, \"us_per_call\": %.3f, \"kernel\": \"%s\", \"kernel_us\": %.3f, \"kernel_GBps\": %.1f, "
                "\"max_abs_dcoef_group0\": %.3g, \"max_abs_dpred_group0\": %.3g, \"groups_not_ok\": %lld}\n",
                (long long)G, (long long)n, k, steps, G * steps / sec, 1e6 * sec / steps, pols_last_kernel_name(ctx), 1e3 * kms,
                alg_bytes / (kms * 1e-3) / 1e9, max_dc, max_dp, (long long)flagged);
    pols_destroy(ctx);
    return (max_dc < 1e-4 && max_dp < 1e-4 && flagged == 0) ? 0 : 1;
}
