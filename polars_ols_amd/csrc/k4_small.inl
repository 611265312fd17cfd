// k4_small.inl -- the per-row K x K solve (K <= 8, f64, in one lane's registers) shared by the lane-per-chunk dynamic kernels
// (k4_rolling.hip) and the row-parallel ones (k3c_scan.hip): solve_normal_equations(.., None, Some(LU)) of
// src/least_squares.rs:732-734 / :277-337 -- Cholesky, LU with partial pivoting when a pivot is not positive.
#pragma once
#include "k1_kernel.inl"   // tri_index, chol_solve

namespace pols {

template <int K> struct K4N { static constexpr int NX = K * (K + 1) / 2; static constexpr int N = NX + K; };

// Cholesky -> LU with partial pivoting (solve_normal_equations(.., None, Some(LU)), :732-734 / :277-337)
template <int K>
__device__ __noinline__ void lu_solve_small(const double *A, const double *b, double *x) {
    double m[K][K], r[K];
    for (int i = 0; i < K; ++i) { r[i] = b[i]; for (int j = 0; j < K; ++j) m[i][j] = A[i * K + j]; }
    for (int j = 0; j < K; ++j) {
        int p = j; double best = fabs(m[j][j]);
        for (int i = j + 1; i < K; ++i) if (fabs(m[i][j]) > best) { best = fabs(m[i][j]); p = i; }
        if (p != j) { for (int c = 0; c < K; ++c) { const double t = m[j][c]; m[j][c] = m[p][c]; m[p][c] = t; } const double t = r[j]; r[j] = r[p]; r[p] = t; }
        const double d = m[j][j];
        for (int i = j + 1; i < K; ++i) {
            const double f = m[i][j] / d;
            for (int c = j + 1; c < K; ++c) m[i][c] -= f * m[j][c];
            r[i] -= f * r[j];
        }
    }
    for (int i = K - 1; i >= 0; --i) {
        double s = r[i];
        for (int c = i + 1; c < K; ++c) s -= m[i][c] * x[c];
        x[i] = s / m[i][i];
    }
}

template <int K>
__device__ __forceinline__ void solve_state(const double (&S)[K4N<K>::N], double alpha, double (&beta)[K]) {
    constexpr int NZ = K + 1;
    double acc[(K + 1) * (K + 2) / 2];
#pragma unroll
    for (int p = 0; p < K; ++p) {
#pragma unroll
        for (int q = p; q < K; ++q) acc[tri_index<NZ>(p, q)] = S[tri_index<K>(p, q)];
        acc[tri_index<NZ>(p, K)] = S[K4N<K>::NX + p];
    }
    acc[tri_index<NZ>(K, K)] = 0.0;
    if (!chol_solve<double, K>(acc, alpha, beta)) {
        double A[K * K], b[K], x[K];
        for (int p = 0; p < K; ++p) {
            for (int q = 0; q < K; ++q) A[p * K + q] = S[p <= q ? tri_index<K>(p, q) : tri_index<K>(q, p)] + (p == q ? alpha : 0.0);
            b[p] = S[K4N<K>::NX + p];
        }
        lu_solve_small<K>(A, b, x);
        for (int p = 0; p < K; ++p) beta[p] = x[p];
    }
}

// DPP move with bound_ctrl (lanes whose source lies outside the row / wave read 0) and every row enabled: unlike common.hpp's dpp_get
// the destination needs no `v_mov_b32 v, 0` in front of every v_mov_b32_dpp -- 2 of the 5 instructions per f64 component and scan step.
// For scans whose combine step is predicated per lane anyway (the segmented scans of k3c_scan.hip / k4c_rolling.hip).
template <int CTRL>
__device__ __forceinline__ double dpp_get0(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 1 / d to f64 accuracy without the IEEE division sequence: v_rcp_f64 + two Newton steps (the solves below run once per ROW)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

// The same solve as a square-root-free L D L' factorisation: K reciprocals instead of K (sqrt + division) pairs -- about a third of
// chol_solve's instructions at K = 6 -- keeping only L and 1 / d in registers.  (S + alpha I) beta = b with S packed upper, b = S[NX ..].
// CLAMP: a pivot below eps x its diagonal entry (non-positive ones included) is replaced by that (no LU, no call, no scratch) -- for callers whose matrix is
// positive definite in exact arithmetic, where such a pivot is rounding noise.  Otherwise `ok` comes back false and beta is unusable.
template <int K, bool CLAMP, int LEN>
__device__ __forceinline__ bool ldl_solve_small(const double (&S)[LEN], double alpha, double (&beta)[K]) {
    static_assert(LEN >= K4N<K>::N, "packed upper triangle + right-hand side");
    double L[K][K], W[K][K], dinv[K];                         // W[i][j] = L[i][j] d_j: the unscaled column entries, kept so that row j's
    bool ok = true;                                           // u[p] = L[j][p] d_p below costs nothing
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const double ajj = S[tri_index<K>(j, j)] + alpha;
        double d = ajj;
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-W[j][p], L[j][p], d);
        if constexpr (CLAMP) d = fmax(d, 0x1p-52 * ajj);      // (one v_max_f64; a pivot below eps x its diagonal entry is noise either way)
        else ok = ok && (d > 0.0);
        dinv[j] = fast_rcp(d);
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            double s = S[tri_index<K>(j, i)];
#pragma unroll
            for (int p = 0; p < j; ++p) s = fma(-W[j][p], L[i][p], s);
            W[i][j] = s;
            L[i][j] = s * dinv[j];
        }
    }
    double t[K];
#pragma unroll
    for (int i = 0; i < K; ++i) {                             // L z = b
        double s = S[K4N<K>::NX + i];
#pragma unroll
        for (int p = 0; p < i; ++p) s = fma(-L[i][p], t[p], s);
        t[i] = s;
    }
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {                        // L' beta = D^-1 z
        double s = t[i] * dinv[i];
#pragma unroll
        for (int p = i + 1; p < K; ++p) s = fma(-L[p][i], beta[p], s);
        beta[i] = s;
    }
    return ok;
}

}  // namespace pols
