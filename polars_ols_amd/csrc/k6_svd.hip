// k6_svd.hip -- K6 "small_svd": fallback solver for the groups the fused static kernels flag (POLS_GROUP_FALLBACK).
//
// Replaces, for those groups only: solve_ols_svd (src/least_squares.rs:183-191, LAPACK dgelsd: minimum-norm least
// squares, singular values below eps * s_max dropped), solve_ridge_svd (:106-168, d = s / (s^2 + alpha)) and the
// Cholesky -> LU / SVD fallback chain of solve_normal_equations (:299-327).  A group is flagged when its Cholesky
// pivot is non-positive (rank-deficient X, n <= k, NaN data) or -- OLS only, where the reference itself uses a
// backward-stable QR -- when a pivot says cond(X)^2 would eat the tolerance of the normal-equation solve.
//
// One 256-thread workgroup per flagged group (a fixed pool of workers strides over all groups and skips the
// healthy ones): the group's sqrt(w)-scaled columns are copied as f64 into a per-worker global scratch, one-sided
// Jacobi (Hestenes) rotations orthogonalise the columns -- every rotation is three block reductions over n and two
// AXPYs, the accumulated V lives in LDS -- and beta = V diag(d) U' y.  One-sided Jacobi resolves small singular
// values to high RELATIVE accuracy, which is what lets it agree with dgelsd on nearly collinear columns; an
// eigen-decomposition of X'X could not.  Not bandwidth-critical: it runs on the rare groups only.
#include "k6_body.inl"

#include <algorithm>

namespace pols {

int k6_launch(pols_ctx *ctx, int dtype, const K6Args &a, int workers) {
    if (a.kt > K6_KMAX) return fail(POLS_ERR_UNSUPPORTED, "svd fallback: %d features > %d", a.kt, K6_KMAX);
    // Short groups whose reference solver is the SVD (n <= k under solve_method = None, anything under "svd") go to K6s, a sub-wave team
    // per group; the host knows from the offsets whether the frame holds any (no extra dispatch on the frames the benchmarks visit).
    K6Args aa = a;
    aa.small_rows = 0;
    const int64_t short_rows = ctx->offs_min_rows;                // fewest rows of a non-empty group
    if (short_rows > 0 && short_rows <= 32 && (a.mode == FIX_MINNORM || (a.mode == FIX_OLS_AUTO && short_rows <= a.kt))) {
        const int64_t top = std::min<int64_t>(32, ctx->offs_max_rows);
        aa.small_rows = top <= 4 ? 4 : top <= 8 ? 8 : top <= 16 ? 16 : 32;
    }
    if (dtype == POLS_F32) hipLaunchKernelGGL(k6_svd_kernel<float>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, aa);
    else hipLaunchKernelGGL(k6_svd_kernel<double>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, aa);
    POLS_HIP(hipGetLastError());
    if (aa.small_rows > 0) return k6s_launch(ctx, dtype, aa);
    return POLS_OK;
}

}  // namespace pols
