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
    aa.small_lo = 0;
    // one K6s launch per team size the frame has groups for (the offsets scan knows: 1-4, 5-8, 9-16, 17-32 rows) -- a frame of 1 000-row groups
    // with 100 000 six-row groups among them ran all of those in 32-lane teams (31 rounds per sweep over 26 zero rows): 1.83 ms, now the 8-lane form
    int top_team = 0;
    for (int b = 3; b >= 0 && top_team == 0; --b) {
        const int lo = b ? (2 << b) : 0;
        if (((ctx->offs_small_mask >> b) & 1) && (a.mode == FIX_MINNORM || (a.mode == FIX_OLS_AUTO && lo < a.kt))) top_team = 4 << b;
    }
    aa.small_rows = top_team;                                     // what the pool leaves to K6s: k6s_takes(mode, n, kt, top_team)
    if (dtype == POLS_F32) hipLaunchKernelGGL(k6_svd_kernel<float>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, aa);
    else hipLaunchKernelGGL(k6_svd_kernel<double>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, aa);
    POLS_HIP(hipGetLastError());
    bool first = true;
    for (int b = 0; b < 4 && (4 << b) <= top_team; ++b) {
        if (!((ctx->offs_small_mask >> b) & 1)) continue;
        K6Args as = aa;
        as.small_rows = 4 << b;
        as.small_lo = first ? -1 : (2 << b);        // (the first launch also owns the flagged EMPTY groups, which the pool leaves to K6s like every n <= top_team)
        first = false;
        int rc = k6s_launch(ctx, dtype, as);
        if (rc) return rc;
    }
    return POLS_OK;
}

}  // namespace pols
