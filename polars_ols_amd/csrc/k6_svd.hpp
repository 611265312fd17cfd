// k6_svd.hpp -- launch interface of K6 (see k6_svd.hip).
#pragma once
#include "common.hpp"
#include "fix_solvers.inl"

namespace pols {

struct K6Args {
    const void *y;
    const void *w;
    const void *x[POLS_MAX_FEATURES];
    const int64_t *offs;
    int64_t n_groups;
    const int32_t *fb_flag;  // the pass is a no-op unless *fb_flag == epoch (some group was flagged in THIS call)
    int32_t epoch;
    int32_t *status;         // groups with POLS_GROUP_FALLBACK are (re)solved, the rest skipped; no fit rows left -> POLS_GROUP_EMPTY
    void *coef;
    void *pred;
    void *resid;
    double *work;            // workers x work_stride doubles
    int64_t work_stride;     // >= (kt + 1) * max_group_rows
    double alpha;            // 0: minimum-norm least squares; > 0: ridge via d = s / (s^2 + alpha)
    double rc_factor;        // singular values below rc_factor * s_max are dropped; < 0: eps * max(fit rows, columns) of the group
    int32_t k_user, kt;
    const uint8_t *valid;    // null policy of the static entry (see common.hpp::null_row_in_fit)
    int32_t null_policy;
    int32_t mode;            // FixMode (fix_solvers.inl)
    int32_t small_rows;      // > 0: flagged groups of at most this many rows whose solver is the SVD belong to K6s (k6s_small.hip); K6 skips them
    int32_t small_lo;        // K6s launches only: groups of more than this many rows (one launch per team size, k6_launch)
};

// Does K6s (a sub-wave team per group, dual Jacobi) take this flagged group?  Decided from (mode, rows, columns) alone so that K6 and K6s
// agree without a hand-over: the SVD is the reference's solver for "svd" on any group and for solve_method = None on a group with no
// more rows than columns (ls.rs:224-231; rows the null policy drops only lower the fit's row count further).
__host__ __device__ inline bool k6s_takes(int mode, int64_t n, int kt, int small_rows) {
    return small_rows > 0 && n <= (int64_t)small_rows && (mode == FIX_MINNORM || (mode == FIX_OLS_AUTO && n <= (int64_t)kt));
}

// launches the K6 pool and, when the frame holds groups short enough for it, K6s
int k6_launch(pols_ctx *ctx, int dtype, const K6Args &a, int workers);
int k6s_launch(pols_ctx *ctx, int dtype, const K6Args &a);

}  // namespace pols
