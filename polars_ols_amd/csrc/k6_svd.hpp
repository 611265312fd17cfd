// k6_svd.hpp -- launch interface of K6 (see k6_svd.hip).
#pragma once
#include "common.hpp"

namespace pols {

// What a flagged group is re-solved with: the solver the REFERENCE runs for the call's (branch, solve_method) on such a group.
enum K6Mode : int32_t {
    K6_MINNORM = 0,    // solve_ols_svd / solve_ridge_svd (ls.rs:106-191): one-sided Jacobi, singular values below rc_factor * s_max dropped
    K6_OLS_AUTO = 1,   // solve_ols with solve_method = None (ls.rs:224-231): pivoted QR when the fit has more rows than columns, else SVD
    K6_OLS_QR = 2,     // solve_ols_qr (ls.rs:195-205): column-pivoted Householder QR, BASIC solution on rank-deficient X (dependent
                       //   columns -> 0; notebooks/polars_ols_demo.ipynb cell 28 prints {1.0, 2.0, -0.0} where "svd" prints {1, 1, 1})
    K6_CHOL_LU = 3,    // solve_ridge None / "chol" (ls.rs:352-363): Cholesky of X'X + alpha I in f64, on failure LU with partial pivoting --
                       //   an exactly singular matrix gives NaN (notebook cell 30), no cut-off of any kind
    K6_LU = 4,         // solve_ridge "lu" (ls.rs:330-333): LU with partial pivoting
};

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
    double rc_factor;        // singular values below rc_factor * s_max are dropped
    int32_t k_user, kt;
    const uint8_t *valid;    // null policy of the static entry (see common.hpp::null_row_in_fit)
    int32_t null_policy;
    int32_t mode;            // K6Mode
};

int k6_launch(pols_ctx *ctx, int dtype, const K6Args &a, int workers);

}  // namespace pols
