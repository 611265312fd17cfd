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
};

int k6_launch(pols_ctx *ctx, int dtype, const K6Args &a, int workers);

}  // namespace pols
