// k2w_resident.hpp -- K2w "gram_mfma_resident, two tiles": OLS / ridge with 17 .. 31 columns, rows held in registers, X read from HBM
// exactly once (see k2w_kernel.inl).  Replaces solve_ols / solve_ridge (src/least_squares.rs:211-240, 342-364) + make_predictions
// (src/expressions.rs:175-195) for these widths whenever the largest group fits the registers of one workgroup.
#pragma once
#include "common.hpp"

namespace pols {

constexpr int K2W_KMIN = 17, K2W_KMAX = 31;   // columns incl. the intercept: [X | 1 | y] fills two 16-column MFMA tiles

struct K2wArgs {
    const void *y;
    const void *w;                       // sample weights or nullptr
    const void *x[32];                   // user feature columns; slots beyond k_user: any loadable column (the target)
    const int64_t *offs;                 // device, n_groups + 1
    int64_t n_groups;
    int64_t n_rows;
    void *coef;                          // n_groups x kt (batch dtype) or nullptr
    void *pred;                          // n_rows or nullptr
    void *resid;                         // n_rows or nullptr
    int32_t *status;                     // n_groups or nullptr
    int32_t k_user, kt;                  // kt = k_user + intercept
    double alpha;                        // ridge penalty
    double pivot_tol;                    // see K1Args::pivot_tol
    int32_t *fb_flag;                    // see K1Args::fb_flag
    int32_t epoch;
    unsigned long long *dbg;             // POLS_TIMELINE=1: 8 s_memtime stamps per group (debug only)
};

// true when a variant keeps every row of the largest group resident (dtype, columns, rows)
bool k2w_fits(int dtype, int kt, int64_t max_group_rows, bool offsets_aligned);
int k2w_launch(pols_ctx *ctx, int dtype, const K2wArgs &a, int64_t max_group_rows);

}  // namespace pols
