// k2_resident.hpp -- K2 "gram_mfma_resident": every static model for up to 16 columns with the group's rows held in
// registers, X read from HBM exactly ONCE whatever the solver (see k2_resident.hip).
//
// Replaces, for ALL groups in one launch:
//   solve_elastic_net          src/least_squares.rs:386-492   (cyclic coordinate descent, Gram form)
//   solve_ridge / solve_ols    src/least_squares.rs:211-240, 342-371 (normal equations: Cholesky, LU, Cholesky -> LU fallback)
//   solve_ols_lu               src/least_squares.rs:264-273   (partial-pivot LU)
//   make_predictions           src/expressions.rs:175-195, 398-405
//   sqrt(w) scaling, intercept, 1/sqrt(w) un-scaling, residuals   polars_ols/least_squares.py:184-196, 234-239
#pragma once
#include "common.hpp"

namespace pols {

constexpr int K2_KMAX = 16;   // columns incl. the intercept: X'X is ONE 16 x 16 tile on the matrix cores

enum K2Solver : int32_t {
    K2_CHOL = 0,            // Cholesky of X'X + alpha I (faer cholesky, ls.rs:288-297)
    K2_LU = 1,              // partial-pivot LU of X'X + alpha I (solve_ols_lu, ls.rs:264-273)
    K2_CD = 2,              // coordinate descent (ls.rs:422-445)
    K2_CD_ACTIVE_SET = 3    // ... with the active set (ls.rs:446-489)
};

struct K2Args {
    const void *y;
    const void *w;                       // sample weights or nullptr
    const void *x[K2_KMAX];              // user feature columns
    const int64_t *offs;                 // device, n_groups + 1
    int64_t n_groups;
    int64_t n_rows;
    void *coef;                          // n_groups x kt (batch dtype) or nullptr
    double *coef64;                      // n_groups x kt f64 or nullptr
    void *pred;                          // n_rows or nullptr
    void *resid;                         // n_rows or nullptr
    int32_t *status;                     // n_groups or nullptr
    int32_t k_user, kt;                  // kt = k_user + intercept
    int32_t solver;                      // K2Solver
    int32_t lu_fallback;                 // K2_CHOL only: a failed factorisation is retried with LU (solve_ridge, ls.rs:358-363)
    double alpha;                        // ridge penalty (K2_CHOL / K2_LU) or the elastic-net alpha (K2_CD*)
    double l1_ratio, tol;
    int64_t max_iter;
    int32_t positive;
    double pivot_tol;                    // see K1Args::pivot_tol
    int32_t *fb_flag;                    // see K1Args::fb_flag
    int32_t epoch;
    unsigned long long *dbg;             // POLS_TIMELINE=1: 8 s_memtime stamps per group (debug only)
};

// true when some variant keeps every row of the largest group resident (dtype, columns, rows)
bool k2_fits(int dtype, int kt, int64_t max_group_rows, bool offsets_aligned);
int k2_launch(pols_ctx *ctx, int dtype, const K2Args &a, int64_t max_group_rows);

}  // namespace pols
