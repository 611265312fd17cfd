// k8_wide.hpp -- K8: the static models for 32 .. 1024 columns (see k8_wide.hip).
#pragma once
#include "common.hpp"

namespace pols {

constexpr int K8_KMAX = 1024;      // features incl. intercept

struct WideArgs {
    const void *const *cols;   // DEVICE table of k_user feature column pointers
    const void *y;
    const void *w;             // sample weights or nullptr
    const int64_t *offs;
    int64_t n_groups, n_rows;
    int32_t k_user, kt;        // kt = k_user + intercept; NZ = kt + 1 (the target is the last column of Z)
    double *gram;              // n_groups x NZ x NZ, row-major
    double *partial;           // splits x n_groups x NZ x NZ
    int32_t splits;
    int64_t rows_per_split;    // multiple of 64
    // solve
    double alpha, l1_ratio, tol, pivot_tol, rc_factor;
    int64_t max_iter;
    int32_t positive, active_set;
    int32_t *status;
    int32_t *fb_flag;
    int32_t epoch;
    void *coef;                // n_groups x kt, batch dtype, or nullptr
    double *coef64;            // n_groups x kt
    double *work;              // min-norm work area: workers x work_stride doubles, each [W: work_w_elems][V, s^2, g]
    int64_t work_stride;
    int64_t work_w_elems;      // (kt + 1) * max_group_rows
    void *pred, *resid;
    // multi-target (solve_multi_target, ls.rs:243-260): m targets share the Gram matrix and ONE factorisation.  Z gets m
    // target columns (NZ = kt + m); coef / coef64 are n_groups x m x kt.  Single-target calls leave these zero.
    const void *const *ycols;  // DEVICE table of m target column pointers (nullptr: the single target `y`)
    void *const *pred_cols;    // DEVICE table of m prediction column pointers (multi-target predict)
    int32_t n_targets;         // 0 or 1: single target
    // null policy (src/expressions.rs:201-296): `rowmask` (one byte per row, 1 = the row takes part in the fit) and `nfit`
    // (fit rows per group) are produced by wide_rowmask_launch from the NaNs of ALL columns (+ the validity bytes); the
    // Gram / min-norm passes give masked rows weight 0 and zero-fill the nulls that stay, the prediction pass zero-fills
    // and, for "drop", masks.  Both pointers are nullptr under "ignore".
    const uint8_t *valid;
    uint8_t *rowmask;
    double *nfit;
    int32_t null_policy;
    int32_t fix_mode;          // FixMode (fix_solvers.inl): the solver the fix-up pass runs on the groups wide_chol flags
};

__host__ __device__ inline int wide_m(const WideArgs &a) { return a.n_targets > 1 ? a.n_targets : 1; }

int wide_rowmask_launch(pols_ctx *ctx, int dtype, const WideArgs &a);   // null policies only: fills a.rowmask / a.nfit
int wide_gram_launch(pols_ctx *ctx, int dtype, const WideArgs &a);
int wide_chol_launch(pols_ctx *ctx, int dtype, const WideArgs &a);      // OLS / ridge; flags what it cannot factor
int wide_cd_launch(pols_ctx *ctx, int dtype, const WideArgs &a);        // elastic net / lasso / non-negative
int wide_minnorm_launch(pols_ctx *ctx, int dtype, const WideArgs &a, int workers);   // flagged groups: Jacobi SVD, minimum norm
int wide_predict_launch(pols_ctx *ctx, int dtype, const WideArgs &a);
// mode="statistics" for 32 .. 127 columns (src/statistics.rs:15-156): needs a.gram (wide_gram_launch), a.coef64 (the dispatcher's
// coefficients) and six f64 output arrays (any may be nullptr); lambda = kwargs.alpha
struct WideStatsOut { double *r2, *mae, *mse, *se, *tv, *pv; double lambda; double *work = nullptr; bool factored = false; };   // work: set by the launcher; factored: see wide_stats_kernel
int wide_stats_launch(pols_ctx *ctx, int dtype, const WideArgs &a, const WideStatsOut &o);
constexpr int K8_STATS_LDS_KMAX = 127;   // statistics: the k x k inverse fits dynamic LDS up to here, beyond it lives in HBM / L2
// `predict` plugin body for wide frames: one coefficient row per input row (coef_rows: n_rows x kt, batch dtype)
int wide_predict_rows_launch(pols_ctx *ctx, int dtype, const WideArgs &a, const void *coef_rows);

}  // namespace pols
