/*
 * pols_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the solve path of azmyrajab/polars_ols
 * (src/least_squares.rs) plus the marshalling / dispatch / prediction pieces of
 * src/expressions.rs and the Python pre-processing of
 * polars_ols/least_squares.py that sit either side of it.  Every function cites
 * the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (polars_ols_amd + libpols_mi355x.so) never
 * links, imports or calls it.
 *
 * PINNING STATUS: the reference is Rust + Polars; neither toolchain exists in
 * the build image, so the reference itself cannot be run or imported here
 * ("oracle/_ref" is unbuildable: needs cargo, ~330 crates, Intel MKL).  The
 * oracle is pinned against (a) every literal known-answer output the reference
 * prints in its README (README.md:50-55,72-76,104,112-113,133-137,162-164) and
 * the literal Woodbury case of src/lib.rs:124-143, and (b) the same third-party
 * numerics the reference's own test-suite uses as ITS oracle (numpy lstsq /
 * solve, sklearn ElasticNet / Ridge; tests/test_ols.py) on the reference's own
 * seeded fixture generator (_make_data, tests/test_ols.py:22-51).  See
 * tests/golden/make_golden.py and tests/test_oracle_golden.py.
 *
 * All arithmetic is f64 (the reference casts every input to Float64:
 * src/expressions.rs:33,47,80).  Matrices are ROW-MAJOR n x k like the ndarray
 * built in src/expressions.rs:26.
 */
#ifndef POLS_ORACLE_H
#define POLS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SolveMethod, src/least_squares.rs:41-65 (0 = None / not given). */
enum { ORC_METHOD_NONE = 0, ORC_METHOD_QR = 1, ORC_METHOD_SVD = 2, ORC_METHOD_CHOL = 3,
       ORC_METHOD_LU = 4, ORC_METHOD_CD = 5, ORC_METHOD_CD_ACTIVE_SET = 6 };

/* NullPolicy, src/least_squares.rs:67-91. */
enum { ORC_NULL_IGNORE = 0, ORC_NULL_ZERO = 1, ORC_NULL_DROP = 2, ORC_NULL_DROP_ZERO = 3,
       ORC_NULL_DROP_Y_ZERO_X = 4, ORC_NULL_DROP_WINDOW = 5 };

/* OLSKwargs, src/expressions.rs:298-308 with the Python defaults of
 * polars_ols/least_squares.py:101-107.  has_* == 0 means Option::None. */
typedef struct {
    double alpha;        /* default 0.0 */
    double l1_ratio;     /* valid iff has_l1_ratio */
    int32_t has_l1_ratio;
    int64_t max_iter;    /* default 1000 */
    double tol;          /* default 1e-5 */
    int32_t positive;    /* default 0 */
    int32_t solve_method;/* ORC_METHOD_* */
    double rcond;        /* valid iff has_rcond */
    int32_t has_rcond;
} orc_ols_params;

/* ---- dense helpers (src/least_squares.rs:20-39, 264-337, 600-666) ---- */
int orc_cholesky_solve(const double *a, int k, const double *b, double *x);          /* 0 ok, 1 not PD */
int orc_lu_solve(const double *a, int k, const double *b, double *x);                /* :264-273 */
int orc_inv(const double *a, int k, int use_cholesky, double *out);                  /* :20-39 */
void orc_outer_product(const double *u, const double *v, int k, double *out);        /* :600-607 */
void orc_woodbury_update(const double *a_inv, const double *u, const double *c, const double *v,
                         int k, int r, int c_is_diag, double *out);                  /* :629-648 */
void orc_update_xtx_inv(const double *xtx_inv, const double *x_update, const double *c_or_null,
                        int k, int r, double *out);                                  /* :651-666 */

/* ---- static solvers ---- */
void orc_solve_ols_qr(const double *y, const double *x, int64_t n, int k, double *beta);   /* :195-205 */
void orc_solve_ols_svd(const double *y, const double *x, int64_t n, int k, int m_targets,
                       double *beta);                                                      /* :183-191 */
void orc_solve_ridge_svd(const double *y, const double *x, int64_t n, int k, int m_targets,
                         double alpha, int has_rcond, double rcond, double *beta);         /* :106-168 */
int orc_solve_ols(const double *y, const double *x, int64_t n, int k, int method,
                  double *beta);                                                           /* :211-240 */
int orc_solve_normal_equations(const double *xtx, const double *xty, int k, int method,
                               int fallback, double *beta);                                /* :277-337 */
int orc_solve_ridge(const double *y, const double *x, int64_t n, int k, double alpha, int method,
                    int has_rcond, double rcond, double *beta);                            /* :342-371 */
int orc_solve_elastic_net(const double *y, const double *x, int64_t n, int k, double alpha,
                          int has_l1_ratio, double l1_ratio, int64_t max_iter, double tol,
                          int positive, int method, double *w, int64_t *n_iter_out);       /* :386-492 */
void orc_solve_multi_target(const double *y, const double *x, int64_t n, int k, int m,
                            double alpha, int has_rcond, double rcond, double *beta);      /* :243-260 */

/* _get_least_squares_coefficients, src/expressions.rs:351-388.  Returns 0 ok,
 * <0 for the reference's panic cases (unsupported method / bad alpha). */
int orc_get_coefficients(const double *y, const double *x, int64_t n, int k,
                         const orc_ols_params *p, double *beta);

/* ---- dynamic solvers ---- */
/* solve_recursive_least_squares, :568-598.  coef_out is n x k row-major. */
void orc_solve_rls(const double *y, const double *x, int64_t n, int k, int has_half_life,
                   double half_life, double initial_state_covariance,
                   const double *initial_state_mean_or_null, const uint8_t *is_valid,
                   double *coef_out);
/* solve_rolling_ols, :848-1032.  min_periods < 0 => None; use_woodbury < 0 => None. */
void orc_solve_rolling_ols(const double *y, const double *x, int64_t n, int k, int64_t window_size,
                           int64_t min_periods, int use_woodbury, double alpha,
                           const uint8_t *is_valid, int null_policy, double *coef_out);

/* ---- marshalling + predictions (src/expressions.rs:22-63, 175-195) ---- */
void orc_construct_features(const double *const *cols, int64_t n, int k, double *x_rowmajor);
void orc_predict_static(const double *x, const double *beta, int64_t n, int k, double *pred);
void orc_predict_dynamic(const double *x, const double *coef, int64_t n, int k, double *pred);

/* ---- batched "grouped frame" driver -------------------------------------
 * What Polars + the plugin do for
 *   pl.col(y).least_squares.<model>(x1..xk, sample_weights=w, add_intercept=..,
 *                                   mode=..).over(group)
 * with rows already sorted by group (group_offsets[G+1]) and null_policy
 * "ignore" (no nulls).  Follows polars_ols/least_squares.py:163-239 (sqrt(w)
 * scaling of y and every feature, intercept appended LAST, predictions
 * *= 1/sqrt(w), residuals = ORIGINAL target - predictions) around
 * src/expressions.rs:390-446.
 *   coef_out : G x kt (kt = k + add_intercept) or NULL
 *   pred_out : N or NULL, resid_out : N or NULL
 *   marshal  : 1 = time/perform the column->row-major copy per group like
 *              expressions.rs:22-63 (always needed for correctness; the flag
 *              only exists so the CPU baseline can report both variants --
 *              with 0 the caller passes x_rowmajor_or_null pre-marshalled).
 *   n_threads: OpenMP threads over groups (Polars' rayon pool analogue).
 * Returns 0 or a negative panic code. */
int orc_batched_least_squares(const double *y, const double *const *x_cols,
                              const double *weights_or_null, int64_t n_rows, int k,
                              const int64_t *group_offsets, int64_t n_groups, int add_intercept,
                              const orc_ols_params *p, double *coef_out, double *pred_out,
                              double *resid_out, int n_threads);

int orc_batched_rls(const double *y, const double *const *x_cols, int64_t n_rows, int k,
                    const int64_t *group_offsets, int64_t n_groups, int has_half_life,
                    double half_life, double initial_state_covariance,
                    const double *initial_state_mean_or_null, const uint8_t *is_valid_or_null,
                    double *coef_out, double *pred_out, int n_threads);

int orc_batched_rolling(const double *y, const double *const *x_cols, int64_t n_rows, int k,
                        const int64_t *group_offsets, int64_t n_groups, int64_t window_size,
                        int64_t min_periods, int use_woodbury, double alpha, int null_policy,
                        const uint8_t *is_valid_or_null, double *coef_out, double *pred_out,
                        int n_threads);

/* statistics side-car, src/statistics.rs:15-156 (mode="statistics"). */
typedef struct { double r2, mae, mse; } orc_residual_metrics;
void orc_residual_metrics_compute(const double *y, const double *pred, int64_t n,
                                  orc_residual_metrics *out);
/* returns 0 ok, 1 if the Cholesky inverse failed (all outputs NaN, :101-111). */
int orc_feature_metrics(const double *x, const double *y, int64_t n, int k, double lambda,
                        double *std_err, double *t_values, double *p_values);
double orc_student_t_two_sided_p(double t, double df);

int orc_max_threads(void);

/* ---- CPU baseline harness (bench.py) ---- */
int orc_batched_least_squares_static(const double *y, const double *const *x_cols, const double *weights, int64_t n_rows, int k,
                                     const int64_t *group_offsets, int64_t n_groups, int add_intercept, const orc_ols_params *p,
                                     double *coef_out, double *pred_out, int n_threads);
/* wall seconds of `passes` repetitions, buffers pre-allocated and first-touched outside the clock; solve_only = 1 excludes the
 * column -> row-major marshalling copy (src/expressions.rs:22-63) from the timed region */
double orc_bench_static(const double *y, const double *const *x_cols, const double *weights, int64_t n_rows, int k,
                        const int64_t *group_offsets, int64_t n_groups, int add_intercept, const orc_ols_params *p,
                        int solve_only, int passes, int n_threads);
/* kind 0: RLS (half_life), kind 1: rolling OLS (window, min_periods, "drop"), ONE sequence on one core */
double orc_bench_dynamic(int kind, const double *y, const double *const *x_cols, int64_t n, int k, double half_life,
                         int64_t window, int64_t min_periods, int passes);

#ifdef __cplusplus
}
#endif
#endif
