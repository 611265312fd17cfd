/*
 * pols_mi355x.h -- C-ABI of libpols_mi355x.so, the MI355X (gfx950) batched
 * least-squares engine that replaces the solve path of azmyrajab/polars_ols.
 *
 * The boundary sits at the seam between the reference's plugin layer
 * (src/expressions.rs) and its solver layer (src/least_squares.rs), i.e. at the
 * `use crate::least_squares::{...}` import of src/expressions.rs:15-18.  Each
 * entry point below names the reference functions it replaces; INTEGRATION.md
 * shows the `extern "C"` block a maintainer of the reference would add to
 * src/expressions.rs to bind them.
 *
 * Conventions
 *  - plain C: pointers + sizes, no C++/torch types, no exceptions or panics cross
 *    the boundary.  Every entry returns POLS_OK (0) or a negative pols_error;
 *    pols_last_error() returns a thread-local message for the last failure.
 *    The reference's `panic!/assert!` cases (src/least_squares.rs:231,335,349,
 *    366,404-413) map to POLS_ERR_PANIC with the reference's message.
 *  - the caller owns every input / output buffer; the library owns only device
 *    scratch and the HIP stream inside a pols_ctx.  A pols_ctx may be used by
 *    one thread at a time; create one per thread (Polars calls plugins from a
 *    rayon pool, README.md:19).
 *  - data layout: struct-of-arrays, exactly what Polars hands a plugin -- one
 *    contiguous buffer per column (src/expressions.rs:22-63 is the copy into a
 *    row-major matrix that this library deletes).  Rows of one group are
 *    contiguous: group g owns rows [group_offsets[g], group_offsets[g+1]).
 *    A single un-grouped call (what the reference plugin receives per group) is
 *    n_groups = 1, group_offsets = {0, n}.
 *  - `mem` says where ALL data pointers of a batch / out live (HOST: the library
 *    stages through its own device scratch, PCIe-inclusive; DEVICE: zero-copy,
 *    asynchronous on the context's stream).  group_offsets and x_cols (the array
 *    of column pointers itself) are always HOST arrays.
 *  - NaN marks undefined rows of rolling outputs, like src/least_squares.rs:864.
 *  - there is NO CPU fallback: if no gfx950 device is usable every compute entry
 *    fails with POLS_ERR_NO_DEVICE.
 */
#ifndef POLS_MI355X_H
#define POLS_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POLS_MAX_FEATURES 32        /* size of the fixed kernel-argument column arrays; wider calls use device pointer tables */
#define POLS_MAX_FEATURES_STATISTICS 1024 /* pols_least_squares_statistics: features incl. the intercept column */
#define POLS_MAX_FEATURES_DYNAMIC 1024 /* pols_recursive_least_squares / pols_rolling_least_squares (the reference's README
                                          benchmark runs them at 100 features; beyond 128 the k x k state of a chunk lives in
                                          HBM: correct, but every row costs O(k^2) L2 traffic) */
#define POLS_MAX_FEATURES_STATIC 1024 /* pols_least_squares / pols_predict: the reference's own wide cases (tests/benchmark.py
                                         at 100 features, test_elastic_net and test_fit_wide up to 1 000) */

typedef enum {
    POLS_OK = 0,
    POLS_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unsorted offsets ...) */
    POLS_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for (message says which) */
    POLS_ERR_HIP = -3,         /* HIP runtime error (message carries hipGetErrorString) */
    POLS_ERR_PANIC = -4,       /* the reference would panic!/assert! on these arguments */
    POLS_ERR_NO_DEVICE = -5    /* no usable gfx950 device; there is no CPU fallback */
} pols_error;

typedef enum { POLS_F32 = 0, POLS_F64 = 1 } pols_dtype;
typedef enum { POLS_MEM_HOST = 0, POLS_MEM_DEVICE = 1 } pols_mem;

/* SolveMethod, src/least_squares.rs:41-65; POLS_SOLVE_AUTO == Option::None. */
typedef enum {
    POLS_SOLVE_AUTO = 0, POLS_SOLVE_QR = 1, POLS_SOLVE_SVD = 2, POLS_SOLVE_CHOL = 3,
    POLS_SOLVE_LU = 4, POLS_SOLVE_CD = 5, POLS_SOLVE_CD_ACTIVE_SET = 6
} pols_solve_method;

/* NullPolicy, src/least_squares.rs:67-91. */
typedef enum {
    POLS_NULL_IGNORE = 0, POLS_NULL_ZERO = 1, POLS_NULL_DROP = 2, POLS_NULL_DROP_ZERO = 3,
    POLS_NULL_DROP_Y_ZERO_X = 4, POLS_NULL_DROP_WINDOW = 5
} pols_null_policy;

/* Per-group status written to pols_out.status. */
typedef enum {
    POLS_GROUP_OK = 0,
    POLS_GROUP_FALLBACK = 1, /* Cholesky failed, the reference's fallback solver was taken (ls.rs:299-327); also every group with FEWER rows
                              * than columns under solve_method None / "svd": the reference picks the SVD for it by shape (ls.rs:224-231) */
    POLS_GROUP_EMPTY = 2,    /* no rows: coefficients are zeros (src/expressions.rs:357-359) */
    POLS_GROUP_NOT_CONVERGED = 3, /* coordinate descent hit max_iter (result still returned, like the reference) */
    POLS_GROUP_BAD_DOF = 4   /* statistics only: degrees of freedom <= 0; the reference panics the whole query here
                                (src/statistics.rs:131-134), a batched launch marks the group, writes NaN standard
                                errors / t / p for it and carries on -- the host decides whether to raise */
} pols_group_status;

typedef struct pols_ctx pols_ctx;

/* ---- context ------------------------------------------------------------ */
int pols_device_count(void);
const char *pols_version(void);
const char *pols_last_error(void);
/* device_id: HIP ordinal.  Creates a private non-blocking stream. */
int pols_create(int device_id, pols_ctx **out);
void pols_destroy(pols_ctx *ctx);
/* Borrow the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream).  The handle is used as
 * given: NULL is HIP's null (legacy default) stream -- which is what torch's default stream is.
 * pols_use_private_stream() goes back to the context's own stream.  A context's scratch buffers are shared by all of its
 * calls: when the stream CHANGES the new stream is made to wait (one event) for what the old one still has in flight, so
 * calls of one context issued under different streams execute in issue order; use one context per stream for concurrency. */
int pols_set_stream(pols_ctx *ctx, void *hip_stream);
int pols_use_private_stream(pols_ctx *ctx);
int pols_synchronize(pols_ctx *ctx);
/* Tuning / diagnostic knobs (engine choices for A/B measurements, debug stamps).  `key` is the name of the matching POLS_*
 * environment variable, with or without the prefix; value NULL restores the default.  The environment is read ONCE, in
 * pols_create(); no compute entry calls getenv. */
int pols_set_option(pols_ctx *ctx, const char *key, const char *value);

/* ---- problem description ------------------------------------------------- */

/* OLSKwargs (src/expressions.rs:298-308) with the Python defaults of
 * polars_ols/least_squares.py:101-107 (see pols_ols_params_default).  has_* == 0 <=> Option::None. */
typedef struct {
    double alpha;
    double l1_ratio;
    int32_t has_l1_ratio;
    int64_t max_iter;
    double tol;
    int32_t positive;
    int32_t solve_method; /* pols_solve_method */
    double rcond;
    int32_t has_rcond;
    int32_t null_policy;  /* pols_null_policy; "ignore" is the OLS default */
} pols_ols_params;
void pols_ols_params_default(pols_ols_params *p);

/* RLSKwargs (src/expressions.rs:310-316; defaults polars_ols/least_squares.py:137-140). */
typedef struct {
    double half_life;
    int32_t has_half_life;
    double initial_state_covariance;        /* default 10.0 */
    const double *initial_state_mean;       /* HOST, n_features + add_intercept values, or NULL */
    int32_t null_policy;                    /* default "drop" */
} pols_rls_params;
void pols_rls_params_default(pols_rls_params *p);

/* RollingKwargs (src/expressions.rs:318-325; defaults polars_ols/least_squares.py:156-160). */
typedef struct {
    int64_t window_size;
    int64_t min_periods;  /* < 0 <=> None -> min(k, window) (ls.rs:860) */
    int32_t use_woodbury; /* < 0 <=> None -> k > 60 (ls.rs:863).  DIVERGENCE: only the default is reproduced -- up to 32 features the
                             window state is X'X (NonWoodburyState, ls.rs:669-735) whatever this field says; from 33 features the
                             inverse is propagated (WoodburyState, :737-787).  use_woodbury = 1 below 33 features (tested by the reference
                             at tests/test_ols.py:718-772) is accepted and ignored: same mathematics, different rounding */
    double alpha;         /* 0 <=> None */
    int32_t null_policy;  /* dataclass default "drop_window"; the namespace method passes "drop" */
} pols_rolling_params;
void pols_rolling_params_default(pols_rolling_params *p);

typedef struct {
    int32_t dtype;                /* pols_dtype of y / x / weights and of every output */
    int32_t mem;                  /* pols_mem of the data pointers below */
    int64_t n_rows;
    int64_t n_groups;
    const int64_t *group_offsets; /* HOST, n_groups + 1 ascending values, [0] == 0, [n_groups] == n_rows */
    int32_t n_features;           /* user features, excluding the intercept */
    const void *y;                /* target column, n_rows */
    const void *const *x_cols;    /* HOST array of n_features column pointers, each n_rows */
    const void *weights;          /* sample_weights column or NULL (polars_ols/least_squares.py:190-196).  A null (NaN) weight acts as
                                     the weight 1e-24 -- sqrt_w = w.sqrt().fill_null(1e-12), least_squares.py:193 -- in every entry:
                                     the fill is a device pass behind this boundary (skipped when null_free is set) */
    const uint8_t *valid;         /* optional row validity, 1 byte per row (1 = valid), or NULL = all valid */
    int32_t add_intercept;        /* append a ones column LAST, named "const" (least_squares.py:184-188) */
    uint64_t offsets_generation;  /* 0: group_offsets is content-checked on every call (hash, then memcmp against the copy the
                                     library keeps of what it last uploaded).  Non-zero: the caller PROMISES that the same
                                     (group_offsets pointer, n_groups, offsets_generation) always names the same content and bumps
                                     the value whenever it rewrites the array -- repeated calls on one frame then cost O(1) on
                                     the host instead of a pass over the offsets */
    int32_t null_free;            /* non-zero: the caller KNOWS that no target / feature / weight value is null (= NaN here) -- what a Polars
                                     / Arrow caller reads off null_count == 0 for free.  The null policy then has nothing to do:
                                     the static entry takes its policy-free kernels and the dynamic entries skip their validity
                                     scan (one pass over the columns + one stream synchronisation per call).  0 = unknown */
} pols_batch;

typedef struct {
    void *coef;      /* static models: n_groups x kt; dynamic (rls / rolling): n_rows x kt; kt = n_features + add_intercept */
    void *pred;      /* n_rows, or NULL */
    void *resid;     /* n_rows: ORIGINAL target - predictions (least_squares.py:239), or NULL */
    int32_t *status; /* n_groups pols_group_status values, or NULL */
} pols_out;

/* ---- compute entries ------------------------------------------------------ */

/* Replaces, for every group in one launch: _get_least_squares_coefficients
 * (src/expressions.rs:351-388) -> solve_ols / solve_ridge / solve_elastic_net
 * (src/least_squares.rs:211-240, 342-371, 386-492) and make_predictions
 * (src/expressions.rs:175-195), including the sqrt(w) pre-scaling, the intercept
 * column and the 1/sqrt(w) un-scaling that polars_ols/least_squares.py:163-239
 * performs around the plugin call. */
int pols_least_squares(pols_ctx *ctx, const pols_batch *b, const pols_ols_params *p, pols_out *o);

/* Replaces solve_recursive_least_squares (src/least_squares.rs:568-598) + the
 * dynamic make_predictions (src/expressions.rs:184,640-645); one sequence per group.
 * RAW columns go in (both dynamic entries): b->weights, b->add_intercept and the null policy are honoured on the device --
 * sqrt(w) scaling of target and features with a null weight acting as 1e-24 (polars_ols/least_squares.py:190-196), the ones
 * column appended LAST, compute_is_valid_mask for p->null_policy from the NaNs (= nulls) of the scaled columns
 * (src/expressions.rs:201-228; skipped when b->valid is given or b->null_free is set), nulls -> 0 (ex.rs:603, 629, 656, 683),
 * predictions masked by the validity (ex.rs:640-645, 695-700) and un-scaled by 1 / sqrt(w) (ls.py:234-235).  coef has
 * kt = n_features + add_intercept columns; p->initial_state_mean has kt values. */
int pols_recursive_least_squares(pols_ctx *ctx, const pols_batch *b, const pols_rls_params *p, pols_out *o);

/* Replaces solve_rolling_ols (src/least_squares.rs:848-1032) + dynamic make_predictions.
 * A window whose sums have no Cholesky factorisation is solved by LU with partial pivoting like the reference (ls.rs:732-734) on the
 * default routes: up to 10 features (the row-parallel tile kernel: null-free frames, the drop family with nulls through a source map,
 * "drop_window" with nulls by masking; tests/test_k4_gpu.py::test_rolling_divergence_band_is_pinned) and, since round 6, 11 to 32 features on
 * null-free frames and under the drop family (the wave-per-chunk kernel lists the rows whose sums it could not invert, a follow-up launch runs
 * the LU on them; ::test_rolling_wide_windows_without_an_inverse_take_the_lu): same kind of answer on the same rows.
 * DIVERGENCE (what is left): "drop_window" on a frame WITH nulls at 11 to 32 features -- there such a window yields NaN coefficients where the
 * reference's LU returns whatever a zero or noise pivot produces (inf / NaN / 1e15-sized numbers);
 * pols_set_option("ROLLING_ENGINE", "chunk") selects the kernels that run the LU at those widths.
 * p->use_woodbury is accepted and does not select a code path: up to 8 features (and wherever the chunk kernels run) the sums are
 * re-factored per row, from 9 features on the inverse is propagated with Sherman-Morrison updates whatever the flag says (the
 * reference's WoodburyState arithmetic, ls.rs:737-787, rebuilt from the sums every 128 rows) -- same mathematics, different rounding. */
int pols_rolling_least_squares(pols_ctx *ctx, const pols_batch *b, const pols_rolling_params *p, pols_out *o);

/* Replaces the `predict` plugin body (src/expressions.rs:706-741): row-wise sum_j x[t,j] * coef[t,j].  `coef` holds one
 * coefficient row per input row (coef_rows == n_rows, batch dtype, where `b->mem` says) -- what Polars broadcasts the
 * coefficient struct to before the plugin sees it; b->add_intercept appends the literal 1.0 feature of
 * polars_ols/least_squares.py:479-483; nulls are the caller's (the Python layer zero-fills or masks, :455-491). */
int pols_predict(pols_ctx *ctx, const pols_batch *b, const void *coef, int64_t coef_rows, void *pred_out);
/* The same with the plugin's `null_policy` kwarg (PredictKwargs, ex.rs:708): nulls are NaNs here; features are zero-filled unless the
 * policy is "ignore" (construct_features_array(.., null_policy != Ignore), :725); under "drop" the rows with a null anywhere come
 * back null (:732-738) -- with NaN as the null that is what the un-filled product already is, so POLS_NULL_DROP computes like
 * POLS_NULL_IGNORE.  pols_predict == pols_predict_policy(.., POLS_NULL_IGNORE, ..). */
int pols_predict_policy(pols_ctx *ctx, const pols_batch *b, const void *coef, int64_t coef_rows, int32_t null_policy, void *pred_out);

/* mode="statistics": replaces the plugin `least_squares_statistics` (src/expressions.rs:468-509) and
 * src/statistics.rs:15-156 for every group of the batch.  Per group, on the sqrt(w)-scaled rows the reference's
 * Python layer hands the plugin (polars_ols/least_squares.py:190-196):
 *   coefficients          from the same dispatcher as pols_least_squares (written to out->coef, batch dtype),
 *   r2, mae, mse          compute_residual_metrics (st.rs:15-37) of those coefficients,
 *   std_err, t, p         compute_feature_metrics (st.rs:79-156): (X'X + alpha I)^-1 by Cholesky (failure -> NaN),
 *                         its own coefficients inv . X'y, RSS / df with df = n - p (alpha == 0) or n - trace(inv),
 *                         two-sided Student-t p-values.
 * The six statistic arrays are always f64 (the reference's struct fields are Float64) and live where `b->mem` says;
 * any of them may be NULL.  out->pred / out->resid are honoured as in pols_least_squares; out->status receives
 * POLS_GROUP_BAD_DOF where the reference would have hit its df > 0 assertion. */
/* Several targets regressed on the same features: replaces the plugin `multi_target_least_squares`
 * (src/expressions.rs:521-591) and solve_multi_target (src/least_squares.rs:243-260).  `b->y` is ignored; `y_cols` holds
 * n_targets column pointers (the fields of the reference's target struct), `pred_cols` n_targets output columns (or NULL),
 * `coef` n_groups x n_targets x (n_features + intercept) in the batch dtype (or NULL), `status` n_groups (or NULL); all
 * live where `b->mem` says.  Unconstrained OLS / ridge with solve_method None or "svd" only, like the reference's Python
 * checks (polars_ols/least_squares.py:303-318, reported as POLS_ERR_PANIC).  Null policies as in the plugin body: the joint
 * validity mask over every target and (unless drop_y_zero_x) every feature (ex.rs:539-548), the fit on the rows it leaves, then
 * predictions for EVERY row from the zero-filled features, masked to NaN under "drop" (ex.rs:566-585). */
int pols_multi_target_least_squares(pols_ctx *ctx, const pols_batch *b, const void *const *y_cols, int32_t n_targets,
                                    const pols_ols_params *p, void *const *pred_cols, void *coef, int32_t *status);

typedef struct pols_stats_out {
    double *r2, *mae, *mse;                 /* n_groups                         */
    double *std_err, *t_values, *p_values;  /* n_groups x (n_features + intercept), row-major */
} pols_stats_out;

int pols_least_squares_statistics(pols_ctx *ctx, const pols_batch *b, const pols_ols_params *p, pols_out *out,
                                  const pols_stats_out *stats);

/* ---- group-key ingestion: `.over(key)` / `group_by(key)` ------------------------------------------------------------
 * The reference's plugin functions never see a key column: Polars partitions the frame on the host and calls them once per
 * group (README.md:19, README.md:57 and :91 `.over("group")`, tests/test_ols.py:110, :384, :860).  The batched entries
 * above take the whole frame with rows sorted by group, so the host needs that partitioning; these entries do it where the
 * columns live.  A pols_layout holds, for one key column of one frame: the stable permutation `order` (sorted position i
 * holds frame row order[i]; rows of a group keep their frame order, which the recursive / rolling models depend on), the
 * groups' offsets and keys (ascending).  Keys are 64-bit integers (the host hashes / dictionary-encodes anything else, as
 * Polars does for its own group-by).  Fewer than 2^32 rows. */
typedef struct pols_layout pols_layout;
/* keys: n_rows int64 where `mem` says.  Synchronises the context's stream (the group count comes back to the host). */
int pols_layout_create(pols_ctx *ctx, const int64_t *keys, int64_t n_rows, int mem, pols_layout **out);
void pols_layout_destroy(pols_layout *layout);
int64_t pols_layout_n_rows(const pols_layout *layout);
int64_t pols_layout_n_groups(const pols_layout *layout);
/* 1 when the key column was already non-decreasing: take / untake are copies, callers may skip them and pass the frame's
 * own columns. */
int pols_layout_is_identity(const pols_layout *layout);
/* host arrays owned by the layout: n_groups + 1 offsets (pass as pols_batch.group_offsets) and n_groups keys */
const int64_t *pols_layout_group_offsets(const pols_layout *layout);
const int64_t *pols_layout_group_keys(const pols_layout *layout);
/* frame order -> group order: dst[c][i] = src[c][order[i]] for n_cols columns whose elements are element_bytes wide: 1
 * (validity bytes), 4 or 8 (f32 / f64 columns), or any other multiple of 4 for row-major tables moved a row at a time
 * (an [n_rows, k] coefficient table is one "column" of k * sizeof(T)-byte elements).  src != dst.  `mem` says where BOTH tables' columns live (host columns are staged through the device). */
int pols_layout_take(pols_ctx *ctx, pols_layout *layout, int element_bytes, const void *const *src_cols, void *const *dst_cols,
                     int32_t n_cols, int mem);
/* group order -> frame order: dst[c][order[i]] = src[c][i]  (predictions / residuals / per-row coefficients back onto the
 * frame, like the Series `.over` returns). */
int pols_layout_untake(pols_ctx *ctx, pols_layout *layout, int element_bytes, const void *const *src_cols, void *const *dst_cols,
                       int32_t n_cols, int mem);
/* out[r] = index (into group_keys / a coefficient table) of the group frame row r belongs to: what broadcasts a per-group
 * coefficient struct over the frame (mode="coefficients" under `.over`, README.md:91). */
int pols_layout_row_groups(pols_ctx *ctx, pols_layout *layout, int64_t *out, int mem);

/* ---- Arrow C Data Interface entry: what the reference's plugin functions do AROUND the solver ---------------------------------
 * `#[polars_expr] fn least_squares(inputs: &[Series], kwargs)` / `least_squares_coefficients` (src/expressions.rs:390-446) receive
 * their Series over the Arrow C Data Interface (one ArrowArray per chunk: values + optional validity BITMAP + element offset),
 * cast them to Float64 with nulls -> NaN and rechunk (convert_polars_to_ndarray :66-103), and return a Series the same way
 * (Float64 with a validity mask :145-158, or a struct of per-feature coefficients with NaN -> null :114-143).  This entry takes
 * the arrays exactly as Polars holds them -- any numeric primitive type, null bitmaps, sliced (offset != 0) and multi-chunk
 * columns -- and hands back a released-by-callback ArrowArray / ArrowSchema pair.  The structs below are the interface's own ABI
 * (https://arrow.apache.org/docs/format/CDataInterface.html); the guard is the one every implementation uses. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif

typedef struct {
    const struct ArrowSchema *schema;         /* format: g f l L i I s S c C (what the reference can cast to Float64) */
    const struct ArrowArray *const *chunks;   /* the Series' chunks, in order; host memory; borrowed for the duration of the call */
    int32_t n_chunks;
} pols_arrow_column;

typedef enum { POLS_MODE_PREDICTIONS = 0, POLS_MODE_RESIDUALS = 1, POLS_MODE_COEFFICIENTS = 2 } pols_output_mode;

/* target / features / weights: inputs[0], inputs[1..] and the sample_weights expression of polars_ols/least_squares.py:163-239
 * (sqrt(w) scaling, the "const" column and the 1/sqrt(w) un-scaling are fused, as in pols_least_squares).  group_offsets NULL
 * = one group holding every row (the per-group call a plugin receives); otherwise rows sorted by group as in pols_batch.
 * Every input Float32 -> computed and returned in f32; anything else -> f64 (the reference's Float64).
 * out / out_schema: caller-allocated structs, filled in; the caller releases them through their `release` callbacks.
 *   predictions / residuals: a primitive array of n_rows values named after the target; nulls where null_policy "drop" masks
 *                            the rows it left out of the fit (residuals: also where the target is null)
 *   coefficients:            a struct array "coefficients" of n_groups rows, one field per feature (+ "const"), NaN -> null */
int pols_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                             const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                             const pols_ols_params *p, int32_t mode, struct ArrowArray *out, struct ArrowSchema *out_schema);

/* The other seven plugin functions (src/expressions.rs:468-741), same conventions: columns as Polars holds them in, a
 * released-by-callback ArrowArray / ArrowSchema pair out; group_offsets NULL = the per-group call; all inputs Float32 -> f32.
 *
 * least_squares_statistics (ex.rs:448-509): a struct array "statistics" with ONE row per group and the fields of
 * statistics_struct_dtype (:448-466): r2, mae, mse (Float64), feature_names (list<str>), coefficients, standard_errors,
 * t_values, p_values (list<Float64>); lists are Arrow large lists ("+L") of large strings ("U") / Float64.  A group whose
 * degrees of freedom are <= 0 fails the call with POLS_ERR_PANIC like the reference's assert (src/statistics.rs:131-134). */
int pols_least_squares_statistics_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features,
                                        int32_t n_features, const pols_arrow_column *weights, const int64_t *group_offsets,
                                        int64_t n_groups, int32_t add_intercept, const pols_ols_params *p, struct ArrowArray *out,
                                        struct ArrowSchema *out_schema);
/* multi_target_least_squares (ex.rs:511-591): `targets` is the STRUCT Series of inputs[0] (format "+s", one numeric field per
 * target; a null struct row is a null in every field); out: a struct array "predictions" of n_rows rows with the targets' field
 * names (multi_target_struct_dtype, :511-519), NaN -> null.  Residuals are the caller's `target - predictions`
 * (polars_ols/least_squares.py:236-239). */
int pols_multi_target_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *targets, const pols_arrow_column *features,
                                          int32_t n_features, const pols_arrow_column *weights, const int64_t *group_offsets,
                                          int64_t n_groups, int32_t add_intercept, const pols_ols_params *p, struct ArrowArray *out,
                                          struct ArrowSchema *out_schema);
/* recursive_least_squares / recursive_least_squares_coefficients (ex.rs:593-646) and rolling_least_squares /
 * rolling_least_squares_coefficients (:648-701); mode POLS_MODE_PREDICTIONS: a primitive array named after the target, null where
 * the validity mask of the null policy masks the row (make_predictions with is_valid, :640-645) or no estimate exists yet;
 * POLS_MODE_COEFFICIENTS: a struct array "coefficients" with ONE ROW PER INPUT ROW, one field per feature (+ "const"), NaN -> null.
 * Quirk kept: the reference's prediction form of RLS ignores initial_state_mean (ex.rs:636) -- pass NULL for it there. */
int pols_recursive_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features,
                                       int32_t n_features, const pols_arrow_column *weights, const int64_t *group_offsets,
                                       int64_t n_groups, int32_t add_intercept, const pols_rls_params *p, int32_t mode,
                                       struct ArrowArray *out, struct ArrowSchema *out_schema);
int pols_rolling_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features,
                                     int32_t n_features, const pols_arrow_column *weights, const int64_t *group_offsets,
                                     int64_t n_groups, int32_t add_intercept, const pols_rolling_params *p, int32_t mode,
                                     struct ArrowArray *out, struct ArrowSchema *out_schema);
/* predict (ex.rs:706-741): `coefficients` is the coefficients STRUCT Series of inputs[0] -- one row per input row (what Polars
 * broadcasts / joins it to), n_features (+ 1 with add_intercept: the pl.lit(1.0) "const" feature of least_squares.py:479-483)
 * numeric fields; null_policy is one of ignore / zero / drop (least_squares.py:474): features are zero-filled unless "ignore"
 * (:725), "drop" nulls the rows with a null anywhere (:732-738).  out: a primitive array named `name` (NULL / "" -> "predictions"). */
int pols_predict_arrow(pols_ctx *ctx, const pols_arrow_column *coefficients, const pols_arrow_column *features, int32_t n_features,
                       int32_t add_intercept, int32_t null_policy, const char *name, struct ArrowArray *out,
                       struct ArrowSchema *out_schema);

/* ---- more than one GPU ------------------------------------------------------------------------------------------------
 * Groups are independent in the reference -- every plugin call sees one group's rows, nothing in src/least_squares.rs carries
 * state across groups, and Polars runs the calls concurrently on its rayon pool (README.md:19) -- so the data path has NO
 * collective: each GPU owns a contiguous range of groups (balanced by rows) and runs the same entries on its shard through its
 * own pols_ctx.  The one exchange step is what Polars does when it concatenates the per-group outputs: re-assembling an output
 * column.  These entries do that over RCCL / xGMI.  RCCL is bound at run time (librccl.so.1; an instance the process already
 * holds is reused), so single-GPU callers never load it.
 *
 *   one process per GPU (MPI / torch.distributed launch):  rank 0 calls pols_comm_unique_id, ships the 128 bytes to every rank
 *       out of band, every rank calls pols_comm_create with its own context.
 *   one process, several GPUs (what a Polars plugin process is):  pols_create per device, then ONE pols_comm_create_all; the
 *       per-device collective calls of one exchange are bracketed by pols_comm_group_begin / _end (or issued from one host
 *       thread per device).
 * Collectives are asynchronous on the context's stream (ordered behind the kernels that produced `local`). */
typedef struct pols_comm pols_comm;
#define POLS_COMM_ID_BYTES 128
/* bounds_out[0 .. world_size]: rank r owns groups [bounds_out[r], bounds_out[r + 1]); boundary r is the first group boundary whose
 * cumulative row count reaches r / world_size of the rows.  A pure function of the offsets: no communication needed to agree. */
int pols_partition_groups(const int64_t *group_offsets, int64_t n_groups, int world_size, int64_t *bounds_out);
int pols_comm_unique_id(void *id_out /* POLS_COMM_ID_BYTES */);
int pols_comm_create(pols_ctx *ctx, const void *id, int world_size, int rank, pols_comm **out);
int pols_comm_create_all(pols_ctx *const *ctxs, int n, pols_comm **out /* n communicators, rank i on ctxs[i]'s device */);
void pols_comm_destroy(pols_comm *comm);
int pols_comm_world_size(const pols_comm *comm);
int pols_comm_rank(const pols_comm *comm);
/* What RCCL itself says about this communicator -- not what the caller passed to pols_comm_create: the rank count and rank the library
 * reports (ncclCommCount / ncclCommUserRank), the HIP device it is bound to (ncclCommCuDevice) with its PCI bus id, and the library's
 * version code (ncclGetVersion).  A measurement line that carries these proves the collective saw N ranks on N different devices. */
typedef struct pols_comm_info {
    int32_t nranks_seen, rank_seen, device, rccl_version;
    char pci_bus_id[32];
} pols_comm_info;
int pols_comm_query(const pols_comm *comm, pols_comm_info *out);
int pols_comm_group_begin(void);
int pols_comm_group_end(void);
/* Every rank receives all rows, in rank order (= group order, the shards being contiguous ranges): `local` holds counts[rank]
 * rows of row_bytes bytes (a [groups x k] coefficient table: row_bytes = k * sizeof(element)), `out` sum(counts) rows; both
 * DEVICE.  One ncclAllGather when every shard has the same size, otherwise one grouped broadcast per owner (all-gatherv). */
int pols_comm_allgather_rows(pols_comm *comm, const void *local, const int64_t *counts, int64_t row_bytes, void *out);
/* Per-row outputs (predictions / residuals) to ONE root, in frame order: grouped ncclSend / ncclRecv -- over point-to-point xGMI
 * the root pulls from its peers over distinct links at once, where a ring all-gather would be bound by one link. */
int pols_comm_gather_rows(pols_comm *comm, const void *local, const int64_t *counts, int64_t row_bytes, int root, void *out_on_root);

/* One process, several GPUs -- the form a Polars plugin process takes: the frame sits in HOST memory (b->mem must be POLS_MEM_HOST),
 * ctxs[0 .. n) are contexts on n devices.  The groups are cut into n contiguous ranges balanced by rows (pols_partition_groups),
 * range r is staged to and solved on ctxs[r]'s device from its own host thread, exactly as pols_least_squares would, and the outputs
 * named in `o` are re-assembled according to out_mem:
 *   POLS_MEM_HOST    o's pointers are host arrays for the WHOLE frame; every device copies its slice into them (no collective;
 *                    comms may be NULL);
 *   POLS_MEM_DEVICE  o's pointers are buffers on ctxs[0]'s device for the whole frame: the coefficient table is all-gathered
 *                    (pols_comm_allgather_rows), predictions / residuals / status are gathered to device 0 (pols_comm_gather_rows)
 *                    over RCCL / xGMI; comms[r] must be rank r of a world of n (pols_comm_create_all).  Returns when device 0 holds them.
 * Replaces what Polars' rayon pool does for `.over(group)` (README.md:19) when more than one GPU is present. */
int pols_least_squares_sharded(pols_ctx *const *ctxs, pols_comm *const *comms, int n, const pols_batch *b, const pols_ols_params *p,
                               pols_out *o, int32_t out_mem);

#ifdef __cplusplus
}
#endif
#endif
