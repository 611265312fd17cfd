// k5_enet.hpp -- K5 "enet_gram_cd": elastic net / lasso / non-negative least squares for every group.
//
// Replaces solve_elastic_net (src/least_squares.rs:386-492) + make_predictions (src/expressions.rs:175-195).
// Three launches:
//   A  gram_stream : one workgroup per group streams the group's columns through LDS in 256-row chunks
//                    (global_load_lds DMA) and accumulates Z^T Z, Z = [sqrt(w) X | sqrt(w) 1 | sqrt(w) y], on the
//                    matrix cores as up to two 16-column tiles (k + 1 <= 32): the ONE pass over X that the
//                    reference repeats ~3k times per sweep (x_j += / dot / -= over n rows, :426-433).  HBM-bound:
//                    b n (k + 1) bytes per group.
//   B  gram_cd     : cyclic coordinate descent on (X^T X, X^T y) -- algebraically the reference's residual-form
//                    update (dot_j = x_j . (y - X w + x_j w_j) = b_j - sum_i G_ji w_i + G_jj w_j), same coordinate
//                    order, alpha * n scaling (:419), soft threshold (:373-379), active-set variant (:446-489) and
//                    the ||w - w_old||_2 < tol stop (:436-444), in f64.  16 lanes per group.
//   C  predict     : pred = X . w (and residuals) with per-group coefficients; also the body of pols_predict.
#pragma once
#include "common.hpp"

namespace pols {

struct GramArgs {
    const void *y;
    const void *w;
    const void *x[POLS_MAX_FEATURES];
    const int64_t *offs;
    int64_t n_groups;
    int64_t n_rows;
    double *gram;        // n_groups x NZ x NZ (row-major, f64), NZ = kt + 1
    int32_t k_user;
    int32_t kt;          // k_user + intercept
    // null policy fused into the staging pass: dropped rows get sqrt(w) = 0, nulls that stay in the fit become 0
    const uint8_t *valid;   // optional, drop family only
    int32_t null_policy;    // pols_null_policy
    double *nvalid;         // n_groups: rows that took part in the fit (the n of alpha * n, ls.rs:419), or nullptr
    int32_t offs_pairs;     // 0: item g is rows [offs[g], offs[g + 1]); 1: [offs[2 g], offs[2 g + 1]) -- segment tables of a LIST of groups (size classes)
};

struct CdArgs {
    const double *gram;  // from gram_stream
    const int64_t *offs;
    int64_t n_groups;
    void *coef;          // n_groups x kt, dtype of the batch (may alias nothing else)
    double *coef64;      // n_groups x kt f64 scratch handed to the prediction pass
    int32_t *status;
    double alpha, l1_ratio, tol;
    double pivot_tol;    // gram_solve only: see K1Args::pivot_tol
    int32_t *fb_flag;    // gram_solve only: see K1Args::fb_flag
    int32_t epoch;
    int64_t max_iter;
    int32_t positive, active_set, kt;
    const double *nvalid;   // per-group number of fit rows under a null policy, or nullptr = offs[g + 1] - offs[g]
    int32_t solver;         // gram_solve only: 0 Cholesky, 1 partial-pivot LU (solve_method = "lu", ls.rs:264-273)
    int32_t lu_fallback;    // gram_solve only: a failed Cholesky is retried with LU (solve_ridge, ls.rs:358-363)
    const int32_t *glist;   // gram_solve only (size classes): item i is group glist[i] -- gram / coef64 / nvalid are indexed by item, offs / coef / status by group
};

struct PredictArgs {
    const void *y;       // for residuals, or nullptr
    const void *w;       // sample weights (reference NaN semantics at w == 0), or nullptr
    const void *x[POLS_MAX_FEATURES];
    const int64_t *offs;
    int64_t n_groups;
    int64_t n_rows;
    const double *coef64;   // n_groups x kt (per-group coefficients) ...
    const void *coef_rows;  // ... or n_rows x kt in the batch dtype (dynamic models / pols_predict)
    void *pred;
    void *resid;
    int32_t k_user, kt;
    // null policy (static models): features are zero-filled for every policy but "ignore" (construct_features_array(.., true),
    // ex.rs:408); "drop" masks the rows that were not part of the fit with NaN (ex.rs:409-417).  `y` must then be set.
    const uint8_t *valid;
    int32_t null_policy;
    const int32_t *gmap;    // SPLIT groups (or nullptr): `offs` cuts long groups into segments, segment g uses the coefficients of group gmap[g]
    int64_t max_item_rows;  // rows of the longest item of `offs` (or 0 = unknown): sizes the workgroups per item
    int32_t offs_pairs;     // see GramArgs::offs_pairs
};

// Long groups cut into segments (one workgroup each in gram_stream / predict): partial Gram matrices (and fit-row counts) of the
// segments [first[g], first[g + 1]) summed into group g's, in segment order.
struct GramReduceArgs {
    const double *part;     // n_segments x NZ x NZ
    const double *nv_part;  // n_segments or nullptr
    const int32_t *first;   // n_groups + 1
    double *gram;           // n_groups x NZ x NZ
    double *nvalid;         // n_groups or nullptr
    int64_t n_groups;
    int32_t nz2;
    int32_t max_segments;   // most segments of one group (0: unknown -> the four-wave form)
    double *slices;         // n_groups x n_slices x NZ x NZ or nullptr: few groups of very many segments -- two-level sum (gram_reduce_launch)
    int32_t n_slices;
};
int gram_reduce_launch(pols_ctx *ctx, const GramReduceArgs &a);

int gram_stream_launch(pols_ctx *ctx, int dtype, const GramArgs &a);
// K5v (k5v_gram.hip): the same pass on the VALU, rows loaded straight into registers -- up to K5V_MAX_KT columns, no sample weights, no
// null policy; gram_stream_launch routes to it (POLS_KG_SINGLE_BUFFER = the round-4 pass: MFMA tiles, one 256-row LDS buffer)
constexpr int K5V_MAX_KT = 10;
constexpr int K5V_MAX_KT_F32 = 13;   // f32 frames: up to 13 columns (105 accumulators + 14 vectors in flight: 254 registers, two waves per SIMD; 14-16 need AGPRs)
bool gram_valu_takes(const pols_ctx *ctx, int dtype, const GramArgs &a);
int gram_valu_launch(pols_ctx *ctx, int dtype, const GramArgs &a);
int gram_cd_launch(pols_ctx *ctx, int dtype, const CdArgs &a);
// OLS / ridge (alpha = ridge penalty) from the streamed Gram: generic kt <= 31, any group size
int gram_solve_launch(pols_ctx *ctx, int dtype, const CdArgs &a);
int predict_launch(pols_ctx *ctx, int dtype, const PredictArgs &a);

}  // namespace pols
