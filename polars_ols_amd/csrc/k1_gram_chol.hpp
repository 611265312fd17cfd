// k1_gram_chol.hpp -- K1 "gram_chol_predict": the static least-squares hot path.
//
// Replaces, for ALL groups in one launch, what the reference does once per plugin call:
//   construct_features_array   src/expressions.rs:22-63   (column -> row-major copy: deleted, we read the columns)
//   solve_ols / solve_ridge    src/least_squares.rs:211-240, 342-364  (QR / Gram+Cholesky on one small problem)
//   make_predictions           src/expressions.rs:175-195 (X . beta)
//   sqrt(w) scaling, intercept, 1/sqrt(w) un-scaling, residuals   polars_ols/least_squares.py:184-196, 234-239
//
// Mapping to gfx950: one TEAM (a 64-lane wave, or a 256-thread workgroup) per group.  Lanes walk the
// row axis of every column with 16-byte loads (lane i owns rows base+VEC*i ..), so each wave instruction
// reads 1 KiB of one column, fully coalesced.  The rows a lane loaded stay in its VGPRs (RC chunks per
// lane), so X is read from HBM exactly once: the Gram matrix Z^T Z of Z = [sqrt(w) X | sqrt(w) y] is
// accumulated per lane (packed upper triangle), reduced across the wave with DPP and across the waves of
// the workgroup through LDS in a fixed order (deterministic, no atomics); every lane then runs the same
// K x K Cholesky + two triangular solves on wave-uniform values, and the prediction X . beta is formed
// from the register-resident rows and stored with 16-byte stores.
//
// Algorithmic HBM traffic per group (n rows, k features, dtype size b): read b*n*(k+1) (+ b*n weights),
// write b*n predictions  ->  cfg2 (f32, n=1000, k=8): 40 000 B.  Flops ~ n*(k+1)(k+2) + 2nk: AI ~ 2.6
// flop/B, so the kernel is HBM-bound; there is no GEMM worth an MFMA tile at k <= 10 (a 16x16 tile would
// be 32 % used and cost 5x the VALU form), the MFMA Gram engine lives in k5/k2 for k > 10.
#pragma once

#include "common.hpp"

namespace pols {

struct K1Args {
    const void *y;
    const void *w;                       // sample weights or nullptr
    const uint8_t *valid;                // row validity bytes or nullptr (drop-family null policies)
    int32_t null_policy;                 // pols_null_policy; anything but "ignore" selects the NULLS kernels
    const void *x[POLS_MAX_FEATURES];    // user feature columns
    const int64_t *offs;                 // device, n_groups + 1
    int64_t n_groups;
    int64_t n_rows;
    void *coef;                          // n_groups x KT or nullptr
    void *pred;                          // n_rows or nullptr
    void *resid;                         // n_rows or nullptr
    int32_t *status;                     // n_groups or nullptr
    double alpha;                        // ridge penalty added to diag(X^T X) (ls.rs:355-356)
    int32_t *fb_flag;                    // device word stamped with `epoch` when any group is flagged (lets K6 exit at once otherwise)
    int32_t epoch;
    double pivot_tol;                    // flag the group for the SVD fallback when a Cholesky pivot d_j <= pivot_tol * G_jj
    int32_t k_user;                      // KT - add_intercept
    int64_t skip_group;                  // K1p: the group a single wave handles after the persistent loop (or -1)
    unsigned long long *dbg;             // POLS_TIMELINE=1: 8 s_memtime stamps per group (debug only)
    int64_t xcd_chunk;                   // 0: workgroup b takes block b of the groups; else b -> (b % 8) * xcd_chunk + b / 8, so that each XCD
                                         // (workgroups are dealt round-robin over the eight) walks one contiguous eighth of every column
    // SIZE CLASSES (round 5): a frame whose group sizes spread widely is served by one launch per size class, each sized for its own largest
    // group.  `glist` (or nullptr) lists the groups of this launch's class in ascending order and n_groups counts THEM: work item i is group
    // glist[i]; class_max_rows (> 0) is the largest group of the class.
    const int32_t *glist;
    int64_t class_max_rows;
};

// Launches the (dtype, KT, team, resident-chunks) variant that fits max_group_rows.
int k1_launch(pols_ctx *ctx, int dtype, int kt, const K1Args &a, int64_t max_group_rows, bool aligned16);

constexpr int K1_MAX_KT = 10;    // register-resident VALU engine: every variant (null-policy family, streamed overflow, K1t, K1p)
constexpr int K1X_MAX_KT = 31;   // ... with the row-resident Cholesky and up to 15 passes, one chunk per lane (k1w_*.hip)
constexpr int K1W_MAX_KT = 15;   // ... its multi-pass resident forms only (three passes up to 12 columns, four beyond)
constexpr int K1M_MAX_KT = 15;   // LDS tile + MFMA engine: [X | y] must fit one 16 x 16 tile

}  // namespace pols
