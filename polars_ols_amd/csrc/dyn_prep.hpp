// dyn_prep.hpp -- what the reference's Python layer and plugin bodies do AROUND the dynamic solvers (recursive / rolling least
// squares), as three device passes behind the C-ABI instead of a dozen host-side column operations:
//   scan     compute_is_valid_mask (src/expressions.rs:201-228) for the call's null policy from the NaNs (= nulls) of the
//            columns: one validity byte per row + "any row invalid" / "any null in a kept row" flags
//   rewrite  polars_ols/least_squares.py:184-196 (sqrt(w) scaling of target and features, the ones column appended LAST) and the
//            NullPolicy::Zero conversion of the inputs (ex.rs:603, 629, 656, 683: nulls -> 0) in one read + write pass; skipped
//            when there are no weights, no intercept and no nulls (the kernels then read the caller's columns in place)
//   post     predictions *= 1 / sqrt(w) (ls.py:234-235) and the is_valid mask of make_predictions (ex.rs:640-645, 695-700)
#pragma once
#include "common.hpp"

namespace pols {

struct DynPrepArgs {
    const void *y, *w;                   // original target / weights (w may be nullptr)
    const void *const *xtab;             // DEVICE table of k_user feature column pointers
    int32_t k_user, add_intercept, null_policy;
    int64_t n_rows;
    uint8_t *valid_out;                  // n_rows bytes
    int32_t *flags;                      // [0] rows left out of the fit, [1] nulls inside kept rows (either > 0 = "some")
    void *y_out;                         // rewritten target
    void *const *xout;                   // DEVICE table of k_user + add_intercept rewritten column pointers
    void *sw_out;                        // sqrt(w) per row (only with weights)
    void *pred;                          // post: predictions in place
    const uint8_t *valid_post;           // post: validity bytes or nullptr (no masking)
};

int dyn_scan_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);
int dyn_rewrite_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);
int dyn_post_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);

}  // namespace pols
