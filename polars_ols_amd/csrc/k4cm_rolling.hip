// k4cm_rolling.hip -- K4c, MASKED: rolling OLS over a FIXED window of rows with validity bytes ("drop_window", src/least_squares.rs:987-1029)
// on the row-parallel tile kernel (k4c_kernel.inl), up to 10 features.  The rows the reference does not solve are rewritten by the fill
// pass below from the per-row table rm_tables_launch builds (dyn_prep.hip).
#include "k4c_kernel.inl"

namespace pols {

int k4cm_launch(pols_ctx *ctx, int dtype, const K4cArgs &a) {
    ctx->last_kernel = dtype == POLS_F32 ? "k4_rolling_tiles_masked_f32" : "k4_rolling_tiles_masked_f64";
    return dtype == POLS_F32 ? k4c_launch_t<float, true>(ctx, a) : k4c_launch_t<double, true>(ctx, a);
}

}  // namespace pols
