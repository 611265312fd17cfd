// k4c_rolling.hip -- K4c on null-free frames (kernel, interface and reference citations: k4c_kernel.inl); the masked form for frames with
// validity bytes is instantiated in k4cm_rolling.hip.
#include "k4c_kernel.inl"

namespace pols {

int k4cm_launch(pols_ctx *ctx, int dtype, const K4cArgs &a);
int k4cg_launch(pols_ctx *ctx, int dtype, const K4cArgs &a);

int k4c_launch(pols_ctx *ctx, int dtype, const K4cArgs &a) {
    if (a.window < 1 || (a.window > k4c_max_window(a.k) && !a.tile_row0) || a.min_periods < 1 || a.min_periods > a.window)
        return fail(POLS_ERR_INVALID, "k4c: window %lld / min_periods %lld outside the row-parallel kernel's range", (long long)a.window, (long long)a.min_periods);
    if (a.valid) return k4cm_launch(ctx, dtype, a);
    if (a.src) return k4cg_launch(ctx, dtype, a);
    ctx->last_kernel = dtype == POLS_F32 ? "k4_rolling_tiles_f32" : "k4_rolling_tiles_f64";
    return dtype == POLS_F32 ? k4c_launch_t<float, false>(ctx, a) : k4c_launch_t<double, false>(ctx, a);
}

}  // namespace pols
