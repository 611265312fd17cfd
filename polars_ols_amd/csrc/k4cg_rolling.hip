// k4cg_rolling.hip -- K4c, GATHER: the "drop" family of rolling OLS on frames WITH nulls (src/least_squares.rs:947-986) on the row-parallel tile
// kernel (k4c_kernel.inl): the kernel's rows are the frame's valid rows read through a source map, the outputs go to the frame's rows
// (dyn_out_gather.inl) -- no compacted copy of the columns, no expansion pass.
#include "k4c_kernel.inl"

namespace pols {

int k4cg_launch(pols_ctx *ctx, int dtype, const K4cArgs &a) {
    ctx->last_kernel = dtype == POLS_F32 ? "k4_rolling_tiles_f32_gathered" : "k4_rolling_tiles_f64_gathered";
    return dtype == POLS_F32 ? k4c_launch_t<float, false, true>(ctx, a) : k4c_launch_t<double, false, true>(ctx, a);
}

}  // namespace pols
