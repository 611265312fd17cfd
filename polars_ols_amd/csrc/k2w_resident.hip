// k2w_resident.hip -- variant choice of K2w (k2w_resident.hpp; kernels: k2w_kernel.inl, instantiated in k2w_f32.hip / k2w_f64.hip).
#include "k2w_resident.hpp"

namespace pols {

template <typename T> int k2w_launch_t(pols_ctx *ctx, const K2wArgs &a, int64_t need);

// rows the chunk grid of the largest group spans: a group that does not start on a 16-byte boundary begins up to VEC - 1 rows early
static int64_t k2w_need(int dtype, int64_t max_group_rows, bool offsets_aligned) {
    const int vec = dtype == POLS_F32 ? 4 : 2;
    return max_group_rows + (offsets_aligned ? 0 : vec - 1);
}

bool k2w_fits(int dtype, int kt, int64_t max_group_rows, bool offsets_aligned) {
    const int vec = dtype == POLS_F32 ? 4 : 2;
    return kt >= K2W_KMIN && kt <= K2W_KMAX && k2w_need(dtype, max_group_rows, offsets_aligned) <= (int64_t)512 * vec;   // (+ n_rows >= vec: caller)
}

int k2w_launch(pols_ctx *ctx, int dtype, const K2wArgs &a, int64_t max_group_rows) {
    if (a.kt < K2W_KMIN || a.kt > K2W_KMAX) return fail(POLS_ERR_UNSUPPORTED, "k2w: %d columns outside %d..%d", a.kt, K2W_KMIN, K2W_KMAX);
    const int64_t need = k2w_need(dtype, max_group_rows, ctx->offs_aligned[dtype == POLS_F32 ? 1 : 0]);
    return dtype == POLS_F32 ? k2w_launch_t<float>(ctx, a, need) : k2w_launch_t<double>(ctx, a, need);
}

}  // namespace pols
