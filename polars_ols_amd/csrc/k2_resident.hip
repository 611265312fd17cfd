// k2_resident.hip -- variant choice of K2 (k2_resident.hpp; kernels: k2_kernel.inl, instantiated in k2_f32.hip / k2_f64.hip).
#include "k2_resident.hpp"

namespace pols {

template <typename T, bool HAS_W> int k2_launch_t(pols_ctx *ctx, const K2Args &a, int64_t need);

// rows the chunk grid of the largest group spans: a group that does not start on a 16-byte boundary begins up to VEC - 1 rows early
static int64_t k2_need(int dtype, int64_t max_group_rows, bool offsets_aligned) {
    const int vec = dtype == POLS_F32 ? 4 : 2;
    return max_group_rows + (offsets_aligned ? 0 : vec - 1);
}

bool k2_fits(int dtype, int kt, int64_t max_group_rows, bool offsets_aligned) {
    const int vec = dtype == POLS_F32 ? 4 : 2;
    return kt >= 1 && kt <= K2_KMAX && k2_need(dtype, max_group_rows, offsets_aligned) <= (int64_t)512 * ((kt == 7 || kt == 8) ? 4 : 2) * vec;   // (+ n_rows >= vec: caller)
    // (four chunks per lane, 8 192 f32 / 4 096 f64 rows: the eight-slot kernel loads all eight column slots whatever kt is, so below seven
    // columns the two-pass streamed path is faster -- 2.8-3.0 TB/s against 1.4-2.7; at 7-8 columns 2.9-4.1 against 2.8,
    // profiles/r05_sweep_k2rc4_ab.txt)
}

int k2_launch(pols_ctx *ctx, int dtype, const K2Args &a, int64_t max_group_rows) {
    if (a.kt < 1 || a.kt > K2_KMAX) return fail(POLS_ERR_UNSUPPORTED, "k2: %d columns > %d", a.kt, K2_KMAX);
    const bool aligned = ctx->offs_aligned[dtype == POLS_F32 ? 1 : 0];
    const int64_t need = k2_need(dtype, max_group_rows, aligned);
    if (a.w) return dtype == POLS_F32 ? k2_launch_t<float, true>(ctx, a, need) : k2_launch_t<double, true>(ctx, a, need);
    return dtype == POLS_F32 ? k2_launch_t<float, false>(ctx, a, need) : k2_launch_t<double, false>(ctx, a, need);
}

}  // namespace pols
