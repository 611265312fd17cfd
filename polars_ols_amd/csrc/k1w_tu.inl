// k1w_tu.inl -- one translation unit of the 16..31-column resident multi-pass K1 kernels: four column counts of one dtype per
// TU (K1W_T, K1W_LO, K1W_FN set by the including .hip; K1_NULLS_TU for the null-policy family, k1nw_*.hip), so that the fully
// unrolled Gram passes compile in parallel.
#define K1_WIDE_TU 1
#include "k1_kernel.inl"

namespace pols {
int K1W_FN(pols_ctx *ctx, int kt, const K1Args &a, int64_t max_rows) {
    switch (kt) {
        case K1W_LO + 0: return k1_launch_wide_kt<K1W_T, K1W_LO + 0>(ctx, a, max_rows);
        case K1W_LO + 1: return k1_launch_wide_kt<K1W_T, K1W_LO + 1>(ctx, a, max_rows);
        case K1W_LO + 2: return k1_launch_wide_kt<K1W_T, K1W_LO + 2>(ctx, a, max_rows);
        case K1W_LO + 3: return k1_launch_wide_kt<K1W_T, K1W_LO + 3>(ctx, a, max_rows);
        default: return fail(POLS_ERR_UNSUPPORTED, "k1w: %d columns are not in this unit", kt);
    }
}
}  // namespace pols
