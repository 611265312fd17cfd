// dyn_prep.hip -- see dyn_prep.hpp.  HBM-bound element-wise passes: one thread per row walks the columns (consecutive lanes =
// consecutive rows of every column: coalesced), nothing to tile.
#include "dyn_prep.hpp"

namespace pols {

template <typename T>
__global__ void __launch_bounds__(256) dyn_scan_kernel(const DynPrepArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int invalid = 0, nulls = 0;
    if (r < a.n_rows) {
        const T yv = static_cast<const T *>(a.y)[r];
        bool ynull = yv != yv, xnull = false;
        for (int j = 0; j < a.k_user; ++j) { const T v = static_cast<const T *>(a.xtab[j])[r]; xnull = xnull || (v != v); }
        if (a.w) { const T wv = static_cast<const T *>(a.w)[r]; (void)wv; }   // a null weight is a weight of 1e-24, not a dropped row (ls.py:193)
        const int pol = a.null_policy;
        bool ok = true;                                                         // "ignore" / "zero": every row is kept (ex.rs:221-226)
        if (pol == POLS_NULL_DROP || pol == POLS_NULL_DROP_ZERO || pol == POLS_NULL_DROP_WINDOW) ok = !ynull && !xnull;   // :209-216
        else if (pol == POLS_NULL_DROP_Y_ZERO_X) ok = !ynull;                                                             // :217-220
        a.valid_out[r] = ok ? 1 : 0;
        invalid = ok ? 0 : 1;
        nulls = (ok && (ynull || xnull)) ? 1 : 0;
    }
    const unsigned long long bi = __ballot(invalid), bn = __ballot(nulls);
    if ((threadIdx.x & 63) == 0) {
        if (bi) atomicAdd(&a.flags[0], __popcll(bi));
        if (bn) atomicAdd(&a.flags[1], __popcll(bn));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) dyn_rewrite_kernel(const DynPrepArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    T sw = T(1);
    if (a.w) {
        T wv = static_cast<const T *>(a.w)[r];
        if (wv != wv) wv = (T)1e-24;                                            // sqrt_w.fill_null(1e-12)  (ls.py:193)
        sw = sqrt(wv);
        static_cast<T *>(a.sw_out)[r] = sw;
    }
    T yv = static_cast<const T *>(a.y)[r];
    static_cast<T *>(a.y_out)[r] = (yv != yv) ? T(0) : yv * sw;                 // NullPolicy::Zero conversion (ex.rs:603, 629, 656, 683)
    for (int j = 0; j < a.k_user; ++j) {
        T v = static_cast<const T *>(a.xtab[j])[r];
        static_cast<T *>(a.xout[j])[r] = (v != v) ? T(0) : v * sw;
    }
    if (a.add_intercept) static_cast<T *>(a.xout[a.k_user])[r] = sw;            // the "const" column, scaled like every feature (ls.py:188-196)
}

template <typename T>
__global__ void __launch_bounds__(256) dyn_post_kernel(const DynPrepArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n_rows) return;
    T p = static_cast<T *>(a.pred)[r];
    if (a.sw_out) p *= T(1) / static_cast<const T *>(a.sw_out)[r];              // predictions *= 1 / sqrt_w  (ls.py:234-235)
    p = nan_if<T>((a.valid_post && !a.valid_post[r]) ? 1u : 0u, p);             // make_predictions(.., is_valid)  (ex.rs:640-645)
    static_cast<T *>(a.pred)[r] = p;
}

template <typename T>
__global__ void __launch_bounds__(256) compact_count_kernel(const CompactArgs a) {
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    unsigned cnt = 0;
    for (int64_t r = s + threadIdx.x; r < e; r += 256) {
        bool ok = !a.valid_in || a.valid_in[r];
        if (a.drop)
            for (int c = 0; c < a.n_mask; ++c) { const T v = static_cast<const T *>(a.in[c])[r]; ok = ok && (v == v); }
        a.vbytes[r] = ok ? 1 : 0;
        cnt += ok ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&a.counts[g], (unsigned long long)cnt);
}

template <typename T>
__global__ void __launch_bounds__(256) compact_scatter_kernel(const CompactArgs a) {
    __shared__ unsigned wave_cnt[4];
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    int64_t base = a.offs_out[g];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t r0 = s; r0 < e; r0 += 256) {                 // (uniform trip count: the barriers below are reached by every lane)
        const int64_t r = r0 + threadIdx.x;
        const bool ok = r < e && a.vbytes[r];
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int w = 0; w < 4; ++w) { before += w < wave ? wave_cnt[w] : 0u; total += wave_cnt[w]; }
        if (ok) {
            const int64_t pos = base + before + __popcll(bal & ((1ull << lane) - 1ull));   // stable: the rows keep their order
            for (int c = 0; c < a.n_cols; ++c) {
                T v = static_cast<const T *>(a.in[c])[r];
                if (v != v) v = (c == a.w_col) ? (T)1e-24 : (a.zero_fill ? T(0) : v);
                static_cast<T *>(a.out[c])[pos] = v;
            }
        }
        base += total;
        __syncthreads();
    }
}

#define COMPACT_LAUNCH(kernel)                                                                                          \
    if (a.n_groups == 0) return POLS_OK;                                                                                \
    if (dtype == POLS_F32) hipLaunchKernelGGL(kernel<float>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a); \
    else hipLaunchKernelGGL(kernel<double>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);                  \
    POLS_HIP(hipGetLastError());                                                                                        \
    return POLS_OK;

int compact_count_launch(pols_ctx *ctx, int dtype, const CompactArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    POLS_HIP(hipMemsetAsync(a.counts, 0, sizeof(unsigned long long) * (size_t)a.n_groups, ctx->stream));
    COMPACT_LAUNCH(compact_count_kernel)
}
int compact_scatter_launch(pols_ctx *ctx, int dtype, const CompactArgs &a) { COMPACT_LAUNCH(compact_scatter_kernel) }

template <typename T>
__global__ void __launch_bounds__(256) mt_predict_kernel(const MtPredictArgs a) {
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const T *cg = static_cast<const T *>(a.coef) + (size_t)g * a.m * a.kt;
    const bool icpt = a.kt != a.k_user;
    for (int64_t r = s + threadIdx.x; r < e; r += 256) {
        T sw = T(1);
        if (a.w) { T wv = static_cast<const T *>(a.w)[r]; if (wv != wv) wv = (T)1e-24; sw = sqrt(wv); }
        for (int t0 = 0; t0 < a.m; t0 += 4) {                 // four targets per sweep over the features
            T acc[4] = {T(0), T(0), T(0), T(0)};
            for (int j = 0; j < a.k_user; ++j) {
                T xv = static_cast<const T *>(a.xtab[j])[r];
                xv = (xv != xv) ? T(0) : xv * sw;             // construct_features_array(.., true): nulls -> 0
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t0 + u < a.m) acc[u] = fma(xv, cg[(size_t)(t0 + u) * a.kt + j], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t0 + u >= a.m) break;
                T pv = acc[u];
                if (icpt) pv = fma(sw, cg[(size_t)(t0 + u) * a.kt + a.kt - 1], pv);
                if (a.w) pv *= T(1) / sw;
                pv = nan_if<T>((a.mask_drop && !a.vbytes[r]) ? 1u : 0u, pv);
                static_cast<T *>(a.ptab[t0 + u])[r] = pv;
            }
        }
    }
}

int mt_predict_launch(pols_ctx *ctx, int dtype, const MtPredictArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    if (dtype == POLS_F32) hipLaunchKernelGGL(mt_predict_kernel<float>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(mt_predict_kernel<double>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

#define DYN_LAUNCH(kernel)                                                                                              \
    if (a.n_rows == 0) return POLS_OK;                                                                                  \
    const unsigned blocks = (unsigned)((a.n_rows + 255) / 256);                                                         \
    if (dtype == POLS_F32) hipLaunchKernelGGL(kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, a);               \
    else hipLaunchKernelGGL(kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, a);                                \
    POLS_HIP(hipGetLastError());                                                                                        \
    return POLS_OK;

int dyn_scan_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a) { DYN_LAUNCH(dyn_scan_kernel) }
int dyn_rewrite_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a) { DYN_LAUNCH(dyn_rewrite_kernel) }
int dyn_post_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a) { DYN_LAUNCH(dyn_post_kernel) }

}  // namespace pols
