// dyn_prep.hip -- see dyn_prep.hpp.  HBM-bound element-wise passes: one thread per row walks the columns (consecutive lanes =
// consecutive rows of every column: coalesced), nothing to tile.
#include "dyn_prep.hpp"
#include "k4_rolling.hpp"

#include <algorithm>

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
    // the host only asks "any?": plain stores of the same value.  (Counting them with one atomicAdd per wave serialised ~150 000 waves
    // on two addresses: 1.53 ms of a 10M-row frame with 3 % nulls, against the 0.1 ms the pass takes to read the frame.)
    const unsigned long long bi = __ballot(invalid), bn = __ballot(nulls);
    if ((threadIdx.x & 63) == 0) {
        if (bi) a.flags[0] = 1;
        if (bn) a.flags[1] = 1;
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
    for (int64_t r = s + (int64_t)blockIdx.y * 256 + threadIdx.x; r < e; r += (int64_t)gridDim.y * 256) {   // (long groups: several workgroups)
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
    const unsigned ny = (unsigned)std::max(1, std::min(1024, a.row_blocks));
    if (dtype == POLS_F32) hipLaunchKernelGGL(mt_predict_kernel<float>, dim3((unsigned)a.n_groups, ny), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(mt_predict_kernel<double>, dim3((unsigned)a.n_groups, ny), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ---------------------------------------------------------------- row compaction (rolling OLS, drop family, frames with nulls)
constexpr int RC_SLAB = 256;

// valid rows of this workgroup's slab at or below each thread's row (inclusive), and the slab's total
__device__ __forceinline__ unsigned rc_slab_prefix(bool ok, unsigned *wave_cnt, unsigned *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = 0, tot = 0;
    for (int w = 0; w < RC_SLAB / 64; ++w) { before += w < wave ? wave_cnt[w] : 0u; tot += wave_cnt[w]; }
    *total = tot;
    return before + (unsigned)__popcll(bal & ((2ull << lane) - 1ull));
}

__global__ void __launch_bounds__(RC_SLAB) rc_count_kernel(const RowCompactArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    unsigned total;
    rc_slab_prefix(r < a.n_rows && a.valid[r], wave_cnt, &total);
    if (threadIdx.x == 0) a.slab_cnt[blockIdx.x] = total;
}

// Exclusive prefix over the slabs, one workgroup per 1 024 of them (see rm_blk_kernel / rm_scan_kernel below: one workgroup looping over the
// 39 000 slabs of a 10M-row frame took 55 us, the two launches of ~40 workgroups take 3 + 5).
__global__ void __launch_bounds__(1024) rc_blk_kernel(const RowCompactArgs a) {
    __shared__ long long part[1024 / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    long long v = s < a.n_slabs ? (long long)a.slab_cnt[s] : 0;
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int w = 0; w < 1024 / 64; ++w) t += part[w];
        a.blk_cnt[blockIdx.x] = (unsigned long long)t;
    }
}

__global__ void __launch_bounds__(1024) rc_scan_kernel(const RowCompactArgs a) {
    __shared__ long long part[1024 / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long before = 0;                                    // the blocks below this one
    for (int64_t b2 = lane; b2 < (int64_t)blockIdx.x; b2 += 64) before += (long long)a.blk_cnt[b2];
    for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off);
    const int64_t s = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const long long v = s < a.n_slabs ? (long long)a.slab_cnt[s] : 0;
    long long incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const long long up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) before += part[w];
    if (s < a.n_slabs) a.slab_base[s] = before + incl - v;
    if (s == a.n_slabs - 1) a.slab_base[a.n_slabs] = before + incl;
}

__global__ void __launch_bounds__(256) rc_groups_kernel(const RowCompactArgs a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g > a.n_groups) return;
    const int64_t r0 = a.offs[g], s = r0 / RC_SLAB;
    int64_t c = a.slab_base[s < a.n_slabs ? s : a.n_slabs];
    if (s < a.n_slabs) {
        int64_t i = s * RC_SLAB;
        if ((reinterpret_cast<uintptr_t>(a.valid) & 3) == 0) {                          // four validity bytes (0 / 1) per load
            for (; i + 4 <= r0; i += 4) {
                const unsigned v4 = *reinterpret_cast<const unsigned *>(a.valid + i);
                c += ((v4 & 0xffu) != 0) + ((v4 & 0xff00u) != 0) + ((v4 & 0xff0000u) != 0) + ((v4 & 0xff000000u) != 0);
            }
        }
        for (; i < r0; ++i) c += a.valid[i] ? 1 : 0;
    }
    a.c_offs[g] = c;
}

// slab_gfirst[s] = compacted offset of the group that holds the slab's first row, for slabs that start INSIDE a group -- one thread per
// slab (a binary search of the offsets) instead of one thread writing every slab of a long group (39 000 stores for a 10 M-row group)
__global__ void __launch_bounds__(256) rc_gfirst_kernel(const RowCompactArgs a) {
    const int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sl >= a.n_slabs) return;
    const int64_t r0 = sl * RC_SLAB;
    int64_t lo = 0, hi = a.n_groups;                         // the last g with offs[g] <= r0
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.offs[mid] <= r0) lo = mid; else hi = mid;
    }
    if (a.offs[lo] < r0 && r0 < a.offs[lo + 1]) a.slab_gfirst[sl] = a.c_offs[lo];
}

template <typename T>
__global__ void __launch_bounds__(RC_SLAB) rc_mask_kernel(const RowCompactArgs a) {       // one thread per row: coalesced down every column
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    if (r >= a.n_rows) return;
    bool ok = !a.valid_in || a.valid_in[r];
    if (a.drop)
        for (int c = 0; c < a.n_mask; ++c) { const T v = static_cast<const T *>(a.in[c])[r]; ok = ok && (v == v); }
    a.valid_out[r] = ok ? 1 : 0;
}

template <typename T>
__global__ void __launch_bounds__(RC_SLAB) rc_scatter_kernel(const RowCompactArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    const bool ok = r < a.n_rows && a.valid[r];
    unsigned total;
    const unsigned incl = rc_slab_prefix(ok, wave_cnt, &total);
    if (ok) {
        const int64_t pos = a.slab_base[blockIdx.x] + incl - 1;                          // stable: the rows keep their order
        for (int c = 0; c < a.n_cols; ++c) {
            T v = static_cast<const T *>(a.in[c])[r];
            if (v != v) v = (c == a.w_col) ? (T)1e-24 : (a.zero_fill ? T(0) : v);
            static_cast<T *>(a.out[c])[pos] = v;
        }
    }
}

// The source map instead of the compacted columns: 4 bytes per valid row written where the scatter copies k + 1 columns
__global__ void __launch_bounds__(RC_SLAB) rc_srcmap_kernel(const RowCompactArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    const bool ok = r < a.n_rows && a.valid[r];
    unsigned total;
    const unsigned incl = rc_slab_prefix(ok, wave_cnt, &total);
    if (ok) a.src[a.slab_base[blockIdx.x] + incl - 1] = (int32_t)r;
}

template <typename T>
__global__ void __launch_bounds__(RC_SLAB) rc_expand_kernel(const RowCompactArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    __shared__ int wave_flag[RC_SLAB / 64];                   // the wave's last sequence start (thread index) or -1
    __shared__ unsigned s_incl[RC_SLAB];
    __shared__ long long s_src[RC_SLAB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * RC_SLAB, r = r0 + threadIdx.x;
    const bool in = r < a.n_rows;
    const bool ok = in && a.valid[r];
    const bool st = in && a.start[r];
    unsigned total;
    const unsigned incl = rc_slab_prefix(ok, wave_cnt, &total);
    s_incl[threadIdx.x] = incl;
    const unsigned long long fw = __ballot(st);
    const unsigned long long fb = fw & ((2ull << lane) - 1ull);                          // sequence starts at or below this lane
    int last = fb ? wave * 64 + (63 - __clzll(fb)) : -1;
    if (lane == 0) wave_flag[wave] = fw ? wave * 64 + (63 - __clzll(fw)) : -1;
    __syncthreads();
    for (int w = wave - 1; w >= 0 && last < 0; --w) last = wave_flag[w];
    // compacted index of the first valid row of this row's sequence (= valid rows of the frame before the sequence's first row)
    const long long base = a.slab_base[blockIdx.x];
    long long gfirst;
    if (last >= 0) gfirst = base + (long long)s_incl[last] - (a.valid[r0 + last] ? 1 : 0);
    else gfirst = a.slab_gfirst[blockIdx.x];
    const long long cidx = base + (long long)incl - 1;       // the last valid row at or before this one
    const long long src = (in && cidx >= gfirst) ? cidx : -1; // none yet in this sequence: NaN (ls.rs:864)
    s_src[threadIdx.x] = src;
    __syncthreads();
    const T qnan = nan_if<T>(1u, T(0));
    const int k = a.k;
    const int64_t rows = a.n_rows - r0 < RC_SLAB ? a.n_rows - r0 : RC_SLAB;
    if (a.coef) {                                             // whole lines: consecutive threads write consecutive values
        T *dst = static_cast<T *>(a.coef) + r0 * k;
        const T *cc = static_cast<const T *>(a.coef_c);
        for (int64_t e = threadIdx.x; e < rows * k; e += RC_SLAB) {
            const int rr = (int)(e / k), j = (int)(e - (int64_t)rr * k);
            const long long sidx = s_src[rr];
            dst[e] = sidx >= 0 ? cc[sidx * k + j] : qnan;
        }
    }
    if (a.pred && in) {
        T p = qnan;
        if (src >= 0) {
            const T *cc = static_cast<const T *>(a.coef_c) + src * k;
            p = T(0);
            for (int j = 0; j < k; ++j) p = fma(static_cast<const T *>(a.in[1 + j])[r], cc[j], p);    // (in[0] is the target)
        }
        static_cast<T *>(a.pred)[r] = p;
    }
}

// ---------------------------------------------------------------- validity prefix of the chunk kernels, on the device
__global__ void __launch_bounds__(RC_SLAB) vt_rows_kernel(const ValidTablesArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    const bool in = r < a.n_rows;
    const bool ok = in && a.valid[r];
    unsigned total;
    const unsigned incl = rc_slab_prefix(ok, wave_cnt, &total);
    if (!in) return;
    int64_t lo = 0, hi = a.n_groups;                         // the group holding row r: the last g with offs[g] <= r
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.offs[mid] <= r) lo = mid; else hi = mid;
    }
    const int64_t g0 = a.offs[lo];
    const int64_t c = a.slab_base[blockIdx.x] + (int64_t)incl - a.c_offs[lo];   // valid rows of the group at or before r
    a.cnt[r] = (int32_t)c;
    if (ok) a.vidx[g0 + c - 1] = (int32_t)(r - g0);
}

__global__ void __launch_bounds__(256) vt_groups_kernel(const ValidTablesArgs a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_groups) return;
    K4Group *G = static_cast<K4Group *>(a.groups) + g;
    const int64_t tot = a.c_offs[g + 1] - a.c_offs[g], mp = a.min_periods;
    // ls.rs:881-891: min_periods_valid = the row at which the min_periods-th valid observation arrives (else it stays min_periods),
    // n_valid = the valid rows counted until then
    G->mpv = tot >= mp ? (int64_t)a.vidx[a.offs[g] + mp - 1] + 1 : mp;
    G->gate_n = tot < mp ? tot : mp;
}

int valid_tables_launch(pols_ctx *ctx, const ValidTablesArgs &a) {
    if (a.n_rows == 0 || a.n_groups == 0) return POLS_OK;
    POLS_HIP(hipMemsetAsync(a.vidx, 0xff, sizeof(int32_t) * (size_t)a.n_rows, ctx->stream));
    hipLaunchKernelGGL(vt_rows_kernel, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    hipLaunchKernelGGL(vt_groups_kernel, dim3((unsigned)((a.n_groups + 255) / 256)), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ---------------------------------------------------------------- per-row solve table of the masked tile kernel ("drop_window" with nulls, K4c)
// Which rows does solve_rolling_ols solve under the fixed window (ls.rs:987-1029)?  Row i of a sequence is NaN before the warm-up row
// mpv - 1 (:864, :939-943), solved when it is that row, or when the state changed (row i is valid, or row i - window is and leaves) and
// the window holds gate_n valid rows (n_valid_window >= n_valid, :1013 / :1022: the valid rows among (i - window, i], from row 1 on while
// i < window -- the saturating_sub of :990), and repeats the last solved row's coefficients otherwise.  All of it is a function of the
// validity bytes.  P(r) = valid rows of the FRAME at or before row r = slab_base[r / 256] + incl[r] (a 16-bit count inside the slab):
//   rm_count_kernel   slab totals + incl[]            (then rm_scan_kernel<false>: slab_base)
//   rm_groups_kernel  per sequence mpv / gate_n, and the flag for the one shape the tile kernel cannot take
//   rm_rows_kernel    every row: solved / NaN / where the last solved row of its slab is;   rm_scan_kernel<true>: ... of the slabs before
__global__ void __launch_bounds__(RC_SLAB) rm_count_kernel(const RollMaskArgs a) {
    __shared__ unsigned wave_cnt[RC_SLAB / 64];
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    unsigned total;
    const unsigned incl = rc_slab_prefix(r < a.n_rows && a.valid[r], wave_cnt, &total);
    if (r < a.n_rows) a.incl[r] = (uint16_t)incl;
    if (threadIdx.x == 0) a.slab_cnt[blockIdx.x] = total;
}

__device__ __forceinline__ int64_t rm_prefix(const RollMaskArgs &a, int64_t r) { return a.slab_base[r >> 8] + (int64_t)a.incl[r]; }

__global__ void __launch_bounds__(256) rm_groups_kernel(const RollMaskArgs a) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= a.n_groups) return;
    const int64_t s = a.offs[g], n = a.offs[g + 1] - s, mp = a.min_periods;
    const int64_t before = s > 0 ? rm_prefix(a, s - 1) : 0, tot = n > 0 ? rm_prefix(a, s + n - 1) - before : 0;   // valid rows before / inside the sequence
    // ls.rs:881-891: min_periods_valid = the row at which the min_periods-th valid observation arrives (else it stays min_periods):
    // the first row of the sequence whose prefix reaches `before + mp` (the prefix is monotone: a binary search)
    int64_t mpv = mp;
    if (tot >= mp) {
        int64_t lo = 0, hi = n - 1;                          // P(s + hi) - before >= mp holds at hi = n - 1
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rm_prefix(a, s + mid) - before >= mp) hi = mid; else lo = mid + 1;
        }
        mpv = lo + 1;
    }
    const int64_t gate = tot < mp ? tot : mp;
    if (n < mp) mpv = n + 1;                                 // :893-900: every row NaN
    a.g_mpv[g] = mpv;
    a.g_gate[g] = (int32_t)gate;
    // a valid row older than the window when the warm-up ends is never subtracted (the sliding loop starts at row mpv, :989): such a
    // sequence is outside the tile kernel's prefix-difference form -- the caller routes the frame to the chunk kernels
    const int64_t jm = mpv - a.window - 1;
    if (n >= mp && jm >= 0 && jm < n && rm_prefix(a, s + jm) - before > 0) atomicOr(a.flag, 1);
}

// The group holding every slab's first row (the last g with offs[g] <= row): a thread per slab, all searches side by side.  Inside rm_rows_kernel
// -- thread 0 of every workgroup, 14 dependent loads in front of everything else -- the search was most of that kernel's 77 us on 10M rows.
// (the table lives in slab_carry until rm_scan_kernel<true> writes the carries there, behind rm_rows_kernel)
__global__ void __launch_bounds__(256) rm_slab_group_kernel(const RollMaskArgs a) {
    const int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (sl >= a.n_slabs) return;
    const int64_t r0 = sl * RC_SLAB;
    int64_t lo = 0, hi = a.n_groups;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.offs[mid] <= r0) lo = mid; else hi = mid;
    }
    a.slab_carry[sl] = lo;
}

__global__ void __launch_bounds__(RC_SLAB) rm_rows_kernel(const RollMaskArgs a) {
    __shared__ int wave_last[RC_SLAB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * RC_SLAB + threadIdx.x;
    const bool in = r < a.n_rows;
    const int64_t g_first = a.slab_carry[blockIdx.x];        // (rm_slab_group_kernel)
    bool solved = false, nan = true;
    if (in) {
        int64_t g = g_first;
        while (g + 1 < a.n_groups && a.offs[g + 1] <= r) ++g;    // (groups that start inside the slab: a short walk)
        const int64_t s = a.offs[g], i = r - s, mpv = a.g_mpv[g], w = a.window;
        if (i >= mpv - 1) {
            nan = false;
            const int64_t is = i >= w ? i - w : 0;
            // :1007-1026: the state changes when row i is valid (it enters) or row i - window is (it leaves); only then is the gate consulted
            const bool changed = a.valid[r] != 0 || (i >= w && a.valid[r - w] != 0);
            solved = (i == mpv - 1) || (changed && rm_prefix(a, r) - rm_prefix(a, s + is) >= (int64_t)a.g_gate[g]);
        }
    }
    // the last solved row of the slab at or before this one: highest set bit of the wave's ballot below the lane, else an earlier wave's last
    const unsigned long long bal = __ballot(solved);
    if (lane == 0) wave_last[wave] = bal ? wave * 64 + (63 - __clzll(bal)) : -1;
    __syncthreads();
    const unsigned long long upto = bal & ((2ull << lane) - 1ull);
    int loc = upto ? wave * 64 + (63 - __clzll(upto)) : -1;
    for (int w2 = wave - 1; w2 >= 0 && loc < 0; --w2) loc = wave_last[w2];
    // 0: solved here; 1: NaN row; 2 + j: repeats row j of its slab (j < 256); 2 + 256: repeats a row before the slab (slab_carry)
    if (in) a.code[r] = (uint16_t)(solved ? 0 : nan ? 1 : 2 + (loc >= 0 ? loc : RC_SLAB));
    if (in) a.solved[r] = solved ? 1 : 0;
    if (threadIdx.x == RC_SLAB - 1) a.slab_last[blockIdx.x] = loc >= 0 ? (int64_t)blockIdx.x * RC_SLAB + loc : -1;
}

// Exclusive prefix over the slabs, one workgroup per 1 024 of them: rm_blk_kernel reduces every block of 1 024 slabs to one value, rm_scan_kernel
// sums the blocks below its own (a few dozen values) and scans its block in LDS -- two launches of ~40 workgroups (3 + 5 us) instead of one
// workgroup looping over 39 000 slabs (55 us on a 10M-row frame; per-block totals by atomics from the producing kernels: 39 000 atomics on 39
// words, 250 us).
template <bool MAXSCAN>
__global__ void __launch_bounds__(1024) rm_blk_kernel(const RollMaskArgs a) {
    __shared__ long long part[1024 / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    long long v = s < a.n_slabs ? (MAXSCAN ? (long long)a.slab_last[s] + 1 : (long long)a.slab_cnt[s]) : 0;    // (MAXSCAN: row + 1, 0 = none)
    for (int off = 32; off >= 1; off >>= 1) { const long long o = __shfl_xor(v, off); v = MAXSCAN ? (v > o ? v : o) : v + o; }
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = part[0];
        for (int w = 1; w < 1024 / 64; ++w) t = MAXSCAN ? (t > part[w] ? t : part[w]) : t + part[w];
        (MAXSCAN ? a.blk_last : a.blk_cnt)[blockIdx.x] = (unsigned long long)t;
    }
}

// MAXSCAN = false: slab_base[s] = valid rows before slab s (slab_base[n_slabs] = all);  true: slab_carry[s] = the last solved row before slab s.
template <bool MAXSCAN>
__global__ void __launch_bounds__(1024) rm_scan_kernel(const RollMaskArgs a) {
    __shared__ long long part[1024 / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto op = [](long long x, long long y) { return MAXSCAN ? (x > y ? x : y) : x + y; };
    const long long ident = MAXSCAN ? -1 : 0;
    long long before = ident;                                // the blocks below this one
    for (int64_t b2 = lane; b2 < (int64_t)blockIdx.x; b2 += 64)
        before = op(before, MAXSCAN ? (long long)a.blk_last[b2] - 1 : (long long)a.blk_cnt[b2]);
    for (int off = 32; off >= 1; off >>= 1) before = op(before, __shfl_xor(before, off));
    const int64_t s = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const long long v = s < a.n_slabs ? (MAXSCAN ? (long long)a.slab_last[s] : (long long)a.slab_cnt[s]) : ident;
    long long incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const long long up = __shfl_up(incl, off);
        if (lane >= off) incl = op(incl, up);
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) before = op(before, part[w]);
    const long long up1 = __shfl_up(incl, 1);
    const long long excl = lane > 0 ? op(before, up1) : before;
    if (s < a.n_slabs) { if (MAXSCAN) a.slab_carry[s] = excl; else a.slab_base[s] = excl; }
    if (!MAXSCAN && s == a.n_slabs - 1) a.slab_base[a.n_slabs] = op(before, incl);
}

// The fill pass behind the masked tile kernel: a row the reference does not solve takes NaN (before the warm-up) or the coefficients of the
// last solved row (a fixed point of this pass: solved rows are never written), and its prediction is recomputed from them (NaN on a
// masked row, src/expressions.rs:695-700).  Reads one byte per row; everything else only for the rows it rewrites.
template <typename T>
__global__ void __launch_bounds__(256) rm_fill_kernel(const RollMaskArgs a) {
    const int64_t r4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;     // four rows per thread: one 32-bit load of their solved bytes
    if (r4 >= a.n_rows) return;
    if (r4 + 4 <= a.n_rows && *reinterpret_cast<const unsigned *>(a.solved + r4) == 0x01010101u) return;   // all four solved here
    const int k = a.k;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    const T qnan = nan_if<T>(1u, T(0));
    for (int64_t r = r4; r < r4 + 4 && r < a.n_rows; ++r) {
        if (a.solved[r]) continue;
        const int code = a.code[r];
        const int64_t slab0 = r & ~(int64_t)(RC_SLAB - 1);
        if (code == 1) {
            if (coef) for (int j = 0; j < k; ++j) coef[r * k + j] = qnan;
            if (pred) pred[r] = qnan;
            continue;
        }
        const int64_t src = code - 2 < RC_SLAB ? slab0 + (code - 2) : a.slab_carry[r >> 8];
        const bool vr = a.valid[r] != 0;
        T c[10], xv[10];                                      // every load of the row issued before the first use (k <= 10)
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            c[j] = (j < k && coef) ? coef[src * k + j] : T(0);
            xv[j] = (j < k && vr) ? static_cast<const T *>(a.x[j])[r] : T(0);
        }
        T p = T(0);
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (j < k) { if (coef) coef[r * k + j] = c[j]; p = fma(xv[j], c[j], p); }
        }
        if (pred) pred[r] = vr ? p : qnan;
    }
}

int roll_mask_tables_launch(pols_ctx *ctx, const RollMaskArgs &a) {
    if (a.n_rows == 0 || a.n_groups == 0) return POLS_OK;
    static_assert(RC_SLAB == 256, "the slab arithmetic of rm_prefix / rm_fill_kernel");
    POLS_HIP(hipMemsetAsync(a.flag, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(rm_count_kernel, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_blk_kernel<false>, dim3((unsigned)((a.n_slabs + 1023) / 1024)), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_scan_kernel<false>, dim3((unsigned)((a.n_slabs + 1023) / 1024)), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_groups_kernel, dim3((unsigned)((a.n_groups + 255) / 256)), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int roll_mask_rows_launch(pols_ctx *ctx, const RollMaskArgs &a) {
    if (a.n_rows == 0 || a.n_groups == 0) return POLS_OK;
    hipLaunchKernelGGL(rm_slab_group_kernel, dim3((unsigned)((a.n_slabs + 255) / 256)), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_rows_kernel, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_blk_kernel<true>, dim3((unsigned)((a.n_slabs + 1023) / 1024)), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(rm_scan_kernel<true>, dim3((unsigned)((a.n_slabs + 1023) / 1024)), dim3(1024), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int roll_mask_fill_launch(pols_ctx *ctx, int dtype, const RollMaskArgs &a) {
    if (a.n_rows == 0 || (!a.coef && !a.pred)) return POLS_OK;
    const dim3 grid((unsigned)((a.n_rows + 1023) / 1024));
    if (dtype == POLS_F32) hipLaunchKernelGGL(rm_fill_kernel<float>, grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(rm_fill_kernel<double>, grid, dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int row_compact_mask_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a) {
    if (a.n_rows == 0) return POLS_OK;
    if (dtype == POLS_F32) hipLaunchKernelGGL(rc_mask_kernel<float>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    else hipLaunchKernelGGL(rc_mask_kernel<double>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
int row_compact_offsets_launch(pols_ctx *ctx, const RowCompactArgs &a0) {
    if (a0.n_rows == 0) return POLS_OK;
    RowCompactArgs a = a0;
    const unsigned n_blk = (unsigned)((a.n_slabs + 1023) / 1024);
    void *bc = nullptr;
    int rc = ensure_scratch(ctx, 26, sizeof(unsigned long long) * (size_t)n_blk, &bc);    // totals of every 1 024 slabs
    if (rc) return rc;
    a.blk_cnt = static_cast<unsigned long long *>(bc);
    hipLaunchKernelGGL(rc_count_kernel, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    hipLaunchKernelGGL(rc_blk_kernel, dim3(n_blk), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(rc_scan_kernel, dim3(n_blk), dim3(1024), 0, ctx->stream, a);
    hipLaunchKernelGGL(rc_groups_kernel, dim3((unsigned)((a.n_groups + 1 + 255) / 256)), dim3(256), 0, ctx->stream, a);
    if (a.slab_gfirst) hipLaunchKernelGGL(rc_gfirst_kernel, dim3((unsigned)((a.n_slabs + 255) / 256)), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
int row_compact_scatter_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a) {
    if (a.n_rows == 0) return POLS_OK;
    if (dtype == POLS_F32) hipLaunchKernelGGL(rc_scatter_kernel<float>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    else hipLaunchKernelGGL(rc_scatter_kernel<double>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
int row_compact_srcmap_launch(pols_ctx *ctx, const RowCompactArgs &a) {
    if (a.n_rows == 0) return POLS_OK;
    hipLaunchKernelGGL(rc_srcmap_kernel, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
int row_compact_expand_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a) {
    if (a.n_rows == 0) return POLS_OK;
    if (dtype == POLS_F32) hipLaunchKernelGGL(rc_expand_kernel<float>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
    else hipLaunchKernelGGL(rc_expand_kernel<double>, dim3((unsigned)a.n_slabs), dim3(RC_SLAB), 0, ctx->stream, a);
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
