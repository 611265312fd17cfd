// k4w_wide.hip -- rolling OLS (K4) and the RLS decayed scan (K3s) for 9..32 features: ONE WAVE PER CHUNK.
//
// Same three passes and the same reference semantics as k4_rolling.hip (solve_rolling_ols, src/least_squares.rs:848-1032;
// RecursiveLeastSquares, :505-545 in information form), but the (X'X, X'y) state of K features no longer fits one lane's
// registers (K (K + 3) / 2 = 560 doubles at K = 32), so a chunk is walked by a whole wave with the state in LDS:
//   * the K x K matrix is stored full and row-major, X'y behind it; a row update touches two matrix rows per step
//     (lanes 0-31 / 32-63), the column pointers are read once per lane;
//   * the per-row solve is the wave-cooperative Cholesky of k5_enet.hip::gram_solve (lane i owns row i of L, odd row
//     stride against bank conflicts), then two shuffle-driven triangular solves; a non-positive pivot takes the LU
//     with partial pivoting, like solve_normal_equations(.., None, Some(LU)) (:732-734 / :277-337);
//   * coefficients live one per lane, so a coefficient row is one coalesced store and the prediction one wave sum.
// The control flow below is k4_walk_kernel's / k3s_walk_kernel's, statement for statement; only the state container
// differs.  All arithmetic is f64.
#include "k4_rolling.hpp"

namespace pols {

constexpr int KW_MAX = 32;

template <typename T>
struct WCtx {
    const K4Args &a;
    int64_t s;
    int first_chunk;
    int K, KK, NS, LP, lane;
    double *S0, *S1, *L, *xs, *rinv, *rv, *xsol;
    const T *mycol;      // lane < K: feature column `lane`; lane == K: the target
    __device__ WCtx(const K4Args &a_, int64_t s_, int fc, double *lds) : a(a_), s(s_), first_chunk(fc) {
        K = a.k; KK = K * K; NS = KK + K; LP = K | 1; lane = threadIdx.x;
        S0 = lds; S1 = S0 + NS; L = S1 + NS; xs = L + K * LP; rinv = xs + KW_MAX + 2; rv = rinv + KW_MAX; xsol = rv + KW_MAX;
        mycol = lane < K ? static_cast<const T *>(a.x[lane]) : static_cast<const T *>(a.y);
    }
    static __host__ __device__ size_t lds_doubles(int K) { return 2 * (size_t)(K * K + K) + (size_t)K * (K | 1) + 4 * KW_MAX + 2; }

    __device__ __forceinline__ bool valid(int64_t i) const { return a.valid ? a.valid[s + i] != 0 : true; }
    __device__ __forceinline__ int64_t cnt(int64_t i) const { return a.cnt ? (int64_t)a.cnt[s + i] : i + 1; }
    __device__ __forceinline__ int64_t vidx(int64_t r) const { return a.vidx ? (int64_t)a.vidx[s + r] : r; }
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

    __device__ __forceinline__ void load_row(int64_t i) const {      // xs[0..K) = x, xs[K] = y
        if (lane <= K) xs[lane] = (double)mycol[s + i];
        sync();
    }
    __device__ __forceinline__ void zero(double *S) const {
        for (int q = lane; q < NS; q += 64) S[q] = 0.0;
        sync();
    }
    __device__ __forceinline__ void scale(double *S, double f) const {
        for (int q = lane; q < NS; q += 64) S[q] *= f;
        sync();
    }
    __device__ __forceinline__ void axpy(double *S, const double *src, double sign) const {
        for (int q = lane; q < NS; q += 64) S[q] += sign * src[q];
        sync();
    }
    // S += sign * [x x', x y] with the row already in xs   (outer_product :600-607, update :714-723)
    __device__ __forceinline__ void add_loaded(double *S, double sign) const {
        const int half = lane >> 5, c = lane & 31;
        if (c < K) {
            const double xc = xs[c];
            for (int p = half; p < K; p += 2) S[p * K + c] += sign * (xs[p] * xc);
        }
        if (lane < K) S[KK + lane] += sign * (xs[lane] * xs[K]);
        sync();
    }
    __device__ __forceinline__ void add_row(double *S, int64_t i, double sign) const { load_row(i); add_loaded(S, sign); }
    // P(i): sum over the valid rows 0..i of the group (i < 0 -> 0), from the chunk prefix + a partial chunk
    __device__ __forceinline__ void prefix(int64_t i, double *P, double sign, int nacc) const {
        if (i < 0) return;
        const int64_t c = i / a.chunk_len;
        axpy(P, a.totals + (size_t)(first_chunk + c) * nacc, sign);
        for (int64_t j = c * a.chunk_len; j <= i; ++j)
            if (valid(j)) add_row(P, j, sign);
    }

    // beta = (S_xx + alpha I)^-1 S_xy; returned one coefficient per lane (lanes >= K: 0)
    __device__ double solve(const double *S, double alpha) const {
        const int half = lane >> 5, c = lane & 31;
        if (c < K)
            for (int p = half; p < K; p += 2) L[p * LP + c] = S[p * K + c] + (p == c ? alpha : 0.0);
        double bi = (lane < K) ? S[KK + lane] : 0.0;
        sync();
        bool ok = true;
        for (int j = 0; j < K; ++j) {
            double d = L[j * LP + j];
            for (int p = 0; p < j; ++p) d -= L[j * LP + p] * L[j * LP + p];
            ok = ok && (d > 0.0);
            const double ri = 1.0 / sqrt(d);
            if (lane == 0) rinv[j] = ri;
            if (lane > j && lane < K) {
                double acc = L[lane * LP + j];
                for (int p = 0; p < j; ++p) acc -= L[lane * LP + p] * L[j * LP + p];
                L[lane * LP + j] = acc * ri;
            }
            sync();
        }
        if (ok) {
            for (int p = 0; p < K; ++p) {                      // forward: t = L^-1 b
                if (lane == p) bi *= rinv[p];
                const double tp = __shfl(bi, p);
                if (lane > p && lane < K) bi -= L[lane * LP + p] * tp;
            }
            for (int p = K - 1; p >= 0; --p) {                 // backward: beta = L^-T t
                if (lane == p) bi *= rinv[p];
                const double bp = __shfl(bi, p);
                if (lane < p) bi -= L[p * LP + lane] * bp;
            }
            return bi;
        }
        // ---- LU with partial pivoting on a fresh copy
        if (c < K)
            for (int p = half; p < K; p += 2) L[p * LP + c] = S[p * K + c] + (p == c ? alpha : 0.0);
        if (lane < K) rv[lane] = S[KK + lane];
        sync();
        for (int j = 0; j < K; ++j) {
            double v = (lane >= j && lane < K) ? fabs(L[lane * LP + j]) : -1.0;
            int idx = lane;
            for (int off = 32; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(v, off);
                const int oi = __shfl_xor(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            const int p = idx;
            if (p != j && p < K) {
                if (lane < K) { const double t = L[j * LP + lane]; L[j * LP + lane] = L[p * LP + lane]; L[p * LP + lane] = t; }
                if (lane == 0) { const double t = rv[j]; rv[j] = rv[p]; rv[p] = t; }
            }
            sync();
            const double d = L[j * LP + j];
            if (lane > j && lane < K) {
                const double f = L[lane * LP + j] / d;
                for (int q = j + 1; q < K; ++q) L[lane * LP + q] -= f * L[j * LP + q];
                rv[lane] -= f * rv[j];
            }
            sync();
        }
        if (lane == 0) {
            for (int i = K - 1; i >= 0; --i) {
                double sacc = rv[i];
                for (int q = i + 1; q < K; ++q) sacc -= L[i * LP + q] * xsol[q];
                xsol[i] = sacc / L[i * LP + i];
            }
        }
        sync();
        return (lane < K) ? xsol[lane] : 0.0;
    }

    // write the coefficient row / prediction of relative row i
    __device__ __forceinline__ void store(int64_t i, double last, T *coef, T *pred) const {
        const int64_t row = s + i;
        if (coef && lane < K) coef[row * K + lane] = (T)last;
        if (pred) {                                            // (features * coefficients).sum_axis(1)  (ex.rs:184)
            load_row(i);
            const double p = wave_sum_row3((lane < K) ? xs[lane] * last : 0.0);
            if (lane == 63) pred[row] = (T)p;
        }
    }
};

// ------------------------------------------------------------------ pass 1: per-chunk totals (decayed for the RLS scan)
template <typename T, bool RLS>
__global__ void __launch_bounds__(64) kw_totals_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    WCtx<T> cx(a, G.start, G.first_chunk, lds);
    const int nacc = cx.NS + (RLS ? 1 : 0);
    double *S = cx.S0;
    cx.zero(S);
    double decay = 1.0;
    for (int64_t i = ch.t0 - G.start; i < ch.t1 - G.start; ++i)
        if (cx.valid(i)) {
            if (RLS) { cx.scale(S, a.ff); decay *= a.ff; }
            cx.add_row(S, i, 1.0);
        }
    for (int q = cx.lane; q < cx.NS; q += 64) a.totals[(size_t)c * nacc + q] = S[q];
    if (RLS && cx.lane == 0) a.totals[(size_t)c * nacc + cx.NS] = decay;
}

// ------------------------------------------------------------------ pass 3: rolling walk (k4_walk_kernel, wave form)
template <typename T>
__global__ void __launch_bounds__(64) kw_rolling_walk_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    WCtx<T> cx(a, G.start, G.first_chunk, lds);
    const int K = cx.K, nacc = cx.NS;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    const int64_t w = a.window, mpv = G.mpv;
    const bool drop = a.drop_mode != 0;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    if (G.all_nan) {                                           // :893-900
        for (int64_t i = rel0; i < rel1; ++i) {
            if (coef && cx.lane < K) coef[(G.start + i) * K + cx.lane] = (T)qnan;
            if (pred && cx.lane == 0) pred[G.start + i] = (T)qnan;
        }
        return;
    }
    const int64_t j_min = drop ? 0 : max(mpv - w, (int64_t)0);
    // min_periods > window under the drop family (ls.rs:869-876 only warns): the warm-up sums c0 > w valid rows but its deque
    // keeps the FIRST w of them (:917-919), so the first w slides pop ranks 0 .. w-1 (rank R - c0 leaves as rank R enters) and
    // ranks [w, c0) are never subtracted at all; afterwards the deque is an ordinary w-row window again.
    const int64_t c0 = drop ? cx.cnt(mpv - 1) : 0;
    const bool longwarm = drop && c0 > w;
    auto old_of = [&](int64_t i) -> int64_t {
        if (!drop) return i - w;
        const int64_t R = cx.cnt(i) - 1;
        const int64_t r = (longwarm && R - c0 < w) ? R - c0 : R - w;
        return r < 0 ? -1 : cx.vidx(r);
    };
    auto gate = [&](int64_t i) -> bool {                       // n_valid_window >= n_valid (:994-997, 1013, 1022)
        const int64_t i_start = i >= w ? i - w : 0;
        return cx.cnt(i) - cx.cnt(i_start) >= G.gate_n;
    };
    auto state_at = [&](int64_t i, double *S) {
        cx.zero(S);
        cx.prefix(i, S, 1.0, nacc);
        const int64_t o = (i >= 0) ? old_of(i) : -1;
        if (o >= j_min && i >= mpv) {
            cx.prefix(o, S, -1.0, nacc);
            cx.prefix(j_min - 1, S, 1.0, nacc);
        }
        if (longwarm && i >= mpv && cx.cnt(i) - 1 - c0 >= w) {  // the warm-up rows the deque never held stay in the sums
            cx.prefix(cx.vidx(c0 - 1), S, 1.0, nacc);
            cx.prefix(cx.vidx(w - 1), S, -1.0, nacc);
        }
    };

    double *S = cx.S0;
    double last = qnan;
    state_at(rel0 - 1, S);
    int64_t prev_old = (rel0 > 0 && rel0 - 1 >= mpv) ? old_of(rel0 - 1) : -1;
    if (rel0 >= mpv && rel0 > 0) {
        if (drop || rel0 - 1 == mpv - 1 || gate(rel0 - 1)) {
            last = cx.solve(S, a.alpha);
        } else {
            for (int64_t ip = rel0 - 2; ip >= mpv - 1; --ip) {
                if (ip == mpv - 1 || gate(ip)) {
                    state_at(ip, cx.S1);
                    last = cx.solve(cx.S1, a.alpha);
                    break;
                }
            }
        }
    }
    for (int64_t i = rel0; i < rel1; ++i) {
        const bool v = cx.valid(i);
        if (v) cx.add_row(S, i, 1.0);
        if (i >= mpv && (v || !drop)) {
            const int64_t no = old_of(i);
            if (drop) {
                if (no != prev_old && no >= 0) cx.add_row(S, no, -1.0);
            } else if (no >= j_min && no >= 0 && cx.valid(no)) {
                cx.add_row(S, no, -1.0);
            }
            prev_old = no;
        }
        if (i >= mpv - 1) {
            const bool do_solve = (i == mpv - 1) || (drop ? v : gate(i));
            if (do_solve) last = cx.solve(S, a.alpha);
        }
        cx.store(i, last, coef, pred);
    }
}

// ------------------------------------------------------------------ pass 3: RLS walk (k3s_walk_kernel, wave form)
template <typename T>
__global__ void __launch_bounds__(64) kw_rls_walk_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    WCtx<T> cx(a, G.start, G.first_chunk, lds);
    const int K = cx.K, nacc = cx.NS + 1;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    double *S = cx.S0;
    for (int q = cx.lane; q < cx.NS; q += 64) S[q] = a.totals[(size_t)c * nacc + q];
    cx.sync();
    const bool seen = rel0 > 0 && cx.cnt(rel0 - 1) > 0;
    double last;
    if (seen) last = cx.solve(S, 0.0);
    else last = (cx.lane < K && a.mean0) ? a.mean0[cx.lane] : 0.0;                  // :519-522
    for (int64_t i = rel0; i < rel1; ++i) {
        if (cx.valid(i)) {
            cx.scale(S, a.ff);
            cx.add_row(S, i, 1.0);
            last = cx.solve(S, 0.0);
        }
        cx.store(i, last, coef, pred);
    }
}

template <typename T>
static int kw_rolling_launch_t(pols_ctx *ctx, const K4Args &a) {
    const size_t lds = sizeof(double) * WCtx<T>::lds_doubles(a.k);
    const int nacc = a.k * a.k + a.k;
    timing_begin(ctx);   // the whole three-launch pass
    hipLaunchKernelGGL((kw_totals_kernel<T, false>), dim3((unsigned)a.n_chunks), dim3(64), lds, ctx->stream, a);
    chunk_scan_launch(ctx, a, nacc, 0);
    hipLaunchKernelGGL((kw_rolling_walk_kernel<T>), dim3((unsigned)a.n_chunks), dim3(64), lds, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
static int kw_rls_launch_t(pols_ctx *ctx, const K4Args &a) {
    const size_t lds = sizeof(double) * WCtx<T>::lds_doubles(a.k);
    const int ns = a.k * a.k + a.k;
    timing_begin(ctx);   // the whole three-launch pass
    hipLaunchKernelGGL((kw_totals_kernel<T, true>), dim3((unsigned)a.n_chunks), dim3(64), lds, ctx->stream, a);
    chunk_scan_launch(ctx, a, ns, 2);
    hipLaunchKernelGGL((kw_rls_walk_kernel<T>), dim3((unsigned)a.n_chunks), dim3(64), lds, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int k4w_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    if (a.k > KW_MAX) return fail(POLS_ERR_UNSUPPORTED, "rolling: %d features > %d", a.k, KW_MAX);
    ctx->last_kernel = dtype == POLS_F32 ? "k4w_rolling_walk_f32" : "k4w_rolling_walk_f64";
    return dtype == POLS_F32 ? kw_rolling_launch_t<float>(ctx, a) : kw_rolling_launch_t<double>(ctx, a);
}

int k3sw_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    if (a.k > KW_MAX) return fail(POLS_ERR_UNSUPPORTED, "rls: %d features > %d", a.k, KW_MAX);
    ctx->last_kernel = dtype == POLS_F32 ? "k3sw_rls_scan_walk_f32" : "k3sw_rls_scan_walk_f64";
    return dtype == POLS_F32 ? kw_rls_launch_t<float>(ctx, a) : kw_rls_launch_t<double>(ctx, a);
}

}  // namespace pols
