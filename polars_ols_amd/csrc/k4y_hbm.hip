// k4y_hbm.hip -- rolling OLS and RLS for 129 .. 1 024 features: k4x_inverse.hip's algorithm (ONE WORKGROUP PER CHUNK propagating the
// INVERSE) with the K x K state of a chunk in an HBM / L2 area owned by its workgroup instead of LDS, and 1 024 threads.  The
// 33 .. 128-feature kernels stay as they were tuned; this file trades their LDS idioms for stride-NT loops so that K may exceed NT.
//
// For more than 60 features the reference itself stops re-factoring X'X per row and propagates (X'X)^-1 with Woodbury
// updates (WoodburyState, src/least_squares.rs:737-787; `use_woodbury` defaults to k > 60, :863); RLS always propagates
// P = A^-1 (RecursiveLeastSquares::update, :531-540).  Same three passes as k4_rolling.hip -- per-chunk totals of the outer
// products, the wave-parallel (decayed) scan, the walk -- but the walk keeps the K x K inverse in LDS (131 KB at K = 128):
//   chunk start   the state (X'X + alpha I, or the decayed information matrix with the prior) is rebuilt from the scanned
//                 totals (+ partial chunks, exactly like state_at() in k4_rolling.hip) and inverted in place by K sweeps of
//                 the symmetric sweep operator (pivots = the squared Cholesky pivots; K^3 multiply-adds once per chunk, so
//                 rounding does not accumulate beyond a chunk);
//   per row       rolling: P -= v v' / (1 + x'v) for the row that enters, P += v v' / (1 - x'v) for the row that leaves
//                 (v = P x), beta = P b;   RLS: the reference's update literally (r, gain, beta += gain * error,
//                 P = P / ff - gain gain' r);   each is O(K^2) over 256 threads and a handful of barriers.
// Control flow (warm-up, gate, forward fill, "drop" family) is k4_walk_kernel's, statement for statement.
#include "k4_rolling.hpp"

namespace pols {

constexpr int KY_MAX = 1024;

// NT threads; GLOBAL: the K x K state lives at a.state + chunk * K * LD instead of LDS (one workgroup owns it, so the barriers that
// order LDS accesses order these too).  Every "thread t owns element t" statement is a stride-NT loop, so K may exceed NT.
template <typename T, int NT, bool GLOBAL>
struct YCtx {
    const K4Args &a;
    int64_t s;
    int first_chunk;
    int K, LD, NS, tid;
    double *P, *xs, *v, *bv, *beta, *red;
    const T *mycol;              // column tid (the target for tid == K): the pointer of the first load_row() stride, kept in a register
    __device__ YCtx(const K4Args &a_, int64_t s_, int fc, double *lds) : a(a_), s(s_), first_chunk(fc) {
        K = a.k; LD = K | 1; NS = K * K + K; tid = threadIdx.x;
        mycol = tid < K ? column(tid) : static_cast<const T *>(a.y);
        P = GLOBAL ? a.state + (size_t)blockIdx.x * K * LD : lds;
        xs = GLOBAL ? lds : P + (size_t)K * LD;
        v = xs + K + 2; bv = v + K; beta = bv + K; red = beta + K;
    }
    static __host__ __device__ size_t lds_doubles(int K) { return (GLOBAL ? 0 : (size_t)K * (K | 1)) + 4 * (size_t)K + 2 + 16; }
    __device__ __forceinline__ const T *column(int j) const { return static_cast<const T *>(a.xtab ? a.xtab[j] : a.x[j]); }
    // (i, c) of element q = tid, tid + NT, ... of a K x K walk, carried: no divide per element (up to a thousand per thread and pivot)
    struct Walk {
        int i, c, di, dc, K;
        __device__ Walk(int tid, int K_) : K(K_) { i = tid / K; c = tid - i * K; di = NT / K; dc = NT - di * K; }
        __device__ __forceinline__ bool more() const { return i < K; }
        __device__ __forceinline__ void next() { i += di; c += dc; if (c >= K) { c -= K; ++i; } }
    };

    __device__ __forceinline__ bool valid(int64_t i) const { return a.valid ? a.valid[s + i] != 0 : true; }
    __device__ __forceinline__ int64_t cnt(int64_t i) const { return a.cnt ? (int64_t)a.cnt[s + i] : i + 1; }
    __device__ __forceinline__ int64_t vidx(int64_t r) const { return a.vidx ? (int64_t)a.vidx[s + r] : r; }

    __device__ __forceinline__ void load_row(int64_t i) const {      // xs[0..K) = x, xs[K] = y
        __syncthreads();
        if (tid <= K) xs[tid] = (double)mycol[s + i];
        for (int j = tid + NT; j <= K; j += NT) xs[j] = (double)(j < K ? column(j) : static_cast<const T *>(a.y))[s + i];
        __syncthreads();
    }
    __device__ double block_sum(double x) const {                     // all NT threads
        const double w = wave_sum_row3(x);
        __syncthreads();
        if ((tid & 63) == 63) red[tid >> 6] = w;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NT / 64; ++q) t += red[q];
        return t;
    }
    // sum_j p[j] * q[j] over the K entries of two LDS vectors
    __device__ double dot(const double *p, const double *q) const {
        double acc = 0.0;
        for (int j = tid; j < K; j += NT) acc += p[j] * q[j];
        return block_sum(acc);
    }
    // ---- raw Gram form (before the inversion): P holds X'X, bv holds X'y
    __device__ void zero() const {
        for (int q = tid; q < K * LD; q += NT) P[q] = 0.0;
        for (int j = tid; j < K; j += NT) bv[j] = 0.0;
        __syncthreads();
    }
    __device__ void gram_axpy(const double *src, double sign) const { // src: K*K (row-major, stride K) then K
        int q = tid;
        for (Walk w(tid, K); w.more(); w.next(), q += NT) P[w.i * LD + w.c] += sign * src[q];
        for (int j = tid; j < K; j += NT) bv[j] += sign * src[K * K + j];
        __syncthreads();
    }
    __device__ void gram_add_row(int64_t i, double sign) const {
        load_row(i);
        for (Walk w(tid, K); w.more(); w.next()) P[w.i * LD + w.c] += sign * (xs[w.i] * xs[w.c]);
        for (int j = tid; j < K; j += NT) bv[j] += sign * (xs[j] * xs[K]);
        __syncthreads();
    }
    __device__ void gram_prefix(int64_t i, double sign, int nacc) const {
        if (i < 0) return;
        const int64_t c = i / a.chunk_len;
        gram_axpy(a.totals + (size_t)(first_chunk + c) * nacc, sign);
        for (int64_t j = c * a.chunk_len; j <= i; ++j)
            if (valid(j)) gram_add_row(j, sign);
    }
    // ---- in-place inverse of P + alpha I by the symmetric sweep operator (result: the inverse, sign fixed at the end)
    __device__ void invert(double alpha) const {
        for (int j = tid; j < K; j += NT) P[j * LD + j] += alpha;
        __syncthreads();
        for (int j = 0; j < K; ++j) {
            const double p = 1.0 / P[j * LD + j];
            for (int t = tid; t < K; t += NT) v[t] = P[t * LD + j];    // column j (= row j)
            __syncthreads();
            for (Walk w(tid, K); w.more(); w.next()) {
                const int i = w.i, c = w.c;
                double val;
                if (i == j && c == j) val = -p;
                else if (i == j) val = v[c] * p;
                else if (c == j) val = v[i] * p;
                else val = P[i * LD + c] - v[i] * v[c] * p;
                P[i * LD + c] = val;
            }
            __syncthreads();
        }
        for (Walk w(tid, K); w.more(); w.next()) P[w.i * LD + w.c] = -P[w.i * LD + w.c];
        __syncthreads();
    }
    // out = P x, thread t owns entry t.  LDS: along ROW t (LD is odd: conflict-free, and consecutive words pair up into wide LDS
    // reads).  HBM / L2: P is symmetric, so down COLUMN t -- consecutive threads, consecutive words.
    __device__ void matvec(const double *x, double *out) const {
        for (int t = tid; t < K; t += NT) {
            double acc = 0.0;
            if (GLOBAL) {
                const double *col = P + t;
                for (int c = 0; c < K; ++c) acc += col[(size_t)c * LD] * x[c];
            } else {
                const double *row = P + t * LD;
                for (int c = 0; c < K; ++c) acc += row[c] * x[c];
            }
            out[t] = acc;
        }
        __syncthreads();
    }
    __device__ void rank1(double coef) const {                        // P += coef * v v'
        for (Walk w(tid, K); w.more(); w.next()) P[w.i * LD + w.c] += coef * (v[w.i] * v[w.c]);
        __syncthreads();
    }
    // Sherman-Morrison: the row in xs enters (sign = +1) or leaves (sign = -1) the window
    __device__ void sm_update(double sign) const {
        matvec(xs, v);
        const double xv = dot(xs, v);
        rank1(-sign / (1.0 + sign * xv));
        for (int j = tid; j < K; j += NT) bv[j] += sign * (xs[j] * xs[K]);
        __syncthreads();
    }
    __device__ void solve_beta() const { matvec(bv, beta); }           // beta = P b
    __device__ void store(int64_t i, bool have, T *coef, T *pred) const {
        const int64_t row = s + i;
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        if (coef)
            for (int j = tid; j < K; j += NT) coef[row * K + j] = (T)(have ? beta[j] : qnan);
        if (pred) {
            load_row(i);
            const double p = dot(xs, beta);
            if (tid == 0) pred[row] = (T)(have ? p : qnan);
        }
    }
};

// ------------------------------------------------------------------ pass 1: per-chunk totals (decayed for RLS)
template <typename T, bool RLS, int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT) ky_totals_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    YCtx<T, NT, GLOBAL> cx(a, G.start, G.first_chunk, lds);
    const int K = cx.K, nacc = cx.NS + (RLS ? 1 : 0);
    cx.zero();
    double decay = 1.0;
    for (int64_t i = ch.t0 - G.start; i < ch.t1 - G.start; ++i)
        if (cx.valid(i)) {
            if (RLS) {
                for (int q = cx.tid; q < K * cx.LD; q += NT) cx.P[q] *= a.ff;
                for (int j = cx.tid; j < K; j += NT) cx.bv[j] *= a.ff;
                decay *= a.ff;
                __syncthreads();
            }
            cx.gram_add_row(i, 1.0);
        }
    double *out = a.totals + (size_t)c * nacc;
    {
        int q = cx.tid;
        for (typename YCtx<T, NT, GLOBAL>::Walk w(cx.tid, K); w.more(); w.next(), q += NT) out[q] = cx.P[w.i * cx.LD + w.c];
    }
    for (int j = cx.tid; j < K; j += NT) out[K * K + j] = cx.bv[j];
    if (RLS && cx.tid == 0) out[cx.NS] = decay;
}

// ------------------------------------------------------------------ pass 3: rolling walk
template <typename T, int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT) ky_rolling_walk_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    YCtx<T, NT, GLOBAL> cx(a, G.start, G.first_chunk, lds);
    const int K = cx.K, nacc = cx.NS;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    const int64_t w = a.window, mpv = G.mpv;
    const bool drop = a.drop_mode != 0;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    if (G.all_nan) {                                           // :893-900
        for (int64_t i = rel0; i < rel1; ++i) {
            if (coef)
                for (int j = cx.tid; j < K; j += NT) coef[(G.start + i) * K + j] = (T)qnan;
            if (pred && cx.tid == 0) pred[G.start + i] = (T)qnan;
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
    auto gram_state_at = [&](int64_t i) {                      // raw (X'X, X'y) after row i has been processed
        cx.zero();
        cx.gram_prefix(i, 1.0, nacc);
        const int64_t o = (i >= 0) ? old_of(i) : -1;
        if (o >= j_min && i >= mpv) {
            cx.gram_prefix(o, -1.0, nacc);
            cx.gram_prefix(j_min - 1, 1.0, nacc);
        }
        if (longwarm && i >= mpv && cx.cnt(i) - 1 - c0 >= w) {  // the warm-up rows the deque never held stay in the sums
            cx.gram_prefix(cx.vidx(c0 - 1), 1.0, nacc);
            cx.gram_prefix(cx.vidx(w - 1), -1.0, nacc);
        }
    };

    bool inverted = false, have = false;                       // `have`: some row at or before the current one produced coefficients
    int64_t prev_old = (rel0 > 0 && rel0 - 1 >= mpv) ? old_of(rel0 - 1) : -1;
    if (rel0 >= mpv && rel0 > 0) {
        // an earlier row already produced coefficients: carry them in.  If the rows just before this chunk were
        // forward-filled (gate closed), the carried coefficients belong to the last row that did solve.
        int64_t ip = rel0 - 1;
        if (!(drop || ip == mpv - 1 || gate(ip))) {
            for (ip = rel0 - 2; ip >= mpv - 1; --ip)
                if (ip == mpv - 1 || gate(ip)) break;
        }
        gram_state_at(ip);
        cx.invert(a.alpha);
        cx.solve_beta();
        have = true;
        if (ip != rel0 - 1) {                                  // rebuild the state of row rel0 - 1 for the walk
            gram_state_at(rel0 - 1);
            cx.invert(a.alpha);
        }
        inverted = true;
    } else {
        gram_state_at(rel0 - 1);
    }
    for (int64_t i = rel0; i < rel1; ++i) {
        const bool vld = cx.valid(i);
        if (vld) {
            if (inverted) { cx.load_row(i); cx.sm_update(1.0); }
            else cx.gram_add_row(i, 1.0);
        }
        if (i >= mpv && (vld || !drop)) {                      // subtract what left the window (always after the first solve)
            const int64_t no = old_of(i);
            bool sub = false;
            if (drop) sub = (no != prev_old && no >= 0);
            else sub = (no >= j_min && no >= 0 && cx.valid(no));
            if (sub) { cx.load_row(no); cx.sm_update(-1.0); }
            prev_old = no;
        }
        if (i >= mpv - 1) {
            const bool do_solve = (i == mpv - 1) || (drop ? vld : gate(i));
            if (do_solve) {
                if (!inverted) { cx.invert(a.alpha); inverted = true; }   // alpha enters once, at the warm-up (:924-926)
                cx.solve_beta();
                have = true;
            }
        }
        cx.store(i, have, coef, pred);
    }
}

// ------------------------------------------------------------------ pass 3: RLS walk (RecursiveLeastSquares::update, literally)
template <typename T, int NT, bool GLOBAL>
__global__ void __launch_bounds__(NT) ky_rls_walk_kernel(const K4Args a) {
    extern __shared__ double lds[];
    const int64_t c = blockIdx.x;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    YCtx<T, NT, GLOBAL> cx(a, G.start, G.first_chunk, lds);
    const int K = cx.K, nacc = cx.NS + 1;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    cx.zero();
    cx.gram_axpy(a.totals + (size_t)c * nacc, 1.0);            // A = P^-1 and b = A beta at the chunk start (prior included)
    cx.invert(0.0);
    cx.solve_beta();                                           // beta = P b (= initial_state_mean before any valid row)
    const double ff = a.ff;
    for (int64_t i = rel0; i < rel1; ++i) {
        if (cx.valid(i)) {
            cx.load_row(i);
            cx.matvec(cx.xs, cx.v);                            // v = P x
            const double xv = cx.dot(cx.xs, cx.v);
            const double xb = cx.dot(cx.xs, cx.beta);
            const double r = 1.0 + xv / ff;                    // :533
            const double err = cx.xs[K] - xb;
            // gain = P x / (r ff);  beta += gain * err;  P = P / ff - gain gain' r = (P - v v' / (r ff)) / ff
            for (int j = cx.tid; j < K; j += NT) cx.beta[j] += cx.v[j] / (r * ff) * err;
            const double coef_vv = -1.0 / (r * ff);
            for (typename YCtx<T, NT, GLOBAL>::Walk w(cx.tid, K); w.more(); w.next())
                cx.P[w.i * cx.LD + w.c] = (cx.P[w.i * cx.LD + w.c] + coef_vv * (cx.v[w.i] * cx.v[w.c])) / ff;
            __syncthreads();
        }
        cx.store(i, true, coef, pred);
    }
}

template <typename T, int NT, bool GLOBAL>
static int ky_launch_t(pols_ctx *ctx, const K4Args &a_in, bool rls) {
    K4Args a = a_in;
    const size_t lds = sizeof(double) * YCtx<T, NT, GLOBAL>::lds_doubles(a.k);
    const int ns = a.k * a.k + a.k;
    if (GLOBAL) {                                              // one K x K state per chunk (scratch slot 3)
        void *st = nullptr;
        int rc = ensure_scratch(ctx, 3, sizeof(double) * (size_t)a.n_chunks * a.k * (a.k | 1), &st);
        if (rc) return rc;
        a.state = static_cast<double *>(st);
    } else {
        static OncePerDevice attr_once;
        if (attr_once.needed(ctx->device)) {
            const void *fns[4] = {reinterpret_cast<const void *>(&ky_totals_kernel<T, false, NT, GLOBAL>), reinterpret_cast<const void *>(&ky_totals_kernel<T, true, NT, GLOBAL>),
                                  reinterpret_cast<const void *>(&ky_rolling_walk_kernel<T, NT, GLOBAL>), reinterpret_cast<const void *>(&ky_rls_walk_kernel<T, NT, GLOBAL>)};
            for (const void *f : fns) POLS_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            attr_once.done(ctx->device);
        }
    }
    timing_begin(ctx);
    if (rls) {
        hipLaunchKernelGGL((ky_totals_kernel<T, true, NT, GLOBAL>), dim3((unsigned)a.n_chunks), dim3(NT), lds, ctx->stream, a);
        chunk_scan_launch(ctx, a, ns, 2);
        hipLaunchKernelGGL((ky_rls_walk_kernel<T, NT, GLOBAL>), dim3((unsigned)a.n_chunks), dim3(NT), lds, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((ky_totals_kernel<T, false, NT, GLOBAL>), dim3((unsigned)a.n_chunks), dim3(NT), lds, ctx->stream, a);
        chunk_scan_launch(ctx, a, ns, 0);
        hipLaunchKernelGGL((ky_rolling_walk_kernel<T, NT, GLOBAL>), dim3((unsigned)a.n_chunks), dim3(NT), lds, ctx->stream, a);
    }
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int k4y_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    if (a.k > KY_MAX) return fail(POLS_ERR_UNSUPPORTED, "rolling: %d features > %d", a.k, KY_MAX);
    ctx->last_kernel = dtype == POLS_F32 ? "k4y_rolling_inverse_hbm_f32" : "k4y_rolling_inverse_hbm_f64";
    return dtype == POLS_F32 ? ky_launch_t<float, 1024, true>(ctx, a, false) : ky_launch_t<double, 1024, true>(ctx, a, false);
}

int k3y_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    if (a.k > KY_MAX) return fail(POLS_ERR_UNSUPPORTED, "rls: %d features > %d", a.k, KY_MAX);
    ctx->last_kernel = dtype == POLS_F32 ? "k3y_rls_inverse_hbm_f32" : "k3y_rls_inverse_hbm_f64";
    return dtype == POLS_F32 ? ky_launch_t<float, 1024, true>(ctx, a, true) : ky_launch_t<double, 1024, true>(ctx, a, true);
}

}  // namespace pols
