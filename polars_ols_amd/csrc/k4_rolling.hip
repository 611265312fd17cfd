// k4_rolling.hip -- K4 "rolling_gram_solve": rolling-window OLS, chunk-parallel.
//
// Replaces solve_rolling_ols (src/least_squares.rs:848-1032: warm-up :907-943, "drop" family :947-986, fixed
// window "drop_window" :987-1029) + the dynamic make_predictions (src/expressions.rs:184, 695-700).
//
// The reference walks a sequence row by row keeping (X'X, X'y) of the current window -- add the new row, subtract
// the row that leaves (NonWoodburyState::update :707-725) -- and solves a K x K system per row (Cholesky -> LU,
// :732-734).  That state is a pure function of the data: S_i = P(i) - P(old(i)) with P the prefix sum of the valid
// rows' outer products and old(i) = i - window (fixed window) or the row of the valid observation `window` ranks
// back ("drop").  So the sequence is cut into chunks, ONE LANE PER CHUNK:
//   pass 1  chunk_totals : per-chunk sums of the outer products                          (1 read of the rows)
//   pass 2  chunk_scan   : exclusive prefix over each group's chunks                      (tiny)
//   pass 3  chunk_walk   : the lane rebuilds S at its chunk start from the prefixes (<= 2 chunk lengths of row
//                          work), then slides add/subtract exactly like the reference and runs the unrolled
//                          Cholesky (LU on failure) per row in registers; coefficients of the rows that the
//                          reference forward-fills (invalid rows, windows with too few observations) are carried.
// A single 1M-row sequence thereby becomes ~4000 independent lanes instead of one dependency chain.  The Woodbury
// variant (:737-787) propagates (X'X)^-1 instead of X'X: same mathematics, different rounding; it is not
// reproduced -- `use_woodbury` is accepted and ignored.  All arithmetic is f64.
#include "k4_rolling.hpp"
#include "k4_small.inl"   // K4N, solve_state (tri_index, chol_solve from k1_kernel.inl)

namespace pols {

template <typename T, int K>
struct K4Ctx {
    const K4Args &a;
    int64_t s;       // group start (absolute row)
    int first_chunk;
    __device__ __forceinline__ bool valid(int64_t i) const { return a.valid ? a.valid[s + i] != 0 : true; }
    __device__ __forceinline__ int64_t cnt(int64_t i) const { return a.cnt ? (int64_t)a.cnt[s + i] : i + 1; }   // valid rows in [0, i]
    __device__ __forceinline__ int64_t vidx(int64_t r) const { return a.vidx ? (int64_t)a.vidx[s + r] : r; }
    __device__ __forceinline__ void load_row(int64_t i, double (&x)[K], double &y) const {
#pragma unroll
        for (int j = 0; j < K; ++j) x[j] = (double)static_cast<const T *>(a.x[j])[s + i];
        y = (double)static_cast<const T *>(a.y)[s + i];
    }
    // PF consecutive rows at once: every load is issued before the first use, so the (L2) latency is paid once per window, not per row
    template <int PF>
    __device__ __forceinline__ void load_window(int64_t i0, int64_t i_end, double (&x)[PF][K], double (&y)[PF], bool (&v)[PF]) const {
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int64_t i = (i0 + r < i_end) ? i0 + r : i_end - 1;    // clamp: the tail re-reads the last row (unused)
            load_row(i, x[r], y[r]);
            v[r] = (i0 + r < i_end) && valid(i);
        }
    }
    __device__ __forceinline__ void add_loaded(double (&S)[K4N<K>::N], const double (&x)[K], double y, double sign) const {
#pragma unroll
        for (int p = 0; p < K; ++p) {
#pragma unroll
            for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] += sign * (x[p] * x[q]);
            S[K4N<K>::NX + p] += sign * (x[p] * y);
        }
    }
    // S += sign * [x x' (packed upper), x y]   (outer_product :600-607, update :714-723)
    __device__ __forceinline__ void add_row(double (&S)[K4N<K>::N], int64_t i, double sign) const {
        double x[K], y;
        load_row(i, x, y);
#pragma unroll
        for (int p = 0; p < K; ++p) {
#pragma unroll
            for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] += sign * (x[p] * x[q]);
            S[K4N<K>::NX + p] += sign * (x[p] * y);
        }
    }
    // P(i): sum over the valid rows 0..i of the group (i < 0 -> 0), from the chunk prefix + a partial chunk
    __device__ __forceinline__ void prefix(int64_t i, double (&P)[K4N<K>::N], double sign) const {
        if (i < 0) return;
        const int64_t c = i / a.chunk_len;
        const double *pb = a.totals + (size_t)(first_chunk + c) * a.tot_cs;
#pragma unroll
        for (int q = 0; q < K4N<K>::N; ++q) P[q] += sign * pb[(size_t)q * a.tot_qs];
        for (int64_t j = c * a.chunk_len; j <= i; ++j)
            if (valid(j)) add_row(P, j, sign);
    }
};

// P = A^-1 (packed upper, like S) from the packed SPD matrix in S[0 .. NX): Cholesky A = L L', M = L^-1 by forward substitution,
// P = M' M.  false on a non-positive pivot.  Used once per chunk by the RLS walk, which then propagates P with the reference's
// own rank-1 update (ls.rs:531-540) instead of re-factoring the information matrix on every row.
template <int K>
__device__ __forceinline__ bool spd_inverse_packed(const double (&S)[K4N<K>::N], double (&P)[K * (K + 1) / 2]) {
    double L[K][K], M[K][K];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        double d = S[tri_index<K>(j, j)];
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-L[j][p], L[j][p], d);
        ok = ok && (d > 0.0);
        const double ri = 1.0 / sqrt(d);
        L[j][j] = ri;                                           // 1 / L_jj
#pragma unroll
        for (int i = j + 1; i < K; ++i) {
            double acc = S[tri_index<K>(j, i)];
#pragma unroll
            for (int p = 0; p < j; ++p) acc = fma(-L[i][p], L[j][p], acc);
            L[i][j] = acc * ri;
        }
    }
#pragma unroll
    for (int c = 0; c < K; ++c) {                               // column c of M = L^-1
        M[c][c] = L[c][c];
#pragma unroll
        for (int i = c + 1; i < K; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int p = c; p < i; ++p) acc = fma(L[i][p], M[p][c], acc);
            M[i][c] = -acc * L[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = i; j < K; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int p = j; p < K; ++p) acc = fma(M[p][i], M[p][j], acc);   // (M'M)_ij, M lower triangular
            P[tri_index<K>(i, j)] = acc;
        }
    return ok;
}

// ------------------------------------------------------------------ pass 1: per-chunk totals
template <typename T, int K>
__global__ void __launch_bounds__(64) k4_totals_kernel(const K4Args a) {
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n_chunks) return;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    K4Ctx<T, K> cx{a, G.start, G.first_chunk};
    double S[K4N<K>::N];
#pragma unroll
    for (int q = 0; q < K4N<K>::N; ++q) S[q] = 0.0;
    constexpr int PF = 4;
    const int64_t i_end = ch.t1 - G.start;
    for (int64_t i0 = ch.t0 - G.start; i0 < i_end; i0 += PF) {
        double xw[PF][K], yw[PF];
        bool vw[PF];
        cx.template load_window<PF>(i0, i_end, xw, yw, vw);
#pragma unroll
        for (int r = 0; r < PF; ++r)
            if (vw[r]) cx.add_loaded(S, xw[r], yw[r], 1.0);
    }
#pragma unroll
    for (int q = 0; q < K4N<K>::N; ++q) a.totals[(size_t)c * a.tot_cs + (size_t)q * a.tot_qs] = S[q];
}

// ------------------------------------------------------------------ pass 2: exclusive (decayed) prefix over a group's chunks
// One wave per (group, component).  The recurrence run' = d * run + t is a scan under the associative operator
// (d1, t1) . (d2, t2) = (d1 d2, d2 t1 + t2): 256 chunks per step (4 per lane), six shuffle rounds, the next tile's loads in flight
// while the current one is combined -- a 1M-row sequence (15 625 chunks) is ~60 steps instead of 15 625 dependent loads.
//   mode 0  plain prefix (rolling): d = 1, carry-in 0
//   mode 1  RLS, packed upper-triangular state of K features (K4N<K>): d = the chunk's decay (slot nacc), carry-in = prior
//   mode 2  RLS, full K x K state (k4w_wide.hip)
// SW waves per (group, component): a long sequence's chunk list is cut into SW segments; every wave first composes its own segment
// (no writes), the segment aggregates meet in LDS, and each wave then re-walks its segment from its carry-in writing the prefixes.
// qb components per workgroup (blockIdx.y covers [qb y, qb (y + 1))): frames of MANY sequences of which few are cut -- a long tail of sequence lengths --
// launched one workgroup per (sequence, component) whatever the sequence (19 000 x 157 workgroups at 12 features for 1 100 cut sequences); with qb = 16
// an uncut sequence costs a sixteenth of them, and each of those only stores the prior (what its one chunk enters with) and leaves.
template <int SW>
__global__ void __launch_bounds__(64 * SW) chunk_scan_kernel(const K4Args a, const int nacc, const int mode, const int qb) {
    __shared__ double segD[SW], segT[SW];
    const int64_t g = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const K4Group G = a.groups[g];
    const int64_t n = G.end - G.start;
    const int64_t nch = (n + a.chunk_len - 1) / a.chunk_len;
    auto prior = [&](int q) -> double {                      // A_0 = I / p0, b_0 = A_0 mean0 (mode 0: nothing)
        if (!mode) return 0.0;
        const int K = a.k;
        const int nx = (mode == 1) ? K * (K + 1) / 2 : K * K;
        if (q < nx) {
            bool diag;
            if (mode == 1) { int i = 0, rem = q; while (rem >= K - i) { rem -= K - i; ++i; } diag = rem == 0; }
            else diag = (q / K) == (q % K);
            return diag ? 1.0 / a.p0 : 0.0;
        }
        return a.mean0 ? a.mean0[q - nx] / a.p0 : 0.0;
    };
    const int q_lo = (int)blockIdx.y * qb, q_hi = q_lo + qb < nacc ? q_lo + qb : nacc;
    if (nch <= 1) {                                          // nothing to scan: the sequence's only chunk enters with the prior
        const int q = q_lo + (int)threadIdx.x;
        if (nch == 1 && q < q_hi) a.totals[(size_t)G.first_chunk * a.tot_cs + (size_t)q * a.tot_qs] = prior(q);
        return;
    }
    for (int q = q_lo; q < q_hi; ++q) {
    if (SW > 1 && q > q_lo) __syncthreads();                // (segD / segT of the previous component have been read)
    double carry = prior(q);
    double *base = a.totals + (size_t)G.first_chunk * a.tot_cs + (size_t)q * a.tot_qs;      // this component's chunk series
    const double *dbase = a.totals + (size_t)G.first_chunk * a.tot_cs + (size_t)nacc * a.tot_qs;   // the decay series (mode != 0)
    const int64_t cs = a.tot_cs;
    constexpr int CPL = 4;                                   // consecutive chunks per lane -> 256 chunks per step
    constexpr int TILE = 64 * CPL;
    const int64_t seg = SW == 1 ? nch : ((nch + SW - 1) / SW + TILE - 1) / TILE * TILE;   // segment length, a multiple of the tile
    const int64_t c_lo = min(nch, (int64_t)wv * seg), c_hi = min(nch, c_lo + seg);
    auto fetch = [&](int64_t c, double (&d)[CPL], double (&t)[CPL]) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const bool in = c + i < c_hi;
            t[i] = in ? base[(size_t)(c + i) * cs] : 0.0;
            d[i] = (in && mode) ? dbase[(size_t)(c + i) * cs] : 1.0;
        }
    };
    auto lane_compose = [&](const double (&d)[CPL], const double (&t)[CPL], double &D, double &T) {
        D = d[0]; T = t[0];
#pragma unroll
        for (int i = 1; i < CPL; ++i) { T = d[i] * T + t[i]; D = D * d[i]; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {             // inclusive scan over the lanes
            const double dp = __shfl_up(D, off), tp = __shfl_up(T, off);
            if (lane >= off) { T = D * tp + T; D = D * dp; }
        }
    };
    if constexpr (SW > 1) {
        // ---- phase 1: this wave's segment as ONE element (Dseg, Tseg)
        double Dseg = 1.0, Tseg = 0.0, dn[CPL], tn[CPL];
        fetch(c_lo + (int64_t)lane * CPL, dn, tn);
        for (int64_t c0 = c_lo; c0 < c_hi; c0 += TILE) {
            double d[CPL], t[CPL], D, T;
#pragma unroll
            for (int i = 0; i < CPL; ++i) { d[i] = dn[i]; t[i] = tn[i]; }
            fetch(c0 + TILE + (int64_t)lane * CPL, dn, tn);
            lane_compose(d, t, D, T);
            const double Dt = __shfl(D, 63), Tt = __shfl(T, 63);
            Tseg = Dt * Tseg + Tt; Dseg = Dseg * Dt;
        }
        if (lane == 0) { segD[wv] = Dseg; segT[wv] = Tseg; }
        __syncthreads();
        for (int sgm = 0; sgm < wv; ++sgm) carry = segD[sgm] * carry + segT[sgm];
    }
    // ---- phase 2: exclusive prefixes of the segment from its carry-in
    double dn[CPL], tn[CPL];
    fetch(c_lo + (int64_t)lane * CPL, dn, tn);
    for (int64_t c0 = c_lo; c0 < c_hi; c0 += TILE) {
        double d[CPL], t[CPL], D, T;
#pragma unroll
        for (int i = 0; i < CPL; ++i) { d[i] = dn[i]; t[i] = tn[i]; }
        fetch(c0 + TILE + (int64_t)lane * CPL, dn, tn);      // prefetch the next tile
        lane_compose(d, t, D, T);
        const double dex = __shfl_up(D, 1), tex = __shfl_up(T, 1);
        double run = (lane == 0) ? carry : dex * carry + tex;   // state entering this lane's first chunk
        const int64_t c = c0 + (int64_t)lane * CPL;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            if (c + i < c_hi) base[(size_t)(c + i) * cs] = run;
            run = d[i] * run + t[i];
        }
        carry = __shfl(D, 63) * carry + __shfl(T, 63);
    }
    }
}

void chunk_scan_launch(pols_ctx *ctx, const K4Args &a, int nacc, int mode) {
    // few long sequences: 8 waves per (group, component); many short ones: one wave each, 16 components per workgroup from 1 024 sequences on
    const bool lng = a.n_chunks / std::max<int64_t>(1, a.n_groups) > 2048;
    const int qb = (!lng && a.n_groups >= 1024) ? 16 : 1;
    const unsigned gy = (unsigned)((nacc + qb - 1) / qb);
    if (lng) hipLaunchKernelGGL(chunk_scan_kernel<8>, dim3((unsigned)a.n_groups, gy), dim3(512), 0, ctx->stream, a, nacc, mode, qb);
    else hipLaunchKernelGGL(chunk_scan_kernel<1>, dim3((unsigned)a.n_groups, gy), dim3(64), 0, ctx->stream, a, nacc, mode, qb);
}

// ------------------------------------------------------------------ pass 3: the walk
template <typename T, int K>
__global__ void __launch_bounds__(64) k4_walk_kernel(const K4Args a) {
    constexpr int N = K4N<K>::N;
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n_chunks) return;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    K4Ctx<T, K> cx{a, G.start, G.first_chunk};
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    const int64_t w = a.window, mpv = G.mpv;
    const bool drop = a.drop_mode != 0;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);

    if (G.all_nan) {                                       // :893-900
        for (int64_t i = rel0; i < rel1; ++i) {
            if (coef) for (int j = 0; j < K; ++j) coef[(G.start + i) * K + j] = (T)qnan;
            if (pred) pred[G.start + i] = (T)qnan;
        }
        return;
    }
    // rows older than j_min are never subtracted: the warm-up adds every valid row of [0, mpv) (:909-921) and the
    // sliding loop only starts subtracting at i = mpv (:989-990)
    const int64_t j_min = drop ? 0 : max(mpv - w, (int64_t)0);
    // min_periods > window under the drop family (ls.rs:869-876 only warns): the warm-up sums c0 > w valid rows but its deque
    // keeps the FIRST w of them (:917-919), so the first w slides pop ranks 0 .. w-1 (rank R - c0 leaves as rank R enters) and
    // ranks [w, c0) are never subtracted at all; afterwards the deque is an ordinary w-row window again.
    const int64_t c0 = drop ? cx.cnt(mpv - 1) : 0;
    const bool longwarm = drop && c0 > w;
    auto old_of = [&](int64_t i) -> int64_t {              // the row that has left the window once row i is in
        if (!drop) return i - w;
        const int64_t R = cx.cnt(i) - 1;
        const int64_t r = (longwarm && R - c0 < w) ? R - c0 : R - w;               // rank of the valid row `window` observations back
        return r < 0 ? -1 : cx.vidx(r);
    };
    auto gate = [&](int64_t i) -> bool {                   // n_valid_window >= n_valid (:994-997, 1013, 1022)
        const int64_t i_start = i >= w ? i - w : 0;        // saturating_sub (:990)
        return cx.cnt(i) - cx.cnt(i_start) >= G.gate_n;
    };
    auto state_at = [&](int64_t i, double (&S)[N]) {       // (X'X, X'y) after row i has been processed
#pragma unroll
        for (int q = 0; q < N; ++q) S[q] = 0.0;
        cx.prefix(i, S, 1.0);
        const int64_t o = (i >= 0) ? old_of(i) : -1;
        if (o >= j_min && i >= mpv) {                      // nothing is subtracted before the sliding loop starts
            cx.prefix(o, S, -1.0);
            cx.prefix(j_min - 1, S, 1.0);
        }
        if (longwarm && i >= mpv && cx.cnt(i) - 1 - c0 >= w) {  // the warm-up rows the deque never held stay in the sums
            cx.prefix(cx.vidx(c0 - 1), S, 1.0);
            cx.prefix(cx.vidx(w - 1), S, -1.0);
        }
    };

    double S[N], last[K];
    state_at(rel0 - 1, S);
    int64_t prev_old = (rel0 > 0 && rel0 - 1 >= mpv) ? old_of(rel0 - 1) : -1;
    bool have_last = false;
#pragma unroll
    for (int j = 0; j < K; ++j) last[j] = qnan;
    if (rel0 >= mpv && rel0 > 0) {                         // some earlier row already produced coefficients: carry them in
        if (drop || rel0 - 1 == mpv - 1 || gate(rel0 - 1)) {
            solve_state<K>(S, a.alpha, last);              // the state has not changed since the last solved row
        } else {                                           // rare: the rows before this chunk were forward-filled
            for (int64_t ip = rel0 - 2; ip >= mpv - 1; --ip) {
                if (ip == mpv - 1 || gate(ip)) {
                    double Sp[N];
                    state_at(ip, Sp);
                    solve_state<K>(Sp, a.alpha, last);
                    break;
                }
            }
        }
        have_last = true;
    }
    (void)have_last;

    constexpr int PF = 2;                                  // entering rows are loaded a window ahead of the solves
    for (int64_t i0 = rel0; i0 < rel1; i0 += PF) {
        double xw[PF][K], yw[PF];
        bool vw[PF];
        cx.template load_window<PF>(i0, rel1, xw, yw, vw);
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int64_t i = i0 + r;
            if (i >= rel1) break;
            const bool v = vw[r];
            if (v) cx.add_loaded(S, xw[r], yw[r], 1.0);        // update XTX w/ latest data point
            if (i >= mpv && (v || !drop)) {                    // subtract what left the window
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
                if (do_solve) solve_state<K>(S, a.alpha, last);
            }
            const int64_t row = G.start + i;
            if (coef) {
#pragma unroll
                for (int j = 0; j < K; ++j) coef[row * K + j] = (T)last[j];
            }
            if (pred) {                                        // (features * coefficients).sum_axis(1)  (ex.rs:184)
                double p = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) p += xw[r][j] * last[j];
                pred[row] = (T)p;
            }
        }
    }
}

template <typename T, int K>
static int k4_launch_k(pols_ctx *ctx, const K4Args &a) {
    const unsigned blocks = (unsigned)((a.n_chunks + 63) / 64);
    timing_begin(ctx);   // the whole three-launch pass
    hipLaunchKernelGGL((k4_totals_kernel<T, K>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    chunk_scan_launch(ctx, a, K4N<K>::N, 0);
    hipLaunchKernelGGL((k4_walk_kernel<T, K>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
static int k4_launch_t(pols_ctx *ctx, const K4Args &a) {
    switch (a.k) {
        case 1: return k4_launch_k<T, 1>(ctx, a);
        case 2: return k4_launch_k<T, 2>(ctx, a);
        case 3: return k4_launch_k<T, 3>(ctx, a);
        case 4: return k4_launch_k<T, 4>(ctx, a);
        case 5: return k4_launch_k<T, 5>(ctx, a);
        case 6: return k4_launch_k<T, 6>(ctx, a);
        case 7: return k4_launch_k<T, 7>(ctx, a);
        case 8: return k4_launch_k<T, 8>(ctx, a);
        default: return fail(POLS_ERR_UNSUPPORTED, "rolling: %d features > %d", a.k, K4_KMAX);
    }
}

// ====================================================================================================================
// K3s: recursive least squares as a decayed scan (information form), same chunk machinery.
//   RecursiveLeastSquares::update (ls.rs:531-540):  P' = P/ff - k k' r,  k = P x / (r ff),  r = 1 + x'P x / ff
//   Sherman-Morrison:  P' = (ff P^-1 + x x')^-1,  and  coef' = P' (ff P^-1 coef + x y)
//   => with A = P^-1, b = A coef:  A' = ff A + x x',  b' = ff b + x y,  coef' = A'^-1 b'   (invalid rows: no change).
template <typename T, int K>
__global__ void __launch_bounds__(64) k3s_totals_kernel(const K4Args a) {
    constexpr int N = K4N<K>::N;
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n_chunks) return;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    K4Ctx<T, K> cx{a, G.start, G.first_chunk};
    double S[N], decay = 1.0;
#pragma unroll
    for (int q = 0; q < N; ++q) S[q] = 0.0;
    constexpr int PF = 4;
    const int64_t i_end = ch.t1 - G.start;
    for (int64_t i0 = ch.t0 - G.start; i0 < i_end; i0 += PF) {
        double xw[PF][K], yw[PF];
        bool vw[PF];
        cx.template load_window<PF>(i0, i_end, xw, yw, vw);
#pragma unroll
        for (int r = 0; r < PF; ++r)
            if (vw[r]) {
#pragma unroll
                for (int q = 0; q < N; ++q) S[q] *= a.ff;
                cx.add_loaded(S, xw[r], yw[r], 1.0);
                decay *= a.ff;
            }
    }
#pragma unroll
    for (int q = 0; q < N; ++q) a.totals[(size_t)c * a.tot_cs + (size_t)q * a.tot_qs] = S[q];
    a.totals[(size_t)c * a.tot_cs + (size_t)N * a.tot_qs] = decay;
}

template <typename T, int K>
__global__ void __launch_bounds__(64) k3s_walk_kernel(const K4Args a) {
    constexpr int N = K4N<K>::N;
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= a.n_chunks) return;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    K4Ctx<T, K> cx{a, G.start, G.first_chunk};
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    double S[N], last[K];
#pragma unroll
    for (int q = 0; q < N; ++q) S[q] = a.totals[(size_t)c * a.tot_cs + (size_t)q * a.tot_qs];
    const bool seen = rel0 > 0 && cx.cnt(rel0 - 1) > 0;           // has any valid row updated the state yet?
    if (seen) solve_state<K>(S, 0.0, last);
    else {
#pragma unroll
        for (int j = 0; j < K; ++j) last[j] = a.mean0 ? a.mean0[j] : 0.0;   // coef = initial_state_mean or zeros (:519-522)
    }
    constexpr int PF = 2;                                         // rows loaded ahead of the updates
    // Inside the chunk the covariance form is propagated, exactly the reference's update: P = A^-1 once per chunk, then
    //   v = P x,  r = 1 + x'v / ff,  gain = v / (r ff),  beta += gain (y - x'beta),  P = (P - v gain') / ff     (ls.rs:531-540)
    // -- about 110 f64 operations per row at K = 6 against ~250 for re-factoring A.  (A holds the prior, so it is SPD; should
    // the factorisation fail anyway, the per-row solve below is the fallback.)
    constexpr int NXP = K * (K + 1) / 2;
    double P[NXP];
    if (spd_inverse_packed<K>(S, P)) {
        if (!seen) {
#pragma unroll
            for (int j = 0; j < K; ++j) last[j] = a.mean0 ? a.mean0[j] : 0.0;
        }
        const double ff = a.ff, iff = 1.0 / a.ff;
        for (int64_t i0 = rel0; i0 < rel1; i0 += PF) {
            double xw[PF][K], yw[PF];
            bool vw[PF];
            cx.template load_window<PF>(i0, rel1, xw, yw, vw);
#pragma unroll
            for (int r = 0; r < PF; ++r) {
                const int64_t i = i0 + r;
                if (i >= rel1) break;
                if (vw[r]) {
                    double v[K], xv = 0.0, xb = 0.0;
#pragma unroll
                    for (int p = 0; p < K; ++p) {
                        double acc = 0.0;
#pragma unroll
                        for (int q = 0; q < K; ++q) acc = fma(P[p <= q ? tri_index<K>(p, q) : tri_index<K>(q, p)], xw[r][q], acc);
                        v[p] = acc;
                        xv = fma(xw[r][p], acc, xv);
                        xb = fma(xw[r][p], last[p], xb);
                    }
                    const double rr = 1.0 + xv * iff;
                    const double gs = 1.0 / (rr * ff);                  // gain = v * gs
                    const double err = yw[r] - xb;
#pragma unroll
                    for (int p = 0; p < K; ++p) last[p] = fma(v[p] * gs, err, last[p]);
#pragma unroll
                    for (int p = 0; p < K; ++p)
#pragma unroll
                        for (int q = p; q < K; ++q) P[tri_index<K>(p, q)] = (P[tri_index<K>(p, q)] - v[p] * (v[q] * gs)) * iff;
                }
                const int64_t row = G.start + i;
                if (coef) {
#pragma unroll
                    for (int j = 0; j < K; ++j) coef[row * K + j] = (T)last[j];
                }
                if (pred) {
                    double p = 0.0;
#pragma unroll
                    for (int j = 0; j < K; ++j) p += xw[r][j] * last[j];
                    pred[row] = (T)p;
                }
            }
        }
        return;
    }
    for (int64_t i0 = rel0; i0 < rel1; i0 += PF) {
        double xw[PF][K], yw[PF];
        bool vw[PF];
        cx.template load_window<PF>(i0, rel1, xw, yw, vw);
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int64_t i = i0 + r;
            if (i >= rel1) break;
            if (vw[r]) {
#pragma unroll
                for (int q = 0; q < N; ++q) S[q] *= a.ff;
                cx.add_loaded(S, xw[r], yw[r], 1.0);
                solve_state<K>(S, 0.0, last);
            }
            const int64_t row = G.start + i;
            if (coef) {
#pragma unroll
                for (int j = 0; j < K; ++j) coef[row * K + j] = (T)last[j];
            }
            if (pred) {
                double p = 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) p += xw[r][j] * last[j];
                pred[row] = (T)p;
            }
        }
    }
}

template <typename T, int K>
static int k3s_launch_k(pols_ctx *ctx, const K4Args &a) {
    const unsigned blocks = (unsigned)((a.n_chunks + 63) / 64);
    timing_begin(ctx);   // the whole three-launch pass
    hipLaunchKernelGGL((k3s_totals_kernel<T, K>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    chunk_scan_launch(ctx, a, K4N<K>::N, 1);
    hipLaunchKernelGGL((k3s_walk_kernel<T, K>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
static int k3s_launch_t(pols_ctx *ctx, const K4Args &a) {
    switch (a.k) {
        case 1: return k3s_launch_k<T, 1>(ctx, a);
        case 2: return k3s_launch_k<T, 2>(ctx, a);
        case 3: return k3s_launch_k<T, 3>(ctx, a);
        case 4: return k3s_launch_k<T, 4>(ctx, a);
        case 5: return k3s_launch_k<T, 5>(ctx, a);
        case 6: return k3s_launch_k<T, 6>(ctx, a);
        case 7: return k3s_launch_k<T, 7>(ctx, a);
        case 8: return k3s_launch_k<T, 8>(ctx, a);
        default: return fail(POLS_ERR_UNSUPPORTED, "rls: %d features > %d", a.k, K4_KMAX);
    }
}

int k3s_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    ctx->last_kernel = dtype == POLS_F32 ? "k3s_rls_scan_walk_f32" : "k3s_rls_scan_walk_f64";
    return dtype == POLS_F32 ? k3s_launch_t<float>(ctx, a) : k3s_launch_t<double>(ctx, a);
}

int k4_launch(pols_ctx *ctx, int dtype, const K4Args &a) {
    ctx->last_kernel = dtype == POLS_F32 ? "k4_rolling_walk_f32" : "k4_rolling_walk_f64";
    return dtype == POLS_F32 ? k4_launch_t<float>(ctx, a) : k4_launch_t<double>(ctx, a);
}

}  // namespace pols
