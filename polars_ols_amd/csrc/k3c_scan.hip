// k3c_scan.hip -- K3c "rls_lookback": recursive least squares for up to 8 features, ROW-PARALLEL and READ-ONCE.
//
// Replaces RecursiveLeastSquares::update + solve_recursive_least_squares (src/least_squares.rs:494-598) and the dynamic
// make_predictions (src/expressions.rs:184, 640-645) for every sequence of a frame in ONE launch that reads every input byte
// once and writes every output byte once, all of it with 16-byte accesses down the row axis.
//
// By Sherman-Morrison the reference's covariance update (P' = P/ff - k k' r, :531-540) is the decayed sum
//     A_t = ff A_{t-1} + x_t x_t',   b_t = ff b_{t-1} + x_t y_t,   beta_t = A_t^-1 b_t     (A = P^-1, b = A beta; invalid rows: no change)
// started from the prior A_0 = I / p0, b_0 = A_0 mean0 (:519-522): a scan under the associative operator
//     (d1, t1) . (d2, t2) = (d1 d2, d2 t1 + t2)      with one element (ff, [x x' | x y]) per valid row.
// Layout: the frame's rows [0, N) -- all sequences back to back, as the boundary delivers them -- are cut into tiles of
// WAVES x 64 x R rows; lane l of a wave owns R CONSECUTIVE rows (its "run"), so a wave instruction moves 64 x 16 bytes of one
// column.  A sequence start is a RESET of the scan (the prior replaces whatever came before), so sequences need no alignment to
// runs, waves or tiles: a frame of ten thousand 1 000-row sequences and one 1 000 000-row sequence are the same kernel.
//   A  every lane composes its run                                    (R x 2 NT flops, registers)
//   B  segmented inclusive scan over the 64 lanes                      (6 DPP steps x (NT + 1) components; a lane only combines
//      with a partner in its own sequence -- exec-masked, so nothing ever crosses a sequence boundary, not even a NaN)
//   C  wave aggregates meet in LDS; the tile's aggregate is PUBLISHED (write-through stores + flag) and the tile's carry-in is
//      found by DECOUPLED LOOK-BACK over the preceding tiles' aggregates / inclusive prefixes (Merrill & Garland): tiles take
//      their index from an atomic ticket, so every predecessor a tile waits for is already running
//   D  every lane walks its R rows from its carry-in: A' = ff A + x x', one K x K solve per row (square-root-free L D L', LU on
//      a non-positive pivot like the reference's Cholesky -> LU chain), coefficients and predictions stored 16 bytes at a time.
// The information matrix is solved directly on every row -- never inverted and propagated -- so a diffuse prior (p0 = 1e6, the
// reference's own test setting, tests/test_ols.py:633-681) costs cond(A) eps ~ 1e-9 like any other solve.
// Bound: HBM, 8 (k + 1) bytes in + 8 (k + 1) bytes out per row (coefficients + predictions, f64).
#include "k4_rolling.hpp"
#include "k4_small.inl"

namespace pols {

template <int NT>
__device__ __forceinline__ void k3c_seg_scan(double &D, double (&Tv)[NT], const int h, const int lane) {
    // inclusive scan over the lanes of (D, T) under (d1, t1) . (d2, t2) = (d1 d2, d2 t1 + t2), restricted to the lanes from h (the
    // highest lane <= this one that starts a segment, -1: none) on: a partner p < lane is combined iff h <= p
    const int li = lane & 15;
    constexpr int CH = NT < 12 ? NT : 12;
#define K3C_STEP(CTRL, RM, OK)                                                                  \
    {                                                                                           \
        const bool ok_ = (OK);                                                                  \
        const double dp = dpp_get<CTRL, RM>(D);                                                 \
        _Pragma("unroll") for (int q0 = 0; q0 < NT; q0 += CH) {   /* CH partner values in flight at a time */ \
            double tp[CH];                                                                      \
            _Pragma("unroll") for (int q = 0; q < CH; ++q) tp[q] = dpp_get<CTRL, RM>(Tv[q0 + q < NT ? q0 + q : NT - 1]); \
            if (ok_) {                                                                          \
                _Pragma("unroll") for (int q = 0; q < CH; ++q)                                  \
                    if (q0 + q < NT) Tv[q0 + q] = fma(D, tp[q], Tv[q0 + q]);                    \
            }                                                                                   \
        }                                                                                       \
        if (ok_) D *= dp;                                                                       \
    }
    K3C_STEP(0x111, 0xf, li >= 1 && h <= lane - 1)                 // row_shr:1
    K3C_STEP(0x112, 0xf, li >= 2 && h <= lane - 2)                 // row_shr:2
    K3C_STEP(0x114, 0xf, li >= 4 && h <= lane - 4)                 // row_shr:4
    K3C_STEP(0x118, 0xf, li >= 8 && h <= lane - 8)                 // row_shr:8
    K3C_STEP(0x142, 0xa, (lane & 16) && h <= (lane & ~15) - 1)     // row_bcast:15 -> rows 1, 3
    K3C_STEP(0x143, 0xc, lane >= 32 && h <= 31)                    // row_bcast:31 -> rows 2, 3
#undef K3C_STEP
}


template <typename T, int R>
__device__ __forceinline__ void k3c_load_run(const void *col, int64_t row0, double (&out)[R]) {
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    static_assert(R % VN == 0, "a run is a whole number of 16-byte vectors");
    const V *p = reinterpret_cast<const V *>(static_cast<const T *>(col) + row0);
#pragma unroll
    for (int i = 0; i < R / VN; ++i) {
        const V v = load_stream(p + i);
#pragma unroll
        for (int j = 0; j < VN; ++j) out[i * VN + j] = (double)vget<T>(v, j);
    }
}

template <typename T, int K, int R, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) k3c_kernel(const K3cArgs a) {
    constexpr int NX = K4N<K>::NX, NT = K4N<K>::N, NCP = K3C_NCP;
    static_assert(NT + 1 <= NCP && NT + 1 <= 64, "one component per lane in the cross-wave steps");
    static_assert(R == 4, "validity / start bytes travel as one 32-bit word per run");
    __shared__ double s_agg[WAVES][NT + 1];      // wave aggregates (slot NT: the decay)
    __shared__ int s_closed[WAVES];              // the aggregate starts at a sequence start inside the wave
    __shared__ double s_wfull[WAVES][NT + 1];    // carry-in of every wave
    __shared__ double s_win[NT + 1];             // look-back (last wave): a window's composite on its way to one component per lane
    __shared__ double s_carry[NT + 1];           // the tile's carry-in
    __shared__ long long s_tile;
    // every lane parks its run here between A and D (its own words only: no synchronisation) -- the scan and the look-back then
    // run without R x (K + 1) row values in registers
    __shared__ T s_x[WAVES][R * (K + 1)][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#define K3C_STAMP(i) do { if (a.dbg && threadIdx.x == 64 * (WAVES - 1)) a.dbg[t * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    if (threadIdx.x == 0) s_tile = (long long)(atomicAdd(a.ticket, 1ull) - a.ticket_base);
    __syncthreads();
    const int64_t t = s_tile;
    const int64_t N = a.n_rows;
    const int64_t row0 = ((t * WAVES + wv) * 64 + lane) * (int64_t)R;
    const double ff = a.ff, ip0 = 1.0 / a.p0;
    K3C_STAMP(0);

    // ---- loads: R consecutive rows of every column, the validity and sequence-start bytes of the run
    double x[R][K], y[R];
    unsigned vbits, sbits;
    if (__all(row0 + R <= N)) {
        double tmp[R];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            k3c_load_run<T, R>(a.x[j], row0, tmp);
#pragma unroll
            for (int r = 0; r < R; ++r) x[r][j] = tmp[r];
        }
        k3c_load_run<T, R>(a.y, row0, y);
        vbits = a.valid ? *reinterpret_cast<const unsigned *>(a.valid + row0) : 0x01010101u;
        sbits = *reinterpret_cast<const unsigned *>(a.start + row0);
    } else {                                     // the wave that holds the end of the frame
        vbits = 0; sbits = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool in = row0 + r < N;
            const int64_t i = in ? row0 + r : 0;
#pragma unroll
            for (int j = 0; j < K; ++j) x[r][j] = in ? (double)static_cast<const T *>(a.x[j])[i] : 0.0;
            y[r] = in ? (double)static_cast<const T *>(a.y)[i] : 0.0;
            if (in && (a.valid ? a.valid[i] != 0 : true)) vbits |= 1u << (8 * r);
            if (in && a.start[i]) sbits |= 1u << (8 * r);
        }
    }
    // Invalid rows leave the fit: their element is the identity (decay 1, nothing added).  Branch-free: the row's values become
    // zeros and its decay 1, so that every lane runs the same instruction stream (their predictions are masked by the caller's
    // post pass, ex.rs:640-645).
    bool st[R];
    double ffr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        st[r] = ((sbits >> (8 * r)) & 0xffu) != 0;
        ffr[r] = ff;
    }
    if (a.valid) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool vr = ((vbits >> (8 * r)) & 0xffu) != 0;
            ffr[r] = vr ? ff : 1.0;
            y[r] = vr ? y[r] : 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) x[r][j] = vr ? x[r][j] : 0.0;
        }
    } else if (!__all(row0 + R <= N)) {
#pragma unroll
        for (int r = 0; r < R; ++r) ffr[r] = (row0 + r < N) ? ff : 1.0;
    }
    double b0[K];
#pragma unroll
    for (int j = 0; j < K; ++j) b0[j] = a.mean0 ? a.mean0[j] * ip0 : 0.0;
    auto add_row = [&](double (&S)[NT], const double (&xr)[K], double yr, double f) { // S = f S + [x x' | x y]
#pragma unroll
        for (int p = 0; p < K; ++p) {
#pragma unroll
            for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] = fma(f, S[tri_index<K>(p, q)], xr[p] * xr[q]);
            S[NX + p] = fma(f, S[NX + p], xr[p] * yr);
        }
    };
    // a sequence starts at row r of some lane's run (rare: one wave-uniform test per row): the prior replaces that lane's state
    auto reset_at = [&](double (&S)[NT], int r) {
        if (__any(st[r])) {
#pragma unroll
            for (int p = 0; p < K; ++p) {
#pragma unroll
                for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] = st[r] ? ((p == q) ? ip0 : 0.0) : S[tri_index<K>(p, q)];
                S[NX + p] = st[r] ? b0[p] : S[NX + p];
            }
        }
    };

    // ---- A: the run as one scan element
    double Tl[NT], Dl = 1.0;
    bool head = false;
#pragma unroll
    for (int q = 0; q < NT; ++q) Tl[q] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        reset_at(Tl, r);
        head = head || st[r];
        Dl = (st[r] ? 1.0 : Dl) * ffr[r];
        add_row(Tl, x[r], y[r], ffr[r]);
#pragma unroll
        for (int j = 0; j < K; ++j) s_x[wv][r * (K + 1) + j][lane] = (T)x[r][j];
        s_x[wv][r * (K + 1) + K][lane] = (T)y[r];
    }

    __builtin_amdgcn_sched_barrier(0);
    K3C_STAMP(1);
    // ---- B: segmented inclusive scan over the lanes
    const unsigned long long hmask = __ballot(head);
    const unsigned long long upto = hmask & (~0ull >> (63 - lane));          // heads at lanes <= lane
    const int h = upto ? 63 - __clzll(upto) : -1;
    k3c_seg_scan<NT>(Dl, Tl, h, lane);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < NT; ++q) s_agg[wv][q] = Tl[q];
        s_agg[wv][NT] = Dl;
        s_closed[wv] = hmask != 0;
    }
    // exclusive value: the inclusive value of the lane below (wave_shr:1); lane 0: the identity
    double ED = dpp_get<0x138>(Dl), ET[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) ET[q] = dpp_get<0x138>(Tl[q]);
    if (lane == 0) ED = 1.0;
    const bool eopen = (hmask & ((1ull << lane) - 1ull)) == 0;                // no sequence start in the lanes below
    __syncthreads();

    // ---- C: across the waves of the tile, one component per lane (lane NT: the decay)
    const int ql = lane <= NT ? lane : NT;
    double run = (lane == NT) ? 1.0 : 0.0;       // composite of the waves below, from the last sequence start among them
    bool wopen = true;                           // ... none among them: the tile's carry-in still has to be prepended
    for (int w2 = 0; w2 < wv; ++w2) {
        const double eD = s_agg[w2][NT], eq = s_agg[w2][ql];
        if (s_closed[w2]) { run = eq; wopen = false; }
        else run = (lane == NT) ? run * eD : fma(eD, run, eq);
    }
    K3C_STAMP(2);
    // the tile's aggregate (every wave computes it: one component per lane, a handful of FMAs)
    double agg = run;
    bool tclosed = !wopen;
    for (int w2 = wv; w2 < WAVES; ++w2) {
        const double eD = s_agg[w2][NT], eq = s_agg[w2][ql];
        if (s_closed[w2]) { agg = eq; tclosed = true; }
        else agg = (lane == NT) ? agg * eD : fma(eD, agg, eq);
    }
    const bool is_prefix = tclosed || t == 0;       // the aggregate IS the inclusive prefix
    // ---- the tile's carry-in: decoupled look-back over published records, two levels (tiles, groups of K3C_GT tiles).
    // Records: one 16-byte granule {value, tag} per component, component-major (granule (q, i) at q * stride + i: a wave that reads 64
    // consecutive records' component q moves 1 KiB), written by ONE write-through (sc1) store each and validated by its own
    // tag = epoch << 3 | kind -- no flag, no ordering between the stores (MI355X_MICROARCH.md, hand-off granules; a reader that meets a
    // mixture of kinds or an old epoch reads again).  kind 1: an aggregate; kind 2: an inclusive prefix (a look-back stops there).
    //   tile record t    its aggregate -- kind 2 when a sequence starts inside the tile (nothing before it matters), else kind 1
    //   group record g   written by the LAST tile of group g to publish (an arrival counter finds it): first the composite of the
    //                    group's tile records (kind 1; kind 2 if one of them stops), then -- after its own look-back over the
    //                    group records below -- the inclusive prefix through the group (kind 2)
    // A tile's carry-in = [prefix through the group below] . [its own group's tiles below it]: at most K3C_GT - 1 + 1 records read
    // (a single look-back over tile records reads up to every resident tile's record when all tiles of one long sequence publish
    // at once: measured 114 KB of write-through traffic and 51 k cycles per tile).  Every wait is for a record whose writer holds a
    // lower ticket, i.e. is already running.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.rec, 0, (int)a.rec_bytes, 0x00020000);
    using U4 = __attribute__((ext_vector_type(4))) unsigned;
    auto publish = [&](double val, unsigned long long kind, int64_t area, int64_t stride, int64_t idx) {
        if (lane <= NT) {
            const unsigned long long vb = (unsigned long long)__double_as_longlong(val), tg = (a.epoch << 3) | kind;
            const U4 g = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)tg, (unsigned)(tg >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (int)(area + (lane * stride + idx) * 16), 0, /*sc1*/ 16);
        }
    };
    // window of up to 64 records first .. last (ascending with the lane, `last` on lane 63): afterwards lane 63 holds their composite
    // from the nearest stop on (or of all of them); records below `first` count as the identity -- and as a stop when below_stops
    auto window = [&](int64_t area, int64_t stride, int64_t first, int64_t last, bool below_stops, double &Dp, double (&Tp)[NT]) -> bool {
        const int64_t p = last - 63 + lane;
        Dp = 1.0;
#pragma unroll
        for (int q = 0; q < NT; ++q) Tp[q] = 0.0;
        bool stop = below_stops && p < first;
        if (p >= first) {
            const int voff = (int)(area + p * 16);
            const int qstride = (int)(stride * 16);
            for (;;) {
                unsigned long long tag0 = 0;
                bool same = true;
#pragma unroll
                for (int q = 0; q <= NT; ++q) {
                    const U4 g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + q * qstride, 0, /*sc1*/ 16);
                    const double val = __longlong_as_double((long long)(((unsigned long long)g[1] << 32) | g[0]));
                    const unsigned long long tg = ((unsigned long long)g[3] << 32) | g[2];
                    if (q == 0) tag0 = tg; else same = same && tg == tag0;
                    if (q < NT) Tp[q] = val; else Dp = val;
                }
                if (same && (tag0 >> 3) == a.epoch) { stop = (tag0 & 7ull) == 2ull; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        const unsigned long long m = __ballot(stop);
        const int hd = m ? 63 - __clzll(m) : -1;
        k3c_seg_scan<NT>(Dp, Tp, hd, lane);
        return hd >= 0;
    };
    // lane 63's composite -> one component per lane (through this wave's LDS words)
    auto spread = [&](double Dp, const double (&Tp)[NT]) -> double {
        __builtin_amdgcn_wave_barrier();
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < NT; ++q) s_win[q] = Tp[q];
            s_win[NT] = Dp;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const double v = s_win[ql];
        __builtin_amdgcn_wave_barrier();
        return v;
    };
    auto compose = [&](double first, double second) -> double {   // first . second (first is the EARLIER rows), one component per lane
        const double sD = k1p_readlane(second, NT);
        return (lane == NT) ? first * second : fma(sD, first, second);
    };
    if (wv == WAVES - 1) {
        constexpr int GT = K3C_GT;
        publish(agg, is_prefix ? 2ull : 1ull, 0, a.tstride, t);
        const int64_t g = t / GT, g0 = g * GT;
        const int64_t size_g = (a.n_tiles - g0 < GT) ? a.n_tiles - g0 : GT;
        unsigned arrived = 0;
        if (lane == 0) arrived = (unsigned)(atomicAdd(a.arrive + g, 1ull) - a.launch_no * (unsigned long long)size_g);
        arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
        double acc = (lane == NT) ? 1.0 : 0.0;     // the carry-in, one component per lane
        bool stopped = false;
        double Dp, Tp[NT];
        if (t > g0) {                               // this group's tiles below this one
            stopped = window(0, a.tstride, g0, t - 1, false, Dp, Tp);
            acc = spread(Dp, Tp);
        }
        K3C_STAMP(3);
        if (arrived == (unsigned)(size_g - 1)) {    // the group's last tile to publish: the group record
            const bool gstop = window(0, a.tstride, g0, g0 + size_g - 1, false, Dp, Tp);
            const double gq = spread(Dp, Tp);
            const bool gprefix = gstop || g == 0;
            publish(gq, gprefix ? 2ull : 1ull, a.grec, a.gstride, g);
            if (!gprefix) {
                double gacc = (lane == NT) ? 1.0 : 0.0;
                int64_t gb = g - 1;
                for (bool gdone = false; !gdone; gb -= 64) {
                    gdone = window(a.grec, a.gstride, 0, gb, true, Dp, Tp);
                    gacc = compose(spread(Dp, Tp), gacc);
                }
                publish(compose(gacc, gq), 2ull, a.grec, a.gstride, g);
            }
        }
        K3C_STAMP(4);
        if (!stopped && g > 0) {                    // the prefix through the group below: one granule per lane
            double gp = 0.0;
            for (;;) {
                unsigned long long tg = (a.epoch << 3) | 2ull;
                if (lane <= NT) {
                    const U4 gr = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(a.grec + (lane * a.gstride + (g - 1)) * 16), 0, /*sc1*/ 16);
                    gp = __longlong_as_double((long long)(((unsigned long long)gr[1] << 32) | gr[0]));
                    tg = ((unsigned long long)gr[3] << 32) | gr[2];
                }
                if (__all(tg == ((a.epoch << 3) | 2ull))) break;
                __builtin_amdgcn_s_sleep(2);
            }
            acc = compose(gp, acc);
        }
        if (lane <= NT) s_carry[lane] = acc;
        K3C_STAMP(5);
    }
    __syncthreads();
    {
        const double cq = s_carry[ql];
        const double runD = k1p_readlane(run, NT);
        const double full = wopen ? ((lane == NT) ? cq * run : fma(runD, cq, run)) : run;
        if (lane <= NT) s_wfull[wv][lane] = full;
    }
    __syncthreads();
    if (eopen) {
#pragma unroll
        for (int q = 0; q < NT; ++q) ET[q] = fma(ED, s_wfull[wv][q], ET[q]);
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- D: the walk; outputs leave FL rows at a time (FL x K values = a whole number of 16-byte vectors)
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N, FL = VN;
    const bool full = row0 + R <= N;
    double beta[K];
#pragma unroll
    for (int r0 = 0; r0 < R; r0 += FL) {
        T cbuf[FL * K], pbuf[FL];
#pragma unroll
        for (int rr = 0; rr < FL; ++rr) {
            const int r = r0 + rr;
            __builtin_amdgcn_sched_barrier(0);   // one row at a time: the scheduler would otherwise start every row's products at once
            double xr[K];
#pragma unroll
            for (int j = 0; j < K; ++j) xr[j] = (double)s_x[wv][r * (K + 1) + j][lane];
            const double yr = (double)s_x[wv][r * (K + 1) + K][lane];
            reset_at(ET, r);
            add_row(ET, xr, yr, ffr[r]);
            ldl_solve_small<K, true>(ET, 0.0, beta);
            double pr = 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) { cbuf[rr * K + j] = (T)beta[j]; pr = fma(xr[j], beta[j], pr); }
            pbuf[rr] = (T)pr;
        }
        if (full) {
            if (a.coef) {
                V *dst = reinterpret_cast<V *>(static_cast<T *>(a.coef) + (row0 + r0) * K);
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    V o;
#pragma unroll
                    for (int j = 0; j < VN; ++j) vset<T>(o, j, cbuf[i * VN + j]);
                    store_stream(dst + i, o);
                }
            }
            if (a.pred) {
                V o;
#pragma unroll
                for (int j = 0; j < VN; ++j) vset<T>(o, j, pbuf[j]);
                store_stream(reinterpret_cast<V *>(static_cast<T *>(a.pred) + row0 + r0), o);
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < FL; ++rr)
                if (row0 + r0 + rr < N) {
                    if (a.coef)
#pragma unroll
                        for (int j = 0; j < K; ++j) static_cast<T *>(a.coef)[(row0 + r0 + rr) * K + j] = cbuf[rr * K + j];
                    if (a.pred) static_cast<T *>(a.pred)[row0 + r0 + rr] = pbuf[rr];
                }
        }
    }
    K3C_STAMP(7);
    if (a.dbg && threadIdx.x == 64 * (WAVES - 1)) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.dbg[t * 8 + 6] = xcc;
    }
#undef K3C_STAMP
}

// sequence-start bytes from the group offsets: start[offs[g]] = 1 for every non-empty group (the caller zero-fills first)
__global__ void k3c_start_kernel(const int64_t *offs, int64_t n_groups, uint8_t *start) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_groups && offs[g] < offs[g + 1]) start[offs[g]] = 1;
}

int k3c_start_flags(pols_ctx *ctx, const int64_t *d_offs, int64_t n_groups, int64_t n_rows, uint8_t *start) {
    POLS_HIP(hipMemsetAsync(start, 0, (size_t)n_rows, ctx->stream));
    hipLaunchKernelGGL(k3c_start_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, ctx->stream, d_offs, n_groups, start);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T, int K>
static int k3c_launch_k(pols_ctx *ctx, const K3cArgs &a) {
    constexpr int R = K3C_R, WAVES = k3c_waves(K);
    hipEvent_t e0, e1;
    const bool timed = timing_pair(ctx, &e0, &e1);
    hipExtLaunchKernelGGL((k3c_kernel<T, K, R, WAVES>), dim3((unsigned)a.n_tiles), dim3(64 * WAVES), 0, ctx->stream, timed ? e0 : nullptr,
                          timed ? e1 : nullptr, 0, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
static int k3c_launch_t(pols_ctx *ctx, const K3cArgs &a) {
    switch (a.k) {
        case 1: return k3c_launch_k<T, 1>(ctx, a);
        case 2: return k3c_launch_k<T, 2>(ctx, a);
        case 3: return k3c_launch_k<T, 3>(ctx, a);
        case 4: return k3c_launch_k<T, 4>(ctx, a);
        case 5: return k3c_launch_k<T, 5>(ctx, a);
        case 6: return k3c_launch_k<T, 6>(ctx, a);
        case 7: return k3c_launch_k<T, 7>(ctx, a);
        case 8: return k3c_launch_k<T, 8>(ctx, a);
        default: return fail(POLS_ERR_UNSUPPORTED, "rls (row-parallel): %d features > %d", a.k, K4_KMAX);
    }
}

int k3c_launch(pols_ctx *ctx, int dtype, const K3cArgs &a) {
    ctx->last_kernel = dtype == POLS_F32 ? "k3s_rls_lookback_f32" : "k3s_rls_lookback_f64";
    return dtype == POLS_F32 ? k3c_launch_t<float>(ctx, a) : k3c_launch_t<double>(ctx, a);
}

}  // namespace pols
