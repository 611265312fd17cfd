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
//   C  wave aggregates meet in LDS -> the tile's aggregate.  The tiles' carry-ins come from a scan over the tile aggregates, and that
//      needs every aggregate first: pass 1 (this kernel, MODE 0) stops after C and writes the tile's record; the records are scanned
//      (k3c_block_scan_kernel: one wave per 64 tiles, k3c_top_scan_kernel: one wave over the block totals) -- or not at all when no
//      sequence is longer than a tile, because then every tile holds a sequence start and tile t's carry-in is tile t - 1's record;
//      pass 2 (MODE 1) repeats A-B from the same rows (their second read: the Infinity Cache holds what pass 1 streamed) and goes
//      on to D.  Measured and not adopted (round 4): a single launch with a decoupled look-back over published records (tickets,
//      tagged write-through granules, two levels of records) -- 96 us on the 1M-row sequence and 488 us on 10 000 x 1 000 rows,
//      every tile waits 27-60 k cycles for the tiles running beside it; the scans done inside pass 1 by the last tile / block to
//      arrive -- pass 1 14 -> 34 us (write-through stores, drained store queues, a two-level serial tail); both record scans in one
//      16-wave workgroup -- 58 -> 79 us on the 1M-row sequence (458 KB through ONE CU's memory pipe instead of 16); the top scan folded
//      into pass 2 (every tile composes the <= 64 block records below its block itself, no third launch): 50 -> 52-56 us, the chain sits
//      in front of every tile's row loads.
//   D  every lane walks its R rows from its carry-in: A' = ff A + x x', one K x K solve per row (square-root-free L D L', LU on
//      a non-positive pivot like the reference's Cholesky -> LU chain), coefficients and predictions stored 16 bytes at a time.
// The information matrix is solved directly on every row -- never inverted and propagated -- so a diffuse prior (p0 = 1e6, the
// reference's own test setting, tests/test_ols.py:633-681) costs cond(A) eps ~ 1e-9 like any other solve.
// Bound: HBM, 8 (k + 1) bytes in + 8 (k + 1) bytes out per row (coefficients + predictions, f64).
#include "k4_rolling.hpp"
#include "k4_small.inl"
#include "dyn_out.inl"

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
        const double dp = dpp_get0<CTRL>(D);                                                 \
        _Pragma("unroll") for (int q0 = 0; q0 < NT; q0 += CH) {   /* CH partner values in flight at a time */ \
            double tp[CH];                                                                      \
            _Pragma("unroll") for (int q = 0; q < CH; ++q) tp[q] = dpp_get0<CTRL>(Tv[q0 + q < NT ? q0 + q : NT - 1]); \
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

// A composite (d, t) of the scan spread over the lanes of a wave: component q of t (q < NT) and the decay d (component NT) sit at lane
// q % 64, slot q / 64 -- one slot up to 63 components (9 features), two beyond (10 features: 65 + 1).
template <int NT>
struct K3cLaneVec {
    static constexpr int NS = (NT + 1 + 63) / 64;
    double v[NS];
    __device__ __forceinline__ void identity(int lane) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = (lane + 64 * s == NT) ? 1.0 : 0.0;
    }
    __device__ __forceinline__ void load(const double *p, int lane) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = (lane + 64 * s <= NT) ? p[lane + 64 * s] : 0.0;
    }
    __device__ __forceinline__ void store(double *p, int lane) const {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (lane + 64 * s <= NT) p[lane + 64 * s] = v[s];
    }
    template <int Q> __device__ __forceinline__ double get() const { return k1p_readlane(v[Q / 64], Q % 64); }
    __device__ __forceinline__ double get_d() const { return get<NT>(); }
    __device__ __forceinline__ void set(int q, double val, int lane) {     // q: wave-uniform
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = (lane + 64 * s == q) ? val : v[s];
    }
    // *this <- (*this) . b   under (d1, t1) . (d2, t2) = (d1 d2, d2 t1 + t2); bd = b's decay
    __device__ __forceinline__ void then(const K3cLaneVec &b, double bd, int lane) {
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = (lane + 64 * s == NT) ? v[s] * b.v[s] : fma(bd, v[s], b.v[s]);
    }
};

template <int NT, int Q>
__device__ __forceinline__ void k3c_unpack(const K3cLaneVec<NT> &c, double (&out)[NT]) {
    if constexpr (Q < NT) { out[Q] = c.template get<Q>(); k3c_unpack<NT, Q + 1>(c, out); }
}

// One wave scans a window of up to 64 records (lane l <-> record first + l, l < count): writes every record's exclusive composite --
// from the last closed record below it in the window, or, nothing closed below, with carry_in (one component per lane, slot NT the
// decay) prepended when there is one -- and whether it is still open; returns "some record of the window is closed" and leaves the
// window's inclusive composite in lane 63's (Dl, Tl).
template <int NT>
__device__ __forceinline__ bool k3c_scan_window(const double *rec, const int32_t *closed, int64_t first, int64_t count, double *excl,
                                                int32_t *open_out, double &Dl, double (&Tl)[NT], bool carry_in_valid,
                                                const K3cLaneVec<NT> &carry_in, const int lane) {
    const bool in = lane < count;
    const int64_t r = first + (in ? lane : 0);
    Dl = 1.0;
#pragma unroll
    for (int q = 0; q < NT; ++q) Tl[q] = in ? rec[r * K3C_NCP + q] : 0.0;
    if (in) Dl = rec[r * K3C_NCP + NT];
    const bool head = in && closed[r] != 0;
    const unsigned long long hm = __ballot(head);
    const unsigned long long up2 = hm & (~0ull >> (63 - lane));
    k3c_seg_scan<NT>(Dl, Tl, up2 ? 63 - __clzll(up2) : -1, lane);
    double ED2 = dpp_get0<0x138>(Dl), ET2[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) ET2[q] = dpp_get0<0x138>(Tl[q]);
    if (lane == 0) {
        ED2 = 1.0;
#pragma unroll
        for (int q = 0; q < NT; ++q) ET2[q] = 0.0;
    }
    const bool open = (hm & ((1ull << lane) - 1ull)) == 0;
    if (carry_in_valid) {                                    // (wave-uniform; the broadcasts run with every lane enabled)
        double cv[NT];
        k3c_unpack<NT, 0>(carry_in, cv);
        const double cd = carry_in.get_d();
        if (open) {
#pragma unroll
            for (int q = 0; q < NT; ++q) ET2[q] = fma(ED2, cv[q], ET2[q]);
            ED2 *= cd;
        }
    }
    if (in) {
#pragma unroll
        for (int q = 0; q < NT; ++q) excl[r * K3C_NCP + q] = ET2[q];
        excl[r * K3C_NCP + NT] = ED2;
        if (open_out) open_out[r] = open ? 1 : 0;
    }
    return hm != 0;
}

// Between the passes, launch 1 (one wave per block of 64 tiles): every tile's carry-in relative to its block's start, the block's record.
template <int NT>
__global__ void __launch_bounds__(64) k3c_block_scan_kernel(const K3cArgs a) {
    const int lane = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int64_t left = a.n_tiles - (blk << 6), cnt = left < 64 ? left : 64;
    double Dl, Tl[NT];
    K3cLaneVec<NT> none;
    none.identity(lane);
    const bool bclosed = k3c_scan_window<NT>(a.rec, a.rec_closed, blk << 6, cnt, a.carry, a.carry_open, Dl, Tl, false, none, lane);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < NT; ++q) a.brec[blk * K3C_NCP + q] = Tl[q];
        a.brec[blk * K3C_NCP + NT] = Dl;
        a.brec_closed[blk] = bclosed ? 1 : 0;
    }
}

// Between the passes, launch 2 (one wave): every block's carry-in, 64 blocks per step, the steps chained.
template <int NT>
__global__ void __launch_bounds__(64) k3c_top_scan_kernel(const K3cArgs a) {
    const int lane = threadIdx.x;
    const int64_t n_blocks = (a.n_tiles + 63) >> 6;
    K3cLaneVec<NT> chain;                                        // composite of the steps so far (from their last closed block), per component
    chain.identity(lane);
    bool chain_valid = false;
    double Dl, Tl[NT];
    for (int64_t b0 = 0; b0 < n_blocks; b0 += 64) {
        const int64_t cnt = n_blocks - b0 < 64 ? n_blocks - b0 : 64;
        const bool anyc = k3c_scan_window<NT>(a.brec, a.brec_closed, b0, cnt, a.bcarry, nullptr, Dl, Tl, chain_valid, chain, lane);
        K3cLaneVec<NT> wq;                                       // the window's composite (lane 63) spread over the lanes
        wq.identity(lane);
#pragma unroll
        for (int q = 0; q < NT; ++q) wq.set(q, k1p_readlane(Tl[q], 63), lane);
        wq.set(NT, k1p_readlane(Dl, 63), lane);
        if (anyc || !chain_valid) chain = wq;
        else chain.then(wq, wq.get_d(), lane);
        chain_valid = true;
    }
}

// LOOK-BACK form, fallback: components [Q0, Q1) of the decayed sums over the a.halo_batches 256-row batches in front of the tile that starts at
// row tbase, re-accumulated by ONE wave; rows before s0 (the sequence's first row) are left out; component q lands in lane q's cv.
template <typename T, int K, int R, int Q0, int Q1>
__device__ __forceinline__ void k3c_slow_halo_piece(const K3cArgs &a, const int64_t tbase, const int64_t s0, const int lane, const double ff, double &cv) {
    constexpr int NX = K4N<K>::NX, NP = Q1 - Q0;
    if constexpr (NP > 0) {
        double hacc[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) hacc[q] = 0.0;
        const double f252 = exp2(252.0 * a.log2ff);
#pragma unroll 1
        for (int j = a.halo_batches - 1; j >= 0; --j) {
            const int64_t base = tbase - 256 * (int64_t)(j + 1) + lane * R;
#pragma unroll
            for (int q = 0; q < NP; ++q) hacc[q] *= f252;
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                const int64_t row = base + r;
                const bool in = row >= s0;                               // (s0 >= 0)
                const int64_t ic = row < 0 ? 0 : row;
                double hx[K];
#pragma unroll
                for (int jj = 0; jj < K; ++jj) hx[jj] = in ? (double)static_cast<const T *>(a.x[jj])[ic] : 0.0;
                const double hyr = in ? (double)static_cast<const T *>(a.y)[ic] : 0.0;
#pragma unroll
                for (int p = 0; p < K; ++p) {
#pragma unroll
                    for (int q = p; q < K; ++q) {
                        constexpr int dummy = 0; (void)dummy;
                        const int idx = tri_index<K>(p, q);
                        if (idx >= Q0 && idx < Q1) hacc[idx - Q0] = fma(ff, hacc[idx - Q0], hx[p] * hx[q]);
                    }
                    if (NX + p >= Q0 && NX + p < Q1) hacc[NX + p - Q0] = fma(ff, hacc[NX + p - Q0], hx[p] * hyr);
                }
            }
        }
        const double wgt = exp2((double)(252 - lane * R) * a.log2ff);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double v = hacc[q] * wgt;
            v += dpp_get0<0x111>(v); v += dpp_get0<0x112>(v); v += dpp_get0<0x114>(v); v += dpp_get0<0x118>(v);   // row sums in lanes 15 / 31 / 47 / 63
            v += dpp_get0<0x142>(v); v += dpp_get0<0x143>(v);                                                      // lane 63: the wave's sum
            const double tot = k1p_readlane(v, 63);
            cv = lane == Q0 + q ? tot : cv;
        }
    }
}

// MODE 0 / 1: the two passes of the scan form; 2: the HALO form (step H); 3: the LOOK-BACK-ONE form -- a finite half-life makes tile t's
// carry-in a function of tile t - 1's LOCAL aggregate alone (no chain), so every tile publishes that aggregate early (step E, before its
// scan) as self-validating granules and picks its predecessor's up after its own scan; a wave whose predecessor has not published within the
// spin limit re-accumulates the halo itself -- correctness never depends on the order in which workgroups are dispatched.
template <typename T, int K, int R, int WAVES, int MODE>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : (K <= 8 ? 2 : 1)) k3c_kernel(const K3cArgs a) {
    constexpr int NX = K4N<K>::NX, NT = K4N<K>::N, NCP = K3C_NCP;
    static_assert(NT + 1 <= NCP && NT + 1 <= 128, "at most two components per lane in the cross-wave steps");
    static_assert(R == 4, "validity / start bytes travel as one 32-bit word per run");
    __shared__ double s_aw[2][WAVES][NT + 1];    // [0]: wave aggregates (slot NT: the decay); [1]: carry-in of every wave
    double (&s_agg)[WAVES][NT + 1] = s_aw[0];
    double (&s_wfull)[WAVES][NT + 1] = s_aw[1];
    __shared__ int s_closed[WAVES];              // the aggregate starts at a sequence start inside the wave
    // every lane parks its run here between A and D (its own words only: no synchronisation) -- the scan and the look-back then
    // run without R x (K + 1) row values in registers
    __shared__ T s_x[MODE >= 1 ? WAVES : 1][R * K][DYN_STAGE_STRIDE];   // (targets and predictions stay in registers)
    __shared__ double s_hagg[MODE >= 2 ? WAVES : 1][NT + 1];   // HALO form: every wave's share of the decayed sums over the rows in front of the tile; LOOK-BACK: every wave's share of the tile's own aggregate
    __shared__ int s_eclosed[MODE == 3 ? WAVES : 1];          // LOOK-BACK: the wave holds a sequence start
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#define K3C_STAMP(i) do { if (a.dbg && threadIdx.x == 64 * (WAVES - 1)) a.dbg[t * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    int64_t t = blockIdx.x;
    if constexpr (MODE == 2) {
        // consecutive tiles on one XCD (workgroup b runs on XCD b % 8): a tile's halo is its neighbours' bodies, an L2 hit when they ran there
        const int64_t per_xcd = (a.n_tiles + 7) / 8;
        t = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (t >= a.n_tiles) return;
    }
    const int64_t N = a.n_rows;
    // packed tiles (single-pass mode): the tile owns the whole sequences of rows [lo, hi) and its lanes start at lo rounded down to
    // a run -- up to three rows of the sequence before, walked from nothing and never stored
    int64_t lo = 0, hi = N, tbase = t * (int64_t)(WAVES * 64 * R);
    if (a.tile_row0) { lo = a.tile_row0[t]; hi = a.tile_row0[t + 1]; tbase = lo & ~(int64_t)3; }
    const int64_t row0 = tbase + (int64_t)((wv * 64 + lane) * R);
    const double ff = a.ff, ip0 = 1.0 / a.p0;
    K3C_STAMP(0);

    if constexpr (MODE == 2) {
        // ---- H: the HALO form (finite half-life, null-free frames; one launch, no records, no scan over the tiles).  With ff = 2^(-1/half_life) a
        // row's state forgets whatever lies more than H rows back to below ff^H: the tile's carry-in
        //     S(t0 - 1) = ff^(t0 - s) prior + sum over q in [max(s, t0 - H), t0) of ff^(t0 - 1 - q) [x_q x_q' | x_q y_q]      (s: the sequence's first row)
        // is re-accumulated from the H = 256 a.halo_batches rows in front of the tile -- sums only, no solve; the host routes here when
        // ff^H <= 2^-36 (1.5e-11 of the state H rows back: five orders below north_star's 1e-6 after a solve of condition 1e4).  The prior's own
        // decay is exact (s comes from a per-tile table).  The 256-row batches in front of the tile are dealt to the body waves newest first
        // (batch j = rows [t0 - 256 (j + 1), t0 - 256 j) goes to wave WAVES - 1 - j % WAVES; a lane: 4 consecutive rows, 16-byte loads like
        // the body's); a wave composes its batches oldest first (Horner in ff^(256 WAVES)), weights the result with its distance to the
        // tile, and the 64 lanes' sums meet in LDS -- in the slots the body rows are parked in afterwards.
        const int nbt = a.halo_batches;
        const int j0 = WAVES - 1 - wv;                                            // this wave's newest batch
        const int64_t s0 = a.tile_seq0[t];                                        // first row of the sequence that holds row t0 - 1 (t0 = 0: 0)
        double hacc[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q) hacc[q] = 0.0;
#pragma unroll 1
        for (int j = j0 + ((nbt - 1 - j0) / WAVES) * WAVES; j >= j0; j -= WAVES) {   // (nbt <= j0: no batch for this wave -- the loop does not run)
            const int64_t base = tbase - 256 * (int64_t)(j + 1) + lane * R;
            if (j >= nbt || !__any(base + R > s0)) continue;                      // the whole batch lies before the sequence (or the frame): nothing yet
            double hx[R][K], hy[R], tmp[R];
            const int64_t lb = base < 0 ? 0 : base;                               // (rows before s0 >= 0 are masked below)
#pragma unroll
            for (int jj = 0; jj < K; ++jj) {
                k3c_load_run<T, R>(a.x[jj], lb, tmp);
#pragma unroll
                for (int r = 0; r < R; ++r) hx[r][jj] = tmp[r];
            }
            k3c_load_run<T, R>(a.y, lb, hy);
            if (!__all(base >= s0)) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const bool in = base + r >= s0;
                    hy[r] = in ? hy[r] : 0.0;
#pragma unroll
                    for (int jj = 0; jj < K; ++jj) hx[r][jj] = in ? hx[r][jj] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < NT; ++q) hacc[q] *= a.ffstep;                     // ff^(256 WAVES - 4): the lane's previous run ended 256 WAVES rows earlier
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int p = 0; p < K; ++p) {
#pragma unroll
                    for (int q = p; q < K; ++q) hacc[tri_index<K>(p, q)] = fma(ff, hacc[tri_index<K>(p, q)], hx[r][p] * hx[r][q]);
                    hacc[NX + p] = fma(ff, hacc[NX + p], hx[r][p] * hy[r]);
                }
            }
        }
        // the lane's last run ends at row t0 - 256 (j0 + 1) + 4 lane + 3: ff^(t0 - 1 - that) scales it to the tile's first row
        const double wgt = exp2((double)(256 * j0 + 252 - lane * R) * a.log2ff);
#pragma unroll
        for (int q = 0; q < NT; ++q) hacc[q] *= wgt;
        // 64 lanes -> 1: two DPP steps leave every quad's sum in its last lane, those 16 lanes write a [NT][17] table, lane q adds up row q
#pragma unroll
        for (int q = 0; q < NT; ++q) hacc[q] += dpp_get0<0x111>(hacc[q]);           // row_shr:1
#pragma unroll
        for (int q = 0; q < NT; ++q) hacc[q] += dpp_get0<0x112>(hacc[q]);           // row_shr:2
        static_assert(sizeof(T) * R * K * DYN_STAGE_STRIDE >= sizeof(double) * NT * 17, "the transpose table fits the wave's parking slots");
        double *tr = reinterpret_cast<double *>(&s_x[wv][0][0]);
        if ((lane & 3) == 3) {
#pragma unroll
            for (int q = 0; q < NT; ++q) tr[q * 17 + (lane >> 2)] = hacc[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < NT) {
            double hs = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) hs += tr[lane * 17 + i];
            s_hagg[wv][lane] = hs;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __builtin_amdgcn_sched_barrier(0);
    }

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
    // zeros and its decay 1, so that every lane runs the same instruction stream (their predictions are NaN: the walk below,
    // ex.rs:640-645).
    bool st[R];
    unsigned fit = 0;                             // bit r: row r is a valid row inside the frame (its decay is ff, else 1)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        st[r] = ((sbits >> (8 * r)) & 0xffu) != 0;
        fit |= 1u << r;
    }
    if (a.valid) {
        fit = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool vr = ((vbits >> (8 * r)) & 0xffu) != 0;
            fit |= vr ? (1u << r) : 0u;
            y[r] = vr ? y[r] : 0.0;
#pragma unroll
            for (int j = 0; j < K; ++j) x[r][j] = vr ? x[r][j] : 0.0;
        }
    } else if (!__all(row0 + R <= N)) {
        fit = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) fit |= (row0 + r < N) ? (1u << r) : 0u;
    }
    auto ffr = [&](int r) -> double { return ((fit >> r) & 1u) ? ff : 1.0; };
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
                S[NX + p] = st[r] ? (a.mean0 ? a.mean0[p] * ip0 : 0.0) : S[NX + p];   // (b_0 = A_0 mean0, re-read here: the path is rare)
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
        Dl = (st[r] ? 1.0 : Dl) * ffr(r);
        add_row(Tl, x[r], y[r], ffr(r));
        if constexpr (MODE >= 1) {
#pragma unroll
            for (int j = 0; j < K; ++j) s_x[wv][r * K + j][lane] = (T)x[r][j];
        }
    }
    const unsigned long long hmask = __ballot(head);
    using U4 = __attribute__((ext_vector_type(4))) unsigned;
    [[maybe_unused]] U4 pg = {0u, 0u, 0u, 0u};               // LOOK-BACK: this lane's granule of the predecessor's record
    if constexpr (MODE == 3) if (a.early_publish) {
        // ---- E: the tile's own aggregate, EARLY (the successor needs it after ITS scan: published now it has a whole scan's time to arrive).
        // A lane's composite Tl carried to the tile's last row is Tl x ff^(rows behind its run) -- every row decays by ff on a null-free frame --
        // from the wave's last sequence start on; the 64 lanes' terms are summed like step H's (two DPP steps + a table in the parking slots,
        // which are still empty: the rows are parked below), half of the components at a time.
        static_assert(K3cLaneVec<NT>::NS == 1, "one state component per lane");
        const int hi_l = hmask ? 63 - __clzll(hmask) : 0;
        const bool incl = lane >= hi_l;
        const double W = exp2((double)(WAVES * 64 * R - (wv * 64 + lane) * R - R) * a.log2ff);
        // (the table: this wave's 2 (NT + 1) doubles of the s_agg | s_wfull region, idle until step B -- three DPP steps leave every 8 lanes' sum in
        // their last lane, those 8 lanes write a [PIECE][9] table, lane q adds up row q)
        constexpr int PIECE = (2 * (NT + 1)) / 9;
        double *tr = &s_aw[0][0][0] + wv * 2 * (NT + 1);
        if constexpr (PIECE == 0) {                           // one feature: two components, the region is too small for a table -- six DPP steps each
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                double v = incl ? Tl[q] * W : 0.0;
                v += dpp_get0<0x111>(v); v += dpp_get0<0x112>(v); v += dpp_get0<0x114>(v); v += dpp_get0<0x118>(v);
                v += dpp_get0<0x142>(v); v += dpp_get0<0x143>(v);
                if (lane == 63) s_hagg[wv][q] = v;
            }
        }
#pragma unroll
        for (int q0 = 0; q0 < (PIECE ? NT : 0); q0 += (PIECE ? PIECE : 1)) {
            double c[PIECE ? PIECE : 1];
#pragma unroll
            for (int q = 0; q < PIECE; ++q) c[q] = (q0 + q < NT && incl) ? Tl[q0 + q < NT ? q0 + q : 0] * W : 0.0;
#pragma unroll
            for (int q = 0; q < PIECE; ++q) c[q] += dpp_get0<0x111>(c[q]);           // row_shr:1
#pragma unroll
            for (int q = 0; q < PIECE; ++q) c[q] += dpp_get0<0x112>(c[q]);           // row_shr:2
#pragma unroll
            for (int q = 0; q < PIECE; ++q) c[q] += dpp_get0<0x114>(c[q]);           // row_shr:4
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if ((lane & 7) == 7) {
#pragma unroll
                for (int q = 0; q < PIECE; ++q) tr[q * 9 + (lane >> 3)] = c[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (lane < PIECE && q0 + lane < NT) {
                double hs = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) hs += tr[lane * 9 + i];
                s_hagg[wv][q0 + lane] = hs;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) s_eclosed[wv] = hmask != 0;
        __syncthreads();
        // the record: NT granules {value, tag = epoch << 1 | "holds a sequence start"} of 16 bytes, each written by ONE write-through store and
        // validated by its own tag (no flag, no fence: MI355X_MICROARCH.md, hand-off granules); tile t's granule q at (32 t + q) 16
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.gran, 0, (int)a.gran_bytes, 0x00020000);
        if (wv == WAVES - 1 && lane < NT) {
            double rec = 0.0;
            unsigned long long closed = 0;
            for (int w2 = WAVES - 1; w2 >= 0; --w2) {         // the waves from the tile's last sequence start on (their terms are already carried to the tile's end)
                rec += s_hagg[w2][lane];
                if (s_eclosed[w2]) { closed = 1; break; }
            }
            const unsigned long long vb = (unsigned long long)__double_as_longlong(rec), tg = (a.epoch << 1) | closed;
            const U4 g = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)tg, (unsigned)(tg >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (int)((t * 32 + lane) * 16), 0, /*sc1*/ 16);
        }
        if (t > 0 && lane < NT) pg = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((t - 1) * 32 + lane) * 16), 0, /*sc1*/ 16);   // first look, behind the scan
    }

    __builtin_amdgcn_sched_barrier(0);
    K3C_STAMP(1);
    // ---- B: segmented inclusive scan over the lanes
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
    double ED = dpp_get0<0x138>(Dl), ET[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) ET[q] = dpp_get0<0x138>(Tl[q]);
    if (lane == 0) ED = 1.0;
    const bool eopen = (hmask & ((1ull << lane) - 1ull)) == 0;                // no sequence start in the lanes below
    __syncthreads();

    // ---- C: across the waves of the tile, one component per lane (lane NT: the decay)
    K3cLaneVec<NT> run;                          // composite of the waves below, from the last sequence start among them
    run.identity(lane);
    bool wopen = true;                           // ... none among them: the tile's carry-in still has to be prepended
    for (int w2 = 0; w2 < wv; ++w2) {
        K3cLaneVec<NT> eq;
        eq.load(&s_agg[w2][0], lane);
        if (s_closed[w2]) { run = eq; wopen = false; }
        else run.then(eq, s_agg[w2][NT], lane);
    }
    K3C_STAMP(2);
    if constexpr (MODE == 0) {
        // pass 1: the tile's record -- its aggregate from the last sequence start inside it on, and whether there is one
        if (wv != WAVES - 1) return;
        bool tclosed = !wopen;
        {
            K3cLaneVec<NT> eq;
            eq.load(&s_agg[wv][0], lane);
            if (s_closed[wv]) { run = eq; tclosed = true; }
            else run.then(eq, s_agg[wv][NT], lane);
        }
        // (all_closed -- no sequence longer than a tile, the host knows: every tile holds a sequence start, tile t + 1's carry-in IS
        // this record and nothing scans them.  Otherwise two small launches between the passes do.)
        run.store(a.rec + t * K3C_NCP, lane);
        if (lane == 0 && !a.all_closed) a.rec_closed[t] = tclosed ? 1 : 0;
        return;
    }
    {
        // pass 2: the tile's carry-in = [its block's carry-in] . [the tiles of its block below it], both written by pass 1
        K3cLaneVec<NT> cq;
        if constexpr (MODE == 2) {
            // HALO form: the carry-in = the prior decayed over the rows since the sequence's first row + the waves' halo sums (step H)
            static_assert(K3cLaneVec<NT>::NS == 1, "one state component per lane");
            const double pw = exp2((double)(tbase - a.tile_seq0[t]) * a.log2ff) * ip0;
            double cv = 0.0;
#pragma unroll
            for (int p = 0; p < K; ++p) {
                cv = lane == tri_index<K>(p, p) ? pw : cv;
                if (a.mean0) cv = lane == NX + p ? pw * a.mean0[p] : cv;
            }
            if (lane < NT) {
#pragma unroll
                for (int w2 = 0; w2 < WAVES; ++w2) cv += s_hagg[w2][lane];
            }
            cq.v[0] = lane == NT ? 1.0 : cv;
        } else if constexpr (MODE == 3) {
            if (!a.early_publish) {
                // LATE publish (POLS_RLS_EARLY=0, A/B): the record falls out of the scan -- the waves' aggregates composed from the tile's last
                // sequence start on (what pass 1 of the scan form writes); the successor then waits a hand-off's latency at this point of ITS life
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.gran, 0, (int)a.gran_bytes, 0x00020000);
                if (wv == WAVES - 1) {
                    K3cLaneVec<NT> full = run, eq;
                    bool tclosed = !wopen;
                    eq.load(&s_agg[wv][0], lane);
                    if (s_closed[wv]) { full = eq; tclosed = true; }
                    else full.then(eq, s_agg[wv][NT], lane);
                    if (lane < NT) {
                        const unsigned long long vb = (unsigned long long)__double_as_longlong(full.v[0]), tg = (a.epoch << 1) | (tclosed ? 1ull : 0ull);
                        const U4 g = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)tg, (unsigned)(tg >> 32)};
                        __builtin_amdgcn_raw_buffer_store_b128(g, rsrc, (int)((t * 32 + lane) * 16), 0, /*sc1*/ 16);
                    }
                }
                if (t > 0 && lane < NT) pg = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((t - 1) * 32 + lane) * 16), 0, /*sc1*/ 16);
            }
            // LOOK-BACK: the carry-in = tile t - 1's record (its aggregate from its last sequence start on, carried to its last row) + -- when it holds
            // no sequence start -- the prior decayed over the rows since the sequence's first row (exact, like the halo form's); what lies
            // further back than tile t - 1 is dropped: ff^(tile rows) <= 2^-36 (the host's route condition)
            const int64_t s0 = a.tile_seq0[t];
            const double pw = exp2((double)(tbase - s0) * a.log2ff) * ip0;
            double pv = 0.0;
#pragma unroll
            for (int p = 0; p < K; ++p) {
                pv = lane == tri_index<K>(p, p) ? pw : pv;
                if (a.mean0) pv = lane == NX + p ? pw * a.mean0[p] : pv;
            }
            double cv = 0.0;
            bool got = t == 0, pclosed = false;
            if (t > 0) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.gran, 0, (int)a.gran_bytes, 0x00020000);
                for (int spins = 0;; ++spins) {
                    const unsigned long long tg = ((unsigned long long)pg[3] << 32) | pg[2];
                    if (__all(lane >= NT || (tg >> 1) == a.epoch)) { got = true; break; }
                    if (spins >= a.spin_limit) break;
                    __builtin_amdgcn_s_sleep(16);
                    if (lane < NT) pg = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((t - 1) * 32 + lane) * 16), 0, /*sc1*/ 16);
                }
                if (got) {
                    cv = lane < NT ? __longlong_as_double((long long)(((unsigned long long)pg[1] << 32) | pg[0])) : 0.0;
                    pclosed = (__builtin_amdgcn_readfirstlane((int)pg[2]) & 1) != 0;
                }
            }
            if (!got) {
                // the predecessor has not published (not dispatched yet?): this wave re-accumulates the halo itself -- a row at a time (8-byte
                // loads) and a third of the components per sweep over the rows, because the registers hold the scan's results.  The slow way,
                // taken under no dispatch order seen so far.
                constexpr int P3 = (NT + 2) / 3;
                k3c_slow_halo_piece<T, K, R, 0, P3>(a, tbase, s0, lane, ff, cv);
                k3c_slow_halo_piece<T, K, R, P3, 2 * P3>(a, tbase, s0, lane, ff, cv);
                k3c_slow_halo_piece<T, K, R, 2 * P3, NT>(a, tbase, s0, lane, ff, cv);
            }
            cq.v[0] = lane == NT ? 1.0 : (pclosed ? cv : cv + pv);
        } else if (a.tile_row0) cq.identity(lane);                               // packed: the tile starts (within a run) at a sequence start
        else if (a.all_closed) cq.load(a.rec + (t > 0 ? t - 1 : 0) * K3C_NCP, lane);   // (tile 0 starts with a sequence start: its carry-in is never used)
        else cq.load(a.carry + t * K3C_NCP, lane);
        if (MODE == 1 && !a.all_closed && a.carry_open[t] && t >= 64) {
            K3cLaneVec<NT> bq;
            if (a.fold_top) {
                // the block's carry-in = the records of the blocks below it, from the last closed one on (what k3c_top_scan_kernel writes
                // to bcarry): up to 63 independent 232-byte loads and one multiply-add each -- cheaper than a dependent launch between the passes
                bq.identity(lane);
                const int64_t blk = t >> 6;
                for (int64_t b2 = 0; b2 < blk; ++b2) {
                    K3cLaneVec<NT> eq;
                    eq.load(a.brec + b2 * K3C_NCP, lane);
                    if (a.brec_closed[b2]) bq = eq;                        // (wave-uniform)
                    else bq.then(eq, eq.get_d(), lane);
                }
            } else bq.load(a.bcarry + (t >> 6) * K3C_NCP, lane);
            bq.then(cq, cq.get_d(), lane);
            cq = bq;
        }
        if (wopen) { cq.then(run, run.get_d(), lane); run = cq; }
        run.store(&s_wfull[wv][0], lane);
    }
    __syncthreads();
    K3C_STAMP(3);
    if (eopen) {
#pragma unroll
        for (int q = 0; q < NT; ++q) ET[q] = fma(ED, s_wfull[wv][q], ET[q]);
    }

    __builtin_amdgcn_sched_barrier(0);
    // ---- D: the walk.  A row's outputs take the LDS slots its inputs were parked in; the wave's 256 rows leave together (dyn_out.inl)
    double beta[K];
    T prd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        __builtin_amdgcn_sched_barrier(0);       // one row at a time: the scheduler would otherwise start every row's products at once
        double xr[K];
#pragma unroll
        for (int j = 0; j < K; ++j) xr[j] = (double)s_x[wv][r * K + j][lane];
        const double yr = y[r];
        reset_at(ET, r);
        add_row(ET, xr, yr, ffr(r));
        ldl_solve_small<K, true>(ET, 0.0, beta);
        double pr = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            s_x[wv][r * K + j][lane] = (T)beta[j];
            pr = fma(xr[j], beta[j], pr);
        }
        prd[r] = (T)pr;
        if (a.valid) prd[r] = ((fit >> r) & 1u) ? prd[r] : nan_if<T>(1u, T(0));   // a row left out of the fit: a null prediction (make_predictions(.., is_valid), ex.rs:640-645)
        asm volatile("" : "+v"(prd[r]));           // computed HERE: left alone the compiler keeps x and beta of every row alive for it (190 VGPRs, not 150)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dyn_wave_copy_out<T, K, K>(&s_x[wv][0][0], lane, tbase + (int64_t)(wv * 64 * R), hi, static_cast<T *>(a.coef), static_cast<T *>(a.pred), lo, prd);
    K3C_STAMP(4);
    K3C_STAMP(5);
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
static int k3c_launch_k(pols_ctx *ctx, const K3cArgs &a0) {
    constexpr int R = K3C_R, WAVES = k3c_waves(K);
    K3cArgs a = a0;
    if (ctx->opt.timeline) {
        void *dbg = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_tiles, &dbg);
        if (rc) return rc;
        a.dbg = static_cast<unsigned long long *>(dbg);
    }
    if constexpr (K <= 6 && WAVES == 4) {
        if (a.tile_seq0 && a.gran) {                                  // LOOK-BACK-ONE form: one launch, tile t = workgroup t
            hipEvent_t e0, e1;
            const bool timed = timing_pair(ctx, &e0, &e1);
            hipExtLaunchKernelGGL((k3c_kernel<T, K, R, WAVES, 3>), dim3((unsigned)a.n_tiles), dim3(64 * WAVES), 0, ctx->stream, timed ? e0 : nullptr,
                                  timed ? e1 : nullptr, 0, a);
            POLS_HIP(hipGetLastError());
            if (a.dbg) return report_timeline(ctx, a.dbg, a.n_tiles, 6, "k3c_rls_rows_lookback");
            return POLS_OK;
        }
    }
    if constexpr (K <= K3C_HALO_KMAX) {
        if (a.tile_seq0) {                                            // HALO form: one launch, the grid rounded up to whole XCD rounds
            hipEvent_t e0, e1;
            const bool timed = timing_pair(ctx, &e0, &e1);            // (one kernel: stamped by its own dispatch packet, the duration rocprofv3 reports)
            hipExtLaunchKernelGGL((k3c_kernel<T, K, R, WAVES, 2>), dim3((unsigned)(((a.n_tiles + 7) / 8) * 8)), dim3(64 * WAVES), 0, ctx->stream,
                                  timed ? e0 : nullptr, timed ? e1 : nullptr, 0, a);
            POLS_HIP(hipGetLastError());
            if (a.dbg) return report_timeline(ctx, a.dbg, a.n_tiles, 6, "k3c_rls_rows_halo");
            return POLS_OK;
        }
    }
    if (a.tile_seq0) return fail(POLS_ERR_UNSUPPORTED, "rls (row-parallel, halo form): %d features > %d", K, K3C_HALO_KMAX);
    timing_begin(ctx);                                            // all launches of the call as one timed span
    a.fold_top = (a.n_tiles + 63) / 64 <= 64 ? 1 : 0;             // (POLS_RLS_ENGINE is not consulted: both forms are the same arithmetic)
    K3cArgs a1 = a;
    a1.dbg = nullptr;
    if (!a.tile_row0) hipLaunchKernelGGL((k3c_kernel<T, K, R, WAVES, 0>), dim3((unsigned)a.n_tiles), dim3(64 * WAVES), 0, ctx->stream, a1);
    if (!a.tile_row0 && !a.all_closed) {                                          // sequences longer than a tile: the records are scanned (two small launches)
        hipLaunchKernelGGL((k3c_block_scan_kernel<K4N<K>::N>), dim3((unsigned)((a.n_tiles + 63) / 64)), dim3(64), 0, ctx->stream, a1);
        if (!a.fold_top) hipLaunchKernelGGL((k3c_top_scan_kernel<K4N<K>::N>), dim3(1), dim3(64), 0, ctx->stream, a1);
    }
    hipLaunchKernelGGL((k3c_kernel<T, K, R, WAVES, 1>), dim3((unsigned)a.n_tiles), dim3(64 * WAVES), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    if (a.dbg) return report_timeline(ctx, a.dbg, a.n_tiles, 6, "k3c_rls_rows");
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
        case 9: return k3c_launch_k<T, 9>(ctx, a);
        case 10: return k3c_launch_k<T, 10>(ctx, a);
        default: return fail(POLS_ERR_UNSUPPORTED, "rls (row-parallel): %d features > %d", a.k, K4_KMAX);
    }
}

int k3c_launch(pols_ctx *ctx, int dtype, const K3cArgs &a) {
    ctx->last_kernel = a.tile_seq0 && a.gran ? (dtype == POLS_F32 ? "k3s_rls_rows_lookback_f32" : "k3s_rls_rows_lookback_f64")
                       : a.tile_seq0         ? (dtype == POLS_F32 ? "k3s_rls_rows_halo_f32" : "k3s_rls_rows_halo_f64")
                                             : (dtype == POLS_F32 ? "k3s_rls_rows_f32" : "k3s_rls_rows_f64");
    return dtype == POLS_F32 ? k3c_launch_t<float>(ctx, a) : k3c_launch_t<double>(ctx, a);
}

}  // namespace pols
