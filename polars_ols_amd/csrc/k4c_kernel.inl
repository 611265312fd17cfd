// k4c_kernel.inl -- K4c "rolling_window_tiles": rolling-window OLS for up to 6 features on null-free frames, ROW-PARALLEL, every
// access a 16-byte one down the row axis, no cross-workgroup dependency.
//
// Replaces solve_rolling_ols (src/least_squares.rs:848-1032) + the dynamic make_predictions (src/expressions.rs:184, 695-700) for
// the frames where every row is valid (then the "drop" deque of :947-986 and the fixed window of :987-1029 are the same thing:
// S_i = alpha I + sum of the last min(i + 1, window) rows' outer products, solved from row min_periods - 1 on) and
// min_periods <= window <= 508.  Everything else -- validity masks, windows beyond the halo, the min_periods > window quirk --
// stays with the lane-per-chunk kernels of k4_rolling.hip.
//
// Layout like K3c: the frame's rows [0, N), all sequences back to back, are cut into tiles of BW x 256 rows (BW body waves, a
// lane owns 4 CONSECUTIVE rows); a sequence start resets the sums.  A window reaches at most `window` rows back, so a tile needs
// nothing from its predecessors but their last HW x 256 > window rows: HW halo waves per workgroup re-read them (an L2 hit when
// the neighbouring tile runs on the same XCD -- the tile -> workgroup map keeps consecutive tiles on one XCD), run the same
// local sums and scan, and retire.  No look-back, no tickets, no inter-workgroup traffic.
//   A  every lane sums its run: [x x' | x y] packed (NT values) + the row count since the last sequence start
//   B  segmented inclusive scan (plain sums) over the lanes, wave totals through LDS: every run's EXCLUSIVE prefix -- the sums from
//      the last sequence start (or the halo's first row) up to the run's first row -- goes to an LDS table, component-major
//   D  body lanes: S before the run = E(own run) - E'(row i0 - window) when no sequence started within the last `window` rows, where
//      E' is the table entry of the run holding that row plus the o = (-window) mod 4 rows in front of it; then per row
//      S += entering row, S -= leaving row (the same add / subtract NonWoodburyState::update performs, :707-725), one K x K solve
//      (L D L'; LU on a non-positive pivot, :732-734), coefficients and predictions stored 16 bytes at a time.
// Bound: HBM, 8 (k + 1) bytes in + 8 (k + 1) bytes out per row (f64), plus the halo re-reads (HW / BW of the input, L2 hits mostly).
#pragma once
#include "k4_rolling.hpp"
#include "k4_small.inl"
#include "dyn_out.inl"
#include "dyn_out_gather.inl"

#include <algorithm>
#include <type_traits>

namespace pols {

template <int NC>
__device__ __forceinline__ void k4c_seg_scan_add(double (&Tv)[NC], const int h, const int lane) {
    // inclusive segmented prefix SUM over the lanes: a partner p < lane is added iff no segment starts in (p, lane], i.e. h <= p
    const int li = lane & 15;
    constexpr int CH = NC < 12 ? NC : 12;
#define K4C_STEP(CTRL, RM, OK)                                                                                  \
    {                                                                                                           \
        const bool ok_ = (OK);                                                                                  \
        _Pragma("unroll") for (int q0 = 0; q0 < NC; q0 += CH) {                                                 \
            double tp[CH];                                                                                      \
            _Pragma("unroll") for (int q = 0; q < CH; ++q) tp[q] = dpp_get0<CTRL>(Tv[q0 + q < NC ? q0 + q : NC - 1]); \
            if (ok_) {                                                                                          \
                _Pragma("unroll") for (int q = 0; q < CH; ++q)                                                  \
                    if (q0 + q < NC) Tv[q0 + q] += tp[q];                                                       \
            }                                                                                                   \
        }                                                                                                       \
    }
    K4C_STEP(0x111, 0xf, li >= 1 && h <= lane - 1)
    K4C_STEP(0x112, 0xf, li >= 2 && h <= lane - 2)
    K4C_STEP(0x114, 0xf, li >= 4 && h <= lane - 4)
    K4C_STEP(0x118, 0xf, li >= 8 && h <= lane - 8)
    K4C_STEP(0x142, 0xa, (lane & 16) && h <= (lane & ~15) - 1)
    K4C_STEP(0x143, 0xc, lane >= 32 && h <= 31)
#undef K4C_STEP
}

// A row whose window sums have no L D L' factorisation, solved the reference's way (Cholesky -> LU with partial pivoting, ls.rs:732-734 /
// :277-337) by ONE wave: the window -- the rows (i - window, i] of the row's sequence, the valid ones under MASKED -- is re-summed from global
// memory, every lane runs the same K x K elimination on one copy of [A | b] in LDS (lds: K (K + 1) + NT doubles of the wave's own), the
// coefficients and the prediction replace the NaNs.  Called by the wave of the tile kernel that met the row, behind its copy-out (round 6: a
// follow-up launch over a list of such rows cost every call ~4.5 us -- a tenth of cfg4r's step -- for a list that is empty on all but
// degenerate frames).  GATHER: the rows are compacted rows -- the window is read through the source map, and the coefficients go to every
// frame row that repeats them (up to the next valid row, the end of the sequence or of the frame).
template <typename T, int K, bool MASKED, bool GATHER>
__device__ __forceinline__ void k4c_lu_fix_row(const K4cArgs &a, const int64_t i, const int lane, double *lds) {
    constexpr int NX = K4N<K>::NX, NT = K4N<K>::N;
    double (*As)[K + 1] = reinterpret_cast<double (*)[K + 1]>(lds);
    double *Ss = lds + K * (K + 1);
    int64_t lo = i - a.window + 1 < 0 ? 0 : i - a.window + 1;
    for (int64_t base = i; base >= lo; base -= 64) {       // the last sequence start at or before row i inside the window
        const int64_t j = base - lane;
        const unsigned long long m = __ballot(j >= lo && a.start[j] != 0);
        if (m) { lo = base - (int64_t)__builtin_ctzll(m); break; }
    }
    double S[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) S[q] = 0.0;
    for (int64_t j = lo + lane; j <= i; j += 64) {
        if constexpr (MASKED) { if (!a.valid[j]) continue; }
        double xr[K];
        const int64_t jf = GATHER ? (int64_t)a.src[j] : j;
#pragma unroll
        for (int p = 0; p < K; ++p) xr[p] = (double)static_cast<const T *>(a.x[p])[jf];
        const double yr = (double)static_cast<const T *>(a.y)[jf];
#pragma unroll
        for (int p = 0; p < K; ++p) {
#pragma unroll
            for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] = fma(xr[p], xr[q], S[tri_index<K>(p, q)]);
            S[NX + p] = fma(xr[p], yr, S[NX + p]);
        }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) S[q] += __shfl_xor(S[q], off);
    }
    // LU with partial pivoting on ONE copy of [A | b] in LDS (faer's partial_piv_lu -- row interchanges on the largest magnitude of the column, the solve behind ls.rs:277-337): lane c owns
    // column c (column K: the right-hand side); per pivot step every lane reads the pivot column (broadcast reads), swaps and updates its own column
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NT; ++q) Ss[q] = S[q];          // (every lane holds the totals; static indices keep S in registers)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane <= K) {
        for (int p = 0; p < K; ++p) {
            double v;
            if (lane < K) {
                const int r0 = p < lane ? p : lane, r1 = p < lane ? lane : p;
                v = Ss[tri_index<K>(r0, r1)] + (p == lane ? a.alpha : 0.0);
            } else v = Ss[NX + p];
            As[p][lane] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int j = 0; j < K; ++j) {
        int pv = j;
        double best = fabs(As[j][j]);
        for (int r2 = j + 1; r2 < K; ++r2) { const double v = fabs(As[r2][j]); if (v > best) { best = v; pv = r2; } }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane <= K && pv != j) { const double t0 = As[j][lane]; As[j][lane] = As[pv][lane]; As[pv][lane] = t0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const double d = As[j][j];
        double f[K];
        for (int r2 = j + 1; r2 < K; ++r2) f[r2] = As[r2][j] / d;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane <= K && lane > j) {
            const double top = As[j][lane];
            for (int r2 = j + 1; r2 < K; ++r2) As[r2][lane] -= f[r2] * top;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    double beta[K];
    for (int p = K - 1; p >= 0; --p) {
        double sacc = As[p][K];
        for (int q = p + 1; q < K; ++q) sacc -= As[p][q] * beta[q];
        beta[p] = sacc / As[p][p];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        double pr = 0.0;
        bool vi = true;
        if constexpr (MASKED) vi = a.valid[i] != 0;
        const int64_t fi = GATHER ? (int64_t)a.src[i] : i;
        for (int p = 0; p < K; ++p) {
            if (a.coef) static_cast<T *>(a.coef)[fi * K + p] = (T)beta[p];
            pr = fma(vi ? (double)static_cast<const T *>(a.x[p])[fi] : 0.0, beta[p], pr);
        }
        if (a.pred) static_cast<T *>(a.pred)[fi] = vi ? (T)pr : nan_if<T>(1u, T(0));     // (a masked row's prediction is a null)
        if constexpr (GATHER) {                                                           // the rows left out behind it repeat its coefficients
            const int64_t fe = i + 1 < a.n_rows ? (int64_t)a.src[i + 1] : a.n_frame;
            for (int64_t f = fi + 1; f < fe && !a.fstart[f] && a.coef; ++f)
                for (int p = 0; p < K; ++p) static_cast<T *>(a.coef)[f * K + p] = (T)beta[p];
        }
    }

}

// MASKED: the FIXED window over rows with validity bytes ("drop_window", ls.rs:987-1029) on frames where no row older than the window
// is still in a warm-up sum (api.hip checks): an invalid row is a zero row -- it enters and leaves the sums as nothing -- every row is
// solved from S_i = E(i) - E(i - window), and the rows the reference does NOT solve (before the warm-up row: NaN; a closed
// n_valid_window gate: the last solved row's coefficients) are rewritten afterwards by the fill pass (k4cm_fill.hip) from a per-row table.
//
// SELF (round 6, windows up to 256 rows on frames that do not pack): NO halo wave.  Four body waves, 1 024-row tiles; only the lanes whose
// window starts in front of the tile -- the first ceil(window / 4) lanes of the tile's first wave -- need anything from outside it, and what
// they need is (a) their LEAVING rows, which they load themselves (L2 hits), and (b) the sum of the rows between their window's first row
// and the tile's first row: a suffix sum over those same leaving rows, one 64-lane scan on that wave alone (total - prefix), parked in LDS
// next to the table.  A quarter of every workgroup used to be halo: 1 303 tiles of 768 rows on 512 tile slots (three rounds) become 977
// tiles of 1 024 rows (two).  Which wave takes the tile's first 256 rows rotates with the tile so that the extra work spreads over the SIMDs.
//
// GATHER (round 6, the "drop" family on frames WITH nulls, ls.rs:947-986): the rows of this kernel are the frame's VALID rows, read through
// the source map a.src from the frame's own columns (8-byte loads: a run's four rows are neighbours in the frame but for the nulls between
// them), and every wave writes the FRAME rows between its first valid row and the next wave's -- coefficients forward-filled inside the
// sequence, NaN predictions on the rows left out (dyn_out_gather.inl).  The compacted copy of the columns and the expansion pass are gone.
template <typename T, int K, int HW, int WAVES, bool MASKED, bool SELF = false, bool GATHER = false>
__global__ void __launch_bounds__(64 * WAVES, (K <= 6 || (K == 7 && HW == 0 && !SELF)) ? 2 : 1) k4c_kernel(const K4cArgs a) {
    static_assert(!SELF || (HW == 0 && WAVES == 4), "the own-halo form: four body waves");
    static_assert(!GATHER || !MASKED, "the gathered rows are all valid");
    constexpr int BW = WAVES - HW, R = 4, RUNS = WAVES * 64;
    constexpr int NX = K4N<K>::NX, NT = K4N<K>::N, NC = NT + 1;   // slot NT: rows since the last sequence start (or the halo's first row)
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *s_E = reinterpret_cast<double *>(smem);              // [NC][RUNS]: exclusive prefix of every run
    // (the same region first holds the tile's rows -- 4 (K + 1) x RUNS values of the batch dtype -- for the hand-over of the leaving rows)
    constexpr size_t TAB_B = sizeof(double) * NC * RUNS > sizeof(T) * 4 * (K + 1) * RUNS ? sizeof(double) * NC * RUNS : sizeof(T) * 4 * (K + 1) * RUNS;
    double *s_agg = reinterpret_cast<double *>(smem + TAB_B);    // [WAVES][NC]: wave totals (from the wave's last sequence start on)
    int *s_closed = reinterpret_cast<int *>(s_agg + WAVES * NC); // [WAVES]
    // SELF: the rows in front of the tile -- [4 (K + 1)][64] values, zero where they do not count, then [NC][64] sums -- and how many of a lane's count
    [[maybe_unused]] double *s_H = reinterpret_cast<double *>(smem + TAB_B + sizeof(double) * WAVES * NC + 64);
    constexpr int HN = NC > 4 * (K + 1) ? NC : 4 * (K + 1);
    [[maybe_unused]] int *s_nin = reinterpret_cast<int *>(s_H + HN * 64);
    const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
#define K4C_STAMP(i) do { if (a.dbg && threadIdx.x == 64 * (WAVES - 1)) a.dbg[t * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
    // consecutive tiles on one XCD (workgroup b runs on XCD b % 8): a tile's halo is its neighbour's body
    const int64_t per_xcd = (a.n_tiles + 7) / 8;
    const int64_t t = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (t >= a.n_tiles) return;
    const int64_t N = a.n_rows, w = a.window;
    // HW == 0: packed tiles -- the tile owns the whole sequences of rows [lo, hi) and starts at lo rounded down to a run
    int64_t lo = 0, hi = N, hs = t * (BW * 256) - HW * 256;      // hs: the halo's first row (may be negative)
    if constexpr (HW == 0 && !SELF) { lo = a.tile_row0[t]; hi = a.tile_row0[t + 1]; hs = lo & ~(int64_t)3; }
    const int wv = SELF ? ((wq + (int)(t & 3)) & 3) : wq;        // the 256-row block of the tile this wave takes (SELF: rotates with the tile)
    const int u = wv * 64 + lane;                                // this lane's run
    const int64_t i0 = hs + (int64_t)u * R;
    const bool inside = i0 >= 0 && i0 + R <= N;
    K4C_STAMP(0);
    const int sh = (int)((w + 3) / 4);                                         // runs between this run and the one holding row i0 - window
    const int o = (int)(sh * 4 - w);                                           // rows of that run in FRONT of row i0 - window
    const int up = u - sh < 0 ? 0 : u - sh;                                    // table index: u - sh >= 0 for body lanes behind a halo (it covers `window`
                                                                               // rows); packed tiles: a negative one is never looked up (use_ep below is false)
    const int64_t rho_run = hs + (int64_t)(u - sh) * R;                        // first row of the run holding row i0 - window
    auto load_row = [&](int64_t i, double (&xr)[K], double &yr) {              // one row, clamped into the frame (callers mask)
        int64_t ic = i < 0 ? 0 : (i >= N ? N - 1 : i);
        if constexpr (GATHER) ic = a.src[ic];
        bool vr = true;
        if constexpr (MASKED) vr = a.valid[ic] != 0;
#pragma unroll
        for (int j = 0; j < K; ++j) xr[j] = vr ? (double)static_cast<const T *>(a.x[j])[ic] : 0.0;
        yr = vr ? (double)static_cast<const T *>(a.y)[ic] : 0.0;
    };
    auto add_row = [&](double (&S)[NC], const double (&xr)[K], double yr, double sign) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
#pragma unroll
            for (int q = p; q < K; ++q) S[tri_index<K>(p, q)] = fma(sign * xr[p], xr[q], S[tri_index<K>(p, q)]);
            S[NX + p] = fma(sign * xr[p], yr, S[NX + p]);
        }
    };

    // ---- A: the run's rows (rows outside the frame: zeros, no sequence start)
    double x[R][K], y[R];
    unsigned sbits = 0;
    [[maybe_unused]] unsigned vmask = 0;                                        // MASKED: byte r != 0 -- row i0 + r is valid
    [[maybe_unused]] int64_t g_ra = 0, g_rb = 0;
    [[maybe_unused]] unsigned g_vb[DYN_GATHER_PRE], g_sb[DYN_GATHER_PRE];
    [[maybe_unused]] const int64_t gwrow0 = hs + (int64_t)wv * 64 * R;                       // GATHER: the compacted rows [gc0, gc1) whose outputs this wave holds
    [[maybe_unused]] const int64_t gc0 = gwrow0 > lo ? gwrow0 : lo, gc1 = gwrow0 + 256 < hi ? gwrow0 + 256 : hi;
    if constexpr (GATHER) {
        // the run's four rows through the source map (one 16-byte load of it when the run lies inside the frame), then a row at a time
        int32_t sr[R];
        bool in[R];
        if (inside) {
            const int4 s4 = *reinterpret_cast<const int4 *>(a.src + i0);
            sr[0] = s4.x; sr[1] = s4.y; sr[2] = s4.z; sr[3] = s4.w;
#pragma unroll
            for (int r = 0; r < R; ++r) in[r] = true;
            sbits = *reinterpret_cast<const unsigned *>(a.start + i0);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t i = i0 + r;
                in[r] = i >= 0 && i < N;
                sr[r] = in[r] ? a.src[i] : 0;
                if (in[r] && a.start[i]) sbits |= 1u << (8 * r);
            }
        }
        // (the frame rows this wave will write and their validity / sequence-start bytes: two more reads of the source map in this round trip,
        // the bytes in the next one, behind the rows)
        dyn_gather_range(g_ra, g_rb, g_vb, g_sb, lane, gc0, gc1, a.src, N, a.n_frame, a.fvalid, a.fstart);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < K; ++j) x[r][j] = in[r] ? (double)static_cast<const T *>(a.x[j])[sr[r]] : 0.0;
            y[r] = in[r] ? (double)static_cast<const T *>(a.y)[sr[r]] : 0.0;
        }
    } else if (__all(inside)) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const V *p = reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + i0);
#pragma unroll
            for (int i = 0; i < R / VN; ++i) {
                const V v = p[i];                        // (not a streaming load: the rows come back as LEAVING rows, from L2)
#pragma unroll
                for (int e = 0; e < VN; ++e) x[i * VN + e][j] = (double)vget<T>(v, e);
            }
        }
        const V *p = reinterpret_cast<const V *>(static_cast<const T *>(a.y) + i0);
#pragma unroll
        for (int i = 0; i < R / VN; ++i) {
            const V v = p[i];
#pragma unroll
            for (int e = 0; e < VN; ++e) y[i * VN + e] = (double)vget<T>(v, e);
        }
        sbits = *reinterpret_cast<const unsigned *>(a.start + i0);
        if constexpr (MASKED) {
            const unsigned vbits = *reinterpret_cast<const unsigned *>(a.valid + i0);
            vmask = vbits;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool vr = ((vbits >> (8 * r)) & 0xffu) != 0;
                y[r] = vr ? y[r] : 0.0;
#pragma unroll
                for (int j = 0; j < K; ++j) x[r][j] = vr ? x[r][j] : 0.0;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r;
            bool in = i >= 0 && i < N;
            const int64_t ic = in ? i : 0;
            if (in && a.start[ic]) sbits |= 1u << (8 * r);
            if constexpr (MASKED) { in = in && a.valid[ic] != 0; if (in) vmask |= 1u << (8 * r); }
#pragma unroll
            for (int j = 0; j < K; ++j) x[r][j] = in ? (double)static_cast<const T *>(a.x[j])[ic] : 0.0;
            y[r] = in ? (double)static_cast<const T *>(a.y)[ic] : 0.0;
        }
    }
    [[maybe_unused]] unsigned solbits = 0;                                     // MASKED: byte r != 0 -- the reference solves row i0 + r (dyn_prep.hip rm_rows_kernel)
    if constexpr (MASKED) {
        if (inside) solbits = *reinterpret_cast<const unsigned *>(a.solved + i0);
        else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (i0 + r >= 0 && i0 + r < N && a.solved[i0 + r]) solbits |= 1u << (8 * r);
        }
    }
    double xo[R][K], yo[R];
    [[maybe_unused]] bool pre[R] = {};                                         // SELF: leaving row r lies in FRONT of the tile (loaded here, not handed over)
    constexpr int NP = (NC + 3) / 4;                                           // SELF: components of the sums in front of the tile per wave
    [[maybe_unused]] double Pp[NP];
    if constexpr (SELF) {
        if (wv == 0) {
            // The rows in front of the tile that this wave's windows reach: lane u's leaving rows i0 - window .. + 3 where they lie before
            // the tile's first row.  They stay in xo / yo for the walk.  Their sum from the window's first row (or the last sequence start
            // in front of the tile, whichever is later) up to the tile's first row is what the lane's prefix E(own run) lacks: a suffix sum
            // over the lanes -- its NC components are shared out over the FOUR waves (the rows travel through LDS, zeroed where they do
            // not count), so that no wave waits for 28 components' worth of scan on this one.
            unsigned hb = 0;                                                   // bit r: such a row starts a sequence
            uint8_t sb[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {                                      // every load first (clamped into the frame, no branch between them): ONE round trip
                const int64_t gi = hs + (u - sh) * R + o + r;
                load_row(gi, xo[r], yo[r]);
                sb[r] = a.start[gi < 0 ? 0 : (gi >= N ? N - 1 : gi)];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int g = (u - sh) * R + o + r;                            // row index relative to the tile's first row
                pre[r] = g < 0 && hs + g >= 0;
                hb |= (pre[r] && sb[r]) ? 1u << r : 0u;
            }
            const unsigned long long hm = __ballot(hb != 0);
            int gstar = -(1 << 30);                                            // the last sequence start in front of the tile (as far back as the windows reach)
            if (hm) {
                const int vs = 63 - __clzll(hm);
                const unsigned hbs = (unsigned)__shfl((int)hb, vs);
                gstar = (vs - sh) * R + o + (31 - __clz(hbs));
            }
            int nin = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool inc = pre[r] && (u - sh) * R + o + r >= gstar;
                nin += inc ? 1 : 0;
#pragma unroll
                for (int j = 0; j < K; ++j) s_H[(r * (K + 1) + j) * 64 + lane] = inc ? xo[r][j] : 0.0;
                s_H[(r * (K + 1) + K) * 64 + lane] = inc ? yo[r] : 0.0;
            }
            s_nin[lane] = nin;
        }
    }
    // The LEAVING rows of a body lane's run -- rows i0 - window .. + 3: rows (o + r) of the runs u - sh and u - sh + 1 -- are rows other
    // lanes of this workgroup have just loaded: they change hands through LDS (the region the prefix table takes over after the next
    // barrier: [row of the run][column][run], batch dtype).  Loaded from global memory instead (round 4 until then: 16-byte loads, L2
    // hits) they went through the CU's memory pipe a second time -- a third of the tile's traffic on the resource this kernel is bound
    // by (10.5 bytes per clock per CU with them).
    {
        T *s_X = reinterpret_cast<T *>(smem);
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < K; ++j) s_X[(size_t)(r * (K + 1) + j) * RUNS + u] = (T)x[r][j];
            s_X[(size_t)(r * (K + 1) + K) * RUNS + u] = (T)y[r];
        }
        __syncthreads();
        if constexpr (SELF) {
            // this wave's components q = 4 i + wq of the 64 lanes' sums in front of the tile: local sums, an inclusive prefix over the lanes,
            // total - exclusive prefix = the lanes from this one on; parked in registers until the rows' LDS can take them (after the next barrier)
            double hx[R][K], hy[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < K; ++j) hx[r][j] = s_H[(r * (K + 1) + j) * 64 + lane];
                hy[r] = s_H[(r * (K + 1) + K) * 64 + lane];
            }
            const double hn = (double)s_nin[lane];
            auto part = [&](auto vc) {
                constexpr int V = decltype(vc)::value;
#pragma unroll
                for (int i = 0; i < NP; ++i) Pp[i] = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int p2 = 0; p2 < K; ++p2) {
#pragma unroll
                        for (int q2 = p2; q2 < K; ++q2)
                            if (tri_index<K>(p2, q2) % 4 == V) Pp[tri_index<K>(p2, q2) / 4] = fma(hx[r][p2], hx[r][q2], Pp[tri_index<K>(p2, q2) / 4]);
                        if ((NX + p2) % 4 == V) Pp[(NX + p2) / 4] = fma(hx[r][p2], hy[r], Pp[(NX + p2) / 4]);
                    }
                }
                if (NT % 4 == V) Pp[NT / 4] = hn;
            };
            switch (wq) {
                case 0: part(std::integral_constant<int, 0>{}); break;
                case 1: part(std::integral_constant<int, 1>{}); break;
                case 2: part(std::integral_constant<int, 2>{}); break;
                default: part(std::integral_constant<int, 3>{}); break;
            }
            k4c_seg_scan_add<NP>(Pp, -1, lane);                                // (no segments: the rows before the last sequence start are zeros)
#pragma unroll
            for (int i = 0; i < NP; ++i) Pp[i] = readlane63(Pp[i]) - dpp_get0<0x138>(Pp[i]);
        }
        if (wv >= HW) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int g = (u - sh) * R + o + r;                                  // row index inside [halo | body]; negative: never used (the
                g = g < 0 ? 0 : g;                                             // sequence then starts inside the tile, `sub` below stays false)
                const int run = g >> 2, rr = g & 3;
                if (SELF && pre[r]) continue;                                  // (already here, from in front of the tile)
#pragma unroll
                for (int j = 0; j < K; ++j) xo[r][j] = (double)s_X[(size_t)(rr * (K + 1) + j) * RUNS + run];
                yo[r] = (double)s_X[(size_t)(rr * (K + 1) + K) * RUNS + run];
            }
        }
    }
    bool st[R];
#pragma unroll
    for (int r = 0; r < R; ++r) st[r] = ((sbits >> (8 * r)) & 0xffu) != 0;
    auto reset_if = [&](double (&S)[NC], bool cond, bool any) {   // the sums start over at a sequence start
        if (any) {
#pragma unroll
            for (int q = 0; q < NC; ++q) S[q] = cond ? 0.0 : S[q];
        }
    };
    double Tl[NC];
    bool head = false;
#pragma unroll
    for (int q = 0; q < NC; ++q) Tl[q] = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        reset_if(Tl, st[r], __any(st[r]));
        head = head || st[r];
        add_row(Tl, x[r], y[r], 1.0);
        Tl[NT] += 1.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    K4C_STAMP(1);

    // ---- B: segmented inclusive prefix over the lanes, then over the waves; the exclusive prefixes go to the LDS table
    const unsigned long long hmask = __ballot(head);
    const unsigned long long upto = hmask & (~0ull >> (63 - lane));
    const int h = upto ? 63 - __clzll(upto) : -1;
    k4c_seg_scan_add<NC>(Tl, h, lane);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < NC; ++q) s_agg[wv * NC + q] = Tl[q];
        s_closed[wv] = hmask != 0;
    }
    double ET[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) ET[q] = dpp_get0<0x138>(Tl[q]);                // wave_shr:1 -- the lane below's inclusive value
    const bool eopen = (hmask & ((1ull << lane) - 1ull)) == 0;                // no sequence start in the lanes below
    __syncthreads();
    if (eopen) {                                                               // prepend the waves below, from their last sequence start
#pragma unroll 1
        for (int w2 = wv - 1; w2 >= 0; --w2) {
#pragma unroll
            for (int q = 0; q < NC; ++q) ET[q] += s_agg[w2 * NC + q];
            if (s_closed[w2]) break;
        }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) s_E[q * RUNS + u] = ET[q];
    if constexpr (SELF) {                                                      // (the rows in front of the tile have been read by every wave: two barriers ago)
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (4 * i + wq < NC) s_H[(4 * i + wq) * 64 + lane] = Pp[i];
    }
    __syncthreads();
    if (wv < HW) return;                                                       // halo waves are done
    K4C_STAMP(2);

    // ---- D: body lanes.  S before the run, then one add / subtract / solve per row.
    double S[NC];
    double cnt_excl = ET[NT];                                                  // rows since the last sequence start, before this run
    // (exact when a sequence starts inside the tile or its halo; otherwise the rows since the halo's first row: >= 256 HW >= window + o + 1)
    // The part of the prefix E(own run) that lies before row i0 - window goes: the table entry of the run holding that row when the
    // sequence started before that run, and the rows of that run in front of row i0 - window that the sequence reaches back to.
    const bool use_ep = cnt_excl >= (double)(w + o + 1);
#pragma unroll
    for (int q = 0; q < NC; ++q) S[q] = ET[q] - (use_ep ? s_E[q * RUNS + up] : 0.0);
    if constexpr (SELF) {
        if (wv == 0) {                                                         // the part of the window in front of the tile (no sequence start in the tile before this run)
            const bool reach = eopen && (int64_t)u * R < w;
#pragma unroll
            for (int q = 0; q < NT; ++q) S[q] += reach ? s_H[q * 64 + lane] : 0.0;
            cnt_excl += reach ? s_H[NT * 64 + lane] : 0.0;                     // (at most `window`: use_ep above stays false, the rows-in-front loop below idle)
        }
    }
    double cnt = cnt_excl;
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);
    const T qnan = nan_if<T>(1u, T(0));
#pragma unroll 1
    for (int r = 0; r < o; ++r) {                                             // (window not a multiple of 4: up to 3 rows, one L2 round trip each)
        const bool gone = cnt_excl >= (double)(w + o - r);                     // the sequence reaches back to row rho_run + r
        if (__any(gone)) {
            double xf[K], yf;
            load_row(rho_run + r, xf, yf);
            if (gone) add_row(S, xf, yf, -1.0);
        }
    }
    K4C_STAMP(3);
    // the prefix table has been read by every body lane: its LDS becomes the staging area of the outputs (dyn_out.inl) -- a row's
    // coefficients and prediction wait there until the wave's 256 rows leave as whole lines
    __syncthreads();                                                           // (the halo waves have retired: they no longer count)
    T *stage = reinterpret_cast<T *>(smem) + (size_t)(wv - HW) * (R * (K + 1) * DYN_STAGE_STRIDE);
    double beta[K];
    unsigned failed = 0;                                                       // bit r: row i0 + r is one the reference solves and its sums had no L D L' factorisation
#pragma unroll
    for (int r = 0; r < R; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        if (__any(st[r])) {
#pragma unroll
            for (int q = 0; q < NC; ++q) S[q] = st[r] ? 0.0 : S[q];
            cnt = st[r] ? 0.0 : cnt;
        }
        add_row(S, x[r], y[r], 1.0);
        cnt += 1.0;
        const bool sub = cnt >= (double)(w + 1);                               // the window is full: row i - window leaves
        if (__any(sub)) {
            if (sub) add_row(S, xo[r], yo[r], -1.0);
            cnt = sub ? (double)w : cnt;                                       // (rows in the window)
        }
        // Cholesky -> LU in the reference (:732-734).  A non-positive pivot here means a window without K independent rows (or one
        // whose X'X is singular to working precision): the reference's LU then divides by a zero or noise pivot and returns
        // inf / NaN / 1e15-sized numbers.  This kernel reports such rows as NaN -- no LU, hence no scratch memory in the launch;
        // POLS_ROLLING_ENGINE=chunk has the LU.
        const bool ok = ldl_solve_small<K, false>(S, a.alpha, beta);
        const bool territory = MASKED ? ((solbits >> (8 * r)) & 0xffu) != 0 : (cnt >= (double)a.min_periods || sub);   // a row the reference solves
        const bool good = MASKED ? ok : (territory && ok);          // (MASKED: the fill pass rewrites the rows that are not solved here)
        failed |= (territory && !ok) ? (1u << r) : 0u;   // no factorisation: the reference goes on to LU (ls.rs:732-734) -- see below
        double pr = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            stage[(r * (K + 1) + j) * DYN_STAGE_STRIDE + lane] = good ? (T)beta[j] : qnan;
            pr = fma(x[r][j], beta[j], pr);
        }
        bool pgood = good;
        if constexpr (MASKED) pgood = good && ((vmask >> (8 * r)) & 0xffu) != 0;   // a masked row's prediction is a null (src/expressions.rs:695-700)
        stage[(r * (K + 1) + K) * DYN_STAGE_STRIDE + lane] = pgood ? (T)pr : qnan;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if constexpr (GATHER) {
        dyn_wave_copy_out_gather<T, K, K + 1>(stage, lane, gwrow0, gc0, gc1, g_ra, g_rb, g_vb, g_sb, a.fvalid, a.fstart, coef, pred);
    } else {
        const T none[4] = {};
        dyn_wave_copy_out<T, K, K + 1>(stage, lane, hs + (int64_t)wv * 64 * R, hi, coef, pred, lo, none);
    }
    // rows whose window sums had no factorisation: this wave runs the reference's LU on them itself (rare -- degenerate windows only; the staging
    // area has left with the copy-out and serves as the elimination's scratch)
    if (__any(failed != 0)) {
        unsigned mine = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = i0 + r;
            if (((failed >> r) & 1u) && row >= lo && row < hi) mine |= 1u << r;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        unsigned long long fm = __ballot(mine != 0);
        while (fm) {                                                        // (wave-uniform)
            const int L = __builtin_ctzll(fm);
            fm &= fm - 1ull;
            const unsigned bits = (unsigned)__shfl((int)mine, L);
            const int64_t rb0 = hs + (int64_t)(wv * 64 + L) * R;
            for (int r = 0; r < R; ++r)
                if ((bits >> r) & 1u) k4c_lu_fix_row<T, K, MASKED, GATHER>(a, rb0 + r, lane, reinterpret_cast<double *>(stage));
        }
    }
    K4C_STAMP(4);
    K4C_STAMP(5);
    if (a.dbg && threadIdx.x == 64 * (WAVES - 1)) a.dbg[t * 8 + 6] = 0;
#undef K4C_STAMP
}

template <typename T, int K, int HW, bool MASKED, bool SELF = false, bool GATHER = false>
static int k4c_launch_h(pols_ctx *ctx, const K4cArgs &a0) {
    constexpr int NC = K4N<K>::N + 1;
    // Waves per workgroup.  The prefix table is NC x 64 WAVES doubles of LDS (K = 6: 14 KiB per wave) and a lane holds its entering
    // and leaving rows through the scan (229 VGPRs: two waves per SIMD).  One halo wave: FOUR waves, two workgroups per CU whose
    // phases interleave (one streams rows in while the other walks) -- measured against one 8-wave workgroup, which repeats
    // 1 / 7 instead of 1 / 3 of the rows: 1M rows 54.5 -> 43.5 us, 10 000 x 1 000 rows 390 -> 326 us.  Two halo waves: eight.
    constexpr int WAVES = HW <= 1 ? 4 : 8;
    K4cArgs a = a0;
    const int64_t tile_rows = (int64_t)(WAVES - HW) * 256;
    static_assert(HW != 0 || (WAVES - HW) * 256 == K4C_PACKED_ROWS, "the packed tile map is built for this tile");
    a.n_tiles = HW == 0 && !SELF ? a.n_packed : (a.n_rows + tile_rows - 1) / tile_rows;
    const int64_t per_xcd = (a.n_tiles + 7) / 8;
    const size_t tab = std::max(sizeof(double) * (size_t)NC * 64 * WAVES, sizeof(T) * (size_t)4 * (K + 1) * 64 * WAVES);   // the tile's rows, then the prefix table
    const size_t lds = std::max(tab + sizeof(double) * WAVES * NC + 64 + (SELF ? sizeof(double) * std::max(NC, 4 * (K + 1)) * 64 + 256 : 0),     // ... + wave totals (+ SELF: the sums in front of the tile)
                                (size_t)(WAVES - HW) * 4 * (K + 1) * DYN_STAGE_STRIDE * sizeof(T));              // ... reused as the output staging area
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k4c_kernel<T, K, HW, WAVES, MASKED, SELF, GATHER>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done(ctx->device);
    }
    if (ctx->opt.timeline) {
        void *dbg = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_tiles, &dbg);
        if (rc) return rc;
        a.dbg = static_cast<unsigned long long *>(dbg);
    }
    hipEvent_t e0, e1;
    const bool timed = timing_pair(ctx, &e0, &e1);
    hipExtLaunchKernelGGL((k4c_kernel<T, K, HW, WAVES, MASKED, SELF, GATHER>), dim3((unsigned)(per_xcd * 8)), dim3(64 * WAVES), (unsigned)lds, ctx->stream, timed ? e0 : nullptr,
                          timed ? e1 : nullptr, 0, a);
    POLS_HIP(hipGetLastError());
    if (a.dbg) return report_timeline(ctx, a.dbg, a.n_tiles, 6, "k4c_rolling_tiles");
    return POLS_OK;
}

template <typename T, int K, bool MASKED, bool GATHER = false>
static int k4c_launch_k(pols_ctx *ctx, const K4cArgs &a) {
    if (a.tile_row0) return k4c_launch_h<T, K, 0, MASKED, false, GATHER>(ctx, a);                // packed tiles: no halo
    // no halo wave: the tile's first wave reaches in front of the tile itself (windows up to 256 rows; at 10 features the table and the parked sums
    // are more than a CU's LDS).  POLS_ROLLING_ENGINE=halowave keeps the halo wave (A/B).
    if constexpr (K <= 9) {
        if (a.window <= 256 && ctx->opt.rolling_engine != 4) return k4c_launch_h<T, K, 0, MASKED, true, GATHER>(ctx, a);
    }
    if (a.window <= 252) return k4c_launch_h<T, K, 1, MASKED, false, GATHER>(ctx, a);            // 256 HW >= 4 ceil(window / 4) + 1
    if constexpr (K <= 6) return k4c_launch_h<T, K, 2, MASKED, false, GATHER>(ctx, a);
    // (7 / 8 features need more than 256 registers: one four-wave workgroup per CU; the eight-wave two-halo form would spill)
    return fail(POLS_ERR_UNSUPPORTED, "rolling (row-parallel): window %lld with %d features needs packed tiles", (long long)a.window, K);
}

template <typename T, bool MASKED, bool GATHER = false>
static int k4c_launch_t(pols_ctx *ctx, const K4cArgs &a) {
    switch (a.k) {
        case 1: return k4c_launch_k<T, 1, MASKED, GATHER>(ctx, a);
        case 2: return k4c_launch_k<T, 2, MASKED, GATHER>(ctx, a);
        case 3: return k4c_launch_k<T, 3, MASKED, GATHER>(ctx, a);
        case 4: return k4c_launch_k<T, 4, MASKED, GATHER>(ctx, a);
        case 5: return k4c_launch_k<T, 5, MASKED, GATHER>(ctx, a);
        case 6: return k4c_launch_k<T, 6, MASKED, GATHER>(ctx, a);
        case 7: return k4c_launch_k<T, 7, MASKED, GATHER>(ctx, a);
        case 8: return k4c_launch_k<T, 8, MASKED, GATHER>(ctx, a);
        case 9: return k4c_launch_k<T, 9, MASKED, GATHER>(ctx, a);
        case 10: return k4c_launch_k<T, 10, MASKED, GATHER>(ctx, a);
        default: return fail(POLS_ERR_UNSUPPORTED, "rolling (row-parallel): %d features > %d", a.k, K4C_KMAX);
    }
}

}  // namespace pols
