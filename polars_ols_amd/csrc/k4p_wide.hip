// k4p_wide.hip -- K4p / K3p: rolling OLS and RLS for 11..32 features, ONE WAVE PER CHUNK, the INVERSE distributed over the wave's registers.
//
// Replaces solve_rolling_ols (src/least_squares.rs:848-1032) on null-free frames and solve_recursive_least_squares (:568-598, validity
// mask included) + the dynamic make_predictions (src/expressions.rs:184) at the widths where a lane can no longer hold a state of its own
// (K3c / K4c stop at 10 features).  The wave-per-chunk kernels of k4w_wide.hip that used to take these widths handle ONE row at a time
// through dependent global loads, an LDS Cholesky with a wave barrier per elimination step and a second load for the prediction --
// 15-30 k cycles per row, 39-44 ms per 10 M rows at 12 features (profiles/r04_bench_dyn_edges.txt).  Here:
//   * rows arrive 32 at a time: lane l loads row i0 + l of every column (coalesced down the row axis; lanes 32-63 load the rows that
//     LEAVE the window), converted to f64 and parked row-major in LDS, from where every row is read back with broadcast 16-byte reads;
//   * the state is P = (X'X)^-1 (rolling) or the RLS covariance, KP x KP padded (KP = 16 or 32), lane (r, seg) holding the CPL = KP^2 / 64
//     entries P[r][seg CPL ..] in registers.  A row costs O(K^2): z = P x is CPL multiply-adds per lane and a 1-2 step cross-segment
//     sum (`v_permlane16/32_swap`), x'z and x'beta one DPP row all-reduce each, z changes hands through a KP-double LDS slot, and
//     P -= g z z' is CPL more.  RLS is the reference's update literally (:531-540); rolling applies Sherman-Morrison for the row that
//     enters and the row that leaves -- what the reference's own WoodburyState does beyond 60 features (:737-787) -- with beta updated
//     incrementally (beta += g z (y - x'beta));
//   * the information form (X'X, X'y) is kept NEXT to the inverse (rolling: the same lanes, CPL registers), exactly as
//     NonWoodburyState::update accumulates it (:707-725).  While a window is young (the first solves of a sequence, where X'X is
//     nearly singular) every row INVERTS it afresh (symmetric sweep operator, one pivot row broadcast through LDS per step: K steps);
//     the inverse is only propagated once a factorisation's smallest pivot ratio says the window is well conditioned, is rebuilt from
//     the sums every 128 rows and at every chunk start, and is dropped again whenever a downdate's denominator collapses -- so rounding
//     never accumulates beyond 128 rows and an ill-conditioned window is always solved from the sums, like the reference does;
//   * a non-positive pivot (fewer than K independent rows in the window) yields NaN where the reference's LU fallback returns
//     inf / NaN / 1e15-sized numbers (the K4c divergence, include/pols_mi355x.h);
//   * coefficients leave one row per store instruction (K consecutive values), predictions 32 rows at a time.
//   * LPS lanes per sequence.  With LPS = 64 a wave walks one chunk; with many chunks in the frame a wave takes SEVERAL -- KP = 16: four
//     chunks, each on one 16-lane DPP row holding a whole 16-column row of P per lane (LPS = 16: the cross-segment sums go, the row sums
//     and the LDS hand-over serve four sequences per instruction: ~30 instead of ~110 instructions per row); KP = 32 (RLS): two chunks on
//     32 lanes each.  The sub-waves run the same code on their own chunk, staging area and LDS slots; where their control flow differs
//     (lengths, validity, a fresh inversion) the others are masked -- every cross-lane operation stays inside a sub-wave.
// Chunks: a sequence of up to 1 024 rows is ONE chunk (no totals, no scan: the state at a sequence start is the prior / empty);
// longer sequences are cut, an RLS chunk starts from the scanned decayed sums (kp_totals + chunk_scan_launch mode 2, inverted once),
// a rolling chunk re-sums the min(window, rel0) rows in front of it (windows up to 1 024 rows; beyond that only single-chunk
// sequences come here).  Everything else (validity masks under rolling, min_periods > window) stays with k4w_wide.hip.
#include "k4_rolling.hpp"

// No implicit contraction in this file: `v += dpp(v)` behind `v = x * z` was fused into fl(partner's product) + (own product, unrounded),
// which gives the two lanes of a pair sums that differ in the last bit -- x'Px then differs from lane to lane, the rank-1 term loses its
// exact symmetry, and the antisymmetric residue of P grows by 1 / ff per row (half_life 21: 2e14 over 1 000 rows).  Every multiply-add
// that should be fused is an explicit fma().
#pragma clang fp contract(off)

namespace pols {

constexpr int KP_REFRESH = 128;        // rolling: rows between two rebuilds of the inverse from the sums
constexpr double KP_SWITCH_RATIO = 1e-4;   // propagate the inverse only from a factorisation whose smallest pivot / diagonal exceeds this

__device__ __forceinline__ double kp_xor32_add(double v) {     // v(l) + v(l ^ 32)
    const unsigned long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double kp_xor16_add(double v) {     // v(l) + v(l ^ 16)
    const unsigned long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}

// KP: padded width (16 or 32).  LPS: lanes per sequence (a multiple of KP): lane l of a sub-wave holds the CPL = KP KP / LPS entries
// P[r][seg CPL ..] with r = l % KP, seg = l / KP; SEQ = 64 / LPS chunks per wave; RB = LPS / 2 rows staged per block (the lower half of
// the sub-wave loads the rows that enter, the upper half the rows that leave).
template <int KP, int LPS>
struct KpGeo {
    static_assert(LPS % KP == 0 && 64 % LPS == 0, "whole segments, whole sub-waves");
    static constexpr int NSEG = LPS / KP;
    static constexpr int CPL = KP / NSEG;          // matrix entries per lane
    static constexpr int SEQ = 64 / LPS;
    static constexpr int RB = LPS / 2;
    static constexpr int XS = KP + 2;              // staged row: KP features (zero padded), the target, one pad (16-byte rows)
    static constexpr int PER_SUB = 2 * RB * XS + 3 * KP;     // doubles of LDS per sub-wave: two staging areas, z, pivot row, diagonal
};

// sum over the segments (lanes of the sub-wave with the same row index r)
template <int KP, int LPS> __device__ __forceinline__ double kp_segsum(double v) {
    constexpr int NSEG = LPS / KP;
    static_assert(NSEG == 1 || (KP == 16 && NSEG == 4) || (KP == 32 && NSEG == 2), "one segment per lane row, or the whole wave");
    if constexpr (NSEG == 1) return v;
    else {
        asm volatile("" : "+v"(v));
        if constexpr (KP == 16) v = kp_xor16_add(v);
        return kp_xor32_add(v);
    }
}
// DPP move with bound_ctrl (every pattern below is a permutation of the row: no lane reads out of range, and no `v_mov_b32 v, 0` is
// needed in front of the move)
template <int CTRL> __device__ __forceinline__ double kp_dpp(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// sum over the KP row indices of a segment (every lane gets the total, bit for bit the same)
template <int KP> __device__ __forceinline__ double kp_rowsum(double v) {
    asm volatile("" : "+v"(v));                    // (the product is rounded before it is shared: every lane ends with the same bits)
    v += kp_dpp<0xB1>(v);                          // quad_perm [1,0,3,2]
    v += kp_dpp<0x4E>(v);                          // quad_perm [2,3,0,1]
    v += kp_dpp<0x141>(v);                         // row_half_mirror
    v += kp_dpp<0x140>(v);                         // row_mirror
    if constexpr (KP == 32) v = kp_xor16_add(v);
    return v;
}
// 1 / x: v_rcp_f64 and two Newton steps (the division sequence is ~15 instructions; the inputs here are sums of squares, far from the
// denormal / overflow cases its scaling exists for)
__device__ __forceinline__ double kp_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void kp_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum_cc P[cc] x[cc] with up to four independent chains (a 16- or 32-long dependent fma chain is 130-260 cycles of latency per row)
template <int CPL>
__device__ __forceinline__ double kp_dot(const double (&P)[CPL], const double (&x)[CPL]) {
    constexpr int NC = CPL >= 16 ? 4 : (CPL >= 8 ? 2 : 1);
    double acc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] = P[q] * x[q];
#pragma unroll
    for (int cc = NC; cc < CPL; ++cc) acc[cc % NC] = fma(P[cc], x[cc], acc[cc % NC]);
    if constexpr (NC == 4) return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    else if constexpr (NC == 2) return acc[0] + acc[1];
    else return acc[0];
}

// One row of the staging area in registers: the lane's CPL features of its segment, feature r, the target.
template <int KP, int LPS>
struct KpRow {
    double xc[KpGeo<KP, LPS>::CPL], xr, y;
    __device__ __forceinline__ void load(const double *row, int r, int c0) {
        constexpr int CPL = KpGeo<KP, LPS>::CPL;
#pragma unroll
        for (int cc = 0; cc < CPL; cc += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(row + c0 + cc);
            xc[cc] = t.x; xc[cc + 1] = t.y;
        }
        xr = row[r];
        y = row[KP];
    }
};

template <typename T, int KP, int LPS>
struct KpCtx {
    using Geo = KpGeo<KP, LPS>;
    static constexpr int CPL = Geo::CPL, XS = Geo::XS, RB = Geo::RB, SEQ = Geo::SEQ;
    const K4Args &a;
    const int K, lane, sub, r, seg, c0;          // lane: inside the sub-wave
    double *xin, *xout, *zb, *rowbuf, *diag;
    __device__ KpCtx(const K4Args &a_, double *lds)
        : a(a_), K(a_.k), lane(threadIdx.x % LPS), sub(threadIdx.x / LPS), r((threadIdx.x % LPS) % KP), seg((threadIdx.x % LPS) / KP),
          c0(((threadIdx.x % LPS) / KP) * CPL) {
        xin = lds + (size_t)sub * Geo::PER_SUB; xout = xin + RB * XS; zb = xout + RB * XS; rowbuf = zb + KP; diag = rowbuf + KP;
    }
    static constexpr size_t lds_doubles() { return (size_t)SEQ * Geo::PER_SUB; }

    // rows [i_in, i_in + RB) of the sequence starting at absolute row s -> xin (lower half of the sub-wave), rows [i_out, ...) -> xout
    // (upper half); rows outside [lo, hi) are staged as zero rows.  Returns the validity bits of the entering rows.
    __device__ __forceinline__ unsigned stage(int64_t s, int64_t i_in, int64_t i_out, int64_t lo, int64_t hi, bool with_out) const {
        const int t = lane % RB;
        const bool outl = lane >= RB;
        const int64_t i = (outl ? i_out : i_in) + t;
        const bool on = (!outl || with_out) && i >= lo && i < hi;
        double *dst = (outl ? xout : xin) + t * XS;
        bool v = false;
        if (on) {
#pragma unroll 4
            for (int j = 0; j < K; ++j) dst[j] = (double)static_cast<const T *>(a.x[j])[s + i];
            dst[KP] = (double)static_cast<const T *>(a.y)[s + i];
            v = outl ? false : (a.valid ? a.valid[s + i] != 0 : true);
        } else if (!outl || with_out) {
            for (int j = 0; j < K; ++j) dst[j] = 0.0;
            dst[KP] = 0.0;
        }
        const unsigned long long m = __ballot(v);
        kp_sync();
        return (unsigned)((m >> (sub * LPS)) & ((1ull << RB) - 1ull));
    }
    __device__ __forceinline__ void zero_pads() const {             // features K .. KP - 1 of every staged row are zero, once
        for (int q = lane; q < 2 * RB * XS; q += LPS) xin[q] = 0.0;
        kp_sync();
    }
    // every lane gets the CPL entries v[c0 ..] of a vector held one entry per row index (lanes of segment 0 publish)
    __device__ __forceinline__ void bcast(double vr, double (&vc)[CPL]) const {
        if (seg == 0) zb[r] = vr;
        kp_sync();
#pragma unroll
        for (int cc = 0; cc < CPL; cc += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(zb + c0 + cc);
            vc[cc] = t.x; vc[cc + 1] = t.y;
        }
        kp_sync();
    }
    // M (symmetric, this lane's CPL entries of row r) -> its inverse, by K steps of the symmetric sweep operator; the pivot row travels
    // through LDS.  ok: every pivot positive; ratio: the smallest pivot / its diagonal entry (how much of the diagonal survives).
    __device__ __forceinline__ void invert(double (&M)[CPL], bool &ok, double &ratio) const {
        double dg = 0.0;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) dg = (c0 + cc == r) ? M[cc] : dg;
        if (r >= c0 && r < c0 + CPL) diag[r] = dg;
        ok = true; ratio = 1.0;
        for (int j = 0; j < K; ++j) {
            if (r == j) {
#pragma unroll
                for (int cc = 0; cc < CPL; cc += 2) *reinterpret_cast<double2 *>(rowbuf + c0 + cc) = double2{M[cc], M[cc + 1]};
            }
            kp_sync();
            double vc[CPL];
#pragma unroll
            for (int cc = 0; cc < CPL; cc += 2) {
                const double2 t = *reinterpret_cast<const double2 *>(rowbuf + c0 + cc);
                vc[cc] = t.x; vc[cc + 1] = t.y;
            }
            const double vr = rowbuf[r], pj = rowbuf[j], dj = diag[j];
            kp_sync();
            ok = ok && (pj > 0.0);
            const double p = kp_rcp(pj), vrp = vr * p;
            ratio = fmin(ratio, pj * kp_rcp(dj));
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                double val = fma(-(vr * vc[cc]), p, M[cc]);                // (v_r v_c first: the result is exactly symmetric)
                val = (r == j) ? vc[cc] * p : val;
                val = (c0 + cc == j) ? ((r == j) ? -p : vrp) : val;
                M[cc] = val;
            }
        }
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) M[cc] = -M[cc];
        ok = ok && (ratio == ratio);
    }
    // (P b)_r for a vector b held one entry per row index
    __device__ __forceinline__ double matvec(const double (&P)[CPL], double br) const {
        double bc[CPL];
        bcast(br, bc);
        return kp_segsum<KP, LPS>(kp_dot<CPL>(P, bc));
    }
    // coefficient row / prediction of relative row i (beta one entry per row index; `good` false -> NaN)
    __device__ __forceinline__ void store_coef(int64_t row, double beta, bool good) const {
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        if (a.coef && seg == 0 && r < K) static_cast<T *>(a.coef)[row * K + r] = (T)(good ? beta : qnan);
    }
};

// ------------------------------------------------------------------ RLS: per-chunk decayed sums (only for sequences cut into chunks)
template <typename T, int KP, int LPS, bool RLS>
__global__ void __launch_bounds__(64) kp_totals_kernel(const K4Args a) {
    constexpr int CPL = KpGeo<KP, LPS>::CPL, XS = KpGeo<KP, LPS>::XS, KP_RB = KpGeo<KP, LPS>::RB;
    __shared__ __attribute__((aligned(16))) double lds[KpCtx<T, KP, LPS>::lds_doubles()];
    KpCtx<T, KP, LPS> cx(a, lds);
    const int64_t ci = (int64_t)blockIdx.x * KpGeo<KP, LPS>::SEQ + cx.sub;
    if (ci >= a.n_chunks) return;                                  // (a whole sub-wave: the others' cross-lane traffic stays inside them)
    const int64_t c = a.order ? (int64_t)a.order[ci] : ci;
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    if (ch.index_in_group == 0 && ch.t1 >= G.end) return;          // the sequence's only chunk: nobody reads its sums (the walk starts it from the prior / from nothing)
    const int K = cx.K;
    cx.zero_pads();
    double S[CPL], b = 0.0, decay = 1.0;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) S[cc] = 0.0;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    for (int64_t i0 = rel0; i0 < rel1; i0 += KP_RB) {
        const unsigned vm = cx.stage(G.start, i0, 0, rel0, rel1, false);
        const int nb = (int)min((int64_t)KP_RB, rel1 - i0);
        for (int t = 0; t < nb; ++t) {
            if (!((vm >> t) & 1)) continue;
            KpRow<KP, LPS> x;
            x.load(cx.xin + t * XS, cx.r, cx.c0);
            const double ffr = RLS ? a.ff : 1.0;                           // (rolling: plain sums of the chunk's valid rows)
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) S[cc] = fma(x.xr, x.xc[cc], ffr * S[cc]);
            b = fma(x.xr, x.y, ffr * b);
            decay *= ffr;
        }
        kp_sync();
    }
    double *out = a.totals + (size_t)c * a.tot_cs;
    if (cx.r < K) {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc)
            if (cx.c0 + cc < K) out[cx.r * K + cx.c0 + cc] = S[cc];
        if (cx.seg == 0) out[K * K + cx.r] = b;
    }
    if (RLS && cx.lane == 0) out[K * K + K] = decay;
}

// ------------------------------------------------------------------ RLS walk (RecursiveLeastSquares::update, literally)
template <typename T, int KP, int LPS>
__global__ void __launch_bounds__(64) kp_rls_walk_kernel(const K4Args a) {
    constexpr int CPL = KpGeo<KP, LPS>::CPL, XS = KpGeo<KP, LPS>::XS, KP_RB = KpGeo<KP, LPS>::RB;
    __shared__ __attribute__((aligned(16))) double lds[KpCtx<T, KP, LPS>::lds_doubles()];
    KpCtx<T, KP, LPS> cx(a, lds);
    const int64_t ci = (int64_t)blockIdx.x * KpGeo<KP, LPS>::SEQ + cx.sub;
    if (ci >= a.n_chunks) return;
    const int64_t c = a.order ? (int64_t)a.order[ci] : ci;       // (work slots in order of chunk length: see K4Args::order)
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    const int K = cx.K, r = cx.r, c0 = cx.c0;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start;
    cx.zero_pads();
    double P[CPL], beta;
    if (rel0 == 0) {                                               // :505-529: P = p0 I, beta = initial_state_mean
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) P[cc] = (c0 + cc == r && r < K) ? a.p0 : 0.0;
        beta = (r < K && a.mean0) ? a.mean0[r] : 0.0;
    } else {                                                       // the scanned information state entering this chunk (prior included)
        const double *tot = a.totals + (size_t)c * a.tot_cs;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) P[cc] = (r < K && c0 + cc < K) ? tot[r * K + c0 + cc] : 0.0;
        const double br = r < K ? tot[K * K + r] : 0.0;
        bool ok; double ratio;
        cx.invert(P, ok, ratio);
        beta = cx.matvec(P, br);
    }
    const double ff = a.ff, iff = 1.0 / a.ff;
    for (int64_t i0 = rel0; i0 < rel1; i0 += KP_RB) {
        const unsigned vm = cx.stage(G.start, i0, 0, rel0, rel1, false);
        const int nb = (int)min((int64_t)KP_RB, rel1 - i0);
        double predv = 0.0;
        for (int t = 0; t < nb; ++t) {
            KpRow<KP, LPS> x;
            x.load(cx.xin + t * XS, r, c0);
            // (requesting row t + 1 before this row's dependent chain was measured: the extra CPL + 2 registers cost a wave per SIMD -- 1.13 ->
            // 1.25 ms at 12 features, 4.10 -> 4.69 ms at 32)
            double xb = kp_rowsum<KP>(x.xr * beta);                        // x'beta: the prediction of a row that does not update
            if ((vm >> t) & 1) {
                const double z = kp_segsum<KP, LPS>(kp_dot<CPL>(P, x.xc));   // (P x)_r
                const double d = kp_rowsum<KP>(x.xr * z);
                const double rr = fma(d, iff, 1.0);                        // :533
                const double g = kp_rcp(rr * ff);
                const double kk = z * g;                                   // gain_r (:534)
                const double err = x.y - xb;
                beta = fma(kk, err, beta);                                 // :536-537
                xb = fma(d * g, err, xb);                                  // x'beta after the update: x'k = x'P x / (r ff)
                // P / ff - k k' r (:538-539).  The product k_r k_c is formed FIRST: it is bitwise the same for (r, c) and (c, r), so P stays
                // exactly symmetric like the reference's (an antisymmetric rounding residue grows by 1 / ff per row: 2e14 over 1 000 rows
                // at half_life 21)
                double kc[CPL];
                cx.bcast(kk, kc);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) P[cc] = fma(-(kk * kc[cc]), rr, P[cc] * iff);
            }
            cx.store_coef(G.start + i0 + t, beta, true);
            predv = (cx.lane == t) ? xb : predv;
        }
        if (a.pred && cx.lane < nb) static_cast<T *>(a.pred)[G.start + i0 + cx.lane] = (T)predv;
        kp_sync();
    }
}

// ------------------------------------------------------------------ rolling walk (min_periods <= window)
// MASKED = false: null-free frames (the "drop" deque and the fixed window are the same sums).
// MASKED = true : the FIXED window over rows with validity bytes ("drop_window", ls.rs:987-1029): a row enters only if it is valid, row
//   i - window leaves only if it is valid, at or beyond j_min = max(mpv - window, 0) (rows older than that are never dropped: the
//   sliding loop only starts at i = mpv, :989-990) and i >= mpv; a row is SOLVED when it is the warm-up row mpv - 1 or its window holds
//   at least gate_n valid rows (the n_valid_window gate, :1013, :1022) and otherwise repeats the last solved row's coefficients.  The
//   validity prefix (K4Args::cnt) and the per-group constants come from the device tables (dyn_prep.hip valid_tables_launch).
struct KpMasks { unsigned vin, vout, gate; };

template <typename T, int KP, int LPS, bool MASKED>
__global__ void __launch_bounds__(64) kp_rolling_walk_kernel(const K4Args a) {
    constexpr int CPL = KpGeo<KP, LPS>::CPL, XS = KpGeo<KP, LPS>::XS, KP_RB = KpGeo<KP, LPS>::RB;
    __shared__ __attribute__((aligned(16))) double lds[KpCtx<T, KP, LPS>::lds_doubles()];
    KpCtx<T, KP, LPS> cx(a, lds);
    const int64_t ci = (int64_t)blockIdx.x * KpGeo<KP, LPS>::SEQ + cx.sub;
    if (ci >= a.n_chunks) return;
    const int64_t c = a.order ? (int64_t)a.order[ci] : ci;       // (work slots in order of chunk length: see K4Args::order)
    const K4Chunk ch = a.chunks[c];
    const K4Group G = a.groups[ch.group];
    const int K = cx.K, r = cx.r, c0 = cx.c0;
    const int64_t rel0 = ch.t0 - G.start, rel1 = ch.t1 - G.start, n = G.end - G.start;
    const int64_t w = a.window, mpv = G.mpv;
    const int64_t j_min = MASKED ? max(mpv - w, (int64_t)0) : 0;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    if (G.all_nan) {                                               // :893-900
        for (int64_t i = rel0; i < rel1; ++i) {
            cx.store_coef(G.start + i, 0.0, false);
            if (a.pred && cx.lane == 0) static_cast<T *>(a.pred)[G.start + i] = (T)qnan;
        }
        return;
    }
    cx.zero_pads();
    double S[CPL], P[CPL], bsum = 0.0, beta = 0.0;
    const uint8_t *valid = a.valid ? a.valid + G.start : nullptr;
    const int32_t *cnt = a.cnt ? a.cnt + G.start : nullptr;
    // what the rows of the block starting at i0 do: which enter, which rows i - window leave, which are solved
    auto masks = [&](int64_t i0, int64_t hi) __attribute__((always_inline)) -> KpMasks {
        const int t = cx.lane % KP_RB;
        const int64_t i = i0 + t;
        bool vin = false, vout = false, gate = false;
        if (i < hi) {
            if (cx.lane < KP_RB) {
                vin = MASKED ? valid[i] != 0 : true;
                if (i >= mpv - 1) {
                    if constexpr (MASKED) {
                        const int64_t is = i >= w ? i - w : 0;          // saturating_sub (:990)
                        gate = (i == mpv - 1) || ((int64_t)cnt[i] - (int64_t)cnt[is] >= G.gate_n);
                    } else gate = true;
                }
            } else {
                const int64_t o = i - w;
                vout = o >= j_min && o >= 0 && i >= mpv && (MASKED ? valid[o] != 0 : true);
            }
        }
        const unsigned long long bi = __ballot(vin), bo = __ballot(vout), bg = __ballot(gate);
        const unsigned long long m = (1ull << KP_RB) - 1ull;
        const int sh = cx.sub * LPS;
        return KpMasks{(unsigned)((bi >> sh) & m), (unsigned)((bo >> (sh + KP_RB)) & m), (unsigned)((bg >> sh) & m)};
    };
    // S, bsum += sign * the valid rows of [lo, hi)
    auto accumulate = [&](int64_t lo, int64_t hi, double sign) __attribute__((always_inline)) {
        for (int64_t j0 = lo; j0 < hi; j0 += KP_RB) {
            cx.stage(G.start, j0, 0, 0, hi, false);
            const int nb = (int)min((int64_t)KP_RB, hi - j0);
            unsigned vm = 0xffffffffu;
            if constexpr (MASKED) {
                const int64_t j = j0 + (cx.lane % KP_RB);
                vm = (unsigned)((__ballot(cx.lane < KP_RB && j < hi && valid[j] != 0) >> (cx.sub * LPS)) & ((1ull << KP_RB) - 1ull));
            }
            for (int t = 0; t < nb; ++t) {
                if (!((vm >> t) & 1)) continue;
                KpRow<KP, LPS> x;
                x.load(cx.xin + t * XS, r, c0);
                const double sx = sign * x.xr;
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) S[cc] = fma(sx, x.xc[cc], S[cc]);
                bsum = fma(sx, x.y, bsum);
            }
            kp_sync();
        }
    };
    // ... the same from the scanned per-chunk totals when the range is long (cut sequences with a window beyond 1 024 rows, expanding
    // windows included): rows [0, p) = the exclusive prefix at p's chunk + the rows of that chunk in front of p
    auto prefix = [&](int64_t p, double sign) __attribute__((always_inline)) {
        const int64_t ci = p / a.chunk_len;
        const double *tot = a.totals + (size_t)(G.first_chunk + ci) * a.tot_cs;
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) S[cc] += (r < K && c0 + cc < K) ? sign * tot[r * K + c0 + cc] : 0.0;
        bsum += r < K ? sign * tot[K * K + r] : 0.0;
        accumulate(ci * a.chunk_len, p, sign);
    };
    auto add_range = [&](int64_t lo, int64_t hi) __attribute__((always_inline)) {
        if (hi <= lo) return;
        if (!a.use_totals || hi - lo <= 2 * (int64_t)a.chunk_len) accumulate(lo, hi, 1.0);
        else { prefix(hi, 1.0); prefix(lo, -1.0); }
    };
    // the sums after row ip (>= 0): the valid rows the sliding loop has not dropped yet (+ alpha I once the warm-up row has passed, :924-926)
    auto state_after = [&](int64_t ip) __attribute__((always_inline)) {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) S[cc] = 0.0;
        bsum = 0.0;
        if (ip < mpv) add_range(0, ip + 1);                        // (nothing is subtracted before the sliding loop starts)
        else {
            const int64_t lo = max(ip - w + 1, (int64_t)0);
            if (j_min > 0) add_range(0, min(j_min, lo));           // rows older than j_min are never dropped
            add_range(lo, ip + 1);
        }
        if (ip >= mpv - 1 && a.alpha != 0.0) {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) S[cc] += (c0 + cc == r && r < K) ? a.alpha : 0.0;
        }
    };
    // the sums inverted afresh INTO P (whatever P held is dead: either nothing was propagated, or it is being rebuilt) -- no second
    // CPL-register copy next to S and P
    auto solve_fresh = [&](bool &ok, double &ratio) __attribute__((always_inline)) -> double {
#pragma unroll
        for (int cc = 0; cc < CPL; ++cc) P[cc] = S[cc];
        cx.invert(P, ok, ratio);
        return cx.matvec(P, bsum);
    };
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) { S[cc] = 0.0; P[cc] = 0.0; }
    double last = 0.0;                                             // MASKED: the coefficients of the last solved row (forward fill)
    bool last_good = false;
    if (rel0 > 0) {
        if constexpr (MASKED) {
            if (rel0 - 1 >= mpv - 1) {
                // the last row at or before rel0 - 1 that was solved: its coefficients are what a closed gate repeats
                int64_t ip = rel0 - 1;
                for (; ip > mpv - 1; --ip) {
                    const int64_t is = ip >= w ? ip - w : 0;
                    if ((int64_t)cnt[ip] - (int64_t)cnt[is] >= G.gate_n) break;
                }
                state_after(ip);
                bool ok; double ratio;
                last = solve_fresh(ok, ratio);
                last_good = ok;
                if (ip != rel0 - 1) state_after(rel0 - 1);
            } else state_after(rel0 - 1);
        } else state_after(rel0 - 1);
    }
    bool inverted = false;
    int since = 0;
    // A window that can never hold 2 K rows is re-inverted on every row (it is never far from singular)
    const bool may_propagate = w >= 2 * (int64_t)K;
    for (int64_t i0 = rel0; i0 < rel1; i0 += KP_RB) {
        cx.stage(G.start, i0, i0 - w, 0, n, i0 + KP_RB > w);
        const KpMasks mk = masks(i0, rel1);
        const int nb = (int)min((int64_t)KP_RB, rel1 - i0);
        double predv = 0.0;
        for (int t = 0; t < nb; ++t) {
            const int64_t i = i0 + t;
            KpRow<KP, LPS> x;
            x.load(cx.xin + t * XS, r, c0);
            // ---- the row enters (NonWoodburyState::update, :707-725)
            if ((mk.vin >> t) & 1) {
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) S[cc] = fma(x.xr, x.xc[cc], S[cc]);
                bsum = fma(x.xr, x.y, bsum);
                if (inverted) {
                    const double z = kp_segsum<KP, LPS>(kp_dot<CPL>(P, x.xc));
                    const double d = kp_rowsum<KP>(x.xr * z), xb = kp_rowsum<KP>(x.xr * beta);
                    const double g = kp_rcp(1.0 + d);
                    beta = fma(g * z, x.y - xb, beta);
                    double zc[CPL];
                    cx.bcast(z, zc);
#pragma unroll
                    for (int cc = 0; cc < CPL; ++cc) P[cc] = fma(-(z * zc[cc]), g, P[cc]);   // (z_r z_c first: P stays exactly symmetric)
                }
            }
            // ---- row i - window leaves
            if ((mk.vout >> t) & 1) {
                KpRow<KP, LPS> o;
                o.load(cx.xout + t * XS, r, c0);
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) S[cc] = fma(-o.xr, o.xc[cc], S[cc]);
                bsum = fma(-o.xr, o.y, bsum);
                if (inverted) {
                    const double z = kp_segsum<KP, LPS>(kp_dot<CPL>(P, o.xc));
                    const double d = kp_rowsum<KP>(o.xr * z), xb = kp_rowsum<KP>(o.xr * beta);
                    const double den = 1.0 - d;
                    if (!(den > 1e-6)) inverted = false;                   // the downdate collapses: back to the sums (wave-uniform)
                    else {
                        const double g = kp_rcp(den);
                        beta = fma(-g * z, o.y - xb, beta);
                        double zc[CPL];
                        cx.bcast(z, zc);
#pragma unroll
                        for (int cc = 0; cc < CPL; ++cc) P[cc] = fma(z * zc[cc], g, P[cc]);
                    }
                }
            }
            if (i == mpv - 1 && a.alpha != 0.0) {                          // alpha enters once, at the warm-up (:924-926)
#pragma unroll
                for (int cc = 0; cc < CPL; ++cc) S[cc] += (c0 + cc == r && r < K) ? a.alpha : 0.0;
                inverted = false;
            }
            bool good = false;
            double bout = 0.0;
            if ((mk.gate >> t) & 1) {
                ++since;
                if (!inverted || since >= KP_REFRESH) {
                    bool ok; double ratio;
                    const double bnew = solve_fresh(ok, ratio);
                    good = ok; bout = bnew;
                    inverted = ok && may_propagate && ratio > KP_SWITCH_RATIO;
                    if (inverted) { beta = bnew; since = 0; }
                } else {
                    good = true; bout = beta;
                }
                if constexpr (MASKED) { last = bout; last_good = good; }
            } else if (MASKED && i >= mpv - 1) {                           // the gate is closed: the last solved row's coefficients
                good = last_good; bout = last;
            }
            cx.store_coef(G.start + i, bout, good);
            // a row the reference solves whose window sums could not be inverted: the reference runs LU there (ls.rs:732-734) -- kp_lu_fix_kernel does,
            // behind this kernel, for the rows put on the list here (rare; on MASKED frames it also rewrites the rows behind that repeat the row)
            if (((mk.gate >> t) & 1) && !good && a.fix_rows && cx.lane == 0) {
                const int idx = atomicAdd(a.fix_count, 1);
                if (idx < a.fix_cap) a.fix_rows[idx] = G.start + i;
            }
            const double p = kp_rowsum<KP>(x.xr * bout);
            predv = (cx.lane == t) ? (good ? p : qnan) : predv;
        }
        if (a.pred && cx.lane < nb) static_cast<T *>(a.pred)[G.start + i0 + cx.lane] = (T)predv;
        kp_sync();
    }
}

// The rows of the list: one wave each re-sums the row's window -- rows (i - window, i] of its sequence -- into [X'X + alpha I | X'y] in LDS (a lane
// owns the entries e = lane, lane + 64, ...; the rows are staged one at a time) and eliminates with partial pivoting, a lane per column, like the
// reference's fallback (faer's partial_piv_lu: row interchanges on the largest magnitude of the column, the solve behind ls.rs:277-337).  Any width up to 32.
// MASKED ("drop_window" with validity bytes, ls.rs:987-1029): the window holds the VALID rows the walk's masks let in and out -- every valid row up
// to i, less the valid rows of [max(mpv - window, 0), i - window] once i >= mpv (the rows in front of that are never subtracted, :989) -- and the
// rows behind i whose gate is closed repeat its coefficients: they are rewritten too.
template <typename T, bool MASKED>
__global__ void __launch_bounds__(64) kp_lu_fix_kernel(const K4Args a) {
    __shared__ double As[32][33], xs[33], bs[32];
    const int lane = threadIdx.x, K = a.k, K1 = K + 1;
    const int n_fix = min(*a.fix_count, (int)a.fix_cap);
    if (blockIdx.x == 0 && lane == 0) *a.fix_next = 0;              // the next call's counter
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int e = blockIdx.x; e < n_fix; e += gridDim.x) {
        const int64_t i = a.fix_rows[e];
        int lo_g = 0, hi_g = a.n_groups;                            // the sequence holding row i: the last g with start <= i
        while (hi_g - lo_g > 1) {
            const int mid = (lo_g + hi_g) >> 1;
            if (a.groups[mid].start <= i) lo_g = mid; else hi_g = mid;
        }
        const K4Group G = a.groups[lo_g];
        const int64_t s = G.start, w = a.window;
        const int64_t rel = i - s;
        // the rows of the window: [lo, i], and on MASKED frames [0, front) in front of it as well (valid rows only)
        int64_t lo = rel - w + 1 < 0 ? 0 : rel - w + 1, front = 0;
        if constexpr (MASKED) {
            const int64_t j_min = G.mpv - w > 0 ? G.mpv - w : 0;
            if (rel < G.mpv) lo = 0;                                // nothing has left yet
            else { if (lo < j_min) lo = j_min; front = j_min < lo ? j_min : lo; }
        }
        for (int q = lane; q < K * K1; q += 64) As[q / K1][q % K1] = 0.0;
        kp_sync();
        for (int part = 0; part < 2; ++part) {
            const int64_t j0 = part == 0 ? 0 : lo, j1 = part == 0 ? front : rel + 1;
            for (int64_t j = j0; j < j1; ++j) {
                if constexpr (MASKED) { if (!a.valid[s + j]) continue; }   // (wave-uniform)
                if (lane < K) xs[lane] = (double)static_cast<const T *>(a.x[lane])[s + j];
                if (lane == K) xs[K] = (double)static_cast<const T *>(a.y)[s + j];
                kp_sync();
                for (int q = lane; q < K * K1; q += 64) { const int p = q / K1, c = q % K1; As[p][c] = fma(xs[p], xs[c], As[p][c]); }
                kp_sync();
            }
        }
        if (lane < K) As[lane][lane] += a.alpha;
        kp_sync();
        for (int j = 0; j < K; ++j) {
            int pv = j;
            double best = fabs(As[j][j]);
            for (int r2 = j + 1; r2 < K; ++r2) { const double v = fabs(As[r2][j]); if (v > best) { best = v; pv = r2; } }
            kp_sync();
            if (lane <= K && pv != j) { const double t0 = As[j][lane]; As[j][lane] = As[pv][lane]; As[pv][lane] = t0; }
            kp_sync();
            const double d = As[j][j];
            if (lane <= K && lane > j) {
                const double top = As[j][lane];
                for (int r2 = j + 1; r2 < K; ++r2) As[r2][lane] -= (As[r2][j] / d) * top;      // (column j itself is not written in this step)
            }
            kp_sync();
        }
        if (lane == 0) {
            double beta[32];
            for (int p = K - 1; p >= 0; --p) {
                double sacc = As[p][K];
                for (int q = p + 1; q < K; ++q) sacc -= As[p][q] * beta[q];
                beta[p] = sacc / As[p][p];
            }
            for (int p = 0; p < K; ++p) bs[p] = beta[p];
        }
        kp_sync();
        // row i, and on MASKED frames the rows behind it that repeat it (gate closed: the walk's masks)
        for (int64_t f = rel; f < G.end - s; ++f) {
            if (f > rel) {
                if constexpr (!MASKED) break;
                const int64_t is = f >= w ? f - w : 0;
                if (f == G.mpv - 1 || (int64_t)a.cnt[s + f] - (int64_t)a.cnt[s + is] >= G.gate_n) break;   // solved on its own
            }
            if (lane == 0) {
                double pr = 0.0;
                for (int p = 0; p < K; ++p) {
                    if (a.coef) static_cast<T *>(a.coef)[(s + f) * K + p] = (T)bs[p];
                    pr = fma((double)static_cast<const T *>(a.x[p])[s + f], bs[p], pr);
                }
                bool vf = true;
                if constexpr (MASKED) vf = a.valid[s + f] != 0;
                if (a.pred) static_cast<T *>(a.pred)[s + f] = (T)(vf ? pr : qnan);
            }
        }
        kp_sync();
    }
}

template <typename T, int KP, int LPS>
static int kp_launch_kp(pols_ctx *ctx, const K4Args &a, bool rls, bool single_chunk) {
    const unsigned blocks = (unsigned)((a.n_chunks + KpGeo<KP, LPS>::SEQ - 1) / KpGeo<KP, LPS>::SEQ);
    timing_begin(ctx);
    if (rls) {
        if (!single_chunk) {
            hipLaunchKernelGGL((kp_totals_kernel<T, KP, LPS, true>), dim3(blocks), dim3(64), 0, ctx->stream, a);
            chunk_scan_launch(ctx, a, a.k * a.k + a.k, 2);
        }
        hipLaunchKernelGGL((kp_rls_walk_kernel<T, KP, LPS>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    } else {
        if (a.use_totals) {                                        // plain per-chunk sums + their exclusive prefix (mode 0)
            hipLaunchKernelGGL((kp_totals_kernel<T, KP, LPS, false>), dim3(blocks), dim3(64), 0, ctx->stream, a);
            chunk_scan_launch(ctx, a, a.k * a.k + a.k, 0);
        }
        if constexpr (KpGeo<KP, LPS>::CPL <= 16) {
            if (a.valid) {
                hipLaunchKernelGGL((kp_rolling_walk_kernel<T, KP, LPS, true>), dim3(blocks), dim3(64), 0, ctx->stream, a);
                if (a.fix_rows) hipLaunchKernelGGL((kp_lu_fix_kernel<T, true>), dim3(128), dim3(64), 0, ctx->stream, a);
            } else {
                hipLaunchKernelGGL((kp_rolling_walk_kernel<T, KP, LPS, false>), dim3(blocks), dim3(64), 0, ctx->stream, a);
                if (a.fix_rows) hipLaunchKernelGGL((kp_lu_fix_kernel<T, false>), dim3(128), dim3(64), 0, ctx->stream, a);
            }
        }
    }
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
static int kp_launch_t(pols_ctx *ctx, const K4Args &a, bool rls, bool single_chunk) {
    // Lanes per sequence.  Few chunks: a whole wave each (the most parallelism per chunk).  From 4 096 chunks on (a wave per SIMD even
    // after packing) up to 16 features: four chunks per wave, each on its own 16-lane row; 17..32 features, RLS: two per wave from 16 384
    // chunks (rolling would need a 32-entry row of the sums AND of the inverse per lane: over the register file).  POLS_K4P_LPS forces one.
    int lps = 64;
    if (a.k <= 16) lps = a.n_chunks >= 4096 ? 16 : 64;
    else lps = (rls && a.n_chunks >= 16384) ? 32 : 64;
    const int want = ctx->opt.k4p_lps;
    if (want == 64) lps = 64;
    else if (want == 16 && a.k <= 16) lps = 16;
    else if (want == 32 && a.k > 16 && rls) lps = 32;
    ctx->last_kernel += lps == 64 ? "" : (lps == 16 ? "_x4" : "_x2");
    if (a.k <= 16) return lps == 16 ? kp_launch_kp<T, 16, 16>(ctx, a, rls, single_chunk) : kp_launch_kp<T, 16, 64>(ctx, a, rls, single_chunk);
    return lps == 32 ? kp_launch_kp<T, 32, 32>(ctx, a, rls, single_chunk) : kp_launch_kp<T, 32, 64>(ctx, a, rls, single_chunk);
}

// single_chunk: no sequence was cut (every chunk starts its sequence).  Rolling: min_periods <= window, window <= 1 024 unless
// single_chunk, and either a null-free frame (a.valid == nullptr) or the fixed window over rows with validity bytes + the device
// validity tables (a.valid, a.cnt, patched groups; "drop_window") -- the caller checks.
int k4p_launch(pols_ctx *ctx, int dtype, const K4Args &a, bool rls, bool single_chunk) {
    if (a.k > 32 || a.k < 1) return fail(POLS_ERR_UNSUPPORTED, "k4p: %d features", a.k);
    if (a.n_chunks <= 0) return POLS_OK;
    ctx->last_kernel = std::string(rls ? "k3p_rls_inverse_wave" : "k4p_rolling_inverse_wave") + (dtype == POLS_F32 ? "_f32" : "_f64");
    K4Args aa = a;
    if (!rls && !ctx->opt.debug_skip_fixup) {          // (POLS_DEBUG_SKIP_FIXUP: the walk's own answer, NaN on such rows -- the test's A/B)
        // the list of rows whose sums could not be inverted (scratch slot 27, shared with K4c: [two counters that take turns][rows])
        const int64_t n_rows = a.groups_end_row;
        void *fx = nullptr;
        const int64_t cap = std::min<int64_t>(std::max<int64_t>(n_rows, 1), (int64_t)1 << 22);
        int rc = ensure_scratch(ctx, 27, 256 + sizeof(int64_t) * (size_t)cap, &fx);
        if (rc) return rc;
        if (ctx->k4c_fix_ptr != fx) { POLS_HIP(hipMemsetAsync(fx, 0, 256, ctx->stream)); ctx->k4c_fix_ptr = fx; ctx->k4c_fix_turn = 0; }
        aa.fix_count = static_cast<int32_t *>(fx) + 32 * (ctx->k4c_fix_turn & 1);
        aa.fix_next = static_cast<int32_t *>(fx) + 32 * ((ctx->k4c_fix_turn + 1) & 1);
        ++ctx->k4c_fix_turn;
        aa.fix_rows = reinterpret_cast<int64_t *>(static_cast<char *>(fx) + 256);
        aa.fix_cap = cap;
    }
    return dtype == POLS_F32 ? kp_launch_t<float>(ctx, aa, rls, single_chunk) : kp_launch_t<double>(ctx, aa, rls, single_chunk);
}

}  // namespace pols
