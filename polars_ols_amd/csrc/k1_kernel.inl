// k1_kernel.inl -- body of K1 (see k1_gram_chol.hpp for the design notes).  Included by k1_{f32,f64}_*.hip and k1n_{f32,f64}_*.hip, a few column counts per unit.
#include "k1_gram_chol.hpp"

#ifndef K1_RAGGED_VECTOR_LOADS
#define K1_RAGGED_VECTOR_LOADS 1
#endif
#ifndef K1T_F32_TWO_PASS
#define K1T_F32_TWO_PASS 1
#endif
#ifndef K1_EDGE_SHIFT
#define K1_EDGE_SHIFT 1        // load_chunk_edge: the chunk crossing the end of the columns is shifted down in registers (0: guarded scalar loads)
#endif
#ifndef K1_EDGE_MASK32
#define K1_EDGE_MASK32 1       // load_chunk_edge: in-group masks from 32-bit distances (0: 64-bit row compares)
#endif
#ifndef K1_KU_STATIC
#define K1_KU_STATIC 1         // K1t: only the last column slot can be the ones column, the k_user test is compile-time for the others
#endif
#ifndef K1T_PASS_MIN_KT
#define K1T_PASS_MIN_KT 5      // K1t: from this many columns the Gram goes through the 16-entry passes + LDS + the row-cooperative Cholesky
#endif
#ifndef K1_SOLVE_ROWS
#define K1_SOLVE_ROWS 1        // multi-pass team kernels: the row-resident right-looking Cholesky at every width (0: LDS left-looking up to 15 columns)
#endif

namespace pols {

// one lane's value to every lane (wave-uniform result)
__device__ __forceinline__ float k1p_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double k1p_readlane(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int NZ>
__host__ __device__ constexpr int tri_index(int i, int j) {  // packed upper triangle, i <= j < NZ
    return i * NZ - (i * (i - 1)) / 2 + (j - i);
}

// One chunk = VEC consecutive rows of every column, held by one lane as the 16-byte vectors it loaded
// (scaled in place by sqrt(w) when there are weights -- no second copy: register pressure decides how many
// groups are in flight per CU, and that is what this latency-bound kernel's throughput is made of).
template <typename T, int KT, bool HAS_W>
struct Chunk {
    using V = typename Vec16<T>::type;
    V x[KT];   // features (sqrt(w)-scaled if HAS_W); slot KT-1 holds the intercept column when there is one
    V y;       // ORIGINAL target (needed for residuals)
    V sw;      // sqrt(w); only meaningful when HAS_W
    unsigned m;  // NULLS kernels: bit v set = row v of the chunk takes part in the fit
};

template <typename T> __device__ __forceinline__ typename Vec16<T>::type vsplat(T v);
template <> __device__ __forceinline__ float4 vsplat<float>(float v) { return float4{v, v, v, v}; }
template <> __device__ __forceinline__ double2 vsplat<double>(double v) { return double2{v, v}; }

template <typename T> __device__ __forceinline__ void vset(typename Vec16<T>::type &v, int i, T x);
template <> __device__ __forceinline__ void vset<float>(float4 &v, int i, float x) {
    if (i == 0) v.x = x; else if (i == 1) v.y = x; else if (i == 2) v.z = x; else v.w = x;
}
template <> __device__ __forceinline__ void vset<double>(double2 &v, int i, double x) { if (i == 0) v.x = x; else v.y = x; }

// the 16-byte loads of one chunk, nothing else: the FAST kernels issue these for ALL resident chunks back to back
// KUS: only the LAST column slot can be the synthesised intercept, so the k_user test is compile-time for the others.  K1t only: there
// it removes four v_mov of 1.0 + a scalar branch per column; in the wave / team kernels the loads it un-serialises cost 6-10 VGPRs
// and an occupancy step (team64_rc1_edge_p2 at 8 columns 77 -> 83: 500 000 groups of 130..252 rows 692 -> 735 us).
template <typename T, int KT, bool HAS_W, bool NT = false, bool KUS = false>
__device__ __forceinline__ void load_chunk_raw(const K1Args &a, int64_t row0, Chunk<T, KT, HAS_W> &c) {
    using V = typename Vec16<T>::type;
    const int ku = a.k_user;                                 // KT - add_intercept: only the LAST slot can be the ones column
    // NT: streaming (`nt`) loads, a compile-time choice -- a run-time branch around the two forms cost the plain path 8 us of 73.
    // (Written out twice on purpose: routing the loads through a small lambda kept the chunk in scratch memory in the ragged
    // kernels -- 736 bytes per lane, 4x slower.)
    if constexpr (NT) {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if ((K1_KU_STATIC && KUS && j < KT - 1) || j < ku) c.x[j] = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + row0));
            else c.x[j] = vsplat<T>(T(1));
        }
        c.y = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0));
        if constexpr (HAS_W) c.sw = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0));
    } else {
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if ((K1_KU_STATIC && KUS && j < KT - 1) || j < ku) c.x[j] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + row0);
            else c.x[j] = vsplat<T>(T(1));
        }
        c.y = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0);
        if constexpr (HAS_W) c.sw = *reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0);
    }
}

template <typename T, int KT, bool HAS_W, bool FAST, bool NULLS = false, bool LOADED = false, bool VEDGE = false>
__device__ __forceinline__ void load_chunk(const K1Args &a, int64_t row0, int64_t s, int64_t e, Chunk<T, KT, HAS_W> &c) {
    constexpr int VEC = Vec16<T>::N;
    const int ku = a.k_user;
    if (LOADED) {
        // the vectors are already in c (load_chunk_raw): only the null policy and the sqrt(w) scaling below are left
    } else if (FAST || (row0 >= s && row0 + VEC <= e)) {
        // whole chunk inside the group: 16-byte loads, all issued before first use
        load_chunk_raw<T, KT, HAS_W>(a, row0, c);
    } else if (K1_RAGGED_VECTOR_LOADS && sizeof(T) == 4 && (VEDGE || e - s <= 512) && row0 + VEC <= a.n_rows) {
        // ragged head / tail of a SHORT f32 group, the 16 bytes of every column still inside the columns: vector loads, then the rows
        // outside [s, e) -- a neighbour's -- zeroed in registers.  Measured per shape (scripts/bench_ragged.py): 12..40 rows 218 ->
        // 203 us, 100..300 rows 111 -> 108 us; but 900..1 020 rows 76 -> 79 us and f64 40..120 rows 828 -> 906 us, hence the limits.
        load_chunk_raw<T, KT, HAS_W>(a, row0, c);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const bool in = (row0 + v >= s) && (row0 + v < e);
#pragma unroll
            for (int j = 0; j < KT; ++j) vset<T>(c.x[j], v, in ? vget<T>(c.x[j], v) : T(0));
            vset<T>(c.y, v, in ? vget<T>(c.y, v) : T(0));
            if constexpr (HAS_W) vset<T>(c.sw, v, in ? vget<T>(c.sw, v) : T(1));
        }
    } else {
        // ragged head / tail of a group: guarded scalar loads, rows outside [s, e) contribute zeros
        // (long groups and f64: full 16-byte loads with the neighbouring group's rows zeroed in registers measured 3-9 % SLOWER)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t r = row0 + v;
            const bool in = (r >= s) && (r < e);
            vset<T>(c.y, v, in ? static_cast<const T *>(a.y)[r] : T(0));
            if constexpr (HAS_W) vset<T>(c.sw, v, in ? static_cast<const T *>(a.w)[r] : T(1));
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                T xv = T(0);
                if (in) xv = (j < ku) ? static_cast<const T *>(a.x[j])[r] : T(1);
                vset<T>(c.x[j], v, xv);
            }
        }
    }
    if constexpr (NULLS) {
        // Null policy (src/expressions.rs:201-296) on the registers: which rows take part in the fit (compute_is_valid_mask),
        // and nulls -> 0 in the features (handle_nulls for the fit; construct_features_array(.., true) for the predictions).
        // The target keeps its nulls: residuals are ORIGINAL target - predictions (ls.py:239).
        const int pol = a.null_policy;
        c.m = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t r = row0 + v;
            bool fit = (r >= s) && (r < e);
            if (fit && pol != POLS_NULL_ZERO) {
                const T yv = vget<T>(c.y, v);
                fit = (yv == yv) && !(a.valid && !a.valid[r]);
                if (null_checks_x(pol)) {
#pragma unroll
                    for (int j = 0; j < KT; ++j) { const T xv = vget<T>(c.x[j], v); fit = fit && (xv == xv); }
                }
            }
            c.m |= fit ? (1u << v) : 0u;
#pragma unroll
            for (int j = 0; j < KT; ++j) vset<T>(c.x[j], v, null_fill<T>(pol, vget<T>(c.x[j], v)));
        }
    }
    if constexpr (HAS_W) {   // sqrt_w scaling of every feature, intercept included (least_squares.py:190-196)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T sw = sqrt(vget<T>(c.sw, v));
            vset<T>(c.sw, v, sw);
#pragma unroll
            for (int j = 0; j < KT; ++j) vset<T>(c.x[j], v, vget<T>(c.x[j], v) * sw);
        }
    }
}

// MODE 0: plain (the y'y slot holds y'y).  MODE 1: null policy, masked -- rows outside the fit contribute nothing, a null target
// that stays (policy ZERO) counts as 0, the (unused) y'y slot counts the rows left in the fit.  MODE 2: null policy, but the caller
// has established that every row of the chunk is in the fit with a non-null target: plain products, same row count.
template <typename T, int KT, bool HAS_W, int MODE>
__device__ __forceinline__ void gram_rows(T (&acc)[(KT + 1) * (KT + 2) / 2], const Chunk<T, KT, HAS_W> &c) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T ys = vget<T>(c.y, v);
        if constexpr (HAS_W) ys *= vget<T>(c.sw, v);
        T mv = T(1);
        if constexpr (MODE == 1) {
            mv = ((c.m >> v) & 1u) ? T(1) : T(0);
            ys = (ys == ys) ? ys * mv : T(0);
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const T xi = MODE == 1 ? vget<T>(c.x[i], v) * mv : vget<T>(c.x[i], v);
#pragma unroll
            for (int j = i; j < KT; ++j) acc[tri_index<NZ>(i, j)] = fma(xi, vget<T>(c.x[j], v), acc[tri_index<NZ>(i, j)]);
            acc[tri_index<NZ>(i, KT)] = fma(xi, ys, acc[tri_index<NZ>(i, KT)]);
        }
        if constexpr (MODE == 0) acc[tri_index<NZ>(KT, KT)] = fma(ys, ys, acc[tri_index<NZ>(KT, KT)]);
        else acc[tri_index<NZ>(KT, KT)] += mv;
    }
}

template <typename T, int KT, bool HAS_W, bool NULLS = false>
__device__ __forceinline__ void gram_accumulate(T (&acc)[(KT + 1) * (KT + 2) / 2], const Chunk<T, KT, HAS_W> &c) {
    if constexpr (!NULLS) {
        gram_rows<T, KT, HAS_W, 0>(acc, c);
    } else {
        // Frames are mostly null-free even when a drop policy is asked for: when no lane of the wave holds a dropped row or a null
        // target in this chunk (one ballot) the masking multiplies are skipped.
        constexpr int VEC = Vec16<T>::N;
        bool clean = c.m == ((1u << VEC) - 1u);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { const T yv = vget<T>(c.y, v); clean = clean && (yv == yv); }
        if (__all(clean)) gram_rows<T, KT, HAS_W, 2>(acc, c);
        else gram_rows<T, KT, HAS_W, 1>(acc, c);
    }
}

// The packed entries [Q0, Q1) only: the multi-pass Gram of the f64 team kernel keeps a third of the accumulators live at a
// time (the guards are compile-time constants under the full unroll).
// NULLS: masked like gram_rows MODE 1 -- rows outside the fit contribute nothing, a null target that stays counts as 0, the
// (unused) y'y slot counts the rows left in the fit.
template <typename T, int KT, bool HAS_W, int Q0, int Q1, bool NULLS = false>
__device__ __forceinline__ void gram_accumulate_range(T (&acc)[Q1 - Q0], const Chunk<T, KT, HAS_W> &c) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T ys = vget<T>(c.y, v);
        if constexpr (HAS_W) ys *= vget<T>(c.sw, v);
        T mv = T(1);
        if constexpr (NULLS) {
            mv = ((c.m >> v) & 1u) ? T(1) : T(0);
            ys = (ys == ys) ? ys * mv : T(0);
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const T xi = NULLS ? vget<T>(c.x[i], v) * mv : vget<T>(c.x[i], v);
#pragma unroll
            for (int j = i; j < KT; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int q = tri_index<NZ>(i, j);
                if (q >= Q0 && q < Q1) acc[q - Q0] = fma(xi, vget<T>(c.x[j], v), acc[q - Q0]);
            }
            const int qy = tri_index<NZ>(i, KT);
            if (qy >= Q0 && qy < Q1) acc[qy - Q0] = fma(xi, ys, acc[qy - Q0]);
        }
        const int qq = tri_index<NZ>(KT, KT);
        if (qq >= Q0 && qq < Q1) {
            if constexpr (NULLS) acc[qq - Q0] += mv;
            else acc[qq - Q0] = fma(ys, ys, acc[qq - Q0]);
        }
    }
}

// ---- team reductions shared by K1t and K1p
__device__ __forceinline__ void k1p_swap_rows(float &a, float &b) {      // a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void k1p_swap_rows(double &a, double &b) {
    const unsigned long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    a = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
    b = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}

// Every lane ends up with the totals of its own team (SUB lanes) in acc[].
template <typename T, int NACC, int SUB>
__device__ __forceinline__ void k1p_team_allreduce(T (&acc)[NACC]) {
    if constexpr (SUB == 64) {
        // reduce-scatter over the wave, then v_readlane: row r holds the totals of the entries 4i + rs_perm(r)
        constexpr int NACC4 = (NACC + 3) / 4;
        T u[NACC4];
        wave_reduce_scatter<T, NACC>(acc, u);
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            const int pq = q & 3;
            acc[q] = k1p_readlane(u[q >> 2], 16 * (pq == 1 ? 2 : (pq == 2 ? 1 : pq)));
        }
    } else if constexpr (SUB == 32) {
        // a team is two 16-lane rows: pair_rows leaves entry 2i in the team's even row and 2i + 1 in its odd row (half the values to
        // all-reduce inside the rows), one row swap of the result with itself hands both back to both rows
#pragma unroll
        for (int i = 0; i < (NACC + 1) / 2; ++i) {
            T w = acc[2 * i];
            pair_rows(w, 2 * i + 1 < NACC ? acc[2 * i + 1] : T(0));
            w = row_allreduce(w);
            T w2 = w;
            k1p_swap_rows(w, w2);
            acc[2 * i] = w;
            if (2 * i + 1 < NACC) acc[2 * i + 1] = w2;
        }
    } else {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = row_allreduce(acc[q]);
    }
}

// one pass of the multi-pass Gram: accumulate the entries [Q0, Q1) over the resident chunks, reduce-scatter inside the wave,
// park the wave partials in LDS at their packed slots (Q0 is a multiple of 4, so slot numbering is unchanged)
template <typename T, int KT, bool HAS_W, int RC, int TEAM, int Q0, int Q1, bool NULLS = false>
__device__ __forceinline__ void gram_pass(const Chunk<T, KT, HAS_W> (&res)[RC], int64_t nch, int tid, int lane, int wave, T *mypart) {
    constexpr int N = Q1 - Q0, N4 = (N + 3) / 4, WAVES = TEAM / 64;
    T acc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = T(0);
#pragma unroll
    for (int rc = 0; rc < RC; ++rc)
        if ((int64_t)rc * TEAM + tid < nch) gram_accumulate_range<T, KT, HAS_W, Q0, Q1, NULLS>(acc, res[rc]);
    T u[N4];
    wave_reduce_scatter<T, N>(acc, u);
    const int row = lane >> 4;
    const int pr = (row == 1) ? 2 : ((row == 2) ? 1 : row);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < N4; ++i) mypart[(Q0 + 4 * i + pr) * WAVES + wave] = u[i];
    }
}

// every pass of QS packed entries, first to last (compile-time recursion: each pass is its own fully unrolled code)
template <typename T, int KT, bool HAS_W, int RC, int TEAM, int QS, int Q0, bool NULLS = false>
__device__ __forceinline__ void gram_passes(const Chunk<T, KT, HAS_W> (&res)[RC], int64_t nch, int tid, int lane, int wave, T *mypart) {
    constexpr int NACC = (KT + 1) * (KT + 2) / 2;
    if constexpr (Q0 < NACC) {
        gram_pass<T, KT, HAS_W, RC, TEAM, Q0, (Q0 + QS < NACC ? Q0 + QS : NACC), NULLS>(res, nch, tid, lane, wave, mypart);
        gram_passes<T, KT, HAS_W, RC, TEAM, QS, Q0 + QS, NULLS>(res, nch, tid, lane, wave, mypart);
    }
}

// 1/sqrt(d).  f32: v_rsq_f32 (1 ulp) + one Newton step instead of the ~25-instruction IEEE sqrt + divide
// chain (the Cholesky is a serial dependency chain, so instruction latency is what it costs); f64: IEEE.
__device__ __forceinline__ float inv_sqrt(float d) {
    const float r = __builtin_amdgcn_rsqf(d);
    return r * fmaf(-0.5f * d * r, r, 1.5f);
}
__device__ __forceinline__ double inv_sqrt(double d) { return 1.0 / sqrt(d); }

// Cholesky (LL^T) of G + alpha I and the two triangular solves, fully unrolled on wave-uniform values.
// Returns false on a non-positive pivot (faer's `cholesky(Side::Lower)` Err, ls.rs:289-299).
template <typename T, int KT>
__device__ __forceinline__ bool chol_solve(const T (&acc)[(KT + 1) * (KT + 2) / 2], T alpha, T (&beta)[KT], T pivot_tol = T(0)) {
    constexpr int NZ = KT + 1;
    T L[KT][KT];
    T rinv[KT];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        T d = acc[tri_index<NZ>(j, j)] + alpha;
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-L[j][p], L[j][p], d);
        // d / G_jj = sin^2 of the angle between column j and the span of the columns before it: a tiny ratio means
        // cond(X)^2 exceeds what the normal equations can deliver -> the group goes to the SVD fallback (K6)
        ok = ok && (d > pivot_tol * (acc[tri_index<NZ>(j, j)] + alpha));
        rinv[j] = inv_sqrt(d);   // 1 / L[j][j]; the solves below only ever divide by the diagonal
#pragma unroll
        for (int i = j + 1; i < KT; ++i) {
            T sacc = acc[tri_index<NZ>(j, i)];
#pragma unroll
            for (int p = 0; p < j; ++p) sacc = fma(-L[i][p], L[j][p], sacc);
            L[i][j] = sacc * rinv[j];
        }
    }
    T t[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        T sacc = acc[tri_index<NZ>(i, KT)];
#pragma unroll
        for (int p = 0; p < i; ++p) sacc = fma(-L[i][p], t[p], sacc);
        t[i] = sacc * rinv[i];
    }
#pragma unroll
    for (int i = KT - 1; i >= 0; --i) {
        T sacc = t[i];
#pragma unroll
        for (int p = i + 1; p < KT; ++p) sacc = fma(-L[p][i], beta[p], sacc);
        beta[i] = sacc * rinv[i];
    }
    return ok;
}

// Wave-cooperative variant for the multi-pass kernel: G (packed upper triangle, wave totals already summed) and the factor
// live in LDS, lane i owns row i of L -- a handful of VGPRs instead of the ~130 the unrolled version keeps live, which is
// what decides how many f64 workgroups fit a CU.  Returns this lane's coefficient (lanes >= KT: 0).
// WIDTH: the lanes that cooperate (64: a wave; 16: one DPP row of K1t, `lane` then counts inside the row).
template <typename T, int KT, int WIDTH = 64>
__device__ __forceinline__ T chol_solve_lds(const T *G, T alpha, T pivot_tol, T *L, T *rinv, int lane, bool &ok) {
    constexpr int NZ = KT + 1;
    if (lane < KT) {
#pragma unroll
        for (int j = 0; j < KT; ++j) L[lane * KT + j] = G[lane <= j ? tri_index<NZ>(lane, j) : tri_index<NZ>(j, lane)] + (lane == j ? alpha : T(0));
    }
    T bi = (lane < KT) ? G[tri_index<NZ>(lane, KT)] : T(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    ok = true;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        T d = L[j * KT + j];
        const T gjj = d;
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-L[j * KT + p], L[j * KT + p], d);
        ok = ok && (d > pivot_tol * gjj);
        const T ri = inv_sqrt(d);
        if (lane == 0) rinv[j] = ri;
        if (lane > j && lane < KT) {
            T sacc = L[lane * KT + j];
#pragma unroll
            for (int p = 0; p < j; ++p) sacc = fma(-L[lane * KT + p], L[j * KT + p], sacc);
            L[lane * KT + j] = sacc * ri;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int p = 0; p < KT; ++p) {                      // forward: t = L^-1 b
        if (lane == p) bi *= rinv[p];
        const T tp = __shfl(bi, p, WIDTH);
        if (lane > p && lane < KT) bi = fma(-L[lane * KT + p], tp, bi);
    }
#pragma unroll
    for (int p = KT - 1; p >= 0; --p) {                 // backward: beta = L^-T t
        if (lane == p) bi *= rinv[p];
        const T bp = __shfl(bi, p, WIDTH);
        if (lane < p) bi = fma(-L[p * KT + lane], bp, bi);
    }
    return bi;
}

// 16+ columns: the left-looking form above re-reads O(KT^3) factor entries from LDS one dependent FMA at a time (~60 k cycles at 31
// columns).  Right-looking, lane i keeps ROW i of the factor in registers (statically indexed): step j broadcasts the pivot
// (v_readlane), scales column j in every lane, broadcasts L[p][j] (lane p's register j) and updates r[p] -= r[j] * L[p][j] --
// KT - j independent readlane + FMA pairs per step, ~1 000 instructions at 31 columns.  The transposed access of the backward
// substitution goes through one LDS copy of the factor.  Returns this lane's coefficient (lanes >= KT: 0).
// lane j of every WIDTH-lane team to all of the team's lanes: v_readlane for a whole wave, DPP row_share for 16-lane rows (one
// instruction per 32 bits either way; j is a constant after unrolling, the switch folds)
template <int WIDTH, typename T>
__device__ __forceinline__ T team_bcast(T v, int j) {
    if constexpr (WIDTH == 64) return k1p_readlane(v, j);
    else if constexpr (WIDTH == 8) {                         // two eight-lane teams per DPP row: lane j of THIS half = row lane j or j + 8
        const T lo = team_bcast<16>(v, j), hi = team_bcast<16>(v, j + 8);
        return (threadIdx.x & 8) ? hi : lo;
    } else {
        switch (j) {
            case 0: return dpp_get<0x150>(v); case 1: return dpp_get<0x151>(v); case 2: return dpp_get<0x152>(v); case 3: return dpp_get<0x153>(v);
            case 4: return dpp_get<0x154>(v); case 5: return dpp_get<0x155>(v); case 6: return dpp_get<0x156>(v); case 7: return dpp_get<0x157>(v);
            case 8: return dpp_get<0x158>(v); case 9: return dpp_get<0x159>(v); case 10: return dpp_get<0x15A>(v); case 11: return dpp_get<0x15B>(v);
            case 12: return dpp_get<0x15C>(v); case 13: return dpp_get<0x15D>(v); case 14: return dpp_get<0x15E>(v); default: return dpp_get<0x15F>(v);
        }
    }
}

template <typename T, int KT, int WIDTH = 64>
__device__ __forceinline__ T chol_solve_rows(const T *G, T alpha, T pivot_tol, T *L, int lane, bool &ok) {
    static_assert(WIDTH == 64 || (WIDTH == 16 && KT <= 16) || (WIDTH == 8 && KT <= 8), "a wave, one 16-lane DPP row per group (K1t), or half of one");
    constexpr int NZ = KT + 1;
    const int li = lane < KT ? lane : KT - 1;                // lanes beyond the matrix mirror the last row (their results are unused)
    T r[KT];
#pragma unroll
    for (int p = 0; p < KT; ++p) r[p] = G[li <= p ? tri_index<NZ>(li, p) : tri_index<NZ>(p, li)] + (li == p ? alpha : T(0));
    T bi = (lane < KT) ? G[tri_index<NZ>(li, KT)] : T(0);
    T g0 = T(0), myrinv = T(1);
#pragma unroll
    for (int p = 0; p < KT; ++p) g0 = (li == p) ? r[p] : g0;  // this lane's original diagonal entry
    ok = true;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const T d = team_bcast<WIDTH>(r[j], j);
        const T gjj = team_bcast<WIDTH>(g0, j);
        ok = ok && (d > pivot_tol * gjj);
        const T ri = inv_sqrt(d);
        myrinv = (lane == j) ? ri : myrinv;
        r[j] *= ri;                                          // lanes i >= j: L[i][j]  (lane j: sqrt(d))
#pragma unroll
        for (int p = j + 1; p < KT; ++p) r[p] = fma(-r[j], team_bcast<WIDTH>(r[j], p), r[p]);
    }
    if (lane < KT) {
#pragma unroll
        for (int p = 0; p < KT; ++p) L[lane * KT + p] = r[p];
    }
    if constexpr (WIDTH != 64) {
        // K1t teams: both substitutions as selects -- an exec-mask branch per step cost more than the FMA it guarded.  (Not for the wave
        // form: there the selects cost 1-4 VGPRs, an occupancy step at 6 f32 columns and an AGPR at 28 f64 ones.)
#pragma unroll
        for (int p = 0; p < KT; ++p) {                       // forward: t = L^-1 b
            const T sc = bi * myrinv;
            bi = (lane == p) ? sc : bi;
            const T tp = team_bcast<WIDTH>(bi, p);
            const T nb = fma(-r[p], tp, bi);
            bi = (lane > p && lane < KT) ? nb : bi;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = KT - 1; p >= 0; --p) {                  // backward: beta = L^-T t (column `li` of the factor, clamped: no guard)
            const T sc = bi * myrinv;
            bi = (lane == p) ? sc : bi;
            const T bp = team_bcast<WIDTH>(bi, p);
            const T nb = fma(-L[p * KT + li], bp, bi);
            bi = (lane < p) ? nb : bi;
        }
    } else {
#pragma unroll
        for (int p = 0; p < KT; ++p) {                       // forward: t = L^-1 b
            if (lane == p) bi *= myrinv;
            const T tp = team_bcast<WIDTH>(bi, p);
            if (lane > p && lane < KT) bi = fma(-r[p], tp, bi);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int p = KT - 1; p >= 0; --p) {                  // backward: beta = L^-T t
            if (lane == p) bi *= myrinv;
            const T bp = team_bcast<WIDTH>(bi, p);
            if (lane < p) bi = fma(-L[p * KT + lane], bp, bi);
        }
    }
    return bi;
}

template <typename T, int KT, bool HAS_W, bool FAST, bool NULLS = false>
__device__ __forceinline__ void predict_store(const K1Args &a, const Chunk<T, KT, HAS_W> &c, const T (&beta)[KT],
                                              int64_t row0, int64_t s, int64_t e) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    V p, r;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) acc = fma(vget<T>(c.x[j], v), beta[j], acc);   // make_predictions on the FIT features (ex.rs:398-405)
        if constexpr (HAS_W) acc *= T(1) / vget<T>(c.sw, v);                         // predictions *= 1/sqrt_w (ls.py:234-235)
        if constexpr (NULLS) {                                                       // "drop" masks the rows that were not fitted (ex.rs:409-417)
            if (a.null_policy == POLS_NULL_DROP) acc = nan_if<T>(((c.m >> v) & 1u) ? 0u : 1u, acc);
        }
        vset<T>(p, v, acc);
        vset<T>(r, v, vget<T>(c.y, v) - acc);                                        // ORIGINAL target - predictions (ls.py:239)
    }
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    if (FAST || (row0 >= s && row0 + VEC <= e)) {
        if (pred) store_stream(reinterpret_cast<V *>(pred + row0), p);
        if (resid) store_stream(reinterpret_cast<V *>(resid + row0), r);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t rr = row0 + v;
            if (rr >= s && rr < e) {
                if (pred) pred[rr] = vget<T>(p, v);
                if (resid) resid[rr] = vget<T>(r, v);
            }
        }
    }
}

// The same with the coefficients read from LDS where they are used (wide multi-pass kernels: KT more live registers in this phase were what
// kept the f32 kernels of 26+ columns at two waves per SIMD -- 173-205 VGPRs however short the Gram passes)
template <typename T, int KT, bool HAS_W, bool FAST, bool NULLS = false>
__device__ __forceinline__ void predict_store_lds(const K1Args &a, const Chunk<T, KT, HAS_W> &c, const T *beta_lds, int64_t row0, int64_t s, int64_t e) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = T(0);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        if (j % 8 == 0) __builtin_amdgcn_sched_barrier(0);                       // (eight coefficients in flight at a time, not KT)
        const T bj = beta_lds[j];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fma(vget<T>(c.x[j], v), bj, acc[v]);   // make_predictions on the FIT features (ex.rs:398-405)
    }
    V p, r;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T pv = acc[v];
        if constexpr (HAS_W) pv *= T(1) / vget<T>(c.sw, v);                          // predictions *= 1/sqrt_w (ls.py:234-235)
        if constexpr (NULLS) {                                                       // "drop" masks the rows that were not fitted (ex.rs:409-417)
            if (a.null_policy == POLS_NULL_DROP) pv = nan_if<T>(((c.m >> v) & 1u) ? 0u : 1u, pv);
        }
        vset<T>(p, v, pv);
        vset<T>(r, v, vget<T>(c.y, v) - pv);                                         // ORIGINAL target - predictions (ls.py:239)
    }
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    if (FAST || (row0 >= s && row0 + VEC <= e)) {
        if (pred) store_stream(reinterpret_cast<V *>(pred + row0), p);
        if (resid) store_stream(reinterpret_cast<V *>(resid + row0), r);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t rr = row0 + v;
            if (rr >= s && rr < e) {
                if (pred) pred[rr] = vget<T>(p, v);
                if (resid) resid[rr] = vget<T>(r, v);
            }
        }
    }
}

// TEAM = 64: four independent waves per 256-thread block, one group each, no LDS, no barriers.
// TEAM = 256: one group per block; cross-wave reduction through LDS with ONE barrier.
// FAST: the host verified that every group starts on a 16-byte boundary, has a multiple of VEC rows and fits
// the RC * TEAM resident chunks -> no ragged-edge and no overflow code (fewer VGPRs, more groups in flight).
// NPASS > 1 (every row resident -- FAST, or the host checked max_rows against the capacity): the Gram is accumulated in NPASS passes
// over the resident registers, each keeping 1 / NPASS of the
// accumulators live -- fewer VGPRs, one more workgroup per CU for the f64 team kernel.
// EDGE (with FAST): the branch-free form for RAGGED frames whose groups stay resident -- every load of every chunk issued up front
// like FAST (unconditional, clamped into the columns), then the rows outside [s, e) -- the head / tail chunks' neighbours' rows --
// zeroed in registers; stores of edge chunks guarded per row.  The general (FAST = false) code loads chunk by chunk inside per-lane
// branches: with the same group sizes it runs 797 vs 679 us (130..252 rows) and 118 vs 91 us (100..300 rows) behind FAST.
template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS = 1, bool NULLS = false, bool NT = false,
          bool EDGE = false, int BLOCK = 256>
__device__ __forceinline__ void k1_body(const K1Args &a, const int64_t bid) {
    static_assert(!EDGE || FAST, "EDGE refines FAST");
    // (tried: a PERSISTENT form of the 256-thread team -- one workgroup per resident slot walking the groups b, b + gridDim.x, ... --
    // 90 registers once the thread index is laundered per group (109 without), and SLOWER: 73.3 us at five workgroups per CU, 75.8 at
    // four, against 66.9 us for the one-shot grid on the same box.  The hardware's workgroup dispatcher refills a finished slot
    // faster than a loop iteration that must drain its stores and re-synchronise.)
    const int tix = threadIdx.x;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
    constexpr int NACC = NZ * (NZ + 1) / 2;
    constexpr int WAVES = TEAM / 64;
    const int lane = tix & 63;
    const int wave = (tix >> 6) % WAVES;
    const int tid = tix % TEAM;
    const int64_t gi = bid * (blockDim.x / TEAM) + tix / TEAM;  // bid: blockIdx.x, or the persistent kernel's walk (blockDim.x: 256)
    if (gi >= a.n_groups) return;  // wave-uniform for TEAM=64; never taken for TEAM=256 (grid == n_groups)
    const int64_t g = a.glist ? (int64_t)a.glist[gi] : gi;      // (size classes: the launch walks its own list of groups)

    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);                      // chunk grid is aligned to 16 bytes in every column
    const int64_t nch = (e - base + VEC - 1) / VEC;

    // the multi-pass Gram has no streamed-overflow path: NPASS > 1 without FAST is launched only when every group fits the registers
    T acc[NACC];
    if constexpr (NPASS == 1) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = T(0);
    }

    // rows beyond register capacity are streamed (Gram pass now, prediction pass at the end); done BEFORE the
    // resident chunks are loaded so the two never share registers
    unsigned long long *dbg = a.dbg ? a.dbg + g * 8 : nullptr;
#define K1_STAMP(i) do { if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    K1_STAMP(0);
    if constexpr (!FAST && NPASS == 1) {
        for (int64_t c = (int64_t)RC * TEAM + tid; c < nch; c += TEAM) {
            Chunk<T, KT, HAS_W> tmp;
            load_chunk<T, KT, HAS_W, false, NULLS>(a, base + c * VEC, s, e, tmp);
            gram_accumulate<T, KT, HAS_W, NULLS>(acc, tmp);
        }
    }
    Chunk<T, KT, HAS_W> res[RC];                             // register-resident rows of this lane
    if constexpr (FAST) {
        // every 16-byte load of every resident chunk is issued before the first use: RC x (KT + 1) requests in flight per lane
        // instead of KT + 1 (a lane without a chunk re-reads chunk 0, which exists whenever the group has rows)
        if (nch > 0) {
#pragma unroll
            for (int rc = 0; rc < RC; ++rc) {
                const int64_t c = (int64_t)rc * TEAM + tid;
                int64_t r0 = base + (c < nch ? c : 0) * VEC;
                if constexpr (EDGE) r0 = r0 > a.n_rows - VEC ? a.n_rows - VEC : r0;   // (the one chunk that crosses the end: re-read below)
                load_chunk_raw<T, KT, HAS_W, NT>(a, r0, res[rc]);
            }
        }
        if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); K1_STAMP(1); }
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t c = (int64_t)rc * TEAM + tid;
            if (c < nch) {
                if constexpr (EDGE) {
                    const int64_t row0 = base + c * VEC;
                    if (row0 + VEC > a.n_rows) {                     // at most one lane per launch
                        load_chunk<T, KT, HAS_W, false, NULLS>(a, row0, s, e, res[rc]);
                    } else {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const bool in = (row0 + v >= s) && (row0 + v < e);
#pragma unroll
                            for (int j = 0; j < KT; ++j) vset<T>(res[rc].x[j], v, in ? vget<T>(res[rc].x[j], v) : T(0));
                            vset<T>(res[rc].y, v, in ? vget<T>(res[rc].y, v) : T(0));
                            if constexpr (HAS_W) vset<T>(res[rc].sw, v, in ? vget<T>(res[rc].sw, v) : T(1));
                        }
                        load_chunk<T, KT, HAS_W, true, NULLS, true>(a, row0, s, e, res[rc]);
                    }
                } else {
                    load_chunk<T, KT, HAS_W, FAST, NULLS, true>(a, base + c * VEC, s, e, res[rc]);
                }
                if constexpr (NPASS == 1) gram_accumulate<T, KT, HAS_W, NULLS>(acc, res[rc]);
            }
        }
    } else {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t c = (int64_t)rc * TEAM + tid;
            if (c < nch) {
                load_chunk<T, KT, HAS_W, FAST, NULLS, false, (TEAM == 256 && RC <= 2)>(a, base + c * VEC, s, e, res[rc]);
                if (rc == RC - 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); K1_STAMP(1); }
                if constexpr (NPASS == 1) gram_accumulate<T, KT, HAS_W, NULLS>(acc, res[rc]);
            }
        }
    }
    K1_STAMP(2);

    // ---- team reduction, fixed order (deterministic): reduce-scatter inside each wave, partials through LDS
    constexpr int NACC4 = (NACC + 3) / 4;
    constexpr int SLOTS = NACC4 * 4;                         // accumulator slots, padded to a multiple of 4
    __shared__ __attribute__((aligned(16))) T part[(BLOCK / TEAM) * (SLOTS * WAVES + 32)];
    T *mypart = part + (tix / TEAM) * (SLOTS * WAVES + 32);   // one region per team
    T *bcast = mypart + SLOTS * WAVES;                       // beta broadcast, 32 slots
    if constexpr (NPASS == 1) {
        T u[NACC4];
        wave_reduce_scatter<T, NACC>(acc, u);
        const int row = lane >> 4;
        const int pr = (row == 1) ? 2 : ((row == 2) ? 1 : row);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < NACC4; ++i) mypart[(4 * i + pr) * WAVES + wave] = u[i];
        }
    } else {
        constexpr int QS = (((NACC + NPASS - 1) / NPASS) + 3) & ~3;    // entries per pass, a multiple of 4
        static_assert(NPASS * QS >= NACC, "the passes cover the packed triangle");
        gram_passes<T, KT, HAS_W, RC, TEAM, QS, 0, NULLS>(res, nch, tid, lane, wave, mypart);
    }
    if constexpr (WAVES > 1) __syncthreads();
    K1_STAMP(3);

    // ---- K x K solve on wave-uniform values: ONE wave per team solves (the others would only burn the
    // SIMDs' VALU issue slots that co-resident workgroups need), beta goes back through LDS
    T beta[KT];
    if constexpr (NPASS > 1) {
        constexpr int TEAMS = BLOCK / TEAM;                        // one scratch set per team of the block
        __shared__ T gsum_s[TEAMS][NACC + 3], lfac_s[TEAMS][KT * KT], lrinv_s[TEAMS][KT];
        T *gsum = gsum_s[tix / TEAM], *lfac = lfac_s[tix / TEAM], *lrinv = lrinv_s[tix / TEAM];
        // f32 from 26 columns: the solving wave PARKS its resident rows in LDS for the duration of the solve (the right-looking factorisation
        // keeps a row of the factor and its broadcasts next to them: 173-205 VGPRs, two waves per SIMD, however short the Gram passes are;
        // without the rows the kernel fits 168 and a third workgroup fits the CU)
        constexpr bool PARK = ((sizeof(T) == 4 && KT >= 25) || (sizeof(T) == 8 && KT >= 18)) && RC == 1 && TEAM == 256;   // (f64: 257 .. 512-row groups, the same step at 18 columns)
        using PV = typename Vec16<T>::type;
        __shared__ __attribute__((aligned(16))) PV park_s[PARK ? (KT + 2) * 64 : 1];
        if (wave == 0) {
            if constexpr (PARK) {
#pragma unroll
                for (int j = 0; j < KT; ++j) park_s[j * 64 + lane] = res[0].x[j];
                park_s[KT * 64 + lane] = res[0].y;
                if constexpr (HAS_W) park_s[(KT + 1) * 64 + lane] = res[0].sw;
            }
            for (int q = lane; q < NACC; q += 64) {     // NACC = 66 at 10 columns
                T t = mypart[q * WAVES];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) t += mypart[q * WAVES + w];
                gsum[q] = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int st = POLS_GROUP_OK;
            T bv = T(0);
            if (e == s || (NULLS && gsum[tri_index<NZ>(KT, KT)] == T(0))) st = POLS_GROUP_EMPTY;   // no row left in the fit -> zeros (ex.rs:357-359)
            else {
                bool ok;
                if constexpr (K1_SOLVE_ROWS || KT > 15) bv = chol_solve_rows<T, KT>(gsum, (T)a.alpha, (T)a.pivot_tol, lfac, lane, ok);
                else bv = chol_solve_lds<T, KT>(gsum, (T)a.alpha, (T)a.pivot_tol, lfac, lrinv, lane, ok);
                if (!ok) { st = POLS_GROUP_FALLBACK; if (tid == 0 && a.fb_flag) *a.fb_flag = a.epoch; }
            }
            if (tid == 0 && a.status) a.status[g] = st;
            if (tid < KT) {
                if (a.coef) static_cast<T *>(a.coef)[g * KT + tid] = bv;
                bcast[tid] = bv;
            }
            if constexpr (PARK) {
                __builtin_amdgcn_sched_barrier(0);                 // (not before the solve is over)
#pragma unroll
                for (int j = 0; j < KT; ++j) res[0].x[j] = park_s[j * 64 + lane];
                res[0].y = park_s[KT * 64 + lane];
                if constexpr (HAS_W) res[0].sw = park_s[(KT + 1) * 64 + lane];
            }
        }
        if constexpr (WAVES > 1) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        if constexpr (!(NPASS > 1 && KT >= 23)) {              // (23+ columns: the prediction phase reads them from LDS as it goes, predict_store_lds)
#pragma unroll
            for (int j = 0; j < KT; ++j) beta[j] = bcast[j];
        }
    } else
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            T t = mypart[q * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += mypart[q * WAVES + w];
            acc[q] = t;
        }
        int st = POLS_GROUP_OK;
        if (e == s || (NULLS && acc[tri_index<NZ>(KT, KT)] == T(0))) {   // features.is_empty() -> zeros (ex.rs:357-359)
#pragma unroll
            for (int j = 0; j < KT; ++j) beta[j] = T(0);
            st = POLS_GROUP_EMPTY;
        } else {
            const bool ok = chol_solve<T, KT>(acc, (T)a.alpha, beta, (T)a.pivot_tol);
            if (!ok) {                                       // K6 re-solves this group
                st = POLS_GROUP_FALLBACK;
                if (tid == 0 && a.fb_flag) *a.fb_flag = a.epoch;
            }
        }
        if (tid == 0 && a.status) a.status[g] = st;
        if (tid < KT) {
            T bv = T(0);
#pragma unroll
            for (int j = 0; j < KT; ++j) bv = (tid == j) ? beta[j] : bv;
            if (a.coef) static_cast<T *>(a.coef)[g * KT + tid] = bv;
            if constexpr (WAVES > 1) bcast[tid] = bv;
        }
    }
    if constexpr (WAVES > 1 && NPASS == 1) {
        __syncthreads();
        if (wave != 0) {
#pragma unroll
            for (int j = 0; j < KT; ++j) beta[j] = bcast[j];
        }
    }

    K1_STAMP(4);
    // ---- fused predictions / residuals from the resident rows, then the streamed overflow rows
    if (a.pred || a.resid) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t c = (int64_t)rc * TEAM + tid;
            if constexpr (NPASS > 1 && KT >= 23) {
                if (c < nch) predict_store_lds<T, KT, HAS_W, (FAST && !EDGE), NULLS>(a, res[rc], bcast, base + c * VEC, s, e);
            } else {
                if (c < nch) predict_store<T, KT, HAS_W, (FAST && !EDGE), NULLS>(a, res[rc], beta, base + c * VEC, s, e);
            }
        }
        if constexpr (!FAST && NPASS == 1) {
            for (int64_t c = (int64_t)RC * TEAM + tid; c < nch; c += TEAM) {
                Chunk<T, KT, HAS_W> tmp;
                load_chunk<T, KT, HAS_W, false, NULLS>(a, base + c * VEC, s, e, tmp);
                predict_store<T, KT, HAS_W, false, NULLS>(a, tmp, beta, base + c * VEC, s, e);
            }
        }
    }
    K1_STAMP(5);
    if (dbg && tid == 0) dbg[6] = 0;
#undef K1_STAMP
}

// workgroup -> block of groups.  The hardware deals workgroups round-robin over the eight XCDs, so the identity map sends neighbouring
// groups -- whose slices share the 128-byte lines at their ends -- to eight different L2s; with xcd_chunk set, XCD x walks the blocks
// [x * xcd_chunk, (x + 1) * xcd_chunk) in order (grid padded to 8 * xcd_chunk; the body returns for blocks past the last group).
__device__ __forceinline__ int64_t k1_block_id(const K1Args &a) {
    const int64_t b = blockIdx.x;
    return a.xcd_chunk ? (b & 7) * a.xcd_chunk + (b >> 3) : b;
}

template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS = 1, bool NULLS = false, bool NT = false,
          bool EDGE = false>
__global__ void __launch_bounds__(256) k1_kernel(const K1Args a) {
    k1_body<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS, NT, EDGE>(a, k1_block_id(a));
}
// SEVERAL teams per workgroup (POLS_K1_WG=2|4, A/B): BLOCK = 512 / 1 024 threads = 2 / 4 times the teams of the 256-thread block, i.e. four
// consecutive groups of the headline shape per workgroup -- their 4 x 4 000-byte column slices are exactly 125 whole 128-byte lines,
// and the teams meet at the workgroup's barriers, so their loads leave together and their stores leave together (the "16 KB pieces,
// burst stores" cell of profiles/r05_probe_matrix.txt).  Same arithmetic, same outputs.
template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k1_kernel_wg(const K1Args a) {
    k1_body<T, KT, HAS_W, TEAM, RC, FAST, NPASS, false, NT, false, BLOCK>(a, k1_block_id(a));
}
// The same body held to 128 VGPRs (four waves per SIMD): the ragged one-chunk-per-lane wave kernel needs 130.
template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS = 1, bool NULLS = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) k1_kernel_occ4(const K1Args a) {
    k1_body<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS>(a, k1_block_id(a));
}

// (the null-policy wave kernel with 16 resident rows sits at exactly 256 VGPRs; one more value and the allocator reaches for an AGPR,
// which halves the occupancy of the unified register file: 80.7 -> 132 us on 10 000 x 1 000 x 8.  Held to two waves per SIMD.)
template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS = 1, bool NULLS = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k1_kernel_occ2(const K1Args a) {
    k1_body<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS>(a, k1_block_id(a));
}

// One chunk of a small ragged group, branch-free: the 16-byte loads of every column issued unconditionally (lanes without a chunk
// re-read the group's first chunk; positions clamped into the columns), then the rows outside [s, e) zeroed in registers.  In
// these kernels nearly every chunk is a head or a tail, so the guarded per-row path was what every wave executed.
template <typename T, int KT, bool HAS_W, bool KUS = false>
__device__ __forceinline__ void load_chunk_edge(const K1Args &a, int64_t row0, int64_t base, bool has, int64_t s, int64_t e,
                                                Chunk<T, KT, HAS_W> &c) {
    constexpr int VEC = Vec16<T>::N;
    int64_t r0 = has ? row0 : base;
    r0 = r0 > a.n_rows - VEC ? a.n_rows - VEC : r0;
    // f32 only, both: in the f64 kernels the older forms allocate 4-9 fewer VGPRs (code-object metadata over 6-10 columns: never a lower
    // occupancy, 3 instead of 2 waves per SIMD at 8 columns x two chunks per lane), in the f32 ones the new forms save up to 28
    constexpr bool SHIFT = K1_EDGE_SHIFT && sizeof(T) == 4, MASK32 = K1_EDGE_MASK32 && sizeof(T) == 4;
    load_chunk_raw<T, KT, HAS_W, false, KUS>(a, r0, c);
    if (!SHIFT && has && row0 + VEC > a.n_rows) {
        load_chunk<T, KT, HAS_W, false>(a, row0, s, e, c);
        return;
    }
    if (SHIFT && has && row0 + VEC > a.n_rows) {     // the chunk that crosses the end of the columns: at most one lane per launch
        // its 16 bytes were read from the last VEC rows: move them down to their slots (the slots past the end are masked below)
        const int shift = (int)(row0 - r0);                  // 1 .. VEC - 1
        for (int k = 0; k < shift; ++k) {
#pragma unroll
            for (int v = 0; v + 1 < VEC; ++v) {
#pragma unroll
                for (int j = 0; j < KT; ++j) vset<T>(c.x[j], v, vget<T>(c.x[j], v + 1));
                vset<T>(c.y, v, vget<T>(c.y, v + 1));
                if constexpr (HAS_W) vset<T>(c.sw, v, vget<T>(c.sw, v + 1));
            }
        }
    }
    // (row0 lies on the chunk grid of a register-resident group: the distances to its ends fit 32 bits, one compare each per row)
    const int lo = has ? (int)(s - row0) : VEC, hi = (int)(e - row0);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const bool in = MASK32 ? ((v >= lo) && (v < hi)) : (has && (row0 + v >= s) && (row0 + v < e));
#pragma unroll
        for (int j = 0; j < KT; ++j) vset<T>(c.x[j], v, in ? vget<T>(c.x[j], v) : T(0));
        vset<T>(c.y, v, in ? vget<T>(c.y, v) : T(0));
        if constexpr (HAS_W) vset<T>(c.sw, v, in ? vget<T>(c.sw, v) : T(1));
    }
    if constexpr (HAS_W) load_chunk<T, KT, HAS_W, true, false, true>(a, row0, s, e, c);   // the sqrt(w) scaling only
}

// all-reduce inside a K1t team of 16 lanes (one DPP row: four fused DPP adds) or 8 lanes (half a row: three)
template <int SUB, typename T>
__device__ __forceinline__ T k1t_team_allreduce(T v) {
    v += dpp_get<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);                                   // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);                                  // row_half_mirror
    if constexpr (SUB >= 16) v += dpp_get<0x140>(v);         // row_mirror
    return v;
}

// REDUCE-SCATTER inside a K1t team.  The all-reduce above costs every entry 4 dependent DPP adds, and handing entry q to the lane
// that parks it in LDS cost a compare + exec-mask branch per entry on top: 540 of the ~1 400 issue slots a wave of the 8-column
// kernel spent (PMC, 500 000 groups of 12..40 rows: 1 084 VALU + 216 SALU per wave, VALU ~65 % busy).  Here each step adds the two
// lanes of a DPP pair and keeps ONE of two entries per lane -- the entry count halves per step, 3 instructions per surviving entry:
// 16 entries -> 8 -> 4 -> 2 -> 1 in 45 instructions, and the lane that ends up with an entry is the one that stores it.
// Step order row_mirror (lanes i, 15 - i), row_half_mirror (i, 7 - i), quad xor 1, quad xor 2: the two lanes of a pair always hold the
// same entries (they agree in the lane bits of the earlier steps), and the four steps together reach all 16 lanes.
template <int CTRL, int M, typename T, int N>
__device__ __forceinline__ void k1t_rs_step(T (&v)[N], bool side) {      // v[j] <- pair total of entry 2j + side, j < (M + 1) / 2
#pragma unroll
    for (int j = 0; j < (M + 1) / 2; ++j) {
        const T lo = v[2 * j] + dpp_get<CTRL>(v[2 * j]);
        if (2 * j + 1 < M) {
            const T hi = v[2 * j + 1] + dpp_get<CTRL>(v[2 * j + 1]);
            v[j] = side ? hi : lo;
        } else {
            v[j] = side ? T(0) : lo;
        }
    }
}
// the entry of slot t a lane holds afterwards is SUB * t + k1t_rs_slot<SUB>(lane)
template <int SUB>
__device__ __forceinline__ int k1t_rs_slot(int l) {
    if constexpr (SUB == 16) return ((l >> 3) & 1) | (((l >> 2) & 1) << 1) | ((l & 1) << 2) | (((l >> 1) & 1) << 3);
    else return ((l >> 2) & 1) | ((l & 1) << 1) | (((l >> 1) & 1) << 2);
}
template <int SUB, typename T, int N>
__device__ __forceinline__ void k1t_team_reduce_scatter(T (&v)[N], int l) {
    static_assert(SUB == 8 || SUB == 16, "half a DPP row or one");
    constexpr int M1 = SUB == 16 ? (N + 1) / 2 : N, M2 = (M1 + 1) / 2, M3 = (M2 + 1) / 2;
    if constexpr (SUB == 16) k1t_rs_step<0x140, N>(v, (l & 8) != 0);
    k1t_rs_step<0x141, M1>(v, (l & 4) != 0);
    k1t_rs_step<0xB1, M2>(v, (l & 1) != 0);
    k1t_rs_step<0x4E, M3>(v, (l & 2) != 0);
}

// The K1t Gram, 16 packed entries per pass (a pass keeps 16 accumulators live whatever the column count): accumulate over the lane's
// resident chunks, reduce-scatter inside the team, park the totals in the team's LDS row.  The y'y entry (the last one) is not needed
// for the coefficients and is left out.
template <typename T, int KT, bool HAS_W, int SUB, int RC, int Q0>
__device__ __forceinline__ void k1t_gram_passes(const Chunk<T, KT, HAS_W> (&res)[RC], int sub, T *grow) {
    constexpr int NUSE = (KT + 1) * (KT + 2) / 2 - 1;
    if constexpr (Q0 < NUSE) {
        constexpr int Q1 = Q0 + 16 < NUSE ? Q0 + 16 : NUSE;
        T p[Q1 - Q0];
#pragma unroll
        for (int q = 0; q < Q1 - Q0; ++q) p[q] = T(0);
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) gram_accumulate_range<T, KT, HAS_W, Q0, Q1>(p, res[rc]);   // (a lane without this chunk holds zeros: no guard, no branch)
        k1t_team_reduce_scatter<SUB>(p, sub);
        const int slot = k1t_rs_slot<SUB>(sub);
#pragma unroll
        for (int t = 0; t < (Q1 - Q0 + SUB - 1) / SUB; ++t) grow[Q0 + SUB * t + slot] = p[t];
        k1t_gram_passes<T, KT, HAS_W, SUB, RC, Q1>(res, sub, grow);
    }
}

#ifndef K1_NULLS_TU
// K1t: groups of at most SUB * 2 * VEC rows -- SUB = 16: 128 f32 / 64 f64 rows (per-asset-per-month sized regressions), FOUR groups
// per wave, one per 16-lane DPP row; SUB = 32: 256 / 128 rows, TWO groups per wave (one more all-reduce step: v_permlane16_swap).  A wave-per-group kernel spends its time in the per-group reduction + Cholesky with most lanes idle (650 M
// groups/s whatever the size below 100 rows); here the reduction is a 4-step row all-reduce, after which every lane holds its own
// group's Gram matrix and the (unrolled, register-resident) Cholesky runs once per wave for four groups.  No LDS, no barriers, no
// early exit (the DPP steps need all 64 lanes active).
template <typename T, int KT, bool HAS_W, int K1T_SUB, int K1T_RC>
__global__ void __launch_bounds__(256) k1t_kernel(const K1Args a) {
    // (no occupancy hint: asking four waves per SIMD of the f64 eight-lane team with two chunks per lane -- 129-170 VGPRs -- spilled 12-192 B)
    static_assert(K1T_SUB == 8 || K1T_SUB == 16 || K1T_SUB == 32, "a team is half a DPP row, one, or two");
    static_assert(K1T_SUB != 8 || KT <= 8, "eight-lane teams: one coefficient per lane");
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
    constexpr int NACC = NZ * (NZ + 1) / 2;
    const int lane = threadIdx.x & 63, sub = lane & (K1T_SUB - 1);
    const int64_t gi = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / K1T_SUB) + (lane / K1T_SUB);
    const bool live = gi < a.n_groups;
    const int64_t g = live ? (a.glist ? (int64_t)a.glist[gi] : gi) : 0;   // (size classes: the launch walks its own list of groups)
    const int64_t s = live ? a.offs[g] : 0, e = live ? a.offs[g + 1] : 0;
    const int64_t base = s & ~(int64_t)(VEC - 1);            // chunk grid aligned to 16 bytes in every column (offsets are >= 0)
    const int nch = (int)((e - base + VEC - 1) / VEC);       // <= K1T_SUB * K1T_RC: the host checked the largest group
    constexpr bool TWO_PASS = (sizeof(T) == 8 || K1T_F32_TWO_PASS) && KT >= K1T_PASS_MIN_KT;
    T acc[TWO_PASS ? 1 : NACC];
    Chunk<T, KT, HAS_W> res[K1T_RC];
    T beta[KT];
    int st = POLS_GROUP_OK;
    if constexpr (TWO_PASS) {
        // f64, 6+ columns: 45 f64 accumulators + the unrolled Cholesky are 230-256 VGPRs (two waves per SIMD).  Like the f64 team
        // kernel, the Gram is taken in two passes over the resident rows (half the accumulators live at a time), the row totals go to
        // LDS, and the row solves cooperatively there (lane i of the row owns row i of L).
        constexpr int TEAMS = 256 / K1T_SUB;
        constexpr int Q1 = ((NACC + 1) / 2 + 3) & ~3;        // entries of the first pass (32-lane teams)
        constexpr int GS = (NACC + 15) & ~15;                // the passes park whole 16-entry slices
        __shared__ T gs[TEAMS][GS], lf[TEAMS][KT * KT], lr[TEAMS][KT], bc[TEAMS][KT];
        const int team = threadIdx.x / K1T_SUB;
#pragma unroll
        for (int rc = 0; rc < K1T_RC; ++rc) {
            const int64_t c = (int64_t)rc * K1T_SUB + sub;
            load_chunk_edge<T, KT, HAS_W, true>(a, base + c * VEC, base, c < nch, s, e, res[rc]);
        }
        if constexpr (K1T_SUB == 32) {                       // a team is two 16-lane rows: the pairing reduction of K1p, two passes
            {
                T p1[Q1];
#pragma unroll
                for (int q = 0; q < Q1; ++q) p1[q] = T(0);
#pragma unroll
                for (int rc = 0; rc < K1T_RC; ++rc)
                    if ((int64_t)rc * K1T_SUB + sub < nch) gram_accumulate_range<T, KT, HAS_W, 0, Q1>(p1, res[rc]);
                k1p_team_allreduce<T, Q1, 32>(p1);
#pragma unroll
                for (int q = 0; q < Q1; ++q) if ((q % K1T_SUB) == sub) gs[team][q] = p1[q];
            }
            {
                T p2[NACC - Q1];
#pragma unroll
                for (int q = 0; q < NACC - Q1; ++q) p2[q] = T(0);
#pragma unroll
                for (int rc = 0; rc < K1T_RC; ++rc)
                    if ((int64_t)rc * K1T_SUB + sub < nch) gram_accumulate_range<T, KT, HAS_W, Q1, NACC>(p2, res[rc]);
                k1p_team_allreduce<T, NACC - Q1, 32>(p2);
#pragma unroll
                for (int q = 0; q < NACC - Q1; ++q) if (((q + Q1) % K1T_SUB) == sub) gs[team][q + Q1] = p2[q];
            }
        } else {
            k1t_gram_passes<T, KT, HAS_W, K1T_SUB, K1T_RC, 0>(res, sub, gs[team]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        T bv = T(0);
        if (e == s) st = POLS_GROUP_EMPTY;
        else {
            bool ok;
            // (32-lane teams: the factor lives in the team's FIRST 16-lane row -- KT <= 16 -- whose lanes broadcast among themselves;
            // the second row runs along on copies of the last matrix row and its results are never read)
            if constexpr (K1T_SUB == 8) bv = chol_solve_rows<T, KT, 8>(gs[team], (T)a.alpha, (T)a.pivot_tol, lf[team], sub, ok);
            else if constexpr (K1_SOLVE_ROWS || K1T_SUB == 32) bv = chol_solve_rows<T, KT, 16>(gs[team], (T)a.alpha, (T)a.pivot_tol, lf[team], sub, ok);
            else bv = chol_solve_lds<T, KT, K1T_SUB>(gs[team], (T)a.alpha, (T)a.pivot_tol, lf[team], lr[team], sub, ok);
            if (!ok) st = POLS_GROUP_FALLBACK;
        }
        if (sub < KT) bc[team][sub] = bv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = bc[team][j];
    } else {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = T(0);
#pragma unroll
    for (int rc = 0; rc < K1T_RC; ++rc) {
        const int64_t c = (int64_t)rc * K1T_SUB + sub;
        load_chunk_edge<T, KT, HAS_W, true>(a, base + c * VEC, base, c < nch, s, e, res[rc]);
        if (c < nch) gram_accumulate<T, KT, HAS_W>(acc, res[rc]);
    }
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        if constexpr (K1T_SUB == 8) acc[q] = k1t_team_allreduce<8>(acc[q]);
        else {
            acc[q] = row_allreduce(acc[q]);
            if constexpr (K1T_SUB == 32) { T t = acc[q]; pair_rows(t, acc[q]); acc[q] = t; }   // rows [r0 + r1, r0 + r1, r2 + r3, r2 + r3]
        }
    }
    if (e == s) {                                            // features.is_empty() -> zeros (ex.rs:357-359)
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = T(0);
        st = POLS_GROUP_EMPTY;
    } else if (!chol_solve<T, KT>(acc, (T)a.alpha, beta, (T)a.pivot_tol)) {
        st = POLS_GROUP_FALLBACK;                            // K6 re-solves this group
    }
    }
    if (live && sub == 0) {
        if (a.status) a.status[g] = st;
        if (st == POLS_GROUP_FALLBACK && a.fb_flag) *a.fb_flag = a.epoch;
    }
    if (live && a.coef && sub < KT) {
        T bv = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) bv = (sub == j) ? beta[j] : bv;
        static_cast<T *>(a.coef)[g * KT + sub] = bv;
    }
    if (a.pred || a.resid) {
#pragma unroll
        for (int rc = 0; rc < K1T_RC; ++rc) {
            const int64_t c = (int64_t)rc * K1T_SUB + sub;
            if (c < nch) predict_store<T, KT, HAS_W, false>(a, res[rc], beta, base + c * VEC, s, e);
        }
    }
}

template <typename T, int KT, bool HAS_W, int K1T_SUB, int K1T_RC>
static int k1t_launch(pols_ctx *ctx, const K1Args &a) {
    char name[96];
    std::snprintf(name, sizeof(name), "k1t_gram_chol_%s_k%d%s_sub%d_rc%d", sizeof(T) == 4 ? "f32" : "f64", KT, HAS_W ? "_w" : "", K1T_SUB, K1T_RC);
    const int64_t per_block = 4 * (64 / K1T_SUB);
    const int64_t blocks = (a.n_groups + per_block - 1) / per_block;
    if (blocks > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    ctx->last_kernel = name;
    K1Args aa = a;
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((k1t_kernel<T, KT, HAS_W, K1T_SUB, K1T_RC>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ev0, ev1, 0, aa);
    else
        hipLaunchKernelGGL((k1t_kernel<T, KT, HAS_W, K1T_SUB, K1T_RC>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, aa);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
#endif

#ifndef K1_NULLS_TU
// K1p: groups of up to a few hundred rows, PERSISTENT waves, 64 / SUB groups per wave, the next groups' rows in flight while these
// are reduced and solved.
//  * Timeline of the one-shot wave-per-group kernel on 500 000 groups of 130..252 rows: 14k cycles per group, ~1 000 VALU
//    instructions of 4 issue cycles each, four waves per SIMD -- the SIMD's issue slots are what is full, not the memory pipe
//    (4.45 TB/s).  A quarter of those instructions are the K x K Cholesky every lane repeats on wave-uniform values, another
//    fifth the wave reduction.  With SUB = 32 (16) lanes per group the same instruction stream reduces and solves 2 (4) groups.
//  * Fewer, fatter waves cannot hide their own load latency, so each wave walks the groups w, w + n_waves, ... and, as soon as it
//    has moved its rows from the LDS staging area into registers, issues the `global_load_lds` DMA of its NEXT rows (and of the
//    offsets after those) into the same area -- no VGPRs held by data in flight, nothing for the compiler's s_waitcnt
//    bookkeeping to trip over: no loads inside the loop's control flow (they would force vmcnt(0) at every join), no LDS reads
//    between the DMA issue and the loop top (the compiler waits for outstanding LDS DMA before any LDS read), outputs stored
//    one iteration late (right behind the DMA in the queue, so that the loop-top wait never sees a young store).
//  * Loads are unconditional and branch-free: lanes without a chunk re-read chunk 0, rows outside [s, e) -- ragged heads / tails,
//    a neighbour's rows -- are zeroed in registers, so ragged frames take the same code.  No barriers: waves never meet.
template <typename T, int KT, bool HAS_W, int SUB, int RC>
__device__ __forceinline__ void k1p_issue(const K1Args &a, int64_t s, int64_t e, typename Vec16<T>::type *stage, int sub) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCOL = KT + 1 + (HAS_W ? 1 : 0);
    const int ku = a.k_user;
    const int64_t base = s - (s % VEC);
    const int64_t nch = (e - base + VEC - 1) / VEC;
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const int64_t c = (int64_t)rc * SUB + sub;
        int64_t rl = base + (c < nch ? c : 0) * VEC;         // lanes without a chunk re-read chunk 0 (masking them off with EXEC measured
                                                             // no faster at 32 lanes per group, 10 % slower at 64)
        if (rl > a.n_rows - VEC) rl = a.n_rows - VEC;        // (the one chunk that crosses the end of the columns: see skip_group)
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            if (j < ku || j >= KT) {                         // slots [ku, KT) are the ones column: nothing to load
                const T *src = static_cast<const T *>(j < ku ? a.x[j] : (j == KT ? a.y : a.w));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + rl),
                                                 (__attribute__((address_space(3))) void *)(stage + (rc * NCOL + j) * 64), 16, 0, 0);
            }
        }
    }
}

// offs[g], offs[g + 1] of every lane's group, DMA'd like the rows (16 bytes per lane into the wave's offsets slot)
__device__ __forceinline__ void k1p_issue_offsets(const K1Args &a, int64_t g, longlong2 *slot) {
    const int64_t gi = g < a.n_groups ? g : a.n_groups - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.offs + gi),
                                     (__attribute__((address_space(3))) void *)slot, 16, 0, 0);
}

// what a group leaves behind for the NEXT iteration to store
template <typename T, int RC>
struct K1pPending {
    using V = typename Vec16<T>::type;
    V p[RC], r[RC];
    T bv;
    int st;
    int64_t g, s, e;
    bool have;
};

template <typename T, int KT, int SUB, int RC>
__device__ __forceinline__ void k1p_flush(const K1Args &a, const K1pPending<T, RC> &pd, int sub) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    if (!pd.have) return;
    if (sub == 0) {
        if (a.status) a.status[pd.g] = pd.st;
        if (pd.st == POLS_GROUP_FALLBACK && a.fb_flag) *a.fb_flag = a.epoch;
    }
    if (a.coef && sub < KT) static_cast<T *>(a.coef)[pd.g * KT + sub] = pd.bv;
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    if (!pred && !resid) return;
    const int64_t base = pd.s - (pd.s % VEC);
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const int64_t row0 = base + ((int64_t)rc * SUB + sub) * VEC;
        if (row0 >= pd.s && row0 + VEC <= pd.e) {
            if (pred) store_stream(reinterpret_cast<V *>(pred + row0), pd.p[rc]);
            if (resid) store_stream(reinterpret_cast<V *>(resid + row0), pd.r[rc]);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t rr = row0 + v;
                if (rr >= pd.s && rr < pd.e) {
                    if (pred) pred[rr] = vget<T>(pd.p[rc], v);
                    if (resid) resid[rr] = vget<T>(pd.r[rc], v);
                }
            }
        }
    }
}

// rows in registers (rows outside [s, e) zeroed) -> Gram, team reduction, Cholesky, predictions; everything is left in `pd`
template <typename T, int KT, bool HAS_W, int SUB, int RC>
__device__ __forceinline__ void k1p_solve(const K1Args &a, int64_t g, int64_t s, int64_t e, bool live, Chunk<T, KT, HAS_W> (&res)[RC],
                                          K1pPending<T, RC> &pd, int sub, unsigned long long *tl = nullptr) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
    constexpr int NACC = NZ * (NZ + 1) / 2;
    T acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = T(0);
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) gram_accumulate<T, KT, HAS_W>(acc, res[rc]);
    if (tl) tl[0] = __builtin_amdgcn_s_memtime();
    k1p_team_allreduce<T, NACC, SUB>(acc);
    if (tl) tl[1] = __builtin_amdgcn_s_memtime();
    T beta[KT];
    int st = POLS_GROUP_OK;
    if (e == s) {                                            // features.is_empty() -> zeros (ex.rs:357-359)
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = T(0);
        st = POLS_GROUP_EMPTY;
    } else if (!chol_solve<T, KT>(acc, (T)a.alpha, beta, (T)a.pivot_tol)) {
        st = POLS_GROUP_FALLBACK;                            // K6 re-solves this group
    }
    T bv = T(0);
#pragma unroll
    for (int j = 0; j < KT; ++j) bv = (sub == j) ? beta[j] : bv;
    pd.bv = bv; pd.st = st; pd.g = g; pd.s = s; pd.e = e; pd.have = live;
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            T pr = T(0);
#pragma unroll
            for (int j = 0; j < KT; ++j) pr = fma(vget<T>(res[rc].x[j], v), beta[j], pr);   // make_predictions on the FIT features (ex.rs:398-405)
            if constexpr (HAS_W) pr *= T(1) / vget<T>(res[rc].sw, v);                         // predictions *= 1/sqrt_w (ls.py:234-235)
            vset<T>(pd.p[rc], v, pr);
            vset<T>(pd.r[rc], v, vget<T>(res[rc].y, v) - pr);                                 // ORIGINAL target - predictions (ls.py:239)
        }
    }
}

template <typename T, int KT, bool HAS_W, int SUB, int RC, bool TL = false>
__global__ void __launch_bounds__(256) k1p_kernel(const K1Args a) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCOL = KT + 1 + (HAS_W ? 1 : 0);
    constexpr int GPW = 64 / SUB;                            // groups per wave
    __shared__ __attribute__((aligned(16))) V stage_s[4][RC * NCOL * 64];
    __shared__ __attribute__((aligned(16))) longlong2 offs_s[4][64];
    const int lane = threadIdx.x & 63, sub = lane & (SUB - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    V *stage = stage_s[wave];
    longlong2 *oslot = offs_s[wave];
    const int64_t step = (int64_t)gridDim.x * 4 * GPW;       // groups all waves take per round
    const int64_t G = a.n_groups;
    const int ku = a.k_user;
    int64_t g = ((int64_t)blockIdx.x * 4 + wave) * GPW + lane / SUB;   // this lane's group
    K1pPending<T, RC> pd;
    pd.have = false;
    int64_t s = 0, e = 0;
    if (g < G) { s = a.offs[g]; e = a.offs[g + 1]; }
    unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0}, tl[2] = {0, 0};   // TL: cycles per phase, summed over this wave's iterations
    k1p_issue<T, KT, HAS_W, SUB, RC>(a, s, e, stage, sub);
    k1p_issue_offsets(a, g + step, oslot);
    const int64_t g_wave0 = ((int64_t)blockIdx.x * 4 + wave) * GPW;   // wave-uniform loop control
#pragma unroll 1
    for (int64_t gw = g_wave0; gw < G; gw += step, g += step) {
        const int64_t base = s - (s % VEC);
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if constexpr (TL) t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // these groups' rows have landed; nothing younger than a whole iteration is
                                                             // outstanding (the previous outputs were stored right behind the DMA)
        if constexpr (TL) t1 = __builtin_amdgcn_s_memtime();
        Chunk<T, KT, HAS_W> res[RC];
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
#pragma unroll
            for (int j = 0; j < KT; ++j) res[rc].x[j] = j < ku ? stage[(rc * NCOL + j) * 64 + lane] : vsplat<T>(T(1));
            res[rc].y = stage[(rc * NCOL + KT) * 64 + lane];
            if constexpr (HAS_W) res[rc].sw = stage[(rc * NCOL + KT + 1) * 64 + lane];
        }
        const longlong2 on = oslot[lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // ... and are in registers: the staging area is free again
        if constexpr (TL) t2 = __builtin_amdgcn_s_memtime();
        const bool live_n = g + step < G;
        const int64_t sn = live_n ? on.x : 0, en = live_n ? on.y : 0;
        if (gw + step < G) {
            k1p_issue<T, KT, HAS_W, SUB, RC>(a, sn, en, stage, sub);
            k1p_issue_offsets(a, g + 2 * step, oslot);
        }
        unsigned long long t2b = 0;
        if constexpr (TL) t2b = __builtin_amdgcn_s_memtime();
        k1p_flush<T, KT, SUB, RC>(a, pd, sub);               // the previous groups' outputs, behind the DMA in the queue
        if constexpr (TL) t3 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t row0 = base + ((int64_t)rc * SUB + sub) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {                  // ragged heads / tails, lanes without a chunk: zero rows
                const bool in = (row0 + v >= s) && (row0 + v < e);
#pragma unroll
                for (int j = 0; j < KT; ++j) vset<T>(res[rc].x[j], v, in ? vget<T>(res[rc].x[j], v) : T(0));
                vset<T>(res[rc].y, v, in ? vget<T>(res[rc].y, v) : T(0));
                if constexpr (HAS_W) vset<T>(res[rc].sw, v, in ? vget<T>(res[rc].sw, v) : T(1));
            }
            if constexpr (HAS_W) load_chunk<T, KT, HAS_W, true, false, true>(a, row0, s, e, res[rc]);   // the sqrt(w) scaling only
        }
        k1p_solve<T, KT, HAS_W, SUB, RC>(a, g, s, e, g < G && g != a.skip_group, res, pd, sub, TL ? tl : nullptr);
        if constexpr (TL) {
            t4 = __builtin_amdgcn_s_memtime();
            tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t2b - t2; tsum[3] += t3 - t2b; tsum[4] += tl[1] - t3; tsum[5] += t4 - tl[1];
            tsum[6] += 1;
        }
        s = sn; e = en;
    }
    k1p_flush<T, KT, SUB, RC>(a, pd, sub);
    if constexpr (TL) {
        if (a.dbg && lane == 0) {
            unsigned long long *d = a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 8;
            unsigned long long c = 0;
            d[0] = 0;
            for (int i = 0; i < 6; ++i) { c += tsum[i]; d[i + 1] = c; }
            d[7] = tsum[6];
        }
    }
    // The group whose last chunk crosses the end of the columns (n_rows not a multiple of the vector width) cannot be DMA'd in
    // 16-byte pieces: one wave takes it here, after the loop, with guarded scalar loads (loads inside the loop's control flow
    // would make every iteration wait for the DMA in flight).
    if (a.skip_group >= 0 && blockIdx.x == 0 && wave == 0) {
        const int64_t gs = a.skip_group;
        const bool mine = lane < SUB;                        // team 0 takes it; the other teams compute on zeros and store nothing
        const int64_t s1 = a.offs[gs], e1 = mine ? a.offs[gs + 1] : s1;
        const int64_t base = s1 - (s1 % VEC);
        Chunk<T, KT, HAS_W> res[RC];
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t row0 = base + ((int64_t)rc * SUB + sub) * VEC;
            const bool has = row0 < e1;
            load_chunk<T, KT, HAS_W, false>(a, has ? row0 : base, s1, has ? e1 : s1, res[rc]);   // [s, s): every row outside -> zeros
        }
        k1p_solve<T, KT, HAS_W, SUB, RC>(a, gs, s1, e1, mine, res, pd, sub);
        k1p_flush<T, KT, SUB, RC>(a, pd, sub);
    }
}

template <typename T, int KT, bool HAS_W, int SUB, int RC>
static int k1p_launch(pols_ctx *ctx, const K1Args &a) {
    char name[96];
    std::snprintf(name, sizeof(name), "k1p_gram_chol_persistent_%s_k%d%s_sub%d_rc%d", sizeof(T) == 4 ? "f32" : "f64", KT, HAS_W ? "_w" : "", SUB, RC);
    static int occ = 0;                                      // resident blocks per CU of this variant (registers and LDS decide)
    if (occ == 0) {
        int o = 0;
        POLS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k1p_kernel<T, KT, HAS_W, SUB, RC>, 256, 0));
        occ = o > 0 ? o : 1;
    }
    constexpr int GPW = 64 / SUB;
    const int64_t want = (a.n_groups + 4 * GPW - 1) / (4 * GPW);
    const int64_t blocks = std::min<int64_t>(want, (int64_t)ctx->num_cus * occ);
    ctx->last_kernel = name;
    K1Args aa = a;
    aa.skip_group = (a.n_rows % Vec16<T>::N == 0) ? -1 : ctx->offs_tail_group;   // the chunk grid crosses the end of the columns
    if constexpr (KT == 8 && !HAS_W) {
        if (ctx->opt.timeline) {                             // phase cycles summed per persistent wave (debug): 7 "stamps" = 6 phases
            void *d = nullptr;
            int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)blocks * 4, &d);
            if (rc) return rc;
            aa.dbg = static_cast<unsigned long long *>(d);
            hipLaunchKernelGGL((k1p_kernel<T, KT, HAS_W, SUB, RC, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, aa);
            POLS_HIP(hipGetLastError());
            std::fprintf(stderr, "[timeline] per persistent wave (%lld waves, ~%.1f iterations each): wait | lds->regs | issue DMA | flush stores | zero+gram+reduce | solve+predict\n",
                         (long long)blocks * 4, (double)a.n_groups / ((double)blocks * 4 * GPW));
            return report_timeline(ctx, aa.dbg, blocks * 4, 7, name);
        }
    }
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((k1p_kernel<T, KT, HAS_W, SUB, RC>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ev0, ev1, 0, aa);
    else
        hipLaunchKernelGGL((k1p_kernel<T, KT, HAS_W, SUB, RC>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, aa);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
#endif

template <typename T, int KT, bool HAS_W, int TEAM, int RC, bool FAST, int NPASS = 1, bool NULLS = false>
static int k1_launch_fast(pols_ctx *ctx, const K1Args &a) {
    char name[96];
    char passes[8] = "";
    if (NPASS > 1) std::snprintf(passes, sizeof(passes), "_p%d", NPASS);
    std::snprintf(name, sizeof(name), "k1_gram_chol_%s_k%d%s_team%d_rc%d%s%s%s", sizeof(T) == 4 ? "f32" : "f64", KT,
                  HAS_W ? "_w" : "", TEAM, RC, FAST ? "_fast" : "", passes, NULLS ? "_nulls" : "");
    // (one team per workgroup -- a finished wave's slot refilled at once instead of waiting for its block-mates -- measured no
    // different: 74.4 vs 74.0 us on configs[1])
    // POLS_K1_WG=2|4: 512- / 1 024-thread workgroups of the 8-column team kernels (see k1_kernel_wg)
    constexpr bool WG_OK = FAST && !NULLS && KT == 8 && ((sizeof(T) == 4 && TEAM == 256 && RC == 1 && (NPASS == 2 || NPASS == 3)) ||
                                                          (sizeof(T) == 8 && TEAM == 128 && RC == 4 && NPASS == 2));
    const int wg = WG_OK && (ctx->opt.k1_wg == 2 || ctx->opt.k1_wg == 4) && !ctx->opt.timeline ? ctx->opt.k1_wg : 1;
    const int block_threads = 256 * wg;
    const int64_t teams_per_block = block_threads / TEAM;
    int64_t blocks = (a.n_groups + teams_per_block - 1) / teams_per_block;
    if (blocks > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    K1Args aa = a;
    aa.xcd_chunk = 0;
    if (ctx->opt.k1_xcd > 0 && blocks >= 64) { aa.xcd_chunk = (blocks + 7) / 8; blocks = aa.xcd_chunk * 8; }
    const bool timeline = ctx->opt.timeline;
    if (timeline) {
        void *d = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_groups, &d);   // slot 3 holds the fix-up work area
        if (rc) return rc;
        aa.dbg = static_cast<unsigned long long *>(d);
    }
    hipEvent_t ev0, ev1;
    constexpr bool OCC4 = sizeof(T) == 4 && TEAM == 64 && RC == 1 && !FAST && !NULLS && NPASS == 1;   // (the general form: single-pass only, see GENERAL)
    // The GENERAL (chunk-by-chunk, streamed-overflow) form only exists for the single-pass kernels: a multi-pass kernel is launched
    // only when every row stays resident (its callers check), and a ragged resident frame takes the branch-free EDGE form -- so the
    // general form of a multi-pass kernel would be code nothing launches (676 instantiations, a quarter of the library).  Frames of
    // fewer than VEC rows never get here (the dispatcher hands them to the fix-up solvers).
    constexpr bool GENERAL = FAST || NPASS == 1;
    void (*kern)(const K1Args) = k1_kernel<T, KT, HAS_W, TEAM, RC, GENERAL ? FAST : true, NPASS, NULLS, false, !GENERAL>;
    if constexpr (!GENERAL)
        std::snprintf(name, sizeof(name), "k1_gram_chol_%s_k%d%s_team%d_rc%d_edge%s%s", sizeof(T) == 4 ? "f32" : "f64", KT, HAS_W ? "_w" : "", TEAM, RC,
                      passes, NULLS ? "_nulls" : "");
    // the f32 wave-per-group FAST kernel (BASELINE configs[1]) reads its columns with `nt` (streaming) loads: every line is used
    // once, so it should not compete for L2 with lines that are -- 73.2-73.3 against 74.3-75.0 us per 400 MB launch
    // (POLS_K1_NT_LOADS=0 selects the plain-load build of the same kernel)
    // (the f64 two-wave kernel of cfg3 as well: 169-173 against 173-174 us per call; the f32 256-thread team from 8 columns: 71.9 vs
    // 72.4 us at 8 -- but 57.8 vs 50.4 us at 6 columns, where the plain loads win by far)
    // (round 6: `nt` loads in the null-policy build of the f32 256-thread team measured no different -- "drop" on a null-free frame 80.8 vs 80.4 us,
    // 5 % null targets 85.5 vs 86.0, profiles/r06_bench_nulls.txt -- so that build keeps its plain loads)
    constexpr bool HAS_NT = ((sizeof(T) == 4 && TEAM == 64 && RC == 4 && NPASS == 1) || (sizeof(T) == 4 && TEAM == 256 && RC == 1 && KT >= 8 && KT <= 10) || (sizeof(T) == 8 && TEAM == 128 && RC == 4 && NPASS == 2 && KT <= 8)) &&
                            FAST && !NULLS;
    if constexpr (HAS_NT) {
        if (ctx->opt.k1_nt_loads != 0) { kern = k1_kernel<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS, true>; std::strcat(name, "_nt"); }
    }
    if constexpr (WG_OK) {
        constexpr bool WNT = HAS_NT;                          // (the loads of the shipped build of this shape)
        if (wg == 2) kern = k1_kernel_wg<T, KT, HAS_W, TEAM, RC, FAST, NPASS, WNT, 512>;
        if (wg == 4) kern = k1_kernel_wg<T, KT, HAS_W, TEAM, RC, FAST, NPASS, WNT, 1024>;
        if (wg > 1) { char sfx[16]; std::snprintf(sfx, sizeof(sfx), "_wg%d", wg); std::strcat(name, sfx); }
    }
    if constexpr (OCC4) {
        if (!ctx->opt.k1_noocc4) kern = k1_kernel_occ4<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS>;
    }
    if constexpr (NULLS && sizeof(T) == 4 && TEAM == 64 && RC == 4 && KT <= 8 && !HAS_W)
        kern = k1_kernel_occ2<T, KT, HAS_W, TEAM, RC, FAST, NPASS, NULLS>;
    if constexpr (!FAST && GENERAL) {
        // ragged frames whose groups all stay resident: the branch-free EDGE form of the FAST kernel instead of the general code
        // (POLS_K1_NOEDGE=1 goes back)
        constexpr int VEC = Vec16<T>::N;
        const int64_t largest = a.class_max_rows > 0 ? a.class_max_rows : ctx->offs_max_rows;   // (of this launch's size class)
        const bool resident = largest + (ctx->offs_aligned[VEC == 4 ? 1 : 0] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (resident && a.n_rows >= VEC && !ctx->opt.k1_noedge && !ctx->opt.timeline) {
            kern = k1_kernel<T, KT, HAS_W, TEAM, RC, true, NPASS, NULLS, false, true>;
            std::snprintf(name, sizeof(name), "k1_gram_chol_%s_k%d%s_team%d_rc%d_edge%s%s", sizeof(T) == 4 ? "f32" : "f64", KT, HAS_W ? "_w" : "", TEAM, RC,
                          passes, NULLS ? "_nulls" : "");
        }
    }
    ctx->last_kernel = name;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(block_threads), 0, ctx->stream, ev0, ev1, 0, aa);
    else
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(block_threads), 0, ctx->stream, aa);
    POLS_HIP(hipGetLastError());
    if (timeline) return report_timeline(ctx, aa.dbg, a.n_groups, 6, name);
    return POLS_OK;
}

template <typename T, int KT, bool HAS_W, int TEAM, int RC>
static int k1_launch_variant(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    constexpr int VEC = Vec16<T>::N;
#ifdef K1_NULLS_TU
    // the null-policy family: the row masks live next to the resident rows; single-pass Gram only.  FAST as below: every load of
    // every resident chunk is in flight before the masks are built (POLS_K1_NOFAST=1: the general code)
    const bool fastn = ctx->offs_aligned[VEC == 4 ? 1 : 0] && max_rows <= (int64_t)RC * TEAM * VEC && !ctx->opt.k1_nofast;
    if constexpr (KT >= 9) {
        // 9-10 columns (8 features + intercept under a null policy): three masked passes, like the plain kernels of these widths
        const bool resident = max_rows + (ctx->offs_aligned[VEC == 4 ? 1 : 0] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (resident && ctx->opt.k1_passes != 1)
            return fastn ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 3, true>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 3, true>(ctx, a);
    }
    if constexpr (sizeof(T) == 4 && KT >= 7 && KT < 9 && TEAM == 64 && RC == 4) {
        // f32 wave kernel with 16 resident rows: with weights, or ragged, the single masked pass needs 267-287 registers (one wave per
        // SIMD); two passes then
        const bool resident = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (resident && (HAS_W || !fastn) && ctx->opt.k1_passes != 1)
            return fastn ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2, true>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2, true>(ctx, a);
    }
    if constexpr (sizeof(T) == 4 && KT >= 6 && KT < 9 && TEAM == 256) {
        // f32 team of 256 (the headline shape under a null policy), RAGGED frames: two masked passes (97 instead of 107 registers) -- 10 000
        // groups of 900..1 020 rows under "drop": 78.4 against 81.6 us.  Aligned frames keep the single pass: two passes measured slower
        // there (5 % null targets 87.5 vs 83.8 us, null-free 80.1 vs 78.8, "zero" 75.5 vs 74.6).  POLS_K1_PASSES=1 goes back.
        const bool resident = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (resident && !fastn && ctx->opt.k1_passes != 1) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2, true>(ctx, a);
    }
    if constexpr (sizeof(T) == 8 && KT >= 6 && KT < 9) {
        // f64, 6+ columns: the single-pass kernel needs 275-400 registers (one wave per SIMD: 300 against 146 us for the plain kernel on
        // 10 000 x 1 000 x 8); two masked passes over the resident rows like the plain f64 kernels.  POLS_K1_PASSES=1 goes back.
        const bool resident = max_rows + (ctx->offs_aligned[0] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (resident && ctx->opt.k1_passes != 1)
            return fastn ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2, true>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2, true>(ctx, a);
    }
    return fastn ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 1, true>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 1, true>(ctx, a);
#else
    // FAST needs every group aligned to the vector width and resident; the offsets scan in upload_offsets() knows
    const bool fast = ctx->offs_aligned[VEC == 4 ? 1 : 0] && max_rows <= (int64_t)RC * TEAM * VEC &&
                      !ctx->opt.k1_nofast;
    // (f32 wave-per-group: the multi-pass form was tried -- 216 -> 187 VGPRs with three passes, still two waves per SIMD because the
    // 16 resident rows alone are 144 registers -- and dropped.)
    if constexpr (sizeof(T) == 8 && TEAM == 256 && (RC == 2 || RC == 4) && KT >= 6) {
        // f64, 6+ columns: the full accumulator set costs a workgroup per CU; POLS_K1_PASSES=1|2|3 overrides
        // (four chunks per lane at 9-10 columns: 160-176 resident registers -- three passes keep the accumulators at 19-22 doubles)
        const int npass = ctx->opt.k1_passes ? ctx->opt.k1_passes : ((KT >= 10 || (RC == 4 && KT >= 9)) ? 3 : 2);
        if (fast && npass == 2) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2>(ctx, a);
        if (fast && npass == 3) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 3>(ctx, a);
        if constexpr (KT >= 9 || RC == 4) {                  // ragged frames whose rows all stay resident: the same passes, general loads
            const bool resident = max_rows + (ctx->offs_aligned[0] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC && !fast;
            if (resident && npass == 2) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2>(ctx, a);
            if (resident && npass == 3) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 3>(ctx, a);
        }
    }
    if constexpr (sizeof(T) == 4 && TEAM == 256 && (RC == 1 || RC == 2 || RC == 4) && KT >= 6 && KT <= 10) {
        // f32, one chunk per lane of a 256-thread team: two passes at 6-8 columns, three at 9-10 (POLS_K1_PASSES=1|2|3 overrides)
        const int npass = ctx->opt.k1_passes ? ctx->opt.k1_passes : (KT >= 9 ? 3 : 2);
        if (fast && npass == 2) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2>(ctx, a);
        if (fast && npass == 3) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 3>(ctx, a);
        const bool resident = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        const int npass_r = ctx->opt.k1_passes ? ctx->opt.k1_passes : ((KT >= 8 && RC == 1) || (KT >= 9 && RC == 4) ? 3 : 2);   // ragged, 8 columns, one chunk: 69.8 (three) vs 71.4 us (two)
        if (!fast && resident && npass_r == 2) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2>(ctx, a);
        if (!fast && resident && npass_r == 3) return k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 3>(ctx, a);
    }
    if constexpr (sizeof(T) == 4 && TEAM == 64 && (RC == 1 || RC == 2) && KT >= 7 && KT <= 8) {
        // f32 wave kernel with 4-8 resident rows per lane, 7-8 columns: the two-pass Gram + row-resident Cholesky with two chunks per
        // lane (100..300 rows: 108.9 -> 104.4 us ragged, 93.0 -> 90.0 aligned) and for ragged one-chunk frames (130..252 rows: 748 ->
        // 732 us); aligned one-chunk frames keep the single pass (686 vs 694 us).  POLS_K1_PASSES=1|2 overrides.
        const bool resident = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        const int npass = ctx->opt.k1_passes ? ctx->opt.k1_passes : ((RC == 2 || !fast) ? 2 : 1);
        if (resident && npass == 2)
            return fast ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2>(ctx, a);
    }
    if constexpr (TEAM == 64 && KT >= 9) {
        // 9-10 columns (8 features + intercept: the smoke() shape): 55-66 accumulators next to the resident rows do not fit the
        // register file; the multi-pass Gram (a third of them live at a time, totals in LDS, row-cooperative Cholesky there)
        // does.  POLS_K1_PASSES=1 goes back to the single pass.
        const int npass = ctx->opt.k1_passes ? ctx->opt.k1_passes : 3;
        const bool resident = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= (int64_t)RC * TEAM * VEC;
        if (npass == 3 && resident) return fast ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 3>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 3>(ctx, a);
        if (npass == 2 && resident) return fast ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true, 2>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false, 2>(ctx, a);
    }
    return fast ? k1_launch_fast<T, KT, HAS_W, TEAM, RC, true>(ctx, a) : k1_launch_fast<T, KT, HAS_W, TEAM, RC, false>(ctx, a);
#endif
}

// Variant choice: smallest team whose registers hold the largest group (so X is read once); groups
// larger than the biggest variant stream their overflow rows twice (Gram pass + prediction pass).
template <typename T, int KT, bool HAS_W>
static int k1_launch_kw(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    constexpr int VEC = Vec16<T>::N;
#ifndef K1_NULLS_TU
    // (f64: measured no better than K1t -- 870 vs 828 us on 500 000 groups of 40..120 rows with 32 lanes per group, 559 vs 450 us on
    // 12..40 rows with 16: 45 f64 accumulators + 8 resident rows leave one or two waves per SIMD -- so not instantiated)
    if constexpr (sizeof(T) == 4) {
        // a few hundred rows per group and enough groups to keep every persistent wave busy for several rounds: K1p
        const int64_t need = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1);
        const bool many = a.n_groups >= (int64_t)ctx->num_cus * 16 * 4;
        const int ps = ctx->opt.k1_persist_sub;              // POLS_K1_PERSIST_SUB=64|32|16: A/B the team width
        // (since the one-shot K1t got the two-pass Gram + row-share Cholesky its 32-lane form does these frames as fast -- 797 vs 814 us
        // on 500 000 groups of 130..252 rows -- without a staging area or a minimum number of groups: K1p runs on request only)
        (void)many;
        if (ctx->opt.k1_persist > 0 && a.n_rows >= VEC && a.n_groups > 0) {
            // 500 000 groups of 40..120 rows: 383 us (16 lanes per group) against 495 us for K1t; 130..252 rows: 815 us (32 lanes)
            // against 850 us for the one-shot wave.  Up to 64 rows K1t's four groups per wave win (210 vs 320 us on 12..40 rows),
            // beyond 256 rows the staging area limits a CU to eight waves and the one-shot wave kernel wins (110 vs 125 us).
            // (two 4-wave blocks per CU need 2 x 4 x (2 * columns + 1) KiB of LDS <= 160 KiB: up to 9 staged columns)
            const bool forced = ctx->opt.k1_persist > 0;
            constexpr bool two_blocks = 2 * (KT + 1 + (HAS_W ? 1 : 0)) + 1 <= 20;
            // (16 lanes per group, up to 128 rows: 383-392 us on 500 000 groups of 40..120 rows against 495 us for K1t when this was
            // written; K1t has since got the two-pass Gram + row-share Cholesky -- 122 instead of 181 registers -- and does 378 us)
            if (need <= 16 * 2 * VEC && (ps == 16 || (ps == 0 && forced))) return k1p_launch<T, KT, HAS_W, 16, 2>(ctx, a);
            if (need <= 32 * 2 * VEC && (ps == 32 || (ps == 0 && (forced || (two_blocks && need > 16 * 2 * VEC))))) return k1p_launch<T, KT, HAS_W, 32, 2>(ctx, a);
            if (need <= 64 * 1 * VEC && ps == 64) return k1p_launch<T, KT, HAS_W, 64, 1>(ctx, a);
            if (need <= 64 * 2 * VEC && (ps == 64 || (ps == 0 && forced))) return k1p_launch<T, KT, HAS_W, 64, 2>(ctx, a);
        }
    }
#endif
#ifndef K1_NULLS_TU
    if (!ctx->opt.k1_notiny && !ctx->opt.timeline && a.n_rows >= VEC) {
        const int64_t need = max_rows + (ctx->offs_aligned[VEC == 4 ? 1 : 0] ? 0 : VEC - 1);
        if constexpr (KT <= 8) {
            // EIGHT-LANE TEAMS (round 3), eight groups per wave: frames whose every group fits eight chunk slots (32 f32 / 16 f64 rows
            // on the chunk grid).  (Splitting a MIXED frame into a list of groups that fit and a list of the rest, two launches, measured
            // no faster -- 198 vs 191 us on 500 000 groups of 12..40 rows: neighbouring groups share cache lines, the two launches
            // fetched 1.43x the bytes of one, PMC FETCH_SIZE -- profiles/r03_*_bucketed_rejected.*)
            const int s8 = ctx->opt.k1t_sub8;                // -1 / 2: the rule below, 1: one chunk per lane only, 0: 16-lane teams
            if (need <= 8 * 1 * VEC && s8 != 0) return k1t_launch<T, KT, HAS_W, 8, 1>(ctx, a);
            // two chunks per lane (the 16 slots of a 16-lane team, twice the groups per wave): 500 000 groups of 12..40 rows, f32, 8 columns
            // 140 vs 160 us, 3 columns 48.5 vs 52.2, 6 columns 110 vs 112; f64 10..16 rows x 8 columns 204 vs 293 -- but 5 columns 95 vs
            // 88.5: the single-pass form (below 6 columns) with 4-5 columns keeps its 16-lane teams
            if (need <= 8 * 2 * VEC && s8 != 0 && s8 != 1 && (KT >= K1T_PASS_MIN_KT || KT <= 3 || s8 == 2)) return k1t_launch<T, KT, HAS_W, 8, 2>(ctx, a);
        }
        if constexpr (KT <= 8 && sizeof(T) == 8 && KT >= K1T_PASS_MIN_KT) {
            // f64, three chunks per lane of an eight-lane team (48 rows: a month or two of trading days per group), eight groups per wave
            // where the 16-lane team's 32 slots were at most 2/3 full
            const int s8 = ctx->opt.k1t_sub8;
            if (need > 16 * 1 * VEC && need <= 8 * 3 * VEC && s8 != 0 && s8 != 1) return k1t_launch<T, KT, HAS_W, 8, 3>(ctx, a);
        }
        if (need <= 16 * 1 * VEC) return k1t_launch<T, KT, HAS_W, 16, 1>(ctx, a);      // one chunk per lane: a third fewer registers
        if (need <= 16 * 2 * VEC) return k1t_launch<T, KT, HAS_W, 16, 2>(ctx, a);
        // four chunks per lane (128 f64 / 256 f32 rows): f64 with 6+ columns has the registers for it since the two-pass form;
        // f32 only on request (POLS_K1T_RC4=1: A/B against the one-chunk wave kernel)
        if (need <= 16 * 4 * VEC) {
            // (... up to 8 columns, 7 with weights: beyond, the four chunks tip the kernel into AGPRs -- 266-314 registers, one wave per SIMD)
            constexpr bool fits = KT <= (HAS_W ? 7 : 8);
            const bool want = ctx->opt.k1t_rc4 >= 0 ? ctx->opt.k1t_rc4 != 0 : (sizeof(T) == 8 && KT >= 6 && fits);
            if constexpr (fits || sizeof(T) == 4) { if (want) return k1t_launch<T, KT, HAS_W, 16, 4>(ctx, a); }
        }
        if constexpr (sizeof(T) == 4 && KT >= 6 && KT <= 8) {
            // (9-10 columns: one wave per group with the three-pass Gram is faster -- 76.7 vs 89.4 us on 50 000 x 200 x (8 + 1))
            // two groups per wave (32-lane teams, two chunks per lane: up to 256 rows) with the two-pass Gram and the row-share
            // Cholesky: ~115 registers where the single-pass form needed 181 (and lost to one wave per group)
            // RAGGED frames only: aligned ones keep the FAST wave kernels (50 000 x 200 x 8: 70.7 vs 73.3 us; x 6: 49.3 vs 54.4), whose
            // ragged form is what loses (500 000 groups of 130..252 rows: 907 us one wave per group, 814 us K1p, 771 us here)
            const bool ragged = !ctx->offs_aligned[1] || ctx->opt.k1_nofast;
            // ... and since the wave kernels' ragged form became branch-free (EDGE) they win again: 737 us.  On request only.
            if (ragged && need <= 32 * 2 * VEC && ctx->opt.k1t_sub32 > 0) return k1t_launch<T, KT, HAS_W, 32, 2>(ctx, a);
            // (three chunks per lane -- up to 384 rows -- measured no better than one wave per group: 109.7 vs 107.9 us on 50 000 groups
            // of 100..300 rows)
        }
        // (SUB = 32, two groups per wave up to 256 / 128 rows, measured SLOWER than one wave per group: 1 022 vs 910 us on 500 000
        // f32 groups of 130..252 rows, 1 815 vs 1 217 us on f64 groups of 40..120 -- the kernel template keeps the variant, nothing
        // launches it)
    }
#endif
    if constexpr (sizeof(T) == 4) {
        // up to 256 f32 rows (a year of trading days): one chunk per lane -- ~36 registers fewer, four waves per SIMD instead of three
        if (max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1) <= 64 * 1 * VEC && !ctx->opt.k1_norc1)
            return k1_launch_variant<T, KT, HAS_W, 64, 1>(ctx, a, max_rows);
    }
    if (max_rows <= 64 * 2 * VEC) return k1_launch_variant<T, KT, HAS_W, 64, 2>(ctx, a, max_rows);
    if constexpr (sizeof(T) == 4) {
        // wave-per-group with 16 rows per lane: no LDS, no barriers, one reduction + one solve per group and
        // twice the groups in flight per CU (2 waves/SIMD x 4 SIMDs = 8) -- POLS_K1_SHAPE=team forces 256 threads
        const bool want_wave = !ctx->opt.k1_shape_team;
        // (unaligned group starts: the chunk grid begins up to VEC - 1 rows before the group)
        // Ragged frames (what `.over(key)` delivers): groups up to 1/8 beyond the 1 024 resident rows stay with the wave kernel, their
        // overflow rows streamed twice -- 83.7 us against 101.2 us for the 256-thread team on 10 000 groups of 950..1 100 rows
        // (scripts/bench_ragged.py); unaligned group starts no longer exclude it either (900..1 020 rows: 88.3 -> 76.9 us).
        // The rule: every group within twice the resident rows (one wave streams its overflow serially) and at most 1/16 of the
        // frame's rows beyond them (each is read twice).
        const int64_t wave_cap = 64 * 4 * VEC;
        const int64_t need = max_rows + (ctx->offs_aligned[1] ? 0 : VEC - 1);
        // Frames whose groups fit one chunk per lane of a 256-thread team (513..1 024 rows, aligned or ragged): four resident rows per lane instead
        // of sixteen -- 68-105 registers, 4-7 waves per SIMD -- and, from 6 columns, the multi-pass Gram + row-resident Cholesky.
        // Interleaved A/B on 10 000 x 1 000 rows (scripts/ab_headline.py, wall clock per call): 2 columns 26.4 vs 29.9 us, 4: 37.7 vs
        // 43.7, 6: 50.4 vs 58.3 (6.35 TB/s), 7: 62.3 vs 65.6, 8: 71.9 vs 73.2, 9: 78.6 vs 81.1, 10: 85.8 vs 89.0.  POLS_K1_SHAPE=wave
        // goes back.  Ragged frames too (their edge chunks take 16-byte loads + rows zeroed in registers): 900..1 020 rows 78.2 -> 71.8 us.
        if (!ctx->opt.k1_shape_wave && need <= 256 * 1 * VEC)
            return k1_launch_variant<T, KT, HAS_W, 256, 1>(ctx, a, max_rows);
        // (the wave kernel with 16 rows per lane -- and its streamed overflow for frames a little beyond 1 024 rows -- is what
        // POLS_K1_SHAPE=wave still uses; by default the two-chunk team takes 1 025..2 048 rows: 950..1 100 rows
        // 88.2 -> 85.6 us)
        const bool wave_default = ctx->opt.k1_shape_wave;
        if (want_wave && wave_default && (need <= wave_cap || (need <= 2 * wave_cap && ctx->offs_wave_overflow * 16 <= a.n_rows)))
            return k1_launch_variant<T, KT, HAS_W, 64, 4>(ctx, a, max_rows);
        if (max_rows <= 256 * 1 * VEC) return k1_launch_variant<T, KT, HAS_W, 256, 1>(ctx, a, max_rows);
#ifndef K1_NULLS_TU
        // 2 049..4 096 rows (ten years of trading days per asset), up to 10 columns, round 5: FOUR chunks per lane of the 256-thread team --
        // 144-176 resident registers, two workgroups per CU -- instead of handing the frame to K2 / K1m (2.1-3.4 TB/s there)
        if (need > 256 * 2 * VEC && need <= 256 * 4 * VEC) return k1_launch_variant<T, KT, HAS_W, 256, 4>(ctx, a, max_rows);   // (up to 10 columns)
#endif
        return k1_launch_variant<T, KT, HAS_W, 256, 2>(ctx, a, max_rows);
    } else {
#ifndef K1_NULLS_TU
        // f64, 6+ columns, aligned groups of up to 1 024 rows: TWO waves per group with 8 rows per lane (218-240 VGPRs, two waves per
        // SIMD) keep four groups in flight per CU where the 256-thread team keeps three -- 153 vs 160 us on 10 000 x 1 000 x 8, 174 vs
        // 177 us with weights (cfg3).  POLS_K1_F64_TEAM=256 goes back; POLS_K1_PASSES=3 splits the Gram in three.
        // Ragged frames (group starts / sizes not multiples of 2 rows) take the same shape without the FAST specialisation as long as
        // every group -- plus the up to VEC - 1 rows its chunk grid starts early -- stays resident; groups of up to 512 rows use half
        // the resident chunks (10 000 groups of 900..1 020 rows: 165 -> 148 us; 50 000 groups of 400..500 rows: 126 -> 78 us).
        if constexpr (KT >= 6) {
            const bool al = ctx->offs_aligned[0] && !ctx->opt.k1_nofast;
            const int64_t need = max_rows + (ctx->offs_aligned[0] ? 0 : VEC - 1);
            // (10 columns: the two-wave team's 8 resident rows x 11 columns leave one wave per SIMD -- the 256-thread team with 4 rows
            // per lane and the three-pass Gram keeps its registers under 168)
            // (9 columns WITH weights: 8 rows x 11 values tip the two-wave team into AGPRs -- 274 registers, one wave per SIMD, 258 us
            // on 10 000 x 1 000 rows where the 256-thread team takes ~185)
            const bool team128 = (KT < 10 && !(KT == 9 && HAS_W)) || need <= 128 * 2 * VEC;
            if (!ctx->opt.k1_f64_team256 && team128 && need <= 128 * 4 * VEC && !ctx->opt.timeline) {
                const int pp = ctx->opt.k1_passes ? ctx->opt.k1_passes : (KT >= 10 ? 3 : 0);   // 10 columns: 66 f64 accumulators -> three passes
                // (round 6: groups of up to 512 rows at 10 columns -- and at 9 WITH weights, 170 VGPRs in two passes -- take three passes over TWO chunks per
                // lane: under 168 registers, three waves per SIMD; they used to run the four-chunk kernel half empty / sit two registers over the step)
                if constexpr (KT == 10 || (KT == 9 && HAS_W)) {
                    constexpr int P3 = (KT == 10 && HAS_W) ? 4 : 3;                             // (10 columns with weights: 171 VGPRs in three)
                    if (need <= 128 * 2 * VEC && (pp == 3 || pp == 0))
                        return al ? k1_launch_fast<T, KT, HAS_W, 128, 2, true, P3>(ctx, a) : k1_launch_fast<T, KT, HAS_W, 128, 2, false, P3>(ctx, a);
                }
                if (pp == 3 && al) return k1_launch_fast<T, KT, HAS_W, 128, 4, true, 3>(ctx, a);
                if (pp == 0 || pp == 2) {
                    if (need <= 128 * 2 * VEC)
                        return al ? k1_launch_fast<T, KT, HAS_W, 128, 2, true, 2>(ctx, a) : k1_launch_fast<T, KT, HAS_W, 128, 2, false, 2>(ctx, a);
                    return al ? k1_launch_fast<T, KT, HAS_W, 128, 4, true, 2>(ctx, a) : k1_launch_fast<T, KT, HAS_W, 128, 4, false, 2>(ctx, a);
                }
            }
        }
        // 1 025..2 048 f64 rows, up to 10 columns (round 5): four chunks per lane of the 256-thread team instead of K2
        {
            const int64_t need4 = max_rows + (ctx->offs_aligned[0] ? 0 : VEC - 1);
            if (need4 > 256 * 2 * VEC && need4 <= 256 * 4 * VEC) return k1_launch_variant<T, KT, HAS_W, 256, 4>(ctx, a, max_rows);   // (up to 10 columns)
        }
#endif
        return k1_launch_variant<T, KT, HAS_W, 256, 2>(ctx, a, max_rows);
    }
}

template <typename T, int KT>
static int k1_launch_kt(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    return a.w ? k1_launch_kw<T, KT, true>(ctx, a, max_rows) : k1_launch_kw<T, KT, false>(ctx, a, max_rows);
}

// 11-15 columns: only the multi-pass forms exist (91 accumulators at 12 features + target: three passes; 136 at 15: four), picked so
// that the resident rows of a lane stay at 4-8 x (k + 1) values.  11-12 columns: one wave up to 512 f32 / 256 f64 rows, two waves
// up to 1 024 / 512, the 256-thread team beyond.  13-15 columns: one chunk per lane (one wave, two waves, four) before two.
// Gram passes of the resident wide kernels on the teams that gain a workgroup by it (round 6): the accumulators a pass may hold are what three waves per
// SIMD (four at 16-18 f32 columns) leave next to the resident rows -- 4 (KT + 1) registers, 4 more with the sqrt(w) vector, ~8-10 more in the null-policy
// builds (measured per build: weights +7..8, masks +8..10).  W: with weights, NL: null-policy family.  At least 16 f32 / 10 f64 per pass.
template <typename T, int KT, bool NL>
constexpr int k1w_short_passes(bool W) {
    constexpr int NACC = (KT + 1) * (KT + 2) / 2;
    const int rows = 4 * (KT + 1) + (W ? 8 : 0) + (NL ? 20 : 0);
    if (sizeof(T) == 4) {
        int acc = KT <= 18 ? 102 - rows : 141 - rows;                 // (16-18 columns: four waves per SIMD; from 19: three; ~20-25 registers of everything else)
        if (acc > 60) acc = 60;
        if (acc < 16) acc = 16;
        return (NACC + acc - 1) / acc;
    }
    int acc = (142 - rows) / 2 - ((KT == 21 || KT == 23) ? 2 : 0);  // (doubles: two registers each; 21 / 23 columns sat a register or two over)
    if (acc > 36) acc = 36;
    if (acc < 10) acc = 10;
    return (NACC + acc - 1) / acc;
}

template <typename T, int KT>
static int k1_launch_wide_kt(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NACC = (KT + 1) * (KT + 2) / 2;
    // passes: three up to 12 columns, four up to 15; beyond, ~60 f32 / ~36 f64 accumulators live at a time (31 columns: 528 entries)
    // Round 6, f32 from 23 columns: the resident rows alone are 4 (KT + 1) registers, and with ~60 accumulators next to them the kernel needs 173-205
    // VGPRs -- two waves per SIMD where 22 columns (165) still run three: 4.4 TB/s at 20 columns, 3.05 at 24 (profiles/r05_bench_k16.txt).  More,
    // shorter passes keep it at 168: the accumulators a pass may hold are what three waves per SIMD leave next to the rows (the products per row
    // are the same however they are split; a pass more is one more reduce-and-barrier round).
    // (from 26 columns the solving wave also parks its rows in LDS while it solves: k1_body, PARK)
    // (16-18 columns the same way one step up: 40 / 36 / 32 accumulators per pass keep the kernel at 128 VGPRs, FOUR waves per SIMD: 4.8 -> 5.2 TB/s;
    // at 19-20 it took parking too and bought 1-2 %: not kept.  The 256-thread team, and the two-wave f32 team up to 25 columns: the other teams gain no
    // workgroup by it and keep NP0.  k1w_short_passes above: per build -- weights and the null-policy masks take their registers off the budget.)
    constexpr int NP0 = KT <= 12 ? 3 : (KT <= 15 ? 4 : (sizeof(T) == 4 ? (NACC + 59) / 60 : (NACC + 35) / 36));      // every other team
    const bool al = ctx->offs_aligned[VEC == 4 ? 1 : 0] && !ctx->opt.k1_nofast;
    const int64_t need = max_rows + (ctx->offs_aligned[VEC == 4 ? 1 : 0] ? 0 : VEC - 1);
#ifdef K1_NULLS_TU
    constexpr bool NL = true;     // the null-policy family: the same shapes with the masked Gram passes (11-15 columns)
#else
    constexpr bool NL = false;
#endif
#define K1W_GO(TEAM, RC)                                                                                                       \
    { constexpr bool SHORT = KT >= 16 && (((TEAM) == 256 || ((TEAM) == 128 && sizeof(T) == 4 && KT <= 25)) && (RC) == 1);            \
      constexpr int NPW = SHORT ? k1w_short_passes<T, KT, NL>(true) : NP0, NPN = SHORT ? k1w_short_passes<T, KT, NL>(false) : NP0;    \
    return a.w ? (al ? k1_launch_fast<T, KT, true, TEAM, RC, true, NPW, NL>(ctx, a) : k1_launch_fast<T, KT, true, TEAM, RC, false, NPW, NL>(ctx, a)) \
               : (al ? k1_launch_fast<T, KT, false, TEAM, RC, true, NPN, NL>(ctx, a) : k1_launch_fast<T, KT, false, TEAM, RC, false, NPN, NL>(ctx, a)); }
#ifndef K1_NULLS_TU
    if constexpr (KT <= 16) {
        // 11-16 columns, groups of at most 32 chunks (128 f32 / 64 f64 rows -- a quarter of daily data against a dozen factors): K1t's four groups
        // per wave, 16 packed Gram entries per pass and the row-cooperative Cholesky, instead of one wave per group with 16 of its 64 lanes
        // holding rows (64-row groups x 12 columns: 1.8 TB/s, profiles/r05_bench_rows_sweep.txt)
        if (!ctx->opt.k1_notiny && !ctx->opt.timeline && a.n_rows >= VEC) {
            if (need <= 16 * 1 * VEC) return a.w ? k1t_launch<T, KT, true, 16, 1>(ctx, a) : k1t_launch<T, KT, false, 16, 1>(ctx, a);
            if constexpr (KT <= 15 || sizeof(T) == 4) {     // (f64 at 16 columns: two chunks per lane are past 256 registers)
                if (need <= 16 * 2 * VEC) return a.w ? k1t_launch<T, KT, true, 16, 2>(ctx, a) : k1t_launch<T, KT, false, 16, 2>(ctx, a);
            }
        }
    }
#endif
    if constexpr (KT <= 12) {
        if constexpr (sizeof(T) == 4) {
            if (need <= 64 * 1 * VEC) { K1W_GO(64, 1); }
        }
        if (need <= 64 * 2 * VEC) { K1W_GO(64, 2); }
        if (need <= 128 * 2 * VEC) { K1W_GO(128, 2); }
    } else if constexpr (KT <= 15) {
        if (need <= 64 * 1 * VEC) { K1W_GO(64, 1); }
        if constexpr (sizeof(T) == 8) {
            if (need <= 64 * 2 * VEC) { K1W_GO(64, 2); }         // f64, up to 256 rows: one wave with two chunks per lane
        }
        if (need <= 128 * 1 * VEC) { K1W_GO(128, 1); }
        if (need <= 256 * 1 * VEC) { K1W_GO(256, 1); }
    } else {                                                 // 16-31 columns: one chunk per lane (4 f32 / 2 f64 rows x up to 32 values)
        if (need <= 64 * 1 * VEC) { K1W_GO(64, 1); }
        if (need <= 128 * 1 * VEC) { K1W_GO(128, 1); }
        if (need <= 256 * 1 * VEC) { K1W_GO(256, 1); }
    }
    if constexpr (KT <= 15) {
        if (need <= 256 * 2 * VEC) { K1W_GO(256, 2); }
    }
#undef K1W_GO
    return fail(POLS_ERR_UNSUPPORTED, "k1: %d columns with %lld-row groups do not stay resident", KT, (long long)max_rows);
}

// One translation unit instantiates the column counts [K1_PART_LO, K1_PART_HI] of one dtype (K1_PART_T) under the entry name
// K1_PART_FN -- the fully unrolled kernels of all 15 column counts in one unit took 5.5 minutes to compile; api.hip picks the unit.
#ifdef K1_PART_FN
int K1_PART_FN(pols_ctx *ctx, int kt, const K1Args &a, int64_t max_rows) {
    switch (kt) {
#if K1_PART_LO <= 1 && 1 <= K1_PART_HI
        case 1: return k1_launch_kt<K1_PART_T, 1>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 2 && 2 <= K1_PART_HI
        case 2: return k1_launch_kt<K1_PART_T, 2>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 3 && 3 <= K1_PART_HI
        case 3: return k1_launch_kt<K1_PART_T, 3>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 4 && 4 <= K1_PART_HI
        case 4: return k1_launch_kt<K1_PART_T, 4>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 5 && 5 <= K1_PART_HI
        case 5: return k1_launch_kt<K1_PART_T, 5>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 6 && 6 <= K1_PART_HI
        case 6: return k1_launch_kt<K1_PART_T, 6>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 7 && 7 <= K1_PART_HI
        case 7: return k1_launch_kt<K1_PART_T, 7>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 8 && 8 <= K1_PART_HI
        case 8: return k1_launch_kt<K1_PART_T, 8>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 9 && 9 <= K1_PART_HI
        case 9: return k1_launch_kt<K1_PART_T, 9>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 10 && 10 <= K1_PART_HI
        case 10: return k1_launch_kt<K1_PART_T, 10>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 11 && 11 <= K1_PART_HI
        case 11: return k1_launch_wide_kt<K1_PART_T, 11>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 12 && 12 <= K1_PART_HI
        case 12: return k1_launch_wide_kt<K1_PART_T, 12>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 13 && 13 <= K1_PART_HI
        case 13: return k1_launch_wide_kt<K1_PART_T, 13>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 14 && 14 <= K1_PART_HI
        case 14: return k1_launch_wide_kt<K1_PART_T, 14>(ctx, a, max_rows);
#endif
#if K1_PART_LO <= 15 && 15 <= K1_PART_HI
        case 15: return k1_launch_wide_kt<K1_PART_T, 15>(ctx, a, max_rows);
#endif
        default: return fail(POLS_ERR_UNSUPPORTED, "k1: %d columns are not in this unit", kt);
    }
}
#endif

}  // namespace pols
