// k2_kernel.inl -- body of K2 "gram_mfma_resident" (interface and reference citations: k2_resident.hpp).
//
// One workgroup of WAVES waves per group.  Every lane keeps RC chunks (16 bytes = VEC consecutive rows) of EVERY column in
// VGPRs -- all loads of a lane are issued before the first use, X is read from HBM exactly once -- and the workgroup then
//   gram    : forms X'X as ONE 16 x 16 tile on the matrix cores (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64, A = B).
//             The matrix cores want lane (c, q) to hold X[row 4t + q][column c]; the registers hold a row per lane.  Each wave
//             therefore transposes its own rows through a PRIVATE 8.25 KB LDS tile, 8 bytes per lane and column at a time
//             (one f64 row or two f32 rows: 16 ds_write_b64, then 16 conflict-free ds_read_b64, one per MFMA for f64 and one
//             per two for f32) -- no workgroup barrier in the whole Gram phase, only wave-local ordering.  Which rows share an
//             MFMA step is irrelevant to a sum over rows, so the f32 form pairs rows (r, r + 1) instead of shuffling.
//             With fewer than 16 columns the target rides in the tile (X'y for free); with exactly 16 (BASELINE configs[4])
//             X'y is accumulated on the VALU from the registers (YV).  f32 tiles are flushed into f64 accumulators after every
//             128-row stage, so an f32 Gram matrix carries f64-summation error.
//   reduce  : per-wave partial tiles -> LDS -> one f64 Gram matrix (fixed order: run-to-run identical).          [1 barrier]
//   solve   : wave 0, in f64, on the matrix in LDS: Cholesky (lane i owns row i of L), partial-pivot LU, Cholesky -> LU
//             fallback, or cyclic coordinate descent with the reference's coordinate order, alpha * n scaling, soft threshold,
//             active set and ||w - w_old|| < tol stop.                                                              [1 barrier]
//   predict : X . beta (+ residuals) from the resident rows, 16-byte streaming stores; 1/sqrt(w) un-scaling as K1.  [1 barrier]
// Bound: HBM, b n (k + 1) (+ b n weights) bytes in, b n out per group -- the elastic net no longer reads X twice.
#include "k1m_kernel.inl"   // Mfma16, Vec16 helpers
#include "k2_resident.hpp"

namespace pols {

constexpr int K2_SLOT_B = 528;                 // one column slot of a wave's tile: 64 lanes x 8 bytes + 16 bytes of padding:
                                               // 132 words = 4 mod 64 -> the 16 column bases of an operand read fall on distinct banks
constexpr int K2_TILE_B = 16 * K2_SLOT_B;      // 8 448 bytes per wave
constexpr int K2_GS = 17;                      // row stride of the f64 Gram matrix in LDS: [16 x 16 | X'y]
constexpr int K2_TAIL_B = (K2_GS * 16 + 16 * 17 + 64) * 8;   // Gram matrix, solver matrix (16 x 17), 64 doubles of vectors

__device__ __forceinline__ void k2_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double k2_bcast(double v, int j) {    // value of lane j (wave-uniform j: v_readlane_b32 with an SGPR lane select) in every lane
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), j);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), j);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- solvers: wave 0, all 64 lanes active, G = [X'X | X'y] in LDS (row stride K2_GS); each returns this lane's coefficient.
// What a serial phase costs when nothing else hides it (scripts/lat_probe.hip, one wave alone on its SIMD): a dependent f64
// FMA 4.7 cycles, a lane broadcast through v_readlane 15, a TAKEN branch 28-33, an LDS round trip 75, sqrt 104.  So the default
// solvers keep their matrix in registers (lane i owns row i), are fully unrolled (no taken branches, no LDS on the chain) and
// pad the system to KC x KC with an identity block instead of testing kt at every step.
__device__ __forceinline__ double k2_rsqrt(double d) {          // v_rsq_f64 + two Newton steps (full f64 accuracy for d > 0)
    double r = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    r = r * fma(-(hd * r), r, 1.5);
    r = r * fma(-(hd * r), r, 1.5);
    return r;
}

// Cholesky of X'X + alpha I (faer cholesky(Side::Lower), ls.rs:288-297) and the two triangular solves; false = failed / flagged pivot
template <int KC>
__device__ __forceinline__ bool k2_chol(const double *G, int kt, double alpha, double pivot_tol, double *T, int lane, double &bi) {
    const int i = (lane & 15) < KC ? (lane & 15) : KC - 1;   // lanes 16..63 repeat lanes 0..15 (broadcasts read lanes 0..KC-1); rows beyond KC mirror the last
    // [X'X + alpha I | X'y] padded to KC x KC with an identity block: lane i builds row i in registers straight from the Gram matrix
    // (KC independent LDS loads).  (Until round 3 the padded system went through a rolled loop over T: per-lane selects like these used
    // to be hoisted out of the persistent group loop into ~64 VGPRs; the caller launders the lane id per group now, and the rolled
    // loop was 4-16 dependent load -> store round trips of ~300 cycles.)
    double row[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        const double gv = (i < kt && c < kt) ? G[i * K2_GS + c] : 0.0;
        row[c] = (i == c) ? ((i < kt) ? gv + alpha : 1.0) : gv;
    }
    bi = (i < kt) ? G[i * K2_GS + kt] : 0.0;
    double gd = 1.0;                                     // this lane's original diagonal entry
#pragma unroll
    for (int c = 0; c < KC; ++c) gd = (i == c) ? row[c] : gd;
    double rme = 1.0;                                    // 1 / L[i][i], kept on lane i (a uniform array would live in 2 KC SGPRs)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        const double d = k2_bcast(row[j], j);
        ok = ok & (d > pivot_tol * k2_bcast(gd, j));     // also false for NaN; no short-circuit: a taken branch is ~30 cycles
        const double ri = k2_rsqrt(d);
        if (i == j) rme = ri;
        row[j] *= ri;                                    // L[i][j] on the lanes below the diagonal
#pragma unroll
        for (int c = j + 1; c < KC; ++c) row[c] = fma(-row[j], k2_bcast(row[j], c), row[c]);   // - L[i][j] L[c][j]
        __builtin_amdgcn_sched_barrier(0);               // a column's broadcasts stay in their column: 2 (KC - j) SGPRs live, not 2 KC^2 / 2
    }
#pragma unroll
    for (int p = 0; p < KC; ++p) {                       // forward: t = L^-1 b
        const double tp = k2_bcast(bi * rme, p);
        bi = (i == p) ? tp : ((i > p) ? fma(-row[p], tp, bi) : bi);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the backward solve needs column i of L on lane i: one transposition through LDS
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < KC; ++c) T[i * K2_GS + c] = row[c];
    }
    k2_wave_sync();
#pragma unroll
    for (int c = 0; c < KC; ++c) row[c] = T[c * K2_GS + i];     // L[c][i]
#pragma unroll
    for (int p = KC - 1; p >= 0; --p) {                  // backward: beta = L^-T t
        const double bp = k2_bcast(bi * rme, p);
        bi = (i == p) ? bp : ((i < p) ? fma(-row[p], bp, bi) : bi);
        __builtin_amdgcn_sched_barrier(0);
    }
    k2_wave_sync();
    return ok;
}

// Partial-pivot LU of X'X + alpha I with the right-hand side carried along (solve_ols_lu, ls.rs:264-273: faer partial_piv_lu).
// Lane i owns row i of the augmented matrix A (16 x 17 in LDS).  false = a zero / NaN pivot (singular to working precision).
__device__ __forceinline__ bool k2_lu(const double *G, int kt, double alpha, double *A, int lane, double &bi) {
    for (int q = lane; q < kt * (kt + 1); q += 64) {
        const int i = q / (kt + 1), j = q - i * (kt + 1);
        A[i * K2_GS + j] = G[i * K2_GS + j] + (i == j ? alpha : 0.0);
    }
    k2_wave_sync();
    bool ok = true;
    for (int j = 0; j < kt; ++j) {
        // pivot: the largest |A[i][j]|, i >= j (first index on ties)
        double v = (lane >= j && lane < kt) ? fabs(A[lane * K2_GS + j]) : -1.0;
        int idx = lane;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(idx, off);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        const int p = __shfl(idx, 0);
        const double piv = A[p * K2_GS + j];
        ok = ok && (fabs(piv) > 0.0);                    // false for 0 and NaN
        if (p != j && lane >= j && lane <= kt) {
            const double t = A[j * K2_GS + lane];
            A[j * K2_GS + lane] = A[p * K2_GS + lane];
            A[p * K2_GS + lane] = t;
        }
        k2_wave_sync();
        if (lane > j && lane < kt) {
            const double f = A[lane * K2_GS + j] / piv;
            for (int c = j + 1; c <= kt; ++c) A[lane * K2_GS + c] = fma(-f, A[j * K2_GS + c], A[lane * K2_GS + c]);
        }
        k2_wave_sync();
    }
    bi = (lane < kt) ? A[lane * K2_GS + kt] : 0.0;
    for (int p = kt - 1; p >= 0; --p) {
        if (lane == p) bi /= A[p * K2_GS + p];
        const double bp = __shfl(bi, p);
        if (lane < p) bi = fma(-A[lane * K2_GS + p], bp, bi);
    }
    return ok && __all((bi == bi) && fabs(bi) <= 1.7e308);   // wave-uniform verdict
}

template <bool POSITIVE>
__device__ __forceinline__ double k2_soft_threshold(double x, double thr) {   // ls.rs:373-379
    if constexpr (POSITIVE) return fmax(x - thr, 0.0);        // sign(x) max(|x| - thr, 0) clamped at 0
    return copysign(fmax(fabs(x) - thr, 0.0), x);
}

// solve_elastic_net (ls.rs:386-492) on (X'X, X'y).  Lane i keeps u_i = (X'y)_i - sum_{k != i} G_ik w_k, which IS the reference's
// x_i . (residuals + x_i w_i) (:428-430): a step of coordinate j is soft_threshold(u_j) / (G_jj + l2) on lane j, a broadcast of
// its change, and one FMA per lane (u_i -= G_ij dw_j, i != j; u_j does not depend on w_j) -- no cross-lane reduction and no
// multiply-add by the diagonal on the dependency chain.  Same coordinate order, alpha * n scaling (:419), soft threshold, active
// set (:446-489) and ||w - w_old||_2 < tol stop (:436-444).
// One sweep, coordinate J onwards, unrolled by recursion (a loop with early exits is re-rolled by the compiler, and every taken
// branch is ~30 cycles on this chain): no branch is taken until the first coordinate beyond kt.
template <bool POSITIVE, bool ACTIVE, bool FULL, int KC, int J>
__device__ __forceinline__ void k2_cd_steps(const double (&g)[KC], int kt, unsigned sweep, double thr, double ime, double tol, int sub,
                                            double &ume, double &wme, double &d2, unsigned &mask) {
    if constexpr (J < KC) {
        if (FULL || J < kt) {                                // FULL: kt == KC, no test (a wave-uniform compare + branch per coordinate otherwise)
            if (!ACTIVE || ((sweep >> J) & 1u)) {
                const double dme = fma(k2_soft_threshold<POSITIVE>(ume, thr), ime, -wme);   // new - old weight, meaningful on lane J (:430-431)
                const double dj = k2_bcast(dme, J);
                ume = fma(-g[J], dj, ume);
                if (sub == J) wme += dme;
                d2 = fma(dj, dj, d2);
                if constexpr (ACTIVE) {
                    if (fabs(k2_bcast(wme, J)) < tol) mask &= ~(1u << J);                   // (:472-476)
                }
            }
            k2_cd_steps<POSITIVE, ACTIVE, FULL, KC, J + 1>(g, kt, sweep, thr, ime, tol, sub, ume, wme, d2, mask);
        }
    }
}

// This lane's row of X'X (zero diagonal) lives in 2 KC VGPRs: no LDS on the dependency chain.
template <bool POSITIVE, bool ACTIVE, bool FULL, int KC>
__device__ __forceinline__ int k2_cd_loop(const double *G, int kt, double n, const K2Args &a, double *T, int lane, double &wout) {
    const int sub = lane & 15;
    const bool in = sub < kt;
    const double alpha_n = a.alpha * n;                  // alpha * n_samples (:419)
    const double thr = alpha_n * a.l1_ratio, l2 = alpha_n * (1.0 - a.l1_ratio);
    const double ime = 1.0 / ((in ? G[sub * K2_GS + sub] : 1.0) + l2);   // 1 / (xtx[[j, j]] + alpha (1 - l1_ratio))  (:431)
    double ume = in ? G[sub * K2_GS + kt] : 0.0;         // w = zeros (:416)
    double wme = 0.0;
    unsigned mask = (1u << kt) - 1u;
    int status = POLS_GROUP_NOT_CONVERGED;
    // this lane's row of X'X with a zero diagonal, padded with zeros: KC independent loads (see k2_chol)
    double g[KC];
#pragma unroll
    for (int j = 0; j < KC; ++j) g[j] = (in && j < kt && j != sub) ? G[sub * K2_GS + j] : 0.0;
    (void)T;
    const double tol2 = a.tol > 0.0 ? a.tol * a.tol : -1.0;   // ||dw||_2 < tol  <=>  ||dw||^2 < tol^2 (a sqrt is ~100 cycles per sweep)
    for (int64_t it = 0; it < a.max_iter; ++it) {
        double d2 = 0.0;
        const unsigned sweep = mask;                     // `for j in active_indices.clone()` (:459)
        k2_cd_steps<POSITIVE, ACTIVE, FULL, KC, 0>(g, kt, sweep, thr, ime, a.tol, sub, ume, wme, d2, mask);
        if (d2 < tol2) { status = POLS_GROUP_OK; break; }     // (:436-444)
    }
    wout = wme;
    return status;
}

template <int KC>
__device__ __forceinline__ int k2_cd(const double *G, int kt, double n, const K2Args &a, double *T, int lane, double &wout) {
    const bool act = a.solver == K2_CD_ACTIVE_SET;
    if (a.positive) return act ? k2_cd_loop<true, true, false, KC>(G, kt, n, a, T, lane, wout) : k2_cd_loop<true, false, false, KC>(G, kt, n, a, T, lane, wout);
    if (act) return k2_cd_loop<false, true, false, KC>(G, kt, n, a, T, lane, wout);
    // the plain cyclic form at full width (cfg5: 16 columns) without the per-coordinate `J < kt` test
    if (kt == KC) return k2_cd_loop<false, false, true, KC>(G, kt, n, a, T, lane, wout);
    return k2_cd_loop<false, false, false, KC>(G, kt, n, a, T, lane, wout);
}

// element v of the result = (row v of the chunk belongs to the group) ? loaded[v + shift] : fill  -- the ragged-edge fix-up
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type k2_fix(const typename Vec16<T>::type &ld, int shift, unsigned keep, T fill) {
    constexpr int VEC = Vec16<T>::N;
    typename Vec16<T>::type out;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T val = fill;
#pragma unroll
        for (int u = 0; u < VEC; ++u) val = (u == v + shift) ? vget<T>(ld, u) : val;
        vset<T>(out, v, ((keep >> v) & 1u) ? val : fill);
    }
    return out;
}

// 8 bytes of a 16-byte vector: one f64 row or two f32 rows
__device__ __forceinline__ double k2_half(const double2 &v, int h) { return h == 0 ? v.x : v.y; }
__device__ __forceinline__ float2 k2_half(const float4 &v, int h) { return h == 0 ? float2{v.x, v.y} : float2{v.z, v.w}; }

// Waves per SIMD the register budget is held to: the resident rows are RC x KC 16-byte vectors (4 VGPRs each) per lane.
__host__ __device__ constexpr int k2_occupancy(int kc, int rc, int waves, bool yv) {
    (void)yv;
    if (waves == 8) return 2;                                // a 512-thread workgroup puts two waves on every SIMD; persistent, one per CU
    if (kc * rc <= 8) return 4;                              // 32 data registers + the solver's 2 KC
    if (kc == 8) return 3;                                   // 64 data registers, 16 for the solver
    return 2;                                                // 16 columns: 64 / 128 data registers + 32 for the solver: anything tighter spills
                                                             // around the solve, and a spill reload there is ~1 us of exposed latency
}

template <typename T, int KC, int WAVES, int RC, bool YV, bool HAS_W>
__global__ void __launch_bounds__(64 * WAVES, k2_occupancy(KC, RC, WAVES, YV)) k2_kernel(const K2Args a, const int pf_waves) {
    using V = typename Vec16<T>::type;
    using M = Mfma16<T>;
    using acc_t = typename M::acc_t;
    using H = decltype(k2_half(V{}, 0));                     // double or float2
    constexpr int VEC = Vec16<T>::N;
    constexpr int TPB = 64 * WAVES;
    static_assert(KC <= K2_KMAX && (!YV || KC == K2_KMAX), "YV is the 16-column case");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *Gs = reinterpret_cast<double *>(smem + (size_t)WAVES * K2_TILE_B);      // [16][17]
    double *As = Gs + K2_GS * 16;                                                     // solver matrix [16][17]
    double *vec = As + 16 * 17;                                                       // [0,16) rinv, [16,32) beta, [32,48) unused
    double *xyp = vec + 64;                                                           // YV: [WAVES][16]
    // PERSISTENT (eight-wave variants: one workgroup per CU, nothing else on the CU to hide the solver): the workgroup walks the
    // groups blockIdx.x, + gridDim.x, ... and, while wave 0 solves group g, waves 1..7 DMA the first chunk (every column) of the
    // first pf_waves waves of the NEXT group into LDS (`global_load_lds`, 1 KiB per wave-instruction) -- a CU's memory pipe
    // delivers ~10 bytes per clock however idle HBM is, so every clock without loads in flight is lost bandwidth.
    unsigned char *pf = smem + (size_t)WAVES * K2_TILE_B + K2_TAIL_B + (YV ? WAVES * 16 * 8 : 0);   // [wave][column][1 KiB]
    constexpr bool PERSISTENT = WAVES == 8;

    bool pf_ready = false;                                   // the prefetch buffer holds chunks of the group being worked on
    // columns never written keep zeros for the whole kernel (their operand lanes must read 0)
    {
        const int first = YV ? a.kt : a.kt + 1;
        unsigned char *tile0 = smem + (size_t)(threadIdx.x >> 6) * K2_TILE_B;
        for (int i = threadIdx.x & 63; i < (16 - first) * (K2_SLOT_B / 8); i += 64)
            reinterpret_cast<double *>(tile0 + (size_t)first * K2_SLOT_B)[i] = 0.0;
    }
#pragma unroll 1
    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
    // The thread index and the column counts are laundered once per group: everything derived from them inside the body (lane
    // predicates, LDS addresses, the solvers' selection masks) would otherwise be loop-invariant, get hoisted out of the
    // persistent loop and sit in ~100 registers on top of the resident rows.
    int tid = threadIdx.x, kt = a.kt, ku = a.k_user;
    asm volatile("" : "+v"(tid), "+s"(kt), "+s"(ku) :: "memory");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool icpt = ku != kt;
    constexpr bool has_w = HAS_W;                            // a template parameter: the sqrt(w) registers only exist when there are weights
    const int ncols = ku + 1 + (has_w ? 1 : 0);              // columns that are loaded: features, target, weights
    unsigned char *mytile = smem + (size_t)wave * K2_TILE_B;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);                      // the chunk grid is aligned to 16 bytes in every column
    const int64_t nch = (e - base + VEC - 1) / VEC;          // <= RC * TPB: the host checked the largest group
    unsigned long long *dbg = a.dbg ? a.dbg + g * 8 : nullptr;
#define K2_STAMP(i) do { if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    K2_STAMP(0);

    // ---- every load of every resident chunk, back to back, with NO divergent branch around a load: the compiler's s_waitcnt
    // insertion only tracks "the first chunk's 17 loads have landed, the second chunk's are still in flight" through straight-line
    // code -- with loads in both arms of a per-lane branch it falls back to vmcnt(0) before the first use, which serialises the
    // matrix-core work behind the whole group's arrival.  So every lane issues the same 16-byte loads: a lane whose chunk crosses
    // the group's edge reads its neighbours' rows too, a lane without a chunk (or whose chunk would cross the end of the frame)
    // reads from a clamped position; a wave-uniform fix-up, taken only by waves that own such a lane, shifts / zeroes afterwards.
    // Every one of the KC column slots is loaded unconditionally too (slots beyond the user's features point at the target
    // column -- the host filled them in -- and are overwritten at use): a load guarded by `j < k_user` leaves the compiler
    // unable to count the loads behind it, and the wait before the first chunk's use degrades to "all but the last two".
    V x[RC][KC], yv[RC], sw[RC];
    unsigned keep[RC];                                       // bit v: row v of the chunk belongs to the group
    int shift[RC];                                           // rows the load position was moved back by (end of the frame)
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const int64_t c = (int64_t)rc * TPB + tid;
        const int64_t row0 = base + c * VEC;
        const bool any = c < nch;
        int64_t rl = any ? row0 : base;
        if (rl > a.n_rows - VEC) rl = a.n_rows - VEC;        // n_rows >= VEC: checked by the host
        shift[rc] = any ? (int)(row0 - rl) : 0;
        keep[rc] = 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) keep[rc] |= (any && row0 + v >= s && row0 + v < e) ? (1u << v) : 0u;
        if (PERSISTENT && rc == 0 && pf_ready && wave < pf_waves) {      // wave-uniform
            // this wave's first chunk was DMA'd into LDS while the previous group was being solved (same clamped positions)
            const unsigned char *mine = pf + (size_t)wave * ncols * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < KC; ++j) x[rc][j] = *reinterpret_cast<const V *>(mine + (size_t)(j < ku ? j : ku) * 1024);
            yv[rc] = *reinterpret_cast<const V *>(mine + (size_t)ku * 1024);
            sw[rc] = *reinterpret_cast<const V *>(mine + (size_t)(has_w ? ku + 1 : ku) * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < KC; ++j) x[rc][j] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + rl);
            yv[rc] = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + rl);
            sw[rc] = *reinterpret_cast<const V *>(static_cast<const T *>(has_w ? a.w : a.y) + rl);
        }
        asm volatile("" ::: "memory");                       // chunk by chunk: the second chunk's loads stay behind the first's
    }

    // ---- Gram: per chunk and 8-byte half, transpose through the wave's own tile and feed the matrix cores
    double accd[4] = {0.0, 0.0, 0.0, 0.0};                   // f32 only: the f64 running sums of the flushed tiles
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    T xy[YV ? KC : 1];
    if constexpr (YV) {
#pragma unroll
        for (int j = 0; j < KC; ++j) xy[j] = T(0);
    }
    const unsigned char *zp = mytile + (size_t)(lane & 15) * K2_SLOT_B + (lane >> 4) * 8;   // operand stream of lane (c, q)
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        // (a wave none of whose lanes owns a row of this chunk -- the upper waves of a group that fills little more than the first chunk
        // -- has only zeros to contribute: it skips the two tile stages and their 32 matrix-core instructions)
        if (rc > 0 && !__any(keep[rc] != 0u)) continue;
        // -- this chunk's registers, prepared only now (nothing above touched them: the wait is for THIS chunk's loads)
        // (a wave-uniform compare + branch is ~10 cycles falling through and ~30 taken: with 16 slots the first 8 are always user
        // columns -- the 16-slot variants start at 9 columns)
        // ... and with every slot a user column (cfg5: 16 features) one test instead of eight: 5.77 vs 6.15 ms there.  f64 only: in the
        // f32 16-slot kernels with weights the extra path cost 48 bytes of scratch at 256 VGPRs.
        if (sizeof(T) == 4 || ku != KC) {
#pragma unroll
            for (int j = (KC == K2_KMAX ? KC / 2 : 0); j < KC; ++j)
                if (j >= ku) x[rc][j] = vsplat<T>((icpt && j == kt - 1) ? T(1) : T(0));      // wave-uniform: intercept / unused slot
        }
        if (!has_w) sw[rc] = vsplat<T>(T(1));
        if (__any(keep[rc] != ((1u << VEC) - 1u))) {         // wave-uniform, no loads inside: a ragged edge somewhere in the wave
#pragma unroll
            for (int j = 0; j < KC; ++j) x[rc][j] = k2_fix<T>(x[rc][j], (j < ku) ? shift[rc] : 0, keep[rc], T(0));
            yv[rc] = k2_fix<T>(yv[rc], shift[rc], keep[rc], T(0));
            sw[rc] = k2_fix<T>(sw[rc], has_w ? shift[rc] : 0, keep[rc], T(1));
        }
        if (has_w) {       // sqrt(w) scaling of every feature, intercept included (least_squares.py:190-196); yv keeps the ORIGINAL target
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T q = sqrt(vget<T>(sw[rc], v));
                vset<T>(sw[rc], v, q);
#pragma unroll
                for (int j = 0; j < KC; ++j) vset<T>(x[rc][j], v, vget<T>(x[rc][j], v) * q);
            }
        }
        V ys;                                                // sqrt(w) y
#pragma unroll
        for (int v = 0; v < VEC; ++v) vset<T>(ys, v, vget<T>(yv[rc], v) * vget<T>(sw[rc], v));
        if constexpr (YV) {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int j = 0; j < KC; ++j) xy[j] = fma(vget<T>(x[rc][j], v), vget<T>(ys, v), xy[j]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // every slot is written, unguarded: the slots from k_user on hold the synthesised columns (ones for the intercept, zeros beyond
            // kt -- what the one-off zero fill of the tiles used to provide), and slot kt is overwritten with the target just below
#pragma unroll
            for (int j = 0; j < KC; ++j) *reinterpret_cast<H *>(mytile + (size_t)j * K2_SLOT_B + lane * 8) = k2_half(x[rc][j], h);
            if constexpr (!YV) *reinterpret_cast<H *>(mytile + (size_t)kt * K2_SLOT_B + lane * 8) = k2_half(ys, h);
            k2_wave_sync();
            if (rc == 0 && h == 0) K2_STAMP(1);              // the first chunk's loads have landed
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                const H v0 = *reinterpret_cast<const H *>(zp + u * 32);
                const H v1 = *reinterpret_cast<const H *>(zp + u * 32 + 32);
                if constexpr (sizeof(T) == 8) {
                    acc0 = M::mma(v0, v0, acc0);
                    acc1 = M::mma(v1, v1, acc1);
                } else {
                    acc0 = M::mma(v0.x, v0.x, acc0);
                    acc1 = M::mma(v0.y, v0.y, acc1);
                    acc0 = M::mma(v1.x, v1.x, acc0);
                    acc1 = M::mma(v1.y, v1.y, acc1);
                }
            }
            if constexpr (sizeof(T) == 4) {                  // 128 rows per flush: an f32 Gram matrix with f64-summation error
#pragma unroll
                for (int r = 0; r < 4; ++r) accd[r] += (double)acc0[r] + (double)acc1[r];
                acc0 = acc_t{0, 0, 0, 0}; acc1 = acc_t{0, 0, 0, 0};
            }
            k2_wave_sync();                                  // the next stage overwrites the tile
        }
    }
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) accd[r] = acc0[r] + acc1[r];
    }
    K2_STAMP(2);
    // ---- per-wave partial tile (and X'y) -> LDS
    {
        double *part = reinterpret_cast<double *>(mytile);
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r * 64 + lane] = accd[r];
        if constexpr (YV) {
            T u[(KC + 3) / 4];
            wave_reduce_scatter<T, KC>(xy, u);
            const int row = lane >> 4;
            if ((lane & 15) == 0) {
#pragma unroll
                for (int i = 0; i < (KC + 3) / 4; ++i) xyp[wave * 16 + 4 * i + rs_perm(row)] = (double)u[i];
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 256; idx += TPB) {
        const int r = idx >> 6, l = idx & 63;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += reinterpret_cast<const double *>(smem + (size_t)w * K2_TILE_B)[idx];
        const int drow = (sizeof(T) == 4) ? (l >> 4) * 4 + r : (l >> 4) + 4 * r;   // C/D layouts of the f32 / f64 16x16x4 MFMA
        Gs[drow * K2_GS + (l & 15)] = v;
    }
    if constexpr (YV) {
        if (tid < 16) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += xyp[w * 16 + tid];
            Gs[tid * K2_GS + 16] = v;
        }
    }
    __syncthreads();
    K2_STAMP(3);
    bool pf_next = false;
    if constexpr (PERSISTENT) {
        const int64_t gn = g + gridDim.x;
        pf_next = pf_waves > 0 && gn < a.n_groups;           // block-uniform
        if (pf_next && wave != 0) {
            const int64_t sn = a.offs[gn], en = a.offs[gn + 1];
            const int64_t basen = sn - (sn % VEC);
            const int64_t nchn = (en - basen + VEC - 1) / VEC;
            const int npieces = pf_waves * ncols;
            for (int p = wave - 1; p < npieces; p += WAVES - 1) {
                const int wt = p / ncols, col = p - wt * ncols;
                const T *src = static_cast<const T *>(col < ku ? a.x[col] : (col == ku ? a.y : a.w));
                const int64_t c = (int64_t)wt * 64 + lane;
                int64_t rl = c < nchn ? basen + c * VEC : basen;            // the clamped position the consumer expects
                if (rl > a.n_rows - VEC) rl = a.n_rows - VEC;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + rl),
                                                 (__attribute__((address_space(3))) void *)(pf + (size_t)p * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed before the barrier below lets anyone read them
        }
    }

    // ---- solve: wave 0, f64
    if (wave == 0) {
        int st = POLS_GROUP_OK;
        double bi = 0.0;
        if (e == s) st = POLS_GROUP_EMPTY;                   // features.is_empty() -> zeros (ex.rs:357-359)
        else if (a.solver == K2_CD || a.solver == K2_CD_ACTIVE_SET) {
            st = k2_cd<KC>(Gs, kt, (double)(e - s), a, As, lane, bi);
        } else {
            bool ok;
            if (a.solver == K2_LU) ok = k2_lu(Gs, kt, a.alpha, As, lane, bi);
            else {
                // (the solve is a chain of ~1 300 dependent-ish instructions whatever kt: 10.9 k cycles of a 27 k-cycle group at 9-15 columns.
                // Padded to 10 / 12 / 14 instead of 16 it does up to half less: 8.3 k cycles at 12.)
                if (KC == K2_KMAX && kt <= 10) ok = k2_chol<10>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi);
                else if (KC == K2_KMAX && kt <= 12) ok = k2_chol<12>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi);
                else if (KC == K2_KMAX && kt <= 14) ok = k2_chol<14>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi);
                else ok = k2_chol<KC>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi);
                if (!ok && a.lu_fallback) ok = k2_lu(Gs, kt, a.alpha, As, lane, bi);   // solve_ridge: Cholesky -> LU (ls.rs:358-363)
            }
            if (!ok) { st = POLS_GROUP_FALLBACK; if (lane == 0 && a.fb_flag) *a.fb_flag = a.epoch; }
        }
        if (lane == 0 && a.status) a.status[g] = st;
        if (lane < 16) {
            const double out = lane < kt ? bi : 0.0;
            vec[16 + lane] = out;
            if (lane < kt) {
                if (a.coef) static_cast<T *>(a.coef)[g * kt + lane] = (T)out;
                if (a.coef64) a.coef64[g * kt + lane] = out;
            }
        }
    }
    __syncthreads();
    K2_STAMP(4);

    // ---- predictions / residuals from the resident rows
    if (a.pred || a.resid) {
        T *pred = static_cast<T *>(a.pred);
        T *resid = static_cast<T *>(a.resid);
        T p[RC][VEC];
#pragma unroll
        for (int rc = 0; rc < RC; ++rc)
#pragma unroll
            for (int v = 0; v < VEC; ++v) p[rc][v] = T(0);
#pragma unroll
        for (int j = 0; j < KC; ++j) {                       // make_predictions (ex.rs:398-405); coefficient j straight from LDS
            const T bj = (T)vec[16 + j];
#pragma unroll
            for (int rc = 0; rc < RC; ++rc)
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[rc][v] = fma(vget<T>(x[rc][j], v), bj, p[rc][v]);
        }
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t c = (int64_t)rc * TPB + tid;
            const int64_t row0 = base + c * VEC;
            if (c < nch) {
                V pv, rv;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    T acc = p[rc][v];
                    if (has_w) acc *= T(1) / vget<T>(sw[rc], v);                                 // predictions *= 1/sqrt_w (ls.py:234-235)
                    vset<T>(pv, v, acc);
                    vset<T>(rv, v, vget<T>(yv[rc], v) - acc);                                    // ORIGINAL target - predictions (ls.py:239)
                }
                if (keep[rc] == ((1u << VEC) - 1u)) {
                    if (pred) store_stream(reinterpret_cast<V *>(pred + row0), pv);
                    if (resid) store_stream(reinterpret_cast<V *>(resid + row0), rv);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int64_t rr = row0 + v;
                        if (rr >= s && rr < e) {
                            if (pred) pred[rr] = vget<T>(pv, v);
                            if (resid) resid[rr] = vget<T>(rv, v);
                        }
                    }
                }
            }
        }
    }
    K2_STAMP(5);
    if (dbg && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[6] = xcc;
    }
#undef K2_STAMP
    pf_ready = pf_next;
    }   // groups of this workgroup
}

template <typename T, int KC, int WAVES, int RC, bool YV, bool HAS_W>
static int k2_launch_v(pols_ctx *ctx, const K2Args &a) {
    size_t lds = (size_t)WAVES * K2_TILE_B + K2_TAIL_B + (YV ? WAVES * 16 * 8 : 0);
    // eight-wave variants are persistent (one workgroup per CU) and use the rest of the CU's 160 KiB LDS as a prefetch buffer for
    // the next group: whole first chunks (every loaded column, 1 KiB each) of as many waves as fit
    int pf_waves = 0;
    unsigned grid = (unsigned)a.n_groups;
    if (WAVES == 8 && !ctx->opt.k2_noprefetch) {
        const int ncols = a.k_user + 1 + (a.w ? 1 : 0);
        pf_waves = (int)std::min<size_t>(WAVES, (160 * 1024 - lds) / ((size_t)ncols * 1024));
        lds += (size_t)pf_waves * ncols * 1024;
        grid = (unsigned)std::min<int64_t>(a.n_groups, (int64_t)ctx->num_cus * k2_occupancy(KC, RC, WAVES, YV) / 2);
    }
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_kernel<T, KC, WAVES, RC, YV, HAS_W>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done(ctx->device);
    }
    static const char *const solver_names[] = {"chol", "lu", "cd", "cdas"};
    char name[96];
    std::snprintf(name, sizeof(name), "k2_gram_mfma_resident_%s_k%d%s_w%d_rc%d%s_%s", sizeof(T) == 4 ? "f32" : "f64", KC, YV ? "yv" : "",
                  WAVES, RC, a.w ? "_w" : "", solver_names[a.solver & 3]);
    ctx->last_kernel = name;
    if (a.n_groups > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    K2Args aa = a;
    if (ctx->opt.timeline) {
        void *d = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_groups, &d);
        if (rc) return rc;
        aa.dbg = static_cast<unsigned long long *>(d);
    }
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((k2_kernel<T, KC, WAVES, RC, YV, HAS_W>), dim3(grid), dim3(64 * WAVES), (unsigned)lds, ctx->stream, ev0, ev1, 0, aa, pf_waves);
    else
        hipLaunchKernelGGL((k2_kernel<T, KC, WAVES, RC, YV, HAS_W>), dim3(grid), dim3(64 * WAVES), lds, ctx->stream, aa, pf_waves);
    POLS_HIP(hipGetLastError());
    if (ctx->opt.timeline) return report_timeline(ctx, aa.dbg, a.n_groups, 6, name);
    return POLS_OK;
}

template <typename T, bool HAS_W, int WAVES, int RC>
static int k2_launch_shape(pols_ctx *ctx, const K2Args &a) {
    if (a.kt == 16) return k2_launch_v<T, 16, WAVES, RC, true, HAS_W>(ctx, a);
    if (a.kt > 8) return k2_launch_v<T, 16, WAVES, RC, false, HAS_W>(ctx, a);
    return k2_launch_v<T, 8, WAVES, RC, false, HAS_W>(ctx, a);
}

// capacity (rows) of the variants, smallest first: {WAVES, RC}
template <typename T, bool HAS_W>
int k2_launch_t(pols_ctx *ctx, const K2Args &a, int64_t need) {
    constexpr int VEC = Vec16<T>::N;
    if (need <= 64 * 1 * VEC) return k2_launch_shape<T, HAS_W, 1, 1>(ctx, a);
    if (need <= 64 * 2 * VEC) return k2_launch_shape<T, HAS_W, 1, 2>(ctx, a);
    if (need <= 256 * 1 * VEC) return k2_launch_shape<T, HAS_W, 4, 1>(ctx, a);
    if (need <= 256 * 2 * VEC) return k2_launch_shape<T, HAS_W, 4, 2>(ctx, a);
    if (need <= 512 * 2 * VEC) return k2_launch_shape<T, HAS_W, 8, 2>(ctx, a);
    // up to eight columns: four chunks per lane (128 + 32 data registers of the 256) -- 8 192 f32 / 4 096 f64 rows stay on chip, twenty
    // to thirty years of trading days per asset; the streamed path reads such groups twice (2.8 TB/s of algorithmic bytes)
    if (a.kt >= 7 && a.kt <= 8 && need <= 512 * 4 * VEC) return k2_launch_v<T, 8, 8, 4, false, HAS_W>(ctx, a);
    return fail(POLS_ERR_UNSUPPORTED, "k2: %lld-row groups exceed the resident capacity", (long long)need);
}

}  // namespace pols
