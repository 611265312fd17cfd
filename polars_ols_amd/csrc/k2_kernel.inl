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

// ---- solvers: wave 0, all 64 lanes active, G = [X'X | X'y] in LDS (row stride K2_GS); each returns this lane's coefficient
// Cholesky of X'X + alpha I, lane i owns row i of L (faer cholesky(Side::Lower), ls.rs:288-297); false = failed / flagged pivot
__device__ __forceinline__ bool k2_chol(const double *G, int kt, double alpha, double pivot_tol, double *L, double *rinv, int lane,
                                        double &bi) {
    for (int q = lane; q < kt * kt; q += 64) {
        const int i = q / kt, j = q - i * kt;
        L[i * 16 + j] = G[i * K2_GS + j] + (i == j ? alpha : 0.0);
    }
    bi = (lane < kt) ? G[lane * K2_GS + kt] : 0.0;
    k2_wave_sync();
    bool ok = true;
    for (int j = 0; j < kt; ++j) {
        double d = L[j * 16 + j];
        const double gjj = d;
        for (int p = 0; p < j; ++p) d = fma(-L[j * 16 + p], L[j * 16 + p], d);
        ok = ok && (d > pivot_tol * gjj);
        const double ri = 1.0 / sqrt(d);
        if (lane == 0) rinv[j] = ri;
        if (lane > j && lane < kt) {
            double sacc = L[lane * 16 + j];
            for (int p = 0; p < j; ++p) sacc = fma(-L[lane * 16 + p], L[j * 16 + p], sacc);
            L[lane * 16 + j] = sacc * ri;
        }
        k2_wave_sync();
    }
    for (int p = 0; p < kt; ++p) {                       // forward: t = L^-1 b
        if (lane == p) bi *= rinv[p];
        const double tp = __shfl(bi, p);
        if (lane > p && lane < kt) bi = fma(-L[lane * 16 + p], tp, bi);
    }
    for (int p = kt - 1; p >= 0; --p) {                  // backward: beta = L^-T t
        if (lane == p) bi *= rinv[p];
        const double bp = __shfl(bi, p);
        if (lane < p) bi = fma(-L[p * 16 + lane], bp, bi);
    }
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

__device__ __forceinline__ double k2_soft_threshold(double x, double thr, bool positive) {   // ls.rs:373-379
    double r = copysign(fmax(fabs(x) - thr, 0.0), x);
    if (positive) r = fmax(r, 0.0);
    return r;
}

// solve_elastic_net (ls.rs:386-492) on (X'X, X'y): lane i keeps q_i = (X'y)_i - sum_k G_ik w_k, so that the reference's
// x_j . (residuals + x_j w_j) (:428-430) is q_j + G_jj w_j on lane j and a step of coordinate j costs every lane one FMA
// (q_i -= G_ij dw_j) -- no cross-lane reduction on the dependency chain, only a broadcast of lane j's step.  Same coordinate
// order, alpha * n scaling (:419), soft threshold, active set (:446-489) and ||w - w_old||_2 < tol stop (:436-444).
__device__ __forceinline__ int k2_cd(const double *G, int kt, double n, const K2Args &a, int lane, double &wout) {
    const int sub = lane & 15;
    const bool in = sub < kt;
    const double alpha_n = a.alpha * n;                  // alpha * n_samples (:419)
    const double thr = alpha_n * a.l1_ratio, l2 = alpha_n * (1.0 - a.l1_ratio);
    const bool positive = a.positive != 0, active_set = a.solver == K2_CD_ACTIVE_SET;
    const double dme = in ? G[sub * K2_GS + sub] : 1.0;  // xtx[[j, j]] (:431)
    const double ime = 1.0 / (dme + l2);
    double qme = in ? G[sub * K2_GS + kt] : 0.0;         // w = zeros (:416)
    double wme = 0.0;
    unsigned mask = (1u << kt) - 1u;
    int status = POLS_GROUP_NOT_CONVERGED;
    // a ROLLED coordinate loop (lane broadcasts take the lane index from an SGPR): the solver's registers sit on top of the
    // resident rows of every wave of the kernel, so it must stay small; the next coordinate's Gram column is fetched from LDS
    // while the current step's dependency chain runs
    for (int64_t it = 0; it < a.max_iter; ++it) {
        double d2 = 0.0;
        const unsigned sweep = mask;                     // `for j in active_indices.clone()` (:459)
        double gnext = in ? G[sub] : 0.0;
        for (int j = 0; j < kt; ++j) {
            const double gj = gnext;
            gnext = in ? G[(j + 1 < kt ? j + 1 : 0) * K2_GS + sub] : 0.0;
            if (!((sweep >> j) & 1u)) continue;
            const double cand = k2_soft_threshold(fma(dme, wme, qme), thr, positive) * ime;   // meaningful on lane j (:430-431)
            const double wj = k2_bcast(cand, j);
            const double dj = k2_bcast(cand - wme, j);
            qme = fma(-gj, dj, qme);
            if (sub == j) wme = cand;
            d2 = fma(dj, dj, d2);
            if (active_set && fabs(wj) < a.tol) mask &= ~(1u << j);   // (:472-476)
        }
        if (sqrt(d2) < a.tol) { status = POLS_GROUP_OK; break; }      // (:436-444)
    }
    wout = wme;
    return status;
}

// 8 bytes of a 16-byte vector: one f64 row or two f32 rows
__device__ __forceinline__ double k2_half(const double2 &v, int h) { return h == 0 ? v.x : v.y; }
__device__ __forceinline__ float2 k2_half(const float4 &v, int h) { return h == 0 ? float2{v.x, v.y} : float2{v.z, v.w}; }

// Waves per SIMD the register budget is held to: the resident rows are RC x KC 16-byte vectors (4 VGPRs each) per lane.
__host__ __device__ constexpr int k2_occupancy(int kc, int rc, int waves, bool yv) {
    if (waves == 8) return kc * rc <= 16 ? 4 : 2;            // a 512-thread workgroup puts two waves on every SIMD
    if (kc * rc <= 8) return 5;
    if (kc * rc <= 16) return (rc == 2 || yv) ? 3 : 4;       // two chunks (or the X'y accumulators) need more than 128 - 64 registers
    return 2;
}

template <typename T, int KC, int WAVES, int RC, bool YV>
__global__ void __launch_bounds__(64 * WAVES, k2_occupancy(KC, RC, WAVES, YV)) k2_kernel(const K2Args a) {
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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);                      // the chunk grid is aligned to 16 bytes in every column
    const int64_t nch = (e - base + VEC - 1) / VEC;          // <= RC * TPB: the host checked the largest group
    const int ku = a.k_user, kt = a.kt;
    const bool icpt = ku != kt, has_w = a.w != nullptr;
    unsigned char *mytile = smem + (size_t)wave * K2_TILE_B;

    // ---- every load of every resident chunk, back to back
    V x[RC][KC], yv[RC], sw[RC];
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const int64_t c = (int64_t)rc * TPB + tid;
        const int64_t row0 = base + c * VEC;
        const bool any = c < nch;
        if (any && row0 >= s && row0 + VEC <= e) {
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                if (j < ku) x[rc][j] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + row0);
                else x[rc][j] = vsplat<T>((icpt && j == kt - 1) ? T(1) : T(0));
            }
            yv[rc] = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0);
            if (has_w) sw[rc] = *reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0);
            else sw[rc] = vsplat<T>(T(1));
        } else {
            // ragged head / tail of the group, or a lane without a chunk: rows outside [s, e) are zero rows
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t r = row0 + v;
                const bool in = any && r >= s && r < e;
                vset<T>(yv[rc], v, in ? static_cast<const T *>(a.y)[r] : T(0));
                vset<T>(sw[rc], v, (in && has_w) ? static_cast<const T *>(a.w)[r] : T(1));
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    T xv = T(0);
                    if (in && j < ku) xv = static_cast<const T *>(a.x[j])[r];
                    else if (in && icpt && j == kt - 1) xv = T(1);
                    vset<T>(x[rc][j], v, xv);
                }
            }
        }
    }
    // columns never written keep zeros for the whole kernel (their operand lanes must read 0)
    {
        const int first = YV ? kt : kt + 1;
        for (int i = lane; i < (16 - first) * (K2_SLOT_B / 8); i += 64)
            reinterpret_cast<double *>(mytile + (size_t)first * K2_SLOT_B)[i] = 0.0;
    }
    // sqrt(w) scaling of every feature, intercept included (least_squares.py:190-196); yv keeps the ORIGINAL target
    if (has_w) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const T q = sqrt(vget<T>(sw[rc], v));
                vset<T>(sw[rc], v, q);
#pragma unroll
                for (int j = 0; j < KC; ++j) vset<T>(x[rc][j], v, vget<T>(x[rc][j], v) * q);
            }
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
#pragma unroll
            for (int j = 0; j < KC; ++j)
                if (j < kt) *reinterpret_cast<H *>(mytile + (size_t)j * K2_SLOT_B + lane * 8) = k2_half(x[rc][j], h);
            if constexpr (!YV) *reinterpret_cast<H *>(mytile + (size_t)kt * K2_SLOT_B + lane * 8) = k2_half(ys, h);
            k2_wave_sync();
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

    // ---- solve: wave 0, f64
    if (wave == 0) {
        int st = POLS_GROUP_OK;
        double bi = 0.0;
        if (e == s) st = POLS_GROUP_EMPTY;                   // features.is_empty() -> zeros (ex.rs:357-359)
        else if (a.solver == K2_CD || a.solver == K2_CD_ACTIVE_SET) {
            st = k2_cd(Gs, kt, (double)(e - s), a, lane, bi);
        } else {
            bool ok;
            if (a.solver == K2_LU) ok = k2_lu(Gs, kt, a.alpha, As, lane, bi);
            else {
                ok = k2_chol(Gs, kt, a.alpha, a.pivot_tol, As, vec, lane, bi);
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
                if (row0 >= s && row0 + VEC <= e) {
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
}

template <typename T, int KC, int WAVES, int RC, bool YV>
static int k2_launch_v(pols_ctx *ctx, const K2Args &a) {
    const size_t lds = (size_t)WAVES * K2_TILE_B + K2_TAIL_B + (YV ? WAVES * 16 * 8 : 0);
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_kernel<T, KC, WAVES, RC, YV>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_once.done(ctx->device);
    }
    static const char *const solver_names[] = {"chol", "lu", "cd", "cdas"};
    char name[96];
    std::snprintf(name, sizeof(name), "k2_gram_mfma_resident_%s_k%d%s_w%d_rc%d%s_%s", sizeof(T) == 4 ? "f32" : "f64", KC, YV ? "yv" : "",
                  WAVES, RC, a.w ? "_w" : "", solver_names[a.solver & 3]);
    ctx->last_kernel = name;
    if (a.n_groups > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((k2_kernel<T, KC, WAVES, RC, YV>), dim3((unsigned)a.n_groups), dim3(64 * WAVES), (unsigned)lds, ctx->stream, ev0, ev1, 0, a);
    else
        hipLaunchKernelGGL((k2_kernel<T, KC, WAVES, RC, YV>), dim3((unsigned)a.n_groups), dim3(64 * WAVES), lds, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T, int WAVES, int RC>
static int k2_launch_shape(pols_ctx *ctx, const K2Args &a) {
    if (a.kt == 16) return k2_launch_v<T, 16, WAVES, RC, true>(ctx, a);
    if (a.kt > 8) return k2_launch_v<T, 16, WAVES, RC, false>(ctx, a);
    return k2_launch_v<T, 8, WAVES, RC, false>(ctx, a);
}

// capacity (rows) of the variants, smallest first: {WAVES, RC}
template <typename T>
int k2_launch_t(pols_ctx *ctx, const K2Args &a, int64_t need) {
    constexpr int VEC = Vec16<T>::N;
    if (need <= 64 * 1 * VEC) return k2_launch_shape<T, 1, 1>(ctx, a);
    if (need <= 64 * 2 * VEC) return k2_launch_shape<T, 1, 2>(ctx, a);
    if (need <= 256 * 1 * VEC) return k2_launch_shape<T, 4, 1>(ctx, a);
    if (need <= 256 * 2 * VEC) return k2_launch_shape<T, 4, 2>(ctx, a);
    if (need <= 512 * 2 * VEC) return k2_launch_shape<T, 8, 2>(ctx, a);
    return fail(POLS_ERR_UNSUPPORTED, "k2: %lld-row groups exceed the resident capacity", (long long)need);
}

}  // namespace pols
