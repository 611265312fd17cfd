// k1_kernel.inl -- body of K1 (see k1_gram_chol.hpp for the design notes).  Included by k1_f32.hip / k1_f64.hip.
#include "k1_gram_chol.hpp"

namespace pols {

template <int NZ>
__host__ __device__ constexpr int tri_index(int i, int j) {  // packed upper triangle, i <= j < NZ
    return i * NZ - (i * (i - 1)) / 2 + (j - i);
}

// One chunk = VEC consecutive rows of every column, held by one lane.
template <typename T, int KT>
struct Chunk {
    static constexpr int VEC = Vec16<T>::N;
    T x[VEC][KT];  // sqrt(w)-scaled features, intercept (if any) in column KT-1
    T y[VEC];      // ORIGINAL target (needed for residuals)
    T sw[VEC];     // sqrt(w); 1 when there are no weights
};

template <typename T, int KT>
__device__ __forceinline__ void load_chunk(const K1Args &a, int64_t row0, int64_t s, int64_t e, Chunk<T, KT> &c) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    const int ku = a.k_user;
    if (row0 >= s && row0 + VEC <= e) {
        // whole chunk inside the group: 16-byte loads, all issued before first use
        V vx[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j)
            if (j < ku) vx[j] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + row0);
        const V vy = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0);
        V vw;
        if (a.w) vw = *reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T sw = a.w ? sqrt(vget<T>(vw, v)) : T(1);
            c.sw[v] = sw;
            c.y[v] = vget<T>(vy, v);
#pragma unroll
            for (int j = 0; j < KT; ++j) c.x[v][j] = (j < ku ? vget<T>(vx[j], v) : T(1)) * sw;
        }
    } else {
        // ragged head / tail of a group: guarded scalar loads, rows outside [s, e) contribute zeros
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t r = row0 + v;
            const bool in = (r >= s) && (r < e);
            const T sw = (in && a.w) ? sqrt(static_cast<const T *>(a.w)[r]) : T(1);
            c.sw[v] = sw;
            c.y[v] = in ? static_cast<const T *>(a.y)[r] : T(0);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                T xv = T(0);
                if (in) xv = (j < ku) ? static_cast<const T *>(a.x[j])[r] : T(1);
                c.x[v][j] = xv * sw;
            }
        }
    }
}

template <typename T, int KT>
__device__ __forceinline__ void gram_accumulate(T (&acc)[(KT + 1) * (KT + 2) / 2], const Chunk<T, KT> &c) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const T ys = c.y[v] * c.sw[v];
#pragma unroll
        for (int i = 0; i < KT; ++i) {
#pragma unroll
            for (int j = i; j < KT; ++j) acc[tri_index<NZ>(i, j)] = fma(c.x[v][i], c.x[v][j], acc[tri_index<NZ>(i, j)]);
            acc[tri_index<NZ>(i, KT)] = fma(c.x[v][i], ys, acc[tri_index<NZ>(i, KT)]);
        }
        acc[tri_index<NZ>(KT, KT)] = fma(ys, ys, acc[tri_index<NZ>(KT, KT)]);
    }
}

// Cholesky (LL^T) of G + alpha I and the two triangular solves, fully unrolled on wave-uniform values.
// Returns false on a non-positive pivot (faer's `cholesky(Side::Lower)` Err, ls.rs:289-299).
template <typename T, int KT>
__device__ __forceinline__ bool chol_solve(const T (&acc)[(KT + 1) * (KT + 2) / 2], T alpha, T (&beta)[KT]) {
    constexpr int NZ = KT + 1;
    T L[KT][KT];
    T rinv[KT];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        T d = acc[tri_index<NZ>(j, j)] + alpha;
#pragma unroll
        for (int p = 0; p < j; ++p) d = fma(-L[j][p], L[j][p], d);
        ok = ok && (d > T(0));
        const T dj = sqrt(d);
        rinv[j] = T(1) / dj;
        L[j][j] = dj;
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

template <typename T, int KT>
__device__ __forceinline__ void predict_store(const K1Args &a, const Chunk<T, KT> &c, const T (&beta)[KT], int64_t row0,
                                              int64_t s, int64_t e) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    T p[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) acc = fma(c.x[v][j], beta[j], acc);   // make_predictions on the FIT features (ex.rs:398-405)
        if (a.w) acc *= T(1) / c.sw[v];                                     // predictions *= 1/sqrt_w (ls.py:234-235)
        p[v] = acc;
    }
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    if (row0 >= s && row0 + VEC <= e) {
        if (pred) {
            V o;
            if constexpr (VEC == 4) o = V{p[0], p[1], p[2], p[3]}; else o = V{p[0], p[1]};
            *reinterpret_cast<V *>(pred + row0) = o;
        }
        if (resid) {
            V o;
            if constexpr (VEC == 4) o = V{c.y[0] - p[0], c.y[1] - p[1], c.y[2] - p[2], c.y[3] - p[3]};
            else o = V{c.y[0] - p[0], c.y[1] - p[1]};
            *reinterpret_cast<V *>(resid + row0) = o;
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int64_t r = row0 + v;
            if (r >= s && r < e) {
                if (pred) pred[r] = p[v];
                if (resid) resid[r] = c.y[v] - p[v];   // ORIGINAL target - predictions (ls.py:239)
            }
        }
    }
}

// TEAM = 64: four independent waves per 256-thread block, one group each, no LDS, no barriers.
// TEAM = 256: one group per block; cross-wave reduction through LDS with ONE barrier.
template <typename T, int KT, int TEAM, int RC>
__global__ void __launch_bounds__(256) k1_kernel(const K1Args a) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
    constexpr int NACC = NZ * (NZ + 1) / 2;
    constexpr int WAVES = TEAM / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (threadIdx.x >> 6) % WAVES;
    const int tid = threadIdx.x % TEAM;
    const int64_t g = (int64_t)blockIdx.x * (256 / TEAM) + threadIdx.x / TEAM;
    if (g >= a.n_groups) return;   // wave-uniform for TEAM=64; never taken for TEAM=256 (grid == n_groups)

    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);                      // chunk grid is aligned to 16 bytes in every column
    const int64_t nch = (e - base + VEC - 1) / VEC;

    Chunk<T, KT> res[RC];                                    // register-resident rows of this lane
    T acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = T(0);

#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const int64_t c = (int64_t)rc * TEAM + tid;
        if (c < nch) {
            load_chunk<T, KT>(a, base + c * VEC, s, e, res[rc]);
            gram_accumulate<T, KT>(acc, res[rc]);
        }
    }
    for (int64_t c = (int64_t)RC * TEAM + tid; c < nch; c += TEAM) {   // rows beyond register capacity: streamed
        Chunk<T, KT> tmp;
        load_chunk<T, KT>(a, base + c * VEC, s, e, tmp);
        gram_accumulate<T, KT>(acc, tmp);
    }

    // ---- team reduction, fixed order (deterministic)
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = wave_sum_row3(acc[q]);
    if constexpr (WAVES == 1) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = readlane63(acc[q]);
    } else {
        __shared__ T part[NACC * WAVES];
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) part[q * WAVES + wave] = acc[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            T t = part[q * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += part[q * WAVES + w];
            acc[q] = t;
        }
    }

    // ---- K x K solve on wave-uniform values
    T beta[KT];
    int st = POLS_GROUP_OK;
    if (e == s) {                       // features.is_empty() -> zeros (ex.rs:357-359)
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = T(0);
        st = POLS_GROUP_EMPTY;
    } else {
        const bool ok = chol_solve<T, KT>(acc, (T)a.alpha, beta);
        if (!ok) st = POLS_GROUP_FALLBACK;   // host re-dispatches this group to the fallback solver
    }
    if (tid == 0 && a.status) a.status[g] = st;
    if (a.coef && tid < KT) {
        T bv = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) bv = (tid == j) ? beta[j] : bv;
        static_cast<T *>(a.coef)[g * KT + tid] = bv;
    }

    // ---- fused predictions / residuals from the resident rows
    if (a.pred || a.resid) {
#pragma unroll
        for (int rc = 0; rc < RC; ++rc) {
            const int64_t c = (int64_t)rc * TEAM + tid;
            if (c < nch) predict_store<T, KT>(a, res[rc], beta, base + c * VEC, s, e);
        }
        for (int64_t c = (int64_t)RC * TEAM + tid; c < nch; c += TEAM) {
            Chunk<T, KT> tmp;
            load_chunk<T, KT>(a, base + c * VEC, s, e, tmp);
            predict_store<T, KT>(a, tmp, beta, base + c * VEC, s, e);
        }
    }
}

template <typename T, int KT, int TEAM, int RC>
static int k1_launch_variant(pols_ctx *ctx, const K1Args &a) {
    char name[96];
    std::snprintf(name, sizeof(name), "k1_gram_chol_%s_k%d_team%d_rc%d", sizeof(T) == 4 ? "f32" : "f64", KT, TEAM, RC);
    const int64_t teams_per_block = 256 / TEAM;
    const int64_t blocks = (a.n_groups + teams_per_block - 1) / teams_per_block;
    if (blocks > 0x7fffffffLL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    ctx->last_kernel = name;
    timing_begin(ctx);
    hipLaunchKernelGGL((k1_kernel<T, KT, TEAM, RC>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

#define K1_VARIANT(T, TN, KT, TEAM, RC) k1_launch_variant<T, KT, TEAM, RC>(ctx, a)

// Variant choice: smallest team whose registers hold the largest group (so X is read once); groups
// larger than the biggest variant stream their overflow rows twice (Gram pass + prediction pass).
template <typename T, int KT>
static int k1_launch_kt(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    constexpr int VEC = Vec16<T>::N;
    constexpr const char *TN = (sizeof(T) == 4) ? "f32" : "f64";
    (void)TN;
    if (max_rows <= 64 * 2 * VEC) {
        if constexpr (sizeof(T) == 4) return K1_VARIANT(T, "f32", KT, 64, 2); else return K1_VARIANT(T, "f64", KT, 64, 2);
    }
    if constexpr (sizeof(T) == 4) {
        if (max_rows <= 256 * 1 * VEC) return K1_VARIANT(T, "f32", KT, 256, 1);
        return K1_VARIANT(T, "f32", KT, 256, 2);
    } else {
        return K1_VARIANT(T, "f64", KT, 256, 2);
    }
}

template <typename T>
int k1_launch_t(pols_ctx *ctx, int kt, const K1Args &a, int64_t max_rows) {
    switch (kt) {
        case 1: return k1_launch_kt<T, 1>(ctx, a, max_rows);
        case 2: return k1_launch_kt<T, 2>(ctx, a, max_rows);
        case 3: return k1_launch_kt<T, 3>(ctx, a, max_rows);
        case 4: return k1_launch_kt<T, 4>(ctx, a, max_rows);
        case 5: return k1_launch_kt<T, 5>(ctx, a, max_rows);
        case 6: return k1_launch_kt<T, 6>(ctx, a, max_rows);
        case 7: return k1_launch_kt<T, 7>(ctx, a, max_rows);
        case 8: return k1_launch_kt<T, 8>(ctx, a, max_rows);
        case 9: return k1_launch_kt<T, 9>(ctx, a, max_rows);
        case 10: return k1_launch_kt<T, 10>(ctx, a, max_rows);
        default: return fail(POLS_ERR_UNSUPPORTED, "k1: %d features (incl. intercept) > %d", kt, K1_MAX_KT);
    }
}

}  // namespace pols
