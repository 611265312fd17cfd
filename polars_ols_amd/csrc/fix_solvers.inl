// fix_solvers.inl -- the solvers the fix-up passes (K6: up to 32 columns, K8: up to 1 024) run on a flagged group's f64 copy,
// chosen so that the group gets what the REFERENCE computes for it (FixMode): a 256-thread workgroup each.
#pragma once
#include "common.hpp"

namespace pols {

// What a flagged group is re-solved with: the solver the reference runs for the call's (branch, solve_method) on such a group.
enum FixMode : int32_t {
    FIX_MINNORM = 0,    // solve_ols_svd / solve_ridge_svd (ls.rs:106-191): one-sided Jacobi, singular values below rc_factor * s_max dropped
    FIX_OLS_AUTO = 1,   // solve_ols with solve_method = None (ls.rs:224-231): pivoted QR when the fit has more rows than columns, else SVD
    FIX_OLS_QR = 2,     // solve_ols_qr (ls.rs:195-205): column-pivoted Householder QR, BASIC solution on rank-deficient X (dependent
                        //   columns -> 0; notebooks/polars_ols_demo.ipynb cell 28 prints {1.0, 2.0, -0.0} where "svd" prints {1, 1, 1})
    FIX_CHOL_LU = 3,    // solve_ridge None / "chol" (ls.rs:352-363): Cholesky of X'X + alpha I in f64, on failure LU with partial pivoting --
                        //   an exactly singular matrix gives NaN (notebook cell 30), no cut-off of any kind
    FIX_LU = 4,         // solve_ridge "lu" (ls.rs:330-333): LU with partial pivoting
};
__host__ __device__ inline bool fix_uses_qr(int mode, double nfit, int kt) {
    // rows the null policy dropped are zero rows of the copy, so "more rows than columns" (ls.rs:224-229) is about the rows in the fit
    return (mode == FIX_OLS_AUTO && nfit > (double)kt) || (mode == FIX_OLS_QR && nfit >= (double)kt);
}
__host__ __device__ inline bool fix_uses_lu(int mode) { return mode == FIX_CHOL_LU || mode == FIX_LU; }

__device__ __forceinline__ double fix_wave_sum(double v) { return readlane63(wave_sum_row3(v)); }   // all 64 lanes active

// solve_ols_qr (ls.rs:195-205; faer col_piv_qr().solve_lstsq) on W = [kt feature columns | m target columns], n rows each, column
// major, in global memory: Householder QR with column pivoting (largest remaining column norm, the FIRST of tied columns),
// rank-revealing like LAPACK dgelsy -- the factorisation stops at the first pivot with |R_jj| <= eps max(n, k) |R_00|, those
// columns get coefficient 0 and the leading block is back-substituted: on the reference's own collinear frame (demo notebook
// cell 28: x3 an exact copy of x2) that is the printed {1.0, 2.0, -0.0}.
// One wave per trailing column (dot, update and the column's new norm in one pass, no workgroup barrier inside a step).
// cidx / cn: kt ints / doubles of LDS.  Result: out[t * kt + j] (any address space), t < m.  NaN data -> every coefficient NaN.
__device__ __forceinline__ void fix_qr_basic(double *W, const int64_t n, const int kt, const int m, int *cidx, double *cn, double *out) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = wv; c < kt; c += 4) {
        const double *col = W + (size_t)c * n;
        double s = 0.0;
        for (int64_t r = lane; r < n; r += 64) s += col[r] * col[r];
        s = fix_wave_sum(s);
        if (lane == 0) { cn[c] = s; cidx[c] = c; }
    }
    __syncthreads();
    const int steps = (int)(n < (int64_t)kt ? n : (int64_t)kt);
    int rank = steps;
    bool bad = false;
    double r00 = 0.0;
    const double rank_tol = 2.220446049250313e-16 * (double)(n > kt ? n : (int64_t)kt);
    for (int j = 0; j < steps; ++j) {
        int best = j;
        double bestn = -1.0;
        for (int c = j; c < kt; ++c) { const double s = cn[c]; bad = bad || (s != s); if (s > bestn) { bestn = s; best = c; } }
        if (bad) break;                                            // (block-uniform: every thread scanned the same LDS words)
        __syncthreads();
        if (tid == 0 && best != j) {
            const int t = cidx[j]; cidx[j] = cidx[best]; cidx[best] = t;
            const double u = cn[j]; cn[j] = cn[best]; cn[best] = u;
        }
        __syncthreads();
        const double normx = sqrt(bestn);
        if (j == 0) r00 = normx;
        if (normx <= rank_tol * r00) { rank = j; break; }
        double *pj = W + (size_t)cidx[j] * n;
        const double alpha = pj[j];
        const double bh = -copysign(normx, alpha), tau = (bh - alpha) / bh, scale = 1.0 / (alpha - bh);
        __syncthreads();                                           // everyone holds alpha before it is overwritten
        for (int64_t r = j + 1 + tid; r < n; r += 256) pj[r] *= scale;
        if (tid == 0) pj[j] = bh;
        __syncthreads();
        for (int c = j + 1 + wv; c < kt + m; c += 4) {             // H = I - tau v v' (v_j = 1) on the trailing columns and on the targets
            double *col = W + (size_t)(c < kt ? cidx[c] : c) * n;
            double d = 0.0;
            for (int64_t r = j + 1 + lane; r < n; r += 64) d += pj[r] * col[r];
            const double w = (fix_wave_sum(d) + col[j]) * tau;
            double nn = 0.0;
            for (int64_t r = j + 1 + lane; r < n; r += 64) { const double v = col[r] - w * pj[r]; col[r] = v; nn += v * v; }
            nn = fix_wave_sum(nn);
            if (lane == 0) { col[j] -= w; if (c < kt) cn[c] = nn; }
        }
        __syncthreads();
    }
    // R z = (Q'y)[:rank], column-oriented, in place in the target's first `rank` entries
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int t = 0; t < m; ++t) {
        double *yt = W + (size_t)(kt + t) * n;
        for (int i = rank - 1; i >= 0; --i) {
            const double *ri = W + (size_t)cidx[i] * n;
            const double zi = yt[i] / ri[i];
            __syncthreads();                                       // every thread holds z_i before yt[i] is overwritten
            for (int p = tid; p < i; p += 256) yt[p] -= ri[p] * zi;
            if (tid == 0) yt[i] = zi;
            __syncthreads();
        }
        for (int i = tid; i < kt; i += 256) out[(size_t)t * kt + cidx[i]] = bad ? qnan : (i < rank ? yt[i] : 0.0);
        __syncthreads();
    }
}

// solve_ridge "lu" / the LU fallback of None / "chol" (ls.rs:264-273, 358-363: faer partial_piv_lu().solve) on G (kt x kt, row
// major) with m right-hand sides B (kt x m, row major: B[i * m + t]), both in global memory or LDS, by the whole workgroup: row
// partial pivoting (first largest |entry|), no cut-off -- an exactly zero pivot divides through and the answer is NaN / inf, which
// is what the reference returns (notebook cell 30).  fcol: kt doubles of LDS.  The solution replaces B.
__device__ __forceinline__ void fix_lu_solve(double *G, double *B, const int kt, const int m, double *fcol, int *piv_s) {
    const int tid = threadIdx.x;
    for (int j = 0; j < kt; ++j) {
        if (tid == 0) {                                            // pivot search (kt <= 1 024 loads from one column: rare path)
            int p = j;
            double best = fabs(G[(size_t)j * kt + j]);
            for (int i = j + 1; i < kt; ++i) { const double v = fabs(G[(size_t)i * kt + j]); if (v > best) { best = v; p = i; } }
            *piv_s = p;
        }
        __syncthreads();
        const int p = *piv_s;
        if (p != j) {
            for (int c = tid; c < kt; c += 256) { const double t = G[(size_t)j * kt + c]; G[(size_t)j * kt + c] = G[(size_t)p * kt + c]; G[(size_t)p * kt + c] = t; }
            for (int t = tid; t < m; t += 256) { const double u = B[(size_t)j * m + t]; B[(size_t)j * m + t] = B[(size_t)p * m + t]; B[(size_t)p * m + t] = u; }
        }
        __syncthreads();
        const double d = G[(size_t)j * kt + j];
        for (int i = j + 1 + tid; i < kt; i += 256) fcol[i] = G[(size_t)i * kt + j] / d;
        __syncthreads();
        const int rem = kt - j - 1;
        for (int q = tid; q < rem * rem; q += 256) {
            const int i = j + 1 + q / rem, c = j + 1 + q % rem;
            G[(size_t)i * kt + c] -= fcol[i] * G[(size_t)j * kt + c];
        }
        for (int q = tid; q < rem * m; q += 256) {
            const int i = j + 1 + q / m, t = q % m;
            B[(size_t)i * m + t] -= fcol[i] * B[(size_t)j * m + t];
        }
        __syncthreads();
    }
    for (int i = kt - 1; i >= 0; --i) {                            // U x = y, column-oriented
        for (int t = tid; t < m; t += 256) B[(size_t)i * m + t] /= G[(size_t)i * kt + i];
        __syncthreads();
        for (int q = tid; q < i * m; q += 256) {
            const int p = q / m, t = q % m;
            B[(size_t)p * m + t] -= G[(size_t)p * kt + i] * B[(size_t)i * m + t];
        }
        __syncthreads();
    }
}

// G = W'W + alpha I (kt x kt) and B = W'Y (kt x m) from the f64 copy, one wave per entry: identical columns give bit-identical
// entries (as in the reference's GEMM), which is what makes an exactly collinear system hit an exactly zero pivot.
__device__ __forceinline__ void fix_gram(const double *W, const int64_t n, const int kt, const int m, const double alpha, double *G, double *B) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t npair = (int64_t)kt * (kt + 1) / 2, ntot = npair + (int64_t)kt * m;
    for (int64_t q = wv; q < ntot; q += 4) {
        int a, b;
        if (q >= npair) { a = (int)((q - npair) / m); b = kt + (int)((q - npair) % m); }
        else { int64_t t = q; a = 0; while (t >= kt - a) { t -= kt - a; ++a; } b = a + (int)t; }
        const double *ca = W + (size_t)a * n, *cb = W + (size_t)b * n;
        double s = 0.0;
        for (int64_t r = lane; r < n; r += 64) s += ca[r] * cb[r];
        s = fix_wave_sum(s);
        if (lane == 0) {
            if (b >= kt) B[(size_t)a * m + (b - kt)] = s;
            else if (a == b) G[(size_t)a * kt + a] = s + alpha;
            else { G[(size_t)a * kt + b] = s; G[(size_t)b * kt + a] = s; }
        }
    }
    __syncthreads();
}

// Cholesky of G (kt x kt) by thread 0, in place: L overwrites the strict lower triangle, its diagonal goes to dg[]; the upper
// triangle keeps G.  A pivot within 16 k eps of its diagonal entry is rounding noise around the exact 0 of a singular matrix and
// fails the factorisation (see api.hip).  On success the m right-hand sides B are solved in place; on failure G is restored.
__device__ __forceinline__ bool fix_chol_solve(double *G, double *B, const int kt, const int m, double *dg) {
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        bool ok = true;
        const double noise = 16.0 * (double)kt * 2.220446049250313e-16;
        for (int j = 0; j < kt && ok; ++j) {
            double d = G[(size_t)j * kt + j];
            const double gjj = d;
            for (int p = 0; p < j; ++p) d -= G[(size_t)j * kt + p] * G[(size_t)j * kt + p];
            if (!(d > noise * gjj)) { ok = false; break; }
            d = sqrt(d);
            dg[j] = d;
            for (int i = j + 1; i < kt; ++i) {
                double s = G[(size_t)i * kt + j];
                for (int p = 0; p < j; ++p) s -= G[(size_t)i * kt + p] * G[(size_t)j * kt + p];
                G[(size_t)i * kt + j] = s / d;
            }
        }
        if (ok) {
            for (int t = 0; t < m; ++t) {
                for (int i = 0; i < kt; ++i) {                     // L z = b, L' x = z
                    double s = B[(size_t)i * m + t];
                    for (int p = 0; p < i; ++p) s -= G[(size_t)i * kt + p] * B[(size_t)p * m + t];
                    B[(size_t)i * m + t] = s / dg[i];
                }
                for (int i = kt - 1; i >= 0; --i) {
                    double s = B[(size_t)i * m + t];
                    for (int p = i + 1; p < kt; ++p) s -= G[(size_t)p * kt + i] * B[(size_t)p * m + t];
                    B[(size_t)i * m + t] = s / dg[i];
                }
            }
        } else {
            for (int i = 0; i < kt; ++i)
                for (int c = 0; c < i; ++c) G[(size_t)i * kt + c] = G[(size_t)c * kt + i];
        }
        ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return ok_s != 0;
}

}  // namespace pols
