// k6_body.inl -- device-side body of K6 (see k6_svd.hip).
#pragma once
#include "k6_svd.hpp"

namespace pols {

constexpr int K6_KMAX = 32;

__device__ __forceinline__ double k6_block_sum(double v, double *red) {   // red: 8 doubles of LDS
    v = wave_sum_row3(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) red[wv] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ double k6_wave_sum(double v) { return readlane63(wave_sum_row3(v)); }   // all 64 lanes active

// solve_ols_qr (ls.rs:195-205; faer col_piv_qr().solve_lstsq) on the worker's f64 copy W = [kt columns | y], n rows each:
// Householder QR with column pivoting (largest remaining column norm, the FIRST of tied columns), rank-revealing like LAPACK
// dgelsy -- the factorisation stops at the first pivot with |R_jj| <= eps max(n, k) |R_00|, those columns get coefficient 0, the
// leading block is back-substituted: on the reference's own collinear frame (demo notebook cell 28: x3 an exact copy of x2)
// that is the printed {1.0, 2.0, -0.0}.
// One wave per trailing column (dot, update and the column's new norm in one pass, no workgroup barrier inside a step).
// cidx / cn: kt ints / doubles of LDS; result in beta[] (LDS), every thread returns after a barrier.
__device__ __forceinline__ void k6_qr_basic(double *W, const int64_t n, const int kt, int *cidx, double *cn, double *beta) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int c = wv; c < kt; c += 4) {
        const double *col = W + (size_t)c * n;
        double s = 0.0;
        for (int64_t r = lane; r < n; r += 64) s += col[r] * col[r];
        s = k6_wave_sum(s);
        if (lane == 0) { cn[c] = s; cidx[c] = c; }
    }
    __syncthreads();
    const int steps = (int)(n < (int64_t)kt ? n : (int64_t)kt);
    int rank = steps;
    bool bad = false;                                              // NaN data: every coefficient NaN, like the reference's QR
    double r00 = 0.0;
    const double rank_tol = 2.220446049250313e-16 * (double)(n > kt ? n : (int64_t)kt);
    double *yv = W + (size_t)kt * n;
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
        for (int c = j + 1 + wv; c <= kt; c += 4) {                // H = I - tau v v' (v_j = 1) on the trailing columns and on y
            double *col = (c < kt) ? W + (size_t)cidx[c] * n : yv;
            double d = 0.0;
            for (int64_t r = j + 1 + lane; r < n; r += 64) d += pj[r] * col[r];
            const double w = (k6_wave_sum(d) + col[j]) * tau;
            double nn = 0.0;
            for (int64_t r = j + 1 + lane; r < n; r += 64) { const double v = col[r] - w * pj[r]; col[r] = v; nn += v * v; }
            nn = k6_wave_sum(nn);
            if (lane == 0) { col[j] -= w; if (c < kt) cn[c] = nn; }
        }
        __syncthreads();
    }
    // R z = (Q'y)[:rank] on wave 0: lane p keeps z_p
    if (wv == 0) {
        double zl = 0.0;
        for (int i = rank - 1; i >= 0; --i) {
            const double rip = (lane > i && lane < rank) ? W[(size_t)cidx[lane] * n + i] * zl : 0.0;
            const double zi = (yv[i] - k6_wave_sum(rip)) / W[(size_t)cidx[i] * n + i];
            if (lane == i) zl = zi;
        }
        if (lane < kt) beta[cidx[lane]] = bad ? __longlong_as_double(0x7ff8000000000000LL) : (lane < rank ? zl : 0.0);
    }
    __syncthreads();
}

// solve_ridge None / "chol" / "lu" (ls.rs:342-364 -> solve_normal_equations :277-337) on the worker's f64 copy: G = X'X + alpha I and
// X'y with one wave per entry (identical columns give bit-identical entries, as in the reference's GEMM), then thread 0 runs the
// reference's chain in f64: Cholesky (a pivot that is not > 0 fails it, faer) -> LU with row partial pivoting, no cut-off --
// an exactly singular system divides by its zero pivot and returns NaN, which is what the reference prints (notebook cell 30).
// G: kt x kt doubles of LDS, bv / beta: kt doubles.
__device__ __forceinline__ void k6_chol_lu(const double *W, const int64_t n, const int kt, const double alpha, const bool try_chol,
                                           double *G, double *bv, double *beta) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int npair = kt * (kt + 1) / 2 + kt;
    for (int q = wv; q < npair; q += 4) {
        int a, b;
        if (q < kt) { a = q; b = kt; }                              // X'y
        else { int t = q - kt; a = 0; while (t >= kt - a) { t -= kt - a; ++a; } b = a + t; }
        const double *ca = W + (size_t)a * n, *cb = W + (size_t)b * n;
        double s = 0.0;
        for (int64_t r = lane; r < n; r += 64) s += ca[r] * cb[r];
        s = k6_wave_sum(s);
        if (lane == 0) {
            if (b == kt) bv[a] = s;
            else { G[a * kt + b] = s + (a == b ? alpha : 0.0); G[b * kt + a] = s; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        bool solved = false;
        if (try_chol) {                                            // in place: L overwrites the strict lower triangle, its diagonal lives
            // in beta[]; the upper triangle keeps G for the LU fallback
            bool ok = true;
            const double noise = 16.0 * (double)kt * 2.220446049250313e-16;   // a pivot within rounding noise of 0 fails (see api.hip)
            for (int j = 0; j < kt && ok; ++j) {
                double d = G[j * kt + j];
                const double gjj = d;
                for (int p = 0; p < j; ++p) d -= G[j * kt + p] * G[j * kt + p];
                if (!(d > noise * gjj)) { ok = false; break; }
                d = sqrt(d);
                beta[j] = d;
                for (int i = j + 1; i < kt; ++i) {
                    double s = G[i * kt + j];
                    for (int p = 0; p < j; ++p) s -= G[i * kt + p] * G[j * kt + p];
                    G[i * kt + j] = s / d;
                }
            }
            if (ok) {
                for (int i = 0; i < kt; ++i) {                     // L z = b, L' x = z
                    double s = bv[i];
                    for (int p = 0; p < i; ++p) s -= G[i * kt + p] * bv[p];
                    bv[i] = s / beta[i];
                }
                for (int i = kt - 1; i >= 0; --i) {
                    double s = bv[i];
                    for (int p = i + 1; p < kt; ++p) s -= G[p * kt + i] * bv[p];
                    bv[i] = s / beta[i];
                }
                for (int i = 0; i < kt; ++i) beta[i] = bv[i];
                solved = true;
            } else {                                               // restore the symmetric matrix from its untouched upper triangle
                for (int i = 0; i < kt; ++i)
                    for (int c = 0; c < i; ++c) G[i * kt + c] = G[c * kt + i];
            }
        }
        if (!solved) {                                             // faer partial_piv_lu().solve (ls.rs:264-273)
            for (int j = 0; j < kt; ++j) {
                int p = j;
                double best = fabs(G[j * kt + j]);
                for (int i = j + 1; i < kt; ++i)
                    if (fabs(G[i * kt + j]) > best) { best = fabs(G[i * kt + j]); p = i; }
                if (p != j) {
                    for (int c = 0; c < kt; ++c) { const double t = G[j * kt + c]; G[j * kt + c] = G[p * kt + c]; G[p * kt + c] = t; }
                    const double t = bv[j]; bv[j] = bv[p]; bv[p] = t;
                }
                const double d = G[j * kt + j];
                for (int i = j + 1; i < kt; ++i) {
                    const double f = G[i * kt + j] / d;
                    for (int c = j + 1; c < kt; ++c) G[i * kt + c] -= f * G[j * kt + c];
                    bv[i] -= f * bv[j];
                }
            }
            for (int i = kt - 1; i >= 0; --i) {
                double s = bv[i];
                for (int p = i + 1; p < kt; ++p) s -= G[i * kt + p] * bv[p];
                bv[i] = s / G[i * kt + i];
            }
            for (int i = 0; i < kt; ++i) beta[i] = bv[i];
        }
    }
    __syncthreads();
}

// The fix-up pass as a device function: worker `worker` of `n_workers` 256-thread workgroups strides over the groups and
// re-solves the flagged ones.
template <typename T>
__device__ __forceinline__ void k6_process(const K6Args &a, const int worker, const int n_workers) {
    __shared__ double V[K6_KMAX * K6_KMAX];
    __shared__ double sv[K6_KMAX], cj[K6_KMAX], beta[K6_KMAX];
    __shared__ double red[8];
    __shared__ int rotated;
    __shared__ int cidx[K6_KMAX];
    const int tid = threadIdx.x;
    const int kt = a.kt, ku = a.k_user;
    if (a.fb_flag && *a.fb_flag != a.epoch) return;               // nothing was flagged in this call (block-uniform)
    double *W = a.work + (size_t)worker * a.work_stride;          // [kt + 1][n] column-major: scaled X columns, then scaled y

    for (int64_t g = worker; g < a.n_groups; g += n_workers) {
        if (a.status[g] != POLS_GROUP_FALLBACK) continue;         // block-uniform
        const int64_t s = a.offs[g], e = a.offs[g + 1];
        const int64_t n = e - s;
        // ---- copy the group as f64, sqrt(w)-scaled, intercept appended last (least_squares.py:184-196)
        const int pol = a.null_policy;
        int nfit_l = 0;
        for (int64_t r = tid; r < n; r += 256) {
            const bool in_fit = null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, s + r);   // dropped rows become zero rows
            nfit_l += in_fit ? 1 : 0;
            const double sw = !in_fit ? 0.0 : (a.w ? sqrt((double)static_cast<const T *>(a.w)[s + r]) : 1.0);
            for (int j = 0; j < kt; ++j) {
                const double x = (j < ku) ? (double)null_fill<T>(pol, static_cast<const T *>(a.x[j])[s + r]) : 1.0;
                W[(size_t)j * n + r] = in_fit ? x * sw : 0.0;
            }
            W[(size_t)kt * n + r] = in_fit ? (double)null_fill<T>(pol, static_cast<const T *>(a.y)[s + r]) * sw : 0.0;
        }
        const double nfit = k6_block_sum((double)nfit_l, red);
        if (nfit == 0.0 && tid == 0) a.status[g] = POLS_GROUP_EMPTY;   // every row dropped by the null policy: zeros, like an empty group
        // ---- the solver the reference runs on this group (K6Mode); rows the null policy dropped are zero rows of W, so "more rows
        //      than columns" (ls.rs:224-229) is about the rows in the fit
        const bool use_qr = (a.mode == K6_OLS_AUTO && nfit > (double)kt) || (a.mode == K6_OLS_QR && nfit >= (double)kt);
        const bool use_lu = a.mode == K6_CHOL_LU || a.mode == K6_LU;
        if (use_qr || use_lu) {
            __syncthreads();
            if (use_qr) k6_qr_basic(W, n, kt, cidx, cj, beta);
            else k6_chol_lu(W, n, kt, a.alpha, a.mode == K6_CHOL_LU, V, cj, beta);
            if (tid < kt) {
                if (nfit == 0.0) beta[tid] = 0.0;
                if (a.coef) static_cast<T *>(a.coef)[g * kt + tid] = (T)beta[tid];
            }
            __syncthreads();
        } else {
        for (int q = tid; q < kt * kt; q += 256) V[q] = ((q / kt) == (q % kt)) ? 1.0 : 0.0;
        __syncthreads();
        // ---- one-sided Jacobi sweeps
        for (int sweep = 0; sweep < 60; ++sweep) {
            if (tid == 0) rotated = 0;
            for (int p = 0; p < kt - 1; ++p) {
                for (int q = p + 1; q < kt; ++q) {
                    double *wp = W + (size_t)p * n, *wq = W + (size_t)q * n;
                    double pa = 0.0, pb = 0.0, pg = 0.0;
                    for (int64_t r = tid; r < n; r += 256) { const double u = wp[r], v = wq[r]; pa += u * u; pb += v * v; pg += u * v; }
                    const double al = k6_block_sum(pa, red), be = k6_block_sum(pb, red), ga = k6_block_sum(pg, red);
                    if (!(ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be))) {      // NaN data also lands here
                        const double zeta = (be - al) / (2.0 * ga);
                        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                        for (int64_t r = tid; r < n; r += 256) {
                            const double u = wp[r], v = wq[r];
                            wp[r] = c * u - sn * v;
                            wq[r] = sn * u + c * v;
                        }
                        if (tid < kt) {
                            const double u = V[tid * kt + p], v = V[tid * kt + q];
                            V[tid * kt + p] = c * u - sn * v;
                            V[tid * kt + q] = sn * u + c * v;
                        }
                        if (tid == 0) rotated = (ga == ga) ? 1 : 0;                  // NaN: give up after this sweep
                    }
                    __syncthreads();
                }
            }
            __syncthreads();
            if (!rotated) break;
        }
        // ---- singular values, coefficients  beta = V diag(d) U' y,  U s = W
        double smax = 0.0;
        for (int j = 0; j < kt; ++j) {
            double pa = 0.0, pd = 0.0;
            const double *wj = W + (size_t)j * n, *yy = W + (size_t)kt * n;
            for (int64_t r = tid; r < n; r += 256) { pa += wj[r] * wj[r]; pd += wj[r] * yy[r]; }
            const double nn = k6_block_sum(pa, red), dot = k6_block_sum(pd, red);
            if (tid == 0) { sv[j] = sqrt(nn); cj[j] = dot; }
            smax = fmax(smax, sqrt(nn));
            if (nn != nn) smax = nn;                                              // NaN propagates to every coefficient
        }
        __syncthreads();
        if (tid < kt) {
            const double cutoff = a.rc_factor * smax;
            double acc = 0.0;
            for (int j = 0; j < kt; ++j) {
                const double sj = sv[j];
                double d;
                if (a.alpha > 0.0) { const double sz = (sj < cutoff) ? 0.0 : sj; d = sz / (sz * sz + a.alpha); }   // :143-148
                else d = (sj > cutoff && sj > 0.0) ? 1.0 / sj : 0.0;                                                // dgelsd
                const double coef = (sj > 0.0) ? d * cj[j] / sj : 0.0;
                acc += V[tid * kt + j] * ((smax != smax) ? smax : coef);
            }
            if (nfit == 0.0) acc = 0.0;
            beta[tid] = acc;
            if (a.coef) static_cast<T *>(a.coef)[g * kt + tid] = (T)acc;
        }
        __syncthreads();
        }
        // ---- predictions / residuals for this group (make_predictions on the fit features, ex.rs:398-405)
        if (a.pred || a.resid) {
            for (int64_t r = tid; r < n; r += 256) {
                const double sw = a.w ? sqrt((double)static_cast<const T *>(a.w)[s + r]) : 1.0;
                double p = 0.0;
                for (int j = 0; j < kt; ++j) {
                    const double x = (j < ku) ? (double)null_fill<T>(pol, static_cast<const T *>(a.x[j])[s + r]) : 1.0;
                    p += (x * sw) * beta[j];
                }
                if (a.w) p *= 1.0 / sw;
                if (pol == POLS_NULL_DROP) p = nan_if<double>(null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, s + r) ? 0u : 1u, p);
                if (a.pred) static_cast<T *>(a.pred)[s + r] = (T)p;
                if (a.resid) static_cast<T *>(a.resid)[s + r] = (T)((double)static_cast<const T *>(a.y)[s + r] - p);
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k6_svd_kernel(const K6Args a) {
    k6_process<T>(a, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace pols
