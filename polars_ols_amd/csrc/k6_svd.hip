// k6_svd.hip -- K6 "small_svd": fallback solver for the groups the fused static kernels flag (POLS_GROUP_FALLBACK).
//
// Replaces, for those groups only: solve_ols_svd (src/least_squares.rs:183-191, LAPACK dgelsd: minimum-norm least
// squares, singular values below eps * s_max dropped), solve_ridge_svd (:106-168, d = s / (s^2 + alpha)) and the
// Cholesky -> LU / SVD fallback chain of solve_normal_equations (:299-327).  A group is flagged when its Cholesky
// pivot is non-positive (rank-deficient X, n <= k, NaN data) or -- OLS only, where the reference itself uses a
// backward-stable QR -- when a pivot says cond(X)^2 would eat the tolerance of the normal-equation solve.
//
// One 256-thread workgroup per flagged group (a fixed pool of workers strides over all groups and skips the
// healthy ones): the group's sqrt(w)-scaled columns are copied as f64 into a per-worker global scratch, one-sided
// Jacobi (Hestenes) rotations orthogonalise the columns -- every rotation is three block reductions over n and two
// AXPYs, the accumulated V lives in LDS -- and beta = V diag(d) U' y.  One-sided Jacobi resolves small singular
// values to high RELATIVE accuracy, which is what lets it agree with dgelsd on nearly collinear columns; an
// eigen-decomposition of X'X could not.  Not bandwidth-critical: it runs on the rare groups only.
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

template <typename T>
__global__ void __launch_bounds__(256) k6_svd_kernel(const K6Args a) {
    __shared__ double V[K6_KMAX * K6_KMAX];
    __shared__ double sv[K6_KMAX], cj[K6_KMAX], beta[K6_KMAX];
    __shared__ double red[8];
    __shared__ int rotated;
    const int tid = threadIdx.x;
    const int kt = a.kt, ku = a.k_user;
    if (a.fb_flag && *a.fb_flag != a.epoch) return;             // nothing was flagged in this call (block-uniform)
    double *W = a.work + (size_t)blockIdx.x * a.work_stride;      // [kt + 1][n] column-major: scaled X columns, then scaled y

    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
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

int k6_launch(pols_ctx *ctx, int dtype, const K6Args &a, int workers) {
    if (a.kt > K6_KMAX) return fail(POLS_ERR_UNSUPPORTED, "svd fallback: %d features > %d", a.kt, K6_KMAX);
    if (dtype == POLS_F32) hipLaunchKernelGGL(k6_svd_kernel<float>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k6_svd_kernel<double>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
