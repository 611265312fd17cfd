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
        if (k6s_takes(a.mode, n, kt, a.small_rows)) continue;     // K6s's (block-uniform)
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
        // ---- the solver the reference runs on this group (FixMode)
        const bool use_qr = fix_uses_qr(a.mode, nfit, kt), use_lu = fix_uses_lu(a.mode);
        if (use_qr || use_lu) {
            __syncthreads();
            if (use_qr) fix_qr_basic(W, n, kt, 1, cidx, cj, beta);
            else {                                                 // V: the Gram matrix, cj: X'y -> the solution, sv: factor diagonal / LU multipliers
                fix_gram(W, n, kt, 1, a.alpha, V, cj);
                if (!(a.mode == FIX_CHOL_LU && fix_chol_solve(V, cj, kt, 1, sv))) fix_lu_solve(V, cj, kt, 1, sv, &rotated);
                if (tid < kt) beta[tid] = cj[tid];
                __syncthreads();
            }
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
            const double rcf = a.rc_factor < 0.0 ? 2.220446049250313e-16 * fmax(nfit, (double)kt) : a.rc_factor;   // eps * max(n, k)
            const double cutoff = rcf * smax;
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
