// k8_wide.hip -- K8: OLS / ridge / elastic net for 32 .. 1024 columns per group.
//
// The reference's own wide cases -- tests/benchmark.py (10 000 x 100), test_elastic_net (up to 1 000 features),
// test_fit_wide (10 rows x up to 1 000 features) -- go through the same dispatcher (src/expressions.rs:351-388) as the
// narrow ones; here the Gram matrix no longer fits a wave, so every step is a workgroup-sized kernel over matrices in
// HBM / L2, and the feature columns arrive through a device table of pointers instead of a fixed kernel-argument array:
//   gram      Z'Z, Z = [sqrt(w) X | sqrt(w) 1 | sqrt(w) y], in 64 x 64 tiles on the f64 matrix cores (16x16x4 MFMA):
//             one workgroup per (tile pair, group, row split); 64-row chunks staged through LDS as f64 with the sqrt(w)
//             scaling applied on the way in; partial Grams per row split are summed in a fixed order (no atomics, so the
//             result is run-to-run identical) by `reduce`, which also mirrors the lower triangle.
//   chol      solve_ols / solve_ridge on the normal equations: right-looking Cholesky in place, the scaled column kept
//             in LDS for the rank-1 trailing update, then the two triangular solves.  Same pivot test as the narrow
//             kernels; what it cannot factor is flagged (POLS_GROUP_FALLBACK).
//   minnorm   flagged groups (n < k as in test_fit_wide, collinear columns as in test_fit_multi_collinear): minimum-norm
//             solution through one-sided Jacobi on whichever of X / X' has fewer columns (dgelsd semantics, ls.rs:183-191;
//             ridge d = s / (s^2 + alpha), :143-148).
//   cd        solve_elastic_net (ls.rs:386-492) in Gram form: per coordinate one workgroup-wide dot of a Gram row with
//             the coefficient vector in LDS; same order, alpha * n scaling, soft threshold, active set and stop rule.
//   predict   X . beta (+ residuals) with the reference's weighted arithmetic.
#include "k8_wide.hpp"
#include "fix_solvers.inl"
#include "k1m_kernel.inl"   // Mfma16
#include "k7_stats.hpp"     // k7_betai

namespace pols {

constexpr int WG_TS = 64;        // tile side (columns of Z)
constexpr int WG_RS = 65;        // LDS row stride of a staged column (doubles)

template <typename T>
__device__ __forceinline__ double wide_z(const WideArgs &a, const void *colp, int z, int64_t r) {   // unscaled Z[r][z]
    const int ku = a.k_user, kt = a.kt;
    if (z < ku) return (double)null_fill<T>(a.null_policy, static_cast<const T *>(colp)[r]);
    if (z >= kt) return (double)null_fill<T>(a.null_policy, static_cast<const T *>(a.ycols ? a.ycols[z - kt] : a.y)[r]);   // target z - kt
    if (z == kt - 1 && ku != kt) return 1.0;
    return 0.0;
}

// ------------------------------------------------------------------------------------------------ row mask (null policies)
// compute_is_valid_mask (ex.rs:201-228) over ALL columns of a row: one workgroup per group, a thread per row, the columns
// walked one after the other (coalesced across the threads).  Also counts the fit rows of the group (the n of alpha * n).
template <typename T>
__global__ void __launch_bounds__(256) wide_rowmask_kernel(const WideArgs a) {
    __shared__ int cnt_s;
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int pol = a.null_policy;
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    int cnt = 0;
    for (int64_t r = s + threadIdx.x; r < e; r += 256) {
        bool fit = true;
        if (null_checks_y(pol)) {
            const T yv = static_cast<const T *>(a.y)[r];
            fit = (yv == yv) && !(a.valid && !a.valid[r]);
            if (fit && null_checks_x(pol))
                for (int j = 0; j < a.k_user && fit; ++j) { const T xv = static_cast<const T *>(a.cols[j])[r]; fit = (xv == xv); }
        }
        a.rowmask[r] = fit ? 1 : 0;
        cnt += fit ? 1 : 0;
    }
    if (cnt) atomicAdd(&cnt_s, cnt);
    __syncthreads();
    if (threadIdx.x == 0) a.nfit[g] = (double)cnt_s;
}

int wide_rowmask_launch(pols_ctx *ctx, int dtype, const WideArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_rowmask_kernel<float>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(wide_rowmask_kernel<double>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ gram
template <typename T>
__global__ void __launch_bounds__(256) wide_gram_kernel(const WideArgs a) {
    using M = Mfma16<double>;
    __shared__ double Zi[WG_TS * WG_RS], Zj[WG_TS * WG_RS];
    __shared__ double sw_s[WG_TS];
    __shared__ const void *cp_i[WG_TS], *cp_j[WG_TS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int NZ = a.kt + wide_m(a), nt = (NZ + WG_TS - 1) / WG_TS;
    int ti = 0, rem = blockIdx.x;                                  // pair index -> (ti <= tj)
    while (rem >= nt - ti) { rem -= nt - ti; ++ti; }
    const int tj = ti + rem;
    const int64_t g = blockIdx.y;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t r_begin = s + (int64_t)blockIdx.z * a.rows_per_split;
    const int64_t r_end = min(e, r_begin + a.rows_per_split);
    if (r_begin >= r_end) return;                                  // `reduce` only sums the splits that hold rows
    if (tid < WG_TS) {
        const int zi = WG_TS * ti + tid, zj = WG_TS * tj + tid;
        cp_i[tid] = zi < a.k_user ? a.cols[zi] : nullptr;
        cp_j[tid] = zj < a.k_user ? a.cols[zj] : nullptr;
    }
    M::acc_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = M::acc_t{0, 0, 0, 0};
    const double *ZJ = (ti == tj) ? Zi : Zj;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += WG_TS) {
        const int rows_here = (int)min((int64_t)WG_TS, r_end - r0);
        __syncthreads();                                           // previous chunk consumed; pointer table visible
        if (tid < WG_TS) {
            double swv = tid < rows_here ? (a.w ? sqrt((double)static_cast<const T *>(a.w)[r0 + tid]) : 1.0) : 0.0;
            if (a.rowmask && tid < rows_here && !a.rowmask[r0 + tid]) swv = 0.0;      // dropped by the null policy
            sw_s[tid] = swv;
        }
        __syncthreads();
        for (int idx = tid; idx < WG_TS * WG_TS; idx += 256) {
            const int c = idx >> 6, r = idx & 63;
            const bool in = r < rows_here;
            const int zi = WG_TS * ti + c;
            Zi[c * WG_RS + r] = (in && zi < NZ) ? wide_z<T>(a, cp_i[c], zi, r0 + r) * sw_s[r] : 0.0;
            if (ti != tj) {
                const int zj = WG_TS * tj + c;
                Zj[c * WG_RS + r] = (in && zj < NZ) ? wide_z<T>(a, cp_j[c], zj, r0 + r) * sw_s[r] : 0.0;
            }
        }
        __syncthreads();
        const int zc = lane & 15, kq = lane >> 4;
        const double *ap = Zi + (16 * wv + zc) * WG_RS + kq;
        const double *bp = ZJ + zc * WG_RS + kq;
#pragma unroll 4
        for (int rr = 0; rr < WG_TS; rr += 4) {
            const double av = ap[rr];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = M::mma(av, bp[16 * t * WG_RS + rr], acc[t]);
        }
    }
    double *P = a.partial + ((size_t)blockIdx.z * a.n_groups + g) * NZ * NZ;
    const int dcol = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = WG_TS * ti + 16 * wv + (lane >> 4) + 4 * reg, j = WG_TS * tj + 16 * t + dcol;
            if (i < NZ && j < NZ) P[(size_t)i * NZ + j] = acc[t][reg];
        }
}

__global__ void __launch_bounds__(256) wide_reduce_kernel(const WideArgs a) {
    const int NZ = a.kt + wide_m(a);
    const int64_t g = blockIdx.y;
    const int64_t n = a.offs[g + 1] - a.offs[g];
    const int nsp = (int)((n + a.rows_per_split - 1) / a.rows_per_split);
    const size_t mat = (size_t)NZ * NZ;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < mat; q += (size_t)gridDim.x * 256) {
        int i = (int)(q / NZ), j = (int)(q - (size_t)i * NZ);
        if (i / WG_TS > j / WG_TS) { const int t = i; i = j; j = t; }        // only tile pairs ti <= tj were computed
        double v = 0.0;
        for (int sp = 0; sp < nsp; ++sp) v += a.partial[((size_t)sp * a.n_groups + g) * mat + (size_t)i * NZ + j];
        a.gram[(size_t)g * mat + q] = v;
    }
}

int wide_gram_launch(pols_ctx *ctx, int dtype, const WideArgs &a) {
    const int NZ = a.kt + wide_m(a), nt = (NZ + WG_TS - 1) / WG_TS, npairs = nt * (nt + 1) / 2;
    char name[64];
    std::snprintf(name, sizeof(name), "k8_wide_gram_%s_k%d", dtype == POLS_F32 ? "f32" : "f64", a.kt);
    ctx->last_kernel = name;
    timing_begin(ctx);
    const dim3 grid((unsigned)npairs, (unsigned)a.n_groups, (unsigned)a.splits);
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_gram_kernel<float>, grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(wide_gram_kernel<double>, grid, dim3(256), 0, ctx->stream, a);
    const unsigned rb = (unsigned)std::min<size_t>(1024, ((size_t)NZ * NZ + 255) / 256);
    hipLaunchKernelGGL(wide_reduce_kernel, dim3(rb, (unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ block reductions
template <int NV>
__device__ __forceinline__ void wide_block_sum(double (&v)[NV], double *red, int nwaves) {   // red: NV * nwaves doubles
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double sacc = wave_sum_row3(v[i]);
        if (lane == 63) red[i * nwaves + wv] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int w = 0; w < nwaves; ++w) t += red[i * nwaves + w];
        v[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------ chol
// Normal equations by a square-root-free Cholesky (L D L') of the AUGMENTED matrix [A b; b' .]: carrying the target row
// through the trailing updates makes the forward substitution free, and leaving the columns unscaled means one workgroup
// barrier per column instead of two -- the factorisation of a k-column group is 2 k barriers in all, which is what
// bounds a single small problem (the reference's own benchmark shape: ONE 10 000 x 100 group).  d_j are the squared
// Cholesky pivots, so the pivot test is the narrow kernels' test.  Up to 127 columns the matrix lives in LDS.
template <typename T, int NTHREADS, bool IN_LDS>
__global__ void __launch_bounds__(NTHREADS) wide_chol_kernel(const WideArgs a) {
    extern __shared__ double a_lds[];
    constexpr int KCAP = IN_LDS ? 128 : K8_KMAX;                   // the LDS-resident variant is only launched below 128 columns
    __shared__ double xs[KCAP], dinv[KCAP], diag0[KCAP];
    __shared__ int ok_s;
    constexpr int NW = NTHREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int kt = a.kt, m = wide_m(a), NZ = kt + m;
    const int64_t g = blockIdx.x;
    const int64_t n = a.nfit ? (int64_t)a.nfit[g] : a.offs[g + 1] - a.offs[g];
    double *Gm = a.gram + (size_t)g * NZ * NZ;
    const int LD = IN_LDS ? (NZ | 1) : NZ;
    double *A = IN_LDS ? a_lds : Gm;
    if (IN_LDS) {
        for (int q = tid; q < NZ * NZ; q += NTHREADS) { const int i = q / NZ, c = q - i * NZ; if (c <= i) A[i * LD + c] = Gm[q]; }
        __syncthreads();
    }
    for (int i = tid; i < kt; i += NTHREADS) { const double d0 = A[(size_t)i * LD + i] + a.alpha; A[(size_t)i * LD + i] = d0; diag0[i] = d0; }
    if (tid == 0) ok_s = 1;
    __syncthreads();
    for (int j = 0; j < kt; ++j) {
        const double d = A[(size_t)j * LD + j];                   // final: column j - 1's update was the last to touch it
        const double di = 1.0 / d;
        if (tid == 0) { dinv[j] = di; if (!(d > a.pivot_tol * diag0[j])) ok_s = 0; }   // d_j / G_jj: see chol_solve (k1_kernel.inl)
        const double *cj = A + j;                                  // column j: cj[i * LD]
        for (int i = j + 1 + wv; i < NZ; i += NW) {                // rows j+1 .. NZ-1 (rows kt .. are the target rows)
            const double li = cj[(size_t)i * LD] * di;
            double *row = A + (size_t)i * LD;
            const int cmax = (i >= kt) ? kt - 1 : i;
            for (int c = j + 1 + lane; c <= cmax; c += 64) row[c] -= li * cj[(size_t)c * LD];
        }
        __syncthreads();
    }
    // A[kt + t][j] = d_j z_j with z the forward solution of target t; back substitution
    //   x_j = (A[kt + t][j] - sum_{i > j} A[i][j] x_i) / d_j
    // fewer fit rows than columns under solve_method None / "svd": the reference takes the SVD by SHAPE (ls.rs:224-231) -- flagged whatever
    // the pivots of this singular matrix happened to round to (alpha > 0, the ridge branch, is positive definite and keeps its answer)
    const bool by_shape = n < (int64_t)kt && a.alpha == 0.0 && (a.fix_mode == FIX_OLS_AUTO || a.fix_mode == FIX_MINNORM);
    int st = POLS_GROUP_OK;
    if (n == 0) st = POLS_GROUP_EMPTY;
    else if (!ok_s || by_shape) { st = POLS_GROUP_FALLBACK; if (tid == 0 && a.fb_flag) *a.fb_flag = a.epoch; }
    for (int t = 0; t < m; ++t) {
        __syncthreads();
        for (int i = tid; i < kt; i += NTHREADS) xs[i] = A[(size_t)(kt + t) * LD + i];
        __syncthreads();
        for (int j = kt - 1; j >= 0; --j) {
            const double xj = xs[j] * dinv[j];                     // every thread: xs[j] is final after the last barrier
            const double *row = A + (size_t)j * LD;
            for (int i = tid; i < j; i += NTHREADS) xs[i] -= row[i] * xj;
            __syncthreads();
            if (tid == 0) xs[j] = xj;
        }
        __syncthreads();
        for (int i = tid; i < kt; i += NTHREADS) {
            const double out = (n == 0) ? 0.0 : xs[i];
            if (a.coef) static_cast<T *>(a.coef)[((size_t)g * m + t) * kt + i] = (T)out;
            a.coef64[((size_t)g * m + t) * kt + i] = out;
        }
    }
    if (tid == 0 && a.status) a.status[g] = st;
}

template <typename T>
static int wide_chol_launch_t(pols_ctx *ctx, const WideArgs &a) {
    const int NZ = a.kt + wide_m(a);
    if (NZ <= 128) {
        const size_t lds = sizeof(double) * (size_t)NZ * (NZ | 1);
        static OncePerDevice attr_once;
        if (attr_once.needed(ctx->device)) {
            POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_chol_kernel<T, 1024, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
            attr_once.done(ctx->device);
        }
        hipLaunchKernelGGL((wide_chol_kernel<T, 1024, true>), dim3((unsigned)a.n_groups), dim3(1024), lds, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((wide_chol_kernel<T, 1024, false>), dim3((unsigned)a.n_groups), dim3(1024), 0, ctx->stream, a);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int wide_chol_launch(pols_ctx *ctx, int dtype, const WideArgs &a) {
    return dtype == POLS_F32 ? wide_chol_launch_t<float>(ctx, a) : wide_chol_launch_t<double>(ctx, a);
}

// ------------------------------------------------------------------------------------------------ minnorm
// Flagged groups: minimum-norm least squares (LAPACK dgelsd semantics of solve_ols_svd, ls.rs:183-191; ridge
// d = s / (s^2 + alpha), :143-148) by one-sided Jacobi on whichever side is smaller:
//   primal (n >= k): the k columns of X (length n) are orthogonalised, X V = U S  =>  beta = V g,  g_c = (w_c . y) / (s_c^2 + alpha)
//   dual   (n <  k): the n columns of X' (length k),            X' V = U S  =>  beta = W g,  g_c = (v_c . y) / (s_c^2 + alpha)
// with w_c the rotated columns, and g_c = 0 below the cut-off.  W, V and the small vectors live in a per-worker area in HBM.
template <typename T>
__global__ void __launch_bounds__(256) wide_svd_kernel(const WideArgs a) {
    __shared__ double red[3 * 4];
    __shared__ int rotated;
    __shared__ int cidx[K8_KMAX];                                  // pivoted QR: column order; LU: nothing
    __shared__ double cn[K8_KMAX];                                 // pivoted QR: trailing column norms; LU: multipliers / factor diagonal
    const int tid = threadIdx.x;
    const int kt = a.kt;
    if (a.fb_flag && *a.fb_flag != a.epoch) return;
    double *W = a.work + (size_t)blockIdx.x * a.work_stride;
    double *Vm = W + a.work_w_elems;
    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
        if (a.status[g] != POLS_GROUP_FALLBACK) continue;
        const int64_t s = a.offs[g];
        const int n = (int)(a.offs[g + 1] - s);
        // the solver the reference runs on this group (FixMode): pivoted QR / Cholesky -> LU work on the columns of X (primal copy)
        const double nfit_g = a.nfit ? a.nfit[g] : (double)n;
        const bool use_qr = fix_uses_qr(a.fix_mode, nfit_g, kt), use_lu = fix_uses_lu(a.fix_mode);
        const bool dual = n < kt && !use_qr && !use_lu;
        const int nc = dual ? n : kt, len = dual ? kt : n;
        const int m = wide_m(a);
        double *yv = W + (size_t)nc * len;                         // the scaled targets, m x n values
        double *s2 = Vm + (size_t)nc * nc, *gsc = s2 + nc;
        for (int r = 0; r < n; r += 1) {
            // one row at a time keeps the sqrt(w) and target reads trivial; the feature reads are strided either way
            double sw = a.w ? sqrt((double)static_cast<const T *>(a.w)[s + r]) : 1.0;
            if (a.rowmask && !a.rowmask[s + r]) sw = 0.0;                // dropped rows become zero rows
            for (int j = tid; j < kt; j += 256) {
                const double z = wide_z<T>(a, j < a.k_user ? a.cols[j] : nullptr, j, s + r) * sw;
                if (dual) W[(size_t)r * len + j] = z; else W[(size_t)j * len + r] = z;
            }
            if (tid < m) yv[(size_t)tid * n + r] = wide_z<T>(a, nullptr, kt + tid, s + r) * sw;
        }
        if (use_qr || use_lu) {
            __syncthreads();
            double *cout = a.coef64 + (size_t)g * m * kt;
            if (use_qr) fix_qr_basic(W, n, kt, m, cidx, cn, cout);
            else {                                                 // Vm: G (kt x kt) then B (kt x m)
                double *Bm = Vm + (size_t)kt * kt;
                fix_gram(W, n, kt, m, a.alpha, Vm, Bm);
                if (!(a.fix_mode == FIX_CHOL_LU && fix_chol_solve(Vm, Bm, kt, m, cn))) fix_lu_solve(Vm, Bm, kt, m, cn, &rotated);
                for (int q = tid; q < kt * m; q += 256) cout[(size_t)(q % m) * kt + q / m] = Bm[q];
                __syncthreads();
            }
            for (int q = tid; q < kt * m; q += 256) {
                if (n == 0) cout[q] = 0.0;
                if (a.coef) static_cast<T *>(a.coef)[(size_t)g * m * kt + q] = (T)cout[q];
            }
            __syncthreads();
            continue;
        }
        for (int q = tid; q < nc * nc; q += 256) Vm[q] = ((q / nc) == (q % nc)) ? 1.0 : 0.0;
        __syncthreads();
        for (int sweep = 0; sweep < 60; ++sweep) {
            if (tid == 0) rotated = 0;
            for (int p = 0; p < nc - 1; ++p)
                for (int q = p + 1; q < nc; ++q) {
                    double *wp = W + (size_t)p * len, *wq = W + (size_t)q * len;
                    double acc[3] = {0.0, 0.0, 0.0};
                    for (int j = tid; j < len; j += 256) { const double u = wp[j], v = wq[j]; acc[0] += u * u; acc[1] += v * v; acc[2] += u * v; }
                    wide_block_sum<3>(acc, red, 4);
                    const double al = acc[0], be = acc[1], ga = acc[2];
                    if (!(ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be))) {
                        const double zeta = (be - al) / (2.0 * ga);
                        const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                        for (int j = tid; j < len; j += 256) { const double u = wp[j], v = wq[j]; wp[j] = c * u - sn * v; wq[j] = sn * u + c * v; }
                        for (int r = tid; r < nc; r += 256) {
                            const double u = Vm[(size_t)r * nc + p], v = Vm[(size_t)r * nc + q];
                            Vm[(size_t)r * nc + p] = c * u - sn * v;
                            Vm[(size_t)r * nc + q] = sn * u + c * v;
                        }
                        if (tid == 0) rotated = (ga == ga) ? 1 : 0;
                    }
                    __syncthreads();
                }
            __syncthreads();
            if (!rotated) break;
        }
        double smax2 = 0.0;
        for (int c = 0; c < nc; ++c) {                             // s_c^2
            double acc[1] = {0.0};
            const double *wc = W + (size_t)c * len;
            for (int j = tid; j < len; j += 256) acc[0] += wc[j] * wc[j];
            wide_block_sum<1>(acc, red, 4);
            if (tid == 0) s2[c] = acc[0];
            smax2 = (acc[0] != acc[0]) ? acc[0] : fmax(smax2, acc[0]);
        }
        __syncthreads();
        const double rcf = a.rc_factor < 0.0 ? 2.220446049250313e-16 * fmax(nfit_g, (double)kt) : a.rc_factor;   // eps * max(n, k)
        const double cutoff = rcf * sqrt(smax2);
        for (int t = 0; t < m; ++t) {
            const double *yt = yv + (size_t)t * n;
            for (int c = tid; c < nc; c += 256) {                  // g_c = (w_c . y  |  v_c . y) / (s_c^2 + alpha), 0 below the cut-off
                double num = 0.0;
                if (dual) { for (int r = 0; r < nc; ++r) num += Vm[(size_t)r * nc + c] * yt[r]; }
                else { const double *wc = W + (size_t)c * len; for (int r = 0; r < len; ++r) num += wc[r] * yt[r]; }
                const double sc = sqrt(s2[c]);
                double gv = (sc > cutoff && sc > 0.0) ? num / (s2[c] + a.alpha) : 0.0;
                if (smax2 != smax2) gv = smax2;
                gsc[c] = gv;
            }
            __syncthreads();
            for (int j = tid; j < kt; j += 256) {
                double b = 0.0;
                if (dual) { for (int c = 0; c < nc; ++c) b += W[(size_t)c * len + j] * gsc[c]; }
                else { for (int c = 0; c < nc; ++c) b += Vm[(size_t)j * nc + c] * gsc[c]; }
                if (n == 0) b = 0.0;
                if (a.coef) static_cast<T *>(a.coef)[((size_t)g * m + t) * kt + j] = (T)b;
                a.coef64[((size_t)g * m + t) * kt + j] = b;
            }
            __syncthreads();
        }
    }
}

int wide_minnorm_launch(pols_ctx *ctx, int dtype, const WideArgs &a, int workers) {
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_svd_kernel<float>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(wide_svd_kernel<double>, dim3((unsigned)workers), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ cd
__device__ __forceinline__ double wide_soft_threshold(double x, double thr, bool positive) {   // ls.rs:373-379
    double r = copysign(fmax(fabs(x) - thr, 0.0), x);
    if (positive) r = fmax(r, 0.0);
    return r;
}

template <typename T>
__global__ void __launch_bounds__(1024) wide_cd_kernel(const WideArgs a) {
    __shared__ double w[K8_KMAX], bv[K8_KMAX], dg[K8_KMAX], red[16];
    __shared__ unsigned char act[K8_KMAX], act_s[K8_KMAX];
    const int tid = threadIdx.x;
    const int kt = a.kt, NZ = kt + 1;
    const int64_t g = blockIdx.x;
    const double n = a.nfit ? a.nfit[g] : (double)(a.offs[g + 1] - a.offs[g]);
    const double *G = a.gram + (size_t)g * NZ * NZ;
    for (int i = tid; i < kt; i += 1024) {
        w[i] = 0.0;                                                // w = zeros (:416)
        bv[i] = G[(size_t)i * NZ + kt];
        dg[i] = G[(size_t)i * NZ + i];                             // xtx[[j, j]] (:431)
        act[i] = 1;
    }
    const double alpha_n = a.alpha * n;                            // alpha * n_samples (:419)
    const double thr = alpha_n * a.l1_ratio, l2 = alpha_n * (1.0 - a.l1_ratio);
    const bool positive = a.positive != 0, active_set = a.active_set != 0;
    int status = (n == 0.0) ? POLS_GROUP_EMPTY : POLS_GROUP_NOT_CONVERGED;
    __syncthreads();
    for (int64_t it = 0; it < a.max_iter && n > 0.0; ++it) {
        for (int i = tid; i < kt; i += 1024) act_s[i] = act[i];    // `for j in active_indices.clone()` (:459)
        __syncthreads();
        double d2 = 0.0;
        for (int j = 0; j < kt; ++j) {
            if (!act_s[j]) continue;                               // block-uniform
            const double *row = G + (size_t)j * NZ;
            const double wj = w[j];
            double part[1] = {0.0};
            for (int i = tid; i < kt; i += 1024) part[0] += row[i] * w[i];
            wide_block_sum<1>(part, red, 16);
            const double dot = bv[j] - part[0] + dg[j] * wj;       // x_j . (residuals + x_j w_j)  (:428-430)
            const double wn = wide_soft_threshold(dot, thr, positive) / (dg[j] + l2);
            const double dw = wn - wj;
            d2 += dw * dw;
            if (tid == 0) { w[j] = wn; if (active_set && fabs(wn) < a.tol) act[j] = 0; }   // (:472-476)
            __syncthreads();                                       // the next coordinate reads the new w[j]
        }
        if (sqrt(d2) < a.tol) { status = POLS_GROUP_OK; break; }   // (:436-444), block-uniform
    }
    for (int i = tid; i < kt; i += 1024) {
        const double out = (n == 0.0) ? 0.0 : w[i];
        if (a.coef) static_cast<T *>(a.coef)[g * kt + i] = (T)out;
        a.coef64[g * kt + i] = out;
    }
    if (tid == 0 && a.status) a.status[g] = status;
}

int wide_cd_launch(pols_ctx *ctx, int dtype, const WideArgs &a) {
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_cd_kernel<float>, dim3((unsigned)a.n_groups), dim3(1024), 0, ctx->stream, a);
    else hipLaunchKernelGGL(wide_cd_kernel<double>, dim3((unsigned)a.n_groups), dim3(1024), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ predict
template <typename T>
__global__ void __launch_bounds__(256) wide_predict_kernel(const WideArgs a) {
    __shared__ double cs[K8_KMAX];
    const int64_t g = blockIdx.y;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int ku = a.k_user, kt = a.kt;
    const int m = wide_m(a), t = blockIdx.z;                       // one target per grid.z slice
    for (int j = threadIdx.x; j < kt; j += 256) cs[j] = a.coef64[((size_t)g * m + t) * kt + j];
    __syncthreads();
    T *pred = static_cast<T *>(a.pred_cols ? a.pred_cols[t] : a.pred), *resid = static_cast<T *>(a.resid);
    for (int64_t r = s + (int64_t)blockIdx.x * 256 + threadIdx.x; r < e; r += (int64_t)gridDim.x * 256) {
        const T sw = a.w ? sqrt(static_cast<const T *>(a.w)[r]) : T(1);
        T p = T(0);
        for (int j = 0; j < ku; ++j) p = fma(null_fill<T>(a.null_policy, static_cast<const T *>(a.cols[j])[r]) * sw, (T)cs[j], p);
        if (ku != kt) p = fma(sw, (T)cs[kt - 1], p);
        if (a.w) p *= T(1) / sw;                                   // (sqrt_w x) . c * (1 / sqrt_w)  (ls.py:190-196, 234-235)
        if (a.null_policy == POLS_NULL_DROP) p = nan_if<T>(a.rowmask[r] ? 0u : 1u, p);   // rows that were not fitted (ex.rs:409-417)
        if (pred) pred[r] = p;
        if (resid) resid[r] = static_cast<const T *>(a.y)[r] - p;
    }
}

int wide_predict_launch(pols_ctx *ctx, int dtype, const WideArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, (a.n_rows / std::max<int64_t>(1, a.n_groups) + 255) / 256));
    const dim3 grid(bx, (unsigned)a.n_groups, (unsigned)wide_m(a));
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_predict_kernel<float>, grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(wide_predict_kernel<double>, grid, dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ statistics
// compute_residual_metrics (st.rs:15-37) + compute_feature_metrics (:79-156) per group, like K7 but with the k x k matrix
// in dynamic LDS: (X'X + lambda I)^-1 by the symmetric sweep operator (a non-positive pivot = the reference's failed
// Cholesky -> NaN standard errors / t / p, :101-111), its diagonal and trace, the side-car's own coefficients inv . X'y,
// then two passes over the group's rows for the mean of the targets and the four sums.
// IN_LDS (up to 127 columns, 256 threads): the matrix lives in dynamic LDS.  Otherwise (up to 1 024 columns, 1 024 threads) it
// lives in a per-group HBM / L2 work area (o.work) -- one workgroup still owns it, so a barrier orders its global accesses.
template <typename T, int NT, bool IN_LDS>
__global__ void __launch_bounds__(NT) wide_stats_kernel(const WideArgs a, const WideStatsOut o) {
    extern __shared__ double sl[];
    constexpr int NW = NT / 64;
    __shared__ double red[4 * NW];
    __shared__ int ok_s;
    const int tid = threadIdx.x;
    const int kt = a.kt, ku = a.k_user, NZ = kt + 1, LD = kt | 1;
    const int64_t g = blockIdx.x;
    double *P = IN_LDS ? sl : o.work + (size_t)g * kt * LD;
    double *v = IN_LDS ? P + (size_t)kt * LD : sl, *bv = v + kt, *binv = bv + kt, *cdis = binv + kt;
    const int64_t s = a.offs[g], e = a.offs[g + 1], n = e - s;
    const double *G = a.gram + (size_t)g * NZ * NZ;
    if (o.factored) {
        // Beyond 127 columns wide_chol factored the Gram matrix IN PLACE: its lower triangle now holds L_ip d_p, its diagonal the
        // pivots d_i (of G + alpha I); the strict upper triangle is untouched.  The matrix comes back from the upper triangle and
        // G_ii + alpha = d_i + sum_{p < i} L_ip^2 d_p.
        for (int i = tid; i < kt; i += NT) {
            double dsum = G[(size_t)i * NZ + i];
            for (int p = 0; p < i; ++p) { const double l = G[(size_t)i * NZ + p]; dsum += l * l / G[(size_t)p * NZ + p]; }
            v[i] = dsum + (o.lambda - a.alpha);
        }
        __syncthreads();
        for (int q = tid; q < kt * kt; q += NT) {
            const int i = q / kt, c = q - i * kt;
            P[i * LD + c] = (i == c) ? v[i] : G[(size_t)(i < c ? i : c) * NZ + (i < c ? c : i)];
        }
    } else {
        for (int q = tid; q < kt * kt; q += NT) { const int i = q / kt, c = q - i * kt; P[i * LD + c] = G[(size_t)i * NZ + c] + (i == c ? o.lambda : 0.0); }
    }
    for (int i = tid; i < kt; i += NT) { bv[i] = G[(size_t)i * NZ + kt]; cdis[i] = a.coef64[g * kt + i]; }
    if (tid == 0) ok_s = 1;
    __syncthreads();
    for (int j = 0; j < kt; ++j) {
        const double d = P[j * LD + j];
        if (tid == 0 && !(d > 0.0)) ok_s = 0;
        const double p = 1.0 / d;
        for (int i = tid; i < kt; i += NT) v[i] = P[i * LD + j];
        __syncthreads();
        const int di = NT / kt, dc = NT - di * kt;                // (i, c) advance by NT elements without a divide per element
        for (int i = tid / kt, c = tid - (tid / kt) * kt; i < kt;) {
            double val;
            if (i == j && c == j) val = -p;
            else if (i == j) val = v[c] * p;
            else if (c == j) val = v[i] * p;
            else val = P[i * LD + c] - v[i] * v[c] * p;
            P[i * LD + c] = val;
            i += di; c += dc;
            if (c >= kt) { c -= kt; ++i; }
        }
        __syncthreads();
    }
    for (int i = tid; i < kt; i += NT) {                           // P now holds -(inverse)
        double acc = 0.0;
        for (int c = 0; c < kt; ++c) acc -= P[i * LD + c] * bv[c];
        binv[i] = acc;                                             // A^-1 X'y  (:116)
        v[i] = -P[i * LD + i];                                     // diag(A^-1)
    }
    __syncthreads();
    const T *yp = static_cast<const T *>(a.y), *wp = static_cast<const T *>(a.w);
    double sums[1] = {0.0};
    for (int64_t r = s + tid; r < e; r += NT) sums[0] += (double)yp[r] * (wp ? sqrt((double)wp[r]) : 1.0);
    wide_block_sum<1>(sums, red, NW);
    const double mean = n ? sums[0] / (double)n : 0.0;             // targets.mean().unwrap_or(0.0)  (:16)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};                           // sse, sae, sst, rss
    for (int64_t r = s + tid; r < e; r += NT) {
        const double sw = wp ? sqrt((double)wp[r]) : 1.0;
        const double yt = (double)yp[r] * sw;
        double p1 = 0.0, p2 = 0.0;
        for (int j = 0; j < kt; ++j) {
            const double x = ((j < ku) ? (double)static_cast<const T *>(a.cols[j])[r] : 1.0) * sw;
            p1 = fma(x, cdis[j], p1);
            p2 = fma(x, binv[j], p2);
        }
        const double e1 = yt - p1, e2 = yt - p2, dm = yt - mean;
        acc[0] += e1 * e1; acc[1] += fabs(e1); acc[2] += dm * dm; acc[3] += e2 * e2;
    }
    wide_block_sum<4>(acc, red, NW);
    double trace = 0.0;
    for (int j = 0; j < kt; ++j) trace += v[j];
    const double nn = (double)n;
    const double df = (o.lambda > 0.0) ? nn - trace : nn - (double)kt;          // :124-128
    const bool ok = ok_s != 0;
    if (tid == 0) {
        if (o.mse) o.mse[g] = acc[0] / nn;
        if (o.mae) o.mae[g] = acc[1] / nn;
        if (o.r2) o.r2[g] = 1.0 - acc[0] / acc[2];
        if (a.status && ok && !(df > 0.0)) a.status[g] = POLS_GROUP_BAD_DOF;
    }
    for (int i = tid; i < kt; i += NT) {
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        double se = nanv, tv = nanv, pv = nanv;
        if (ok && df > 0.0) {
            se = sqrt(acc[3] / df * fabs(v[i]));
            tv = binv[i] / se;
            pv = (tv != tv) ? nanv : k7_betai(0.5 * df, 0.5, df / (df + tv * tv));
        }
        if (o.se) o.se[g * kt + i] = se;
        if (o.tv) o.tv[g * kt + i] = tv;
        if (o.pv) o.pv[g * kt + i] = pv;
    }
}

int wide_stats_launch(pols_ctx *ctx, int dtype, const WideArgs &a, const WideStatsOut &o_in) {
    if (a.kt > K8_KMAX) return fail(POLS_ERR_UNSUPPORTED, "statistics: %d features (incl. intercept) > %d", a.kt, K8_KMAX);
    if (a.n_groups == 0) return POLS_OK;
    WideStatsOut o = o_in;
    if (a.kt <= K8_STATS_LDS_KMAX) {
        const size_t lds = sizeof(double) * ((size_t)a.kt * (a.kt | 1) + 4 * (size_t)a.kt);
        static OncePerDevice attr_once;
        if (attr_once.needed(ctx->device)) {
            POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_stats_kernel<float, 256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wide_stats_kernel<double, 256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            attr_once.done(ctx->device);
        }
        if (dtype == POLS_F32) hipLaunchKernelGGL((wide_stats_kernel<float, 256, true>), dim3((unsigned)a.n_groups), dim3(256), lds, ctx->stream, a, o);
        else hipLaunchKernelGGL((wide_stats_kernel<double, 256, true>), dim3((unsigned)a.n_groups), dim3(256), lds, ctx->stream, a, o);
    } else {
        // the solve is over: its work area (slot 3) is free to hold one kt x kt matrix per group
        void *w = nullptr;
        int rc = ensure_scratch(ctx, 3, sizeof(double) * (size_t)a.n_groups * a.kt * (a.kt | 1), &w);
        if (rc) return rc;
        o.work = static_cast<double *>(w);
        const size_t lds = sizeof(double) * 4 * (size_t)a.kt;
        if (dtype == POLS_F32) hipLaunchKernelGGL((wide_stats_kernel<float, 1024, false>), dim3((unsigned)a.n_groups), dim3(1024), lds, ctx->stream, a, o);
        else hipLaunchKernelGGL((wide_stats_kernel<double, 1024, false>), dim3((unsigned)a.n_groups), dim3(1024), lds, ctx->stream, a, o);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) wide_predict_rows_kernel(const WideArgs a, const T *coef) {
    const int ku = a.k_user, kt = a.kt;
    T *pred = static_cast<T *>(a.pred);
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n_rows; r += (int64_t)gridDim.x * 256) {
        const T *c = coef + r * kt;
        T p = T(0);
        for (int j = 0; j < ku; ++j) p = fma(null_fill<T>(a.null_policy, static_cast<const T *>(a.cols[j])[r]), c[j], p);   // (features * coefficients).sum_axis(1)
        if (ku != kt) p += c[kt - 1];
        pred[r] = p;
    }
}

int wide_predict_rows_launch(pols_ctx *ctx, int dtype, const WideArgs &a, const void *coef_rows) {
    if (a.n_rows == 0) return POLS_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>(4096, (a.n_rows + 255) / 256);
    if (dtype == POLS_F32) hipLaunchKernelGGL(wide_predict_rows_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, a, static_cast<const float *>(coef_rows));
    else hipLaunchKernelGGL(wide_predict_rows_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, a, static_cast<const double *>(coef_rows));
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
