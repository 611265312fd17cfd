// k7_stats.hip -- K7 "group_statistics": regression statistics for every group in one launch.
//
// Replaces compute_residual_metrics (src/statistics.rs:15-37) and compute_feature_metrics (:79-156) as called by the
// plugin least_squares_statistics (src/expressions.rs:468-509).  One 256-thread workgroup per group:
//   wave 0   reads the group's Gram matrix (already streamed by gram_stream), factors A = X'X + lambda I by Cholesky
//            in LDS (failure -> NaN standard errors / t / p, :101-111), forms M = L^-1 one column per lane, and from it
//            diag(A^-1) = column norms of M, trace(A^-1), and the side-car's own coefficients M'(M X'y) (:116);
//   all      one pass over the group's rows (L2-warm after the Gram pass for small groups) for the mean of the
//            targets, then a second for  sum e^2, sum |e|, sum (y - mean)^2  with the DISPATCHER's coefficients
//            (:15-37) and the RSS of the side-car's coefficients (:119-123);
//   lanes<k  df = n - p or n - trace (:124-128), se = sqrt(sigma^2 |inv_jj|), t = beta / se, and the two-sided
//            Student-t p-value 2 (1 - cdf(|t|)) == I_{df/(df+t^2)}(df/2, 1/2) by Lentz' continued fraction.
// Everything is f64, like the reference (which casts its inputs to f64, src/expressions.rs:22-63).
#include "k7_stats.hpp"

namespace pols {

constexpr int K7_KMAX = 31;

template <int NV>
__device__ __forceinline__ void k7_block_sum(double (&v)[NV], double (*red)[4]) {   // red: NV x 4 doubles of LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum_row3(v[i]);
        if (lane == 63) red[i][wv] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i][0] + red[i][1] + red[i][2] + red[i][3];
}

// wave 0 of the group: A = X'X + lambda I factored, A^-1 X'y and diag(A^-1) (src/statistics.rs:100-121), the dispatcher's coefficients
template <typename T>
__device__ __forceinline__ void k7_small_solve(const StatsArgs &a, int64_t g, int lane, double *L, double *M, double *rinv, double *bvec,
                                               double *tvec, double *binv, double *cdis, double *dg, int *okflag_p) {
    const int kt = a.kt, NZ = kt + 1;
    const double *G = a.gram + (size_t)g * NZ * NZ;
    int &okflag = *okflag_p;
    {
        for (int q = lane; q < kt * kt; q += 64) {
            const int i = q / kt, j = q - i * kt;
            L[q] = G[i * NZ + j] + (i == j ? a.lambda : 0.0);
        }
        if (lane < kt) {
            bvec[lane] = G[lane * NZ + kt];
            cdis[lane] = (double)static_cast<const T *>(a.coef)[g * kt + lane];
        }
        __builtin_amdgcn_wave_barrier();
        bool ok = true;
        for (int j = 0; j < kt; ++j) {
            double d = L[j * kt + j];
            for (int p = 0; p < j; ++p) d -= L[j * kt + p] * L[j * kt + p];
            ok = ok && (d > 0.0);                                   // also false for NaN
            const double ri = 1.0 / sqrt(d);
            if (lane == 0) rinv[j] = ri;
            if (lane > j && lane < kt) {
                double acc = L[lane * kt + j];
                for (int p = 0; p < j; ++p) acc -= L[lane * kt + p] * L[j * kt + p];
                L[lane * kt + j] = acc * ri;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // M = L^-1, column c on lane c (forward substitution against e_c)
        if (lane < kt) {
            const int c = lane;
            for (int i = 0; i < c; ++i) M[i * kt + c] = 0.0;
            M[c * kt + c] = rinv[c];
            for (int i = c + 1; i < kt; ++i) {
                double acc = 0.0;
                for (int p = c; p < i; ++p) acc += L[i * kt + p] * M[p * kt + c];
                M[i * kt + c] = -acc * rinv[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < kt) {
            double t = 0.0;
            for (int j = 0; j <= lane; ++j) t += M[lane * kt + j] * bvec[j];
            tvec[lane] = t;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < kt) {
            double bi = 0.0, dd = 0.0;
            for (int p = lane; p < kt; ++p) { const double m = M[p * kt + lane]; bi += m * tvec[p]; dd += m * m; }
            binv[lane] = bi;                                        // A^-1 X'y          (:116)
            dg[lane] = dd;                                          // diag(A^-1)
        }
        if (lane == 0) okflag = ok ? 1 : 0;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k7_stats_kernel(const StatsArgs a) {
    __shared__ double L[K7_KMAX * K7_KMAX], M[K7_KMAX * K7_KMAX];
    __shared__ double rinv[K7_KMAX], bvec[K7_KMAX], tvec[K7_KMAX], binv[K7_KMAX], cdis[K7_KMAX], dg[K7_KMAX];
    __shared__ double red[4][4];
    __shared__ int okflag;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t g = blockIdx.x;
    const int kt = a.kt, ku = a.k_user, NZ = kt + 1;
    const int64_t s = a.offs[g], e = a.offs[g + 1], n = e - s;
    const double *G = a.gram + (size_t)g * NZ * NZ;

    if (wv == 0) k7_small_solve<T>(a, g, lane, L, M, rinv, bvec, tvec, binv, cdis, dg, &okflag);
    __syncthreads();

    const T *yp = static_cast<const T *>(a.y), *wp = static_cast<const T *>(a.w);
    double sums[1] = {0.0};
    for (int64_t r = s + tid; r < e; r += 256) {
        const double sw = wp ? sqrt((double)wp[r]) : 1.0;
        sums[0] += (double)yp[r] * sw;
    }
    k7_block_sum<1>(sums, red);
    const double mean = n ? sums[0] / (double)n : 0.0;             // targets.mean().unwrap_or(0.0)  (:16)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};                           // sse, sae, sst, rss
    for (int64_t r = s + tid; r < e; r += 256) {
        const double sw = wp ? sqrt((double)wp[r]) : 1.0;
        const double yt = (double)yp[r] * sw;
        double p1 = 0.0, p2 = 0.0;
        for (int j = 0; j < kt; ++j) {
            const double x = ((j < ku) ? (double)static_cast<const T *>(a.x[j])[r] : 1.0) * sw;
            p1 = fma(x, cdis[j], p1);
            p2 = fma(x, binv[j], p2);
        }
        const double e1 = yt - p1, e2 = yt - p2, dm = yt - mean;
        acc[0] += e1 * e1; acc[1] += fabs(e1); acc[2] += dm * dm; acc[3] += e2 * e2;
    }
    k7_block_sum<4>(acc, red);

    const double nn = (double)n;
    double trace = 0.0;
    for (int j = 0; j < kt; ++j) trace += dg[j];
    const double df = (a.lambda > 0.0) ? nn - trace : nn - (double)kt;          // :124-128
    const bool ok = okflag != 0;
    if (tid == 0) {
        if (a.mse) a.mse[g] = acc[0] / nn;
        if (a.mae) a.mae[g] = acc[1] / nn;
        if (a.r2) a.r2[g] = 1.0 - acc[0] / acc[2];
        if (a.status && ok && !(df > 0.0)) a.status[g] = POLS_GROUP_BAD_DOF;
    }
    if (tid < kt) {
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        double se = nanv, tv = nanv, pv = nanv;
        if (ok && df > 0.0) {
            const double sigma2 = acc[3] / df;
            se = sqrt(sigma2 * fabs(dg[tid]));
            tv = binv[tid] / se;
            pv = (tv != tv) ? nanv : k7_betai(0.5 * df, 0.5, df / (df + tv * tv));
        }
        if (a.se) a.se[g * kt + tid] = se;
        if (a.tv) a.tv[g * kt + tid] = tv;
        if (a.pv) a.pv[g * kt + tid] = pv;
    }
}

// ---- long groups: prepare (per group) / segment sums (per segment) / finish (per group)
template <typename T>
__global__ void __launch_bounds__(64) k7_prepare_kernel(const StatsArgs a) {
    __shared__ double L[K7_KMAX * K7_KMAX], M[K7_KMAX * K7_KMAX];
    __shared__ double rinv[K7_KMAX], bvec[K7_KMAX], tvec[K7_KMAX], binv[K7_KMAX], cdis[K7_KMAX], dg[K7_KMAX];
    __shared__ int okflag;
    const int64_t g = blockIdx.x;
    const int lane = threadIdx.x, kt = a.kt;
    k7_small_solve<T>(a, g, lane, L, M, rinv, bvec, tvec, binv, cdis, dg, &okflag);
    __syncthreads();
    double *P = a.prep + (size_t)g * (3 * kt + 1);
    if (lane < kt) { P[lane] = cdis[lane]; P[kt + lane] = binv[lane]; P[2 * kt + lane] = dg[lane]; }
    if (lane == 0) P[3 * kt] = (double)okflag;
}

// one workgroup per segment, ONE pass: sums of (y - c), (y - c)^2 with c = the group's first (scaled) target -- the mean and the total
// sum of squares follow without a second pass and without the cancellation of raw moments -- and of the three residual forms
template <typename T>
__global__ void __launch_bounds__(256) k7_segment_kernel(const StatsArgs a) {
    __shared__ double red[5][4];
    __shared__ double cdis[K7_KMAX], binv[K7_KMAX];
    const int tid = threadIdx.x;
    const int64_t sgi = blockIdx.x, g = a.seg_map[sgi];
    const int kt = a.kt, ku = a.k_user;
    const int64_t s = a.seg_offs[sgi], e = a.seg_offs[sgi + 1], gs = a.offs[g];
    const double *P = a.prep + (size_t)g * (3 * kt + 1);
    if (tid < kt) { cdis[tid] = P[tid]; binv[tid] = P[kt + tid]; }
    __syncthreads();
    const T *yp = static_cast<const T *>(a.y), *wp = static_cast<const T *>(a.w);
    const double c = (a.offs[g + 1] > gs) ? (double)yp[gs] * (wp ? sqrt((double)wp[gs]) : 1.0) : 0.0;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};                      // sum (y - c), sum (y - c)^2, sse, sae, rss
    for (int64_t r = s + tid; r < e; r += 256) {
        const double sw = wp ? sqrt((double)wp[r]) : 1.0;
        const double yt = (double)yp[r] * sw;
        double p1 = 0.0, p2 = 0.0;
        for (int j = 0; j < kt; ++j) {
            const double x = ((j < ku) ? (double)static_cast<const T *>(a.x[j])[r] : 1.0) * sw;
            p1 = fma(x, cdis[j], p1);
            p2 = fma(x, binv[j], p2);
        }
        const double e1 = yt - p1, e2 = yt - p2, dc = yt - c;
        acc[0] += dc; acc[1] += dc * dc; acc[2] += e1 * e1; acc[3] += fabs(e1); acc[4] += e2 * e2;
    }
    k7_block_sum<5>(acc, red);
    if (tid < 5) a.seg_part[(size_t)sgi * 5 + tid] = acc[tid];
}

template <typename T>
__global__ void __launch_bounds__(64) k7_finish_kernel(const StatsArgs a) {
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x, kt = a.kt;
    const int64_t n = a.offs[g + 1] - a.offs[g];
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int v = a.seg_first[g] + tid; v < a.seg_first[g + 1]; v += 64)   // lane l: segments l, l + 64, ...; then a fixed tree over the lanes:
        for (int i = 0; i < 5; ++i) acc[i] += a.seg_part[(size_t)v * 5 + i];   // the same sums whatever ran first
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = __shfl(wave_sum_row3(acc[i]), 63);
    const double nn = (double)n;
    const double sst = n ? acc[1] - acc[0] * acc[0] / nn : 0.0;    // sum (y - mean)^2 from the shifted sums
    const double *P = a.prep + (size_t)g * (3 * kt + 1);
    double trace = 0.0;
    for (int j = 0; j < kt; ++j) trace += P[2 * kt + j];
    const double df = (a.lambda > 0.0) ? nn - trace : nn - (double)kt;          // :124-128
    const bool ok = P[3 * kt] != 0.0;
    if (tid == 0) {
        if (a.mse) a.mse[g] = acc[2] / nn;
        if (a.mae) a.mae[g] = acc[3] / nn;
        if (a.r2) a.r2[g] = 1.0 - acc[2] / sst;
        if (a.status && ok && !(df > 0.0)) a.status[g] = POLS_GROUP_BAD_DOF;
    }
    if (tid < kt) {
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        double se = nanv, tv = nanv, pv = nanv;
        if (ok && df > 0.0) {
            const double sigma2 = acc[4] / df;
            se = sqrt(sigma2 * fabs(P[2 * kt + tid]));
            tv = P[kt + tid] / se;
            pv = (tv != tv) ? nanv : k7_betai(0.5 * df, 0.5, df / (df + tv * tv));
        }
        if (a.se) a.se[g * kt + tid] = se;
        if (a.tv) a.tv[g * kt + tid] = tv;
        if (a.pv) a.pv[g * kt + tid] = pv;
    }
}

template <typename T>
static int k7_split_launch_t(pols_ctx *ctx, const StatsArgs &a) {
    hipLaunchKernelGGL(k7_prepare_kernel<T>, dim3((unsigned)a.n_groups), dim3(64), 0, ctx->stream, a);
    hipLaunchKernelGGL(k7_segment_kernel<T>, dim3((unsigned)a.n_seg), dim3(256), 0, ctx->stream, a);
    hipLaunchKernelGGL(k7_finish_kernel<T>, dim3((unsigned)a.n_groups), dim3(64), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int k7_stats_launch(pols_ctx *ctx, int dtype, const StatsArgs &a) {
    if (a.kt > K7_KMAX) return fail(POLS_ERR_UNSUPPORTED, "statistics: %d features (incl. intercept) > %d", a.kt, K7_KMAX);
    if (a.n_groups == 0) return POLS_OK;
    if (a.seg_offs) return dtype == POLS_F32 ? k7_split_launch_t<float>(ctx, a) : k7_split_launch_t<double>(ctx, a);
    if (dtype == POLS_F32) hipLaunchKernelGGL(k7_stats_kernel<float>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k7_stats_kernel<double>, dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
