// k6s_small.hip -- K6s: the minimum-norm solver for SHORT flagged groups, a sub-wave team per group.
//
// `solve_ols` sends a group with no more rows than columns to the SVD (`least_squares.rs:211-240` -> `solve_ols_svd`, `:183-191`,
// LAPACK dgelsd: the minimum-norm solution), and "svd" sends every group there.  Such groups fail K1's Cholesky (X'X is singular) and
// used to queue for K6's pool of 256-thread workgroups -- one group at a time per workgroup, a primal one-sided Jacobi over kt columns
// that never converges on the kt - n null directions: 65 s for 2.5 M groups of 4 rows x 8 features (profiles/r04_bench_shape_cliffs.txt).
// Here a team of 4 / 8 / 16 / 32 lanes takes a group, LANE r HOLDS DATA ROW r in registers (kt features and the target, sqrt(w)-scaled,
// rows the null policy drops as zero rows), and the Jacobi runs on the DUAL: the columns of B = A' (kt x n) are the data rows, so
// rotating PAIRS OF LANES until the rows are mutually orthogonal gives A = R S U' with the rows' norms as singular values -- at most
// n <= 32 of them, the null space never enters.  The target rides along as one more component of every row (y' R), hence
//      beta = sum_c  a_c * (y'R)_c / (s_c^2 + alpha)          (s_c above the cut-off; alpha = 0: the dgelsd solution)
// and one team all-reduce per coefficient finishes the group.  A round pairs every lane of the team with one partner (round-robin
// tournament: TEAM - 1 rounds per sweep); the partner's row arrives through `ds_bpermute` and BOTH lanes of a pair compute the same
// rotation from bit-identical sums.  16 / 8 / 4 / 2 groups per wave; no LDS, no barriers, no scratch.
//
// Which groups: flagged (POLS_GROUP_FALLBACK), at most `small_rows` (= TEAM) rows, and a (branch, solve_method) whose reference
// solver for such a group is the SVD -- `k6s_takes` below, the same predicate K6 uses to leave them alone.
#include "k6_svd.hpp"

namespace pols {

template <int TEAM>
__device__ __forceinline__ double k6s_team_sum(double v) {
#pragma unroll
    for (int m = 1; m < TEAM; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
template <int TEAM>
__device__ __forceinline__ double k6s_team_max(double v) {
#pragma unroll
    for (int m = 1; m < TEAM; m <<= 1) v = fmax(v, __shfl_xor(v, m));
    return v;
}

template <typename T, int TEAM, int KTB>
__global__ void __launch_bounds__(256) k6s_kernel(const K6Args a) {
    // solve_method = None picks the SVD BY SHAPE (ls.rs:224-231: "n > k ? QR : SVD"), so a group with fewer rows than columns comes here
    // whether or not its Cholesky noticed: in f32 the last pivot of a rank-deficient Gram matrix can come out as positive noise above the
    // flagging threshold (one group in 2 600 in tests/test_k6_gpu.py) -- the reference never factors such a group at all.
    const bool by_shape = a.mode == FIX_OLS_AUTO || a.mode == FIX_MINNORM;   // ("svd": the SVD on every group -- K1's Cholesky answer stands in for it only at full column rank)
    if (!by_shape && a.fb_flag && *a.fb_flag != a.epoch) return;  // nothing was flagged in this call
    const int tix = threadIdx.x, lane = tix & 63, sub = lane % TEAM, team0 = lane - sub;
    const int kt = a.kt, ku = a.k_user, pol = a.null_policy;
    const int64_t g = (int64_t)blockIdx.x * (256 / TEAM) + tix / TEAM;
    int64_t s = 0, e = 0;
    bool live = g < a.n_groups;
    bool flagged = false;
    if (live) { s = a.offs[g]; e = a.offs[g + 1]; flagged = a.status[g] == POLS_GROUP_FALLBACK; }
    const int64_t n = e - s;
    const bool shaped = by_shape && n > 0 && n <= (int64_t)kt;    // n < kt: singular by construction; n == kt: the reference still takes the SVD
    live = live && (flagged || shaped) && k6s_takes(a.mode, n, kt, a.small_rows) && n > (int64_t)a.small_lo;   // (small_lo: the next smaller team size's launch has the shorter groups)
    if (!__any(live)) return;                                      // (wave-uniform; from here on every lane stays active: the shuffles need them)

    // ---- lane `sub` = data row s + sub: kt features and the target, sqrt(w)-scaled; dropped rows and lanes beyond the group: zero rows
    const bool has = live && sub < n;
    const int64_t r = s + sub;
    double v[KTB + 1];
#pragma unroll
    for (int j = 0; j <= KTB; ++j) v[j] = 0.0;
    bool in_fit = false;
    if (has) {
        in_fit = null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, r);
        if (in_fit) {
            const double sw = a.w ? sqrt((double)static_cast<const T *>(a.w)[r]) : 1.0;
#pragma unroll
            for (int j = 0; j < KTB; ++j)
                if (j < kt) v[j] = (j < ku ? (double)null_fill<T>(pol, static_cast<const T *>(a.x[j])[r]) : 1.0) * sw;
            v[KTB] = (double)null_fill<T>(pol, static_cast<const T *>(a.y)[r]) * sw;
        }
    }
    const double nfit = k6s_team_sum<TEAM>(in_fit ? 1.0 : 0.0);

    // ---- one-sided Jacobi over the rows (Hestenes), round-robin pairing: in round t lane TEAM - 1 meets lane t, the others i + j = 2t
    // (mod TEAM - 1).  Same rotation formula and stopping rule as K6's primal sweeps.
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll 1
        for (int t = 0; t < TEAM - 1; ++t) {
            int partner;
            if (sub == TEAM - 1) partner = t;
            else if (sub == t) partner = TEAM - 1;
            else { partner = 2 * t - sub; partner += partner < 0 ? TEAM - 1 : 0; partner -= partner >= TEAM - 1 ? TEAM - 1 : 0; }
            const int src = team0 + partner;
            double o[KTB + 1];
#pragma unroll
            for (int j = 0; j <= KTB; ++j) o[j] = __shfl(v[j], src);
            double nm = 0.0, no = 0.0, ga = 0.0;
#pragma unroll
            for (int j = 0; j < KTB; ++j) { nm = fma(v[j], v[j], nm); no = fma(o[j], o[j], no); ga = fma(v[j], o[j], ga); }
            const bool low = sub < partner;
            const double al = low ? nm : no, be = low ? no : nm;
            double c = 1.0, sn = 0.0;
            if (!(ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be))) {            // NaN data also lands here
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                c = 1.0 / sqrt(1.0 + tt * tt);
                sn = c * tt;
                rotated = rotated || (ga == ga);                                 // NaN: give up after this sweep
            }
            const double sg = low ? -sn : sn;                                    // low row: c u - s v; high row: s u + c v
#pragma unroll
            for (int j = 0; j <= KTB; ++j) v[j] = fma(sg, o[j], c * v[j]);
        }
        if (!__any(rotated)) break;
    }

    // ---- singular values = row norms; beta = sum_c a_c (y'R)_c / (s_c^2 + alpha) over the rows above the cut-off
    double s2 = 0.0;
#pragma unroll
    for (int j = 0; j < KTB; ++j) s2 = fma(v[j], v[j], s2);
    const double sv = sqrt(s2);
    const double smax = k6s_team_max<TEAM>(sv);
    const double bad = k6s_team_sum<TEAM>((s2 != s2 || v[KTB] != v[KTB]) ? 1.0 : 0.0);      // NaN data: every coefficient NaN
    const double rcf = a.rc_factor < 0.0 ? 2.220446049250313e-16 * fmax(nfit, (double)kt) : a.rc_factor;   // eps * max(n, k)
    const double cutoff = rcf * smax;
    double gc;
    if (a.alpha > 0.0) gc = (sv < cutoff) ? 0.0 : v[KTB] / (s2 + a.alpha);                   // ls.rs:143-148
    else gc = (sv > cutoff && sv > 0.0) ? v[KTB] / s2 : 0.0;                                 // dgelsd
    double beta[KTB];
#pragma unroll
    for (int j = 0; j < KTB; ++j) {
        double bj = k6s_team_sum<TEAM>(v[j] * gc);
        bj = bad != 0.0 ? __longlong_as_double(0x7ff8000000000000LL) : bj;
        beta[j] = nfit == 0.0 ? 0.0 : bj;
    }
    if (live) {
        // (a square group that factored keeps its status: nothing failed, it only gets the solver the reference runs on it)
        if (sub == 0 && (nfit == 0.0 || (!flagged && n < (int64_t)kt)))   // every row dropped by the null policy: zeros, like an empty group
            a.status[g] = nfit == 0.0 ? POLS_GROUP_EMPTY : POLS_GROUP_FALLBACK;
        if (a.coef) {
#pragma unroll
            for (int j = 0; j < KTB; ++j)
                if (j < kt && (j % TEAM) == sub) static_cast<T *>(a.coef)[g * kt + j] = (T)beta[j];
        }
    }
    // ---- predictions / residuals of the group's rows (make_predictions on the fit features, ex.rs:398-405)
    if (has && (a.pred || a.resid)) {
        double p = 0.0;
#pragma unroll
        for (int j = 0; j < KTB; ++j)
            if (j < kt) p = fma(j < ku ? (double)null_fill<T>(pol, static_cast<const T *>(a.x[j])[r]) : 1.0, beta[j], p);
        if (pol == POLS_NULL_DROP) p = nan_if<double>(in_fit ? 0u : 1u, p);
        if (a.pred) static_cast<T *>(a.pred)[r] = (T)p;
        if (a.resid) static_cast<T *>(a.resid)[r] = (T)((double)static_cast<const T *>(a.y)[r] - p);
    }
}

template <typename T, int TEAM>
static void k6s_launch_ktb(pols_ctx *ctx, const K6Args &a) {
    const unsigned blocks = (unsigned)((a.n_groups + (256 / TEAM) - 1) / (256 / TEAM));
    if (a.kt <= 8) hipLaunchKernelGGL((k6s_kernel<T, TEAM, 8>), dim3(blocks), dim3(256), 0, ctx->stream, a);
    else if (a.kt <= 16) hipLaunchKernelGGL((k6s_kernel<T, TEAM, 16>), dim3(blocks), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL((k6s_kernel<T, TEAM, 32>), dim3(blocks), dim3(256), 0, ctx->stream, a);
}

template <typename T>
static void k6s_launch_team(pols_ctx *ctx, const K6Args &a) {
    switch (a.small_rows) {
        case 4: k6s_launch_ktb<T, 4>(ctx, a); break;
        case 8: k6s_launch_ktb<T, 8>(ctx, a); break;
        case 16: k6s_launch_ktb<T, 16>(ctx, a); break;
        default: k6s_launch_ktb<T, 32>(ctx, a); break;
    }
}

// a.small_rows in {4, 8, 16, 32}: the team size; groups with more rows stay with K6
int k6s_launch(pols_ctx *ctx, int dtype, const K6Args &a) {
    if (a.kt > 32 || a.n_groups <= 0) return POLS_OK;
    if ((a.n_groups + 1) / 2 > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    if (dtype == POLS_F32) k6s_launch_team<float>(ctx, a);
    else k6s_launch_team<double>(ctx, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
