// k3_rls.hip -- K3 "rls_rank1": recursive least squares, one 64-lane wave per sequence (group).
//
// Replaces RecursiveLeastSquares::update + solve_recursive_least_squares (src/least_squares.rs:494-598) and
// the dynamic make_predictions (src/expressions.rs:184, 640-645) for every group of a frame in one launch.
//
// The recursion is a strict dependency chain over rows (P_t depends on P_{t-1}), so a sequence cannot use
// more than one wave; throughput comes from many sequences in flight.  Inside the wave the K x K covariance
// lives ONE ENTRY PER LANE (lane = 8*i + j, K <= 8): the rank-1 update
//     r = 1 + x'Px / ff ;  k = Px / (r ff) ;  beta += k (y - x'beta) ;  P = P / ff - (k k') r      (:531-540)
// needs P x along rows and along columns (P is symmetric, so the column reduction delivers (Px)_j to the lane
// that also holds (Px)_i), which are 3-step cross-lane reductions: DPP quad_perm / row_half_mirror inside the
// 8-lane row, DPP row_ror:8 + v_permlane16_swap + v_permlane32_swap across rows.  All arithmetic is f64 like
// the reference.  Rows are staged 64 at a time: each lane loads one row of every column (coalesced 512 B per
// column), parks it in LDS, and the per-step operand reads run ahead of the dependency chain; coefficients and
// predictions of a block are collected in LDS and written back with coalesced stores.
//
// Algorithmic HBM bytes per row: b (k + 1) read (+1 validity byte) and b k (coefficients) and/or b
// (predictions) written -- cfg4 (k = 6, f64): 56 B in, 8..56 B out.  Bound: the serial chain (one sequence) or
// VALU issue (many sequences); never HBM.
#include "common.hpp"
#include "k3_rls.hpp"

namespace pols {

constexpr int K3_BLK = 64;     // rows staged per block
constexpr int K3_KMAX = 8;

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) { return dpp_get<CTRL>(v); }

// sum over the 8 lanes that share i (lane bits 0..2); every lane gets the total
__device__ __forceinline__ double row8_allreduce(double v) {
    v += dpp_f64<0xB1>(v);    // xor 1
    v += dpp_f64<0x4E>(v);    // xor 2
    v += dpp_f64<0x141>(v);   // row_half_mirror: lane l <-> 7 - l inside each 8
    return v;
}

__device__ __forceinline__ double swap16_allreduce(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
__device__ __forceinline__ double swap32_allreduce(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    return __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
// sum over the 8 lanes that share j (lane bits 3..5); every lane gets the total
__device__ __forceinline__ double col8_allreduce(double v) {
    v += dpp_f64<0x128>(v);   // row_ror:8 -> lane l <-> l ^ 8 inside each 16
    v = swap16_allreduce(v);  // l <-> l ^ 16
    v = swap32_allreduce(v);  // l <-> l ^ 32
    return v;
}

template <typename T>
__global__ void __launch_bounds__(64) k3_rls_kernel(const K3Args a) {
    __shared__ double xs[K3_BLK][K3_KMAX + 2];   // [row][x_0..x_{k-1}, y, valid]
    __shared__ double cs[K3_BLK][K3_KMAX + 1];   // [row][beta_0..beta_{k-1}, pred]
    const int lane = threadIdx.x;
    const int i = lane >> 3, j = lane & 7;
    const int k = a.k;
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const bool act = (i < k) && (j < k);

    const double ff = a.forgetting_factor;
    double P = (act && i == j) ? a.initial_state_covariance : 0.0;                   // P0 = lam * I (:520)
    double beta = (i < k && a.mean0) ? a.mean0[i] : 0.0;                            // coef (:519-522), replicated over j
    T *coef = static_cast<T *>(a.coef);
    T *pred = static_cast<T *>(a.pred);

    for (int64_t t0 = s; t0 < e; t0 += K3_BLK) {
        const int nb = (int)min((int64_t)K3_BLK, e - t0);
        // ---- stage one block of rows (lane = row)
        if (lane < nb) {
            const int64_t r = t0 + lane;
            for (int c = 0; c < k; ++c) xs[lane][c] = (double)static_cast<const T *>(a.x[c])[r];
            xs[lane][k] = (double)static_cast<const T *>(a.y)[r];
            xs[lane][k + 1] = a.valid ? (double)a.valid[r] : 1.0;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- the recursion
        for (int st = 0; st < nb; ++st) {
            const double xi = (i < k) ? xs[st][i] : 0.0;
            const double xj = (j < k) ? xs[st][j] : 0.0;
            const double y = xs[st][k];
            const bool valid = xs[st][k + 1] != 0.0;
            double pr = col8_allreduce(xi * beta);                // x . coef (pre-update), same in every lane
            if (valid) {                                         // RecursiveLeastSquares::update (:531-540)
                const double Pxi = row8_allreduce(P * xj);        // (P x)_i
                const double Pxj = col8_allreduce(P * xi);        // (P x)_j  (P symmetric)
                const double q = col8_allreduce(xi * Pxi);        // x' P x
                const double r = 1.0 + q / ff;
                const double den = r * ff;
                const double ki = Pxi / den, kj = Pxj / den;      // kalman gain entries i and j
                const double resid = y - pr;
                beta = beta + ki * resid;
                P = P / ff - (ki * kj) * r;
                if (!act) P = 0.0;
                pr = pr + (q / den) * resid;                      // x . coef (post-update): x'k = x'Px / (r ff)
            }
            if (j == 0 && i < k) cs[st][i] = beta;                // coefficients[t, :] = coef (:592-594)
            if (lane == 0) cs[st][k] = pr;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- write the block back, coalesced
        if (coef) {
            const int total = nb * k;
            for (int q = lane; q < total; q += 64) coef[t0 * k + q] = (T)cs[q / k][q - (q / k) * k];
        }
        if (pred && lane < nb) pred[t0 + lane] = (T)cs[lane][k];
        __builtin_amdgcn_wave_barrier();
    }
}

int k3_launch(pols_ctx *ctx, int dtype, const K3Args &a) {
    if (a.k > K3_KMAX) return fail(POLS_ERR_UNSUPPORTED, "rls: %d features > %d", a.k, K3_KMAX);
    if (a.n_groups > 0x7fffffffLL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    ctx->last_kernel = dtype == POLS_F32 ? "k3_rls_f32" : "k3_rls_f64";
    timing_begin(ctx);
    if (dtype == POLS_F32) hipLaunchKernelGGL(k3_rls_kernel<float>, dim3((unsigned)a.n_groups), dim3(64), 0, ctx->stream, a);
    else hipLaunchKernelGGL(k3_rls_kernel<double>, dim3((unsigned)a.n_groups), dim3(64), 0, ctx->stream, a);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
