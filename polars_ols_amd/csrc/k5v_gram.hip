// k5v_gram.hip -- K5v: the streamed Gram pass on the VALU, for up to ten columns (sample weights and null policies applied on the registers as the rows are loaded).
//
// The streamed path (K5: a Gram pass, the small solve, a prediction pass) serves every group that is too long to stay in registers --
// and the one-regression-over-the-whole-frame call of the reference's README (long groups cut into segments).  Its Gram kernel puts
// Z'Z (Z = [X | y]) on the matrix cores in 16-column tiles, which is the right shape at 16 or 31 columns; at 9 columns a third of the tile
// is useful, half of that is the mirror image, and on this part the f32 / f64 MFMA peaks EQUAL the vector peaks (157 / 78.6 TFLOP/s), so
// the tile only costs: the f64 16x16x4 MFMA is 64 cycles for 4 rows -- 54 % of the matrix pipe at the HBM rate for 72-byte rows, behind
// LDS staging with a barrier per chunk.  gram_stream measured 3.8 / 4.2 TB/s (f32 / f64, 8 features) before and 4.4 / 4.0 after 4 KB
// pieces and double buffering (profiles/r05_long_groups_ab.txt).
//
// Here every lane loads 16-byte vectors (4 f32 / 2 f64 rows) of each column straight into registers -- 4 KB pieces per column per
// workgroup step, nothing staged -- and accumulates the (kt + 1)(kt + 2) / 2 packed products of its own rows: 45 FMAs per row at 8
// columns, 19 % of the vector pipe at the HBM rate.  The lanes' sums are converted to f64, reduce-scattered over the wave into the wave's
// f64 totals in LDS, and added across the four waves in wave order: the Gram matrix leaves in f64 in the layout gram_stream_kernel writes,
// for the same consumers (gram_reduce, gram_solve, gram_cd).  f32 frames: a lane's f32 partial is flushed into those f64 totals every
// K5V_FLUSH_STEPS steps (128 of its rows) -- an item is as long as its group or segment, up to millions of rows in the statistics entry,
// and the MFMA pass this kernel replaced flushed to f64 every 64 rows per wave: that flush is what holds the streamed f32 paths to 1e-4.
#include "k5_enet.hpp"

namespace pols {

__device__ __forceinline__ void k5v_scale(float4 &z, const float (&sw)[4]) { z.x *= sw[0]; z.y *= sw[1]; z.z *= sw[2]; z.w *= sw[3]; }
__device__ __forceinline__ void k5v_scale(double2 &z, const double (&sw)[2]) { z.x *= sw[0]; z.y *= sw[1]; }

template <int NZ>
__host__ __device__ constexpr int k5v_tri(int i, int j) { return i * NZ - i * (i - 1) / 2 + (j - i); }   // packed upper triangle, i <= j < NZ

// HAS_W: sample weights -- every column of [X | 1 | y] scaled by sqrt(w) as it is loaded (least_squares.py:190-196).
// NULLS: a null policy (src/expressions.rs:201-296), what gram_stream_kernel's prep pass does in LDS done on the registers: rows the policy drops
// (null target / null feature / validity byte 0, by policy) become zero rows, nulls that stay in the fit become 0, and the rows left in the fit are
// counted into a.nvalid (the n of alpha * n, and "no row left" = an empty group).
template <typename T, int KT, bool HAS_W, bool NULLS>
__global__ void __launch_bounds__(256) gram_valu_kernel(const GramArgs a) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1, NACC = NZ * (NZ + 1) / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs_pairs ? a.offs[2 * g] : a.offs[g], e = a.offs_pairs ? a.offs[2 * g + 1] : a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int ku = a.k_user;                                 // KT or KT - 1: only the last slot can be the intercept column

    T acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = T(0);
    const int pol = a.null_policy;
    int nfit = 0;                                            // NULLS: this lane's rows that take part in the fit

    // lanes -> wave: f64 reduce-scatter, 32 entries at a time (an f64 copy of all of them would double the registers at 11-16 columns), ADDED
    // to the wave's totals in LDS (every entry has one owner lane: no race); acc starts over
    constexpr int SL = 32, NSL = (NACC + SL - 1) / SL;
    __shared__ double part[4][NSL * SL];
    for (int q = lane; q < NSL * SL; q += 64) part[wave][q] = 0.0;
    auto flush = [&]() {
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            double accd[SL];
#pragma unroll
            for (int q = 0; q < SL; ++q) accd[q] = (sl * SL + q < NACC) ? (double)acc[(sl * SL + q < NACC) ? sl * SL + q : 0] : 0.0;
            double u[SL / 4];
            wave_reduce_scatter<double, SL>(accd, u);       // u[i], in every lane of 16-lane row r: the wave total of entry 4 i + rs_perm(r) of the slice
            if ((lane & 15) == 0) {
                const int r = rs_perm(lane >> 4);
#pragma unroll
                for (int i = 0; i < SL / 4; ++i) part[wave][sl * SL + 4 * i + r] += u[i];
            }
        }
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = T(0);
    };
    constexpr int K5V_FLUSH_STEPS = 32;                      // f32: 32 steps x 4 rows per lane between flushes
    int since = 0;

    // (a counted loop, the same trip count in every lane: the flush is a cross-lane step and must meet the whole wave -- lanes past the
    // item's last row skip the body and rejoin)
    const int64_t n_steps = (e - base + 256 * VEC - 1) / (256 * VEC);
    for (int64_t it = 0; it < n_steps; ++it) {
        const int64_t row0 = base + (it * 256 + tid) * VEC;
        if constexpr (sizeof(T) == 4) {
            if (since == K5V_FLUSH_STEPS) { flush(); since = 0; }
            ++since;
        }
        if (row0 >= e) continue;
        V z[NZ];
        if (row0 >= s && row0 + VEC <= e) {
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j < KT - 1 || j < ku) z[j] = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + row0));
                else { if constexpr (VEC == 4) z[j] = V{T(1), T(1), T(1), T(1)}; else z[j] = V{T(1), T(1)}; }
            }
            z[KT] = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0));
            if constexpr (HAS_W || NULLS) {
                T sw[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) sw[v] = T(1);
                if constexpr (HAS_W) {
                    const V wv = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0));
#pragma unroll
                    for (int v = 0; v < VEC; ++v) sw[v] = sqrt(vget<T>(wv, v));
                }
                if constexpr (NULLS) {
                    bool ok[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ok[v] = true;
                    if (a.valid && null_checks_y(pol)) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) ok[v] = a.valid[row0 + v] != 0;
                    }
                    if (null_checks_y(pol)) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { const T yv = vget<T>(z[KT], v); ok[v] = ok[v] && (yv == yv); }
                    }
                    if (null_checks_x(pol)) {
#pragma unroll
                        for (int j = 0; j < KT; ++j)
#pragma unroll
                            for (int v = 0; v < VEC; ++v) { const T xv = vget<T>(z[j], v); ok[v] = ok[v] && (xv == xv); }   // (the ones column is never a null)
                    }
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { nfit += ok[v] ? 1 : 0; sw[v] = ok[v] ? sw[v] : T(0); }
#pragma unroll
                    for (int j = 0; j < NZ; ++j) {             // nulls that stay -> 0 (handle_nulls, ex.rs:257-296); dropped rows -> zero rows (a NaN x 0 would stay NaN)
                        T t[VEC];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) t[v] = ok[v] ? null_fill<T>(pol, vget<T>(z[j], v)) : T(0);
                        if constexpr (VEC == 4) z[j] = V{t[0], t[1], t[2], t[3]}; else z[j] = V{t[0], t[1]};
                    }
                }
#pragma unroll
                for (int j = 0; j < NZ; ++j) k5v_scale(z[j], sw);
            }
        } else {                                             // the segment's first / last vector: rows outside it are zero rows
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                T t[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int64_t r = row0 + v;
                    const bool in = r >= s && r < e;
                    if (j == KT) t[v] = in ? static_cast<const T *>(a.y)[r] : T(0);
                    else if (j < KT - 1 || j < ku) t[v] = in ? static_cast<const T *>(a.x[j])[r] : T(0);
                    else t[v] = in ? T(1) : T(0);
                    bool use = in;
                    if constexpr (NULLS) {
                        use = in && null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, r);
                        t[v] = use ? null_fill<T>(pol, t[v]) : T(0);
                        if (j == KT) nfit += use ? 1 : 0;
                    }
                    if constexpr (HAS_W) t[v] = use ? t[v] * sqrt(static_cast<const T *>(a.w)[r]) : T(0);
                }
                if constexpr (VEC == 4) z[j] = V{t[0], t[1], t[2], t[3]}; else z[j] = V{t[0], t[1]};
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int i = 0; i < NZ; ++i) {
                const T zi = vget<T>(z[i], v);
#pragma unroll
                for (int j = i; j < NZ; ++j) acc[k5v_tri<NZ>(i, j)] = fma(zi, vget<T>(z[j], v), acc[k5v_tri<NZ>(i, j)]);
            }
    }

    flush();
    __shared__ int nfit_s;
    if constexpr (NULLS) { if (tid == 0) nfit_s = 0; }
    __syncthreads();
    if constexpr (NULLS) {
        if (nfit) atomicAdd(&nfit_s, nfit);
        __syncthreads();
        if (tid == 0 && a.nvalid) a.nvalid[g] = (double)nfit_s;
    }
    double *G = a.gram + (size_t)g * NZ * NZ;
    for (int t = tid; t < NZ * NZ; t += 256) {
        const int i = t / NZ, j = t - i * NZ;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int q = lo * NZ - lo * (lo - 1) / 2 + (hi - lo);
        G[t] = ((part[0][q] + part[1][q]) + part[2][q]) + part[3][q];
    }
}

template <typename T, int KT, bool HAS_W, bool NULLS>
static void k5v_go_w(pols_ctx *ctx, const GramArgs &a) {
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1)) hipExtLaunchKernelGGL((gram_valu_kernel<T, KT, HAS_W, NULLS>), dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, ev0, ev1, 0, a);
    else hipLaunchKernelGGL((gram_valu_kernel<T, KT, HAS_W, NULLS>), dim3((unsigned)a.n_groups), dim3(256), 0, ctx->stream, a);
}
template <typename T, int KT>
static void k5v_go(pols_ctx *ctx, const GramArgs &a) {
    const bool nulls = a.null_policy != POLS_NULL_IGNORE;
    if (a.w) { if (nulls) k5v_go_w<T, KT, true, true>(ctx, a); else k5v_go_w<T, KT, true, false>(ctx, a); }
    else { if (nulls) k5v_go_w<T, KT, false, true>(ctx, a); else k5v_go_w<T, KT, false, false>(ctx, a); }
}

template <typename T>
static int k5v_launch_t(pols_ctx *ctx, const GramArgs &a) {
    switch (a.kt) {
        case 1: k5v_go<T, 1>(ctx, a); break;
        case 2: k5v_go<T, 2>(ctx, a); break;
        case 3: k5v_go<T, 3>(ctx, a); break;
        case 4: k5v_go<T, 4>(ctx, a); break;
        case 5: k5v_go<T, 5>(ctx, a); break;
        case 6: k5v_go<T, 6>(ctx, a); break;
        case 7: k5v_go<T, 7>(ctx, a); break;
        case 8: k5v_go<T, 8>(ctx, a); break;
        case 9: k5v_go<T, 9>(ctx, a); break;
        case 10: k5v_go<T, 10>(ctx, a); break;
        default:
            if constexpr (sizeof(T) == 4) {                  // f32 only: 78-105 accumulators + 12-14 vectors in flight fit 256 registers; f64 would need twice that
                switch (a.kt) {
                    case 11: k5v_go<T, 11>(ctx, a); break;
                    case 12: k5v_go<T, 12>(ctx, a); break;
                    case 13: k5v_go<T, 13>(ctx, a); break;
                    default: return fail(POLS_ERR_UNSUPPORTED, "k5v: %d columns", a.kt);
                }
                break;
            }
            return fail(POLS_ERR_UNSUPPORTED, "k5v: %d columns", a.kt);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

bool gram_valu_takes(const pols_ctx *ctx, int dtype, const GramArgs &a) {
    return a.kt >= 1 && a.kt <= (dtype == POLS_F32 ? K5V_MAX_KT_F32 : K5V_MAX_KT) && (a.null_policy == POLS_NULL_IGNORE ? !a.nvalid : a.nvalid != nullptr) && !ctx->opt.kg_single_buffer;
}

int gram_valu_launch(pols_ctx *ctx, int dtype, const GramArgs &a) {
    if (a.n_groups > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    char name[96];
    std::snprintf(name, sizeof(name), "k5_gram_stream_%s_valu%s%s_k%d", dtype == POLS_F32 ? "f32" : "f64", a.w ? "_w" : "", a.null_policy != POLS_NULL_IGNORE ? "_nulls" : "", a.kt);
    ctx->last_kernel = name;
    return dtype == POLS_F32 ? k5v_launch_t<float>(ctx, a) : k5v_launch_t<double>(ctx, a);
}

}  // namespace pols
