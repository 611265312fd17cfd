// k1m_kernel.inl -- K1m "gram_chol_predict, LDS tile + MFMA Gram": same contract as K1 (k1_gram_chol.hpp), other mapping.
//
// One 256-thread workgroup per group.
//   stage   : the group's columns (k features, y, optional w) are DMA'd HBM -> LDS with
//             `global_load_lds_dwordx4` (1 KiB per wave-instruction, coalesced down the row axis, no VGPR
//             round trip), laid out column-major with a column stride of 16 B x odd so that both the MFMA
//             operand reads (lane = (column, row-quad)) and the row-parallel prediction reads are
//             bank-conflict free.
//   gram    : Z^T Z for Z = [sqrt(w) X | 1 | sqrt(w) y] as a 16 x 16 tile on the matrix cores:
//             v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 with A = B = one LDS value per lane (lane l feeds
//             Z[row 4t + (l >> 4)][column l & 15]); each wave takes every 4th 8-row step, two accumulators
//             hide the dependent-issue latency; partial tiles are summed across the 4 waves through LDS in a
//             fixed order.  Accumulator cost: 8 VGPRs instead of the (k+1)(k+2)/2 of the VALU form.
//   solve   : every lane runs the unrolled K x K Cholesky + triangular solves on wave-uniform values.
//   predict : X . beta from the LDS-resident tile, 16-byte coalesced stores.  X is read from HBM once.
#include "k1_kernel.inl"

namespace pols {

template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
    using acc_t = __attribute__((ext_vector_type(4))) float;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    static __device__ __forceinline__ constexpr int slot(int i, int j) { return (i & 3) * 64 + (i >> 2) * 16 + j; }
};
template <> struct Mfma16<double> {
    using acc_t = __attribute__((ext_vector_type(4))) double;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ constexpr int slot(int i, int j) { return (i >> 2) * 64 + (i & 3) * 16 + j; }
};

// Row stride (elements) of the LDS tile: >= rows, a multiple of the 16-byte vector, and (stride in 16-byte
// units) odd -> the 16 column bases fall on 16 distinct 16-byte slots of the 256-byte LDS bank row.
template <typename T>
static inline int k1m_row_stride(int64_t max_rows) {
    constexpr int VEC = Vec16<T>::N;
    int64_t need = max_rows + (VEC - 1);          // head slack: the tile starts at the 16-byte boundary below s
    need = (need + 7) & ~(int64_t)7;              // the Gram loop walks 8-row steps
    int64_t units = (need + VEC - 1) / VEC;
    if ((units & 1) == 0) units += 1;
    return (int)(units * VEC);
}

template <typename T, int KT>
__global__ void __launch_bounds__(256) k1m_kernel(const K1Args a, const int rs, const int ncols) {
    using V = typename Vec16<T>::type;
    using M = Mfma16<T>;
    using acc_t = typename M::acc_t;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NZ = KT + 1;
    static_assert(NZ <= 16, "one 16x16 MFMA tile holds [X | y]");
    constexpr int NACC = NZ * (NZ + 1) / 2;
    constexpr int RPP = 64 * VEC;  // rows one DMA wave-instruction moves

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *tile = reinterpret_cast<T *>(smem);                 // [ncols][rs]: x_0..x_{ku-1}, y, (w)
    T *part = tile + (size_t)ncols * rs;                   // [4 waves][4 regs][64 lanes]; wave 0's slab later holds the sum

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int head = (int)(s - base);
    const int span = (int)(e - base);                      // tile rows [head, span) belong to the group
    const int ku = a.k_user;
    const bool has_w = a.w != nullptr;
    const bool icpt = ku != KT;

    // ---- stage: HBM -> LDS, one 1 KiB piece per wave-instruction
    const int ppc = (span + RPP - 1) / RPP;
    const int npieces = ncols * ppc;
    for (int p = wave; p < npieces; p += 4) {
        const int col = p / ppc, q = p - col * ppc;
        const T *src = static_cast<const T *>(col < ku ? a.x[col] : (col == ku ? a.y : a.w));
        const int row0 = q * RPP + lane * VEC;
        const int64_t grow = base + row0;
        T *ldst = tile + (size_t)col * rs + q * RPP;       // wave-uniform; the DMA adds lane * 16 bytes itself
        if (row0 < span) {
            if (grow + VEC <= a.n_rows) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + grow),
                                                 (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
            } else {  // last chunk of the whole frame when n_rows is not a multiple of the vector width
#pragma unroll
                for (int v = 0; v < VEC; ++v) ldst[lane * VEC + v] = (grow + v < a.n_rows) ? src[grow + v] : T(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (has_w) {  // w <- sqrt(w) once (sqrt_w of least_squares.py:193); NaN for w < 0 propagates like the reference
        T *wc = tile + (size_t)(ku + 1) * rs;
        for (int r = tid; r < span; r += 256) wc[r] = sqrt(wc[r]);
        __syncthreads();
    }

    // ---- Gram on the matrix cores
    const int zc = lane & 15, kq = lane >> 4;
    const bool z_real = (zc < ku) || (zc == KT);
    const bool z_one = icpt && (zc == KT - 1);
    const T *zcol = tile + (size_t)(zc < ku ? zc : ku) * rs;
    const T *wcol = tile + (size_t)(ku + 1) * rs;
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int t = wave * 8; t < span; t += 32) {
        int r0, r1;
        T v0, v1, w0 = T(1), w1 = T(1);
        if constexpr (sizeof(T) == 4) {   // one ds_read_b64: rows t + 2kq, t + 2kq + 1
            r0 = t + 2 * kq; r1 = r0 + 1;
            const float2 vv = *reinterpret_cast<const float2 *>(zcol + r0);
            v0 = vv.x; v1 = vv.y;
            if (has_w) { const float2 ww = *reinterpret_cast<const float2 *>(wcol + r0); w0 = ww.x; w1 = ww.y; }
        } else {                          // two ds_read_b64: rows t + kq, t + 4 + kq
            r0 = t + kq; r1 = r0 + 4;
            v0 = zcol[r0]; v1 = zcol[r1];
            if (has_w) { w0 = wcol[r0]; w1 = wcol[r1]; }
        }
        const bool in0 = (r0 >= head) && (r0 < span), in1 = (r1 >= head) && (r1 < span);
        const T a0 = in0 ? (z_one ? w0 : (z_real ? v0 * w0 : T(0))) : T(0);
        const T a1 = in1 ? (z_one ? w1 : (z_real ? v1 * w1 : T(0))) : T(0);
        acc0 = M::mma(a0, a0, acc0);
        acc1 = M::mma(a1, a1, acc1);
    }
    acc0 += acc1;

    // ---- cross-wave sum, fixed order: waves 1..3 publish, wave 0 adds them to its own tile
    if (wave != 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(wave * 4 + r) * 64 + lane] = acc0[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T t = acc0[r];
#pragma unroll
            for (int w = 1; w < 4; ++w) t += part[(w * 4 + r) * 64 + lane];
            part[r * 64 + lane] = t;
        }
    }
    __syncthreads();

    // ---- K x K solve (every lane, wave-uniform values)
    T gacc[NACC];
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = i; j < NZ; ++j) gacc[tri_index<NZ>(i, j)] = part[M::slot(i, j)];
    T beta[KT];
    int st = POLS_GROUP_OK;
    if (e == s) {
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = T(0);
        st = POLS_GROUP_EMPTY;
    } else {
        const bool ok = chol_solve<T, KT>(gacc, (T)a.alpha, beta);
        if (!ok) st = POLS_GROUP_FALLBACK;
    }
    if (tid == 0 && a.status) a.status[g] = st;
    if (a.coef && tid < KT) {
        T bv = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) bv = (tid == j) ? beta[j] : bv;
        static_cast<T *>(a.coef)[g * KT + tid] = bv;
    }

    // ---- predictions / residuals from the LDS tile
    if (a.pred || a.resid) {
        T *pred = static_cast<T *>(a.pred);
        T *resid = static_cast<T *>(a.resid);
        const T *ycol = tile + (size_t)ku * rs;
        for (int row0 = tid * VEC; row0 < span; row0 += 256 * VEC) {
            T p[VEC], sw[VEC];
            if (has_w) {
                const V wv = *reinterpret_cast<const V *>(wcol + row0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) sw[v] = vget<T>(wv, v);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) sw[v] = T(1);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) p[v] = T(0);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j < ku) {
                    const V xv = *reinterpret_cast<const V *>(tile + (size_t)j * rs + row0);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) p[v] = fma(vget<T>(xv, v) * sw[v], beta[j], p[v]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) p[v] = fma(sw[v], beta[j], p[v]);   // intercept column (ones * sqrt_w)
                }
            }
            if (has_w) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[v] *= T(1) / sw[v];                  // least_squares.py:234-235
            }
            const V yv = *reinterpret_cast<const V *>(ycol + row0);
            if (row0 >= head && row0 + VEC <= span) {
                if (pred) {
                    V o;
                    if constexpr (VEC == 4) o = V{p[0], p[1], p[2], p[3]}; else o = V{p[0], p[1]};
                    *reinterpret_cast<V *>(pred + base + row0) = o;
                }
                if (resid) {
                    V o;
                    if constexpr (VEC == 4) o = V{yv.x - p[0], yv.y - p[1], yv.z - p[2], yv.w - p[3]};
                    else o = V{yv.x - p[0], yv.y - p[1]};
                    *reinterpret_cast<V *>(resid + base + row0) = o;
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int r = row0 + v;
                    if (r >= head && r < span) {
                        if (pred) pred[base + r] = p[v];
                        if (resid) resid[base + r] = vget<T>(yv, v) - p[v];
                    }
                }
            }
        }
    }
}

template <typename T>
static inline size_t k1m_lds_bytes(int rs, int ncols) {
    return sizeof(T) * ((size_t)ncols * rs + 4 * 4 * 64);
}

template <typename T, int KT>
static int k1m_launch_kt(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    const int ncols = a.k_user + 1 + (a.w ? 1 : 0);
    const int rs = k1m_row_stride<T>(max_rows);
    const size_t lds = k1m_lds_bytes<T>(rs, ncols);
    static bool attr_set = false;
    if (!attr_set) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1m_kernel<T, KT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    char name[96];
    std::snprintf(name, sizeof(name), "k1m_gram_mfma_%s_k%d_lds%zu", sizeof(T) == 4 ? "f32" : "f64", KT, lds);
    ctx->last_kernel = name;
    timing_begin(ctx);
    hipLaunchKernelGGL((k1m_kernel<T, KT>), dim3((unsigned)a.n_groups), dim3(256), lds, ctx->stream, a, rs, ncols);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// true when the largest group's tile fits the 160 KiB LDS of a CU
template <typename T>
bool k1m_fits(int k_user, bool has_w, int64_t max_rows) {
    if (max_rows > (1 << 20)) return false;
    const int ncols = k_user + 1 + (has_w ? 1 : 0);
    return k1m_lds_bytes<T>(k1m_row_stride<T>(max_rows), ncols) <= 160 * 1024;
}

template <typename T>
int k1m_launch_t(pols_ctx *ctx, int kt, const K1Args &a, int64_t max_rows) {
    switch (kt) {
        case 1: return k1m_launch_kt<T, 1>(ctx, a, max_rows);
        case 2: return k1m_launch_kt<T, 2>(ctx, a, max_rows);
        case 3: return k1m_launch_kt<T, 3>(ctx, a, max_rows);
        case 4: return k1m_launch_kt<T, 4>(ctx, a, max_rows);
        case 5: return k1m_launch_kt<T, 5>(ctx, a, max_rows);
        case 6: return k1m_launch_kt<T, 6>(ctx, a, max_rows);
        case 7: return k1m_launch_kt<T, 7>(ctx, a, max_rows);
        case 8: return k1m_launch_kt<T, 8>(ctx, a, max_rows);
        case 9: return k1m_launch_kt<T, 9>(ctx, a, max_rows);
        case 10: return k1m_launch_kt<T, 10>(ctx, a, max_rows);
        case 11: return k1m_launch_kt<T, 11>(ctx, a, max_rows);
        case 12: return k1m_launch_kt<T, 12>(ctx, a, max_rows);
        case 13: return k1m_launch_kt<T, 13>(ctx, a, max_rows);
        case 14: return k1m_launch_kt<T, 14>(ctx, a, max_rows);
        case 15: return k1m_launch_kt<T, 15>(ctx, a, max_rows);
        default: return fail(POLS_ERR_UNSUPPORTED, "k1m: %d features (incl. intercept) > 15", kt);
    }
}

}  // namespace pols
