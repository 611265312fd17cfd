// k1m_kernel.inl -- K1m "gram_chol_predict, LDS tile + MFMA Gram": same contract as K1 (k1_gram_chol.hpp), other mapping.
//
// One 256-thread workgroup per group.
//   stage   : the group's columns (k features, y, optional w) are DMA'd HBM -> LDS with
//             `global_load_lds_dwordx4` (1 KiB per wave-instruction, coalesced down the row axis, no VGPR
//             round trip), laid out column-major with a column stride of 16 B x odd so that both the MFMA
//             operand reads (lane = (column, row-quad)) and the row-parallel prediction reads are
//             bank-conflict free.
//   prep    : rows of the tile outside the group (alignment head, 8-row tail pad) are zeroed; with sample
//             weights every column is scaled by sqrt(w) in place (least_squares.py:190-196), so the Gram
//             loop needs no masks, selects or multiplies.
//   gram    : Z^T Z for Z = [sqrt(w) X | sqrt(w) 1 | sqrt(w) y] as a 16 x 16 tile on the matrix cores:
//             v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 with A = B = one LDS value per lane (lane l feeds
//             Z[row 4t + (l >> 4)][column l & 15]).  Each wave owns a contiguous quarter of the 8-row steps;
//             the loop body is ONE ds_read_b64 + TWO MFMAs per step (lanes of the unused tile columns read a
//             block of zeros, the intercept lanes a block of ones), two accumulators hide the dependent-issue
//             latency.  Partial tiles are summed across the 4 waves through LDS in a fixed order.
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

constexpr int K1M_CONST_ELEMS = 32;  // zeros block + ones block: 32 elements each (>= one 4-step batch of reads)

template <typename T, int KT, bool HAS_W>
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
    T *zeros = part + 4 * 4 * 64;                          // K1M_CONST_ELEMS zeros, then K1M_CONST_ELEMS ones
    T *ones = zeros + K1M_CONST_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int head = (int)(s - base);
    const int span = (int)(e - base);                      // tile rows [head, span) belong to the group
    const int span8 = (span + 7) & ~7;
    const int ku = a.k_user;
    const bool icpt = ku != KT;
    unsigned long long *dbg = a.dbg ? a.dbg + g * 8 : nullptr;
#define K1M_STAMP(i) do { if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    K1M_STAMP(0);

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
    if (tid < 2 * K1M_CONST_ELEMS) zeros[tid] = (tid < K1M_CONST_ELEMS) ? T(0) : T(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    K1M_STAMP(1);

    // ---- prep: zero the rows outside [head, span); with weights scale every column by sqrt(w) in place
    if constexpr (HAS_W) {
        T *wc = tile + (size_t)(ku + 1) * rs;
        for (int row0 = tid * VEC; row0 < span8; row0 += 256 * VEC) {
            V wv = *reinterpret_cast<V *>(wc + row0);
            T sw[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int r = row0 + v;
                sw[v] = (r >= head && r < span) ? sqrt(vget<T>(wv, v)) : T(0);   // NaN for w < 0, like the reference
            }
            if constexpr (VEC == 4) wv = V{sw[0], sw[1], sw[2], sw[3]}; else wv = V{sw[0], sw[1]};
            *reinterpret_cast<V *>(wc + row0) = wv;
            for (int c = 0; c <= ku; ++c) {
                V xv = *reinterpret_cast<V *>(tile + (size_t)c * rs + row0);
                T t[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int r = row0 + v;
                    t[v] = (r >= head && r < span) ? vget<T>(xv, v) * sw[v] : T(0);
                }
                if constexpr (VEC == 4) xv = V{t[0], t[1], t[2], t[3]}; else xv = V{t[0], t[1]};
                *reinterpret_cast<V *>(tile + (size_t)c * rs + row0) = xv;
            }
        }
    } else {
        const int npad = head + (span8 - span);            // < 4 + 8 rows
        for (int i = tid; i < npad * ncols; i += 256) {
            const int c = i / npad, k = i - c * npad;
            const int r = (k < head) ? k : span + (k - head);
            tile[(size_t)c * rs + r] = T(0);
        }
    }
    __syncthreads();

    // ---- Gram on the matrix cores: per step one ds_read_b64 (two rows of this lane's column) + two MFMAs
    const int zc = lane & 15, kq = lane >> 4;
    const int rlane = (sizeof(T) == 4) ? 2 * kq : kq;
    const int nsteps = span8 >> 3;
    const int per_wave = (nsteps + 3) >> 2;
    const int t_begin = min(nsteps, wave * per_wave), t_end = min(nsteps, t_begin + per_wave);
    const T *zp;      // this lane's operand stream
    int zinc;         // elements per 8-row step: 8 for tile columns, 0 for the constant blocks
    if (zc < ku) { zp = tile + (size_t)zc * rs + t_begin * 8 + rlane; zinc = 8; }
    else if (zc == KT) { zp = tile + (size_t)ku * rs + t_begin * 8 + rlane; zinc = 8; }                       // y
    else if (icpt && zc == KT - 1) {
        if constexpr (HAS_W) { zp = tile + (size_t)(ku + 1) * rs + t_begin * 8 + rlane; zinc = 8; }          // sqrt(w) * 1
        else { zp = ones; zinc = 0; }
    } else { zp = zeros; zinc = 0; }
    acc_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    int n = t_end - t_begin;
    for (; n >= 4; n -= 4) {
        T v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (sizeof(T) == 4) {
                const float2 vv = *reinterpret_cast<const float2 *>(zp + u * 8);   // rows r, r + 1
                v0[u] = vv.x; v1[u] = vv.y;
            } else {
                v0[u] = zp[u * 8]; v1[u] = zp[u * 8 + 4];                          // rows r, r + 4
            }
        }
        // constant-block lanes keep re-reading the same 4 steps' worth of zeros / ones (zinc == 0)
        zp += 4 * zinc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc0 = M::mma(v0[u], v0[u], acc0);
            acc1 = M::mma(v1[u], v1[u], acc1);
        }
    }
    for (; n > 0; --n) {
        T v0, v1;
        if constexpr (sizeof(T) == 4) { const float2 vv = *reinterpret_cast<const float2 *>(zp); v0 = vv.x; v1 = vv.y; }
        else { v0 = zp[0]; v1 = zp[4]; }
        zp += zinc;
        acc0 = M::mma(v0, v0, acc0);
        acc1 = M::mma(v1, v1, acc1);
    }
    acc0 += acc1;
    K1M_STAMP(2);

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
    K1M_STAMP(3);

    // ---- K x K solve (every lane, wave-uniform values)
    T gacc[NACC];
#pragma unroll
    for (int i = 0; i < NZ; ++i)
#pragma unroll
        for (int j = i; j < NZ; ++j) gacc[tri_index<NZ>(i, j)] = part[M::slot(i, j)];
    if constexpr (!HAS_W) {
        // the ones block also fed the pad rows (where every other column is zero): only sum(1*1) is off
        if (icpt) gacc[tri_index<NZ>(KT - 1, KT - 1)] = (T)(span - head);
    }
    T beta[KT];
    int st = POLS_GROUP_OK;
    if (e == s) {
#pragma unroll
        for (int j = 0; j < KT; ++j) beta[j] = T(0);
        st = POLS_GROUP_EMPTY;
    } else {
        const bool ok = chol_solve<T, KT>(gacc, (T)a.alpha, beta, (T)a.pivot_tol);
        if (!ok) { st = POLS_GROUP_FALLBACK; if (tid == 0 && a.fb_flag) *a.fb_flag = a.epoch; }
    }
    if (tid == 0 && a.status) a.status[g] = st;
    if (a.coef && tid < KT) {
        T bv = T(0);
#pragma unroll
        for (int j = 0; j < KT; ++j) bv = (tid == j) ? beta[j] : bv;
        static_cast<T *>(a.coef)[g * KT + tid] = bv;
    }
    K1M_STAMP(4);

    // ---- predictions / residuals from the LDS tile (already sqrt(w)-scaled when HAS_W)
    if (a.pred || a.resid) {
        T *pred = static_cast<T *>(a.pred);
        T *resid = static_cast<T *>(a.resid);
        const T *ycol = tile + (size_t)ku * rs;
        const T *wcol = tile + (size_t)(ku + 1) * rs;
        for (int row0 = tid * VEC; row0 < span; row0 += 256 * VEC) {
            T p[VEC], sw[VEC], yo[VEC];
            if constexpr (HAS_W) {
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
                    for (int v = 0; v < VEC; ++v) p[v] = fma(vget<T>(xv, v), beta[j], p[v]);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) p[v] = fma(sw[v], beta[j], p[v]);   // intercept column (ones * sqrt_w)
                }
            }
            if constexpr (HAS_W) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[v] *= T(1) / sw[v];                  // least_squares.py:234-235
            }
            const bool full = (row0 >= head) && (row0 + VEC <= span);
            if (resid) {   // residuals use the ORIGINAL target (least_squares.py:239); the tile's y is scaled when HAS_W
                if constexpr (HAS_W) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int r = row0 + v;
                        yo[v] = (r >= head && r < span) ? static_cast<const T *>(a.y)[base + r] : T(0);
                    }
                } else {
                    const V yv = *reinterpret_cast<const V *>(ycol + row0);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) yo[v] = vget<T>(yv, v);
                }
            }
            if (full) {
                if (pred) {
                    V o;
                    if constexpr (VEC == 4) o = V{p[0], p[1], p[2], p[3]}; else o = V{p[0], p[1]};
                    store_stream(reinterpret_cast<V *>(pred + base + row0), o);
                }
                if (resid) {
                    V o;
                    if constexpr (VEC == 4) o = V{yo[0] - p[0], yo[1] - p[1], yo[2] - p[2], yo[3] - p[3]};
                    else o = V{yo[0] - p[0], yo[1] - p[1]};
                    store_stream(reinterpret_cast<V *>(resid + base + row0), o);
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int r = row0 + v;
                    if (r >= head && r < span) {
                        if (pred) pred[base + r] = p[v];
                        if (resid) resid[base + r] = yo[v] - p[v];
                    }
                }
            }
        }
    }
    K1M_STAMP(5);
    if (dbg && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[6] = xcc;
    }
#undef K1M_STAMP
}

template <typename T>
static inline size_t k1m_lds_bytes(int rs, int ncols) {
    return sizeof(T) * ((size_t)ncols * rs + 4 * 4 * 64 + 2 * K1M_CONST_ELEMS);
}

template <typename T, int KT, bool HAS_W>
static int k1m_launch_kw(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    const int ncols = a.k_user + 1 + (HAS_W ? 1 : 0);
    const int rs = k1m_row_stride<T>(max_rows);
    const size_t lds = k1m_lds_bytes<T>(rs, ncols);
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1m_kernel<T, KT, HAS_W>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done(ctx->device);
    }
    char name[96];
    std::snprintf(name, sizeof(name), "k1m_gram_mfma_%s_k%d%s_lds%zu", sizeof(T) == 4 ? "f32" : "f64", KT, HAS_W ? "_w" : "", lds);
    ctx->last_kernel = name;
    K1Args aa = a;
    const bool timeline = ctx->opt.timeline;
    if (timeline) {
        void *d = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_groups, &d);
        if (rc) return rc;
        aa.dbg = static_cast<unsigned long long *>(d);
    }
    timing_begin(ctx);
    hipLaunchKernelGGL((k1m_kernel<T, KT, HAS_W>), dim3((unsigned)a.n_groups), dim3(256), lds, ctx->stream, aa, rs, ncols);
    timing_end(ctx);
    POLS_HIP(hipGetLastError());
    if (timeline) return report_timeline(ctx, aa.dbg, a.n_groups, 6, name);
    return POLS_OK;
}

template <typename T, int KT>
static int k1m_launch_kt(pols_ctx *ctx, const K1Args &a, int64_t max_rows) {
    return a.w ? k1m_launch_kw<T, KT, true>(ctx, a, max_rows) : k1m_launch_kw<T, KT, false>(ctx, a, max_rows);
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
