// k5_enet.hip -- K5: streamed MFMA Gram, Gram-form coordinate descent, prediction pass (see k5_enet.hpp).
#include "k5_enet.hpp"
#include "k1m_kernel.inl"   // Mfma16, k1m_row_stride, K1M_CONST_ELEMS

namespace pols {

constexpr int KG_CR = 256;   // rows per LDS chunk

// ------------------------------------------------------------------------------------------------ A: gram_stream
// YV (exactly 16 columns of X incl. the intercept, i.e. cfg5's shape): the target would sit alone in a second 16-column tile
// and double the MFMA count for one useful column -- and the f64 16x16x4 MFMA is a 64-cycle instruction on this part.
// X'y is then accumulated on the VALU from the operands the lanes already hold (one extra LDS read + FMA per step) and only
// tile (0, 0) goes through the matrix cores; y'y is not produced (nothing on this path reads it).
template <typename T, int NT, bool HAS_W, bool YV = false>
__global__ void __launch_bounds__(256) gram_stream_kernel(const GramArgs a, const int rs, const int ncols, const int tile_elems, const int nbuf, const int cr) {
    using V = typename Vec16<T>::type;
    using M = Mfma16<T>;
    using acc_t = typename M::acc_t;
    constexpr int VEC = Vec16<T>::N;
    constexpr int RPP = 64 * VEC;
    constexpr int NPAIR = NT * (NT + 1) / 2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [nbuf][ncols][rs]; the first is re-used for the cross-wave partial tiles at the end.  nbuf == 2 (narrow tiles): the DMA of chunk
    // c + 1 is issued before the MFMAs of chunk c, so a workgroup has a chunk in flight the whole time instead of one exposed load
    // latency per 256 rows (gram_stream at 3.8 / 4.2 TB/s f32 / f64 on 8 features, profiles/r05_kernel_stats_long_groups.csv)
    T *const tile0 = reinterpret_cast<T *>(smem);
    T *zeros = tile0 + (size_t)nbuf * tile_elems;
    T *ones = zeros + K1M_CONST_ELEMS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs_pairs ? a.offs[2 * g] : a.offs[g], e = a.offs_pairs ? a.offs[2 * g + 1] : a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int head = (int)(s - base);
    const int64_t span = e - base;                  // rows [head, span) relative to base belong to the group
    const int ku = a.k_user, kt = a.kt, NZ = kt + 1;
    const bool icpt = ku != kt;

    const int zc = lane & 15, kq = lane >> 4;
    const int rlane = (sizeof(T) == 4) ? 2 * kq : kq;
    // operand stream of this lane for each 16-column tile of Z
    int zsrc[NT];      // tile column index, -1 = zeros block, -2 = ones block
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int z = 16 * t + zc;
        if (z < ku) zsrc[t] = z;
        else if (icpt && z == kt - 1) zsrc[t] = HAS_W ? ku + 1 : -2;
        else if (z == kt) zsrc[t] = ku;
        else zsrc[t] = -1;
    }
    const bool need11 = (NT == 2) && (NZ > 17);     // tile (1,1) only carries y'y when k = 16

    acc_t acc[NPAIR][2];
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) { acc[p][0] = acc_t{0, 0, 0, 0}; acc[p][1] = acc_t{0, 0, 0, 0}; }
    T xy0 = T(0), xy1 = T(0);                       // YV: this lane's share of (X'y)[zc]
    double accd[NPAIR][4], xyd = 0.0;               // f64 totals (f32: flushed per chunk; f64: filled once at the end)
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) { accd[p][0] = 0.0; accd[p][1] = 0.0; accd[p][2] = 0.0; accd[p][3] = 0.0; }

    if (tid < 2 * K1M_CONST_ELEMS) zeros[tid] = (tid < K1M_CONST_ELEMS) ? T(0) : T(1);
    __shared__ int nfit_s;
    if (tid == 0) nfit_s = 0;
    int nfit = 0;                                   // rows of the group that take part in the fit (null policies)
    const int pol = a.null_policy;
    const int nld = (HAS_W && !a.w) ? ncols - 1 : ncols;   // no weights column to stage: the prep pass writes ones

    // ---- stage one chunk: HBM -> LDS (asynchronous; the caller waits for vmcnt(0) and the barrier)
    auto stage = [&](const int64_t c0, T *const tl) __attribute__((always_inline)) {
        const int rows_here = (int)min((int64_t)cr, span - c0);
        const int ppc = (rows_here + RPP - 1) / RPP;
        for (int p = wave; p < nld * ppc; p += 4) {
            const int col = p / ppc, q = p - col * ppc;
            const T *src = static_cast<const T *>(col < ku ? a.x[col] : (col == ku ? a.y : a.w));
            const int row0 = q * RPP + lane * VEC;
            const int64_t grow = base + c0 + row0;
            T *ldst = tl + (size_t)col * rs + q * RPP;
            if (row0 < rows_here) {
                if (grow + VEC <= a.n_rows) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + grow),
                                                     (__attribute__((address_space(3))) void *)ldst, 16, 0, 0);
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ldst[lane * VEC + v] = (grow + v < a.n_rows) ? src[grow + v] : T(0);
                }
            }
        }
    };
    if (nbuf == 2 && span > 0) stage(0, tile0);
    int buf = 0;
    for (int64_t c0 = 0; c0 < span; c0 += cr, buf ^= (nbuf - 1)) {
        const int rows_here = (int)min((int64_t)cr, span - c0);
        const int rows8 = (rows_here + 7) & ~7;
        T *const tile = tile0 + (size_t)buf * tile_elems;
        if (nbuf == 1) stage(c0, tile);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                            // this chunk has landed; every wave is done with the other buffer
        if (nbuf == 2 && c0 + cr < span) stage(c0 + cr, tile0 + (size_t)(buf ^ 1) * tile_elems);
        // ---- prep: zero rows outside the group, apply sqrt(w)
        const int64_t lo = head - c0, hi = span - c0;       // valid chunk rows: lo <= r < hi
        if constexpr (HAS_W) {
            T *wc = tile + (size_t)(ku + 1) * rs;
            for (int row0 = tid * VEC; row0 < rows8; row0 += 256 * VEC) {
                V wv;
                if (a.w) wv = *reinterpret_cast<V *>(wc + row0);
                T sw[VEC];
                bool ok[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const int r = row0 + v; ok[v] = (r >= lo && r < hi); sw[v] = a.w ? sqrt(vget<T>(wv, v)) : T(1); }
                if (pol != POLS_NULL_IGNORE) {      // which rows leave the fit (compute_is_valid_mask, ex.rs:201-228)
                    if (a.valid && null_checks_y(pol)) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) if (ok[v]) ok[v] = a.valid[base + c0 + row0 + v] != 0;
                    }
                    const int c_lo = null_checks_x(pol) ? 0 : ku, c_hi = null_checks_y(pol) ? ku : -1;
                    for (int c = c_lo; c <= c_hi; ++c) {
                        const V xv = *reinterpret_cast<V *>(tile + (size_t)c * rs + row0);
#pragma unroll
                        for (int v = 0; v < VEC; ++v) { const T x = vget<T>(xv, v); ok[v] = ok[v] && (x == x); }
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) { if (!ok[v]) sw[v] = T(0); else ++nfit; }
                if constexpr (VEC == 4) wv = V{sw[0], sw[1], sw[2], sw[3]}; else wv = V{sw[0], sw[1]};
                *reinterpret_cast<V *>(wc + row0) = wv;
                for (int c = 0; c <= ku; ++c) {
                    V xv = *reinterpret_cast<V *>(tile + (size_t)c * rs + row0);
                    T t[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) t[v] = ok[v] ? null_fill<T>(pol, vget<T>(xv, v)) * sw[v] : T(0);   // handle_nulls (ex.rs:257-296)
                    if constexpr (VEC == 4) xv = V{t[0], t[1], t[2], t[3]}; else xv = V{t[0], t[1]};
                    *reinterpret_cast<V *>(tile + (size_t)c * rs + row0) = xv;
                }
            }
            __syncthreads();
        } else {
            const int nhead = (c0 == 0) ? head : 0;
            const int ntail = rows8 - rows_here;
            const int npad = nhead + ntail;
            for (int i = tid; i < npad * ncols; i += 256) {
                const int c = i / npad, k = i - c * npad;
                const int r = (k < nhead) ? k : rows_here + (k - nhead);
                tile[(size_t)c * rs + r] = T(0);
            }
            if (npad > 0) __syncthreads();          // (workgroup-uniform: only a segment's first and last chunk have rows to blank)
        }
        // ---- MFMA: each wave takes a contiguous quarter of this chunk's 8-row steps
        const int nsteps = rows8 >> 3;
        const int per_wave = (nsteps + 3) >> 2;
        const int t_begin = min(nsteps, wave * per_wave), t_end = min(nsteps, t_begin + per_wave);
        const T *zp[NT];
        int zinc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (zsrc[t] >= 0) { zp[t] = tile + (size_t)zsrc[t] * rs + t_begin * 8 + rlane; zinc[t] = 8; }
            else { zp[t] = (zsrc[t] == -2) ? ones : zeros; zinc[t] = 0; }
        }
        const T *yp = tile + (size_t)ku * rs + t_begin * 8 + rlane;      // YV: the target rows matching this lane's operands
        for (int left = t_end - t_begin; left > 0;) {
        // (f32: at most eight steps = 64 rows per wave between two flushes into the f64 sums, whatever the chunk length)
        const int nb = (sizeof(T) == 4 && left > 8) ? 8 : left;
        left -= nb;
        for (int n = nb; n > 0; --n) {
            T v0[NT], v1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (sizeof(T) == 4) { const float2 vv = *reinterpret_cast<const float2 *>(zp[t]); v0[t] = vv.x; v1[t] = vv.y; }
                else { v0[t] = zp[t][0]; v1[t] = zp[t][4]; }
                zp[t] += zinc[t];
            }
            if constexpr (YV) {
                T y0, y1;
                if constexpr (sizeof(T) == 4) { const float2 yy = *reinterpret_cast<const float2 *>(yp); y0 = yy.x; y1 = yy.y; }
                else { y0 = yp[0]; y1 = yp[4]; }
                yp += 8;
                xy0 = fma(v0[0], y0, xy0);
                xy1 = fma(v1[0], y1, xy1);
            }
            int p = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = ti; tj < NT; ++tj, ++p) {
                    if (ti == 1 && !need11) continue;
                    acc[p][0] = M::mma(v0[ti], v0[tj], acc[p][0]);
                    acc[p][1] = M::mma(v1[ti], v1[tj], acc[p][1]);
                }
        }
        if constexpr (sizeof(T) == 4) {
            // f32: every 64 rows per wave the tile is flushed into f64 running sums, so the Gram matrix of an f32
            // frame carries f64-summation error -- what holds the f32 paths fed from here to the 1e-4 parity bound
#pragma unroll
            for (int p = 0; p < NPAIR; ++p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) accd[p][r] += (double)acc[p][0][r] + (double)acc[p][1][r];
                acc[p][0] = acc_t{0, 0, 0, 0}; acc[p][1] = acc_t{0, 0, 0, 0};
            }
            if constexpr (YV) { xyd += (double)xy0 + (double)xy1; xy0 = T(0); xy1 = T(0); }
        }
        }
        if (nbuf == 1) __syncthreads();             // the next chunk's DMA overwrites the tile
    }
    __syncthreads();                                // the partial tiles below re-use the first buffer
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) accd[p][r] = acc[p][0][r] + acc[p][1][r];
        if constexpr (YV) xyd = xy0 + xy1;
    }

    if (HAS_W && a.nvalid) {
        if (nfit) atomicAdd(&nfit_s, nfit);
        __syncthreads();
        if (tid == 0) a.nvalid[g] = (double)nfit_s;
    }
    // ---- cross-wave sum (fixed order) and write-out of the (symmetric) Gram matrix in f64
    double *part = reinterpret_cast<double *>(tile0);   // [pair][wave][reg * 64 + lane], f64
    if constexpr (YV) {                                     // X'y: fold the four row-quarters of the wave, then the waves
        double xy = xyd;
        xy += __shfl_xor(xy, 16);
        xy += __shfl_xor(xy, 32);
        if (lane < 16) part[NPAIR * 4 * 256 + wave * 16 + lane] = xy;
    }
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(p * 4 + wave) * 256 + r * 64 + lane] = accd[p][r];
    }
    __syncthreads();
    double *G = a.gram + (size_t)g * NZ * NZ;
    const int r = tid >> 6, l = tid & 63;
    const int drow = (sizeof(T) == 4) ? (l >> 4) * 4 + r : (l >> 4) + 4 * r;   // C/D layouts of the f32 / f64 16x16x4 MFMA
    const int dcol = l & 15;
    int p = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj, ++p) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[(p * 4 + w) * 256 + tid];
            const int i = 16 * ti + drow, j = 16 * tj + dcol;
            if (i < NZ && j < NZ) {
                if (!HAS_W && icpt && i == kt - 1 && j == kt - 1) v = (double)(e - s);   // the ones block also fed pad rows
                G[i * NZ + j] = v;
                if (ti != tj) G[j * NZ + i] = v;
            }
        }
    if constexpr (YV) {
        if (tid < 16) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[NPAIR * 4 * 256 + w * 16 + tid];
            G[tid * NZ + kt] = v;
            G[kt * NZ + tid] = v;
        }
        if (tid == 16) G[kt * NZ + kt] = 0.0;               // y'y is not computed on this path
    }
}

template <typename T, int NT, bool HAS_W, bool YV = false>
static int gram_stream_launch_t(pols_ctx *ctx, const GramArgs &a) {
    const int ncols = a.k_user + 1 + (HAS_W ? 1 : 0);
    // rows per LDS chunk: 256, doubled while a column's piece stays within 4 KB and the tile within 48 KB (three workgroups per CU).  The
    // 1 KB pieces of a 256-row f32 chunk streamed at 3.8 TB/s (2 KB, f64: 4.2) where 4 KB pieces of the same columns stream at 5.5
    // (profiles/r05_probe_matrix.txt, r05_kernel_stats_long_groups.csv)
    int cr = KG_CR;
    if (!ctx->opt.kg_single_buffer)
        while ((size_t)cr * 2 * sizeof(T) <= 4096 && (size_t)ncols * k1m_row_stride<T>(cr * 2) * sizeof(T) <= 48 * 1024) cr *= 2;
    const int rs = k1m_row_stride<T>(cr);
    const int npair = NT * (NT + 1) / 2;
    // the tile is re-used for the cross-wave partial tiles, which are f64 whatever T is
    const int tile_elems = std::max(ncols * rs, (int)((npair * 4 * 256 + (YV ? 64 : 0)) * (sizeof(double) / sizeof(T))));
    const int nbuf = (!ctx->opt.kg_single_buffer && sizeof(T) * (size_t)tile_elems <= 24 * 1024) ? 2 : 1;   // (wider tiles: two workgroups per CU do the same job)
    const size_t lds = sizeof(T) * ((size_t)nbuf * tile_elems + 2 * K1M_CONST_ELEMS);
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gram_stream_kernel<T, NT, HAS_W, YV>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_once.done(ctx->device);
    }
    char name[96];
    std::snprintf(name, sizeof(name), "k5_gram_stream_%s_nt%d%s%s_k%d", sizeof(T) == 4 ? "f32" : "f64", NT, HAS_W ? "_w" : "", YV ? "_yv" : "", a.kt);
    ctx->last_kernel = name;
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((gram_stream_kernel<T, NT, HAS_W, YV>), dim3((unsigned)a.n_groups), dim3(256), (unsigned)lds, ctx->stream, ev0, ev1, 0, a, rs, ncols, tile_elems, nbuf, cr);
    else
        hipLaunchKernelGGL((gram_stream_kernel<T, NT, HAS_W, YV>), dim3((unsigned)a.n_groups), dim3(256), lds, ctx->stream, a, rs, ncols, tile_elems, nbuf, cr);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

template <int NW>   // waves per workgroup
__global__ void __launch_bounds__(64 * NW) gram_reduce_kernel(const GramReduceArgs a) {
    // one workgroup per (group, 64 matrix entries, slice of the segment list): a lane per entry (coalesced), the waves take a share of
    // the slice's segments each, eight loads in flight per lane (a plain loop was one load latency per segment: 243 us for the 2 048
    // segments of a 10M-row group); the wave sums meet in LDS in wave order.  gridDim.z > 1 (ONE regression over a 10 M-row frame:
    // ~2 000 segments, and two workgroups to sum them -- 20 us of latency chains): the segment list is cut into gridDim.z slices whose
    // sums go to `slices`, and gram_reduce_final_kernel adds those in slice order.  The order of the additions depends on nothing but
    // the segment list and the launch shape, which the frame alone decides.
    __shared__ double part[NW][64];
    const int64_t g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v0 = a.first[g], v1 = a.first[g + 1];
    const int nz_ = (int)gridDim.z, z = (int)blockIdx.z;
    if (nz_ > 1) {
        const int per_z = (v1 - v0 + nz_ - 1) / nz_;
        v0 = v0 + z * per_z < v1 ? v0 + z * per_z : v1;
        v1 = v0 + per_z < v1 ? v0 + per_z : v1;
    }
    const int per = (v1 - v0 + NW - 1) / NW, va = v0 + wave * per < v1 ? v0 + wave * per : v1, vb = va + per < v1 ? va + per : v1;
    const int e = blockIdx.y * 64 + lane;
    double acc = 0.0;
    if (e < a.nz2) {
        const double *p = a.part + e;
        int v = va;
        for (; v + 8 <= vb; v += 8) {
            double t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = p[(size_t)(v + i) * a.nz2];
            acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; v < vb; ++v) acc += p[(size_t)v * a.nz2];
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < a.nz2) {
        double t = part[0][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) t += part[w][lane];          // wave order
        if (nz_ > 1) a.slices[((size_t)g * nz_ + z) * a.nz2 + e] = t;
        else a.gram[(size_t)g * a.nz2 + e] = t;
    }
    if (a.nvalid && blockIdx.y == 0 && z == 0 && wave == 0) {     // (counts: exact in any order; one thread's dependent loop was ~1 us per segment)
        double n = 0.0;
        for (int v = a.first[g] + lane; v < a.first[g + 1]; v += 64) n += a.nv_part[v];
        n = wave_sum_row3(n);
        if (lane == 63) a.nvalid[g] = n;
    }
}

__global__ void __launch_bounds__(64) gram_reduce_final_kernel(const GramReduceArgs a, const int n_slices) {
    const int64_t g = blockIdx.x;
    const int e = blockIdx.y * 64 + threadIdx.x;
    if (e >= a.nz2) return;
    const double *p = a.slices + (size_t)g * n_slices * a.nz2 + e;
    double t = p[0];
    for (int z = 1; z < n_slices; ++z) t += p[(size_t)z * a.nz2];   // slice order
    a.gram[(size_t)g * a.nz2 + e] = t;
}

int gram_reduce_launch(pols_ctx *ctx, const GramReduceArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    const unsigned eb = (unsigned)((a.nz2 + 63) / 64);
    if (a.slices && a.n_slices > 1) {
        hipLaunchKernelGGL(gram_reduce_kernel<4>, dim3((unsigned)a.n_groups, eb, (unsigned)a.n_slices), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(gram_reduce_final_kernel, dim3((unsigned)a.n_groups, eb), dim3(64), 0, ctx->stream, a, a.n_slices);
    } else if (a.max_segments >= 256) {
        // (one workgroup per (group, 64 entries): with ~2 000 segments in ONE group the four-wave form was 31 us of a 225-us call)
        hipLaunchKernelGGL(gram_reduce_kernel<16>, dim3((unsigned)a.n_groups, eb), dim3(1024), 0, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(gram_reduce_kernel<4>, dim3((unsigned)a.n_groups, eb), dim3(256), 0, ctx->stream, a);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

int gram_stream_launch(pols_ctx *ctx, int dtype, const GramArgs &a) {
    if (gram_valu_takes(ctx, dtype, a)) return gram_valu_launch(ctx, dtype, a);
    if (a.kt + 1 > 32) return fail(POLS_ERR_UNSUPPORTED, "gram_stream: %d features (incl. intercept) > 31", a.kt);
    const bool two = a.kt + 1 > 16, w = a.w != nullptr || a.null_policy != POLS_NULL_IGNORE;   // null policies ride on the sqrt(w) prep pass
    if (a.kt == 16 && !ctx->opt.kg_noyv) {   // the target would be alone in the second tile: X'y on the VALU
        if (dtype == POLS_F32) return w ? gram_stream_launch_t<float, 1, true, true>(ctx, a) : gram_stream_launch_t<float, 1, false, true>(ctx, a);
        return w ? gram_stream_launch_t<double, 1, true, true>(ctx, a) : gram_stream_launch_t<double, 1, false, true>(ctx, a);
    }
    if (dtype == POLS_F32) {
        if (two) return w ? gram_stream_launch_t<float, 2, true>(ctx, a) : gram_stream_launch_t<float, 2, false>(ctx, a);
        return w ? gram_stream_launch_t<float, 1, true>(ctx, a) : gram_stream_launch_t<float, 1, false>(ctx, a);
    }
    if (two) return w ? gram_stream_launch_t<double, 2, true>(ctx, a) : gram_stream_launch_t<double, 2, false>(ctx, a);
    return w ? gram_stream_launch_t<double, 1, true>(ctx, a) : gram_stream_launch_t<double, 1, false>(ctx, a);
}

// ------------------------------------------------------------------------------------------------ B: gram_cd
constexpr int CD_KMAX = 31;   // 16 lanes per group up to 16 columns, 32 lanes beyond

__device__ __forceinline__ double soft_threshold(double x, double thr, bool positive) {   // ls.rs:373-379
    const double mag = fmax(fabs(x) - thr, 0.0);
    double r = copysign(mag, x);
    if (positive) r = fmax(r, 0.0);
    return r;
}

template <typename T, int LPG>   // LPG lanes per group = the most columns it handles
__global__ void __launch_bounds__(64) gram_cd_kernel(const CdArgs a) {
    constexpr int CD_KMAX = LPG;
    const int lane = threadIdx.x, sub = lane & (LPG - 1);
    const int64_t grp = (int64_t)blockIdx.x * (64 / LPG) + (lane / LPG);
    const bool live = grp < a.n_groups;
    const int kt = a.kt, NZ = kt + 1;
    const int64_t gi = live ? grp : 0;
    const double *G = a.gram + (size_t)gi * NZ * NZ;
    const double n = a.nvalid ? a.nvalid[gi] : (double)(a.offs[gi + 1] - a.offs[gi]);

    double col[CD_KMAX], w[CD_KMAX], b[CD_KMAX], diag[CD_KMAX];
#pragma unroll
    for (int i = 0; i < CD_KMAX; ++i) {
        const bool in = i < kt;
        col[i] = (in && sub < kt) ? G[i * NZ + sub] : 0.0;     // column `sub` of X'X
        b[i] = in ? G[i * NZ + kt] : 0.0;                     // X'y
        diag[i] = in ? G[i * NZ + i] : 1.0;                   // xtx[[j, j]] (:431)
        w[i] = 0.0;                                           // w = zeros (:416)
    }
    const double alpha_n = a.alpha * n;                       // alpha * n_samples (:419)
    const double thr = alpha_n * a.l1_ratio, l2 = alpha_n * (1.0 - a.l1_ratio);
    const bool positive = a.positive != 0, active_set = a.active_set != 0;
    double wme = 0.0;                                         // w[sub], this lane's own coordinate
    unsigned mask = (kt >= 32) ? 0xffffffffu : ((1u << kt) - 1u);
    bool done = !live || n == 0.0;
    int status = (n == 0.0) ? POLS_GROUP_EMPTY : POLS_GROUP_NOT_CONVERGED;

    for (int64_t it = 0; it < a.max_iter; ++it) {
        if (__all(done)) break;
        double d2 = 0.0;
        const unsigned sweep = mask;                          // `for j in active_indices.clone()` (:459)
#pragma unroll
        for (int j = 0; j < CD_KMAX; ++j) {
            if (j >= kt) break;
            if (!((sweep >> j) & 1u)) continue;
            double sdot = row_allreduce(col[j] * wme);        // sum_i G[j][i] w[i] over the group's lanes
            if (LPG == 32) sdot += __shfl_xor(sdot, 16);
            const double dot = b[j] - sdot + diag[j] * w[j];  // x_j . (residuals + x_j w_j)  (:428-430)
            const double wn = soft_threshold(dot, thr, positive) / (diag[j] + l2);   // (:430-431)
            if (!done) {
                const double dw = wn - w[j];
                d2 += dw * dw;
                w[j] = wn;
                if (sub == j) wme = wn;
                if (active_set && fabs(wn) < a.tol) mask &= ~(1u << j);   // (:472-476)
            }
        }
        if (!done && sqrt(d2) < a.tol) { done = true; status = POLS_GROUP_OK; }    // (:436-444)
    }
    if (live) {
        if (sub < kt) {
            const double out = (n == 0.0) ? 0.0 : wme;
            if (a.coef) static_cast<T *>(a.coef)[grp * kt + sub] = (T)out;
            if (a.coef64) a.coef64[grp * kt + sub] = out;
        }
        if (sub == 0 && a.status) a.status[grp] = status;
    }
}

int gram_cd_launch(pols_ctx *ctx, int dtype, const CdArgs &a) {
    if (a.kt > CD_KMAX) return fail(POLS_ERR_UNSUPPORTED, "elastic net: %d features (incl. intercept) > %d", a.kt, CD_KMAX);
    if (a.kt <= 16) {
        const unsigned blocks = (unsigned)((a.n_groups + 3) / 4);
        if (dtype == POLS_F32) hipLaunchKernelGGL((gram_cd_kernel<float, 16>), dim3(blocks), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((gram_cd_kernel<double, 16>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    } else {
        const unsigned blocks = (unsigned)((a.n_groups + 1) / 2);
        if (dtype == POLS_F32) hipLaunchKernelGGL((gram_cd_kernel<float, 32>), dim3(blocks), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL((gram_cd_kernel<double, 32>), dim3(blocks), dim3(64), 0, ctx->stream, a);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ B': gram_solve
// OLS / ridge from a streamed Gram matrix for any kt <= 31 (groups too big for K1 / K1m, or 16..31 features):
// one wave per group, G + alpha I in LDS, lane i owns row i of the Cholesky factor (left-looking), then the two
// triangular solves column by column with a lane broadcast per step.  Same pivot test as chol_solve (ls.rs:289-299).
constexpr int GS_KMAX = 31;

template <typename T>
__global__ void __launch_bounds__(256) gram_solve_kernel(const CdArgs a) {
    __shared__ double Ls[4][GS_KMAX * GS_KMAX];
    __shared__ double rs[4][GS_KMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t grp = (int64_t)blockIdx.x * 4 + wv;
    if (grp >= a.n_groups) return;                     // wave-uniform; no block barriers below
    const int kt = a.kt, NZ = kt + 1;
    const double *G = a.gram + (size_t)grp * NZ * NZ;
    double *L = Ls[wv];
    double *rinv = rs[wv];
    const int64_t gid = a.glist ? (int64_t)a.glist[grp] : grp;   // (size classes: item grp of the launch is group gid of the frame)
    const int64_t n = a.nvalid ? (int64_t)a.nvalid[grp] : a.offs[gid + 1] - a.offs[gid];
    for (int q = lane; q < kt * kt; q += 64) {
        const int i = q / kt, j = q - i * kt;
        L[q] = G[i * NZ + j] + (i == j ? a.alpha : 0.0);
    }
    double bi = (lane < kt) ? G[lane * NZ + kt] : 0.0;          // X'y, one entry per lane
    __builtin_amdgcn_wave_barrier();
    bool ok = true;
    for (int j = 0; j < kt && a.solver != 1; ++j) {
        double d = L[j * kt + j];
        const double gjj = d;
        for (int p = 0; p < j; ++p) d -= L[j * kt + p] * L[j * kt + p];
        ok = ok && (d > a.pivot_tol * gjj);
        const double ri = 1.0 / sqrt(d);
        if (lane == 0) rinv[j] = ri;
        if (lane > j && lane < kt) {
            double sacc = L[lane * kt + j];
            for (int p = 0; p < j; ++p) sacc -= L[lane * kt + p] * L[j * kt + p];
            L[lane * kt + j] = sacc * ri;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // forward: t = L^-1 b   (lane i holds b_i / t_i)
    for (int p = 0; p < kt; ++p) {
        if (lane == p) bi *= rinv[p];
        const double tp = __shfl(bi, p);
        if (lane > p && lane < kt) bi -= L[lane * kt + p] * tp;
    }
    // backward: beta = L^-T t
    for (int p = kt - 1; p >= 0; --p) {
        if (lane == p) bi *= rinv[p];
        const double bp = __shfl(bi, p);
        if (lane < p) bi -= L[p * kt + lane] * bp;
    }
    // solve_method = "lu" (solve_ols_lu, ls.rs:264-273), or solve_ridge's Cholesky -> LU fallback (ls.rs:358-363): partial-pivot LU
    // of the same matrix, lane i owns row i (the factor's storage is re-initialised from the Gram matrix)
    if (a.solver == 1 || (!ok && a.lu_fallback)) {
        for (int q = lane; q < kt * kt; q += 64) {
            const int i = q / kt, j = q - i * kt;
            L[q] = G[i * NZ + j] + (i == j ? a.alpha : 0.0);
        }
        bi = (lane < kt) ? G[lane * NZ + kt] : 0.0;            // the right-hand side rides along in registers, permuted with the rows
        __builtin_amdgcn_wave_barrier();
        ok = true;
        for (int j = 0; j < kt; ++j) {
            double v = (lane >= j && lane < kt) ? fabs(L[lane * kt + j]) : -1.0;
            int idx = lane;
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const double ov = __shfl_xor(v, off);
                const int oi = __shfl_xor(idx, off);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            const int pr = __shfl(idx, 0);
            const double piv = L[pr * kt + j];
            ok = ok && (fabs(piv) > 0.0);
            const double bj = __shfl(bi, j), bp = __shfl(bi, pr);
            if (pr != j) {
                if (lane >= j && lane < kt) { const double t = L[j * kt + lane]; L[j * kt + lane] = L[pr * kt + lane]; L[pr * kt + lane] = t; }
                if (lane == j) bi = bp;
                if (lane == pr) bi = bj;
            }
            __builtin_amdgcn_wave_barrier();
            const double brow = __shfl(bi, j);
            if (lane > j && lane < kt) {
                const double f = L[lane * kt + j] / piv;
                for (int c = j + 1; c < kt; ++c) L[lane * kt + c] = fma(-f, L[j * kt + c], L[lane * kt + c]);
                bi = fma(-f, brow, bi);
            }
            __builtin_amdgcn_wave_barrier();
        }
        for (int p = kt - 1; p >= 0; --p) {
            if (lane == p) bi /= L[p * kt + p];
            const double bp = __shfl(bi, p);
            if (lane < p) bi = fma(-L[lane * kt + p], bp, bi);
        }
        ok = ok && __all(lane >= kt || ((bi == bi) && fabs(bi) <= 1.7e308));
    }
    int st = POLS_GROUP_OK;
    if (n == 0) { bi = 0.0; st = POLS_GROUP_EMPTY; }
    else if (!ok) { st = POLS_GROUP_FALLBACK; if (lane == 0 && a.fb_flag) *a.fb_flag = a.epoch; }
    if (lane < kt) {
        if (a.coef) static_cast<T *>(a.coef)[gid * kt + lane] = (T)bi;
        if (a.coef64) a.coef64[grp * kt + lane] = bi;
    }
    if (lane == 0 && a.status) a.status[gid] = st;
}

int gram_solve_launch(pols_ctx *ctx, int dtype, const CdArgs &a) {
    if (a.kt > GS_KMAX) return fail(POLS_ERR_UNSUPPORTED, "%d features (incl. intercept) > %d", a.kt, GS_KMAX);
    const unsigned blocks = (unsigned)((a.n_groups + 3) / 4);
    if (dtype == POLS_F32) hipLaunchKernelGGL(gram_solve_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(gram_solve_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, a);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// ------------------------------------------------------------------------------------------------ C: predict
// pred[r] = sum_j x_j[r] * c_j with c = per-group coefficients (f64) or per-row coefficients (dynamic models,
// src/expressions.rs:184); residuals = y - pred.  With sample weights the reference's arithmetic is kept:
// (sqrt_w x) . c * (1 / sqrt_w)  (least_squares.py:190-196, 234-235), so w == 0 gives NaN here too.
template <typename T, int JB>   // JB: columns loaded per batch (4 / 8 / 16: the smallest that covers the features keeps the registers, hence the waves per SIMD, for them)
__global__ void __launch_bounds__(256) predict_kernel(const PredictArgs a) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs_pairs ? a.offs[2 * g] : a.offs[g], e = a.offs_pairs ? a.offs[2 * g + 1] : a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int ku = a.k_user, kt = a.kt;
    const bool icpt = ku != kt;
    const double *cg = a.coef64 ? a.coef64 + (size_t)(a.gmap ? a.gmap[g] : g) * kt : nullptr;
    const T *crow = static_cast<const T *>(a.coef_rows);
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    const int pol = a.null_policy;
    // blockIdx.y: long items (segments of a split group, 10 000-row groups) are covered by several ONE-SHOT workgroups instead of one
    // workgroup looping -- every resident workgroup looping in step made the loads come in bursts (4.0 TB/s where K1's one-shot
    // workgroups stream 5.5)
    for (int64_t row0 = base + ((int64_t)blockIdx.y * 256 + threadIdx.x) * VEC; row0 < e; row0 += (int64_t)gridDim.y * 256 * VEC) {
        const bool full = (row0 >= s) && (row0 + VEC <= e);
        T p[VEC], sw[VEC], yv[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { p[v] = T(0); sw[v] = T(1); yv[v] = T(0); }
        if (a.w) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { const int64_t r = row0 + v; if (r >= s && r < e) sw[v] = sqrt(static_cast<const T *>(a.w)[r]); }
        }
        // columns in batches of JB: every load of a batch is issued before the first FMA, so a thread has JB 16-byte
        // requests in flight instead of one (the per-column loop was a chain of full memory latencies)
        for (int j0 = 0; j0 < kt; j0 += JB) {
            T xb[JB][VEC];
            T cb[JB];                               // this batch's group coefficients: wave-uniform loads, all issued before the first wait
            if (cg) {
#pragma unroll
                for (int u = 0; u < JB; ++u) cb[u] = (j0 + u < kt) ? (T)cg[j0 + u] : T(0);
            }
            if (full) {
                V tv[JB];
#pragma unroll
                for (int u = 0; u < JB; ++u)
                    if (j0 + u < ku) tv[u] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j0 + u]) + row0);
#pragma unroll
                for (int u = 0; u < JB; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) xb[u][v] = (j0 + u < ku) ? vget<T>(tv[u], v) : T(1);
            } else {
#pragma unroll
                for (int u = 0; u < JB; ++u)
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int64_t r = row0 + v;
                        xb[u][v] = (j0 + u < ku) ? ((r >= s && r < e) ? static_cast<const T *>(a.x[j0 + u])[r] : T(0)) : T(1);
                    }
            }
#pragma unroll
            for (int u = 0; u < JB; ++u) {
                const int j = j0 + u;
                if (j < kt) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const int64_t r = row0 + v;
                        T c;
                        if (cg) c = cb[u];
                        else c = (r >= s && r < e) ? crow[r * kt + j] : T(0);
                        p[v] = fma(null_fill<T>(pol, xb[u][v]) * sw[v], c, p[v]);
                    }
                }
            }
        }
        (void)icpt;
        if (a.w) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) p[v] *= T(1) / sw[v];
        }
        if (resid || pol == POLS_NULL_DROP) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { const int64_t r = row0 + v; if (r >= s && r < e) yv[v] = static_cast<const T *>(a.y)[r]; }
        }
        if (pol == POLS_NULL_DROP) {                // mask the rows that were not part of the fit (ex.rs:409-417)
            unsigned dropped = 0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t r = row0 + v;
                if (r >= s && r < e && !null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, r)) dropped |= 1u << v;
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) p[v] = nan_if<T>((dropped >> v) & 1u, p[v]);
        }
        if (full) {
            if (pred) { V o; if constexpr (VEC == 4) o = V{p[0], p[1], p[2], p[3]}; else o = V{p[0], p[1]}; store_stream(reinterpret_cast<V *>(pred + row0), o); }
            if (resid) {
                V o;
                if constexpr (VEC == 4) o = V{yv[0] - p[0], yv[1] - p[1], yv[2] - p[2], yv[3] - p[3]}; else o = V{yv[0] - p[0], yv[1] - p[1]};
                store_stream(reinterpret_cast<V *>(resid + row0), o);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t r = row0 + v;
                if (r >= s && r < e) { if (pred) pred[r] = p[v]; if (resid) resid[r] = yv[v] - p[v]; }
            }
        }
    }
}

// The same for the static models -- per-group coefficients, kt <= JB (predictions and / or residuals; sample weights and the "drop" mask as template forms) --
// without the general kernel's control flow: there, the column pointers and the coefficients were fetched one by one inside the divergent
// `full` branch, each behind its own s_waitcnt (eight dependent L2 round trips per workgroup before the first FMA: 4.06 TB/s on
// 8 f32 features where the same frame streams at 5.5 through K1).  Here every index is a compile-time constant: one s_load for the pointers,
// one for the coefficients, JB vector loads in flight, one wait.
// HAS_W: the reference's weighted arithmetic, (sqrt(w) x) . c * (1 / sqrt(w)) (see predict_kernel).  DROP: null_policy "drop" -- the rows that were not
// part of the fit get a NaN prediction (ex.rs:409-417), decided from the feature vectors already in registers + the target + the validity bytes.
template <typename T, int JB, bool HAS_W = false, bool DROP = false>
__global__ void __launch_bounds__(256) predict_groups_kernel(const PredictArgs a) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    const int64_t g = blockIdx.x;
    const int64_t s = a.offs_pairs ? a.offs[2 * g] : a.offs[g], e = a.offs_pairs ? a.offs[2 * g + 1] : a.offs[g + 1];
    const int64_t base = s - (s % VEC);
    const int ku = a.k_user, kt = a.kt;
    const double *cg = a.coef64 + (size_t)(a.gmap ? a.gmap[g] : g) * kt;
    T *pred = static_cast<T *>(a.pred);
    T *resid = static_cast<T *>(a.resid);
    const int pol = a.null_policy;
    T c[JB];
#pragma unroll
    for (int u = 0; u < JB; ++u) c[u] = (u < kt) ? (T)cg[u] : T(0);   // (zero beyond kt: the unrolled sums below run over all JB slots)
    for (int64_t row0 = base + ((int64_t)blockIdx.y * 256 + threadIdx.x) * VEC; row0 < e; row0 += (int64_t)gridDim.y * 256 * VEC) {
        T p[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) p[v] = T(0);
        if ((row0 >= s) && (row0 + VEC <= e)) {
            V tv[JB];
#pragma unroll
            for (int u = 0; u < JB; ++u)
                if (u < ku) tv[u] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[u]) + row0);
            T sw[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) sw[v] = T(1);
            if constexpr (HAS_W) {
                const V wv = *reinterpret_cast<const V *>(static_cast<const T *>(a.w) + row0);
#pragma unroll
                for (int v = 0; v < VEC; ++v) sw[v] = sqrt(vget<T>(wv, v));
            }
#pragma unroll
            for (int u = 0; u < JB; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const T xv = (u < ku) ? null_fill<T>(pol, vget<T>(tv[u], v)) : T(1);
                    p[v] = fma(HAS_W ? xv * sw[v] : xv, c[u], p[v]);
                }
            if constexpr (HAS_W) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) p[v] *= T(1) / sw[v];
            }
            V yv;
            if (DROP || resid) yv = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + row0);   // (workgroup-uniform)
            if constexpr (DROP) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const T y1 = vget<T>(yv, v);
                    bool fit = (y1 == y1) && !(a.valid && !a.valid[row0 + v]);
#pragma unroll
                    for (int u = 0; u < JB; ++u)
                        if (u < ku) { const T x1 = vget<T>(tv[u], v); fit = fit && (x1 == x1); }
                    p[v] = nan_if<T>(fit ? 0u : 1u, p[v]);
                }
            }
            if (pred) {
                V o;
                if constexpr (VEC == 4) o = V{p[0], p[1], p[2], p[3]}; else o = V{p[0], p[1]};
                store_stream(reinterpret_cast<V *>(pred + row0), o);
            }
            if (resid) {                                     // ORIGINAL target - prediction (least_squares.py:239)
                V o;
                if constexpr (VEC == 4) o = V{yv.x - p[0], yv.y - p[1], yv.z - p[2], yv.w - p[3]}; else o = V{yv.x - p[0], yv.y - p[1]};
                store_stream(reinterpret_cast<V *>(resid + row0), o);
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int64_t r = row0 + v;
                if (r >= s && r < e) {
                    const T sw = HAS_W ? sqrt(static_cast<const T *>(a.w)[r]) : T(1);
#pragma unroll
                    for (int u = 0; u < JB; ++u) {
                        const T xv = (u < ku) ? null_fill<T>(pol, static_cast<const T *>(a.x[u])[r]) : T(1);
                        p[v] = fma(HAS_W ? xv * sw : xv, c[u], p[v]);
                    }
                    if constexpr (HAS_W) p[v] *= T(1) / sw;
                    if constexpr (DROP) p[v] = nan_if<T>(null_row_in_fit<T>(pol, a.valid, a.y, a.x, ku, r) ? 0u : 1u, p[v]);
                    if (pred) pred[r] = p[v];
                    if (resid) resid[r] = static_cast<const T *>(a.y)[r] - p[v];
                }
            }
        }
    }
}

int predict_launch(pols_ctx *ctx, int dtype, const PredictArgs &a) {
    if (a.n_groups == 0) return POLS_OK;
    const int64_t per_wg = 256 * (dtype == POLS_F32 ? 4 : 2);
    int64_t gy = std::max<int64_t>(1, (a.max_item_rows + per_wg - 1) / per_wg);
    gy = std::min<int64_t>(std::min<int64_t>(gy, 1024), std::max<int64_t>(1, 16384 / a.n_groups));   // (short items of the same frame get gy - 1 empty workgroups each)
    if (ctx->opt.predict_loop) gy = 1;
    const dim3 grid((unsigned)a.n_groups, (unsigned)gy);
    const int jb = a.kt <= 4 ? 4 : (a.kt <= 8 ? 8 : (a.kt <= 12 ? 12 : 16));
    if (a.coef64 && (a.pred || a.resid) && ((a.null_policy != POLS_NULL_DROP && !a.resid) || a.y) && a.kt <= 16 && !ctx->opt.predict_loop) {
#define POLS_PREDICT_GROUPS_GO(T, W, D)                                                                                     \
    do {                                                                                                                    \
        if (jb == 4) hipLaunchKernelGGL((predict_groups_kernel<T, 4, W, D>), grid, dim3(256), 0, ctx->stream, a);           \
        else if (jb == 8) hipLaunchKernelGGL((predict_groups_kernel<T, 8, W, D>), grid, dim3(256), 0, ctx->stream, a);      \
        else if (jb == 12) hipLaunchKernelGGL((predict_groups_kernel<T, 12, W, D>), grid, dim3(256), 0, ctx->stream, a);    \
        else hipLaunchKernelGGL((predict_groups_kernel<T, 16, W, D>), grid, dim3(256), 0, ctx->stream, a);                  \
    } while (0)
#define POLS_PREDICT_GROUPS_T(T)                                                                                            \
    do {                                                                                                                    \
        if (a.null_policy == POLS_NULL_DROP) { if (a.w) POLS_PREDICT_GROUPS_GO(T, true, true); else POLS_PREDICT_GROUPS_GO(T, false, true); } \
        else { if (a.w) POLS_PREDICT_GROUPS_GO(T, true, false); else POLS_PREDICT_GROUPS_GO(T, false, false); }            \
    } while (0)
        if (dtype == POLS_F32) POLS_PREDICT_GROUPS_T(float);
        else POLS_PREDICT_GROUPS_T(double);
#undef POLS_PREDICT_GROUPS_T
#undef POLS_PREDICT_GROUPS_GO
        POLS_HIP(hipGetLastError());
        return POLS_OK;
    }
#define POLS_PREDICT_GO(T)                                                                                          \
    do {                                                                                                            \
        if (jb == 4) hipLaunchKernelGGL((predict_kernel<T, 4>), grid, dim3(256), 0, ctx->stream, a);                \
        else if (jb <= 8) hipLaunchKernelGGL((predict_kernel<T, 8>), grid, dim3(256), 0, ctx->stream, a);           \
        else hipLaunchKernelGGL((predict_kernel<T, 16>), grid, dim3(256), 0, ctx->stream, a);                       \
    } while (0)
    if (dtype == POLS_F32) POLS_PREDICT_GO(float);
    else POLS_PREDICT_GO(double);
#undef POLS_PREDICT_GO
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

}  // namespace pols
