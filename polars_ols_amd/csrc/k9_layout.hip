// k9_layout.hip -- K9 "group_layout": `.over(key)` / `group_by(key)` ingestion on the device.
//
// The reference never sees a key column: Polars partitions the frame on the host and calls the plugin once per group
// (README.md:19, :57 `.over("group")`; tests/test_ols.py:110,384,860).  The batched engine wants the opposite -- every
// group in one launch, rows sorted by group (pols_batch.group_offsets) -- so the partitioning moves here:
//   key_probe      one sweep: min / max key and "is the column already non-decreasing" (then nothing moves at all);
//   key_rebase     key - min as an unsigned radix key of only as many bits as the key RANGE needs (10 000 dense group ids
//                  = 14 bits = two 8-bit radix passes instead of eight) + the row iota;
//   rocprim        stable LSD radix_sort_pairs (the device-wide sort is the library's, like a plain GEMM would be hipBLASLt's;
//                  stability keeps each group's rows in frame order, which RLS / rolling depend on), run_length_encode for
//                  the group keys / sizes, exclusive_scan for the offsets;
//   take / untake  frame order <-> group order for up to 32 columns per launch, a column at a time, always WALKING THE FRAME in
//                  XCD-sized slabs (see slab_block): the group-sorted side is then touched at one slowly advancing front per
//                  group, which an XCD's L2 turns into whole-line traffic -- a scatter on the way in, a gather through the
//                  inverse permutation on the way back; with more groups than an L2 holds fronts for, a plain gather;
//   row_groups     the group id of every frame row (what broadcasts a per-group coefficient struct back over the frame).
// All of it is HBM-bound integer / byte work: no LDS tricks, just coalesced streams and as few passes as the key range allows.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

#include <cstdlib>
#include <cstring>
#include <vector>

struct pols_layout {
    int device = 0;
    int64_t n = 0, n_groups = 0;
    bool identity = true;                 // keys were already non-decreasing: order == iota, nothing to move
    uint32_t *order = nullptr;            // device [n]: sorted position i holds frame row order[i]   (NULL when identity)
    uint32_t *inverse = nullptr;          // device [n]: frame row j sits at sorted position inverse[j] (built on first use)
    int64_t *d_offsets = nullptr;         // device [n_groups + 1]
    // one device allocation for order | inverse | offsets (hipMalloc / hipFree are ~100 us each): the offsets only move out of it
    // when there are more groups than the reserve foresaw
    void *arena = nullptr;
    uint32_t *inverse_slot = nullptr;
    bool offsets_own = false;
    std::vector<int64_t> offsets, keys;   // host copies: group_offsets for pols_batch, one key per group
};

namespace pols {

struct KeyProbe { long long mn, mx; int unsorted; int pad; };

// One partial per workgroup, reduced on the host after the (anyway needed) copy back: 2 048 same-address 64-bit atomics made this
// sweep 154 us on 10M keys.  Two keys per 16-byte load, the right-hand neighbour for the order test re-read from L1.
constexpr int PROBE_BLOCKS_PER_CU = 8;
__global__ void __launch_bounds__(256) key_probe_kernel(const int64_t *__restrict__ keys, int64_t n, KeyProbe *__restrict__ out) {
    __shared__ long long smn[4], smx[4];
    __shared__ int sun[4];
    long long mn = 0x7fffffffffffffffLL, mx = -0x7fffffffffffffffLL - 1;
    int uns = 0;
    const int64_t stride = (int64_t)gridDim.x * 512;
    const bool al = (reinterpret_cast<uintptr_t>(keys) & 15) == 0;
#pragma unroll 4
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += stride) {
        long long k0, k1, k2;
        if (al && i + 2 < n) {
            const longlong2 kk = *reinterpret_cast<const longlong2 *>(keys + i);
            k0 = kk.x; k1 = kk.y; k2 = keys[i + 2];
        } else {
            k0 = keys[i];
            k1 = i + 1 < n ? keys[i + 1] : k0;
            k2 = i + 2 < n ? keys[i + 2] : k1;
        }
        const long long lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
        mn = lo < mn ? lo : mn;
        mx = hi > mx ? hi : mx;
        uns |= (k1 < k0) | (k2 < k1);
    }
    for (int off = 32; off; off >>= 1) {
        const long long a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
        uns |= __shfl_xor(uns, off);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { smn[wv] = mn; smx[wv] = mx; sun[wv] = uns; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; uns |= sun[w]; }
        KeyProbe r = {mn, mx, uns, 0};
        out[blockIdx.x] = r;
    }
}

template <typename U>
__global__ void __launch_bounds__(256) key_rebase_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t mn,
                                                         U *__restrict__ u, uint32_t *__restrict__ iota) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u[i] = (U)((uint64_t)keys[i] - (uint64_t)mn);
    iota[i] = (uint32_t)i;
}

template <typename U>
__global__ void __launch_bounds__(256) key_restore_kernel(const U *__restrict__ u, int64_t n, int64_t mn, int64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = (int64_t)((uint64_t)u[i] + (uint64_t)mn);
}

__global__ void __launch_bounds__(256) invert_kernel(const uint32_t *__restrict__ order, int64_t n, uint32_t *__restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[order[i]] = (uint32_t)i;
}

constexpr int TAKE_COLS = 32;
struct TakeArgs {
    const void *src[TAKE_COLS];
    void *dst[TAKE_COLS];
    const uint32_t *index;               // dst[c][i] = src[c][index[i]]
    int64_t n;
    int n_cols;
};

// The hardware deals workgroups round-robin over the 8 XCDs, each with its own 4 MB L2.  A kernel that walks the frame in row
// order touches, per group, one slowly advancing "front" of the group-sorted column (a group's rows keep their frame order),
// i.e. n_groups cache lines at a time -- L2-sized for thousands of groups, but only if a line's 16-32 visits come from ONE XCD.
// So the frame is cut into 8 contiguous slabs, one per XCD: logical block = (b % 8) * (blocks / 8) + b / 8  (gridDim.x is a
// multiple of 8; logical blocks past the end return).
__device__ __forceinline__ int64_t slab_block(bool slab) {
    if (!slab) return blockIdx.x;
    const unsigned per = gridDim.x >> 3;
    return (int64_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

// blockIdx.y = column, one thread = VEC consecutive output rows of it: dst[i] = src[index[i]].  Workgroups are dispatched
// x-fastest, so a launch walks the frame one column at a time (a 10M-row f32 column is 40 MB and stays in the 256 MB Infinity
// Cache while it is gathered); the index is re-read per column (4 coalesced bytes per row); the VEC gathers of a thread are
// all issued before its one VEC-wide non-temporal store.  SLAB: the walk is in frame order (untake through the inverse
// permutation) -- see slab_block().
template <typename E, int VEC, bool SLAB>
__global__ void __launch_bounds__(256) take_kernel(const TakeArgs a) {
    const int64_t i0 = (slab_block(SLAB) * 256 + threadIdx.x) * VEC;
    if (i0 >= a.n) return;
    const E *s = static_cast<const E *>(a.src[blockIdx.y]);
    E *d = static_cast<E *>(a.dst[blockIdx.y]) + i0;
    typedef E VecE __attribute__((ext_vector_type(VEC)));
    typedef uint32_t VecI __attribute__((ext_vector_type(VEC)));
    if (i0 + VEC <= a.n) {
        const VecI idx = *reinterpret_cast<const VecI *>(a.index + i0);      // hipMalloc'ed, i0 a multiple of VEC: aligned
        VecE vv;
#pragma unroll
        for (int v = 0; v < VEC; ++v) vv[v] = s[idx[v]];
        if ((reinterpret_cast<uintptr_t>(d) & (sizeof(VecE) - 1)) == 0) {
            __builtin_nontemporal_store(vv, reinterpret_cast<VecE *>(d));
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) d[v] = vv[v];
        }
    } else {
        for (int v = 0; i0 + v < a.n; ++v) d[v] = s[a.index[i0 + v]];
    }
}

// The same permutation as a scatter that walks the FRAME: dst[index[j]] = src[j] with index = the inverse permutation.  Reads are
// perfectly coalesced; the writes land on the groups' fronts and are merged into whole lines by the XCD's L2 before they leave it
// (slab_block).  Gathering in group order instead reads every row of a group from a different line of the frame column with no
// reuse inside any L2: 128 bytes over the fabric per 4-byte element (measured 0.17 ms per 10M-row f32 column against 0.0x ms here).
// Only worth it while the fronts fit an L2 (move_columns decides).
template <typename E, int VEC>
__global__ void __launch_bounds__(256) put_kernel(const TakeArgs a) {
    const int64_t j0 = (slab_block(true) * 256 + threadIdx.x) * VEC;
    if (j0 >= a.n) return;
    const E *s = static_cast<const E *>(a.src[blockIdx.y]) + j0;
    E *d = static_cast<E *>(a.dst[blockIdx.y]);
    typedef E VecE __attribute__((ext_vector_type(VEC)));
    typedef uint32_t VecI __attribute__((ext_vector_type(VEC)));
    if (j0 + VEC <= a.n) {
        const VecI idx = *reinterpret_cast<const VecI *>(a.index + j0);
        VecE vv;
        if ((reinterpret_cast<uintptr_t>(s) & (sizeof(VecE) - 1)) == 0) {
            vv = __builtin_nontemporal_load(reinterpret_cast<const VecE *>(s));
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) vv[v] = s[v];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) d[idx[v]] = vv[v];
    } else {
        for (int v = 0; j0 + v < a.n; ++v) d[a.index[j0 + v]] = s[v];
    }
}

// Rows of `words` 4-byte words (a per-row coefficient table [n, k]): one thread per word, coalesced along the row both ways.
__global__ void __launch_bounds__(256) take_rows_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                        const uint32_t *__restrict__ index, int64_t n, int words) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * words) return;
    const int64_t i = t / words;
    const int w = (int)(t - i * words);
    __builtin_nontemporal_store(src[(int64_t)index[i] * words + w], dst + t);
}

// out[order[i]] = the group holding sorted position i.  Walking the SORTED side keeps the binary search over the offsets coherent
// (a wave's 64 positions sit in one or two groups, so its log2(G) probes are broadcasts); the 8-byte writes scatter over the
// frame.  Walking the frame instead (coalesced writes, divergent probes) measured 0.185 ms against 0.135 ms on 10M rows.
__global__ void __launch_bounds__(256) row_groups_kernel(const int64_t *__restrict__ offs, int64_t n_groups, const uint32_t *__restrict__ order,
                                                         int64_t n, int64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t lo = 0, hi = n_groups;       // largest g with offs[g] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    out[order ? order[i] : i] = lo;
}

static inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

template <typename U>
static int sort_and_segment(pols_ctx *ctx, pols_layout *L, const int64_t *d_keys, int64_t mn, int bits, bool sorted) {
    const int64_t n = L->n;
    const size_t nb = round256(sizeof(U) * (size_t)n), ib = round256(sizeof(uint32_t) * (size_t)n);
    // slot 9: [u_in | u_out | iota | unique keys (U) | counts (u32) | run count | rocprim temp]
    size_t t_sort = 0, t_rle = 0, t_scan = 0;
    U *np_u = nullptr;
    uint32_t *np_i = nullptr;
    int64_t *np_o = nullptr;
    POLS_HIP((rocprim::radix_sort_pairs(nullptr, t_sort, (const U *)np_u, np_u, (const uint32_t *)np_i, np_i, (size_t)n, 0u, (unsigned)bits,
                                         ctx->stream)));
    POLS_HIP((rocprim::run_length_encode(nullptr, t_rle, (const U *)np_u, (unsigned int)n, np_u, np_i, np_i, ctx->stream)));
    POLS_HIP((rocprim::exclusive_scan(nullptr, t_scan, (const uint32_t *)np_i, np_o, (int64_t)0, (size_t)n + 1, rocprim::plus<int64_t>(), ctx->stream)));
    // (the tail of the region also carries the unique keys home as int64: up to n of them)
    const size_t tmpb = round256(std::max(std::max(t_sort, sizeof(int64_t) * (size_t)n), std::max(t_rle, t_scan)));
    void *base = nullptr;
    int rc = ensure_scratch(ctx, 9, 3 * nb + 2 * ib + 256 + tmpb, &base);
    if (rc) return rc;
    char *p = static_cast<char *>(base);
    U *u_in = reinterpret_cast<U *>(p);            p += nb;
    U *u_out = reinterpret_cast<U *>(p);           p += nb;
    U *uniq = reinterpret_cast<U *>(p);            p += nb;
    uint32_t *iota = reinterpret_cast<uint32_t *>(p); p += ib;
    uint32_t *counts = reinterpret_cast<uint32_t *>(p); p += ib;
    uint32_t *n_runs = reinterpret_cast<uint32_t *>(p); p += 256;
    void *tmp = p;

    hipLaunchKernelGGL(key_rebase_kernel<U>, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_keys, n, mn, u_in, iota);
    const U *sorted_keys = u_in;
    if (!sorted) {
        POLS_HIP(hipMalloc(&L->arena, 2 * ib + round256(sizeof(int64_t) * (size_t)(std::min<int64_t>(n, (int64_t)1 << 20) + 1))));
        L->order = static_cast<uint32_t *>(L->arena);
        L->inverse_slot = reinterpret_cast<uint32_t *>(static_cast<char *>(L->arena) + ib);
        size_t t = tmpb;
        POLS_HIP((rocprim::radix_sort_pairs(tmp, t, (const U *)u_in, u_out, (const uint32_t *)iota, L->order, (size_t)n, 0u,
                                             (unsigned)bits, ctx->stream)));
        sorted_keys = u_out;
        L->identity = false;
    }
    {
        size_t t = tmpb;
        POLS_HIP((rocprim::run_length_encode(tmp, t, sorted_keys, (unsigned int)n, uniq, counts, n_runs, ctx->stream)));
    }
    uint32_t runs = 0;
    POLS_HIP(hipMemcpyAsync(&runs, n_runs, sizeof(runs), hipMemcpyDeviceToHost, ctx->stream));
    POLS_HIP(hipStreamSynchronize(ctx->stream));
    L->n_groups = runs;
    if (L->arena && (int64_t)runs <= ((int64_t)1 << 20)) {
        L->d_offsets = reinterpret_cast<int64_t *>(static_cast<char *>(L->arena) + 2 * ib);
    } else {
        POLS_HIP(hipMalloc(&L->d_offsets, sizeof(int64_t) * ((size_t)runs + 1)));
        L->offsets_own = true;
    }
    {
        // counts[runs] is never read as a group size: the scan of runs + 1 inputs only needs its first `runs` values to
        // produce offsets[0 .. runs]; the slot exists (counts has n >= runs entries or the 256-byte pad behind it).
        size_t t = tmpb;
        POLS_HIP((rocprim::exclusive_scan(tmp, t, (const uint32_t *)counts, L->d_offsets, (int64_t)0, (size_t)runs + 1,
                                           rocprim::plus<int64_t>(), ctx->stream)));
    }
    L->offsets.resize((size_t)runs + 1);
    L->keys.resize(runs);
    POLS_HIP(hipMemcpyAsync(L->offsets.data(), L->d_offsets, sizeof(int64_t) * ((size_t)runs + 1), hipMemcpyDeviceToHost, ctx->stream));
    // unique keys back to int64 in place of u_in (no longer needed), then home
    int64_t *k64 = reinterpret_cast<int64_t *>(tmp);
    if (runs) {
        hipLaunchKernelGGL(key_restore_kernel<U>, dim3(blocks_for(runs, 256)), dim3(256), 0, ctx->stream, (const U *)uniq, (int64_t)runs, mn, k64);
        POLS_HIP(hipMemcpyAsync(L->keys.data(), k64, sizeof(int64_t) * (size_t)runs, hipMemcpyDeviceToHost, ctx->stream));
    }
    POLS_HIP(hipStreamSynchronize(ctx->stream));
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

static int ensure_inverse(pols_ctx *ctx, pols_layout *L) {
    if (L->identity || L->inverse) return POLS_OK;
    L->inverse = L->inverse_slot;
    hipLaunchKernelGGL(invert_kernel, dim3(blocks_for(L->n, 256)), dim3(256), 0, ctx->stream, (const uint32_t *)L->order, L->n, L->inverse);
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

static int move_columns(pols_ctx *ctx, pols_layout *L, int dtype_bytes, const void *const *src, void *const *dst, int n_cols, int mem,
                        bool to_group_order) {
    if (!L) return fail(POLS_ERR_INVALID, "layout is NULL");
    if (L->device != ctx->device) return fail(POLS_ERR_INVALID, "layout belongs to device %d, context to %d", L->device, ctx->device);
    if (n_cols < 0 || (n_cols && (!src || !dst))) return fail(POLS_ERR_INVALID, "src / dst column tables are NULL");
    if (dtype_bytes != 1 && (dtype_bytes < 4 || (dtype_bytes & 3)))
        return fail(POLS_ERR_INVALID, "element size must be 1 byte or a multiple of 4 bytes (got %d)", dtype_bytes);
    const int64_t n = L->n;
    const size_t colb = (size_t)dtype_bytes * (size_t)n;
    for (int c = 0; c < n_cols; ++c)
        if (!src[c] || !dst[c]) return fail(POLS_ERR_INVALID, "column %d is NULL", c);
    if (n == 0 || n_cols == 0) return POLS_OK;
    const hipMemcpyKind kind = mem == POLS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (L->identity) {                                       // nothing moves: plain copies
        for (int c = 0; c < n_cols; ++c)
            if (src[c] != dst[c])
                POLS_HIP(hipMemcpyAsync(dst[c], src[c], colb, mem == POLS_MEM_HOST ? hipMemcpyHostToHost : kind, ctx->stream));
        if (mem == POLS_MEM_HOST) POLS_HIP(hipStreamSynchronize(ctx->stream));
        return POLS_OK;
    }
    // frame -> group order: scatter along the frame while the groups' fronts (one line per group, two while a front crosses a line)
    // fit an XCD's 4 MB L2 next to the streams passing through it; gather in group order otherwise.  POLS_K9_TAKE=gather|scatter
    // overrides (profiling).
    bool scatter = to_group_order && dtype_bytes <= 8 && L->n_groups <= 12288;
    if (ctx->opt.k9_take == 1) scatter = false;
    if (ctx->opt.k9_take == 2) scatter = to_group_order && dtype_bytes <= 8;
    int rc = (to_group_order && !scatter) ? POLS_OK : ensure_inverse(ctx, L);
    if (rc) return rc;
    const uint32_t *index = (to_group_order && !scatter) ? L->order : L->inverse;
    const bool slab = !to_group_order || scatter;
    const unsigned nblk = slab ? ((blocks_for(n, 1024) + 7u) & ~7u) : blocks_for(n, 1024);
    for (int c0 = 0; c0 < n_cols; c0 += TAKE_COLS) {
        const int nc = std::min(TAKE_COLS, n_cols - c0);
        TakeArgs a;
        std::memset(&a, 0, sizeof(a));
        a.index = index;
        a.n = n;
        a.n_cols = nc;
        char *stage = nullptr;
        if (mem == POLS_MEM_HOST) {                          // PCIe-inclusive convenience path: stage in, move, stage out
            void *s = nullptr;
            if ((rc = ensure_scratch(ctx, 1, 2 * round256(colb) * (size_t)nc, &s))) return rc;
            stage = static_cast<char *>(s);
            for (int c = 0; c < nc; ++c) {
                POLS_HIP(hipMemcpyAsync(stage + round256(colb) * c, src[c0 + c], colb, hipMemcpyHostToDevice, ctx->stream));
                a.src[c] = stage + round256(colb) * c;
                a.dst[c] = stage + round256(colb) * (nc + c);
            }
        } else {
            for (int c = 0; c < nc; ++c) {
                if (src[c0 + c] == dst[c0 + c]) return fail(POLS_ERR_INVALID, "column %d: a permutation cannot run in place", c0 + c);
                a.src[c] = src[c0 + c];
                a.dst[c] = dst[c0 + c];
            }
        }
        const dim3 grid(nblk, nc);
        if (scatter) {
            if (dtype_bytes == 4) hipLaunchKernelGGL((put_kernel<uint32_t, 4>), grid, dim3(256), 0, ctx->stream, a);
            else if (dtype_bytes == 8) hipLaunchKernelGGL((put_kernel<uint64_t, 4>), grid, dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL((put_kernel<uint8_t, 4>), grid, dim3(256), 0, ctx->stream, a);
        } else if (dtype_bytes == 4) {
            if (slab) hipLaunchKernelGGL((take_kernel<uint32_t, 4, true>), grid, dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL((take_kernel<uint32_t, 4, false>), grid, dim3(256), 0, ctx->stream, a);
        } else if (dtype_bytes == 8) {
            if (slab) hipLaunchKernelGGL((take_kernel<uint64_t, 4, true>), grid, dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL((take_kernel<uint64_t, 4, false>), grid, dim3(256), 0, ctx->stream, a);
        } else if (dtype_bytes == 1) {
            if (slab) hipLaunchKernelGGL((take_kernel<uint8_t, 4, true>), grid, dim3(256), 0, ctx->stream, a);
            else hipLaunchKernelGGL((take_kernel<uint8_t, 4, false>), grid, dim3(256), 0, ctx->stream, a);
        } else {
            const int words = dtype_bytes / 4;
            if (n * words > (int64_t)0x7fffffff * 256) return fail(POLS_ERR_UNSUPPORTED, "group_layout: table too large for one launch");
            for (int c = 0; c < nc; ++c)
                hipLaunchKernelGGL(take_rows_kernel, dim3(blocks_for(n * words, 256)), dim3(256), 0, ctx->stream,
                                   static_cast<const uint32_t *>(a.src[c]), static_cast<uint32_t *>(a.dst[c]), index, n, words);
        }
        POLS_HIP(hipGetLastError());
        if (mem == POLS_MEM_HOST) {
            for (int c = 0; c < nc; ++c)
                POLS_HIP(hipMemcpyAsync(dst[c0 + c], stage + round256(colb) * (nc + c), colb, hipMemcpyDeviceToHost, ctx->stream));
            POLS_HIP(hipStreamSynchronize(ctx->stream));
        }
    }
    return POLS_OK;
}

}  // namespace pols

using namespace pols;

extern "C" {

int pols_layout_create(pols_ctx *ctx, const int64_t *keys, int64_t n_rows, int mem, pols_layout **out) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    POLS_HIP(hipSetDevice(ctx->device));
    if (!out) return fail(POLS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rows < 0 || (n_rows && !keys)) return fail(POLS_ERR_INVALID, "keys is NULL / n_rows < 0");
    if (n_rows >= ((int64_t)1 << 32)) return fail(POLS_ERR_UNSUPPORTED, "group_layout: %lld rows (row indices are 32-bit)", (long long)n_rows);
    if (mem != POLS_MEM_HOST && mem != POLS_MEM_DEVICE) return fail(POLS_ERR_INVALID, "mem must be POLS_MEM_HOST or POLS_MEM_DEVICE");
    pols_layout *L = new pols_layout;
    L->device = ctx->device;
    L->n = n_rows;
    auto bail = [&](int rc) { pols_layout_destroy(L); return rc; };
    if (n_rows == 0) {
        L->offsets.assign(1, 0);
        *out = L;
        return POLS_OK;
    }
    const int64_t *d_keys = keys;
    int rc;
    if (mem == POLS_MEM_HOST) {
        void *d = nullptr;
        if ((rc = ensure_scratch(ctx, 1, sizeof(int64_t) * (size_t)n_rows, &d))) return bail(rc);
        if (hipMemcpyAsync(d, keys, sizeof(int64_t) * (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
            return bail(fail(POLS_ERR_HIP, "group_layout: key upload failed"));
        d_keys = static_cast<const int64_t *>(d);
    }
    const unsigned nblk = std::min<unsigned>(blocks_for(n_rows, 512), (unsigned)std::max(ctx->num_cus, 1) * PROBE_BLOCKS_PER_CU);
    void *pr = nullptr;
    if ((rc = ensure_scratch(ctx, 6, sizeof(KeyProbe) * nblk, &pr))) return bail(rc);
    std::vector<KeyProbe> part(nblk);
    hipLaunchKernelGGL(key_probe_kernel, dim3(nblk), dim3(256), 0, ctx->stream, d_keys, n_rows, static_cast<KeyProbe *>(pr));
    if (hipMemcpyAsync(part.data(), pr, sizeof(KeyProbe) * nblk, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return bail(fail(POLS_ERR_HIP, "group_layout: key probe failed: %s", hipGetErrorString(hipGetLastError())));
    KeyProbe h = part[0];
    for (unsigned b = 1; b < nblk; ++b) {
        h.mn = std::min(h.mn, part[b].mn);
        h.mx = std::max(h.mx, part[b].mx);
        h.unsorted |= part[b].unsorted;
    }
    const uint64_t range = (uint64_t)h.mx - (uint64_t)h.mn;
    int bits = 1;
    while (bits < 64 && (range >> bits)) ++bits;
    rc = bits <= 32 ? sort_and_segment<uint32_t>(ctx, L, d_keys, h.mn, bits, !h.unsorted)
                    : sort_and_segment<uint64_t>(ctx, L, d_keys, h.mn, bits, !h.unsorted);
    if (rc) return bail(rc);
    ctx->last_kernel = h.unsorted ? (bits <= 32 ? "k9_group_layout_sort_u32" : "k9_group_layout_sort_u64") : "k9_group_layout_presorted";
    *out = L;
    return POLS_OK;
}

void pols_layout_destroy(pols_layout *L) {
    if (!L) return;
    (void)hipSetDevice(L->device);
    if (L->arena) (void)hipFree(L->arena);
    if (L->d_offsets && L->offsets_own) (void)hipFree(L->d_offsets);
    delete L;
}

int64_t pols_layout_n_rows(const pols_layout *L) { return L ? L->n : -1; }
int64_t pols_layout_n_groups(const pols_layout *L) { return L ? L->n_groups : -1; }
int pols_layout_is_identity(const pols_layout *L) { return L ? (L->identity ? 1 : 0) : -1; }
const int64_t *pols_layout_group_offsets(const pols_layout *L) { return L ? L->offsets.data() : nullptr; }
const int64_t *pols_layout_group_keys(const pols_layout *L) { return L ? L->keys.data() : nullptr; }

int pols_layout_take(pols_ctx *ctx, pols_layout *L, int element_bytes, const void *const *src_cols, void *const *dst_cols, int32_t n_cols, int mem) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    POLS_HIP(hipSetDevice(ctx->device));
    return move_columns(ctx, L, element_bytes, src_cols, dst_cols, n_cols, mem, true);
}

int pols_layout_untake(pols_ctx *ctx, pols_layout *L, int element_bytes, const void *const *src_cols, void *const *dst_cols, int32_t n_cols, int mem) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    POLS_HIP(hipSetDevice(ctx->device));
    return move_columns(ctx, L, element_bytes, src_cols, dst_cols, n_cols, mem, false);
}

int pols_layout_row_groups(pols_ctx *ctx, pols_layout *L, int64_t *out, int mem) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    POLS_HIP(hipSetDevice(ctx->device));
    if (!L || !out) return fail(POLS_ERR_INVALID, "layout / out is NULL");
    if (L->device != ctx->device) return fail(POLS_ERR_INVALID, "layout belongs to device %d, context to %d", L->device, ctx->device);
    if (L->n == 0) return POLS_OK;
    int64_t *d_out = out;
    if (mem == POLS_MEM_HOST) {
        void *d = nullptr;
        int rc = ensure_scratch(ctx, 1, sizeof(int64_t) * (size_t)L->n, &d);
        if (rc) return rc;
        d_out = static_cast<int64_t *>(d);
    }
    hipLaunchKernelGGL(row_groups_kernel, dim3(blocks_for(L->n, 256)), dim3(256), 0, ctx->stream, (const int64_t *)L->d_offsets, L->n_groups,
                       (const uint32_t *)L->order, L->n, d_out);
    POLS_HIP(hipGetLastError());
    if (mem == POLS_MEM_HOST) {
        POLS_HIP(hipMemcpyAsync(out, d_out, sizeof(int64_t) * (size_t)L->n, hipMemcpyDeviceToHost, ctx->stream));
        POLS_HIP(hipStreamSynchronize(ctx->stream));
    }
    return POLS_OK;
}

}  // extern "C"
