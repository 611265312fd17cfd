// common.hpp -- context, error plumbing and wave-level primitives shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pols_mi355x.h"

namespace pols {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define POLS_HIP(call)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ::pols::fail(POLS_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                         \
    } while (0)

// ---------------------------------------------------------------- context
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct TimedLaunch {
    hipEvent_t start, stop;
};

// Tuning / diagnostic knobs.  Parsed ONCE from the POLS_* environment variables in pols_create() (a launch path never calls
// getenv) and changeable afterwards through pols_set_option(); every field's default is the shipped behaviour.
struct Options {
    bool timeline = false;        // POLS_TIMELINE       s_memtime phase stamps of K1 / K1m (debug)
    bool k1_noocc4 = false;       // POLS_K1_NOOCC4      do not hold the ragged one-chunk wave kernel to 128 VGPRs
    bool k1_nofast = false;       // POLS_K1_NOFAST      never take the FAST (aligned, resident) specialisations
    bool k1_notiny = false;       // POLS_K1_NOTINY      never take K1t (four groups per wave)
    bool k1_norc1 = false;        // POLS_K1_NORC1       never take the one-chunk-per-lane wave kernel
    bool k1_shape_team = false;   // POLS_K1_SHAPE=team  f32: 256-thread teams instead of wave-per-group
    bool k1_shape_wave = false;   // POLS_K1_SHAPE=wave  f32: wave-per-group even where the 256-thread team is the default
    bool k1_f64_team256 = false;  // POLS_K1_F64_TEAM=256
    bool no_classes = false;      // POLS_NO_CLASSES     static K1 path: one launch sized for the largest group whatever the spread of group sizes (the round-4 form)
    bool kg_single_buffer = false; // POLS_KG_SINGLE_BUFFER  streamed Gram: one LDS tile (the round-4 form), A/B for the double-buffered DMA
    bool predict_loop = false;    // POLS_PREDICT_LOOP   prediction pass: one looping workgroup per item (the round-4 form)
    bool kg_noyv = false;         // POLS_KG_NOYV        streamed Gram: keep the target in a second MFMA tile at 16 columns
    int k1_passes = 0;            // POLS_K1_PASSES      0: default
    int k1t_rc4 = -1;             // POLS_K1T_RC4        -1: default rule
    int k1t_sub8 = -1;            // POLS_K1T_SUB8       eight-lane K1t teams: -1 default rule (frames that fit 16 chunk slots), 0 never, 1 only frames that fit 8
    int static_engine = 0;        // POLS_STATIC_ENGINE  0 auto, 1 "stream" (three launches), 2 "k2" (wherever it fits), 3 "nok2"
    int rls_engine = 0;           // POLS_RLS_ENGINE     0 auto, 1 "seq" (K3), 2 "scan" (K3c up to 8 features, else the chunk kernels), 3 "chunk" (lane-per-chunk K3s), 4 "halo" (K3c: the halo form where the look-back form would run)
    int rls_early = 0;            // POLS_RLS_EARLY      K3c look-back form: 0 = the tile's record falls out of its scan (default), 1 = computed and published before it (step E; A/B: 34.6 vs 31.2 us on cfg4)
    int rls_spin_limit = -1;      // POLS_RLS_SPINS      K3c look-back form: polls before a wave falls back to the halo (-1: default 64; 0: always fall back -- the test of that path)
    int rolling_engine = 0;       // POLS_ROLLING_ENGINE 0 auto (K4c tiles where they apply), 1 "chunk" (lane-per-chunk K4), 2 "halo" (no packed tiles), 3 "nocompact" (the drop family with nulls stays with the chunk kernels), 4 "halowave" (K4c keeps its halo wave: A/B of the own-halo form), 5 "scatter" (the drop family with nulls: compacted columns + expansion pass instead of the source map)
    int k1_engine = 0;            // POLS_K1_ENGINE      0 auto, 1 "valu", 2 "mfma"
    int k9_take = 0;              // POLS_K9_TAKE        0 auto, 1 "gather", 2 "scatter"
    int k1_nt_loads = -1;         // POLS_K1_NT_LOADS    -1: default rule, 0 / 1
    bool no_split = false;        // POLS_NO_SPLIT       streamed static path: long groups stay one workgroup each (A/B of the segment split)
    bool k2_noprefetch = false;   // POLS_K2_NOPREFETCH  eight-wave K2: one workgroup per group, no next-group prefetch into LDS
    bool k1_noedge = false;       // POLS_K1_NOEDGE      ragged resident frames: the general chunk-by-chunk code instead of the branch-free EDGE kernels
    int k1t_sub32 = -1;           // POLS_K1T_SUB32      -1: default (on), 0: never two groups per wave in the one-shot kernel
    int k1_persist_sub = 0;       // POLS_K1_PERSIST_SUB 0: default rule, 64 / 32 / 16 lanes per group
    int k1_persist = -1;          // POLS_K1_PERSIST     -1: default rule (enough groups), 0 never, 1 whenever the groups fit K1p
    bool debug_skip_fixup = false; // POLS_DEBUG_SKIP_FIXUP  measurement switch: the fix-up dispatch behind a static solve is skipped (flagged groups keep their unusable coefficients)
    bool k1_rc2_wide = true;      // POLS_K1_RC2_WIDE    9-10 columns: K1's two- / four-chunk team beyond 1 024 rows instead of K1m / K2 (0: the round-4 rule, A/B)
    int seg_target = 0;           // POLS_SEG_TARGET     streamed static path: rows per segment of a cut group (0: the default rule)
    int k4p_lps = 0;              // POLS_K4P_LPS        K4p / K3p (k4p_wide.hip): lanes per sequence, 0 auto, 64 / 16 (up to 16 features) / 32 (17..32, RLS)
    int k1_wg = 0;                // POLS_K1_WG          8-column team kernels: 2 / 4 = 512- / 1 024-thread workgroups (2 / 4 times the groups per workgroup, A/B)
    int k1_xcd = 0;               // POLS_K1_XCD         resident K1 kernels: 1 = XCD-contiguous workgroup -> group map (each XCD walks one eighth of the frame)
};
void options_from_env(Options &o);
// key: the variable's name with or without the POLS_ prefix (case-insensitive); value NULL = back to the default.  False = unknown key.
bool options_set(Options &o, const char *key, const char *value);

}  // namespace pols

struct pols_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t switch_event = nullptr;       // pols_set_stream: the new stream waits for what the old one still has in flight
    int num_cus = 0;
    // device scratch (grow-only): [0] group offsets, [1] inputs for HOST batches, [2] outputs, [3] fix-up work area,
    // [4] staged targets / statistics of HOST batches, [5] Gram matrices / chunk totals / staged coefficients, [6] RLS prior mean,
    // [7] status words, [8] K3c tile / block records, [9] group-key ingestion (K9), [10] chunk / group tables of the dynamic kernels
    // (nothing else may take this slot: the tables are cached across calls), [11] timeline stamps, [12] Arrow ingestion,
    // [13] collective staging, [14] dynamic-path prep / null-policy compaction (dyn_prep.hip), [15] their host-batch outputs
    // [16] sequence-start bytes of the row-parallel dynamic kernels (K3c / K4c), [17] null-weight-filled copy of a DEVICE batch's weights column (static entries), [18] first rows of K3c's packed tiles, [19..22] row compaction of the rolling entry (columns, coefficients, start bytes, tile map), [23] segment tables + partial Gram matrices of the streamed static path,
    // [24] group lists of the size classes, [25] K3c halo form: first row of the sequence in front of every tile, [26] segment tables of the last size class, [27] K4c: rows without a factorisation (the LU list), [28] K3c look-back form: the tiles' record granules
    pols::Scratch scratch[29];
    pols::Options opt;
    bool timing = false;
    int timing_stride = 1;                   // time every n-th eligible launch (pols_timing_enable(ctx, n))
    int timing_tick = 0;
    bool timing_open = false;                // a timing_begin() is waiting for its timing_end()
    std::vector<pols::TimedLaunch> timed;   // pool of event pairs
    size_t timed_used = 0;
    std::string last_kernel;
    // cache of the last uploaded group_offsets so steady-state calls on the same frame do not re-upload metadata.  A hit is
    // either PROMISED by the caller (same host pointer, count and non-zero pols_batch.offsets_generation) or VERIFIED: same
    // count, same content hash AND a memcmp against the host copy kept here -- a hash collision cannot alias two frames.
    const int64_t *offs_host = nullptr;
    uint64_t offs_generation = 0;
    std::vector<int64_t> offs_copy;
    int64_t offs_n = -1;
    uint64_t offs_sum = 0;
    uint64_t offs_id = 0;                    // bumps whenever different offsets are uploaded (keys the chunk-table cache)
    // pinned staging ring for small host -> device uploads (offsets, pointer tables): the copy is asynchronous and the
    // caller's array may be freed on return; a slot is reused only after the event recorded behind its copy has completed
    struct PinnedSlot { void *ptr = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    PinnedSlot pinned[4];
    int pinned_next = 0;
    int64_t offs_max_rows = 0;
    int64_t offs_min_rows = 0;               // fewest rows of a NON-EMPTY group (0: no group has rows)
    int64_t offs_hist_cnt[48] = {0}, offs_hist_rows[48] = {0};   // groups / rows by size bucket b: 2^(b-1) < rows <= 2^b (bucket 0: 0 or 1 rows): the size-class split of ls_core
    int32_t offs_small_mask = 0;             // bit b: some group has (b ? 2 << b : 0) < rows <= 4 << b, b = 0..3 (1-4, 5-8, 9-16, 17-32 rows): K6s team sizes
    int64_t offs_tail_group = -1;            // last group with rows (K1p hands it to one wave when n_rows is not a multiple of the vector width)
    int64_t offs_wave_overflow = 0;          // sum over groups of the rows beyond 1 021 (see k1_launch_kw)
    int32_t *fb_flag = nullptr;              // device word, see K1Args::fb_flag
    int32_t epoch = 0;
    bool offs_aligned[2] = {false, false};   // every group start AND size a multiple of 2 / of 4 rows
    // cache of the chunk tables of the dynamic kernels (scratch slot 4) for mask-free batches: rebuilt only when the
    // offsets, min_periods or the chunk length change
    struct { uint64_t offs_id = 0; int64_t n_groups = -1, n_rows = -1, mp = -1, n_chunks = 0; int32_t chunk_len = 0;
             const void *tab = nullptr; size_t b_groups = 0; } chunk_cache;
    // K3c (k3c_scan.hip), scratch slot 18: first row of every PACKED tile (tiles cut at sequence starts, single-pass mode); n_tiles 0 =
    // this frame does not pack (a sequence longer than a tile, or tiles too empty)
    struct { const void *ptr = nullptr; uint64_t offs_id = 0; int64_t n_groups = -1, n_rows = -1, tile_rows = 0, n_tiles = 0; } k3c;
    // K3c halo form, scratch slot 25: per tile the first row of the sequence that holds the row in front of it
    struct { const void *ptr = nullptr; uint64_t offs_id = 0; int64_t n_groups = -1, n_rows = -1, tile_rows = 0; } k3h;
    // segment tables of the streamed static path (scratch slot 23: long groups cut into segments): rebuilt when other offsets arrive
    struct { const void *ptr = nullptr; uint64_t offs_id = 0; int64_t cut[3] = {0, 0, 0}, n[4] = {0, 0, 0, 0}; int n_cut = 0;
             std::vector<int32_t> host_last; } class_cache;   // group lists of the size classes (slot 24; host_last: the last class' ids, for its segment tables)
    struct { const void *ptr = nullptr; uint64_t offs_id = 0; int64_t n_groups = -1, n_rows = -1, seg_target = 0, n_seg = 0, max_len = 0, max_seg = 0, class_key = 0, n_items = 0; size_t nz2 = 0; bool nulls = false; } seg_cache[2];   // [0] whole-frame tables (slot 23), [1] the last size class' tables (slot 26)
    const void *k3c_gran_ptr = nullptr;      // K3c look-back form (scratch slot 28): the slot's address when it was last zeroed; the launches' running tag
    unsigned long long k3c_epoch = 0;
    size_t k3c_gran_cap = 0;
    const void *k4c_fix_ptr = nullptr;       // K4c's LU list (scratch slot 27): the slot's address when its counters were last zeroed, and whose turn it is
    uint64_t k4c_fix_turn = 0;
    // sequence-start bytes (scratch slot 16): rebuilt when other offsets arrive
    struct { const void *ptr = nullptr; uint64_t offs_id = 0; int64_t n_groups = -1, n_rows = -1; } start_flags;
};

namespace pols {

int ensure_scratch(pols_ctx *ctx, int slot, size_t bytes, void **out);
// uploads group offsets (cached), returns device pointer and max group size; generation: pols_batch.offsets_generation
int upload_offsets(pols_ctx *ctx, const int64_t *offs, int64_t n_groups, const int64_t **d_offs, int64_t *max_rows,
                   uint64_t generation = 0);
// asynchronous upload of a small host array through the context's pinned ring (no stream synchronisation; `src` may be
// freed on return)
int upload_small(pols_ctx *ctx, void *dst_device, const void *src, size_t bytes);
// timing helpers: no-ops unless ctx->timing
void timing_begin(pols_ctx *ctx);
void timing_end(pols_ctx *ctx);
// single-kernel form: hands out the next event pair for hipExtLaunchKernelGGL, which stamps them with the kernel's own
// begin / end -- the duration rocprofv3 reports, without the event packets' latency inside the bracket.  False = timing off.
bool timing_pair(pols_ctx *ctx, hipEvent_t *start, hipEvent_t *stop);
// POLS_TIMELINE=1 debugging: synchronise, read n_stamps s_memtime stamps per group, print phase statistics to stderr
int report_timeline(pols_ctx *ctx, const unsigned long long *d_dbg, int64_t n_groups, int n_stamps, const char *name);

inline size_t dtype_size(int dtype) { return dtype == POLS_F32 ? 4 : 8; }
inline size_t round256(size_t b) { return (b + 255) & ~(size_t)255; }

// hipFuncSetAttribute is per device: a kernel's launcher keeps one of these masks and applies the attribute once per device.  Two
// host threads racing on the same device both apply it (idempotent) before either publishes the bit.
struct OncePerDevice {
    std::atomic<uint64_t> mask{0};
    bool needed(int device) const { return (mask.load(std::memory_order_acquire) & (1ull << (device & 63))) == 0; }
    void done(int device) { mask.fetch_or(1ull << (device & 63), std::memory_order_release); }
};

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

// DPP controls (gfx9): quad_perm = 0x00..0xFF, row_mirror 0x140, row_half_mirror 0x141,
// row_bcast:15 0x142, row_bcast:31 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(dpp_mov<CTRL, ROW_MASK>(__float_as_int(v)));
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_get(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_mov<CTRL, ROW_MASK>((int)(b & 0xffffffffLL));
    const int hi = dpp_mov<CTRL, ROW_MASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the 64 lanes of a wave; the total is valid in lanes 48..63 (row 3).  All 64 lanes must
// be active.  4 intra-row steps (xor1, xor2, half-mirror, mirror) + row_bcast15 + row_bcast31.
template <typename T>
__device__ __forceinline__ T wave_sum_row3(T v) {
    v += dpp_get<0xB1>(v);         // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);         // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);        // row_half_mirror
    v += dpp_get<0x140>(v);        // row_mirror
    v += dpp_get<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    v += dpp_get<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

// ---- reduce-scatter over a wave (gfx950 v_permlane32_swap / v_permlane16_swap) -------------------------
// pair_lo_hi(a, b): afterwards `a` holds, in lanes 0-31, a[l] + a[l+32] and, in lanes 32-63, b[l-32] + b[l].
__device__ __forceinline__ void pair_halves(float &a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void pair_halves(double &a, double b) {
    const unsigned long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    a = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) +
        __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
// pair_rows(a, b): afterwards `a` holds per 16-lane row [a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3].
__device__ __forceinline__ void pair_rows(float &a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void pair_rows(double &a, double b) {
    const unsigned long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ba, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ba >> 32), (unsigned)(bb >> 32), false, false);
    a = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) +
        __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
// all-reduce inside each 16-lane row: 4 DPP steps that fuse into v_add_*_dpp (no temporaries)
template <typename T>
__device__ __forceinline__ T row_allreduce(T v) {
    v += dpp_get<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);  // row_half_mirror
    v += dpp_get<0x140>(v);  // row_mirror
    return v;
}

// Sum N per-lane values over the wave.  On return u[i] (i < (N+3)/4) holds, in every lane of 16-lane row r,
// the wave total of v[4*i + rs_perm(r)]; v[] is clobbered.  ~2.5 N VALU ops instead of 6 N, and the live
// register count halves at each of the first two steps.
__device__ __forceinline__ constexpr int rs_perm(int row) { return row == 1 ? 2 : (row == 2 ? 1 : row); }

template <typename T, int N>
__device__ __forceinline__ void wave_reduce_scatter(T (&v)[N], T (&u)[(N + 3) / 4]) {
    constexpr int N2 = (N + 1) / 2, N4 = (N + 3) / 4;
#pragma unroll
    for (int i = 0; i < N2; ++i) {
        const T b = (2 * i + 1 < N) ? v[2 * i + 1] : T(0);
        pair_halves(v[2 * i], b);      // v[2i]: lanes 0-31 <- total(v[2i]) over halves, lanes 32-63 <- total(v[2i+1])
    }
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        T a = v[4 * i];
        const T b = (2 * i + 1 < N2) ? v[4 * i + 2] : T(0);
        pair_rows(a, b);               // rows: [v4i, v4i+2, v4i+1, v4i+3]
        u[i] = row_allreduce(a);
    }
}

__device__ __forceinline__ float readlane63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double readlane63(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- null policies of the static entry (src/expressions.rs:201-296).  A null is a NaN; for the drop family a 0 in the
// optional validity bytes drops the row as well.
__device__ __forceinline__ bool null_checks_x(int policy) {   // rows with a null FEATURE leave the fit
    return policy == POLS_NULL_DROP || policy == POLS_NULL_DROP_ZERO || policy == POLS_NULL_DROP_WINDOW;
}
__device__ __forceinline__ bool null_checks_y(int policy) {   // rows with a null TARGET leave the fit
    return null_checks_x(policy) || policy == POLS_NULL_DROP_Y_ZERO_X;
}
// is row r (absolute) part of the fit?  x[] are the feature columns, y the target
template <typename T>
__device__ __forceinline__ bool null_row_in_fit(int policy, const uint8_t *valid, const void *y, const void *const *x, int ku, int64_t r) {
    if (policy == POLS_NULL_IGNORE || policy == POLS_NULL_ZERO) return true;
    if (valid && !valid[r]) return false;
    const T yv = static_cast<const T *>(y)[r];
    if (yv != yv) return false;
    if (null_checks_x(policy))
        for (int j = 0; j < ku; ++j) { const T v = static_cast<const T *>(x[j])[r]; if (v != v) return false; }
    return true;
}
// bit-level select, so that no floating-point reasoning of the optimiser is involved
template <typename T> __device__ __forceinline__ T nan_if(unsigned cond, T v);
template <> __device__ __forceinline__ float nan_if<float>(unsigned cond, float v) {
    return __uint_as_float(cond ? 0x7fc00000u : __float_as_uint(v));
}
template <> __device__ __forceinline__ double nan_if<double>(unsigned cond, double v) {
    return __longlong_as_double(cond ? 0x7ff8000000000000LL : __double_as_longlong(v));
}
template <typename T>
__device__ __forceinline__ T null_fill(int policy, T v) { return (policy != POLS_NULL_IGNORE && v != v) ? T(0) : v; }

template <typename T> struct Vec16;  // 16-byte vector of T
template <> struct Vec16<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double2; static constexpr int N = 2; };

// Streaming 16-byte store of an output that this launch never reads again: the `nt` hint keeps it from displacing the
// input stream in L2 (measured on the 9-reads-1-write pattern of the static kernels: 75.1 -> 71.5 us per 400 MB).
__device__ __forceinline__ void store_stream(float4 *dst, const float4 &v) {
    __builtin_nontemporal_store(v.x, &dst->x); __builtin_nontemporal_store(v.y, &dst->y);
    __builtin_nontemporal_store(v.z, &dst->z); __builtin_nontemporal_store(v.w, &dst->w);
}
__device__ __forceinline__ void store_stream(double2 *dst, const double2 &v) {
    __builtin_nontemporal_store(v.x, &dst->x); __builtin_nontemporal_store(v.y, &dst->y);
}

// Streaming 16-byte load of an input this launch reads exactly once (`nt`: no L2 allocation priority for the line).
__device__ __forceinline__ float4 load_stream(const float4 *p) {
    using F = __attribute__((ext_vector_type(4))) float;
    const F v = __builtin_nontemporal_load(reinterpret_cast<const F *>(p));
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ double2 load_stream(const double2 *p) {
    using D = __attribute__((ext_vector_type(2))) double;
    const D v = __builtin_nontemporal_load(reinterpret_cast<const D *>(p));
    return double2{v.x, v.y};
}

template <typename T> __device__ __forceinline__ T vget(const typename Vec16<T>::type &v, int i);
template <> __device__ __forceinline__ float vget<float>(const float4 &v, int i) {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}
template <> __device__ __forceinline__ double vget<double>(const double2 &v, int i) { return i == 0 ? v.x : v.y; }

#endif  // __HIPCC__

}  // namespace pols
