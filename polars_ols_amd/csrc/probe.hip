// probe.hip -- pols_stream_probe: one pass over a batch's columns with the arithmetic removed.
//
// A measurement aid, not part of the reference interface: bench.py's `roofline.stream_ceiling`.  The static kernels are HBM-bound, and
// the rate HBM admits depends on the traffic mix (k + 1 column streams read, one written) -- on MI355X a plain 16-byte copy of this
// mix reaches 5.7-5.9 TB/s of the 8 TB/s the datasheet names (profiles/r01_bw_probe.txt).  This kernel reads every feature column,
// the target and the weights of the caller's batch exactly like K1 does (16-byte streaming loads down the row axis, 256 threads per
// workgroup, one chunk per lane), adds them up (one VALU add per loaded value: nothing a memory-bound kernel could feel) and writes the
// sum over the predictions column with streaming stores.  Same buffers, same frames, same launch timing as the kernel it is read beside.
#include "common.hpp"

#include <algorithm>
#include <cstring>

namespace pols {

struct ProbeArgs {
    const void *cols[POLS_MAX_FEATURES + 2];
    int32_t n_cols;
    int64_t n_rows;
    void *out;
};

template <typename T>
__global__ void __launch_bounds__(256) stream_probe_kernel(const ProbeArgs a) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    const int64_t row0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (row0 >= a.n_rows) return;
    T *out = static_cast<T *>(a.out);
    if (row0 + VEC <= a.n_rows) {
        T s[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[v] = T(0);
        for (int j0 = 0; j0 < a.n_cols; j0 += 9) {              // nine loads in flight per lane, like the headline kernel's chunk
            V t[9];
#pragma unroll
            for (int u = 0; u < 9; ++u)
                if (j0 + u < a.n_cols) t[u] = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.cols[j0 + u]) + row0));
#pragma unroll
            for (int u = 0; u < 9; ++u)
                if (j0 + u < a.n_cols) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) s[v] += vget<T>(t[u], v);
                }
        }
        V o;
        if constexpr (VEC == 4) o = V{s[0], s[1], s[2], s[3]}; else o = V{s[0], s[1]};
        store_stream(reinterpret_cast<V *>(out + row0), o);
    } else {
        for (int64_t r = row0; r < a.n_rows; ++r) {
            T s = T(0);
            for (int j = 0; j < a.n_cols; ++j) s += static_cast<const T *>(a.cols[j])[r];
            out[r] = s;
        }
    }
}

// mode 1: the same traffic as a PERSISTENT grid-stride stream -- no per-workgroup launch ramp or tail, the next piece's loads issued
// before the current piece is stored (18 x 16-byte loads in flight per lane): what the memory system admits for this mix when nothing
// else (dispatch, ramp, per-group arithmetic) is in the way.  Pieces of 256 x VEC rows, like K1's chunks.
template <typename T>
__global__ void __launch_bounds__(256) stream_probe_persistent_kernel(const ProbeArgs a, const int64_t n_pieces) {
    using V = typename Vec16<T>::type;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NL = 9;
    T *out = static_cast<T *>(a.out);
    const int nc = a.n_cols < NL ? a.n_cols : NL;            // (the probe's configs have at most 9 + 1 streams; extra columns: second pass)
    int64_t piece = blockIdx.x;
    V cur[NL], nxt[NL];
    auto issue = [&](int64_t pc, V (&t)[NL]) {
        const int64_t row0 = (pc * 256 + threadIdx.x) * VEC;
#pragma unroll
        for (int u = 0; u < NL; ++u)
            if (u < nc) t[u] = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.cols[u]) + row0));
    };
    if (piece < n_pieces) issue(piece, cur);
    for (; piece < n_pieces; piece += gridDim.x) {
        const int64_t np = piece + gridDim.x;
        if (np < n_pieces) issue(np, nxt);
        T s[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[v] = T(0);
#pragma unroll
        for (int u = 0; u < NL; ++u)
            if (u < nc) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) s[v] += vget<T>(cur[u], v);
            }
        const int64_t row0 = (piece * 256 + threadIdx.x) * VEC;
        for (int j = NL; j < a.n_cols; ++j) {                // streams beyond nine (weights at 8 features + target)
            const V t = load_stream(reinterpret_cast<const V *>(static_cast<const T *>(a.cols[j]) + row0));
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] += vget<T>(t, v);
        }
        V o;
        if constexpr (VEC == 4) o = V{s[0], s[1], s[2], s[3]}; else o = V{s[0], s[1]};
        store_stream(reinterpret_cast<V *>(out + row0), o);
#pragma unroll
        for (int u = 0; u < NL; ++u) cur[u] = nxt[u];
    }
}

}  // namespace pols

extern "C" int pols_stream_probe_ex(pols_ctx *ctx, const pols_batch *b, void *pred_out, int mode);
extern "C" int pols_stream_probe(pols_ctx *ctx, const pols_batch *b, void *pred_out) { return pols_stream_probe_ex(ctx, b, pred_out, 0); }

extern "C" int pols_stream_probe_ex(pols_ctx *ctx, const pols_batch *b, void *pred_out, int mode) {
    using namespace pols;
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    if (!b || !pred_out) return fail(POLS_ERR_INVALID, "batch / pred_out is NULL");
    if (b->mem != POLS_MEM_DEVICE) return fail(POLS_ERR_INVALID, "the stream probe takes a DEVICE batch");
    if (b->dtype != POLS_F32 && b->dtype != POLS_F64) return fail(POLS_ERR_INVALID, "dtype");
    if (b->n_features < 0 || b->n_features > POLS_MAX_FEATURES || !b->y || (b->n_features > 0 && !b->x_cols))
        return fail(POLS_ERR_INVALID, "the stream probe takes up to %d feature columns and a target", POLS_MAX_FEATURES);
    POLS_HIP(hipSetDevice(ctx->device));
    ProbeArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int j = 0; j < b->n_features; ++j) {
        if (!b->x_cols[j] || (reinterpret_cast<uintptr_t>(b->x_cols[j]) & 15)) return fail(POLS_ERR_INVALID, "column %d is NULL or not 16-byte aligned", j);
        a.cols[a.n_cols++] = b->x_cols[j];
    }
    a.cols[a.n_cols++] = b->y;
    if (b->weights) a.cols[a.n_cols++] = b->weights;
    a.n_rows = b->n_rows;
    a.out = pred_out;
    if (b->n_rows <= 0) return POLS_OK;
    const int vec = b->dtype == POLS_F32 ? 4 : 2;
    const int64_t blocks = (b->n_rows + 256 * (int64_t)vec - 1) / (256 * (int64_t)vec);
    if (blocks > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many rows for one launch");
    ctx->last_kernel = b->dtype == POLS_F32 ? "stream_probe_f32" : "stream_probe_f64";
    hipEvent_t ev0, ev1;
    const bool timed = timing_pair(ctx, &ev0, &ev1);
    const int64_t full_pieces = b->n_rows / (256 * (int64_t)vec);
    if (mode == 1 && full_pieces > 0 && full_pieces * 256 * vec == b->n_rows) {
        ctx->last_kernel = b->dtype == POLS_F32 ? "stream_probe_persistent_f32" : "stream_probe_persistent_f64";
        const unsigned grid = (unsigned)std::min<int64_t>(full_pieces, (int64_t)std::max(1, ctx->num_cus) * 4);
        if (b->dtype == POLS_F32)
            hipExtLaunchKernelGGL(stream_probe_persistent_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, timed ? ev0 : nullptr,
                                  timed ? ev1 : nullptr, 0, a, full_pieces);
        else
            hipExtLaunchKernelGGL(stream_probe_persistent_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, timed ? ev0 : nullptr,
                                  timed ? ev1 : nullptr, 0, a, full_pieces);
        POLS_HIP(hipGetLastError());
        return POLS_OK;
    }
    if (b->dtype == POLS_F32) {
        if (timed) hipExtLaunchKernelGGL(stream_probe_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(stream_probe_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    } else {
        if (timed) hipExtLaunchKernelGGL(stream_probe_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(stream_probe_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    }
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}
