// arrow.hip -- the plugin-side marshalling behind the C-ABI: Arrow C Data Interface in, Arrow C Data Interface out, one entry per
// plugin body of src/expressions.rs (pols_*_arrow, include/pols_mi355x.h):
//   least_squares / least_squares_coefficients (:390-446)                  pols_least_squares_arrow
//   least_squares_statistics (:448-509)                                     pols_least_squares_statistics_arrow
//   multi_target_least_squares (:511-591; inputs[0] is a STRUCT Series)     pols_multi_target_least_squares_arrow
//   recursive_least_squares[_coefficients] (:593-646)                       pols_recursive_least_squares_arrow
//   rolling_least_squares[_coefficients] (:648-701)                         pols_rolling_least_squares_arrow
//   predict (:706-741; inputs[0] is the coefficients STRUCT, one row per row) pols_predict_arrow
//
// Replaces what src/expressions.rs does around the solvers:
//   convert_polars_to_ndarray (:66-103)     cast every input Series to Float64, null -> NaN, rechunk, copy into an ndarray
//   construct_features_array (:22-63)       the column -> row-major copy (deleted: the kernels read columns)
//   compute_is_valid_mask / handle_nulls    (:201-296) -- here: validity BITMAPS become NaNs in the staged columns, which is what
//                                           the kernels' fused null policies key on
//   convert_array_to_struct_series (:114-143), mask_predictions (:145-158)   struct-of-coefficients / nullable predictions out
// A Polars Series reaches a plugin as one Arrow array per chunk: values buffer + optional validity bitmap + an element offset
// (slices share buffers).  Each chunk is copied to the device as it is (raw values, raw bitmap bytes) and ONE pass of
// `arrow_ingest_kernel` per chunk casts (i8..u64 / f32 / f64 -> the compute dtype), applies the bitmap (null -> NaN, or the
// fill value for weights) and lands the rows at the chunk's position in a contiguous device column -- the cast, the
// fill_null and the rechunk of :80-91 in one HBM pass behind the PCIe copy, instead of three host passes per column.
#include <algorithm>
#include <cstdlib>
#include <limits>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"

namespace pols {

template <typename S, typename T>
__global__ void __launch_bounds__(256) arrow_ingest_kernel(const S *src, const uint8_t *bits, int bit_offset, int64_t n, T fill, T *dst, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool valid = true;
    if (bits) { const int64_t b = bit_offset + i; valid = (bits[b >> 3] >> (b & 7)) & 1; }
    dst[i * stride] = valid ? (T)src[i] : fill;
}

// a struct Series' own validity bitmap: a null STRUCT row is a null in every field
template <typename T>
__global__ void __launch_bounds__(256) arrow_parent_mask_kernel(const uint8_t *bits, int bit_offset, int64_t n, T fill, T *dst, int64_t stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t b = bit_offset + i;
    if (!((bits[b >> 3] >> (b & 7)) & 1)) dst[i * stride] = fill;
}

// validity bitmap of an output column: bit i = value i is not NaN; returns the null count through a device counter
template <typename T>
__global__ void __launch_bounds__(256) arrow_validity_kernel(const T *v, int64_t n, uint8_t *bits, unsigned long long *nulls) {
    const int64_t byte = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (byte * 8 >= n) return;
    unsigned m = 0, cnt = 0;
    for (int b = 0; b < 8; ++b) {
        const int64_t i = byte * 8 + b;
        if (i < n) { const T x = v[i]; if (x == x) m |= 1u << b; else ++cnt; }
    }
    bits[byte] = (uint8_t)m;
    if (cnt) atomicAdd(nulls, (unsigned long long)cnt);
}

struct ArrowType { int bytes; char code; };   // code: the format character

static bool arrow_type(const char *fmt, ArrowType *t) {
    if (!fmt || !fmt[0] || fmt[1]) return false;
    switch (fmt[0]) {
        case 'g': *t = {8, 'g'}; return true;   // float64
        case 'f': *t = {4, 'f'}; return true;   // float32
        case 'l': *t = {8, 'l'}; return true;   // int64
        case 'L': *t = {8, 'L'}; return true;
        case 'i': *t = {4, 'i'}; return true;   // int32
        case 'I': *t = {4, 'I'}; return true;
        case 's': *t = {2, 's'}; return true;   // int16
        case 'S': *t = {2, 'S'}; return true;
        case 'c': *t = {1, 'c'}; return true;   // int8
        case 'C': *t = {1, 'C'}; return true;
        default: return false;
    }
}

template <typename T>
static int ingest_launch(pols_ctx *ctx, char code, const void *src, const uint8_t *bits, int bit_offset, int64_t n, T fill, T *dst, int64_t stride) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (n == 0) return POLS_OK;
#define INGEST(S) hipLaunchKernelGGL((arrow_ingest_kernel<S, T>), dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const S *>(src), bits, bit_offset, n, fill, dst, stride)
    switch (code) {
        case 'g': INGEST(double); break;
        case 'f': INGEST(float); break;
        case 'l': INGEST(int64_t); break;
        case 'L': INGEST(uint64_t); break;
        case 'i': INGEST(int32_t); break;
        case 'I': INGEST(uint32_t); break;
        case 's': INGEST(int16_t); break;
        case 'S': INGEST(uint16_t); break;
        case 'c': INGEST(int8_t); break;
        case 'C': INGEST(uint8_t); break;
        default: return fail(POLS_ERR_UNSUPPORTED, "arrow format '%c'", code);
    }
#undef INGEST
    POLS_HIP(hipGetLastError());
    return POLS_OK;
}

// One logical column as the kernels want it: a primitive Series, or ONE FIELD of a struct Series (multi-target targets, the
// coefficients struct of `predict`).  A chunk is (values array, first element, length) plus -- for a struct field -- the parent
// chunk whose validity bitmap nulls whole rows.  Arrow: a struct array's offset applies to its children on top of their own.
struct ChunkView {
    const ArrowArray *a;         // the primitive array that holds the values
    int64_t offset, length;      // first element / number of elements of `a` this chunk covers
    const ArrowArray *parent;    // struct chunk with nulls of its own, or nullptr
};
struct ColView {
    ArrowType type;
    const char *name = "";
    std::vector<ChunkView> chunks;
    int64_t rows() const { int64_t n = 0; for (const auto &c : chunks) n += c.length; return n; }
    bool no_nulls() const {                                    // Arrow carries the null counts (-1 = not computed: unknown)
        for (const auto &c : chunks) if (c.a->null_count != 0 || c.parent) return false;
        return true;
    }
    size_t raw_bytes() const {                                 // staging for the largest chunk: values + bitmap bytes
        size_t m = 256;
        for (const auto &c : chunks) m = std::max(m, round256((size_t)c.length * type.bytes) + round256((size_t)c.length / 8 + 16));
        return m;
    }
};

static int primitive_ok(const ArrowArray *a, const char *what, int i) {
    if (!a || a->length < 0 || a->offset < 0 || a->n_buffers < 2 || !a->buffers || (a->length && !a->buffers[1]))
        return fail(POLS_ERR_INVALID, "%s: chunk %d is not a primitive array (2 buffers)", what, i);
    return POLS_OK;
}

static int view_primitive(const pols_arrow_column *c, const char *what, ColView *v) {
    if (!c || !c->schema || (!c->chunks && c->n_chunks) || c->n_chunks < 0) return fail(POLS_ERR_INVALID, "%s: NULL schema / chunks", what);
    if (!arrow_type(c->schema->format, &v->type))
        return fail(POLS_ERR_UNSUPPORTED, "%s: arrow format '%s' (numeric primitives only: the reference casts to Float64, src/expressions.rs:80)",
                    what, c->schema->format ? c->schema->format : "(null)");
    v->name = c->schema->name ? c->schema->name : "";
    v->chunks.clear();
    for (int i = 0; i < c->n_chunks; ++i) {
        const ArrowArray *a = c->chunks[i];
        int rc = primitive_ok(a, what, i);
        if (rc) return rc;
        v->chunks.push_back({a, a->offset, a->length, nullptr});
    }
    return POLS_OK;
}

// the fields of a struct Series ("+s"): one ColView per field
static int view_struct(const pols_arrow_column *c, const char *what, std::vector<ColView> *fields) {
    if (!c || !c->schema || (!c->chunks && c->n_chunks) || c->n_chunks < 0) return fail(POLS_ERR_INVALID, "%s: NULL schema / chunks", what);
    const ArrowSchema *sc = c->schema;
    if (!sc->format || std::strcmp(sc->format, "+s") != 0 || sc->n_children < 1 || !sc->children)
        return fail(POLS_ERR_PANIC, "%s must be of polars struct dtype", what);   // ex.rs:513-517, :712-714 (`expect`)
    fields->assign((size_t)sc->n_children, ColView());
    for (int64_t f = 0; f < sc->n_children; ++f) {
        ColView &v = (*fields)[(size_t)f];
        const ArrowSchema *fs = sc->children[f];
        if (!fs || !arrow_type(fs->format, &v.type))
            return fail(POLS_ERR_UNSUPPORTED, "%s: field %d has arrow format '%s' (numeric primitives only)", what, (int)f,
                        fs && fs->format ? fs->format : "(null)");
        v.name = fs->name ? fs->name : "";
        for (int i = 0; i < c->n_chunks; ++i) {
            const ArrowArray *pa = c->chunks[i];
            if (!pa || pa->length < 0 || pa->offset < 0 || pa->n_children != sc->n_children || !pa->children)
                return fail(POLS_ERR_INVALID, "%s: chunk %d is not a struct array of %d fields", what, i, (int)sc->n_children);
            const ArrowArray *a = pa->children[f];
            int rc = primitive_ok(a, what, i);
            if (rc) return rc;
            if (a->length < pa->offset + pa->length) return fail(POLS_ERR_INVALID, "%s: field %d of chunk %d is shorter than its struct", what, (int)f, i);
            const bool pnull = pa->null_count != 0 && pa->n_buffers >= 1 && pa->buffers && pa->buffers[0];
            v.chunks.push_back({a, a->offset + pa->offset, pa->length, pnull ? pa : nullptr});
        }
    }
    return POLS_OK;
}

// One column: every chunk host -> device (values, bitmap bytes), cast + null-fill into dst[0], dst[stride], ...  `raw` is a device
// staging area of at least v.raw_bytes() re-used chunk after chunk (stream order keeps that safe).
template <typename T>
static int ingest_view(pols_ctx *ctx, const ColView &v, char *raw, T fill, T *dst, int64_t stride = 1) {
    int64_t row = 0;
    const ArrowType &t = v.type;
    for (const ChunkView &cv : v.chunks) {
        const ArrowArray *a = cv.a;
        const int64_t n = cv.length;
        if (n == 0) continue;
        const char *values = static_cast<const char *>(a->buffers[1]) + (size_t)cv.offset * t.bytes;
        const uint8_t *bitmap = (a->null_count != 0) ? static_cast<const uint8_t *>(a->buffers[0]) : nullptr;
        const bool same = (t.code == (sizeof(T) == 8 ? 'g' : 'f'));
        if (same && !bitmap && stride == 1) {                 // already the compute dtype, no nulls: straight into place
            POLS_HIP(hipMemcpyAsync(dst + row, values, (size_t)n * t.bytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            const size_t vbytes = (size_t)n * t.bytes;
            uint8_t *dbits = nullptr;
            int bit_offset = 0;
            POLS_HIP(hipMemcpyAsync(raw, values, vbytes, hipMemcpyHostToDevice, ctx->stream));
            if (bitmap) {
                const int64_t b0 = cv.offset >> 3, b1 = (cv.offset + n + 7) >> 3;
                dbits = reinterpret_cast<uint8_t *>(raw + round256(vbytes));
                bit_offset = (int)(cv.offset & 7);
                POLS_HIP(hipMemcpyAsync(dbits, bitmap + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, ctx->stream));
            }
            int rc = ingest_launch<T>(ctx, t.code, raw, dbits, bit_offset, n, fill, dst + row * stride, stride);
            if (rc) return rc;
        }
        if (cv.parent) {                                      // null struct rows: null in this field too
            const ArrowArray *pa = cv.parent;
            const int64_t b0 = pa->offset >> 3, b1 = (pa->offset + n + 7) >> 3;
            uint8_t *dbits = reinterpret_cast<uint8_t *>(raw);
            POLS_HIP(hipMemcpyAsync(dbits, static_cast<const uint8_t *>(pa->buffers[0]) + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL((arrow_parent_mask_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dbits,
                               (int)(pa->offset & 7), n, fill, dst + row * stride, stride);
            POLS_HIP(hipGetLastError());
        }
        row += n;
    }
    return POLS_OK;
}

// ---- output arrays: malloc'd buffers handed over with release callbacks, as the interface prescribes.  An output is BUILT with
// release set from its first allocation on (so that release_array can unwind a half-built tree: it tolerates NULL children /
// buffers) and PUBLISHED to the caller's structs only when complete; on any error the partial tree is released here.
static void release_array(ArrowArray *a) {
    if (!a || !a->release) return;
    for (int64_t i = 0; a->children && i < a->n_children; ++i) {
        if (!a->children[i]) continue;
        if (a->children[i]->release) a->children[i]->release(a->children[i]);
        std::free(a->children[i]);
    }
    std::free(a->children);
    for (int64_t i = 0; a->buffers && i < a->n_buffers; ++i) std::free(const_cast<void *>(a->buffers[i]));
    std::free(a->buffers);
    a->release = nullptr;
}

static void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; s->children && i < s->n_children; ++i) {
        if (!s->children[i]) continue;
        if (s->children[i]->release) s->children[i]->release(s->children[i]);
        std::free(s->children[i]);
    }
    std::free(s->children);
    std::free(const_cast<char *>(s->format));
    std::free(const_cast<char *>(s->name));
    s->release = nullptr;
}

static char *dup_cstr(const char *s) {
    const size_t n = std::strlen(s) + 1;
    char *d = static_cast<char *>(std::malloc(n));
    std::memcpy(d, s, n);
    return d;
}

static void make_schema(ArrowSchema *s, const char *format, const char *name, int64_t n_children) {
    std::memset(s, 0, sizeof(*s));
    s->format = dup_cstr(format);
    s->name = dup_cstr(name ? name : "");
    s->flags = 2;   // ARROW_FLAG_NULLABLE
    s->n_children = n_children;
    s->children = n_children ? static_cast<ArrowSchema **>(std::calloc((size_t)n_children, sizeof(ArrowSchema *))) : nullptr;
    for (int64_t i = 0; i < n_children; ++i) s->children[i] = static_cast<ArrowSchema *>(std::calloc(1, sizeof(ArrowSchema)));
    s->release = release_schema;
}

// an array with `n_buffers` (NULL) buffers and `n_children` zeroed children, releasable from here on
static void make_array(ArrowArray *a, int64_t length, int64_t n_buffers, int64_t n_children) {
    std::memset(a, 0, sizeof(*a));
    a->length = length;
    a->n_buffers = n_buffers;
    a->buffers = static_cast<const void **>(std::calloc((size_t)std::max<int64_t>(n_buffers, 1), sizeof(void *)));
    a->n_children = n_children;
    a->children = n_children ? static_cast<ArrowArray **>(std::calloc((size_t)n_children, sizeof(ArrowArray *))) : nullptr;
    for (int64_t i = 0; i < n_children; ++i) a->children[i] = static_cast<ArrowArray *>(std::calloc(1, sizeof(ArrowArray)));
    a->release = release_array;
}

// device column of n values (element stride `stride`) -> host primitive array; NaN -> null when `nan_is_null` (validity bitmap +
// null count made on the device).  `dscratch`: n * sizeof(T) (only for stride != 1) + n / 8 + 512 bytes of device scratch.
template <typename T>
static int export_column(pols_ctx *ctx, const T *dvals, int64_t n, bool nan_is_null, char *dscratch, ArrowArray *out, int64_t stride = 1) {
    make_array(out, n, 2, 0);
    // Both host buffers exist BEFORE the first copy is queued, and no early return leaves a copy in flight behind it: the caller
    // answers an error with release_array(out), which frees these buffers -- a DMA still landing in them would write freed memory.
    const int64_t nbytes = (n + 7) / 8;
    void *hv = std::malloc(std::max<size_t>(1, (size_t)n * sizeof(T)));
    void *hb = (nan_is_null && n) ? std::malloc((size_t)nbytes + sizeof(unsigned long long)) : nullptr;   // bitmap, then the null count behind it
    out->buffers[1] = hv;                                     // owned by `out` from here on: an early return leaks nothing
    out->buffers[0] = hb;
    if (!hv || (nan_is_null && n && !hb)) return fail(POLS_ERR_INVALID, "out of host memory");
    unsigned long long *hn = nullptr;
    auto queued = [&](hipError_t e, const char *what) -> int {   // a failed call behind queued copies: drain the stream, then report
        if (e == hipSuccess) return POLS_OK;
        (void)hipStreamSynchronize(ctx->stream);
        return fail(POLS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    int rc;
    if (n && stride != 1) {                                   // column j of a row-major table -> contiguous
        T *dcol = reinterpret_cast<T *>(dscratch);
        if ((rc = queued(hipMemcpy2DAsync(dcol, sizeof(T), dvals, sizeof(T) * (size_t)stride, sizeof(T), (size_t)n, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpy2DAsync"))) return rc;
        dvals = dcol;
        dscratch += round256((size_t)n * sizeof(T));
    }
    if (n && (rc = queued(hipMemcpyAsync(hv, dvals, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync"))) return rc;
    if (nan_is_null && n) {
        uint8_t *dbits = reinterpret_cast<uint8_t *>(dscratch);
        unsigned long long *dcnt = reinterpret_cast<unsigned long long *>(dscratch + round256((size_t)nbytes));
        if ((rc = queued(hipMemsetAsync(dcnt, 0, sizeof(*dcnt), ctx->stream), "hipMemsetAsync"))) return rc;
        hipLaunchKernelGGL((arrow_validity_kernel<T>), dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, ctx->stream, dvals, n, dbits, dcnt);
        if ((rc = queued(hipGetLastError(), "arrow_validity_kernel"))) return rc;
        hn = reinterpret_cast<unsigned long long *>(static_cast<char *>(hb) + nbytes);
        if ((rc = queued(hipMemcpyAsync(hb, dbits, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync"))) return rc;
        if ((rc = queued(hipMemcpyAsync(hn, dcnt, sizeof(*hn), hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync"))) return rc;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(POLS_ERR_HIP, "stream synchronisation failed");   // (nothing of ours is queued any more)
    if (hn) {
        unsigned long long nulls;
        std::memcpy(&nulls, hn, sizeof(nulls));
        out->null_count = (int64_t)nulls;
        if (nulls == 0) { std::free(const_cast<void *>(out->buffers[0])); out->buffers[0] = nullptr; }   // no nulls: no bitmap
    }
    return POLS_OK;
}

// A struct array of `n` rows with one primitive field per column.  cols[j] = (device pointer, element stride): the columns of a
// row-major [n x k] table (stride k) or separate contiguous columns (stride 1).  NaN -> null per field (ex.rs:114-143).
template <typename T>
static int export_struct(pols_ctx *ctx, const std::vector<std::pair<const T *, int64_t>> &cols, const std::vector<std::string> &names,
                         int64_t n, bool nan_is_null, const char *struct_name, char *dscratch, ArrowArray *out, ArrowSchema *out_schema) {
    const int k = (int)cols.size();
    ArrowArray arr;
    make_array(&arr, n, 1, k);
    for (int j = 0; j < k; ++j) {
        int rc = export_column<T>(ctx, cols[(size_t)j].first, n, nan_is_null, dscratch, arr.children[j], cols[(size_t)j].second);
        if (rc) { release_array(&arr); return rc; }
    }
    make_schema(out_schema, "+s", struct_name, k);
    for (int j = 0; j < k; ++j) make_schema(out_schema->children[j], sizeof(T) == 4 ? "f" : "g", names[(size_t)j].c_str(), 0);
    *out = arr;                                               // published complete
    return POLS_OK;
}

// names of the coefficient struct's fields: the features' names (their index when unnamed), "const" LAST (least_squares.py:188)
static std::vector<std::string> coefficient_names(const std::vector<ColView> &feat, bool add_intercept) {
    std::vector<std::string> names;
    for (size_t j = 0; j < feat.size(); ++j) names.push_back(feat[j].name && feat[j].name[0] ? std::string(feat[j].name) : std::to_string(j));
    if (add_intercept) names.push_back("const");
    return names;
}

// ---- the inputs of one call on the device
template <typename T>
struct Ingested {
    T *y = nullptr, *w = nullptr;
    std::vector<const void *> x;
    std::vector<T *> targets;     // multi-target: the fields of the target struct
    char *free_area = nullptr;    // what is left of the scratch slot behind the columns (outputs, export scratch)
    size_t free_bytes = 0;
    bool null_free = false;
};

// Stages target (or the fields of a target struct), features and weights as contiguous device columns of T in scratch slot 12,
// leaving `extra` bytes behind them.  Null -> NaN; a null weight acts as the weight 1e-24 (sqrt_w.fill_null(1e-12),
// least_squares.py:193).
template <typename T>
static int ingest_all(pols_ctx *ctx, const std::vector<ColView> &targets, const std::vector<ColView> &feat, const ColView *weights,
                      int64_t n_rows, size_t extra, Ingested<T> *in, T feature_fill = std::numeric_limits<T>::quiet_NaN()) {
    const size_t colb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1));
    size_t raw_cap = 256;
    for (const auto &v : targets) raw_cap = std::max(raw_cap, v.raw_bytes());
    for (const auto &v : feat) raw_cap = std::max(raw_cap, v.raw_bytes());
    if (weights) raw_cap = std::max(raw_cap, weights->raw_bytes());
    const size_t n_in = targets.size() + feat.size() + (weights ? 1 : 0);
    void *base = nullptr;
    int rc = ensure_scratch(ctx, 12, colb * n_in + raw_cap + extra, &base);
    if (rc) return rc;
    char *q = static_cast<char *>(base);
    char *raw = q + colb * n_in;
    const T nan = std::numeric_limits<T>::quiet_NaN();
    in->null_free = true;
    for (const auto &v : targets) {
        T *d = reinterpret_cast<T *>(q); q += colb;
        if ((rc = ingest_view<T>(ctx, v, raw, nan, d))) return rc;
        in->targets.push_back(d);
        in->null_free = in->null_free && v.no_nulls();
    }
    in->y = in->targets.empty() ? nullptr : in->targets[0];
    for (const auto &v : feat) {
        T *d = reinterpret_cast<T *>(q); q += colb;
        if ((rc = ingest_view<T>(ctx, v, raw, feature_fill, d))) return rc;
        in->x.push_back(d);
        in->null_free = in->null_free && v.no_nulls();
    }
    if (weights) {
        in->w = reinterpret_cast<T *>(q); q += colb;
        if ((rc = ingest_view<T>(ctx, *weights, raw, (T)1e-24, in->w))) return rc;
        in->null_free = in->null_free && weights->no_nulls();
    }
    in->free_area = raw + raw_cap;
    in->free_bytes = extra;
    return POLS_OK;
}

template <typename T>
static void fill_batch(pols_batch *b, const Ingested<T> &in, int64_t n_rows, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept) {
    std::memset(b, 0, sizeof(*b));
    b->dtype = sizeof(T) == 4 ? POLS_F32 : POLS_F64;
    b->mem = POLS_MEM_DEVICE;
    b->n_rows = n_rows; b->n_groups = n_groups; b->group_offsets = group_offsets;
    b->n_features = (int32_t)in.x.size(); b->y = in.y; b->x_cols = in.x.data(); b->weights = in.w; b->add_intercept = add_intercept;
    b->null_free = in.null_free ? 1 : 0;                      // Arrow carries the null counts: the policy-free kernels for free
}

// common argument checks + views of target / features / weights; *all_f32: every input is Float32 (the f32 build-side mode;
// anything else computes in the reference's Float64, ex.rs:33,47,80)
static int view_inputs(const pols_arrow_column *features, int32_t n_features, const pols_arrow_column *weights, int32_t add_intercept,
                       int max_features, int64_t n_rows, std::vector<ColView> *feat, ColView *wv, bool *all_f32) {
    if (n_features < 1) return fail(POLS_ERR_INVALID, "must pass at least 2 series");   // ex.rs:72
    if (n_features + (add_intercept ? 1 : 0) > max_features) return fail(POLS_ERR_UNSUPPORTED, "%d features", n_features);
    if (!features) return fail(POLS_ERR_INVALID, "NULL argument");
    feat->assign((size_t)n_features, ColView());
    int rc;
    for (int j = 0; j < n_features; ++j) {
        if ((rc = view_primitive(&features[j], "feature", &(*feat)[(size_t)j]))) return rc;
        if ((*feat)[(size_t)j].rows() != n_rows) return fail(POLS_ERR_INVALID, "all input series passed must be of equal length");   // ex.rs:96-100
        *all_f32 = *all_f32 && (*feat)[(size_t)j].type.code == 'f';
    }
    if (weights) {
        if ((rc = view_primitive(weights, "sample_weights", wv))) return rc;
        if (wv->rows() != n_rows) return fail(POLS_ERR_INVALID, "all input series passed must be of equal length");
        *all_f32 = *all_f32 && wv->type.code == 'f';
    }
    return POLS_OK;
}

static size_t export_scratch_bytes(int64_t n, size_t elem) { return round256((size_t)std::max<int64_t>(n, 1) * elem) + round256((size_t)std::max<int64_t>(n, 1) / 8 + 16) + 512; }

// ---- least_squares / least_squares_coefficients (ex.rs:390-446)
template <typename T>
static int arrow_ls(pols_ctx *ctx, const ColView &target, const std::vector<ColView> &feat, const ColView *weights, const int64_t *group_offsets,
                    int64_t n_groups, int32_t add_intercept, const pols_ols_params *p, int32_t mode, ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = target.rows();
    const int kt = (int)feat.size() + (add_intercept ? 1 : 0);
    const size_t outb = mode == POLS_MODE_COEFFICIENTS ? round256(sizeof(T) * (size_t)std::max<int64_t>(n_groups, 1) * kt)
                                                       : round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1));
    Ingested<T> in;
    int rc = ingest_all<T>(ctx, {target}, feat, weights, n_rows, outb + export_scratch_bytes(std::max(n_rows, n_groups), sizeof(T)), &in);
    if (rc) return rc;
    char *dout = in.free_area, *dscr = dout + outb;
    pols_batch b;
    fill_batch(&b, in, n_rows, group_offsets, n_groups, add_intercept);
    pols_out o;
    std::memset(&o, 0, sizeof(o));
    if (mode == POLS_MODE_COEFFICIENTS) o.coef = dout;
    else if (mode == POLS_MODE_PREDICTIONS) o.pred = dout;
    else o.resid = dout;
    if ((rc = pols_least_squares(ctx, &b, p, &o))) return rc;
    if (mode != POLS_MODE_COEFFICIENTS) {
        // predictions: nulls only where the "drop" policy masks the rows it left out of the fit (mask_predictions, ex.rs:145-158,
        // :409-417) -- those rows carry NaN; residuals inherit the target's nulls the same way.  Under every other policy a NaN
        // stays a NaN value (the reference's fill_null_with_values(NaN) data flows straight through, ex.rs:84-86).
        const bool mask = p->null_policy == POLS_NULL_DROP || mode == POLS_MODE_RESIDUALS;
        ArrowArray arr;
        if ((rc = export_column<T>(ctx, reinterpret_cast<const T *>(dout), n_rows, mask, dscr, &arr))) { release_array(&arr); return rc; }
        make_schema(out_schema, sizeof(T) == 4 ? "f" : "g", target.name, 0);   // named after the target (ex.rs:404)
        *out = arr;
        return POLS_OK;
    }
    // coefficients: a struct with one field per feature, one row per group, NaN -> null (ex.rs:114-143)
    std::vector<std::pair<const T *, int64_t>> cols;
    for (int j = 0; j < kt; ++j) cols.push_back({reinterpret_cast<const T *>(dout) + j, (int64_t)kt});
    return export_struct<T>(ctx, cols, coefficient_names(feat, add_intercept != 0), n_groups, true, "coefficients", dscr, out, out_schema);
}

// ---- recursive_least_squares[_coefficients] / rolling_least_squares[_coefficients] (ex.rs:593-701): one coefficient ROW per input row
template <typename T, typename Params, typename Entry>
static int arrow_dynamic(pols_ctx *ctx, const ColView &target, const std::vector<ColView> &feat, const ColView *weights,
                         const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept, const Params *p, Entry entry, int32_t mode,
                         ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = target.rows();
    const int kt = (int)feat.size() + (add_intercept ? 1 : 0);
    const size_t outb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1) * (mode == POLS_MODE_COEFFICIENTS ? kt : 1));
    Ingested<T> in;
    int rc = ingest_all<T>(ctx, {target}, feat, weights, n_rows, outb + export_scratch_bytes(n_rows, sizeof(T)), &in);
    if (rc) return rc;
    char *dout = in.free_area, *dscr = dout + outb;
    pols_batch b;
    fill_batch(&b, in, n_rows, group_offsets, n_groups, add_intercept);
    pols_out o;
    std::memset(&o, 0, sizeof(o));
    if (mode == POLS_MODE_COEFFICIENTS) o.coef = dout; else o.pred = dout;
    if ((rc = entry(ctx, &b, p, &o))) return rc;
    if (mode == POLS_MODE_COEFFICIENTS) {
        std::vector<std::pair<const T *, int64_t>> cols;
        for (int j = 0; j < kt; ++j) cols.push_back({reinterpret_cast<const T *>(dout) + j, (int64_t)kt});
        return export_struct<T>(ctx, cols, coefficient_names(feat, add_intercept != 0), n_rows, true, "coefficients", dscr, out, out_schema);
    }
    // make_predictions with the validity mask (ex.rs:640-645, 695-700): masked rows are nulls; rows before the rolling warm-up carry
    // NaN coefficients -> NaN predictions, which the Python layer turns into nulls as well (least_squares.py:407-408)
    ArrowArray arr;
    if ((rc = export_column<T>(ctx, reinterpret_cast<const T *>(dout), n_rows, true, dscr, &arr))) { release_array(&arr); return rc; }
    make_schema(out_schema, sizeof(T) == 4 ? "f" : "g", target.name, 0);
    *out = arr;
    return POLS_OK;
}

// ---- multi_target_least_squares (ex.rs:511-591): target STRUCT in, struct of predictions (same field names) out
template <typename T>
static int arrow_multi_target(pols_ctx *ctx, const std::vector<ColView> &targets, const std::vector<ColView> &feat, const ColView *weights,
                              const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept, const pols_ols_params *p,
                              ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = targets[0].rows();
    const int m = (int)targets.size();
    const size_t colb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1));
    Ingested<T> in;
    int rc = ingest_all<T>(ctx, targets, feat, weights, n_rows, colb * (size_t)m + export_scratch_bytes(n_rows, sizeof(T)), &in);
    if (rc) return rc;
    char *dout = in.free_area, *dscr = dout + colb * (size_t)m;
    pols_batch b;
    fill_batch(&b, in, n_rows, group_offsets, n_groups, add_intercept);
    std::vector<const void *> yc(in.targets.begin(), in.targets.end());
    std::vector<void *> pc;
    std::vector<std::pair<const T *, int64_t>> cols;
    std::vector<std::string> names;
    for (int t = 0; t < m; ++t) {
        pc.push_back(dout + colb * (size_t)t);
        cols.push_back({reinterpret_cast<const T *>(pc.back()), 1});
        names.push_back(targets[(size_t)t].name);
    }
    if ((rc = pols_multi_target_least_squares(ctx, &b, yc.data(), m, p, pc.data(), nullptr, nullptr))) return rc;
    // convert_array_to_struct_series (ex.rs:114-143): NaN -> null, which is also how the "drop" mask of :575-583 reaches the caller
    return export_struct<T>(ctx, cols, names, n_rows, true, "predictions", dscr, out, out_schema);
}

// ---- predict (ex.rs:706-741): coefficients STRUCT (one row per input row) x features -> predictions
template <typename T>
static int arrow_predict(pols_ctx *ctx, const std::vector<ColView> &coefs, const std::vector<ColView> &feat, int32_t add_intercept,
                         int32_t null_policy, const char *name, ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = coefs[0].rows();
    const int kt = (int)coefs.size();
    const size_t tabb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1) * kt), colb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1));
    size_t raw_coef = 256;
    for (const auto &v : coefs) raw_coef = std::max(raw_coef, v.raw_bytes());
    Ingested<T> in;
    // construct_features_array(&inputs[1..], null_policy != Ignore) (:725): nulls become 0 unless "ignore"; under "drop" the rows with
    // a null anywhere are masked (:732-738) -- here: they keep NaN, which the export turns into nulls
    const T fill = (null_policy == POLS_NULL_IGNORE || null_policy == POLS_NULL_DROP) ? std::numeric_limits<T>::quiet_NaN() : T(0);
    int rc = ingest_all<T>(ctx, {}, feat, nullptr, n_rows, tabb + raw_coef + colb + export_scratch_bytes(n_rows, sizeof(T)), &in, fill);
    if (rc) return rc;
    T *dtab = reinterpret_cast<T *>(in.free_area);
    char *raw = in.free_area + tabb;
    T *dpred = reinterpret_cast<T *>(raw + raw_coef);
    char *dscr = reinterpret_cast<char *>(dpred) + colb;
    for (int j = 0; j < kt; ++j)                              // field j -> column j of the row-major [n_rows x kt] table pols_predict takes
        if ((rc = ingest_view<T>(ctx, coefs[(size_t)j], raw, std::numeric_limits<T>::quiet_NaN(), dtab + j, kt))) return rc;
    const int64_t one[2] = {0, n_rows};
    pols_batch b;
    fill_batch(&b, in, n_rows, one, 1, add_intercept);
    b.y = in.x[0];                                            // predict has no target; the batch checks want a column
    if ((rc = pols_predict(ctx, &b, dtab, n_rows, dpred))) return rc;
    ArrowArray arr;
    if ((rc = export_column<T>(ctx, dpred, n_rows, null_policy == POLS_NULL_DROP, dscr, &arr))) { release_array(&arr); return rc; }
    make_schema(out_schema, sizeof(T) == 4 ? "f" : "g", name, 0);
    *out = arr;
    return POLS_OK;
}

// ---- least_squares_statistics (ex.rs:448-509): ONE struct row per group:
//   {r2, mae, mse: f64; feature_names: list<str>; coefficients, standard_errors, t_values, p_values: list<f64>}
// Lists are large lists ("+L", 64-bit offsets) of large strings ("U") / Float64 -- Polars' own physical types; values are always
// Float64 like the reference's struct fields, and a NaN stays a NaN value (Series::new on f64 makes no nulls).
static int list_f64(ArrowArray *a, ArrowSchema *s, const char *name, const double *host_table, int64_t n, int k) {
    make_array(a, n, 2, 1);
    int64_t *offs = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * (size_t)(n + 1)));
    double *vals = static_cast<double *>(std::malloc(std::max<size_t>(1, sizeof(double) * (size_t)n * k)));
    if (!offs || !vals) { std::free(offs); std::free(vals); return fail(POLS_ERR_INVALID, "out of host memory"); }
    for (int64_t i = 0; i <= n; ++i) offs[i] = i * k;
    std::memcpy(vals, host_table, sizeof(double) * (size_t)n * k);
    a->buffers[1] = offs;
    make_array(a->children[0], n * k, 2, 0);
    a->children[0]->buffers[1] = vals;
    make_schema(s, "+L", name, 1);
    make_schema(s->children[0], "g", "item", 0);
    return POLS_OK;
}

static int list_names(ArrowArray *a, ArrowSchema *s, const char *name, const std::vector<std::string> &names, int64_t n) {
    const int k = (int)names.size();
    make_array(a, n, 2, 1);
    size_t per = 0;
    for (const auto &nm : names) per += nm.size();
    int64_t *offs = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * (size_t)(n + 1)));
    int64_t *soffs = static_cast<int64_t *>(std::malloc(sizeof(int64_t) * (size_t)(n * k + 1)));
    char *data = static_cast<char *>(std::malloc(std::max<size_t>(1, per * (size_t)n)));
    if (!offs || !soffs || !data) { std::free(offs); std::free(soffs); std::free(data); return fail(POLS_ERR_INVALID, "out of host memory"); }
    int64_t pos = 0;
    for (int64_t i = 0; i < n; ++i) {
        offs[i] = i * k;
        for (int j = 0; j < k; ++j) {
            soffs[i * k + j] = pos;
            std::memcpy(data + pos, names[(size_t)j].data(), names[(size_t)j].size());
            pos += (int64_t)names[(size_t)j].size();
        }
    }
    offs[n] = n * k; soffs[n * k] = pos;
    a->buffers[1] = offs;
    make_array(a->children[0], n * k, 3, 0);
    a->children[0]->buffers[1] = soffs;
    a->children[0]->buffers[2] = data;
    make_schema(s, "+L", name, 1);
    make_schema(s->children[0], "U", "item", 0);
    return POLS_OK;
}

template <typename T>
static int arrow_statistics(pols_ctx *ctx, const ColView &target, const std::vector<ColView> &feat, const ColView *weights,
                            const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept, const pols_ols_params *p,
                            ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = target.rows(), G = std::max<int64_t>(n_groups, 1);
    const int kt = (int)feat.size() + (add_intercept ? 1 : 0);
    const size_t coefb = round256(sizeof(T) * (size_t)G * kt), sb = round256(sizeof(double) * (size_t)G), tb = round256(sizeof(double) * (size_t)G * kt);
    const size_t statb = round256(sizeof(int32_t) * (size_t)G);
    Ingested<T> in;
    int rc = ingest_all<T>(ctx, {target}, feat, weights, n_rows, coefb + 3 * sb + 3 * tb + statb, &in);
    if (rc) return rc;
    char *q = in.free_area;
    void *dcoef = q; q += coefb;
    pols_stats_out so;
    so.r2 = reinterpret_cast<double *>(q); q += sb; so.mae = reinterpret_cast<double *>(q); q += sb; so.mse = reinterpret_cast<double *>(q); q += sb;
    so.std_err = reinterpret_cast<double *>(q); q += tb; so.t_values = reinterpret_cast<double *>(q); q += tb; so.p_values = reinterpret_cast<double *>(q); q += tb;
    int32_t *dstat = reinterpret_cast<int32_t *>(q);
    pols_batch b;
    fill_batch(&b, in, n_rows, group_offsets, n_groups, add_intercept);
    pols_out o;
    std::memset(&o, 0, sizeof(o));
    o.coef = dcoef; o.status = dstat;
    if ((rc = pols_least_squares_statistics(ctx, &b, p, &o, &so))) return rc;
    // everything is small (per group): home in one go
    std::vector<T> hcoef((size_t)n_groups * kt);
    std::vector<double> h3((size_t)n_groups * 3), ht((size_t)n_groups * kt * 3), hc64((size_t)n_groups * kt);
    std::vector<int32_t> hstat((size_t)n_groups);
    if (n_groups) {
        POLS_HIP(hipMemcpyAsync(hcoef.data(), dcoef, sizeof(T) * hcoef.size(), hipMemcpyDeviceToHost, ctx->stream));
        const double *src3[3] = {so.r2, so.mae, so.mse}, *srct[3] = {so.std_err, so.t_values, so.p_values};
        for (int i = 0; i < 3; ++i) {
            POLS_HIP(hipMemcpyAsync(h3.data() + (size_t)i * n_groups, src3[i], sizeof(double) * (size_t)n_groups, hipMemcpyDeviceToHost, ctx->stream));
            POLS_HIP(hipMemcpyAsync(ht.data() + (size_t)i * n_groups * kt, srct[i], sizeof(double) * (size_t)n_groups * kt, hipMemcpyDeviceToHost, ctx->stream));
        }
        POLS_HIP(hipMemcpyAsync(hstat.data(), dstat, sizeof(int32_t) * hstat.size(), hipMemcpyDeviceToHost, ctx->stream));
        POLS_HIP(hipStreamSynchronize(ctx->stream));
    }
    for (int64_t g = 0; g < n_groups; ++g)                    // the reference asserts df > 0 and panics the whole query (statistics.rs:131-134)
        if (hstat[(size_t)g] == POLS_GROUP_BAD_DOF) return fail(POLS_ERR_PANIC, "Degrees of freedom <= 0. Cannot compute standard errors.");
    for (size_t i = 0; i < hcoef.size(); ++i) hc64[i] = (double)hcoef[i];
    ArrowArray arr;
    make_array(&arr, n_groups, 1, 8);
    make_schema(out_schema, "+s", "statistics", 8);
    static const char *const scalar_names[3] = {"r2", "mae", "mse"};
    for (int i = 0; i < 3 && !rc; ++i) {
        ArrowArray *c = arr.children[i];
        make_array(c, n_groups, 2, 0);
        void *v = std::malloc(std::max<size_t>(1, sizeof(double) * (size_t)n_groups));
        if (!v) { rc = fail(POLS_ERR_INVALID, "out of host memory"); break; }
        std::memcpy(v, h3.data() + (size_t)i * n_groups, sizeof(double) * (size_t)n_groups);
        c->buffers[1] = v;
        make_schema(out_schema->children[i], "g", scalar_names[i], 0);
    }
    if (!rc) rc = list_names(arr.children[3], out_schema->children[3], "feature_names", coefficient_names(feat, add_intercept != 0), n_groups);
    if (!rc) rc = list_f64(arr.children[4], out_schema->children[4], "coefficients", hc64.data(), n_groups, kt);
    static const char *const list_names3[3] = {"standard_errors", "t_values", "p_values"};
    for (int i = 0; i < 3 && !rc; ++i)
        rc = list_f64(arr.children[5 + i], out_schema->children[5 + i], list_names3[i], ht.data() + (size_t)i * n_groups * kt, n_groups, kt);
    if (rc) { release_array(&arr); release_schema(out_schema); return rc; }
    *out = arr;
    return POLS_OK;
}

}  // namespace pols

using namespace pols;

namespace {
struct CommonViews {
    ColView target, wv;
    std::vector<ColView> feat;
    bool all_f32 = true;
    int64_t n_rows = 0;
    int64_t one[2] = {0, 0};
};

// target (primitive) + features + weights + the single-group default of a per-group plugin call
int common_views(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                 const pols_arrow_column *weights, const int64_t **group_offsets, int64_t *n_groups, int32_t add_intercept, int max_features,
                 const void *p, struct ArrowArray *out, struct ArrowSchema *out_schema, CommonViews *cv) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    if (!target || !features || !p || !out || !out_schema) return fail(POLS_ERR_INVALID, "NULL argument");
    int rc = view_primitive(target, "target", &cv->target);
    if (rc) return rc;
    cv->n_rows = cv->target.rows();
    cv->all_f32 = cv->target.type.code == 'f';
    if ((rc = view_inputs(features, n_features, weights, add_intercept, max_features, cv->n_rows, &cv->feat, &cv->wv, &cv->all_f32))) return rc;
    cv->one[1] = cv->n_rows;
    if (!*group_offsets) { *group_offsets = cv->one; *n_groups = 1; }   // the call a plugin receives per group: one group, all rows
    POLS_HIP(hipSetDevice(ctx->device));
    return POLS_OK;
}
}  // namespace

extern "C" {

int pols_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                             const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                             const pols_ols_params *p, int32_t mode, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    if (mode < POLS_MODE_PREDICTIONS || mode > POLS_MODE_COEFFICIENTS) return fail(POLS_ERR_INVALID, "mode %d", mode);
    CommonViews cv;
    int rc = common_views(ctx, target, features, n_features, weights, &group_offsets, &n_groups, add_intercept, POLS_MAX_FEATURES_STATIC, p, out, out_schema, &cv);
    if (rc) return rc;
    const ColView *w = weights ? &cv.wv : nullptr;
    // compute dtype: f32 only when EVERY input is Float32 (a build-side mode; the reference always computes in f64, ex.rs:33,47,80)
    if (cv.all_f32) return arrow_ls<float>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, mode, out, out_schema);
    return arrow_ls<double>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, mode, out, out_schema);
}

int pols_least_squares_statistics_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                                        const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                                        const pols_ols_params *p, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    CommonViews cv;
    int rc = common_views(ctx, target, features, n_features, weights, &group_offsets, &n_groups, add_intercept, POLS_MAX_FEATURES_STATISTICS, p, out, out_schema, &cv);
    if (rc) return rc;
    const ColView *w = weights ? &cv.wv : nullptr;
    if (cv.all_f32) return arrow_statistics<float>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, out, out_schema);
    return arrow_statistics<double>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, out, out_schema);
}

int pols_recursive_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                                       const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                                       const pols_rls_params *p, int32_t mode, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    if (mode != POLS_MODE_PREDICTIONS && mode != POLS_MODE_COEFFICIENTS) return fail(POLS_ERR_INVALID, "mode %d (predictions or coefficients)", mode);
    CommonViews cv;
    int rc = common_views(ctx, target, features, n_features, weights, &group_offsets, &n_groups, add_intercept, POLS_MAX_FEATURES_DYNAMIC, p, out, out_schema, &cv);
    if (rc) return rc;
    const ColView *w = weights ? &cv.wv : nullptr;
    if (cv.all_f32) return arrow_dynamic<float>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, pols_recursive_least_squares, mode, out, out_schema);
    return arrow_dynamic<double>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, pols_recursive_least_squares, mode, out, out_schema);
}

int pols_rolling_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                                     const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                                     const pols_rolling_params *p, int32_t mode, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    if (mode != POLS_MODE_PREDICTIONS && mode != POLS_MODE_COEFFICIENTS) return fail(POLS_ERR_INVALID, "mode %d (predictions or coefficients)", mode);
    CommonViews cv;
    int rc = common_views(ctx, target, features, n_features, weights, &group_offsets, &n_groups, add_intercept, POLS_MAX_FEATURES_DYNAMIC, p, out, out_schema, &cv);
    if (rc) return rc;
    const ColView *w = weights ? &cv.wv : nullptr;
    if (cv.all_f32) return arrow_dynamic<float>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, pols_rolling_least_squares, mode, out, out_schema);
    return arrow_dynamic<double>(ctx, cv.target, cv.feat, w, group_offsets, n_groups, add_intercept, p, pols_rolling_least_squares, mode, out, out_schema);
}

int pols_multi_target_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *targets, const pols_arrow_column *features, int32_t n_features,
                                          const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups, int32_t add_intercept,
                                          const pols_ols_params *p, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    if (!targets || !features || !p || !out || !out_schema) return fail(POLS_ERR_INVALID, "NULL argument");
    std::vector<ColView> tv, feat;
    ColView wv;
    int rc = view_struct(targets, "the first series in a multi-target regression", &tv);   // ex.rs:513-517
    if (rc) return rc;
    const int64_t n_rows = tv[0].rows();
    bool all_f32 = true;
    for (const auto &v : tv) all_f32 = all_f32 && v.type.code == 'f';
    if ((rc = view_inputs(features, n_features, weights, add_intercept, POLS_MAX_FEATURES_STATIC - (int)tv.size(), n_rows, &feat, &wv, &all_f32))) return rc;
    const int64_t one[2] = {0, n_rows};
    if (!group_offsets) { group_offsets = one; n_groups = 1; }
    POLS_HIP(hipSetDevice(ctx->device));
    const ColView *w = weights ? &wv : nullptr;
    if (all_f32) return arrow_multi_target<float>(ctx, tv, feat, w, group_offsets, n_groups, add_intercept, p, out, out_schema);
    return arrow_multi_target<double>(ctx, tv, feat, w, group_offsets, n_groups, add_intercept, p, out, out_schema);
}

int pols_predict_arrow(pols_ctx *ctx, const pols_arrow_column *coefficients, const pols_arrow_column *features, int32_t n_features,
                       int32_t add_intercept, int32_t null_policy, const char *name, struct ArrowArray *out, struct ArrowSchema *out_schema) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    if (!coefficients || !features || !out || !out_schema) return fail(POLS_ERR_INVALID, "NULL argument");
    if (null_policy != POLS_NULL_IGNORE && null_policy != POLS_NULL_ZERO && null_policy != POLS_NULL_DROP)
        return fail(POLS_ERR_PANIC, "'null_policy' must be one of {drop, ignore, zero}");   // least_squares.py:474
    std::vector<ColView> cv, feat;
    ColView wv;
    int rc = view_struct(coefficients, "the first input series to predict function", &cv);   // ex.rs:712-714
    if (rc) return rc;
    const int64_t n_rows = cv[0].rows();
    bool all_f32 = true;
    for (const auto &v : cv) all_f32 = all_f32 && v.type.code == 'f';
    if ((rc = view_inputs(features, n_features, nullptr, add_intercept, POLS_MAX_FEATURES_STATIC, n_rows, &feat, &wv, &all_f32))) return rc;
    if ((int)cv.size() != n_features + (add_intercept ? 1 : 0))
        return fail(POLS_ERR_PANIC, "number of coefficients must match number of features!");   // ex.rs:718-722
    POLS_HIP(hipSetDevice(ctx->device));
    const char *nm = (name && name[0]) ? name : "predictions";   // .alias(name or "predictions"), least_squares.py:491
    if (all_f32) return arrow_predict<float>(ctx, cv, feat, add_intercept, null_policy, nm, out, out_schema);
    return arrow_predict<double>(ctx, cv, feat, add_intercept, null_policy, nm, out, out_schema);
}

}  // extern "C"
