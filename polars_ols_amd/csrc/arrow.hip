// arrow.hip -- the plugin-side marshalling of the static path behind the C-ABI: Arrow C Data Interface in, Arrow C Data
// Interface out (pols_least_squares_arrow, include/pols_mi355x.h).
//
// Replaces what src/expressions.rs does around the solver for `least_squares` / `least_squares_coefficients` (:390-446):
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
#include <cstdlib>
#include <limits>

#include "common.hpp"

namespace pols {

template <typename S, typename T>
__global__ void __launch_bounds__(256) arrow_ingest_kernel(const S *src, const uint8_t *bits, int bit_offset, int64_t n, T fill, T *dst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool valid = true;
    if (bits) { const int64_t b = bit_offset + i; valid = (bits[b >> 3] >> (b & 7)) & 1; }
    dst[i] = valid ? (T)src[i] : fill;
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
static int ingest_launch(pols_ctx *ctx, char code, const void *src, const uint8_t *bits, int bit_offset, int64_t n, T fill, T *dst) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (n == 0) return POLS_OK;
#define INGEST(S) hipLaunchKernelGGL((arrow_ingest_kernel<S, T>), dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const S *>(src), bits, bit_offset, n, fill, dst)
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

static int64_t column_rows(const pols_arrow_column *c) {
    int64_t n = 0;
    for (int i = 0; i < c->n_chunks; ++i) n += c->chunks[i]->length;
    return n;
}

static int check_column(const pols_arrow_column *c, const char *what, ArrowType *t) {
    if (!c || !c->schema || (!c->chunks && c->n_chunks) || c->n_chunks < 0) return fail(POLS_ERR_INVALID, "%s: NULL schema / chunks", what);
    if (!arrow_type(c->schema->format, t))
        return fail(POLS_ERR_UNSUPPORTED, "%s: arrow format '%s' (numeric primitives only: the reference casts to Float64, src/expressions.rs:80)",
                    what, c->schema->format ? c->schema->format : "(null)");
    for (int i = 0; i < c->n_chunks; ++i) {
        const ArrowArray *a = c->chunks[i];
        if (!a || a->length < 0 || a->offset < 0 || a->n_buffers < 2 || !a->buffers || (a->length && !a->buffers[1]))
            return fail(POLS_ERR_INVALID, "%s: chunk %d is not a primitive array (2 buffers)", what, i);
    }
    return POLS_OK;
}

// One column: every chunk host -> device (values, bitmap bytes), cast + null-fill into dst[0 .. n_rows).  `raw` is a device
// staging area of raw_cap bytes re-used chunk after chunk (stream order keeps that safe).
template <typename T>
static int ingest_column(pols_ctx *ctx, const pols_arrow_column *c, const ArrowType &t, char *raw, T fill, T *dst) {
    int64_t row = 0;
    for (int i = 0; i < c->n_chunks; ++i) {
        const ArrowArray *a = c->chunks[i];
        const int64_t n = a->length;
        if (n == 0) continue;
        const char *values = static_cast<const char *>(a->buffers[1]) + (size_t)a->offset * t.bytes;
        const uint8_t *bitmap = (a->null_count != 0) ? static_cast<const uint8_t *>(a->buffers[0]) : nullptr;
        const bool same = (t.code == (sizeof(T) == 8 ? 'g' : 'f'));
        if (same && !bitmap) {                                // already the compute dtype, no nulls: straight into place
            POLS_HIP(hipMemcpyAsync(dst + row, values, (size_t)n * t.bytes, hipMemcpyHostToDevice, ctx->stream));
            row += n;
            continue;
        }
        const size_t vbytes = (size_t)n * t.bytes;
        char *dvals = raw;
        uint8_t *dbits = nullptr;
        int bit_offset = 0;
        POLS_HIP(hipMemcpyAsync(dvals, values, vbytes, hipMemcpyHostToDevice, ctx->stream));
        if (bitmap) {
            const int64_t b0 = a->offset >> 3, b1 = (a->offset + n + 7) >> 3;
            dbits = reinterpret_cast<uint8_t *>(raw + round256(vbytes));
            bit_offset = (int)(a->offset & 7);
            POLS_HIP(hipMemcpyAsync(dbits, bitmap + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, ctx->stream));
        }
        int rc = ingest_launch<T>(ctx, t.code, dvals, dbits, bit_offset, n, fill, dst + row);
        if (rc) return rc;
        row += n;
    }
    return POLS_OK;
}

// ---- output arrays: malloc'd buffers handed over with release callbacks, as the interface prescribes
static void release_array(ArrowArray *a) {
    if (!a || !a->release) return;
    for (int64_t i = 0; i < a->n_children; ++i) {
        if (a->children[i]->release) a->children[i]->release(a->children[i]);
        std::free(a->children[i]);
    }
    std::free(a->children);
    for (int64_t i = 0; i < a->n_buffers; ++i) std::free(const_cast<void *>(a->buffers[i]));
    std::free(a->buffers);
    a->release = nullptr;
}

static void release_schema(ArrowSchema *s) {
    if (!s || !s->release) return;
    for (int64_t i = 0; i < s->n_children; ++i) {
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

static void make_primitive(ArrowArray *a, void *values, void *validity, int64_t n, int64_t null_count) {
    std::memset(a, 0, sizeof(*a));
    a->length = n;
    a->null_count = null_count;
    a->n_buffers = 2;
    a->buffers = static_cast<const void **>(std::calloc(2, sizeof(void *)));
    a->buffers[0] = validity;
    a->buffers[1] = values;
    a->release = release_array;
}

// device column of n values -> host primitive array; NaN -> null when `nan_is_null` (validity bitmap + null count made on the device)
template <typename T>
static int export_column(pols_ctx *ctx, const T *dvals, int64_t n, bool nan_is_null, char *dscratch, ArrowArray *out) {
    void *hv = std::malloc(std::max<size_t>(1, (size_t)n * sizeof(T)));
    void *hb = nullptr;
    unsigned long long nulls = 0;
    if (!hv) return fail(POLS_ERR_INVALID, "out of host memory");
    if (n) POLS_HIP(hipMemcpyAsync(hv, dvals, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    if (nan_is_null && n) {
        const int64_t nbytes = (n + 7) / 8;
        uint8_t *dbits = reinterpret_cast<uint8_t *>(dscratch);
        unsigned long long *dcnt = reinterpret_cast<unsigned long long *>(dscratch + round256((size_t)nbytes));
        POLS_HIP(hipMemsetAsync(dcnt, 0, sizeof(*dcnt), ctx->stream));
        hipLaunchKernelGGL((arrow_validity_kernel<T>), dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, ctx->stream, dvals, n, dbits, dcnt);
        POLS_HIP(hipGetLastError());
        hb = std::malloc((size_t)nbytes);
        POLS_HIP(hipMemcpyAsync(hb, dbits, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
        POLS_HIP(hipMemcpyAsync(&nulls, dcnt, sizeof(nulls), hipMemcpyDeviceToHost, ctx->stream));
    }
    POLS_HIP(hipStreamSynchronize(ctx->stream));
    if (hb && nulls == 0) { std::free(hb); hb = nullptr; }     // no nulls: no bitmap (the interface allows a NULL validity buffer)
    make_primitive(out, hv, hb, n, (int64_t)nulls);
    return POLS_OK;
}

template <typename T>
static int arrow_ls(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                    const pols_arrow_column *weights, const ArrowType *types, const int64_t *group_offsets, int64_t n_groups,
                    int32_t add_intercept, const pols_ols_params *p, int32_t mode, ArrowArray *out, ArrowSchema *out_schema) {
    const int64_t n_rows = column_rows(target);
    const int kt = n_features + (add_intercept ? 1 : 0);
    const size_t colb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_rows, 1));
    // raw staging: the largest chunk of any column (values + bitmap bytes)
    size_t raw_cap = 256;
    auto grow = [&](const pols_arrow_column *c, const ArrowType &t) {
        for (int i = 0; i < c->n_chunks; ++i)
            raw_cap = std::max(raw_cap, round256((size_t)c->chunks[i]->length * t.bytes) + round256((size_t)c->chunks[i]->length / 8 + 16));
    };
    grow(target, types[0]);
    for (int j = 0; j < n_features; ++j) grow(&features[j], types[1 + j]);
    if (weights) grow(weights, types[1 + n_features]);
    const int n_in = 1 + n_features + (weights ? 1 : 0);
    const size_t coefb = round256(sizeof(T) * (size_t)std::max<int64_t>(n_groups, 1) * kt);
    const size_t outb = mode == POLS_MODE_COEFFICIENTS ? coefb + round256(sizeof(T) * (size_t)std::max<int64_t>(n_groups, 1)) : colb;
    const size_t bitsb = round256((size_t)std::max<int64_t>(n_rows, n_groups) / 8 + 16) + 256;
    void *base = nullptr;
    int rc = ensure_scratch(ctx, 12, colb * n_in + raw_cap + outb + bitsb, &base);
    if (rc) return rc;
    char *q = static_cast<char *>(base);
    T *dy = reinterpret_cast<T *>(q); q += colb;
    std::vector<const void *> dx((size_t)n_features);
    const T nan = std::numeric_limits<T>::quiet_NaN();
    char *raw = static_cast<char *>(base) + colb * n_in;
    if ((rc = ingest_column<T>(ctx, target, types[0], raw, nan, dy))) return rc;
    for (int j = 0; j < n_features; ++j) {
        T *d = reinterpret_cast<T *>(q); q += colb;
        if ((rc = ingest_column<T>(ctx, &features[j], types[1 + j], raw, nan, d))) return rc;
        dx[(size_t)j] = d;
    }
    T *dw = nullptr;
    if (weights) {
        dw = reinterpret_cast<T *>(q); q += colb;
        // sqrt_w = w.sqrt().fill_null(1e-12) (least_squares.py:193): a null weight acts as the weight 1e-24
        if ((rc = ingest_column<T>(ctx, weights, types[1 + n_features], raw, (T)1e-24, dw))) return rc;
    }
    char *dout = raw + raw_cap;
    char *dbits = dout + outb;

    pols_batch b;
    std::memset(&b, 0, sizeof(b));
    b.dtype = sizeof(T) == 4 ? POLS_F32 : POLS_F64;
    b.mem = POLS_MEM_DEVICE;
    b.n_rows = n_rows; b.n_groups = n_groups; b.group_offsets = group_offsets;
    b.n_features = n_features; b.y = dy; b.x_cols = dx.data(); b.weights = dw; b.add_intercept = add_intercept;
    auto no_nulls = [](const pols_arrow_column *c) {
        for (int i = 0; i < c->n_chunks; ++i) if (c->chunks[i]->null_count != 0) return false;   // (-1 = not computed: unknown)
        return true;
    };
    b.null_free = no_nulls(target) ? 1 : 0;                   // Arrow carries the null counts: the policy-free kernels for free
    for (int j = 0; j < n_features && b.null_free; ++j) b.null_free = no_nulls(&features[j]) ? 1 : 0;
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
        if ((rc = export_column<T>(ctx, reinterpret_cast<const T *>(dout), n_rows, mask, dbits, out))) return rc;
        make_schema(out_schema, sizeof(T) == 4 ? "f" : "g", target->schema->name, 0);   // named after the target (ex.rs:404)
        return POLS_OK;
    }
    // coefficients: a struct with one field per feature (named like the feature, its index when unnamed; the intercept is
    // "const", appended last, least_squares.py:188), one row per group, NaN -> null (ex.rs:114-143)
    T *dcol = reinterpret_cast<T *>(dout + coefb);
    std::memset(out, 0, sizeof(*out));
    out->length = n_groups;
    out->n_buffers = 1;
    out->buffers = static_cast<const void **>(std::calloc(1, sizeof(void *)));
    out->n_children = kt;
    out->children = static_cast<ArrowArray **>(std::calloc((size_t)std::max(kt, 1), sizeof(ArrowArray *)));
    out->release = release_array;
    make_schema(out_schema, "+s", "coefficients", kt);
    for (int j = 0; j < kt; ++j) {
        out->children[j] = static_cast<ArrowArray *>(std::calloc(1, sizeof(ArrowArray)));
        // column j of the [n_groups x kt] table -> contiguous
        POLS_HIP(hipMemcpy2DAsync(dcol, sizeof(T), reinterpret_cast<const T *>(dout) + j, sizeof(T) * (size_t)kt, sizeof(T), (size_t)n_groups,
                                  hipMemcpyDeviceToDevice, ctx->stream));
        if ((rc = export_column<T>(ctx, dcol, n_groups, true, dbits, out->children[j]))) return rc;
        char idx[16];
        std::snprintf(idx, sizeof(idx), "%d", j);
        const char *name = j < n_features ? features[j].schema->name : "const";
        make_schema(out_schema->children[j], sizeof(T) == 4 ? "f" : "g", (name && name[0]) ? name : idx, 0);
    }
    return POLS_OK;
}

}  // namespace pols

using namespace pols;

extern "C" int pols_least_squares_arrow(pols_ctx *ctx, const pols_arrow_column *target, const pols_arrow_column *features, int32_t n_features,
                                        const pols_arrow_column *weights, const int64_t *group_offsets, int64_t n_groups,
                                        int32_t add_intercept, const pols_ols_params *p, int32_t mode, struct ArrowArray *out,
                                        struct ArrowSchema *out_schema) {
    if (!ctx) return fail(POLS_ERR_INVALID, "ctx is NULL");
    if (!target || !features || !p || !out || !out_schema) return fail(POLS_ERR_INVALID, "NULL argument");
    if (n_features < 1) return fail(POLS_ERR_INVALID, "must pass at least 2 series");   // ex.rs:72
    if (n_features + (add_intercept ? 1 : 0) > POLS_MAX_FEATURES_STATIC) return fail(POLS_ERR_UNSUPPORTED, "%d features", n_features);
    if (mode < POLS_MODE_PREDICTIONS || mode > POLS_MODE_COEFFICIENTS) return fail(POLS_ERR_INVALID, "mode %d", mode);
    std::vector<ArrowType> types((size_t)n_features + 2);
    int rc = check_column(target, "target", &types[0]);
    if (rc) return rc;
    const int64_t n_rows = column_rows(target);
    bool all_f32 = types[0].code == 'f';
    for (int j = 0; j < n_features; ++j) {
        if ((rc = check_column(&features[j], "feature", &types[(size_t)1 + j]))) return rc;
        if (column_rows(&features[j]) != n_rows) return fail(POLS_ERR_INVALID, "all input series passed must be of equal length");   // ex.rs:96-100
        all_f32 = all_f32 && types[(size_t)1 + j].code == 'f';
    }
    if (weights) {
        if ((rc = check_column(weights, "sample_weights", &types[(size_t)1 + n_features]))) return rc;
        if (column_rows(weights) != n_rows) return fail(POLS_ERR_INVALID, "all input series passed must be of equal length");
        all_f32 = all_f32 && types[(size_t)1 + n_features].code == 'f';
    }
    const int64_t one[2] = {0, n_rows};
    if (!group_offsets) { group_offsets = one; n_groups = 1; }   // the call a plugin receives per group: one group, all rows
    POLS_HIP(hipSetDevice(ctx->device));
    // compute dtype: f32 only when EVERY input is Float32 (a build-side mode; the reference always computes in f64, ex.rs:33,47,80)
    if (all_f32) return arrow_ls<float>(ctx, target, features, n_features, weights, types.data(), group_offsets, n_groups, add_intercept, p, mode, out, out_schema);
    return arrow_ls<double>(ctx, target, features, n_features, weights, types.data(), group_offsets, n_groups, add_intercept, p, mode, out, out_schema);
}
