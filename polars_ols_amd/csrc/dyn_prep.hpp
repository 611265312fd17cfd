// dyn_prep.hpp -- what the reference's Python layer and plugin bodies do AROUND the dynamic solvers (recursive / rolling least
// squares), as three device passes behind the C-ABI instead of a dozen host-side column operations:
//   scan     compute_is_valid_mask (src/expressions.rs:201-228) for the call's null policy from the NaNs (= nulls) of the
//            columns: one validity byte per row + "any row invalid" / "any null in a kept row" flags
//   rewrite  polars_ols/least_squares.py:184-196 (sqrt(w) scaling of target and features, the ones column appended LAST) and the
//            NullPolicy::Zero conversion of the inputs (ex.rs:603, 629, 656, 683: nulls -> 0) in one read + write pass; skipped
//            when there are no weights, no intercept and no nulls (the kernels then read the caller's columns in place)
//   post     predictions *= 1 / sqrt(w) (ls.py:234-235) and the is_valid mask of make_predictions (ex.rs:640-645, 695-700)
#pragma once
#include "common.hpp"

namespace pols {

struct DynPrepArgs {
    const void *y, *w;                   // original target / weights (w may be nullptr)
    const void *const *xtab;             // DEVICE table of k_user feature column pointers
    int32_t k_user, add_intercept, null_policy;
    int64_t n_rows;
    uint8_t *valid_out;                  // n_rows bytes
    int32_t *flags;                      // [0] rows left out of the fit, [1] nulls inside kept rows (either > 0 = "some")
    void *y_out;                         // rewritten target
    void *const *xout;                   // DEVICE table of k_user + add_intercept rewritten column pointers
    void *sw_out;                        // sqrt(w) per row (only with weights)
    void *pred;                          // post: predictions in place
    const uint8_t *valid_post;           // post: validity bytes or nullptr (no masking)
};

// handle_nulls for the entries that work on FILTERED rows (mode="statistics", src/expressions.rs:469-471; the multi-target fit,
// :539-548): rows the policy drops leave (stable compaction inside every group), surviving nulls become 0 where the policy says
// so.  Two passes, one workgroup per group: count (validity bytes + rows kept per group), then -- once the host has turned the
// counts into the compacted offsets it needs anyway -- scatter.
struct CompactArgs {
    const void *const *in;       // DEVICE table of n_cols column pointers; the columns that decide validity come first
    void *const *out;            // DEVICE table of n_cols compacted column pointers
    int32_t n_cols, n_mask;      // columns [0, n_mask) are checked for nulls (NaN) when `drop`
    int32_t w_col;               // index of the sample-weights column (null -> 1e-24, least_squares.py:193) or -1
    int32_t drop, zero_fill;
    const uint8_t *valid_in;     // caller's validity bytes (ANDed in) or nullptr
    const int64_t *offs;         // DEVICE group offsets of the input
    const int64_t *offs_out;     // DEVICE group offsets of the output (scatter pass)
    unsigned long long *counts;  // rows kept per group (count pass; zeroed by the launcher)
    uint8_t *vbytes;             // n_rows validity bytes: written by the count pass, read by the scatter pass
    int64_t n_groups;
};
int compact_count_launch(pols_ctx *ctx, int dtype, const CompactArgs &a);
int compact_scatter_launch(pols_ctx *ctx, int dtype, const CompactArgs &a);

// Multi-target predictions under the drop policies (src/expressions.rs:566-585): the coefficients were fitted on the filtered rows;
// every ORIGINAL row is predicted from its zero-filled features and its group's coefficients, "drop" masks the rows left out.
struct MtPredictArgs {
    const void *const *xtab;     // DEVICE table of k_user feature column pointers (original rows)
    const void *w;               // sample weights or nullptr: (sqrt(w) x) . c * (1 / sqrt(w)) like the reference's arithmetic
    const void *coef;            // n_groups x m x kt, batch dtype
    void *const *ptab;           // DEVICE table of m prediction column pointers
    const uint8_t *vbytes;       // row validity from the compaction pass
    const int64_t *offs;         // DEVICE group offsets (original rows)
    int64_t n_groups;
    int32_t k_user, kt, m, mask_drop;
    int32_t row_blocks;          // workgroups per group (long groups), 1 .. 1024
};
int mt_predict_launch(pols_ctx *ctx, int dtype, const MtPredictArgs &a);

// Rolling OLS under the drop family on a frame WITH nulls (src/least_squares.rs:947-986: the window is a deque of the last `window`
// VALID rows, a row left out repeats the last coefficients) = the null-free problem on the valid rows, then every original row takes
// the coefficients of the last valid row at or before it in its sequence.  Slab-parallel (256 rows per workgroup, any group sizes):
//   count   valid rows per slab                              scan    exclusive prefix over the slabs (one workgroup)
//   groups  compacted offset of every group + what a slab that starts inside a group needs to know about it
//   scatter the valid rows of every column, order kept       [the row-parallel rolling kernel runs on the compacted frame]
//   expand  coefficients forward-filled onto the original rows (whole lines), predictions from the original rows
struct RowCompactArgs {
    const uint8_t *valid;        // n_rows validity bytes (device)
    const uint8_t *start;        // n_rows sequence-start bytes of the ORIGINAL frame (k3c_start_flags)
    const int64_t *offs;         // DEVICE group offsets of the original frame, n_groups + 1
    int64_t n_rows, n_groups, n_slabs;
    uint32_t *slab_cnt;          // n_slabs
    int64_t *slab_base;          // n_slabs + 1: valid rows before the slab; [n_slabs] = all of them
    int64_t *c_offs;             // n_groups + 1: group offsets of the compacted frame
    int64_t *slab_gfirst;        // n_slabs: compacted offset of the group that holds the slab's first row when it started before the slab
    const void *const *in;       // DEVICE table of n_cols column pointers (target first, then the features)
    void *const *out;            // DEVICE table of n_cols compacted column pointers
    int32_t n_cols, k;
    const void *coef_c;          // expand: compacted n_valid x k coefficients
    void *coef, *pred;           // expand: n_rows x k / n_rows (either may be nullptr)
    // mask pass (handle_nulls for the static entries that work on filtered rows): valid[r] = valid_in[r] && no NaN in columns [0, n_mask)
    uint8_t *valid_out;          // written by row_compact_mask_launch (then `valid` points at it)
    const uint8_t *valid_in;     // caller's validity bytes or nullptr
    int32_t n_mask, drop;        // drop == 0: only valid_in decides
    // scatter: what a surviving null becomes -- the weights column -> 1e-24 (least_squares.py:193), others -> 0 when zero_fill
    int32_t w_col, zero_fill;
    unsigned long long *blk_cnt; // valid rows of every 1 024 slabs (set by row_compact_offsets_launch)
    int32_t *src;                // source map (row_compact_srcmap_launch): src[c] = the frame row compacted row c came from, n_valid entries
};
int row_compact_mask_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a);
int row_compact_offsets_launch(pols_ctx *ctx, const RowCompactArgs &a);            // count + scan + groups
int row_compact_scatter_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a);
int row_compact_expand_launch(pols_ctx *ctx, int dtype, const RowCompactArgs &a);
int row_compact_srcmap_launch(pols_ctx *ctx, const RowCompactArgs &a);             // instead of the scatter: the tile kernel gathers through src (k4c_kernel.inl, GATHER)

// The validity prefix of the chunk-parallel dynamic kernels (K4Args::cnt / vidx and the per-group warm-up constants of
// solve_rolling_ols, ls.rs:881-891) built ON THE DEVICE from device validity bytes: the host-side build copied the bytes home, walked
// every row and uploaded two int32 tables -- 40+ ms of a 10 M-row call whose kernels take 3 (profiles/r04_bench_dyn_nulls.txt).
// Needs row_compact_offsets_launch's slab_base / c_offs of the same frame.
struct ValidTablesArgs {
    const uint8_t *valid;
    const int64_t *offs;         // DEVICE group offsets, n_groups + 1
    int64_t n_rows, n_groups, n_slabs;
    const int64_t *slab_base;    // valid rows before each 256-row slab
    const int64_t *c_offs;       // valid rows before each group
    int32_t *cnt;                // out: inclusive count of valid rows inside the group, per row
    int32_t *vidx;               // out: row (relative to the group) of the r-th valid row, slot offs[g] + r (pre-filled with -1)
    void *groups;                // K4Group[n_groups]: mpv and gate_n are patched (k4_rolling.hpp)
    int64_t min_periods;
};
int valid_tables_launch(pols_ctx *ctx, const ValidTablesArgs &a);

// The per-row solve table of the masked tile kernel (rolling OLS, "drop_window" on frames with validity bytes; k4c_kernel.inl MASKED):
// roll_mask_tables_launch -- validity prefix + per-sequence warm-up constants + the "never-dropped row" flag the host reads before it
// routes; roll_mask_rows_launch -- every row's code and the slabs' carries; roll_mask_fill_launch -- behind the kernel.
struct RollMaskArgs {
    const uint8_t *valid;
    const int64_t *offs;         // DEVICE group offsets, n_groups + 1
    int64_t n_rows, n_groups, n_slabs;
    uint32_t *slab_cnt;          // n_slabs: valid rows of every 256-row slab
    int64_t *slab_base;          // n_slabs + 1: valid rows before the slab
    int64_t *c_offs;             // n_groups + 1: valid rows before every group
    uint16_t *incl;              // n_rows: valid rows of the slab at or before the row
    int64_t *g_mpv;              // n_groups: min_periods_valid (n + 1: the whole sequence is NaN)
    int32_t *g_gate;             // n_groups: n_valid of the gate
    int32_t *flag;               // bit 0: some sequence keeps a valid row older than its window in the warm-up sum
    unsigned long long *blk_cnt, *blk_last;   // per 1 024 slabs: valid rows / (last solved row + 1) (rm_blk_kernel)
    uint8_t *solved;             // n_rows: 1 on the rows the reference solves
    uint16_t *code;              // n_rows: 0 solved here; 1 NaN; 2 + j: repeats row j of its slab; 2 + 256: repeats slab_carry[slab]
    int64_t *slab_last, *slab_carry;   // n_slabs each
    int64_t window, min_periods;
    // fill pass
    const void *x[10];
    void *coef, *pred;
    int32_t k;
};
int roll_mask_tables_launch(pols_ctx *ctx, const RollMaskArgs &a);
int roll_mask_rows_launch(pols_ctx *ctx, const RollMaskArgs &a);
int roll_mask_fill_launch(pols_ctx *ctx, int dtype, const RollMaskArgs &a);

int dyn_scan_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);
int dyn_rewrite_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);
int dyn_post_launch(pols_ctx *ctx, int dtype, const DynPrepArgs &a);

}  // namespace pols
