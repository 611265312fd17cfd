// k4_rolling.hpp -- K4 "rolling_gram_solve": rolling-window OLS for every group (see k4_rolling.hip).
#pragma once
#include "common.hpp"

#include <cmath>

namespace pols {

constexpr int K4_KMAX = 8;

// One chunk of consecutive rows of one group, walked by one lane.
struct K4Chunk {
    int64_t t0, t1;      // absolute rows [t0, t1)
    int32_t group;
    int32_t index_in_group;
};

// Per-group constants of solve_rolling_ols (src/least_squares.rs:858-900).
struct K4Group {
    int64_t start, end;      // absolute rows
    int64_t mpv;             // min_periods_valid (:881-891), relative to the group start
    int64_t gate_n;          // n_valid as the loop of :883-891 leaves it (the n_valid_window gate of :1013,:1022)
    int32_t first_chunk;
    int32_t all_nan;         // early return of :893-900
};

struct K4Args {
    const void *y;
    const void *x[POLS_MAX_FEATURES];
    const void *const *xtab;     // DEVICE table of k column pointers: used instead of x[] beyond 32 features (k4x_inverse.hip)
    const uint8_t *valid;        // or nullptr = all valid
    const int32_t *cnt;          // inclusive count of valid rows inside the group, per row; nullptr when valid == nullptr
    const int32_t *vidx;         // row (relative to the group) of the r-th valid row, per row slot; nullptr when valid == nullptr
    const K4Chunk *chunks;
    const K4Group *groups;
    int64_t n_chunks;
    int32_t n_groups;
    double *totals;              // per-chunk sums, then exclusive prefix at the chunk start: element (chunk c, component q) lives at
                                 // totals[c * tot_cs + q * tot_qs].  Lane-per-chunk kernels use component-major storage
                                 // (tot_cs = 1, tot_qs = n_chunks: consecutive lanes = consecutive chunks = consecutive words, for the
                                 // totals writes, the scan and the walk alike); wave / workgroup-per-chunk kernels chunk-major.
    int64_t tot_cs, tot_qs;
    void *coef;                  // n_rows x k or nullptr
    void *pred;                  // n_rows or nullptr
    int64_t window;
    double alpha;
    int32_t k;
    int32_t drop_mode;           // 1: window over the last `window` VALID rows (:947-986); 0: fixed window (:987-1029)
    int32_t chunk_len;
    // RLS in information form (k3s_launch): totals rows carry one extra slot (the chunk's decay factor)
    double ff;                   // forgetting factor (ls.rs:513-517)
    double p0;                   // initial_state_covariance: A_0 = I / p0
    const double *mean0;         // device, k values or nullptr
    double *state;               // k4x beyond 128 features: one k x (k | 1) matrix per chunk in HBM (set by the launcher)
    const int32_t *order;        // k4p / k3p (or nullptr): work slot i takes chunk order[i] -- the chunks sorted by length, longest first, so that the chunks that
                                 // share a wave (up to four) are about equally long; chunk ids, totals and the group tables are untouched
    // k4p rolling on null-free frames: rows whose window sums could not be inverted, for kp_lu_fix_kernel (set by k4p_launch; see K4cArgs)
    int64_t *fix_rows;
    int32_t *fix_count, *fix_next;
    int64_t fix_cap, groups_end_row;   // groups_end_row: rows of the frame the tables describe (the caller sets it)
    int32_t use_totals;          // k4p rolling: sequences are cut AND the window exceeds 1 024 rows -- the chunk-start sums come from the scanned
                                 // per-chunk totals (prefix differences) instead of re-summing the window in front of the chunk
};

int k4_launch(pols_ctx *ctx, int dtype, const K4Args &a);
// Chunk-parallel RLS for long sequences: A_t = ff A_{t-1} + x x', b_t = ff b_{t-1} + x y (valid rows), beta_t = A_t^-1 b_t,
// which is the reference's P-form recursion (ls.rs:531-540) by the Sherman-Morrison identity.
int k3s_launch(pols_ctx *ctx, int dtype, const K4Args &a);
// 9..32 features: one wave per chunk, state in LDS (k4w_wide.hip).  Totals rows are k*k + k (+ 1 for the RLS decay) doubles.
int k4w_launch(pols_ctx *ctx, int dtype, const K4Args &a);
int k3sw_launch(pols_ctx *ctx, int dtype, const K4Args &a);
// 9..32 features, one wave per chunk, the inverse / covariance distributed over the wave's registers (k4p_wide.hip): RLS (any validity
// mask) and rolling OLS on null-free frames with min_periods <= window.  single_chunk: no sequence was cut into several chunks.
int k4p_launch(pols_ctx *ctx, int dtype, const K4Args &a, bool rls, bool single_chunk);
// 33..128 features: one workgroup per chunk, the inverse propagated in LDS (k4x_inverse.hip).  Totals rows as for k4w.
int k4x_launch(pols_ctx *ctx, int dtype, const K4Args &a);
int k3x_launch(pols_ctx *ctx, int dtype, const K4Args &a);
constexpr int K4X_KMAX = 128;
// 129..1024 features: the same with the per-chunk state in HBM / L2 and 1 024 threads (k4y_hbm.hip)
int k4y_launch(pols_ctx *ctx, int dtype, const K4Args &a);
int k3y_launch(pols_ctx *ctx, int dtype, const K4Args &a);
constexpr int K4Y_KMAX = 1024;
// pass 2 of every chunk-parallel kernel: exclusive prefix of the chunk totals, one wave per (group, component);
// mode 0 plain (rolling), 1 / 2 decayed with the RLS prior as carry-in (packed / full K x K state), see k4_rolling.hip
void chunk_scan_launch(pols_ctx *ctx, const K4Args &a, int nacc, int mode);

constexpr int KC_KMAX = 10;      // the row-parallel dynamic kernels: K3c up to 9 features, K4c up to 10 (512 registers at one wave per SIMD, a 137 KB table)
// ---- K3c: row-parallel, read-once RLS for up to KC_KMAX features (k3c_scan.hip) ------------------------------------------------
constexpr int K3C_KMAX = 10;      // (its cross-wave steps spread the state components over the lanes: one per lane up to 9 features, two at 10)
constexpr int K3C_NCP = 72;      // doubles per tile record: k (k + 1) / 2 + k + 1 <= 66
constexpr int K3C_R = 4;         // consecutive rows per lane
#ifndef K3C_WAVES_SMALL
#define K3C_WAVES_SMALL 4
#endif
constexpr int k3c_waves(int k) { return k <= 6 ? K3C_WAVES_SMALL : 2; }    // waves per tile (each parks R (k + 1) row values per lane in LDS)
constexpr int64_t k3c_tile_rows(int k) { return (int64_t)K3C_R * 64 * k3c_waves(k); }
struct K3cArgs {
    const void *y;
    const void *x[KC_KMAX];
    const uint8_t *valid;              // validity bytes or nullptr = every row valid
    const uint8_t *start;              // 1 on the first row of every sequence (k3c_start_flags)
    int64_t n_rows;
    void *coef, *pred;                 // n_rows x k / n_rows, batch dtype, 16-byte aligned; either may be nullptr
    const double *mean0;               // device, k values, or nullptr
    double ff, p0;
    // pass 1 -> pass 2: one record per tile (K3C_NCP doubles: the aggregate from the tile's last sequence start on, slot NT its
    // decay) and whether the tile holds a sequence start; per tile the composite of the tiles of its 64-tile block below it and
    // whether it is still open (no sequence start among them); the same per block: records, carry-ins
    double *rec, *carry, *brec, *bcarry;
    int32_t *rec_closed, *carry_open, *brec_closed;
    int64_t n_tiles;
    int32_t fold_top;                  // set by the launcher: at most 64 blocks of tiles -- pass 2 composes its block's carry-in from the block records itself
                                       // (a loop over <= 63 lane-vector records) and the one-wave top-level scan launch is skipped
    int32_t all_closed;                // no sequence is longer than a tile: every tile holds a sequence start, tile t's carry-in is tile t - 1's record
    const int64_t *tile_row0;          // PACKED tiles (or nullptr): tile t owns rows [tile_row0[t], tile_row0[t + 1]), whole sequences only, and
                                       // loads from tile_row0[t] & ~3 on -- no carry-in, so one launch (pass 2 alone) does the frame
    unsigned long long *dbg;           // POLS_TIMELINE: 8 words per tile (s_memtime stamps of the tile's last wave) or nullptr
    int32_t k;
    // HALO form (finite half-life, null-free frames, K3C_HALO_KMAX features): tile t's carry-in is re-accumulated from the 256 halo_batches rows in front of
    // it -- one launch, no records.  tile_seq0[t]: first row of the sequence that holds row t * tile_rows - 1 (tile 0: 0); nullptr = the scan forms above.
    const int64_t *tile_seq0;
    int32_t halo_batches;              // 256-row batches: ff^(256 halo_batches) <= 2^-36
    double log2ff, ffstep;             // log2(ff); ff^(256 waves - 4): a lane's runs of its wave's consecutive batches are 256 waves rows apart
    // LOOK-BACK-ONE form (up to 6 features, halo_batches <= 4: the rows that matter lie inside the tile in front): every tile publishes its own
    // aggregate as granules, tile t reads tile t - 1's; nullptr = the halo form
    void *gran;                        // n_tiles x 32 granules of 16 bytes {value, tag}, zeroed when allocated
    int64_t gran_bytes;
    unsigned long long epoch;          // this launch's tag (never 0, never reused on this area)
    int32_t spin_limit;                // polls before a wave gives up on its predecessor and re-accumulates the halo itself
    int32_t early_publish;             // 1: the record is computed and published before the tile's scan (step E); 0: it falls out of the scan (A/B)
};
constexpr int K3C_HALO_KMAX = 9;      // (one state component per lane in the cross-wave steps)
constexpr int K3C_HALO_MAX_BATCHES = 8;
// 256-row batches the halo form re-reads in front of every tile for this forgetting factor (0: it does not apply -- no / too long a half-life)
inline int32_t k3c_halo_batches(double ff) {
    if (!(ff > 0.0) || !(ff < 1.0)) return 0;
    const double need = -36.0 / std::log2(ff);              // ff^need = 2^-36 = 1.5e-11
    const int32_t nb = need <= 256.0 ? 1 : (int32_t)std::ceil(need / 256.0);
    return nb <= K3C_HALO_MAX_BATCHES ? nb : 0;
}
int k3c_launch(pols_ctx *ctx, int dtype, const K3cArgs &a);
int k3c_start_flags(pols_ctx *ctx, const int64_t *d_offs, int64_t n_groups, int64_t n_rows, uint8_t *start);

// ---- K4c: row-parallel rolling OLS on null-free frames, window <= K4C_MAX_WINDOW (k4c_rolling.hip) ------------------------------
constexpr int K4C_KMAX = 10;                // 7 .. 10 features: more than 256 registers (256 + 24 .. 190 AGPRs), one four-wave workgroup per CU instead of two
constexpr int64_t K4C_MAX_WINDOW = 508;   // two halo waves: 512 >= 4 ceil(window / 4) + 1
struct K4cArgs {
    const void *y;
    const void *x[KC_KMAX];
    const uint8_t *start;              // 1 on the first row of every sequence (k3c_start_flags)
    const uint8_t *valid;              // MASKED form ("drop_window" with nulls): validity bytes, 4-byte aligned; nullptr: every row valid
    const uint8_t *solved;             // MASKED: 1 on the rows the reference solves (dyn_prep.hip rm_rows_kernel), 4-byte aligned
    int64_t n_rows, n_tiles;           // n_tiles is set by the launcher
    void *coef, *pred;                 // n_rows x k / n_rows, batch dtype, 16-byte aligned; either may be nullptr
    int64_t window, min_periods;       // 1 <= min_periods <= window <= K4C_MAX_WINDOW
    double alpha;
    unsigned long long *dbg;           // POLS_TIMELINE: 8 words per tile or nullptr (set by the launcher)
    int32_t k;
    const int64_t *tile_row0;          // PACKED tiles (or nullptr), K3cArgs::tile_row0's table for 1 024-row tiles: whole sequences per tile, so no
    int64_t n_packed;                  // window reaches outside it and the halo waves go (n_packed tiles)
    // GATHER ("drop" family on a frame with nulls): the kernel's rows are the VALID rows of the frame -- n_rows of them, `start` their
    // sequence-start flags -- read THROUGH src from the frame's own columns x / y, and the outputs go to the FRAME's rows (forward fill,
    // dyn_out_gather.inl): no compacted copy of the columns, no expansion pass.  nullptr: the rows are the columns' rows.
    const int32_t *src;                // src[c]: the frame row of compacted row c
    int64_t n_frame;                   // rows of the frame (coef / pred are n_frame x k / n_frame)
    const uint8_t *fvalid, *fstart;    // the frame's validity and sequence-start bytes
};
// widest window of the halo forms (packed tiles take any window: nothing reaches outside a tile)
constexpr int64_t k4c_max_window(int k) { return k <= 6 ? K4C_MAX_WINDOW : 252; }
constexpr int64_t K4C_PACKED_ROWS = 1024;    // rows of a packed tile (four body waves)
int k4c_launch(pols_ctx *ctx, int dtype, const K4cArgs &a);

}  // namespace pols
