// k2w_kernel.inl -- K2w "gram_mfma_resident, two tiles": OLS / ridge for 17 .. 31 columns (the intercept counted) with the group's
// rows held in registers, so that X is read from HBM exactly ONCE whatever the group length up to the resident capacity -- the
// shapes that used to take the three-launch streamed path (Gram pass, solve, prediction pass: X read twice, ~2 TB/s).
//
// Replaces, per group: construct_features_array + solve_ols / solve_ridge (src/least_squares.rs:211-240, 342-364; normal equations,
// Cholesky) + make_predictions (src/expressions.rs:175-195) + the sqrt(w) / intercept / un-scaling / residual steps of
// polars_ols/least_squares.py:184-196, 234-239.  A failed or flagged factorisation marks the group POLS_GROUP_FALLBACK for the
// fix-up pass (K6), exactly like K1 / K2.
//
// One PERSISTENT workgroup of WAVES (2, 4 or 8) waves walks the groups blockIdx.x, + gridDim.x, ...  Per group:
//   load    : every lane keeps ONE 16-byte chunk (2 f64 / 4 f32 consecutive rows) of EVERY column in VGPRs -- 32 column slots, all
//             loaded unconditionally from clamped positions (K2's scheme: no load inside a divergent branch), 128 VGPRs of data.
//   gram    : Z = [sqrt(w) X | sqrt(w) 1 | sqrt(w) y] has up to 32 columns = two 16-column halves; Z'Z is THREE 16 x 16 tiles on the
//             matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32): (0,0) = A0'A0, (0,1) = A0'A1, (1,1) = A1'A1 with A_h the
//             operand stream of half h.  The matrix cores want lane (c, q) to hold Z[row 4t + q][column c]; the registers hold a
//             row per lane, so each wave transposes through its PRIVATE 8.25 KB LDS tile (K2's, 16 column slots): half 0 is written,
//             its 16 operands are read into registers, half 1 is written over it, and its operands are consumed one at a time
//             next to the three MFMAs of the step.  No workgroup barrier in the Gram phase.  f32 tiles are flushed to f64 per stage.
//   reduce  : per-wave partial tiles -> LDS -> one f64 32 x 32 matrix, fixed order (run-to-run identical).             [2 barriers]
//   solve   : one wave, f64: [A | b] padded with an identity block, lane i keeps row i in registers, Gauss-Jordan elimination without
//             pivoting with v_readlane broadcasts (steps beyond kt skipped): no substitution passes (k2w_chol below).
//   predict : X . beta (+ residuals) from the resident rows, 16-byte streaming stores.                                  [1 barrier]
// Bound: HBM, b n (k + 1) (+ b n weights) bytes in, b n out per group.
#pragma once
#include <cstddef>
#include "k2_kernel.inl"
#include "k2w_resident.hpp"

namespace pols {

constexpr int K2W_KC = 32;                       // column slots (two 16-column MFMA tiles)
constexpr int K2W_GS = 33;                       // row stride of the f64 matrices in LDS
constexpr int K2W_TAIL_B = (2 * 32 * K2W_GS + 64) * 8;   // Z'Z, the solver's matrix, 64 doubles of vectors

// (X'X + alpha I) beta = X'y on the padded system (the reference: faer cholesky(Side::Lower) + two triangular solves, ls.rs:288-297);
// false = failed / flagged pivot.  G: Z'Z with X'y in column kt; Tm: 32 x 33 scratch.  Lane i keeps row i of [A | b] in registers and the
// wave runs Gauss-Jordan elimination WITHOUT pivoting: at step j every row but row j subtracts its multiple of row j (v_readlane
// broadcasts of row j's entries), so the matrix ends up diagonal and beta_i = b_i / a_ii in every lane at once.  The pivots are the
// squares of Cholesky's diagonal (the same Schur complements), so the positive-definiteness / conditioning test is the same test;
// what goes is the two substitution passes -- 2 kt dependent broadcast -> multiply -> subtract steps of ~80 cycles each -- and the
// transposition between them: in this layout a row update costs the same instructions whether 31 - j rows take part (Cholesky) or
// 31 (here).  Measured (f64, 1 000-row groups, per group): 31 columns 20.7 k -> see DESIGN cycles, 20 columns 13.7 k -> see DESIGN.
// Round 3, measured against publishing the pivot column through LDS once per step instead of the broadcasts: the LDS form was SLOWER
// (31 columns x 1 000 rows: 614 vs 515 us per 5 000 groups).
// KB: the padded size of the system, 20 / 24 / 28 / 32 (kt <= KB); the elimination is unrolled over KB.
__device__ __forceinline__ double k2w_rcp(double d) {           // v_rcp_f64 + two Newton steps
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

template <int KB>
__device__ __forceinline__ bool k2w_chol(const double *G, int kt, double alpha, double pivot_tol, double *Tm, int lane, double &bi) {
    constexpr int KC = KB;
    const int i = (lane & 31) < KB ? (lane & 31) : KB - 1;   // lanes 32..63 repeat lanes 0..31 (broadcasts read lanes 0..kt-1); rows beyond KB mirror the last
    // [X'X + alpha I | X'y] padded with an identity block, built in LDS: lane (row, half) writes half a row with independent loads,
    // unrolled (the lane id is laundered per group in the caller, so nothing here is hoisted out of the persistent loop)
    {
        constexpr int HW = (KC + 2) / 2;                 // columns per half row, the right-hand side included
        const int r = lane & 31, c0 = (lane >> 5) * HW;
        double v[HW];
#pragma unroll
        for (int cc = 0; cc < HW; ++cc) {
            const int c = c0 + cc;
            const int src = (c == KC) ? kt : c;          // column kt of G holds X'y
            v[cc] = (r < kt && (c < kt || c == KC)) ? G[r * K2W_GS + src] : 0.0;
        }
#pragma unroll
        for (int cc = 0; cc < HW; ++cc) {
            const int c = c0 + cc;
            double w = v[cc];
            if (c < KC && r == c) w = (r < kt) ? w + alpha : 1.0;
            if (r < KC && c <= KC) Tm[r * K2W_GS + c] = w;
        }
    }
    k2_wave_sync();
    double row[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) row[c] = Tm[i * K2W_GS + c];
    bi = Tm[i * K2W_GS + KC];
    const double gd = Tm[i * K2W_GS + i];
    k2_wave_sync();                                      // (Tm belongs to the next group's build)
    double rme = 1.0;                                    // 1 / pivot i, kept on lane i
    bool ok = true;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        if (j < kt) {                                    // wave-uniform: the identity block needs no elimination
            const double d = k2_bcast(row[j], j);
            ok = ok & (d > pivot_tol * k2_bcast(gd, j)); // also false for NaN
            const double ri = k2w_rcp(d);
            if (i == j) rme = ri;
            const double f = (i == j) ? 0.0 : row[j] * ri;                       // this row's multiple of row j (row j itself stays)
#pragma unroll
            for (int c = j + 1; c < KC; ++c) row[c] = fma(-f, k2_bcast(row[c], j), row[c]);
            bi = fma(-f, k2_bcast(bi, j), bi);
            __builtin_amdgcn_sched_barrier(0);           // a step's broadcasts stay in their step
        }
    }
    bi *= rme;                                           // beta_i (rows beyond kt: 0 x 1)
    return ok;
}

// npf: the first npf of a wave's loaded columns (features, then the target, then the weights) of the workgroup's NEXT group are DMA'd
// into LDS by the waves that are not solving, during the solve
// (`global_load_lds`, 1 KiB per wave-instruction) -- a CU's memory pipe delivers ~10 bytes per clock however idle HBM is, and a
// 31-column x 1 000-row f64 group is 264 KB of it.  Pieces 0..7 of a wave sit in that wave's own Gram tile (idle between the
// reduce and the next group's transposes; read back by the same wave before it writes the tile: no barrier), the rest in the LDS
// beyond the solver's matrices, K2W_PF_DED pieces per wave.
constexpr int K2W_PF_TILE = 8;                   // 1 KiB pieces inside a wave's 8.25 KiB tile
template <typename T, int WAVES, bool HAS_W>
__global__ void __launch_bounds__(64 * WAVES, 2) k2w_kernel(const K2wArgs a, const int npf, const int pf_ded) {
    using V = typename Vec16<T>::type;
    using M = Mfma16<T>;
    using acc_t = typename M::acc_t;
    using H = decltype(k2_half(V{}, 0));                     // double or float2
    constexpr int VEC = Vec16<T>::N;
    constexpr int TPB = 64 * WAVES;
    constexpr int KC = K2W_KC;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *Gs = reinterpret_cast<double *>(smem + (size_t)WAVES * K2_TILE_B);      // [32][33]
    double *As = Gs + 32 * K2W_GS;                                                    // solver matrix [32][33]
    double *vec = As + 32 * K2W_GS;                                                   // [0, 32) beta, [32, 64) the column pointers (prefetch)
    unsigned long long *s_xptr = reinterpret_cast<unsigned long long *>(vec + 32);
    unsigned char *pfd = smem + (size_t)WAVES * K2_TILE_B + K2W_TAIL_B;               // [wave][pf_ded][1 KiB]
    // piece (consumer wave wt, loaded column col < npf) lives at:
    auto pf_at = [&](int wt, int col) -> unsigned char * {
        return col < K2W_PF_TILE ? smem + (size_t)wt * K2_TILE_B + (size_t)col * 1024
                                 : pfd + ((size_t)wt * pf_ded + (col - K2W_PF_TILE)) * 1024;
    };
    bool pf_ready = false;                                   // the pieces hold columns of the group being worked on
    if (npf > 0) {
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (threadIdx.x == j) s_xptr[j] = reinterpret_cast<unsigned long long>(a.x[j]);
    }                                                        // (read after several barriers)

#pragma unroll 1
    for (int64_t g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
    // the thread index and the column counts are laundered once per group (see k2_kernel: nothing derived from them may be hoisted
    // out of the persistent loop into registers that sit on top of the resident rows)
    int tid = threadIdx.x, kt = a.kt, ku = a.k_user;
    asm volatile("" : "+v"(tid), "+s"(kt), "+s"(ku) :: "memory");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool icpt = ku != kt;
    constexpr bool has_w = HAS_W;
    unsigned char *mytile = smem + (size_t)wave * K2_TILE_B;
    const int64_t s = a.offs[g], e = a.offs[g + 1];
    const int64_t base = s - (s % VEC);                      // the chunk grid is aligned to 16 bytes in every column
    const int64_t nch = (e - base + VEC - 1) / VEC;          // <= TPB: the host checked the largest group
    unsigned long long *dbg = a.dbg ? a.dbg + g * 8 : nullptr;
#define K2W_STAMP(i) do { if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
    K2W_STAMP(0);

    // ---- every load of the resident chunk, back to back, no divergent branch around a load (k2_kernel's scheme): lanes whose chunk
    // crosses the group's edge read their neighbours' rows too, lanes without a chunk read from a clamped position; a wave-uniform
    // fix-up shifts / zeroes afterwards.  Column slots beyond the user's features are not loaded (wave-uniform).
    V x[KC], yv, sw;
    unsigned keep = 0;                                       // bit v: row v of the chunk belongs to the group
    int shift;                                               // rows the load position was moved back by (end of the frame)
    const int64_t row0 = base + (int64_t)tid * VEC;
    {
        const bool any = tid < nch;
        int64_t rl = any ? row0 : base;
        if (rl > a.n_rows - VEC) rl = a.n_rows - VEC;        // n_rows >= VEC: checked by the host
        shift = any ? (int)(row0 - rl) : 0;
#pragma unroll
        for (int v = 0; v < VEC; ++v) keep |= (any && row0 + v >= s && row0 + v < e) ? (1u << v) : 0u;
        const int np = pf_ready ? npf : 0;                   // block-uniform: columns < np wait in LDS (same clamped positions)
#pragma unroll
        for (int j = 0; j < KC; ++j) {                       // (one chunk per lane: every load is awaited before the first use anyway, so a
            if (j < 16 || j < ku) {                          // wave-uniform skip; the first 16 slots are always user columns (kt >= 17))
                if (j < np) x[j] = *reinterpret_cast<const V *>(pf_at(wave, j) + lane * 16);
                else x[j] = *reinterpret_cast<const V *>(static_cast<const T *>(a.x[j]) + rl);
            }
        }
        if (ku < np) yv = *reinterpret_cast<const V *>(pf_at(wave, ku) + lane * 16);
        else yv = *reinterpret_cast<const V *>(static_cast<const T *>(a.y) + rl);
        if (has_w && ku + 1 < np) sw = *reinterpret_cast<const V *>(pf_at(wave, ku + 1) + lane * 16);
        else sw = *reinterpret_cast<const V *>(static_cast<const T *>(has_w ? a.w : a.y) + rl);
    }

    // ---- this chunk's registers
#pragma unroll
    for (int j = 16; j < KC; ++j)                            // (a taken wave-uniform branch is ~30 cycles: only the slots that CAN be synthetic)
        if (j >= ku) x[j] = vsplat<T>((icpt && j == kt - 1) ? T(1) : T(0));      // wave-uniform: intercept / unused slot
    if (!has_w) sw = vsplat<T>(T(1));
    if (__any(keep != ((1u << VEC) - 1u))) {                 // wave-uniform, no loads inside: a ragged edge somewhere in the wave
#pragma unroll
        for (int j = 0; j < KC; ++j) x[j] = k2_fix<T>(x[j], (j < ku) ? shift : 0, keep, T(0));
        yv = k2_fix<T>(yv, shift, keep, T(0));
        sw = k2_fix<T>(sw, has_w ? shift : 0, keep, T(1));
    }
    if (has_w) {       // sqrt(w) scaling of every feature, intercept included (least_squares.py:190-196); yv keeps the ORIGINAL target
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T q = sqrt(vget<T>(sw, v));
            vset<T>(sw, v, q);
#pragma unroll
            for (int j = 0; j < KC; ++j) vset<T>(x[j], v, vget<T>(x[j], v) * q);
        }
    }
    V ys;                                                    // sqrt(w) y
#pragma unroll
    for (int v = 0; v < VEC; ++v) vset<T>(ys, v, vget<T>(yv, v) * vget<T>(sw, v));

    // ---- Gram: per 8-byte half of the chunk, transpose both 16-column halves of Z through the wave's tile and feed the matrix cores
    double accd[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) accd[t][r] = 0.0;
    acc_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
    const unsigned char *zp = mytile + (size_t)(lane & 15) * K2_SLOT_B + (lane >> 4) * 8;   // operand stream of lane (c, q)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // half 0 of Z: columns 0 .. 15 (always features: kt >= 17)
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<H *>(mytile + (size_t)j * K2_SLOT_B + lane * 8) = k2_half(x[j], h);
        k2_wave_sync();
        if (h == 0) K2W_STAMP(1);                            // this wave's loads have landed
        H va[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) va[u] = *reinterpret_cast<const H *>(zp + u * 32);
        k2_wave_sync();                                      // everyone has its operands: the tile may be overwritten
        // half 1: columns 16 .. 31 -- features, intercept, zeros (x[j] already holds them) and the target in slot kt - 16
#pragma unroll
        for (int j = 16; j < KC; ++j) {
            const H vj = (j == kt) ? k2_half(ys, h) : k2_half(x[j], h);          // wave-uniform select
            *reinterpret_cast<H *>(mytile + (size_t)(j - 16) * K2_SLOT_B + lane * 8) = vj;
        }
        k2_wave_sync();
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const H vb = *reinterpret_cast<const H *>(zp + u * 32);
            if constexpr (sizeof(T) == 8) {
                acc00 = M::mma(va[u], va[u], acc00);
                acc01 = M::mma(va[u], vb, acc01);
                acc11 = M::mma(vb, vb, acc11);
            } else {
                acc00 = M::mma(va[u].x, va[u].x, acc00);
                acc01 = M::mma(va[u].x, vb.x, acc01);
                acc11 = M::mma(vb.x, vb.x, acc11);
                acc00 = M::mma(va[u].y, va[u].y, acc00);
                acc01 = M::mma(va[u].y, vb.y, acc01);
                acc11 = M::mma(vb.y, vb.y, acc11);
            }
        }
        if constexpr (sizeof(T) == 4) {                      // 128 rows per flush: an f32 Gram matrix with f64-summation error
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accd[0][r] += (double)acc00[r]; accd[1][r] += (double)acc01[r]; accd[2][r] += (double)acc11[r];
            }
            acc00 = acc_t{0, 0, 0, 0}; acc01 = acc_t{0, 0, 0, 0}; acc11 = acc_t{0, 0, 0, 0};
        }
        k2_wave_sync();                                      // the next stage overwrites the tile
    }
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { accd[0][r] = acc00[r]; accd[1][r] = acc01[r]; accd[2][r] = acc11[r]; }
    }
    K2W_STAMP(2);
    // ---- per-wave partial tiles -> LDS (3 x 256 doubles = 6 KB of the wave's own tile)
    {
        double *part = reinterpret_cast<double *>(mytile);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[t * 256 + r * 64 + lane] = accd[t][r];
    }
    __syncthreads();
    for (int idx = tid; idx < 3 * 256; idx += TPB) {
        const int t = idx >> 8, q = idx & 255, r = q >> 6, l = q & 63;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += reinterpret_cast<const double *>(smem + (size_t)w * K2_TILE_B)[idx];
        const int drow = (sizeof(T) == 4) ? (l >> 4) * 4 + r : (l >> 4) + 4 * r;   // C/D layouts of the f32 / f64 16x16x4 MFMA
        const int dcol = l & 15;
        const int gi = (t == 2 ? 16 : 0) + drow, gj = (t == 0 ? 0 : 16) + dcol;     // tile (0,0), (0,1), (1,1)
        Gs[gi * K2W_GS + gj] = v;
        if (t == 1) Gs[gj * K2W_GS + gi] = v;                                       // (1,0) = (0,1)'
    }
    __syncthreads();
    K2W_STAMP(3);
    // which wave solves: workgroups that share a CU put their solver on different SIMDs (a workgroup's wave w runs on SIMD (w mod 4)
    // when the CU fills in order; with every workgroup solving on its wave 0 the co-resident solves all queue on one SIMD while three idle)
    const unsigned per_cu_round = gridDim.x >= 8 / WAVES ? gridDim.x / (8 / WAVES) : 1u;   // (workgroup b and b + per_cu_round tend to share a CU)
    const int solver = WAVES == 8 ? 0 : (int)((blockIdx.x / per_cu_round) % WAVES);
    bool pf_next = false;
    {
        const int64_t gn = g + gridDim.x;
        pf_next = npf > 0 && gn < a.n_groups;                // block-uniform
        if (pf_next && wave != solver) {                     // (the tiles' partial sums were consumed before the barrier above)
            const int producer = wave < solver ? wave : wave - 1;             // 0 .. WAVES - 2
            const int64_t sn = a.offs[gn], en = a.offs[gn + 1];
            const int64_t basen = sn - (sn % VEC);
            const int64_t nchn = (en - basen + VEC - 1) / VEC;
            const int npieces = WAVES * npf;
            for (int p = producer; p < npieces; p += WAVES - 1) {
                const int col = p / WAVES, wt = p - col * WAVES;                // column-major: every wave's first columns first
                // (a.x[col] with a run-time col: from the LDS copy of the pointer table made at kernel start -- indexing the kernel
                // arguments at run time left the f64 build with a 36-byte private segment)
                const T *xcol = reinterpret_cast<const T *>(s_xptr[col < ku ? col : 0]);
                const T *src = col < ku ? xcol : static_cast<const T *>(col == ku ? a.y : a.w);
                const int64_t c = (int64_t)wt * 64 + lane;
                int64_t rl = c < nchn ? basen + c * VEC : basen;               // the clamped position the consumer expects
                if (rl > a.n_rows - VEC) rl = a.n_rows - VEC;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + rl),
                                                 (__attribute__((address_space(3))) void *)pf_at(wt, col), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // landed before the barrier below lets anyone read them
        }
    }

    // ---- solve: one wave, f64
    if (wave == solver) {
        int st = POLS_GROUP_OK;
        double bi = 0.0;
        if (e == s) st = POLS_GROUP_EMPTY;                   // features.is_empty() -> zeros (ex.rs:357-359)
        else if (!(kt <= 20 ? k2w_chol<20>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi)
                   : kt <= 24 ? k2w_chol<24>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi)
                   : kt <= 28 ? k2w_chol<28>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi)
                              : k2w_chol<32>(Gs, kt, a.alpha, a.pivot_tol, As, lane, bi))) {
            st = POLS_GROUP_FALLBACK;
            if (lane == 0 && a.fb_flag) *a.fb_flag = a.epoch;
        }
        if (lane == 0 && a.status) a.status[g] = st;
        if (lane < 32) {
            const double out = lane < kt ? bi : 0.0;
            vec[lane] = out;
            if (lane < kt && a.coef) static_cast<T *>(a.coef)[g * kt + lane] = (T)out;
        }
    }
    __syncthreads();
    K2W_STAMP(4);

    // ---- predictions / residuals from the resident rows
    if (a.pred || a.resid) {
        T *pred = static_cast<T *>(a.pred);
        T *resid = static_cast<T *>(a.resid);
        T p[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) p[v] = T(0);
#pragma unroll
        for (int j = 0; j < KC; ++j) {                       // make_predictions (ex.rs:398-405); coefficient j straight from LDS (0 beyond kt)
            const T bj = (T)vec[j];
#pragma unroll
            for (int v = 0; v < VEC; ++v) p[v] = fma(vget<T>(x[j], v), bj, p[v]);
        }
        if (tid < nch) {
            V pv, rv;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                T acc = p[v];
                if (has_w) acc *= T(1) / vget<T>(sw, v);                                         // predictions *= 1/sqrt_w (ls.py:234-235)
                vset<T>(pv, v, acc);
                vset<T>(rv, v, vget<T>(yv, v) - acc);                                            // ORIGINAL target - predictions (ls.py:239)
            }
            if (keep == ((1u << VEC) - 1u)) {
                if (pred) store_stream(reinterpret_cast<V *>(pred + row0), pv);
                if (resid) store_stream(reinterpret_cast<V *>(resid + row0), rv);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const int64_t rr = row0 + v;
                    if (rr >= s && rr < e) {
                        if (pred) pred[rr] = vget<T>(pv, v);
                        if (resid) resid[rr] = vget<T>(rv, v);
                    }
                }
            }
        }
    }
    K2W_STAMP(5);
    if (dbg && tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[6] = xcc;
    }
#undef K2W_STAMP
    __syncthreads();                                         // vec / Gs / the tiles are rewritten by the next group
    pf_ready = pf_next;
    }   // groups of this workgroup
}

template <typename T, int WAVES, bool HAS_W>
static int k2w_launch_v(pols_ctx *ctx, const K2wArgs &a) {
    size_t lds = (size_t)WAVES * K2_TILE_B + K2W_TAIL_B;
    // eight waves = one persistent workgroup per CU: the rest of the CU's 160 KiB LDS (and the idle Gram tiles) take the next group's
    // first columns while wave 0 solves (POLS_K2_NOPREFETCH=1: off)
    int npf = 0, pf_ded = 0;
    // (two-wave workgroups: measured slower with it, f64 31 / 24 columns x 200 rows 2.29 / 2.16 against 2.36 / 2.24 TB/s -- four workgroups
    // per CU already keep the memory pipe busy and the DMA competes with their loads; four waves: f32 31 columns x 1 000 rows 2.76 -> 2.87)
    if (WAVES >= 4 && !ctx->opt.k2_noprefetch) {
        const int ncols = a.k_user + 1 + (a.w ? 1 : 0);
        pf_ded = (int)(((size_t)160 * 1024 * WAVES / 8 - lds) / 1024 / WAVES);   // (8 / WAVES workgroups share the CU's LDS)
        npf = std::min(ncols, K2W_PF_TILE + pf_ded);
        pf_ded = std::max(0, npf - K2W_PF_TILE);
        lds += (size_t)WAVES * pf_ded * 1024;
    }
    // persistent: two waves per SIMD by the register budget -> one 8-wave or two 4-wave workgroups per CU
    const unsigned grid = (unsigned)std::min<int64_t>(a.n_groups, (int64_t)ctx->num_cus * (8 / WAVES));
    static OncePerDevice attr_once;
    if (attr_once.needed(ctx->device)) {
        POLS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k2w_kernel<T, WAVES, HAS_W>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done(ctx->device);
    }
    char name[96];
    std::snprintf(name, sizeof(name), "k2w_gram_mfma_resident2_%s_k%d_w%d%s_chol", sizeof(T) == 4 ? "f32" : "f64", a.kt, WAVES, a.w ? "_w" : "");
    ctx->last_kernel = name;
    if (a.n_groups > 0x7ffffff0LL) return fail(POLS_ERR_UNSUPPORTED, "too many groups for one launch");
    K2wArgs aa = a;
    if (ctx->opt.timeline) {
        void *d = nullptr;
        int rc = ensure_scratch(ctx, 11, sizeof(unsigned long long) * 8 * (size_t)a.n_groups, &d);
        if (rc) return rc;
        aa.dbg = static_cast<unsigned long long *>(d);
    }
    hipEvent_t ev0, ev1;
    if (timing_pair(ctx, &ev0, &ev1))
        hipExtLaunchKernelGGL((k2w_kernel<T, WAVES, HAS_W>), dim3(grid), dim3(64 * WAVES), (unsigned)lds, ctx->stream, ev0, ev1, 0, aa, npf, pf_ded);
    else
        hipLaunchKernelGGL((k2w_kernel<T, WAVES, HAS_W>), dim3(grid), dim3(64 * WAVES), lds, ctx->stream, aa, npf, pf_ded);
    POLS_HIP(hipGetLastError());
    if (ctx->opt.timeline) return report_timeline(ctx, aa.dbg, a.n_groups, 6, name);
    return POLS_OK;
}

template <typename T>
int k2w_launch_t(pols_ctx *ctx, const K2wArgs &a, int64_t need) {
    constexpr int VEC = Vec16<T>::N;
    // (two waves per SIMD by the register budget: 8 / WAVES workgroups per CU -- the short groups' serial solves overlap four deep)
    if (need <= 128 * VEC) return a.w ? k2w_launch_v<T, 2, true>(ctx, a) : k2w_launch_v<T, 2, false>(ctx, a);
    if (need <= 256 * VEC) return a.w ? k2w_launch_v<T, 4, true>(ctx, a) : k2w_launch_v<T, 4, false>(ctx, a);
    if (need <= 512 * VEC) return a.w ? k2w_launch_v<T, 8, true>(ctx, a) : k2w_launch_v<T, 8, false>(ctx, a);
    return fail(POLS_ERR_UNSUPPORTED, "k2w: %lld-row groups exceed the resident capacity", (long long)need);
}

}  // namespace pols
