// dyn_out_gather.inl -- copy-out of the row-parallel rolling kernel when it runs on the VALID rows of a frame with nulls ("drop" family,
// src/least_squares.rs:947-986: the window is a deque of the last `window` valid rows, a row left out repeats the last coefficients).
//
// The kernel's rows are the compacted rows c = 0 .. Nc - 1; src[c] is the frame row compacted row c came from.  A wave that holds the
// outputs of compacted rows [c0, c1) in its staging area (dyn_out.inl's layout) writes the FRAME rows [src[c0], src[c1]) -- from row 0
// for c0 == 0, to the frame's end for c1 == Nc -- so every frame row is written exactly once, by the wave that holds the last valid row
// at or before it:
//   coefficients  the last valid row's at or before the row IN ITS SEQUENCE, NaN before the sequence's first valid row (:864);
//   prediction    a valid row's own, NaN on a row left out (make_predictions(.., is_valid), src/expressions.rs:640-645).
// 64 frame rows per step, a lane per row: validity and sequence-start bytes -> ballots -> the staging slot of every row; the K coefficient
// values of the 64 rows leave as 64 K consecutive values (lane l stores values l, l + 64, ...), whole lines but for the two ends.
// This replaces the expansion pass (a read of the compacted table + a write of the frame's) of the three-pass form.
#pragma once
#include "dyn_out.inl"

namespace pols {

constexpr int DYN_GATHER_PRE = 5;      // steps whose validity / sequence-start bytes are requested with the run's rows (5 x 64 frame rows: 256 valid rows + their nulls)

// the frame rows a wave writes, and the bytes of the first DYN_GATHER_PRE steps -- requested EARLY (next to the kernel's own loads: first touched
// here, they come from HBM, and a step that waits for its bytes serialises five HBM latencies per wave: 0.52 against 0.28 ms for the kernel
// on 10 000 x 1 000 rows x 6 with 3 % nulls)
// (plain scalars and register arrays: a struct of them ended up in scratch)
__device__ __forceinline__ void dyn_gather_range(int64_t &ra, int64_t &rb, unsigned (&vb)[DYN_GATHER_PRE], unsigned (&sb)[DYN_GATHER_PRE], const int lane,
                                                 const int64_t c0, const int64_t c1, const int32_t *src, const int64_t Nc, const int64_t N,
                                                 const uint8_t *ovalid, const uint8_t *ostart) {
    ra = 0; rb = 0;
    if (c0 < c1) {                                                   // (wave-uniform)
        ra = c0 == 0 ? 0 : (int64_t)src[c0];
        rb = c1 < Nc ? (int64_t)src[c1] : N;
    }
#pragma unroll
    for (int c = 0; c < DYN_GATHER_PRE; ++c) {
        const int64_t row = ra + 64 * c + lane;
        const int64_t rc = row < rb ? row : (rb > 0 ? rb - 1 : 0);   // clamped: every lane loads, the value is masked at use
        vb[c] = ovalid[rc];
        sb[c] = ostart[rc];
    }
}

// one step: the 64 frame rows from `base` on (vb / sb: this lane's validity / sequence-start byte; count / last_valid / last_start: the state over [ra, base))
template <typename T, int K, int SLOTS>
__device__ __forceinline__ void dyn_gather_step(const T *stage, const int lane, const int64_t base, const int64_t rb, const unsigned vb, const unsigned sb,
                                                const int first_local, int &count, int64_t &last_valid, int64_t &last_start, T *coef, T *pred) {
    const T qnan = nan_if<T>(1u, T(0));
    const unsigned long long below = (2ull << lane) - 1ull;          // lanes at or below this one
    const bool in = base + lane < rb;
    const unsigned long long vm = __ballot(in && vb != 0), sm = __ballot(in && sb != 0);
    const bool mine = in && vb != 0;
    const unsigned long long vl = vm & below, sl = sm & below;
    const int cnt = count + __popcll(vl);
    const int64_t lv = vl ? base + (63 - __clzll(vl)) : last_valid;
    const int64_t ls = sl ? base + (63 - __clzll(sl)) : last_start;
    const bool none = cnt == 0 || ls > lv;                           // no valid row yet in the frame / in this row's sequence
    const int slot = none ? -1 : first_local + cnt;
    if (coef) {
        T *dst = coef + base * K;
        const int64_t nvals = (rb - base < 64 ? rb - base : 64) * K;
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int m = q * 64 + lane, row = m / K, col = m - row * K;
            const int sr = __shfl(slot, row);
            const int srr = sr < 0 ? 0 : sr;
            const T v0 = stage[((srr & 3) * SLOTS + col) * DYN_STAGE_STRIDE + (srr >> 2)];
            if (m < nvals) dst[m] = sr < 0 ? qnan : v0;
        }
    }
    if (pred && in) {
        const int s0 = slot < 0 ? 0 : slot;
        const T p0 = stage[((s0 & 3) * SLOTS + (SLOTS == K ? K - 1 : K)) * DYN_STAGE_STRIDE + (s0 >> 2)];
        pred[base + lane] = (mine && slot >= 0) ? p0 : qnan;
    }
    count += __popcll(vm);
    last_valid = vm ? base + (63 - __clzll(vm)) : last_valid;
    last_start = sm ? base + (63 - __clzll(sm)) : last_start;
}

template <typename T, int K, int SLOTS = K + 1>
__device__ __forceinline__ void dyn_wave_copy_out_gather(const T *stage, const int lane, const int64_t wrow0, const int64_t c0, const int64_t c1,
                                                         const int64_t ra, const int64_t rb, const unsigned (&gvb)[DYN_GATHER_PRE],
                                                         const unsigned (&gsb)[DYN_GATHER_PRE], const uint8_t *ovalid, const uint8_t *ostart,
                                                         T *coef, T *pred) {
    if (c0 >= c1) return;                                            // (wave-uniform)
    int count = 0;                                                   // valid rows of [ra, this step)
    int64_t last_valid = -1, last_start = -1;                        // frame rows, over [ra, this step)
    const int first_local = (int)(c0 - wrow0) - 1;                   // staging slot of compacted row c0, less one
#pragma unroll
    for (int c = 0; c < DYN_GATHER_PRE; ++c) {
        if (ra + 64 * c < rb)                                        // (wave-uniform)
            dyn_gather_step<T, K, SLOTS>(stage, lane, ra + 64 * c, rb, gvb[c], gsb[c], first_local, count, last_valid, last_start, coef, pred);
    }
    for (int64_t base = ra + 64 * DYN_GATHER_PRE; base < rb; base += 64) {   // a long run of nulls behind the wave's rows: a step at a time
        unsigned vb = 0, sb = 0;
        if (base + lane < rb) { vb = ovalid[base + lane]; sb = ostart[base + lane]; }
        dyn_gather_step<T, K, SLOTS>(stage, lane, base, rb, vb, sb, first_local, count, last_valid, last_start, coef, pred);
    }
}

}  // namespace pols
