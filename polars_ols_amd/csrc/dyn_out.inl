// dyn_out.inl -- coalesced copy-out of the row-parallel dynamic kernels (k3c_scan.hip, k4c_rolling.hip).
//
// A lane of those kernels owns 4 CONSECUTIVE rows, so its K coefficients x 4 rows are 4 K contiguous values of the row-major
// coefficient table -- and a 16-byte store per lane scatters a wave's store instruction over 64 different lines, 16 bytes each.
// Measured on the 1M-row sequence (k = 6, f64): the walk took 28.6 k ticks per tile with those stores and 5.3 k without them,
// and every other phase of the kernel doubled too (the partial-line writes clog the CU's memory pipe behind which the next
// tile's loads queue).  So the outputs of a wave's 256 rows go to LDS first -- into the slots the rows' inputs were parked in,
// value j of row 4 l + r at stage[(r (K + 1) + j) STRIDE + l], slot K the prediction -- and leave as whole lines: every store
// instruction of the wave writes 1 KiB of consecutive addresses.  STRIDE = 65: the transposed reads below then spread over the banks.
#pragma once
#include "common.hpp"

namespace pols {

constexpr int DYN_STAGE_STRIDE = 65;

// SLOTS = K + 1: slot K of a row holds its prediction; SLOTS = K: the predictions arrive in registers (pr[r]: row 4 l + r) and take the
// first four slots once the coefficients have left.
template <typename T, int K, int SLOTS = K + 1>
__device__ __forceinline__ void dyn_wave_copy_out(T *stage, const int lane, const int64_t wrow0, const int64_t N, T *coef, T *pred,
                                                  const int64_t lo, const T (&pr)[4]) {
    // rows [lo, N) of the wave's 256 are stored (N: the end of the frame or of a packed tile, lo: the first row of a packed tile)
    using V = typename Vec16<T>::type;
    constexpr int VN = Vec16<T>::N;
    auto at = [&](int row_local, int j) -> T { return stage[((row_local & 3) * SLOTS + j) * DYN_STAGE_STRIDE + (row_local >> 2)]; };
    const bool whole = wrow0 + 256 <= N && wrow0 >= lo;              // wave-uniform
    if (coef) {
        T *dst = coef + wrow0 * K;
#pragma unroll
        for (int i = 0; i < 4 * K / VN; ++i) {
            const int m0 = (i * 64 + lane) * VN;                     // first of this lane's VN consecutive values of the wave's 256 K
            V o;
#pragma unroll
            for (int e = 0; e < VN; ++e) {
                const int m = m0 + e, rl = m / K, j = m - rl * K;
                vset<T>(o, e, at(rl, j));
            }
            const int64_t ra = wrow0 + m0 / K, rb = wrow0 + (m0 + VN - 1) / K;   // rows of the first / last value
            if (whole || (ra >= lo && rb < N)) store_stream(reinterpret_cast<V *>(dst + m0), o);
            else {
#pragma unroll
                for (int e = 0; e < VN; ++e) {
                    const int64_t re = wrow0 + (m0 + e) / K;
                    if (re >= lo && re < N) dst[m0 + e] = vget<T>(o, e);
                }
            }
        }
    }
    if (pred) {
        T *dst = pred + wrow0;
        if constexpr (SLOTS == K) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[(r * SLOTS + K - 1) * DYN_STAGE_STRIDE + lane] = pr[r];   // (slot K - 1 of row r: read as at(row, K - 1))
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int i = 0; i < 4 / VN; ++i) {
            const int m0 = (i * 64 + lane) * VN;
            V o;
#pragma unroll
            for (int e = 0; e < VN; ++e) vset<T>(o, e, at(m0 + e, SLOTS == K ? K - 1 : K));
            if (whole || (wrow0 + m0 >= lo && wrow0 + m0 + VN - 1 < N)) store_stream(reinterpret_cast<V *>(dst + m0), o);
            else {
#pragma unroll
                for (int e = 0; e < VN; ++e)
                    if (wrow0 + m0 + e >= lo && wrow0 + m0 + e < N) dst[m0 + e] = vget<T>(o, e);
            }
        }
    }
}

template <typename T, int K>
__device__ __forceinline__ void dyn_wave_copy_out(T *stage, const int lane, const int64_t wrow0, const int64_t N, T *coef, T *pred) {
    const T none[4] = {};
    dyn_wave_copy_out<T, K, K + 1>(stage, lane, wrow0, N, coef, pred, 0, none);
}

}  // namespace pols
