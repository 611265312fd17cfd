// lat_probe.hip -- single-wave dependent-chain latencies on gfx950 (s_memtime ticks per operation): what a serial solver phase
// (Cholesky chain, coordinate descent) pays per step when nothing else hides it.  hipcc --offload-arch=gfx950 -O3 lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 512

template <int MODE>
__global__ void __launch_bounds__(64) probe(double *out, unsigned long long *ticks, double seed, int zero) {
    const int lane = threadIdx.x;
    double a = seed + lane, b = 1.0000001, c = 1e-9;
    float af = (float)a, bf = 1.0000001f, cf = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MODE == 0) {            // dependent v_fma_f64
#pragma unroll
        for (int i = 0; i < N; ++i) a = fma(a, b, c);
    } else if constexpr (MODE == 1) {     // dependent v_fma_f32
#pragma unroll
        for (int i = 0; i < N; ++i) af = fmaf(af, bf, cf);
    } else if constexpr (MODE == 2) {     // dependent v_add_f64 / v_max_f64 pairs
#pragma unroll
        for (int i = 0; i < N / 2; ++i) { a = a + c; a = fmax(a, b); }
    } else if constexpr (MODE == 3) {     // v_fma_f64 -> v_readlane (lo, hi) -> v_fma_f64 with the SGPR pair
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const long long bits = __double_as_longlong(a);
            const int lo = __builtin_amdgcn_readlane((int)bits, i & 15), hi = __builtin_amdgcn_readlane((int)(bits >> 32), i & 15);
            const double s = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
            a = fma(s, c, a);
        }
    } else if constexpr (MODE == 4) {     // the same through __shfl (ds_bpermute)
#pragma unroll
        for (int i = 0; i < N; ++i) { const double s = __shfl(a, i & 15); a = fma(s, c, a); }
    } else if constexpr (MODE == 5) {     // rolled loop: one dependent f64 fma + loop overhead (taken branch per iteration)
#pragma unroll 1
        for (int i = 0; i < N + zero; ++i) a = fma(a, b, c);
    } else if constexpr (MODE == 6) {     // DPP row all-reduce of an f64 (4 steps) feeding the next
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            double v = a;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            a = v * c + 1.0;
        }
    } else if constexpr (MODE == 7) {     // LDS round trip: ds_write_b64 -> ds_read_b64 (other lane) -> fma
        __shared__ double buf[64];
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            buf[lane] = a;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            a = fma(buf[(lane + 1) & 63], c, a);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else if constexpr (MODE == 8) {     // dependent v_mfma_f64_16x16x4 on one accumulator
        using acc_t = __attribute__((ext_vector_type(4))) double;
        acc_t acc = {a, a, a, a};
#pragma unroll
        for (int i = 0; i < N / 4; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
        a = acc[0] + acc[1] + acc[2] + acc[3];
    } else if constexpr (MODE == 9) {     // two independent v_mfma_f64_16x16x4 accumulators, alternating
        using acc_t = __attribute__((ext_vector_type(4))) double;
        acc_t acc0 = {a, a, a, a}, acc1 = {b, b, b, b};
#pragma unroll
        for (int i = 0; i < N / 8; ++i) { acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, b, acc1, 0, 0, 0); }
        a = acc0[0] + acc1[1];
    } else if constexpr (MODE == 10) {    // dependent f64 divide
#pragma unroll
        for (int i = 0; i < N / 8; ++i) a = b / a;
    } else if constexpr (MODE == 11) {    // dependent f64 sqrt
#pragma unroll
        for (int i = 0; i < N / 8; ++i) a = sqrt(a + b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a + af;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; unsigned long long *ticks;
    hipMalloc(&out, 64 * 8 * 8); hipMalloc(&ticks, 8 * 8);
    const char *names[] = {"v_fma_f64 dependent", "v_fma_f32 dependent", "v_add_f64 + v_max_f64 dependent", "fma -> readlane x2 -> fma (f64)",
                           "fma -> __shfl (bpermute) -> fma (f64)", "rolled loop: fma_f64 + branch", "4-step shfl_xor row all-reduce (f64) + fma",
                           "LDS write -> read other lane -> fma", "v_mfma_f64_16x16x4 dependent", "v_mfma_f64_16x16x4 two accumulators",
                           "f64 divide dependent", "f64 sqrt dependent"};
    const int ops[] = {N, N, N, N, N, N, N / 4, N / 4, N / 4, N / 4, N / 8, N / 8};
    for (int m = 0; m < 12; ++m) {
        unsigned long long best = ~0ULL;
        for (int rep = 0; rep < 5; ++rep) {
            switch (m) {
#define CASE(i) case i: hipLaunchKernelGGL(probe<i>, dim3(1), dim3(64), 0, 0, out, ticks, 1.5, 0); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11)
            }
            hipDeviceSynchronize();
            unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            if (t < best) best = t;
        }
        printf("%-48s %8.1f ticks per op  (%llu ticks / %d)\n", names[m], (double)best / ops[m], best, ops[m]);
    }
    // s_memtime tick rate against the wall clock
    return 0;
}
