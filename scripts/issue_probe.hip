// issue_probe.hip -- VALU issue rates on gfx950 in s_memtime ticks per instruction: independent f64 / f32 FMA streams, DPP moves,
// with 1, 2 or 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 256
template <int MODE>
__global__ void probe(double *out, unsigned long long *ticks, double seed) {
    const int lane = threadIdx.x & 63;
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + lane + i; f[i] = (float)a[i]; }
    const double b = 1.0000001, c = 1e-9;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(a[i]), "+v"(f[i]) :: "memory"); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 0" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(a[i]), "+v"(f[i]) :: "memory"); }
    if constexpr (MODE == 0) {           // 8 independent f64 FMA chains
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = fma(a[j], b, c);
    } else if constexpr (MODE == 1) {    // 8 independent f32 FMA chains
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(f[j], 1.0000001f, 1e-9f);
    } else if constexpr (MODE == 2) {    // f64 mul + add pairs, independent
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = a[j] * b + (j & 1 ? c : -c) * a[(j + 1) & 7];
    } else if constexpr (MODE == 3) {    // dpp mov pairs + f64 fma (the scan step's mix)
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long long bits = __double_as_longlong(a[j]);
                const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, 0x111, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), 0x111, 0xf, 0xf, false);
                a[j] = fma(b, __longlong_as_double(((long long)hi << 32) | (unsigned)lo), a[j]);
            }
    } else if constexpr (MODE == 5) {    // 2 independent f64 chains
#pragma unroll
        for (int i = 0; i < N * 4; ++i) { a[0] = fma(a[0], b, c); a[1] = fma(a[1], b, c); }
    } else if constexpr (MODE == 6) {    // 4 independent f64 chains
#pragma unroll
        for (int i = 0; i < N * 2; ++i) { a[0] = fma(a[0], b, c); a[1] = fma(a[1], b, c); a[2] = fma(a[2], b, c); a[3] = fma(a[3], b, c); }
    } else if constexpr (MODE == 4) {    // single dependent f64 chain
#pragma unroll
        for (int i = 0; i < N * 8; ++i) a[0] = fma(a[0], b, c);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { asm volatile("" :: "v"(a[i]), "v"(f[i]) : "memory"); }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> void run(const char *name, int per) {
    double *out; unsigned long long *t;
    hipMalloc(&out, 8 * 4096); hipMalloc(&t, 8 * 64);
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, t, 1.0);
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, out, t, 1.0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(threads / 64);
        hipMemcpy(h.data(), t, 8 * h.size(), hipMemcpyDeviceToHost);
        double m = 0; for (auto v : h) m += (double)v; m /= h.size();
        printf("%-28s waves/CU=%2d (%.1f per SIMD): %.2f ticks per instruction per wave -> SIMD issues one per %.2f ticks\n", name, threads / 64, threads / 256.0, m / (N * 8 * per), m / (N * 8 * per) / (threads >= 256 ? threads / 256.0 : 1.0));
    }
}
int main() {
    run<0>("f64 fma x8 independent", 1); run<1>("f32 fma x8 independent", 1); run<2>("f64 mul+fma x8", 2); run<3>("2 dpp mov + f64 fma", 3); run<4>("f64 fma dependent", 1); run<5>("f64 fma 2 chains", 1); run<6>("f64 fma 4 chains", 1);
    return 0;
}
