// probe_reread.hip -- does the second read of a long group come out of the Infinity Cache?
// One workgroup per segment of SEG bytes; the workgroup streams its segment once or twice (16-byte loads, sums kept live).
// The fused long-group kernel (Gram pass, solve, prediction pass in ONE workgroup) only pays if pass 2 is much cheaper than an HBM read.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_reread.hip -o gpurun_out/probe_reread && gpurun_out/probe_reread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int THREADS>
__global__ void __launch_bounds__(THREADS) reread_kernel(const float4 *__restrict__ src, float *__restrict__ out, long seg_vec, int passes, int wr) {
    const float4 *p = src + (long)blockIdx.x * seg_vec;
    float4 acc = {0, 0, 0, 0};
    for (int pass = 0; pass < passes; ++pass) {
        for (long i = threadIdx.x; i < seg_vec; i += 4 * THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { long j = i + (long)u * THREADS; v[u] = j < seg_vec ? p[j] : float4{0, 0, 0, 0}; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y * (pass + 1); acc.z += v[u].z; acc.w += v[u].w; }
        }
        __syncthreads();
    }
    if (wr || acc.x == 123.456f) out[(long)blockIdx.x * THREADS + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int THREADS>
static double run(const float4 *d, float *o, long seg_bytes, long n_segs, int passes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long seg_vec = seg_bytes / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(reread_kernel<THREADS>, dim3((unsigned)n_segs), dim3(THREADS), 0, 0, d, o, seg_vec, passes, 0);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(reread_kernel<THREADS>, dim3((unsigned)n_segs), dim3(THREADS), 0, 0, d, o, seg_vec, passes, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const long total = 4L << 30;
    float4 *d; float *o;
    CK(hipMalloc(&d, total)); CK(hipMalloc(&o, 64 << 20));
    CK(hipMemset(d, 0, total));
    std::printf("%10s %8s %7s | %9s %9s | %s\n", "seg", "threads", "segs", "1 pass", "2 passes", "TB/s of the bytes ONCE (1 pass / 2 passes), pass-2 cost as a fraction of pass 1");
    const long segs[] = {128 << 10, 256 << 10, 384 << 10, 512 << 10, 768 << 10, 1 << 20, 2 << 20, 4 << 20};
    for (long sb : segs) {
        const long n = total / sb;
        for (int th : {256, 512, 1024}) {
            double t1, t2;
            if (th == 256) { t1 = run<256>(d, o, sb, n, 1); t2 = run<256>(d, o, sb, n, 2); }
            else if (th == 512) { t1 = run<512>(d, o, sb, n, 1); t2 = run<512>(d, o, sb, n, 2); }
            else { t1 = run<1024>(d, o, sb, n, 1); t2 = run<1024>(d, o, sb, n, 2); }
            std::printf("%8ldKB %8d %7ld | %7.3fms %7.3fms | %5.2f %5.2f  %4.2f\n", sb >> 10, th, n, t1, t2, total / t1 * 1e-9, total / t2 * 1e-9, (t2 - t1) / t1);
        }
    }
    return 0;
}
