// bw_probe.hip -- what HBM bandwidth does the K1 access pattern admit with the math removed?
//   hipcc --offload-arch=gfx950 -O3 scripts/bw_probe.hip -o gpurun_out/bw_probe && gpurun_out/bw_probe
// Pattern A: plain float4 copy (read N, write N)                      -- the guide's "achievable" reference
// Pattern B: K column streams read, 1 written, wave-per-group (1000 rows = 4 float4 per lane per column)
// Pattern C: same with a 256-thread team per group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

// read-only sweep: every block sums a contiguous slice; the result is written only if it is "impossible" (keeps the loads alive)
__global__ void read4(const float4 *__restrict__ in, float *sink, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = {0, 0, 0, 0};
    for (; i < n; i += stride) { const float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// per-workgroup re-read: a block streams its own SEG bytes (pass 1), idles `spin` iterations, streams the same bytes again
// (pass 2): is the second pass served on-die (L2 / Infinity Cache) when the whole chip is doing the same thing?
__global__ void __launch_bounds__(256) reread(const float4 *__restrict__ in, float *sink, int seg_vec, int spin, int passes) {
    const float4 *p = in + (size_t)blockIdx.x * seg_vec;
    float4 acc = {0, 0, 0, 0};
    for (int ps = 0; ps < passes; ++ps) {
        for (int i = threadIdx.x; i < seg_vec; i += 256) { const float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        for (int k = 0; k < spin; ++k) __builtin_amdgcn_s_sleep(8);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

struct Cols { const float *x[9]; float *out; };

template <int TEAM, int RC, int NT = 0>   // NT: 1 = nontemporal stores, 2 = nontemporal loads too
__global__ void __launch_bounds__(256) stream9(Cols c, int rows, int groups) {
    const int tid = threadIdx.x % TEAM;
    const long g = (long)blockIdx.x * (256 / TEAM) + threadIdx.x / TEAM;
    if (g >= groups) return;
    const long base = g * rows;
    float4 v[RC][9];
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const long r = (long)(rc * TEAM + tid) * 4;
        if (r < rows) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const float4 *src = reinterpret_cast<const float4 *>(c.x[j] + base + r);
                if (NT == 2) { v[rc][j].x = __builtin_nontemporal_load(&src->x); v[rc][j].y = __builtin_nontemporal_load(&src->y);
                               v[rc][j].z = __builtin_nontemporal_load(&src->z); v[rc][j].w = __builtin_nontemporal_load(&src->w); }
                else v[rc][j] = *src;
            }
        }
    }
#pragma unroll
    for (int rc = 0; rc < RC; ++rc) {
        const long r = (long)(rc * TEAM + tid) * 4;
        if (r < rows) {
            float4 s = v[rc][0];
#pragma unroll
            for (int j = 1; j < 9; ++j) { s.x += v[rc][j].x; s.y += v[rc][j].y; s.z += v[rc][j].z; s.w += v[rc][j].w; }
            float4 *dst = reinterpret_cast<float4 *>(c.out + base + r);
            if (NT >= 1) { __builtin_nontemporal_store(s.x, &dst->x); __builtin_nontemporal_store(s.y, &dst->y);
                           __builtin_nontemporal_store(s.z, &dst->z); __builtin_nontemporal_store(s.w, &dst->w); }
            else *dst = s;
        }
    }
}

int main() {
    const int groups = 10000, rows = 1000;
    const size_t N = (size_t)groups * rows;
    Cols c;
    for (int j = 0; j < 9; ++j) { float *p; CK(hipMalloc(&p, N * 4)); CK(hipMemset(p, 0x3c, N * 4)); c.x[j] = p; }
    CK(hipMalloc(&c.out, N * 4));
    float4 *big_in, *big_out;
    const size_t NB = (size_t)200 * 1000 * 1000 / 16;   // 200 MB
    CK(hipMalloc(&big_in, NB * 16)); CK(hipMalloc(&big_out, NB * 16)); CK(hipMemset(big_in, 1, NB * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char *name, double bytes) {
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0, 0);
        const int reps = 50;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.2f us  %7.0f GB/s\n", name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    };
    timeit([&] { copy4<<<2048, 256>>>(big_in, big_out, NB); }, "A copy float4 200MB r + 200MB w", 2.0 * NB * 16);
    timeit([&] { copy4<<<8192, 256>>>(big_in, big_out, NB); }, "A' copy float4 (8192 blocks)", 2.0 * NB * 16);
    const double bytes9 = 10.0 * N * 4;
    timeit([&] { stream9<64, 4><<<groups / 4, 256>>>(c, rows, groups); }, "B 9r+1w wave-per-group rc4", bytes9);
    timeit([&] { stream9<256, 1><<<groups, 256>>>(c, rows, groups); }, "C 9r+1w team256 rc1", bytes9);
    timeit([&] { stream9<128, 2><<<groups / 2, 256>>>(c, rows, groups); }, "D 9r+1w team128 rc2", bytes9);
    timeit([&] { stream9<64, 4, 1><<<groups / 4, 256>>>(c, rows, groups); }, "B1 wave-per-group, nt stores", bytes9);
    timeit([&] { stream9<64, 4, 2><<<groups / 4, 256>>>(c, rows, groups); }, "B2 wave-per-group, nt loads+stores", bytes9);
    timeit([&] { stream9<256, 1, 1><<<groups, 256>>>(c, rows, groups); }, "C1 team256, nt stores", bytes9);
    // ---- does a re-read come from the Infinity Cache?
    float *sink; CK(hipMalloc(&sink, 64));
    float4 *huge; const size_t NH = (size_t)2000 * 1000 * 1000 / 16;   // 2 GB
    CK(hipMalloc(&huge, NH * 16)); CK(hipMemset(huge, 1, NH * 16));
    for (size_t mb : {32, 64, 128, 192, 256, 400, 1000, 2000}) {
        const size_t nv = mb * 1000 * 1000 / 16;
        char name[64]; snprintf(name, sizeof(name), "E read-only sweep of %4zu MB", mb);
        timeit([&] { read4<<<4096, 256>>>(huge, sink, nv); }, name, (double)nv * 16);
    }
    // 88 KB and 272 KB segments (cfg3 / cfg5 groups), 2 GB total so that pass 1 is an HBM stream
    for (int seg_kb : {88, 272}) {
        const int seg_vec = seg_kb * 1024 / 16;
        const int blocks = (int)(NH / seg_vec);
        for (int passes : {1, 2}) {
            char name[64]; snprintf(name, sizeof(name), "F %3d KB per block, %d pass(es)", seg_kb, passes);
            timeit([&] { reread<<<blocks, 256>>>(huge, sink, seg_vec, 0, passes); }, name, (double)blocks * seg_vec * 16);
        }
    }
    return 0;
}
