// probe_matrix.hip -- what does the memory system of an MI355X admit for the static kernels' traffic (k + 1 column streams read down the
// row axis, one written), and which of the things K1 holds FIXED decides it?
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_matrix.hip -o scripts/probe_matrix.bin && scripts/probe_matrix.bin > profiles/r05_probe_matrix.txt
// The two probes of rounds 3-4 (csrc/probe.hip) both stream 4 KB per column per workgroup with consecutive pieces on consecutive
// workgroup ids -- i.e. dealt round-robin over the 8 XCDs, exactly K1's own mapping -- so they could not say anything about piece size,
// XCD locality or the read / write mix.  This program varies those, on the headline frame's shape (nine f32 column streams of 10^7 rows,
// one output stream; three rotated frames = 1.2 GB so that no launch finds its input in the 256 MB Infinity Cache):
//   map    rr  : workgroup b -> piece b (round-robin over XCDs)          xcd : b -> (b % 8) * ceil(P / 8) + b / 8 (each XCD's L2 streams one
//                                                                              contiguous eighth of every column)
//   piece  4 / 16 / 64 KB per column per workgroup (1 / 4 / 16 sub-pieces of 256 lanes x 16 bytes, the next sub-piece's loads in flight
//          before the current one is consumed)
//   rows   1024 (pieces are whole 128-byte lines) or 1000 (the headline's groups: 4 000-byte slices that straddle lines, 250 of 256 lanes)
//   mix    9R+1W, 9R (no store), 1W (no loads)
//   loads  plain, nt, LDS-DMA (global_load_lds_dwordx4, read back with ds_read_b128)
//   stores plain / nt per sub-piece, or held in registers and written in one burst at the end of the piece ("burst")
// One line per cell: microseconds per launch (wall clock over back-to-back launches) and TB/s of algorithmic bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Cols { const float *x[9]; float *out; };

using F4 = __attribute__((ext_vector_type(4))) float;

template <int LOADK> __device__ __forceinline__ F4 ld(const float *p) {
    if constexpr (LOADK == 1) return __builtin_nontemporal_load(reinterpret_cast<const F4 *>(p));
    else return *reinterpret_cast<const F4 *>(p);
}
template <int STOREK> __device__ __forceinline__ void st(float *p, F4 v) {
    if constexpr (STOREK == 1) __builtin_nontemporal_store(v, reinterpret_cast<F4 *>(p));
    else *reinterpret_cast<F4 *>(p) = v;
}

// MIX 0: 9R + 1W, 1: 9R, 2: 1W.  LOADK 0 plain, 1 nt, 2 LDS-DMA.  STOREK 0 plain, 1 nt, 2 nt burst at the end of the piece.
template <int MIX, int LOADK, int STOREK, int SUBP>
__global__ void __launch_bounds__(256) probe(const Cols c, const int rows_per_sub, const long n_pieces, const long xcd_chunk, float *sink) {
    const long b = blockIdx.x;
    const long piece = xcd_chunk ? (b & 7) * xcd_chunk + (b >> 3) : b;
    if (piece >= n_pieces) return;
    const int tid = threadIdx.x;
    const bool act = tid * 4 < rows_per_sub;
    const long row0 = piece * SUBP * (long)rows_per_sub + (act ? tid * 4 : 0);
    constexpr int NBUF = SUBP > 1 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float stage[LOADK == 2 ? NBUF * 9 * 1024 : 4];
    F4 acc = {0, 0, 0, 0};
    F4 held[STOREK == 2 ? SUBP : 1];

    if constexpr (MIX == 2) {
#pragma unroll
        for (int s = 0; s < SUBP; ++s) {
            const F4 o = {(float)tid, (float)s, 1.f, 2.f};
            if (act) st<STOREK == 0 ? 0 : 1>(c.out + row0 + (long)s * rows_per_sub, o);
        }
        return;
    } else if constexpr (LOADK == 2) {
        // every wave DMAs 64 lanes x 16 bytes of each column into its own 1 KiB slots (9 per buffer) and reads them back itself: no barrier
        const int wave = tid >> 6, lane = tid & 63;
        auto issue = [&](int s, int buf) {
#pragma unroll
            for (int j = 0; j < 9; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(c.x[j] + row0 + (long)s * rows_per_sub),
                                                 (__attribute__((address_space(3))) void *)(stage + ((buf * 9 + j) * 4 + wave) * 256), 16, 0, 0);
        };
        issue(0, 0);
#pragma unroll
        for (int s = 0; s < SUBP; ++s) {
            if (s + 1 < SUBP) { issue(s + 1, (s + 1) & 1); asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            F4 t = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 9; ++j) t += *reinterpret_cast<const F4 *>(stage + (((s & 1) * 9 + j) * 4 + wave) * 256 + lane * 4);
            if constexpr (MIX == 0) {
                if constexpr (STOREK == 2) held[s] = t;
                else if (act) st<STOREK>(c.out + row0 + (long)s * rows_per_sub, t);
            } else acc += t;
            if (s + 2 < SUBP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is re-filled two steps on
        }
    } else {
        F4 cur[9], nxt[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) cur[j] = ld<LOADK>(c.x[j] + row0);
#pragma unroll
        for (int s = 0; s < SUBP; ++s) {
            if (s + 1 < SUBP) {
#pragma unroll
                for (int j = 0; j < 9; ++j) nxt[j] = ld<LOADK>(c.x[j] + row0 + (long)(s + 1) * rows_per_sub);
            }
            F4 t = cur[0];
#pragma unroll
            for (int j = 1; j < 9; ++j) t += cur[j];
            if constexpr (MIX == 0) {
                if constexpr (STOREK == 2) held[s] = t;
                else if (act) st<STOREK>(c.out + row0 + (long)s * rows_per_sub, t);
            } else acc += t;
            if (s + 1 < SUBP) {
#pragma unroll
                for (int j = 0; j < 9; ++j) cur[j] = nxt[j];
            }
        }
    }
    if constexpr (MIX == 0 && STOREK == 2) {
#pragma unroll
        for (int s = 0; s < SUBP; ++s)
            if (act) st<1>(c.out + row0 + (long)s * rows_per_sub, held[s]);
    }
    if constexpr (MIX == 1) {
        if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
    }
}

struct Cell { int mix, loadk, storek, subp; };
using Kern = void (*)(const Cols, const int, const long, const long, float *);

template <int MIX, int LOADK, int STOREK>
static Kern pick_subp(int subp) {
    switch (subp) {
        case 1: return probe<MIX, LOADK, STOREK, 1>;
        case 4: return probe<MIX, LOADK, STOREK, 4>;
        default: return probe<MIX, LOADK, STOREK, 16>;
    }
}
template <int MIX, int LOADK>
static Kern pick_store(int storek, int subp) {
    switch (storek) {
        case 0: return pick_subp<MIX, LOADK, 0>(subp);
        case 1: return pick_subp<MIX, LOADK, 1>(subp);
        default: return pick_subp<MIX, LOADK, 2>(subp);
    }
}
template <int MIX>
static Kern pick_load(int loadk, int storek, int subp) {
    switch (loadk) {
        case 0: return pick_store<MIX, 0>(storek, subp);
        case 1: return pick_store<MIX, 1>(storek, subp);
        default: return pick_store<MIX, 2>(storek, subp);
    }
}
static Kern pick(const Cell &c) {
    switch (c.mix) {
        case 0: return pick_load<0>(c.loadk, c.storek, c.subp);
        case 1: return pick_load<1>(c.loadk, 0, c.subp);
        default: return pick_store<2, 0>(c.storek, c.subp);
    }
}

int main(int argc, char **argv) {
    const long N = 10000000;
    const int FRAMES = 3, REPS = argc > 1 ? std::atoi(argv[1]) : 60;
    // argv[2] = bytes of SKEW between consecutive columns of a frame (0: one hipMalloc per column -- 2 MiB-aligned bases, 40 MiB apart, what
    // torch's allocator hands out too; > 0: the ten columns of a frame carved out of ONE allocation, column j at j * (N * 4 + skew)): do the
    // ten streams collide in the HBM channel / bank mapping when their bases are congruent modulo a large power of two?
    const long SKEW = argc > 2 ? std::atol(argv[2]) : 0;
    Cols fr[FRAMES];
    for (int f = 0; f < FRAMES; ++f) {
        if (SKEW > 0) {
            char *big; const size_t stride = (size_t)N * 4 + (size_t)SKEW, total = stride * 10 + (1 << 20);
            CK(hipMalloc(&big, total)); CK(hipMemset(big, 0x3c, total));
            for (int j = 0; j < 9; ++j) fr[f].x[j] = reinterpret_cast<const float *>(big + stride * j);
            fr[f].out = reinterpret_cast<float *>(big + stride * 9);
            continue;
        }
        for (int j = 0; j < 9; ++j) { float *p; CK(hipMalloc(&p, (N + 4096) * 4)); CK(hipMemset(p, 0x3c, (N + 4096) * 4)); fr[f].x[j] = p; }
        CK(hipMalloc(&fr[f].out, (N + 4096) * 4));
    }
    printf("# column base skew: %ld bytes%s\n", SKEW, SKEW ? " (one allocation per frame)" : " (one hipMalloc per column)");
    for (int j = 0; j < 3; ++j) printf("# frame 0 column %d at %p\n", j, (const void *)fr[0].x[j]);
    const bool QUICK = argc > 3;      // argv[3]: only the headline's cells (9R+1W and 9R, 4 / 16 KB, rows = 1000, nt loads)
    float *sink; CK(hipMalloc(&sink, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *mixn[] = {"9R+1W", "9R", "1W"}, *ldn[] = {"plain", "nt", "lds-dma"}, *stn[] = {"plain", "nt", "burst"};
    printf("# probe matrix: 9 f32 column streams of 1e7 rows (+ 1 written), three rotated frames, %d launches per cell\n", REPS);
    printf("# %-6s %-4s %5s %5s %-8s %-6s %9s %8s %8s\n", "mix", "map", "piece", "rows", "loads", "stores", "us/launch", "TB/s", "blocks");
    double best[3] = {0, 0, 0};
    char bestn[3][128] = {"", "", ""};
    for (int mix = 0; mix < 3; ++mix)
        for (int rows : {1024, 1000})
            for (int subp : {1, 4, 16})
                for (int map = 0; map < 2; ++map)
                    for (int loadk = 0; loadk < 3; ++loadk)
                        for (int storek = 0; storek < 3; ++storek) {
                            if (mix == 1 && storek) continue;
                            if (mix == 2 && loadk) continue;
                            if (storek == 2 && subp == 1) continue;       // burst == nt with one sub-piece
                            if (QUICK && (mix == 2 || rows != 1000 || subp == 16 || loadk != 1 || map == 1 || storek == 0)) continue;
                            const Cell cell{mix, loadk, storek, subp};
                            Kern k = pick(cell);
                            const long subs = N / rows, n_pieces = subs / subp;      // whole pieces only (the tail is outside the byte count too)
                            const long chunk = map ? (n_pieces + 7) / 8 : 0;
                            const long blocks = map ? chunk * 8 : n_pieces;
                            const double bytes = (double)n_pieces * subp * rows * 4.0 * ((mix != 2 ? 9 : 0) + (mix != 1 ? 1 : 0));
                            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, s, fr[w % FRAMES], rows, n_pieces, chunk, sink);
                            CK(hipStreamSynchronize(s));
                            CK(hipEventRecord(e0, s));
                            for (int r = 0; r < REPS; ++r) hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), 0, s, fr[r % FRAMES], rows, n_pieces, chunk, sink);
                            CK(hipEventRecord(e1, s));
                            CK(hipEventSynchronize(e1));
                            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                            const double us = ms * 1000.0 / REPS, tbs = bytes / (us * 1e-6) / 1e12;
                            printf("  %-6s %-4s %4dK %5d %-8s %-6s %9.2f %8.3f %8ld\n", mixn[mix], map ? "xcd" : "rr", subp * 4, rows, mix == 2 ? "-" : ldn[loadk],
                                   mix == 1 ? "-" : stn[storek], us, tbs, blocks);
                            if (tbs > best[mix]) {
                                best[mix] = tbs;
                                snprintf(bestn[mix], sizeof(bestn[mix]), "%s %dK rows=%d loads=%s stores=%s", map ? "xcd" : "rr", subp * 4, rows, ldn[loadk], stn[storek]);
                            }
                            fflush(stdout);
                        }
    for (int m = 0; m < 3; ++m) printf("# best %-6s %.3f TB/s  (%s)\n", mixn[m], best[m], bestn[m]);
    return 0;
}
