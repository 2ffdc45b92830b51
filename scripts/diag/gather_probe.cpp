// Random-piece gather bandwidth on MI355X (dev tool): every wave keeps DEPTH x 64 x 16 B in flight, each group of
// PIECE/16 lanes reading one contiguous PIECE-byte piece of a random row.  Answers: how much of the 8 TB/s do
// 128 B / 256 B / 512 B / whole-row random pieces get?   hipcc -O3 --offload-arch=gfx950 gather_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void probe(const v4f *tab, int64_t rows, int row_f4, int piece_f4, int iters, float *sink)
{
    const int lane = threadIdx.x & 63;
    uint32_t s = (blockIdx.x * 512 + threadIdx.x) / piece_f4 * 2654435761u + 12345u;   // one rng per lane group
    v4f acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        v4f v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            s = s * 1664525u + 1013904223u;
            const int64_t row = (int64_t)((s >> 4) % (uint32_t)rows);
            s = s * 1664525u + 1013904223u;
            const int piece = (int)((s >> 8) % (uint32_t)(row_f4 / piece_f4));
            v[d] = tab[row * row_f4 + piece * piece_f4 + (lane % piece_f4)];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

// The MLP kernel's pattern: 32 rows per wave, one 128-B line per row per step, lane (pl, h) takes 4 x 16 B of it
// over 4 instructions.  MODE 0: f4 = 4 (i >> 1) + (i & 1) + 2 h (every instruction touches all 32 lines, 32 B each);
// MODE 1: lanes 8 r + c read row 8 i + r whole (one instruction per line).
template <int MODE, int STEPS>
__global__ __launch_bounds__(512) void probe_mlp(const v4f *tab, int64_t rows, int row_f4, int iters, float *sink)
{
    const int lane = threadIdx.x & 63, pl = lane & 31, h = lane >> 5;
    uint32_t s0 = (blockIdx.x * 8 + (threadIdx.x >> 6)) * 2654435761u + 777u;
    v4f acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // 32 random rows per wave per iteration (the same for every lane: derived from s0, it and the row slot)
        auto rowof = [&](int slot) { uint32_t x = (s0 + it * 40503u + slot * 9176u) * 2246822519u; x ^= x >> 15; x *= 3266489917u; x ^= x >> 13; return (int64_t)(x % (uint32_t)rows); };
        v4f v[STEPS][4];
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) v[st][i] = tab[rowof(pl) * row_f4 + st * 8 + 4 * (i >> 1) + (i & 1) + 2 * h];
                else v[st][i] = tab[rowof(8 * i + (lane >> 3)) * row_f4 + st * 8 + (lane & 7)];
            }
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += v[st][i];
    }
    if (acc.x == 123.456f) sink[0] = acc.y + acc.z + acc.w;
}

int main()
{
    const int64_t rows = 1000000; const int row_f4 = 640;      // 2560-B rows, 2.56 GB
    v4f *tab; hipMalloc(&tab, rows * row_f4 * 16); hipMemset(tab, 0, rows * row_f4 * 16);
    float *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 64, grid = 256 * 4;
    for (int mode = 0; mode < 2; ++mode)
        for (int steps : {1, 2, 4}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
#define PM(M, S) if (mode == M && steps == S) hipLaunchKernelGGL((probe_mlp<M, S>), dim3(grid), dim3(512), 0, 0, tab, rows, row_f4, iters, sink);
                PM(0, 1) PM(0, 2) PM(0, 4) PM(1, 1) PM(1, 2) PM(1, 4)
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)grid * 512 * 16 * 4 * steps * iters;
            printf("mlp pattern mode %d (%s)  %d x 128 B per row in flight  %.2f TB/s\n", mode, mode ? "line per 8 lanes" : "4 touches per line", steps, bytes / best / 1e9);
        }
    for (int piece_f4 : {4, 8, 16, 32, 64}) {
        for (int depth : {4, 8, 16}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (depth == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(512), 0, 0, tab, rows, row_f4, piece_f4, iters, sink);
                if (depth == 8) hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(512), 0, 0, tab, rows, row_f4, piece_f4, iters, sink);
                if (depth == 16) hipLaunchKernelGGL(probe<16>, dim3(grid), dim3(512), 0, 0, tab, rows, row_f4, piece_f4, iters, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double bytes = (double)grid * 512 * 16 * depth * iters;
            printf("piece %4d B  depth %2d (%5.1f KB in flight / CU at 2 blocks)  %.2f TB/s\n", piece_f4 * 16, depth,
                   2 * 512 * 16.0 * depth / 1024, bytes / best / 1e9);
        }
    }
    return 0;
}
