// Dev tool (not part of the product): what a wave's chain of dependent v_mfma_f32_32x32x16_bf16 costs on gfx950, alone on its SIMD
// and beside a second wave, with one or two independent chains per wave and with vector work between the matrix instructions.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/diag/mfma_chain_probe.cpp -o mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS, int VALU>
__global__ __launch_bounds__(512) void probe(const float *in, float *out, unsigned long long *ticks, int iters)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[lane + i]; b[i] = (__bf16)in[64 + lane + i]; }
    v16f acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float m0 = in[lane], m1 = in[lane + 1], m2 = in[lane + 2], m3 = in[lane + 3];
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % CHAINS], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU; ++v) {       // VALU instructions per matrix instruction, independent of the chain
                if (v & 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(m1), "v"(m2));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m3) : "v"(m1), "v"(m2));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(acc[0]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = m0 + m3;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (lane == 0) ticks[4096 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime() - r0;
}

template <int CHAINS, int VALU>
static void run(const char *what, int waves, const float *in, float *out, unsigned long long *ticks)
{
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<CHAINS, VALU>), dim3(blocks), dim3(waves * 64), 0, 0, in, out, ticks, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<CHAINS, VALU>), dim3(blocks), dim3(waves * 64), 0, 0, in, out, ticks, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8192);
    hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
    const int nw = blocks * waves;
    double s = 0, r = 0, lo = 1e30, hi = 0; for (int i = 0; i < nw; ++i) { s += (double)h[i]; r += (double)h[4096 + i]; lo = std::min(lo, (double)h[i]); hi = std::max(hi, (double)h[i]); }
    const double per = s / nw / iters;             // ticks per 12 matrix instructions, per wave
    printf("%-30s %d wave(s)/SIMD chains %d VALU/MFMA %d : %6.1f ticks per 12 MFMA per wave (%.1f .. %.1f), %.3f GHz, event %.3f ms = %.0f TFLOP/s\n", what, waves / 4, CHAINS, VALU, per,
           lo / iters, hi / iters, s / r * 0.1, ms, 12.0 * 32768 * iters * nw / ms / 1e9);
}

int main()
{
    float *in, *out; unsigned long long *ticks;
    hipMalloc(&in, 4096);
    {   // operands: zeros (M2D_PROBE_ZERO=1) or random normal-ish values -- the clock the part holds depends on what the MFMAs multiply
        std::vector<float> hin(1024, 0.f);
        unsigned s = 12345u;
        if (!getenv("M2D_PROBE_ZERO")) for (auto &x : hin) { float t = 0.f; for (int i = 0; i < 12; ++i) { s = s * 1664525u + 1013904223u; t += ((s >> 8) & 0xffffff) / 16777216.0f - 0.5f; } x = t; }
        hipMemcpy(in, hin.data(), 4096, hipMemcpyHostToDevice);
    } hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 8192 * 8);
    for (int w : {4, 8}) {
        run<1, 0>("one chain", w, in, out, ticks);
        run<2, 0>("two chains", w, in, out, ticks);
        run<3, 0>("three chains", w, in, out, ticks);
        run<1, 1>("one chain + 1 VALU per MFMA", w, in, out, ticks);
        run<1, 2>("one chain + 2 VALU per MFMA", w, in, out, ticks);
        run<1, 4>("one chain + 4 VALU per MFMA", w, in, out, ticks);
        run<1, 6>("one chain + 6 VALU per MFMA", w, in, out, ticks);
        run<2, 4>("two chains + 4 VALU per MFMA", w, in, out, ticks);
    }
    return 0;
}
