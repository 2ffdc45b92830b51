// Phase timing of m2d_mlp_mfma (dev tool; not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DM2D_MLP_DIAG=1 scripts/diag/mlp_diag.cpp -o mlp_diag
#include "../../foodrec_amd/csrc/m2d_catalogue.hip"
#include "../../foodrec_amd/csrc/m2d_mlp.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main()
{
    const int64_t U = 200000, I = 100000, B = 1 << 20;
    const int C = 4, E = 128;
    const size_t K = (C + 1) * E;
    m2d_engine h;
    h.U = U; h.I = I; h.C = C; h.E = E; h.a = 0.99f; h.b = 1.0f - 0.99f;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); h.num_cu = prop.multiProcessorCount;
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    auto dev = [&](size_t n, float scale) { std::vector<float> v(n); for (auto &x : v) x = rnd() * scale; float *d; hipMalloc(&d, n * 4); hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice); return d; };
    h.pm = dev(U * K, 1.f); h.re = dev(I * E, 1.f); h.ce = dev(C * E, 1.f);
    { std::vector<float> c(I * C, 1.0f); float *d; hipMalloc(&d, c.size() * 4); hipMemcpy(d, c.data(), c.size() * 4, hipMemcpyHostToDevice); h.dish_cats = d; }
    h.mlp_w1 = dev(K * 256, 0.1f); h.mlp_b1 = dev(256, 0.1f); h.mlp_w2 = dev(256 * 64, 0.1f); h.mlp_b2 = dev(64, 0.1f);
    h.mlp_w3 = dev(64, 0.1f); h.mlp_b3 = 0.f; h.mlp_h1 = 256; h.mlp_h2 = 64;
    hipMalloc(&h.err_dev, 16); hipMemset(h.err_dev, 0, 16);
    std::vector<int32_t> hu(B), hi(B);
    for (int64_t i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; hu[i] = (s >> 4) % U; s = s * 1664525u + 1013904223u; hi[i] = (s >> 4) % I; }
    int32_t *du, *di; float *out;
    hipMalloc(&du, B * 4); hipMalloc(&di, B * 4); hipMalloc(&out, B * 4);
    hipMemcpy(du, hu.data(), B * 4, hipMemcpyHostToDevice); hipMemcpy(di, hi.data(), B * 4, hipMemcpyHostToDevice);
#if M2D_MLP_DIAG
    unsigned long long *dbg; hipMalloc(&dbg, 4096 * 8 * 8); hipMemset(dbg, 0, 4096 * 8 * 8); g_m2d_mlp_diag_buffer = dbg;
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    m2d_launch_score_pairs_mlp(&h, du, di, B, out, nullptr);
    hipDeviceSynchronize();
    float best = 1e9f;
    std::vector<float> all;
    for (int it = 0; it < 25; ++it) {                       // this kernel's clock wanders: report best and median of 25
        hipEventRecord(e0);
        m2d_launch_score_pairs_mlp(&h, du, di, B, out, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        all.push_back(ms);
    }
    std::sort(all.begin(), all.end());
    printf("median of 25: %.3f ms   ", all[12]);
#if M2D_MLP_DIAG
    {
        std::vector<unsigned long long> hd(2048 * 8);
        hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
        double p = 0, l1 = 0, b = 0, l23 = 0, nt = 0, vm = 0, gw = 0, mz = 0;
        for (int w = 0; w < 2048; ++w) { p += hd[w*8]; l1 += hd[w*8+1]; b += hd[w*8+2]; l23 += hd[w*8+3]; nt += hd[w*8+4]; vm += hd[w*8+5]; gw += hd[w*8+6]; mz += hd[w*8+7]; }
        printf("per tile per wave (cycles): prologue %.0f  layer1 %.0f (10 chunks)  vmcnt-wait %.0f  barrier %.0f  layers2-3 %.0f  gather-wait %.0f  make_z %.0f\n", p / nt, l1 / nt, vm / nt, b / nt, l23 / nt, gw / nt, mz / nt);
    }
#endif
    const double fl = 2.0 * (K * 256 + 256 * 64 + 64) * (double)B;
    printf("M2D_MLP_DIAG=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", M2D_MLP_DIAG, best, fl / best / 1e9, fl / best / 1e9 / 1.573);
    return 0;
}
