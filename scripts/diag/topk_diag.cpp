// Ablation timing of m2d_topk_mfma (dev tool; not part of the product).  Build one binary per mask:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DM2D_DIAG=<mask> scripts/diag/topk_diag.cpp -o topk_diag_<mask>
// (a unity build of the retrieval units: -DM2D_DIAG reaches every kernel)
#include "../../foodrec_amd/csrc/m2d_catalogue_dense.hip"
#include "../../foodrec_amd/csrc/m2d_catalogue_plan.hip"
#include "../../foodrec_amd/csrc/m2d_catalogue_scan_f32.hip"
#include "../../foodrec_amd/csrc/m2d_catalogue_scan_bf16.hip"
#include "../../foodrec_amd/csrc/m2d_catalogue_merge.hip"
#include "../../foodrec_amd/csrc/m2d_catalogue_repair.hip"

int m2d_ensure_finite_scan(m2d_engine *, hipStream_t) { return M2D_OK; }   // (m2d_abi.hip is not part of this binary)

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

int main(int argc, char **argv)
{
    const int64_t U = getenv("M2D_DIAG_USERS") ? atoll(getenv("M2D_DIAG_USERS")) : 65536, I = getenv("M2D_DIAG_DISHES") ? atoll(getenv("M2D_DIAG_DISHES")) : 100000;
    const int C = 4, E = 64, k = 10;
    m2d_engine h;
    h.U = U; h.I = I; h.C = C; h.E = E; h.a = 0.99f; h.b = 1.0f - 0.99f;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    h.num_cu = prop.multiProcessorCount;
    const size_t K = (C + 1) * E;
    std::vector<float> pm(U * K), re(I * E), ce(C * E), cats(I * C, 1.0f);
    if (getenv("M2D_DIAG_PRUNE")) h.opt_topk_prune = atoi(getenv("M2D_DIAG_PRUNE"));
    if (getenv("M2D_DIAG_FORM")) h.opt_topk_form = atoi(getenv("M2D_DIAG_FORM"));      // 3 / 4: the hi x hi first form always / never
    if (argc > 1) h.opt_variant = atoi(argv[1]);
    unsigned s = 1;
    auto rnd1 = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.0f - 0.5f; };   // 24 bits: 16-bit draws tie often enough to send thousands of users through the tie repair
    // ~N(0, 1/E): twelve uniforms, scaled (the benchmark's tables); uniform non-empty category subsets
    auto rnd = [&]() { float t = 0.f; for (int i = 0; i < 12; ++i) t += rnd1(); return t * 0.125f; };
    if (getenv("M2D_DIAG_PATTERNS")) for (int64_t d = 0; d < I; ++d) { s = s * 1664525u + 1013904223u; const int pt = 1 + (int)((s >> 10) % 15u); for (int c = 0; c < C; ++c) cats[d * C + c] = (pt >> c) & 1 ? 1.0f : 0.0f; }
    for (auto &x : pm) x = rnd();
    for (auto &x : re) x = rnd();
    for (auto &x : ce) x = rnd();
    float *dpm, *dre, *dce, *dcats, *outs; int32_t *users, *outi;
    hipMalloc(&dpm, pm.size() * 4); hipMalloc(&dre, re.size() * 4); hipMalloc(&dce, ce.size() * 4);
    hipMalloc(&dcats, cats.size() * 4); hipMalloc(&outs, U * k * 4); hipMalloc(&outi, U * k * 4);
    hipMalloc(&users, U * 4); hipMalloc(&h.err_dev, 32); hipMemset(h.err_dev, 0, 32);
    h.nonfinite_dev = h.err_dev + 4; h.finite_scan_pending = false;
    hipMemcpy(dpm, pm.data(), pm.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dre, re.data(), re.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dce, ce.data(), ce.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcats, cats.data(), cats.size() * 4, hipMemcpyHostToDevice);
    std::vector<int32_t> hu(U);
    for (int64_t i = 0; i < U; ++i) hu[i] = (int32_t)i;
    hipMemcpy(users, hu.data(), U * 4, hipMemcpyHostToDevice);
    h.pm = dpm; h.re = dre; h.ce = dce; h.dish_cats = dcats;
#if M2D_DIAG & 16
    unsigned long long *dbg; hipMalloc(&dbg, (size_t)1048576 * 8 * 8); hipMemset(dbg, 0, (size_t)1048576 * 8 * 8);
    g_m2d_diag_buffer = dbg;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    m2d_launch_topk_users(&h, users, U, k, outs, outi, nullptr);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        int rc = m2d_launch_topk_users(&h, users, U, k, outs, outi, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rc) printf("rc=%d %s\n", rc, h.last_error.c_str());
        if (ms < best) best = ms;
    }
    // M2D_DIAG_REPS calls queued back to back (no host wait between them), a few times over: the rate, and with M2D_DIAG & 16 the
    // in-kernel clock of the LAST launch, that the chip HOLDS under this load (a handful of launches read high)
    if (getenv("M2D_DIAG_REPS")) {
        const int reps = atoi(getenv("M2D_DIAG_REPS"));
        for (int round = 0; round < 4; ++round) {
            hipEventRecord(e0);
            for (int it = 0; it < reps; ++it) m2d_launch_topk_users(&h, users, U, k, outs, outi, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("round %d: %d calls back to back, %.4f ms per call\n", round, reps, ms / reps);
        }
    }
#if M2D_DIAG & 16
    {
        std::vector<unsigned long long> hd((size_t)1048576 * 8);
        hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost);
        if (getenv("M2D_DIAG_DUMP")) {                      // raw records of the LAST launch, for scripts/diag/scan_breakdown.py
            FILE *f = fopen(getenv("M2D_DIAG_DUMP"), "wb");
            size_t nrec = hd.size() / 8;
            while (nrec > 0 && hd[(nrec - 1) * 8 + 6] == 0) --nrec;
            if (f) { fwrite(hd.data(), 64, nrec, f); fclose(f); printf("dumped %zu wave records, %d users per block\n", nrec, h.topk_block_users); }
        }
        double m = 0, e = 0, b = 0, sl = 0, ns = 0, st = 0, ck = 0, rt = 0, nw = 0, maxck = 0;
        for (int w = 0; w < 1048576; ++w) { m += hd[w*8]; e += hd[w*8+1]; b += hd[w*8+2]; sl += hd[w*8+3]; ns += hd[w*8+4]; st += hd[w*8+5]; ck += hd[w*8+6]; rt += hd[w*8+7]; nw += hd[w*8+6] != 0; if ((double)hd[w*8+6] > maxck) maxck = (double)hd[w*8+6]; }
#if M2D_DIAG & 32
        printf("first stage of an item (steps 1 .. 10): %.0f candidate steps per wave-item, %.0f cycles each; whole item: %.0f candidate steps, %.0f cycles each\n", e / nw, e ? rt / e : 0.0, ns / nw, ns ? sl / ns : 0.0);
        rt = 0;
#endif
        if (rt > 0) printf("in-kernel clock: %.0f waves, %.0f s_memtime ticks per wave (longest %.0f) over %.1f us (s_memrealtime, 100 MHz) = %.3f GHz; all waves together %.3e ticks\n", nw, ck / nw, maxck, rt / nw / 100.0, ck / rt * 0.1, ck);
        // pipelined bf16 kernel: d[0] = interleaved body, d[1] = sorted_insert calls, d[2] = stage wait + barrier, d[3] = slow path
        printf("per step per wave (cycles): body %.0f  slow path %.0f (%.1f%% of steps, %.0f each, %.2f inserts each)  wait+barrier %.0f  [%.0f steps/wave]\n",
               m / st, sl / st, 100.0 * ns / st, ns ? sl / ns : 0.0, ns ? e / ns : 0.0, b / st, st / (nw > 0 ? nw : 1));
    }
#endif
#if M2D_DIAG & 4096
    {   // time line of workgroup (0, 0): waves w and w + 4 share a SIMD (8-wave blocks)
        std::vector<unsigned long long> tr(8 * 512 * 4);
        hipMemcpy(tr.data(), dbg + 7000000, tr.size() * 8, hipMemcpyDeviceToHost);
        const unsigned long long t00 = tr[(0 * 512 + 1) * 4];
        const int q0 = getenv("M2D_TRACE_FROM") ? atoi(getenv("M2D_TRACE_FROM")) : 41;
        for (int w : {0, 4}) {
            printf("wave %d: step: start (+gap since the previous step's end) | pre-body  body  post\n", w);
            for (int q = q0; q < q0 + 40; ++q) {
                const unsigned long long *e = &tr[((size_t)w * 512 + q) * 4], *pe = &tr[((size_t)w * 512 + q - 1) * 4];
                printf("  q %3d: %8lld (+%4lld) | %5lld %5lld %5lld\n", q, (long long)(e[0] - t00), (long long)(e[0] - pe[3]), (long long)(e[1] - e[0]), (long long)(e[2] - e[1]), (long long)(e[3] - e[2]));
            }
        }
    }
#endif
    const double flops = 2.0 * K * (double)U * (double)I;
    printf("M2D_DIAG=%d  %s  %.3f ms  %.1f dense-equivalent TFLOP/s; executed bf16 flops %.0f TFLOP/s = %.3f of 2500\n", M2D_DIAG, h.last_kernel, best, flops / best / 1e9,
           6.0 * E * (double)U * (double)I / best / 1e9, 6.0 * E * (double)U * (double)I / best / 1e9 / 2500.0);
    return 0;
}
