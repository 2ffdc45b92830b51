/* Plain-C host program against include/m2d.h -- no Python, no torch: what a non-Python caller of the
 * C ABI looks like.  Built and run by tests/test_gpu_c_abi.py on the GPU box:
 *   hipcc -x c ... is not needed: this file only needs a C compiler, the HIP runtime API for memory
 *   (hip_runtime_api.h is plain C) and libm2d.so.
 * Scores one hand-checkable pair (3.4625) and a small random batch against a scalar
 * restatement of Model_Recommender.py:56-97 written inline. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "m2d.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { printf("FAIL %s -> %d (%s)\n", #x, rc_, m2d_last_error(h)); return 1; } } while (0)
#define HCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static float ref_score(const float *pm, const float *re, const float *ce, int C, int E, int u, int d, const float *m)
{
    float sc = 0.f, sd = 0.f, n = 0.f;
    for (int c = 0; c < C; ++c) {
        n += m[c];
        for (int e = 0; e < E; ++e) {
            sc += pm[(size_t)u * (C + 1) * E + e] * (m[c] * ce[c * E + e]);
            sd += re[(size_t)d * E + e] * (m[c] * pm[((size_t)u * (C + 1) + c + 1) * E + e]);
        }
    }
    const float a = 0.99f, b = 1.0f - a;
    return a * (sc / n) + b * (sd / n);
}

int main(void)
{
    m2d_engine *h = NULL;
    /* --- the hand KAT --- */
    {
        const float pm[10] = {1, 2, 1, 0, 0, 1, 2, 2, 3, -1}, re[2] = {0.5f, -1}, ce[8] = {1, 1, 2, 0, 0, 2, -1, 1};
        const float m[4] = {1, 0, 1, 0};
        const int32_t u0 = 0, d0 = 0;
        int32_t *du, *dd; float *dm, *dout, out = 0;
        CHECK(m2d_create(pm, re, ce, 1, 1, 4, 2, 0.99f, 0, M2D_TABLES_HOST, &h));
        HCHECK(hipMalloc((void **)&du, 4)); HCHECK(hipMalloc((void **)&dd, 4));
        HCHECK(hipMalloc((void **)&dm, 16)); HCHECK(hipMalloc((void **)&dout, 4));
        HCHECK(hipMemcpy(du, &u0, 4, hipMemcpyHostToDevice)); HCHECK(hipMemcpy(dd, &d0, 4, hipMemcpyHostToDevice));
        HCHECK(hipMemcpy(dm, m, 16, hipMemcpyHostToDevice));
        CHECK(m2d_score_pairs(h, du, dd, dm, 1, dout, NULL));
        CHECK(m2d_check(h, NULL, NULL, NULL));
        HCHECK(hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost));
        printf("KAT score = %.6f\n", out);
        if (fabsf(out - 3.4625f) > 1e-5f) { printf("FAIL KAT\n"); return 1; }
        CHECK(m2d_destroy(h));
    }
    /* --- random batch, E = 64 --- */
    {
        const int U = 300, I = 200, C = 4, E = 64, B = 5000;
        float *pm = malloc(sizeof(float) * U * (C + 1) * E), *re = malloc(sizeof(float) * I * E), *ce = malloc(sizeof(float) * C * E);
        float *m = malloc(sizeof(float) * B * C), *out = malloc(sizeof(float) * B);
        int32_t *us = malloc(4 * B), *ds = malloc(4 * B);
        unsigned s = 7;
#define RND() (s = s * 1664525u + 1013904223u, ((s >> 8) & 0xffff) / 65536.0f - 0.5f)
        for (int i = 0; i < U * (C + 1) * E; ++i) pm[i] = RND() * 0.25f;
        for (int i = 0; i < I * E; ++i) re[i] = RND() * 0.25f;
        for (int i = 0; i < C * E; ++i) ce[i] = RND() * 0.25f;
        for (int i = 0; i < B; ++i) {
            s = s * 1664525u + 1013904223u; us[i] = (s >> 8) % U;
            s = s * 1664525u + 1013904223u; ds[i] = (s >> 8) % I;
            for (int c = 0; c < C; ++c) { s = s * 1664525u + 1013904223u; m[i * C + c] = (s >> 16) & 1; }
            if (m[i * C] + m[i * C + 1] + m[i * C + 2] + m[i * C + 3] == 0) m[i * C + 1] = 1;
        }
        int32_t *du, *dd; float *dm, *dout;
        CHECK(m2d_create(pm, re, ce, U, I, C, E, 0.99f, 0, M2D_TABLES_HOST, &h));
        HCHECK(hipMalloc((void **)&du, 4 * B)); HCHECK(hipMalloc((void **)&dd, 4 * B));
        HCHECK(hipMalloc((void **)&dm, 4 * B * C)); HCHECK(hipMalloc((void **)&dout, 4 * B));
        HCHECK(hipMemcpy(du, us, 4 * B, hipMemcpyHostToDevice)); HCHECK(hipMemcpy(dd, ds, 4 * B, hipMemcpyHostToDevice));
        HCHECK(hipMemcpy(dm, m, 4 * B * C, hipMemcpyHostToDevice));
        hipStream_t st; HCHECK(hipStreamCreate(&st));
        CHECK(m2d_score_pairs(h, du, dd, dm, B, dout, st));          /* a non-default stream */
        CHECK(m2d_check(h, st, NULL, NULL));
        HCHECK(hipMemcpy(out, dout, 4 * B, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int i = 0; i < B; ++i) {
            const double e = fabs(out[i] - ref_score(pm, re, ce, C, E, us[i], ds[i], m + i * C));
            if (e > worst) worst = e;
        }
        printf("batch of %d: max |err| = %.3g\n", B, worst);
        if (worst > 1e-4) { printf("FAIL batch\n"); return 1; }
        /* an out-of-range id is reported, not clamped */
        const int32_t bad = U; int64_t bv = -1, bi = -1;
        HCHECK(hipMemcpy(du + 17, &bad, 4, hipMemcpyHostToDevice));
        CHECK(m2d_score_pairs(h, du, dd, dm, B, dout, st));
        const int rc = m2d_check(h, st, &bv, &bi);
        printf("bad id -> rc %d value %lld index %lld (%s)\n", rc, (long long)bv, (long long)bi, m2d_last_error(h));
        if (rc != M2D_ERR_BAD_USER_ID || bv != U || bi != 17) { printf("FAIL error path\n"); return 1; }
        /* --- a training step from plain C (SGD, lr 0.5): loss = mean sigmoid-CE of the scores checked above
         * (Model_Recommender.py:99-104), and one step along the clipped negative gradient lowers it --- */
        HCHECK(hipMemcpy(du + 17, &us[17], 4, hipMemcpyHostToDevice));
        float *y = malloc(4 * B), *dy, *dres, res[4];
        double loss = 0;
        for (int i = 0; i < B; ++i) {
            y[i] = (float)(i & 1);
            const double sc = ref_score(pm, re, ce, C, E, us[i], ds[i], m + i * C);
            loss += (sc > 0 ? sc : 0) - sc * y[i] + log1p(exp(-fabs(sc)));
        }
        loss /= B;
        HCHECK(hipMalloc((void **)&dy, 4 * B)); HCHECK(hipMalloc((void **)&dres, 16));
        HCHECK(hipMemcpy(dy, y, 4 * B, hipMemcpyHostToDevice));
        CHECK(m2d_train_begin(h, M2D_LEARNER_SGD, 0.5f, 5.0f, st));
        CHECK(m2d_train_step(h, du, dd, dm, dy, B, 1, dres, st));
        CHECK(m2d_check(h, st, NULL, NULL));
        HCHECK(hipMemcpy(res, dres, 16, hipMemcpyDeviceToHost));
        printf("train step: loss %.6f (host %.6f), gradient norm %.4g, clip scale %.3g, lr %.3g\n", res[0], loss, res[1], res[2], res[3]);
        if (fabs(res[0] - loss) > 1e-5 || res[2] != 1.0f || res[3] != 0.5f) { printf("FAIL train loss\n"); return 1; }
        /* the engine owns its copy of host tables; m2d_train_slot has nothing to copy for SGD */
        if (m2d_train_slot(h, 2, 0, dres, 0, st) != M2D_ERR_INVALID_ARG) { printf("FAIL sgd slot\n"); return 1; }
        /* apply = 0: the loss_value fetch alone, on the updated tables */
        CHECK(m2d_train_step(h, du, dd, dm, dy, B, 0, dres, st));
        CHECK(m2d_check(h, st, NULL, NULL));
        float res2[4];
        HCHECK(hipMemcpy(res2, dres, 16, hipMemcpyDeviceToHost));
        printf("loss after the step: %.6f\n", res2[0]);
        if (!(res2[0] < res[0])) { printf("FAIL loss did not decrease\n"); return 1; }
        CHECK(m2d_train_end(h));
        CHECK(m2d_destroy(h));
    }
    printf("C ABI OK (version %d)\n", m2d_abi_version());
    return 0;
}
