/* CPU oracle (C) for the Market2Dish Recommender scoring path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of Model.inference, /root/reference/Code/Recommender/Model_Recommender.py:56-97,
 * one pair at a time, float32 throughout, sums taken in row-major (c, e) order.  It is the checker
 * for tests/ and the `cpu_baseline` leg of bench.py; nothing under foodrec_amd/ links or loads it.
 *
 * PARITY UNPINNED for the arithmetic: the reference runs these ops inside TensorFlow 1.x, which is
 * not available here, and ships no golden vectors.  This file is pinned only against the hand KAT
 *  and against oracle/m2d_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC; no -ffast-math: 0/0 must stay NaN).
 */
#include <stddef.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* return codes */
#define M2D_ORACLE_OK 0
#define M2D_ORACLE_BAD_USER (-1)
#define M2D_ORACLE_BAD_ITEM (-2)

int m2d_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* One pair.  Line numbers are Model_Recommender.py. */
static float score_one(const float *um /* [(C+1)*E] PM[user] :57 */,
                       const float *it /* [E] RE[item] :63 */,
                       const float *ce /* [C*E] */, const float *m /* [C] categories[:, :, 0] */,
                       int C, int E, float a, float b)
{
    float sum_cat = 0.0f;  /* :75 */
    float sum_dish = 0.0f; /* :90 */
    float n = 0.0f;        /* :77 */
    for (int c = 0; c < C; ++c) {
        const float mc = m[c];
        const float *cec = ce + (size_t)c * E;
        const float *ul = um + (size_t)(c + 1) * E; /* U_low[c] :59 */
        n += mc;
        for (int e = 0; e < E; ++e) {
            const float dish_category = mc * cec[e];       /* :67 */
            sum_cat += um[e] * dish_category;              /* :71 (U_high = row 0), :75 */
            const float dish_memory = mc * ul[e];          /* :82 */
            sum_dish += it[e] * dish_memory;               /* :86, :90 */
        }
    }
    const float high = sum_cat / n;  /* :79  (0/0 -> NaN) */
    const float low = sum_dish / n;  /* :92 */
    return a * high + b * low;       /* :95-96 */
}

/* Score B pairs.  `coef` is rounded to float32 and `1 - coef` is taken in float32 (:17, :96).
 * nthreads <= 0 -> all OpenMP threads.  Out-of-range ids are an error (TF-CPU GatherV2 raises). */
int m2d_oracle_score_pairs(const float *pm, const float *re, const float *ce, int64_t U, int64_t I,
                           int32_t C, int32_t E, float coef, const int32_t *users,
                           const int32_t *items, const float *cats /* [B, C] */, int64_t B,
                           float *out, int nthreads)
{
    const float a = coef;
    const float b = 1.0f - a;
    int err = M2D_ORACLE_OK;
    for (int64_t i = 0; i < B; ++i) {
        if (users[i] < 0 || users[i] >= U) return M2D_ORACLE_BAD_USER;
        if (items[i] < 0 || items[i] >= I) return M2D_ORACLE_BAD_ITEM;
    }
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int64_t i = 0; i < B; ++i) {
        out[i] = score_one(pm + (size_t)users[i] * (size_t)(C + 1) * E, re + (size_t)items[i] * E, ce,
                           cats + (size_t)i * C, C, E, a, b);
    }
    (void)nthreads;
    return err;
}

/* The graph as TensorFlow executes it (Model_Recommender.py:57-96): every op materialises its
 * [B, C, E] result before the next one reads it.  `scratch` must hold 4*B*C*E floats.  Used by
 * bench.py's cpu_baseline so the baseline pays the same temporaries the TF graph pays. */
int m2d_oracle_score_pairs_materialised(const float *pm, const float *re, const float *ce, int64_t U,
                                        int64_t I, int32_t C, int32_t E, float coef,
                                        const int32_t *users, const int32_t *items,
                                        const float *cats, int64_t B, float *out, float *scratch,
                                        int nthreads)
{
    const float a = coef;
    const float b = 1.0f - a;
    const size_t CE_ = (size_t)C * E;
    float *t_dc = scratch;                 /* Dish_Category  :67 */
    float *t_cs = scratch + (size_t)B * CE_;     /* category_score :71 */
    float *t_dm = scratch + 2 * (size_t)B * CE_; /* Dish_Memory    :82 */
    float *t_ds = scratch + 3 * (size_t)B * CE_; /* dish_score     :86 */
    for (int64_t i = 0; i < B; ++i) {
        if (users[i] < 0 || users[i] >= U) return M2D_ORACLE_BAD_USER;
        if (items[i] < 0 || items[i] >= I) return M2D_ORACLE_BAD_ITEM;
    }
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#endif
#define PAR _Pragma("omp parallel for num_threads(nthreads) schedule(static)")
    PAR for (int64_t i = 0; i < B; ++i)
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < E; ++e)
                t_dc[i * CE_ + (size_t)c * E + e] = cats[i * C + c] * ce[(size_t)c * E + e];
    PAR for (int64_t i = 0; i < B; ++i) {
        const float *uh = pm + (size_t)users[i] * (size_t)(C + 1) * E;
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < E; ++e)
                t_cs[i * CE_ + (size_t)c * E + e] = uh[e] * t_dc[i * CE_ + (size_t)c * E + e];
    }
    PAR for (int64_t i = 0; i < B; ++i) {
        const float *ul = pm + (size_t)users[i] * (size_t)(C + 1) * E + E;
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < E; ++e)
                t_dm[i * CE_ + (size_t)c * E + e] = cats[i * C + c] * ul[(size_t)c * E + e];
    }
    PAR for (int64_t i = 0; i < B; ++i) {
        const float *it = re + (size_t)items[i] * E;
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < E; ++e)
                t_ds[i * CE_ + (size_t)c * E + e] = it[e] * t_dm[i * CE_ + (size_t)c * E + e];
    }
    PAR for (int64_t i = 0; i < B; ++i) {
        float s1 = 0.0f, s2 = 0.0f, n = 0.0f;
        for (size_t k = 0; k < CE_; ++k) s1 += t_cs[i * CE_ + k];
        for (size_t k = 0; k < CE_; ++k) s2 += t_ds[i * CE_ + k];
        for (int c = 0; c < C; ++c) n += cats[i * C + c];
        out[i] = a * (s1 / n) + b * (s2 / n);
    }
#undef PAR
    return M2D_ORACLE_OK;
}
