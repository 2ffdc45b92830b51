// Fused pair-score kernels for gfx950 (MI355X): one pass over
//   gather PM[user] ((C+1)*E floats, contiguous)      Model_Recommender.py:57-62
//   gather RE[item] (E floats)                        Model_Recommender.py:63-66
//   masked category dot (high level)                  Model_Recommender.py:67-80
//   masked low-level dot                              Model_Recommender.py:82-93
//   blend                                             Model_Recommender.py:95-97
// with none of the [B, C, E] temporaries the TF graph materialises.
//
// Mapping (wave64).  A wave owns 64 consecutive pairs ("chunk"): lane i reads pair i's two ids and
// its mask with coalesced loads and finally stores pair i's score, so every id/mask/score access is
// one full 256-B (or 1-KiB) wave transaction.  Inside the chunk the wave is cut into 64/LPP groups
// of LPP lanes; a group walks its LPP pairs one per step, lane j of the group holding float4 column
// j of every row, so each row read is LPP*16 contiguous bytes (256 B at E = 64).  The Category_Embedding
// fragment a lane needs (C float4) never changes and lives in registers.  PF steps of row loads
// are kept in flight ahead of the arithmetic; the rows of Personal_Memory are read once and can
// be marked non-temporal so they do not evict the (re-used) dish rows from L2 / Infinity Cache.
//
// HBM-bound: ~1 flop per byte.  Algorithmic bytes per pair = (C+2)*E*4 + C*4 + 12.
#include "m2d_engine.h"

namespace {

struct ScoreArgs {
    const float *pm;
    const float *re;
    const float *ce;
    const int32_t *users;
    const int32_t *items;
    const float *cats;  // [B, C] (explicit feed) or [I, C] (by dish)
    const float *hv;    // extension: per-dish high-level vectors [I, E] (ingredient table), or null
    float *out;
    int64_t B;
    int64_t U;
    int64_t I;
    int64_t user_base;
    int32_t E;
    int32_t C;
    float a;
    float b;
    int32_t *err;
    int32_t skip_masked;   // do not fetch the Personal_Memory rows of categories whose mask weight is 0 (their products are 0)
    const int32_t *nonfinite;   // device word: some table value is inf / NaN -> 0 * row is not 0 (:82), every row is fetched
    const float *uh;       // [U, 4] derived table <U_high[u], CE_c> (m2d_build_user_high), or null: read U_high and multiply
};

__device__ __forceinline__ void latch_error(int32_t *err, int code, int64_t value, int64_t index)
{
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int32_t)value;
        err[2] = (int32_t)(index & 0xffffffff);
        err[3] = (int32_t)(index >> 32);
    }
}

// Rows of weight-0 categories may be left out only while every table value is finite: 0 * inf = NaN at
// Model_Recommender.py:82-90.  One scalar load per wave (the word is set by the table scan / the engine's writers).
__device__ __forceinline__ bool skip_rows(const ScoreArgs &p)
{
    return p.skip_masked != 0 && __builtin_amdgcn_readfirstlane(*p.nonfinite) == 0;
}

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4f ld4(const v4f *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

__device__ __forceinline__ float dot4(const v4f &x, const v4f &y, float acc)
{
    acc = fmaf(x.x, y.x, acc);
    acc = fmaf(x.y, y.y, acc);
    acc = fmaf(x.z, y.z, acc);
    acc = fmaf(x.w, y.w, acc);
    return acc;
}

__device__ __forceinline__ v4f scale4(float s, const v4f &x) { return s * x; }

// Sum over the LPP lanes of a group; every lane of the group ends with the total.
template <int LPP>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int off = LPP / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// C == 4 categories, E % 4 == 0, E / 4 <= LPP.  FULL: E / 4 == LPP (no idle lanes).
// HV (build-defined extension): the high-level operand sum_c m_c CE_c / n of
// Model_Recommender.py:67-79 is replaced by a resident per-dish vector H[d] (the normalised multi-hot
// ingredient sum), i.e. high = <U_high, H[d]>; the low-level path is unchanged.
template <int LPP, int PF, bool BYDISH, bool NT, bool FULL, bool HV, bool UH = false>
__global__ __launch_bounds__(256) void m2d_score_pairs_c4(ScoreArgs p)
{
    constexpr int C = 4;
    const int lane = threadIdx.x & 63;
    const int j = lane & (LPP - 1);
    const int E4 = FULL ? LPP : (p.E >> 2);
    const bool col_ok = FULL || (j < E4);
    const int jc = col_ok ? j : 0;  // idle lanes re-read column 0 (in bounds), contribution zeroed
    const int64_t nchunks = (p.B + 63) >> 6;
    // wave-uniform loop control lives in SGPRs: no shuffle below ever runs under a partial EXEC mask
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    const bool skipm = skip_rows(p);
    const v4f *pm4 = reinterpret_cast<const v4f *>(p.pm);
    const v4f *re4 = reinterpret_cast<const v4f *>(p.re);
    const v4f *ce4 = reinterpret_cast<const v4f *>(p.ce);
    const v4f *hv4 = reinterpret_cast<const v4f *>(p.hv);
    const size_t urow4 = (size_t)(C + 1) * E4;  // float4 per user block

    v4f cef[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        cef[c] = ce4[(size_t)c * E4 + jc];
        if (!col_ok) cef[c] = v4f{0.f, 0.f, 0.f, 0.f};
    }

    for (int64_t chunk = wave0; chunk < nchunks; chunk += nwaves) {
        const int64_t pi = (chunk << 6) + lane;
        const bool valid = pi < p.B;
        int32_t uid = valid ? p.users[pi] : (int32_t)p.user_base;
        int32_t did = valid ? p.items[pi] : 0;
        int64_t ul = (int64_t)uid - p.user_base;
        bool bad = false;
        if (ul < 0 || ul >= p.U) {
            latch_error(p.err, M2D_ERR_BAD_USER_ID, uid, pi);
            ul = 0;
            bad = true;
        }
        if (did < 0 || (int64_t)did >= p.I) {
            latch_error(p.err, M2D_ERR_BAD_ITEM_ID, did, pi);
            did = 0;
            bad = true;
        }
        v4f m = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const v4f *cp = reinterpret_cast<const v4f *>(p.cats);
            m = BYDISH ? cp[did] : cp[pi];
        }
        const int32_t ul32 = (int32_t)ul;
        // categories whose weight is exactly 0 contribute 0 * U_low[c] = 0 to :82-:90: their rows are not fetched
        // (a NaN weight compares unequal to 0 and keeps its row)
        const int32_t act = skipm ? ((m.x != 0.f ? 1 : 0) | (m.y != 0.f ? 2 : 0) | (m.z != 0.f ? 4 : 0) | (m.w != 0.f ? 8 : 0)) : 15;
        // high-level part from the derived table: sum_c m_c <U_high, CE_c> (the U_high row is then not fetched at all)
        // (UH is its own instantiation: the literal kernel's code, and with it its bits, stay what they were)
        constexpr bool use_uh = UH && !HV;
        v4f uhv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (use_uh) uhv = reinterpret_cast<const v4f *>(p.uh)[ul];

        v4f ub[PF][C + 1];
        v4f ib[PF];
        v4f hb[PF];
        float my_high = 0.f, my_low = 0.f;

        auto issue = [&](int s, int slot) {
            const int32_t us = __shfl(ul32, s, LPP);
            const int32_t ds = __shfl(did, s, LPP);
            const int32_t as = __shfl(act, s, LPP);
            const v4f *pu = pm4 + (size_t)us * urow4 + jc;
            if constexpr (use_uh) ub[slot][0] = v4f{0.f, 0.f, 0.f, 0.f};
            else ub[slot][0] = ld4<NT>(pu);
#pragma unroll
            for (int r = 1; r <= C; ++r) {
                ub[slot][r] = v4f{0.f, 0.f, 0.f, 0.f};
                if ((as >> (r - 1)) & 1) ub[slot][r] = ld4<NT>(pu + (size_t)r * E4);
            }
            ib[slot] = re4[(size_t)ds * E4 + jc];
            if constexpr (HV) hb[slot] = hv4[(size_t)ds * E4 + jc];
        };

#pragma unroll
        for (int k = 0; k < PF; ++k) issue(k, k);

        for (int s0 = 0; s0 < LPP; s0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int s = s0 + k;
                const float m0 = __shfl(m.x, s, LPP);
                const float m1 = __shfl(m.y, s, LPP);
                const float m2 = __shfl(m.z, s, LPP);
                const float m3 = __shfl(m.w, s, LPP);
                const float mc[C] = {m0, m1, m2, m3};
                float hs = 0.f, ls = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if constexpr (!HV && !use_uh) {
                        const v4f dish_category = scale4(mc[c], cef[c]);    // :67
                        hs = dot4(ub[k][0], dish_category, hs);                // :71, :75
                    }
                    const v4f dish_memory = scale4(mc[c], ub[k][c + 1]);    // :82
                    ls = dot4(ib[k], dish_memory, ls);                         // :86, :90
                }
                if constexpr (HV) hs = dot4(ub[k][0], hb[k], hs);
                if (!FULL && !col_ok) {
                    ls = 0.f;
                    if (HV) hs = 0.f;
                }
                if (s + PF < LPP) issue(s + PF, k);
                if constexpr (HV || !use_uh) hs = group_sum<LPP>(hs);
                ls = group_sum<LPP>(ls);
                if (j == s) {
                    my_high = hs;
                    my_low = ls;
                }
            }
        }
        if (valid) {
            const float n = (m.x + m.y) + (m.z + m.w);                         // :77
            if constexpr (use_uh) my_high = fmaf(m.w, uhv.w, fmaf(m.z, uhv.z, fmaf(m.y, uhv.y, m.x * uhv.x)));
            const float high = HV ? my_high : my_high / n;                     // :79 (H[d] is already normalised)
            const float low = my_low / n;                                      // :92
            float score = m2d_blend_unfused(p.a, high, p.b, low);     // :95-96, no fma contraction
            if (bad) score = __builtin_nanf("");
            p.out[pi] = score;
        }
    }
}

// Derived table for serving: uh[u][c] = <U_high[u], CE_c>, c < 4.  The high-level sum of Model_Recommender.py:67-79 is
// sum_c m_c uh[u][c] / n -- the same products in another order -- so a pair reads 16 bytes of this table (16 B x U: it
// lives in L2 / the Infinity Cache) instead of the E x 4-byte U_high row from HBM.  One group of E/4 lanes per user.
template <int LPP>
__global__ __launch_bounds__(256) void m2d_build_user_high(const float *pm, const float *ce, int64_t U, int32_t E, float *out)
{
    constexpr int C = 4;
    const int lane = threadIdx.x & 63, j = lane & (LPP - 1), E4 = E >> 2;
    const bool col_ok = j < E4;
    const int jc = col_ok ? j : 0;
    const v4f *ce4 = reinterpret_cast<const v4f *>(ce);
    v4f cef[C];
#pragma unroll
    for (int c = 0; c < C; ++c) cef[c] = col_ok ? ce4[(size_t)c * E4 + jc] : v4f{0.f, 0.f, 0.f, 0.f};
    const int64_t gpw = 64 / LPP;
    const int64_t g0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * gpw + lane / LPP;
    for (int64_t u = g0; u < ((U + gpw - 1) / gpw) * gpw; u += (int64_t)gridDim.x * 4 * gpw) {   // whole groups: full-wave shuffles
        const bool ok = u < U;
        const v4f x = reinterpret_cast<const v4f *>(pm)[(size_t)(ok ? u : 0) * (C + 1) * E4 + jc];
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = group_sum<LPP>(dot4(x, cef[c], 0.f));
        if (ok && j == 0) reinterpret_cast<v4f *>(out)[u] = v4f{h[0], h[1], h[2], h[3]};
    }
}

// Latency form of the kernel above for small batches (serving calls, the reference's own 51 pairs per sess.run):
// a group of LPP lanes takes ONE pair per pass instead of walking LPP pairs one after another, so a batch of B
// pairs is spread over B * LPP / 64 waves and finishes in about one row-gather latency instead of up to 64 of
// them in sequence.  Same per-lane arithmetic and the same group reduction: bit-identical scores.
template <int LPP, bool BYDISH, bool FULL, bool HV>
__global__ __launch_bounds__(256) void m2d_score_pairs_c4_small(ScoreArgs p)
{
    constexpr int C = 4, GPW = 64 / LPP;                    // pairs per wave per pass
    const int lane = threadIdx.x & 63;
    const int j = lane & (LPP - 1);
    const int E4 = FULL ? LPP : (p.E >> 2);
    const bool col_ok = FULL || (j < E4);
    const int jc = col_ok ? j : 0;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const v4f *pm4 = reinterpret_cast<const v4f *>(p.pm);
    const v4f *re4 = reinterpret_cast<const v4f *>(p.re);
    const v4f *ce4 = reinterpret_cast<const v4f *>(p.ce);
    const v4f *hv4 = reinterpret_cast<const v4f *>(p.hv);
    const size_t urow4 = (size_t)(C + 1) * E4;
    const bool skipm = skip_rows(p);
    v4f cef[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        cef[c] = ce4[(size_t)c * E4 + jc];
        if (!col_ok) cef[c] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t base = wave0 * GPW; base < p.B; base += nwaves * GPW) {
        const int64_t pi = base + lane / LPP;
        const bool valid = pi < p.B;
        int32_t uid = valid ? p.users[pi] : (int32_t)p.user_base;
        int32_t did = valid ? p.items[pi] : 0;
        int64_t ul = (int64_t)uid - p.user_base;
        bool bad = false;
        if (ul < 0 || ul >= p.U) {
            if (j == 0) latch_error(p.err, M2D_ERR_BAD_USER_ID, uid, pi);
            ul = 0;
            bad = true;
        }
        if (did < 0 || (int64_t)did >= p.I) {
            if (j == 0) latch_error(p.err, M2D_ERR_BAD_ITEM_ID, did, pi);
            did = 0;
            bad = true;
        }
        v4f m = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const v4f *cp = reinterpret_cast<const v4f *>(p.cats);
            m = BYDISH ? cp[did] : cp[pi];
        }
        const v4f *pu = pm4 + (size_t)ul * urow4 + jc;
        v4f ub[C + 1];
        const float mw[C] = {m.x, m.y, m.z, m.w};
        ub[0] = pu[0];
#pragma unroll
        for (int r = 1; r <= C; ++r) {                     // a category of weight 0 contributes 0: its row is not fetched
            ub[r] = v4f{0.f, 0.f, 0.f, 0.f};
            if (!skipm || mw[r - 1] != 0.f) ub[r] = pu[(size_t)r * E4];
        }
        const v4f ib = re4[(size_t)did * E4 + jc];
        v4f hb = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HV) hb = hv4[(size_t)did * E4 + jc];
        const float mc[C] = {m.x, m.y, m.z, m.w};
        float hs = 0.f, ls = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if constexpr (!HV) {
                const v4f dish_category = scale4(mc[c], cef[c]);    // :67
                hs = dot4(ub[0], dish_category, hs);                   // :71, :75
            }
            const v4f dish_memory = scale4(mc[c], ub[c + 1]);       // :82
            ls = dot4(ib, dish_memory, ls);                            // :86, :90
        }
        if constexpr (HV) hs = dot4(ub[0], hb, hs);
        if (!FULL && !col_ok) {
            ls = 0.f;
            if (HV) hs = 0.f;
        }
        hs = group_sum<LPP>(hs);
        ls = group_sum<LPP>(ls);
        if (valid && j == 0) {
            const float n = (m.x + m.y) + (m.z + m.w);                         // :77
            const float high = HV ? hs : hs / n;                               // :79
            const float low = ls / n;                                          // :92
            float score = m2d_blend_unfused(p.a, high, p.b, low);     // :95-96
            if (bad) score = __builtin_nanf("");
            p.out[pi] = score;
        }
    }
}

// The throughput form for category counts other than 4 (1 <= C <= 8, E % 4 == 0, E / 4 <= LPP): the layout of
// m2d_score_pairs_c4 -- a group of LPP lanes walks its LPP pairs, a float4 of every row per lane, two pairs' rows in
// flight -- with the category loop unrolled to 8 and every c >= C step skipped by a wave-uniform branch (no 0 * row
// products are added for categories the model does not have).  Mask weights are read as scalars (a pair's C weights
// are not a float4).  One wave per pair, as m2d_score_pairs_generic does it, reaches 1.3-2.1 G pairs/s on config-2
// sized tables where this form reaches the rate of the C = 4 kernel.
template <int LPP, bool BYDISH, bool NT>
__global__ __launch_bounds__(256) void m2d_score_pairs_cn(ScoreArgs p)
{
    constexpr int CM = 8, PF = 2;
    const int C = p.C;
    const int lane = threadIdx.x & 63;
    const int j = lane & (LPP - 1);
    const int E4 = p.E >> 2;
    const bool col_ok = j < E4;
    const int jc = col_ok ? j : 0;  // idle lanes re-read column 0 (in bounds), contribution zeroed
    const int64_t nchunks = (p.B + 63) >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);

    const v4f *pm4 = reinterpret_cast<const v4f *>(p.pm);
    const v4f *re4 = reinterpret_cast<const v4f *>(p.re);
    const v4f *ce4 = reinterpret_cast<const v4f *>(p.ce);
    const size_t urow4 = (size_t)(C + 1) * E4;  // float4 per user block
    const bool skipm = skip_rows(p);

    v4f cef[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        cef[c] = v4f{0.f, 0.f, 0.f, 0.f};
        if (c < C && col_ok) cef[c] = ce4[(size_t)c * E4 + jc];
    }

    for (int64_t chunk = wave0; chunk < nchunks; chunk += nwaves) {
        const int64_t pi = (chunk << 6) + lane;
        const bool valid = pi < p.B;
        int32_t uid = valid ? p.users[pi] : (int32_t)p.user_base;
        int32_t did = valid ? p.items[pi] : 0;
        int64_t ul = (int64_t)uid - p.user_base;
        bool bad = false;
        if (ul < 0 || ul >= p.U) {
            latch_error(p.err, M2D_ERR_BAD_USER_ID, uid, pi);
            ul = 0;
            bad = true;
        }
        if (did < 0 || (int64_t)did >= p.I) {
            latch_error(p.err, M2D_ERR_BAD_ITEM_ID, did, pi);
            did = 0;
            bad = true;
        }
        float mk[CM];
        int32_t act = 0;
        float n = 0.f;
        {
            const float *mrow = p.cats + (BYDISH ? (size_t)did : (size_t)pi) * C;
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                mk[c] = 0.f;
                if (c < C) {
                    if (valid) mk[c] = mrow[c];
                    n += mk[c];                                                // :77
                    // weight exactly 0: 0 * U_low[c] = 0 at :82-:90, the row is not fetched (a NaN weight keeps its row)
                    act |= (!skipm || mk[c] != 0.f) ? (1 << c) : 0;
                }
            }
        }
        const int32_t ul32 = (int32_t)ul;

        v4f ub[PF][CM + 1];
        v4f ib[PF];
        float my_high = 0.f, my_low = 0.f;

        auto issue = [&](int s, int slot) {
            const int32_t us = __shfl(ul32, s, LPP);
            const int32_t ds = __shfl(did, s, LPP);
            const int32_t as = __shfl(act, s, LPP);
            const v4f *pu = pm4 + (size_t)us * urow4 + jc;
            ub[slot][0] = ld4<NT>(pu);
#pragma unroll
            for (int r = 1; r <= CM; ++r) {
                ub[slot][r] = v4f{0.f, 0.f, 0.f, 0.f};
                if ((as >> (r - 1)) & 1) ub[slot][r] = ld4<NT>(pu + (size_t)r * E4);     // bits >= C are never set
            }
            ib[slot] = re4[(size_t)ds * E4 + jc];
        };

#pragma unroll
        for (int k = 0; k < PF; ++k) issue(k, k);

        for (int s0 = 0; s0 < LPP; s0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int s = s0 + k;
                float hs = 0.f, ls = 0.f;
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C) {                                               // wave-uniform
                        const float mc = __shfl(mk[c], s, LPP);
                        const v4f dish_category = scale4(mc, cef[c]);       // :67
                        hs = dot4(ub[k][0], dish_category, hs);             // :71, :75
                        const v4f dish_memory = scale4(mc, ub[k][c + 1]);   // :82
                        ls = dot4(ib[k], dish_memory, ls);                  // :86, :90
                    }
                }
                if (!col_ok) {
                    hs = 0.f;
                    ls = 0.f;
                }
                if (s + PF < LPP) issue(s + PF, k);
                hs = group_sum<LPP>(hs);
                ls = group_sum<LPP>(ls);
                if (j == s) {
                    my_high = hs;
                    my_low = ls;
                }
            }
        }
        if (valid) {
            const float high = my_high / n;                                    // :79
            const float low = my_low / n;                                      // :92
            float score = m2d_blend_unfused(p.a, high, p.b, low);     // :95-96, no fma contraction
            if (bad) score = __builtin_nanf("");
            p.out[pi] = score;
        }
    }
}

// Any C (<= 64), any E: one wave per pair, lanes stride over e.  Slow path for shapes the
// vectorised kernels do not cover (E % 4 != 0, E > 256, C > 8) and the latency path of C != 4.
template <bool BYDISH>
__global__ __launch_bounds__(256) void m2d_score_pairs_generic(ScoreArgs p)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int C = p.C, E = p.E;
    const bool skipm = skip_rows(p);
    for (int64_t pi = wave0; pi < p.B; pi += nwaves) {
        int32_t uid = p.users[pi];
        int32_t did = p.items[pi];
        int64_t ul = (int64_t)uid - p.user_base;
        bool bad = false;
        if (ul < 0 || ul >= p.U) {
            if (lane == 0) latch_error(p.err, M2D_ERR_BAD_USER_ID, uid, pi);
            ul = 0;
            bad = true;
        }
        if (did < 0 || (int64_t)did >= p.I) {
            if (lane == 0) latch_error(p.err, M2D_ERR_BAD_ITEM_ID, did, pi);
            did = 0;
            bad = true;
        }
        const float *um = p.pm + (size_t)ul * (size_t)(C + 1) * E;
        const float *it = p.re + (size_t)did * E;
        const float *mrow = p.cats + (BYDISH ? (size_t)did * C : (size_t)pi * C);
        float hs = 0.f, ls = 0.f, n = 0.f;
        for (int c = 0; c < C; ++c) {
            const float mc = mrow[c];
            n += mc;
            if (skipm && mc == 0.f) continue;      // 0 * row = 0: the row is not fetched
            for (int e = lane; e < E; e += 64) {
                if (!p.hv) hs = fmaf(um[e], mc * p.ce[(size_t)c * E + e], hs);
                ls = fmaf(it[e], mc * um[(size_t)(c + 1) * E + e], ls);
            }
        }
        if (p.hv)
            for (int e = lane; e < E; e += 64) hs = fmaf(um[e], p.hv[(size_t)did * E + e], hs);
        hs = group_sum<64>(hs);
        ls = group_sum<64>(ls);
        if (lane == 0) {
            float score = m2d_blend_unfused(p.a, p.hv ? hs : hs / n, p.b, ls / n);
            if (bad) score = __builtin_nanf("");
            p.out[pi] = score;
        }
    }
}

template <int LPP, bool FULL, bool BYDISH>
void launch_c4(const ScoreArgs &a, int pf, bool nt, bool small, dim3 grid, hipStream_t st, const char **name)
{
#define M2D_CASE(PFV, NTV)                                                                           \
    if (pf == PFV && nt == NTV) {                                                                    \
        hipLaunchKernelGGL((m2d_score_pairs_c4<LPP, PFV, BYDISH, NTV, FULL, false>), grid, dim3(256), 0, st, a); \
        return;                                                                                      \
    }
    if (small) {  // latency form: one pair per group per pass
        *name = a.hv ? "m2d_score_pairs_c4_small_hv" : "m2d_score_pairs_c4_small";
        if (a.hv) hipLaunchKernelGGL((m2d_score_pairs_c4_small<LPP, BYDISH, FULL, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((m2d_score_pairs_c4_small<LPP, BYDISH, FULL, false>), grid, dim3(256), 0, st, a);
        return;
    }
    if (a.hv) {   // extension kernel: one configuration (PF 2, non-temporal user rows)
        *name = "m2d_score_pairs_c4_hv";
        hipLaunchKernelGGL((m2d_score_pairs_c4<LPP, 2, BYDISH, true, FULL, true>), grid, dim3(256), 0, st, a);
        return;
    }
    *name = "m2d_score_pairs_c4";
    if (a.uh) {   // high-level sum from the derived table: one configuration (PF 2, non-temporal user rows), as the default
        *name = "m2d_score_pairs_c4_uh";
        if (pf == 4) hipLaunchKernelGGL((m2d_score_pairs_c4<LPP, 4, BYDISH, true, FULL, false, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((m2d_score_pairs_c4<LPP, 2, BYDISH, true, FULL, false, true>), grid, dim3(256), 0, st, a);
        return;
    }
    M2D_CASE(1, false) M2D_CASE(1, true) M2D_CASE(2, false) M2D_CASE(2, true)
    M2D_CASE(4, false) M2D_CASE(4, true)
#undef M2D_CASE
}

template <bool BYDISH>
int launch_any(m2d_engine *h, const ScoreArgs &a, hipStream_t st)
{
    const int64_t nchunks = (a.B + 63) >> 6;
    int pf = h->opt_prefetch;
    if (pf != 1 && pf != 2 && pf != 4) pf = 2;
    const bool nt = h->opt_nt != 0;
    const int E4 = a.E >> 2;
    const bool vec_ok = (a.C == 4) && (a.E % 4 == 0) && E4 <= 64 && h->opt_variant != 9;
    int64_t blocks;
    const int64_t cap = (int64_t)h->num_cu * (h->opt_blocks_per_cu > 0 ? h->opt_blocks_per_cu : 8);
    // up to 8192 pairs: the latency form (option "variant" = 11 forces the throughput form, 12 the latency form)
    const bool small = vec_ok && ((a.B <= 8192 && h->opt_variant != 11) || h->opt_variant == 12);
    if (vec_ok) {
        const int lpp = E4 <= 8 ? 8 : E4 <= 16 ? 16 : E4 <= 32 ? 32 : 64;
        blocks = small ? (a.B * lpp / 64 + 3) / 4 + 1 : (nchunks + 3) / 4;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        dim3 grid((unsigned)blocks);
        const char **nm = &h->last_kernel;
        if (E4 == 8) launch_c4<8, true, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 == 16) launch_c4<16, true, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 == 32) launch_c4<32, true, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 == 64) launch_c4<64, true, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 < 8) launch_c4<8, false, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 < 16) launch_c4<16, false, BYDISH>(a, pf, nt, small, grid, st, nm);
        else if (E4 < 32) launch_c4<32, false, BYDISH>(a, pf, nt, small, grid, st, nm);
        else launch_c4<64, false, BYDISH>(a, pf, nt, small, grid, st, nm);
    } else if (a.C <= 8 && (a.E % 4 == 0) && E4 <= 64 && !a.hv && a.B > 8192 && h->opt_variant != 9) {
        blocks = (nchunks + 3) / 4;
        if (blocks > cap) blocks = cap;
        dim3 grid((unsigned)blocks);
        h->last_kernel = "m2d_score_pairs_cn";
#define M2D_CN(L)                                                                                        \
    do {                                                                                                 \
        if (nt) hipLaunchKernelGGL((m2d_score_pairs_cn<L, BYDISH, true>), grid, dim3(256), 0, st, a);    \
        else hipLaunchKernelGGL((m2d_score_pairs_cn<L, BYDISH, false>), grid, dim3(256), 0, st, a);      \
    } while (0)
        if (E4 <= 8) M2D_CN(8);
        else if (E4 <= 16) M2D_CN(16);
        else if (E4 <= 32) M2D_CN(32);
        else M2D_CN(64);
#undef M2D_CN
    } else {
        if (a.C > 64) {
            h->last_error = "num_categories > 64 is not supported";
            return M2D_ERR_UNSUPPORTED;
        }
        blocks = (a.B + 3) / 4;
        if (blocks > cap) blocks = cap;
        dim3 grid((unsigned)blocks);
        h->last_kernel = "m2d_score_pairs_generic";
        hipLaunchKernelGGL((m2d_score_pairs_generic<BYDISH>), grid, dim3(256), 0, st, a);
    }
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}

}  // namespace

int m2d_ensure_user_high(m2d_engine *h, hipStream_t stream)
{
    if (h->user_high_valid) return M2D_OK;
    if (!h->user_high) M2D_HIP_TRY(h, hipMalloc((void **)&h->user_high, (size_t)h->U * 4 * sizeof(float)));
    const int E4 = h->E >> 2;
    const int lpp = E4 <= 8 ? 8 : E4 <= 16 ? 16 : E4 <= 32 ? 32 : 64;
    int64_t blocks = (h->U * lpp / 64 + 3) / 4 + 1;
    const int64_t cap = (int64_t)h->num_cu * 16;
    if (blocks > cap) blocks = cap;
    dim3 grid((unsigned)blocks);
    if (lpp == 8) hipLaunchKernelGGL((m2d_build_user_high<8>), grid, dim3(256), 0, stream, h->pm, h->ce, h->U, h->E, h->user_high);
    else if (lpp == 16) hipLaunchKernelGGL((m2d_build_user_high<16>), grid, dim3(256), 0, stream, h->pm, h->ce, h->U, h->E, h->user_high);
    else if (lpp == 32) hipLaunchKernelGGL((m2d_build_user_high<32>), grid, dim3(256), 0, stream, h->pm, h->ce, h->U, h->E, h->user_high);
    else hipLaunchKernelGGL((m2d_build_user_high<64>), grid, dim3(256), 0, stream, h->pm, h->ce, h->U, h->E, h->user_high);
    M2D_HIP_TRY(h, hipGetLastError());
    h->user_high_valid = true;
    return M2D_OK;
}

int m2d_launch_score_pairs(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                           bool by_dish, int64_t B, float *out, hipStream_t stream, bool use_ingredients)
{
    if (B == 0) return M2D_OK;
    {
        const int rc = m2d_ensure_finite_scan(h, stream);
        if (rc != M2D_OK) return rc;
    }
    ScoreArgs a;
    a.nonfinite = h->nonfinite_dev;
    a.hv = use_ingredients ? h->dish_high : nullptr;
    a.pm = h->pm; a.re = h->re; a.ce = h->ce;
    a.users = users; a.items = items; a.cats = cats; a.out = out;
    a.B = B; a.U = h->U; a.I = h->I; a.user_base = h->user_base;
    a.E = h->E; a.C = h->C; a.a = h->a; a.b = h->b; a.err = h->err_dev;
    a.skip_masked = h->opt_skip_masked;
    a.uh = nullptr;
    // opt-in ("user_high_table"): batches large enough to pay for a pass over U_high take the high-level sum from the derived
    // table (rebuilt when Personal_Memory / Category_Embedding changed); smaller ones never do, so a call's scores depend
    // on its inputs and its size only, not on what ran before
    if (!use_ingredients && h->opt_user_high && h->C == 4 && h->E % 4 == 0 && h->E <= 256 && h->opt_variant != 9 &&
        (h->opt_prefetch == 2 || h->opt_prefetch == 4) && h->opt_nt != 0 && B >= (1 << 18)) {
        int rc = m2d_ensure_user_high(h, stream);
        if (rc != M2D_OK) return rc;
        a.uh = h->user_high;
    }
    return by_dish ? launch_any<true>(h, a, stream) : launch_any<false>(h, a, stream);
}
