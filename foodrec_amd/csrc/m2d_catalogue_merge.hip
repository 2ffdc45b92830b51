// What follows a pattern-grouped (or dense) scan: the dish ranges' partial lists merged into a user's list, lists shorter than k
// completed, tied users listed for the repair, and near-tied lists finished in the repair's plain-f32 arithmetic (m2d_topk_refine).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

// nsplit sorted partial lists per user -> final top-k.  LPU lanes per user (LPU = nsplit rounded up to a power of two,
// <= 64), lane w holding the head of split w's list; each of the k rounds is an argmax over the group by xor
// shuffles and the winning lane steps to its next entry.  (The first form of this kernel walked all nsplit * k
// entries from ONE thread through a scratch-memory pointer array: ~0.2 ms for a single user with 64 splits, most
// of that call's latency.)  Splits cover increasing dish ranges, so on equal scores the lower split (= lower lane)
// wins, which keeps ties in ascending-id order.
// (returns, in the lane that finished a user's list, the user's refinement word -- see m2d_topk_refine -- or -1)
template <int LPU>
__device__ __forceinline__ int32_t merge_splits_body(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k,
                                                     float *out_scores, int32_t *out_ids, const float *tie_in, float *tie_out,
                                                     int32_t *tie_list, int64_t I, const float *ex_in, float *ex_out,
                                                     const float *plan, int32_t *rcount)
{
    int32_t ret_ent = -1;
    const int lane = threadIdx.x & 63, w = lane & (LPU - 1);
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPU;
    const bool live = u < nU && w < nsplit;
    const float *s = ps + ((size_t)(live ? u : 0) * nsplit + (live ? w : 0)) * k;
    const int32_t *id = pi + ((size_t)(live ? u : 0) * nsplit + (live ? w : 0)) * k;
    int ptr = 0;
    float hs = live ? s[0] : 0.f;
    int32_t hi = live ? id[0] : -1;                          // -1: this list is exhausted (or the lane is idle)
    float last = 0.f;
    int32_t last_i = -1;
    // (the final pass of a call that finishes near-tied lists -- m2d_topk_refine -- also does m2d_topk_refine_flag's work)
    const float d2 = (plan && rcount && u < nU) ? 2.f * plan[(size_t)u * 8 + 7] : 0.f;
    int n_real = 0;
    bool near = false;
    for (int o = 0; o < k; ++o) {                           // wave-uniform trip count: the shuffles see a full EXEC
        float bs = hs;
        int32_t bi = hi;
        int bw = w;
        // The group's best head: the other candidate wins if this one is exhausted, or it ranks strictly ahead (`ahead`: NaN
        // after every number), or ties from a lower split.  That is a total order, so it is one unsigned 64-bit key -- high
        // word: the score's ordered image (+-0 alike, NaN = 1, an exhausted list = 0), low word: LPU - 1 - split -- and the
        // reduction is a max of keys with the head (score, id) as payload.  Partners inside a row of 16 lanes come by DPP (one
        // VALU each; 8 and 4 by mirror images, which reach the same maximum): the xor butterfly of three ds_bpermute and
        // twenty VALU per step was what this kernel spent its time on (all 64 lanes work for 64 / LPU users).
        uint32_t kh;
        {
            const float z = hs + 0.f;                        // -0 -> +0
            const int32_t b = __float_as_int(z);
            const uint32_t ord = (uint32_t)(b ^ ((b >> 31) & 0x7fffffff)) ^ 0x80000000u;      // order-preserving, > 1 for every number (-inf: 0x007fffff)
            kh = hi < 0 ? 0u : (z != z ? 1u : ord);
        }
        uint32_t kl = (uint32_t)(LPU - 1 - w);
#pragma unroll
        for (int off = LPU / 2; off >= 1; off >>= 1) {
            float os;
            int32_t oi;
            uint32_t oh, ol;
            if (off >= 16) {
                os = __shfl_xor(bs, off, 64); oi = __shfl_xor(bi, off, 64);
                oh = (uint32_t)__shfl_xor((int)kh, off, 64); ol = (uint32_t)__shfl_xor((int)kl, off, 64);
            } else {
                constexpr int Q1 = 0xB1, Q2 = 0x4E, HM = 0x141, RM = 0x140;      // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
#define M2D_DPP_PARTNER(x) (off == 1 ? __builtin_amdgcn_update_dpp(0, (x), Q1, 0xf, 0xf, false) : off == 2 ? __builtin_amdgcn_update_dpp(0, (x), Q2, 0xf, 0xf, false) : off == 4 ? __builtin_amdgcn_update_dpp(0, (x), HM, 0xf, 0xf, false) : __builtin_amdgcn_update_dpp(0, (x), RM, 0xf, 0xf, false))
                os = __int_as_float(M2D_DPP_PARTNER(__float_as_int(bs)));
                oi = M2D_DPP_PARTNER(bi);
                oh = (uint32_t)M2D_DPP_PARTNER((int)kh);
                ol = (uint32_t)M2D_DPP_PARTNER((int)kl);
#undef M2D_DPP_PARTNER
            }
            const bool take = (((unsigned long long)oh << 32) | ol) > (((unsigned long long)kh << 32) | kl);
            if (take) { bs = os; bi = oi; kh = oh; kl = ol; }
        }
        bw = LPU - 1 - (int)kl;
        if (u < nU && w == 0) {
            out_scores[u * k + o] = bi >= 0 ? bs : __builtin_nanf("");
            out_ids[u * k + o] = bi;
        }
        if (live && bi >= 0 && bw == w) {                   // this lane's head was taken: step to its next entry
            ++ptr;
            hi = ptr < k ? id[ptr] : -1;
            hs = ptr < k ? s[ptr] : 0.f;
        }
        if (bi >= 0 && bs == bs) {
            near = near || (n_real > 0 && last - bs < d2);
            ++n_real;
        }
        last = bs;
        last_i = bi;
    }
    // The merged list's tie value (see grouped_publish): its last score if a split's own tie value is that score, or a
    // score left at the head of some split's list equals it -- which of the tied dishes made the list was then decided
    // by split order (the scan order of the pattern-grouped kernels), not by dish id.  The dense kernels pass no tie_in.
    if (tie_in) {
        const bool t = live && last_i >= 0 && ((hi >= 0 && hs == last) || tie_in[(size_t)u * nsplit + w] == last);
        int tv = t ? 1 : 0;
#pragma unroll
        for (int off = LPU / 2; off >= 1; off >>= 1) tv |= __shfl_xor(tv, off, 64);
        const bool any = tv != 0;
        bool to_repair = false;
        int refine_ent = -1;
        if (ex_in) {
            // what the merged list leaves out (LeftOut, grouped_publish): the best two of every split's own left-out scores and
            // of what this merge left behind in the splits' lists (two entries of each suffice)
            LeftOut o = M2D_LEFTOUT_NONE;
            if (live) {
                const float *e = ex_in + ((size_t)u * nsplit + w) * 8;
                o.s1 = fmaxf(e[0], -INFINITY); o.i1 = __float_as_int(e[1]); o.s2 = fmaxf(e[2], -INFINITY); o.i2 = __float_as_int(e[3]);
                o.s3 = fmaxf(e[4], -INFINITY);
                for (int q = 0; q < 3; ++q)                  // what this merge left behind in the split's list: three entries suffice
                    if (hi >= 0 && ptr + q < k && id[ptr + q] >= 0) left_out_merge(o, s[ptr + q], id[ptr + q]);
            }
#pragma unroll
            for (int off = LPU / 2; off >= 1; off >>= 1) {
                const float o1 = __shfl_xor(o.s1, off, 64), o2 = __shfl_xor(o.s2, off, 64), o3 = __shfl_xor(o.s3, off, 64);
                const int32_t j1 = __shfl_xor(o.i1, off, 64), j2 = __shfl_xor(o.i2, off, 64);
                left_out_merge(o, o1, j1);
                left_out_merge(o, o2, j2);
                left_out_merge(o, o3, -1);
            }
            if (u < nU && w == 0) {
                float *e = ex_out + (size_t)u * 8;
                e[0] = o.s1; e[1] = __int_as_float(o.i1); e[2] = o.s2; e[3] = __int_as_float(o.i2); e[4] = o.s3;
            }
            // m2d_topk_refine_flag's decision, from the values at hand.  A tie at the list's end IS a near-tie (gap 0, the tied
            // dish among the left-out ones): with the refinement on, it is settled there -- the same (score desc, id asc) ranking
            // in the same arithmetic, over the handful of dishes that can matter instead of the user's patterns -- and the repair
            // is left with the users that have three or more dishes that close (copies of dishes, all-zero users)
            if (rcount && plan && n_real > 0) {
                int nex = 0;
                if (n_real == k) {
                    const float lim = last - d2;
                    nex = (o.s1 >= lim && o.i1 >= 0 ? 1 : 0) + (o.s2 >= lim && o.i2 >= 0 ? 1 : 0);
                    to_repair = o.s3 >= lim;
                }
                if (!to_repair && (near || nex > 0)) refine_ent = (int32_t)u | (nex << 30);
            }
        }
        if (u < nU && w == 0) {
            tie_out[u] = any ? last : __builtin_nanf("");
            // the final pass of a pattern-grouped call also does m2d_topk_tie_compact's work: a tied user joins the repair's
            // list, everybody else's list is finished here (this lane wrote it: its own stores, in program order)
            if (tie_list) {
                if (to_repair || (any && !(rcount && plan))) tie_list[1 + atomicAdd(&tie_list[0], 1)] = (int32_t)u;
                else fill_absent_user(out_scores + u * k, out_ids + u * k, k, I);
                if (to_repair) atomicAdd(&rcount[1], 1);
            }
        }
        if (rcount && tie_list && u < nU && w == 0) {
            rcount[8 + u] = refine_ent;                      // a word per user, no atomics (m2d_topk_refine compacts)
            ret_ent = refine_ent;
        }
    }
    return ret_ent;
}

template <int LPU>
__global__ __launch_bounds__(256) void m2d_topk_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k,
                                                             float *out_scores, int32_t *out_ids, const float *tie_in, float *tie_out,
                                                             int32_t *tie_list, int64_t I, const float *ex_in = nullptr, float *ex_out = nullptr,
                                                             const float *plan = nullptr, int32_t *rcount = nullptr)
{
    (void)merge_splits_body<LPU>(ps, pi, nU, nsplit, k, out_scores, out_ids, tie_in, tie_out, tie_list, I, ex_in, ex_out, plan, rcount);
}

// Users with fewer than k ranked dishes (every finite-scored dish is already in their list, the rest of the
// catalogue scored NaN -- an empty category mask, Model_Recommender.py:79 -- or -inf): append the dishes not
// in the list in ascending id with a NaN score, which is where heapq.nlargest-style "NaN last" puts them.
__global__ void m2d_topk_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I)
{
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < nU) fill_absent_user(scores + u * k, ids + u * k, k, I);
}

// the users whose final tie value is set (not NaN), as a list: [0] count, [1 + f] position in the call
// -- and the lists of everybody else finished on the way (m2d_topk_fill_absent's work; a listed user's list is finished by
// the kernel that rewrites it)
__global__ __launch_bounds__(256) void m2d_topk_tie_compact(const float *tie_final, int64_t nU, int32_t *tie_list, float *scores,
                                                            int32_t *ids, int k, int64_t I, int refined)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= nU) return;
    // refined: m2d_topk_refine_flag decides about the tied users too (they are near-tied lists with a gap of 0)
    if (!refined && tie_final[u] == tie_final[u]) tie_list[1 + atomicAdd(&tie_list[0], 1)] = (int32_t)u;
    else fill_absent_user(scores + u * k, ids + u * k, k, I);
}

// ---- m2d_topk_refine: near-tied lists are finished in the tie repair's arithmetic ----------------------------------------------
// A scan kernel's score s~ lies within delta_u (plan record word 7) of the same score c in the repair's plain-f32 arithmetic
// (m2d_topk_repair_scan: the ranking every kernel's lists are DEFINED by).  Where every gap between neighbouring entries of a
// user's final list, and between its last entry and the best score left out, is above 2 delta_u, the list is already c's
// ranking.  Elsewhere -- a few per cent of the users -- the candidates (the list, plus the best left-out dish if it is that
// close) are scored again in that arithmetic, step for step (pattern sums of the low-level rows, a float4 column per lane,
// the 16-lane rotation sum, m2d_blend_unfused), and sorted (score desc, id asc); a user with TWO left-out scores that close
// joins the repair's list and is re-ranked over its relevant patterns.  So the split-bf16 kernel (the default) and the
// exact-f32 kernel return the same dish ids: the difference between their products only ever decided near-ties.
// One wave per user; most leave after reading their list.  Tie-listed users are skipped (the repair rewrites them).
// one thread per user: is the list near-tied?  -> the user's word at counter[8 + u] (position in the call | left-out dishes to take
// along << 30, or -1), or -- three left-out scores that close -- the repair's list.  (Launches with dish ranges decide this in their last merge pass.)
__device__ __forceinline__ int32_t refine_flag_user(const RefineArgs &p, const int64_t u)
{
    const int k = p.k;
    const float *os = p.out_scores + (size_t)u * k;
    const int32_t *oi = p.out_ids + (size_t)u * k;
    const float d2 = 2.f * p.plan[(size_t)u * 8 + 7];
    int n = 0;
    bool near = false;
    float prev = 0.f;
    for (int q = 0; q < k; ++q) {                           // ranked entries come first (NaN / absent ones behind them)
        const float sv = os[q];
        if (oi[q] < 0 || sv != sv) break;
        near = near || (q > 0 && prev - sv < d2);
        prev = sv;
        ++n;
    }
    int nex = 0;
    if (n == k) {                                           // a full list: how many left-out scores are within 2 delta of its last
        const float *ex = p.ex + (size_t)u * 8;
        const float lim = prev - d2;
        nex = (ex[0] >= lim && __float_as_int(ex[1]) >= 0 ? 1 : 0) + (ex[2] >= lim && __float_as_int(ex[3]) >= 0 ? 1 : 0);
        if (ex[4] >= lim) {                                 // three of them: the repair ranks this user over its patterns
            p.tie_list[1 + atomicAdd(&p.tie_list[0], 1)] = (int32_t)u;
            atomicAdd(&p.counter[1], 1);
            return -1;
        }
    }
    if (n == 0 || !(near || nex > 0)) return -1;
    return (int32_t)u | (nex << 30);
}

__global__ __launch_bounds__(256) void m2d_topk_refine_flag(RefineArgs p)
{
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= p.nU) return;
    p.counter[8 + u] = refine_flag_user(p, u);
}

// 32 lanes per listed user, a lane per candidate (k <= 16 list entries + at most two left-out dishes): the candidate's score in
// the repair's arithmetic.  A lane emulates the sixteen lanes the repair gives a dish: partial j = the fma chain over float4
// columns j, j + 16, ... of the row (it.x w.x first ... as there), then the rotation sum's tree -- (p_j + p_j+8), then + the
// same of j + 4, of j + 2, of j + 1; the adds commute, so the tree does not depend on the rotations' direction.
// the listed users of a block (s_list[0 .. count): position in the call | left-out dishes to take along << 30), 32 lanes each
template <int CH>                                           // float4 columns per emulated lane: E <= 64 CH
__device__ __forceinline__ void refine_listed_users(const RefineArgs &p, const int32_t *s_list, const int count, v4f (*s_w)[32])
{
    constexpr int C = 4;
    const int lane = threadIdx.x & 63, half = lane >> 5, i = lane & 31;
    const int k = p.k, E = p.E, E4 = E >> 2, W = (C + 1) * E;
    for (int f = (threadIdx.x >> 6) * 2 + half; f < ((count + 1) & ~1); f += 8) {
        const bool fvalid = f < count;                      // (both halves of a wave run the same trip count: the shuffles below see a full EXEC)
        const int32_t ent = fvalid ? s_list[f] : 0;
        const int64_t u = ent & 0x3fffffff;
        const int nex = fvalid ? (int)((uint32_t)ent >> 30) : 0;
        float *os = p.out_scores + (size_t)u * k;
        int32_t *oi = p.out_ids + (size_t)u * k;
        const float sv = (fvalid && i < k) ? os[i] : 0.f;
        const int32_t iv = (fvalid && i < k) ? oi[i] : -1;
        const bool real = i < k && iv >= 0 && sv == sv;
        const unsigned long long rb = __ballot(real);
        const int n = __builtin_popcount((uint32_t)(rb >> (32 * half)));
        const int nc = n + nex;
        const float *ex = p.ex + (size_t)u * 8;
        int32_t d = iv;
        if (fvalid && i >= n && i < nc) d = __float_as_int(ex[1 + 2 * (i - n)]);
        const bool mine = fvalid && i < nc && d >= 0;       // (a left-out record always names its dishes: the guard costs nothing and keeps
                                                            //  the loads below inside the tables whatever a record holds)
        float sc = -INFINITY;
        int64_t ul = fvalid ? (int64_t)p.users[u] - p.user_base : 0;
        if (ul < 0 || ul >= p.U) ul = 0;                    // latched by the scan kernel
        const v4f *um4 = reinterpret_cast<const v4f *>(p.pm + (size_t)ul * W);
        float hc[C];                                        // <U_high, CE_c>: the plan's words, as in the planned repair (repair_alpha)
#pragma unroll
        for (int c = 0; c < C; ++c) hc[c] = p.plan[(size_t)u * 8 + 1 + c];
        int pt = 0;
        if (mine) {
            const v4f m = *reinterpret_cast<const v4f *>(p.cats + (size_t)d * C);
            pt = (m.x != 0.f ? 1 : 0) | (m.y != 0.f ? 2 : 0) | (m.z != 0.f ? 4 : 0) | (m.w != 0.f ? 8 : 0);
        }
        const v4f *row = reinterpret_cast<const v4f *>(p.re) + (size_t)(mine ? d : 0) * E4;
        v4f it[16 * CH];                                    // the candidate's row: fetched once, before the pattern loop
#pragma unroll
        for (int q = 0; q < 16 * CH; ++q) it[q] = (mine && q < E4) ? row[q] : v4f{0.f, 0.f, 0.f, 0.f};
        // The low-level operand w_P = sum of the pattern's U_low rows is the same for every candidate of a pattern: the 32 lanes
        // of a user work it out a float4 column each (as the repair's wp table: the rows added in category order from zero) and
        // pass it through LDS, pattern by pattern -- a user's candidates carry one or two patterns, almost always
        bool todo = mine && pt != 0;
        v4f *wb = &s_w[(threadIdx.x >> 5)][0];
        for (;;) {
            const unsigned long long tm = __ballot(todo);
            if (tm == 0ull) break;                          // wave-uniform
            const uint32_t hm = (uint32_t)(tm >> (32 * half));
            const int cur = __shfl(pt, half * 32 + (hm ? __builtin_ctz(hm) : 0), 64) * (hm ? 1 : 0);      // this half's pattern of the round (0: none)
            if (i < E4 && cur) {
                v4f w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const v4f z = {0.f, 0.f, 0.f, 0.f};
                    const v4f r = um4[(c + 1) * E4 + i];
                    w += ((cur >> c) & 1) ? r : z;
                }
                wb[i] = w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (todo && pt == cur) {
                float part[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    float lo = 0.f;
#pragma unroll
                    for (int ch = 0; ch < CH; ++ch) {
                        const int q = jj + 16 * ch;
                        if (q < E4) {
                            const v4f w = wb[q];
                            lo = fmaf(it[q].x, w.x, fmaf(it[q].y, w.y, fmaf(it[q].z, w.z, fmaf(it[q].w, w.w, lo))));
                        }
                    }
                    part[jj] = lo;
                }
                float s8[8], s4[4], s2[2];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) s8[jj] = part[jj] + part[jj + 8];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) s4[jj] = s8[jj] + s8[jj + 4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) s2[jj] = s4[jj] + s4[jj + 2];
                const float lo = s2[0] + s2[1];
                const float npat = (float)__builtin_popcount(pt);
                sc = fmaxf(repair_score_planned(repair_alpha(p.a, hc, pt), p.b, lo / npat), -INFINITY);
                todo = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();                // (the next round overwrites wb)
        }
        // a candidate's rank = how many candidates of its user are ahead of it in (score desc, id asc)
        int rank = 0;
        for (int o = 0; o < 18; ++o) {                      // (k <= 16, two more at most)
            const float s2v = __shfl(sc, half * 32 + o, 64);
            const int32_t i2v = __shfl(d, half * 32 + o, 64);
            rank += (o < nc && (s2v > sc || (s2v == sc && i2v < d))) ? 1 : 0;
        }
        if (mine && rank < n) { os[rank] = sc; oi[rank] = d; }
    }
}


template <int CH>
__global__ __launch_bounds__(256) void m2d_topk_refine(RefineArgs p)
{
    // a block takes 64 users: their words are compacted in LDS (a list with one global counter cost 40 us of serialised atomics)
    __shared__ int32_t s_list[64];
    __shared__ int s_count;
    __shared__ v4f s_w[8][32];                              // per half-wave: the pattern's low-level operand, a float4 column per lane
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int64_t uu = (int64_t)blockIdx.x * 64 + threadIdx.x;
        const int32_t e = uu < p.nU ? p.counter[8 + uu] : -1;
        if (e != -1) s_list[atomicAdd(&s_count, 1)] = e;
    }
    __syncthreads();
    const int count = s_count;
    if (count == 0) return;
    if (threadIdx.x == 0) atomicAdd(&p.counter[0], count);
    refine_listed_users<CH>(p, s_list, count, s_w);
}

}  // namespace

void m2d_launch_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *out_s, int32_t *out_i,
                             hipStream_t st, const float *tie_in, float *tie_out, int32_t *tie_list,
                             int64_t I, const float *ex_in, float *ex_out, const float *plan,
                             int32_t *rcount)
{
    int lpu = 1;
    while (lpu < nsplit) lpu <<= 1;
    const unsigned grid = (unsigned)((nU * lpu + 255) / 256);
#define M2D_MERGE(L) if (lpu == L) hipLaunchKernelGGL(m2d_topk_merge_splits<L>, dim3(grid), dim3(256), 0, st, ps, pi, nU, nsplit, k, out_s, out_i, tie_in, tie_out, tie_list, I, ex_in, ex_out, plan, rcount);
    M2D_MERGE(1) M2D_MERGE(2) M2D_MERGE(4) M2D_MERGE(8) M2D_MERGE(16) M2D_MERGE(32) M2D_MERGE(64)
#undef M2D_MERGE
}

// More than 64 partial lists per user: two passes of the merge above -- groups of 64 consecutive splits first (a
// "user" of that pass is one (user, group)), then the per-group winners.  nsplit must be a multiple of 64 then;
// tmp_s / tmp_i hold nU * (nsplit / 64) * k entries.  Consecutive groups are consecutive dish ranges, so the
// lower-split-wins tie rule carries through both passes.
// tie: [nU * nsplit] values of the splits, then room for the nU * (nsplit / 64) of the first pass, then the nU final ones
// ex: [nU * nsplit] x 8 floats of the splits' left-out scores (LeftOut), then room for the first pass's nU * (nsplit / 64), then the
// nU final ones -- laid out like `tie`; null = not kept
void m2d_launch_merge_splits2(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *tmp_s, int32_t *tmp_i,
                              float *out_s, int32_t *out_i, hipStream_t st, float *tie, float *tie_final, int32_t *tie_list, int64_t I,
                              float *ex, float *ex_final, const float *plan, int32_t *rcount)
{
    if (nsplit <= 64) {
        m2d_launch_merge_splits(ps, pi, nU, nsplit, k, out_s, out_i, st, tie, tie_final, tie_list, I, ex, ex_final, plan, rcount);
        return;
    }
    const int G = nsplit / 64;
    float *tie_mid = tie + (size_t)nU * nsplit;
    float *ex_mid = ex ? ex + (size_t)nU * nsplit * 8 : nullptr;
    m2d_launch_merge_splits(ps, pi, nU * G, 64, k, tmp_s, tmp_i, st, tie, tie_mid, nullptr, 0, ex, ex_mid);      // a "user" of this pass is (user, group)
    m2d_launch_merge_splits(tmp_s, tmp_i, nU, G, k, out_s, out_i, st, tie_mid, tie_final, tie_list, I, ex_mid, ex_final, plan, rcount);
}

void m2d_topk_launch_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I, hipStream_t st)
{
    hipLaunchKernelGGL(m2d_topk_fill_absent, dim3((unsigned)((nU + 127) / 128)), dim3(128), 0, st, scores, ids, nU, k, I);
}

void m2d_topk_launch_tie_compact(const float *tie_final, int64_t nU, int32_t *tie_list, float *scores, int32_t *ids, int k, int64_t I,
                                 int refined, hipStream_t st)
{
    hipLaunchKernelGGL(m2d_topk_tie_compact, dim3((unsigned)((nU + 255) / 256)), dim3(256), 0, st, tie_final, nU, tie_list, scores, ids, k, I,
                       refined);
}

// near-tied lists: finished in the repair's arithmetic (may add to the repair's list).  flag_pass: a launch with ONE dish range has
// no merge pass to decide who is near-tied -- m2d_topk_refine_flag does
void m2d_topk_launch_refine(const RefineArgs &f, bool flag_pass, hipStream_t st)
{
    if (flag_pass) hipLaunchKernelGGL(m2d_topk_refine_flag, dim3((unsigned)((f.nU + 255) / 256)), dim3(256), 0, st, f);
    if (f.E <= 64) hipLaunchKernelGGL(m2d_topk_refine<1>, dim3((unsigned)((f.nU + 63) / 64)), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(m2d_topk_refine<2>, dim3((unsigned)((f.nU + 63) / 64)), dim3(256), 0, st, f);
}
