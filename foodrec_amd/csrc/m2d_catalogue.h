// Full-catalogue retrieval (m2d_topk_users) for gfx950: what its translation units share.
//
//   m2d_catalogue_dense.hip       dish vectors Dt[d]; m2d_topk_mfma (dense [users x (C+1)E] . [(C+1)E x dishes], exact f32: weighted
//                            masks, k > 16, the ingredient table beyond E = 64), m2d_topk_generic (any shape, one block per user)
//   m2d_catalogue_plan.hip        0/1 masks: the pattern-sorted dish table, the call's plan (per-user bounds, relevant patterns, the sort,
//                            the launch order), the launcher of a pattern-grouped call, m2d_launch_topk_users' dispatch
//   m2d_catalogue_scan_f32.hip    m2d_topk_grouped: the pattern-grouped scan on v_mfma_f32_32x32x2_f32 (exact f32; zero-padded widths)
//   m2d_catalogue_scan_bf16.hip   m2d_topk_grouped_bf16 / _bf16_pipe2: the same scan on split-bf16 MFMA (the default, E = 64 / 128)
//   m2d_catalogue_merge.hip       dish ranges' partial lists -> a user's list; near-tied lists finished in plain f32 (m2d_topk_refine)
//   m2d_catalogue_repair.hip      users whose k-th score is tied three ways or more: re-ranked over their patterns in id order
//
// Reference behaviour all of it reproduces: score = Model_Recommender.py:67-96 per (user, dish), ranking = heapq.nlargest
// (evaluate.py:63: score descending, ties to the lower dish id, NaN last).
#pragma once

#include <math.h>

#include <type_traits>

#include "m2d_engine.h"

// What is written is what runs: no floating-point contraction in this file.  hipcc's default (-ffp-contract=fast) fuses a
// multiply into a following add wherever it sees one -- ALSO through __fmul_rn / __fadd_rn, and not in every copy of an
// unrolled loop: the repair scan's blend a * alpha + b * low came out as v_pk_mul + v_add for the first of a group's two
// dishes in flight and as v_mul + v_fmac (one rounding fewer) for the second, so a re-ranked user's last score bit depended on
// which of the two places a dish landed in (found when the scan's dish order began to depend on the listed users' masks).
// Every fused multiply-add in this file is an explicit fmaf or an MFMA.
#pragma clang fp contract(off)

// Timing-only ablation hooks for scripts/diag/topk_diag.cpp (never defined in the product build):
// bit 0 = no epilogue, bit 1 = no LDS-DMA refill, bit 2 = no per-stage barrier/wait, bit 3 = epilogue
// fast path only (no insertions).  Outputs are wrong.
#ifndef M2D_DIAG
#define M2D_DIAG 0
#endif
__attribute__((unused)) static unsigned long long *g_m2d_diag_buffer = nullptr;   // set by scripts/diag only
#if M2D_DIAG & 16
#define STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#else
#define STAMP(x)
#endif

#define M2D_INTERNAL __attribute__((visibility("hidden")))

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ---- argument blocks that cross translation units (global namespace: one type for every unit) ------------------------------
struct GroupedArgs {
    const float *pm;         // [U, (C+1) E]
    const float *ce;         // [C, E]
    const float *rs;         // [slots, EW]  Recipe_Embedding rows sorted by (pattern, norm bucket, dish id); with the
                             //              ingredient extension [H[d] | RE[d]], EW = 2 E
    const __bf16 *rs16;      // the same rows as split bf16 (hi | lo blocks per 32-row tile)
    const int32_t *perm;     // [slots]      slot -> dish id (-1 = padding)
    const int32_t *tile_info;
    const int32_t *grp;      // [0..15] first slot of each pattern's group, [40..55] rows per pattern
    const int32_t *users;
    int64_t nU, U, user_base;
    int32_t k, nsplit;
    int64_t tiles;
    float a, b;
    float *out_scores;
    int32_t *out_ids;
    int32_t *err;
    unsigned long long *dbg;   // scripts/diag only
    int32_t e_real;            // padded form only: the tables' E (rows of `rs` are zero-padded to the kernel's E)
    int32_t prog_limit;        // hi x hi first form, E = 64: polls of the progress words before a wave gives up (M2D_ERR_KERNEL_TIMEOUT)
    float *tie_val;            // [nU, nsplit] the list's last score when a tie decides what it holds (tie_at_boundary), else NaN
    const float *plan;         // [nU, 8] per user of the call: scan-start bound, <U_high, CE_c> x 4, relevant-pattern mask (m2d_topk_user_plan)
    const int32_t *order;      // [nU] position in the launch -> index into users / plan (users sorted by pattern mask), or null
    unsigned long long *tiles_scanned;   // diagnostic: 32-dish tiles the blocks stepped through
    const int32_t *items;      // [user blocks x nsplit] launch order of a pruned scan: block * nsplit + split, longest first; or null
    float *ex_out;             // [nU, nsplit, 8] per (user, dish range): the best two scores left out of the range's list with their dish ids
                               // (bits), the third best such score -- what m2d_topk_refine needs to finish near-tied lists; null = not kept
    int32_t *shared_thr;       // word 6 of the plan records (stride 8): the user's running threshold over ALL dish ranges, as an
                               // ordered key (thr_key); null = every (block, range) item keeps to its own lists
};

struct RepairArgs {
    const float *pm, *re, *ce, *cats, *hv;      // hv: per-dish high-level vectors of the ingredient extension, or null
    const int32_t *users;
    const int32_t *tie_list;                    // [0] listed users, [1 + f] position of listed user f in the call
    int64_t nU, U, I, user_base;
    int32_t C, E, k;
    int32_t cap;                                // listed users the scan / merge pair handles (the rest: m2d_topk_repair_rest)
    float a, b;
    const float *rows;                          // the pattern-sorted f32 dish table (GroupedArgs::rs), row stride ew floats
    const int32_t *perm;                        // slot -> dish id
    const int32_t *grp;                         // [0..15] first slot of each pattern's group, [40..55] rows per pattern
    const float *plan;                          // the call's plan records (words 1-4: <U_high, CE_c>, word 5: relevant-pattern mask), or null
    int32_t ew;
    int32_t all_patterns;                       // 1: the scan reads every pattern's dishes whatever the masks say (A/B)
    float *part_s;                              // [cap, REPAIR_SPLITS, k] partial lists
    int32_t *part_i;
    float *out_scores;                          // [nU, k]
    int32_t *out_ids;
};

struct RefineArgs {
    const float *pm, *re, *ce, *cats, *plan, *tie_final, *ex;
    const int32_t *users;
    int32_t *tie_list, *counter;                            // counter: [0] users refined, [1] sent on to the repair, [2] length of the list at [8..]
    int64_t nU, U, I, user_base;
    int32_t E, k;
    float a, b;
    float *out_scores;
    int32_t *out_ids;
};

// a pattern-grouped scan launch: which instantiation (m2d_catalogue_scan_f32.hip / m2d_catalogue_scan_bf16.hip)
struct ScanShape {
    int E;           // kernel width: 32 / 64 / 128 / 256 floats per dish row (with the ingredient table: [H[d] | RE[d]], twice the embedding)
    int KR;          // list slots per lane: 10 or 16
    bool bf16x3;     // split-bf16 MFMA (E = 64 / 128); else exact f32
    bool hv;         // ingredient rows (pipelined split-bf16 kernel only)
    bool pad;        // dish rows zero-padded to E floats (exact f32 only)
    bool pipe;       // split bf16: the pipelined form (else the first form)
    int waves;       // waves per block: 8 (256 users) or 4 (128 users; pipelined split bf16, E = 64)
    bool keep;       // the lists' left-out scores are kept for m2d_topk_refine (GroupedArgs::ex_out)
    bool apx;        // pipelined split bf16, E = 64, 8 waves: hi x hi product first, cross products for the tiles with a candidate
};

// ---- launchers other units call (all enqueue on `st`; int results are M2D_* codes) ---------------------------------------------
M2D_INTERNAL int m2d_topk_dense_launch(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores, int32_t *out_ids,
                                       hipStream_t st);
M2D_INTERNAL int m2d_topk_scan_f32_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);
M2D_INTERNAL int m2d_topk_scan_bf16_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);
M2D_INTERNAL void m2d_launch_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *out_s, int32_t *out_i,
                                          hipStream_t st, const float *tie_in = nullptr, float *tie_out = nullptr, int32_t *tie_list = nullptr,
                                          int64_t I = 0, const float *ex_in = nullptr, float *ex_out = nullptr, const float *plan = nullptr,
                                          int32_t *rcount = nullptr);
M2D_INTERNAL void m2d_launch_merge_splits2(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *tmp_s, int32_t *tmp_i,
                                           float *out_s, int32_t *out_i, hipStream_t st, float *tie, float *tie_final, int32_t *tie_list,
                                           int64_t I, float *ex = nullptr, float *ex_final = nullptr, const float *plan = nullptr,
                                           int32_t *rcount = nullptr);
M2D_INTERNAL void m2d_topk_launch_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I, hipStream_t st);
M2D_INTERNAL void m2d_topk_launch_tie_compact(const float *tie_final, int64_t nU, int32_t *tie_list, float *scores, int32_t *ids, int k,
                                              int64_t I, int refined, hipStream_t st);
M2D_INTERNAL void m2d_topk_launch_refine(const RefineArgs &f, bool flag_pass, hipStream_t st);
M2D_INTERNAL int m2d_topk_launch_repair(m2d_engine *h, const RepairArgs &r, bool hv, hipStream_t st);

namespace {

__device__ __forceinline__ bool ahead(float v, float w)
{
    // does v rank strictly before w?  NaN ranks after everything.
    return (v > w) || (w != w && v == v);
}

// Insert (x, id) into a descending register list of N slots and drop the last one -- for EVERY lane at once,
// with no per-lane predicate and no serial chain through the slots:
//     new[i] = med3(old[i-1], x, old[i])          (old[-1] = +inf)
// which is old[i-1] when x goes above slot i-1, x when it lands in slot i, and old[i] otherwise; a lane whose x
// does not beat its last slot is left unchanged.  The ids follow the same two compares.  Equal scores keep
// the earlier arrival first (x > old[i] is strict), so ties stay in ascending-id order.  A NaN x is demoted
// to -inf and can never enter.  About 4 VALU per slot, dependency depth 2 (the earlier compare-exchange
// sweep spent ~50 cycles per slot on VALU <-> mask round trips).
// v_cndmask_b32 with an explicit lane mask (hipcc turned the equivalent nested ?: into exec-masked branches)
__device__ __forceinline__ int32_t lane_select(unsigned long long mask, int32_t if_set, int32_t if_clear)
{
    int32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

template <int N>
__device__ __forceinline__ void sorted_insert(float (&ls)[N], int32_t (&li)[N], float x, int32_t id)
{
    x = fmaxf(x, -INFINITY);                     // maxNum: NaN -> -inf
    unsigned long long above[N];                 // lane masks: x ranks above slot i
#pragma unroll
    for (int i = 0; i < N; ++i) above[i] = __ballot(x > ls[i]);
    float ns[N];
    int32_t ni[N];
    ns[0] = fmaxf(ls[0], x);
    ni[0] = lane_select(above[0], id, li[0]);
#pragma unroll
    for (int i = 1; i < N; ++i) {
        ns[i] = __builtin_amdgcn_fmed3f(ls[i - 1], x, ls[i]);
        ni[i] = lane_select(above[i - 1], li[i - 1], lane_select(above[i], id, li[i]));
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ls[i] = ns[i];
        li[i] = ni[i];
    }
}

// sorted_insert written in place, from the last slot up: slot i takes med3(old[i-1], x, old[i]) while slots < i still
// hold their old values, so no copy of the list is made (4 VALU per slot) and any contiguous range of slots
// [lo, hi) can be done on its own -- the pipelined kernel spreads the ranges over the gaps between its MFMAs.
// Ranges must be applied from the highest slots down.
template <int N, int LO, int HI>
__device__ __forceinline__ void sorted_insert_range(float (&ls)[N], int32_t (&li)[N], const float x, const int32_t id)
{
#pragma unroll
    for (int i = HI - 1; i >= LO; --i) {
        const unsigned long long ab = __ballot(x > ls[i]);
        if (i > 0) {
            const unsigned long long ab1 = __ballot(x > ls[i - 1]);
            li[i] = lane_select(ab1, li[i - 1], lane_select(ab, id, li[i]));
            ls[i] = __builtin_amdgcn_fmed3f(ls[i - 1], x, ls[i]);
        } else {
            li[0] = lane_select(ab, id, li[0]);
            ls[0] = fmaxf(ls[0], x);
        }
    }
}

// (the scores of slots [lo, hi) are final HERE: keeps the v_med3 of a range in the MFMA gap it was written into -- the
//  compiler otherwise collects them behind the last MFMA of the step, where nothing hides them)
template <int N, int LO, int HI>
__device__ __forceinline__ void pin_range(float (&ls)[N])
{
#pragma unroll
    for (int i = LO; i < HI; ++i) asm volatile("" : "+v"(ls[i]));
}

template <int N>
__device__ __forceinline__ void sorted_insert_inplace(float (&ls)[N], int32_t (&li)[N], float x, const int32_t id)
{
    x = fmaxf(x, -INFINITY);                     // maxNum: NaN -> -inf
    sorted_insert_range<N, 0, N>(ls, li, x, id);
}

// Tie bookkeeping of the pattern-grouped kernels.  Their lists keep equal scores in SCAN order (mask pattern, row-norm
// bucket, dish id), heapq.nlargest keeps them in id order (evaluate.py:63).  Which dishes a user's final list holds
// differs between the two only if a score that fell off the end of a list -- or was refused at its end -- EQUALS the
// k-th score of the final list.  A lane therefore carries one bit, "such an event happened at the value my last entry
// holds now" (set by the event, cleared when the last entry rises: four VALU / SALU per insertion, candidate path
// only); where lists are merged the bit counts only if that value is the merged list's last one, and a score left
// behind by the merge that equals it counts too.  A user whose final k-th score is tied this way is re-ranked in id
// order by m2d_topk_repair_ties (on N(0, 1/E) tables: a few users in a million); ties that stay inside a list are put
// into id order when the list is finished (m2d_topk_fill_absent).  Thresholds are compared with >= so that an equal
// score reaches the insertion.
__device__ __forceinline__ bool tie_at_boundary(float x, float old_last, float new_last)
{
    return fminf(x, old_last) == new_last && new_last > -INFINITY;
}

__device__ __forceinline__ unsigned long long tie_update(unsigned long long mask, float x, float old_last, float new_last)
{
    return (mask & ~__ballot(new_last != old_last)) | __ballot(tie_at_boundary(x, old_last, new_last));
}

// What a lane's list leaves out (round 4, index-exact lists): the best and the second-best score that reached the lane's
// insertion and is not in its list -- refused at the list's end, or pushed off it.  A score within 2 delta of the user's final
// k-th score always reaches an insertion (the threshold compares are relaxed by 2 delta), so if such a score exists outside the
// final list, the largest one is here, whatever thresholds the launch's shape produced.
struct LeftOut {
    float s1;
    int32_t i1;                                            // slot of s1 (a dish id once published)
    float s2;
    int32_t i2;
    float s3;                                              // the third best such score (its dish is not kept: three that close go to the repair)
};
#define M2D_LEFTOUT_NONE LeftOut{-INFINITY, -1, -INFINITY, -1, -INFINITY}

// (score, id) into the best three of a LeftOut
__device__ __forceinline__ void left_out_merge(LeftOut &o, const float cs, const int32_t ci)
{
    const float c = fmaxf(cs, -INFINITY);                  // NaN (an empty slot): never
    const bool a1 = c > o.s1, a2 = c > o.s2;
    o.s3 = a2 ? o.s2 : fmaxf(o.s3, c);
    o.i2 = a1 ? o.i1 : (a2 ? ci : o.i2);
    o.s2 = a1 ? o.s1 : (a2 ? c : o.s2);
    o.i1 = a1 ? ci : o.i1;
    o.s1 = a1 ? c : o.s1;
}

// `keep_from` = the lane's last entry after the insertion, less 2 delta: the user's final k-th score is not below a lane's last
// entry, so a score under keep_from can never come within 2 delta of it -- nearly every pushed-off entry, as lists' gaps are a
// hundred times 2 delta.  One ballot then settles the wave (the bookkeeping itself is 14 VALU: it cost the scan 6 % when every
// insertion paid it).
__device__ __forceinline__ void left_out_note(LeftOut &o, const float x, const int32_t idx, const float old_last, const int32_t old_last_id,
                                              const float keep_from)
{
    const float es = fminf(x, old_last);                   // what is out after this insertion: x itself, or the entry it pushed off
    if (__ballot(es >= keep_from) == 0ull) return;          // wave-uniform
    const bool refused = !(x > old_last);                  // the insertion is strict: an equal score stays out
    left_out_merge(o, es >= keep_from ? es : -INFINITY, refused ? idx : old_last_id);
}

// (used by the kernels that finish a list: see m2d_topk_fill_absent below)
__device__ __forceinline__ void fill_absent_user(float *s, int32_t *id, const int k, const int64_t I)
{
    int n = 0;
    while (n < k && id[n] >= 0) ++n;
    // bit-equal scores inside the list: ascending dish id, as heapq.nlargest leaves them (evaluate.py:63); the
    // pattern-grouped kernels deliver them in scan order
    for (int q = 1; q < n; ++q) {
        for (int r = q; r > 0 && s[r - 1] == s[r] && id[r - 1] > id[r]; --r) {
            const int32_t t = id[r - 1];
            id[r - 1] = id[r];
            id[r] = t;
        }
    }
    for (int64_t d = 0; n < k && d < I; ++d) {
        bool present = false;
        for (int q = 0; q < n; ++q) present = present || (id[q] == (int32_t)d);
        if (!present) {
            id[n] = (int32_t)d;
            s[n] = __builtin_nanf("");
            ++n;
        }
    }
}

// The tie repair's scratch: REPAIR_SPLITS partial lists for each of up to REPAIR_CAP listed users (m2d_catalogue_repair.hip)
constexpr int REPAIR_SPLITS = 64, REPAIR_CAP = 1024;

__device__ __forceinline__ float row16_sum(float x)
{
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));
    return x;
}

// The ranking arithmetic of the tie repair and of m2d_topk_refine ("c" in their comments).  With the call's plan at hand the
// high-level part is alpha_P EXACTLY as the scan kernels form it -- a (hs inv_n) from the plan's <U_high, CE_c> words -- so that
// a scan kernel's score and c differ only by what the low-level contraction's arithmetic does (split bf16: ~1e-5 of |w||r|),
// not by two summation orders of the thirty-times larger high-level dot products: that is what keeps the near-tie margin
// (plan word 7) small enough for a few per cent of the users.  Without a plan (the first-form split-bf16 kernel, the
// ingredient table): the reference's blend of the two normalised sums, Model_Recommender.py:79, :92, :95-96.
__device__ __forceinline__ float repair_alpha(const float a, const float (&hc)[4], const int pt)
{
    float hs = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) hs += ((pt >> c) & 1) ? hc[c] : 0.f;
    const float inv_n = 1.0f / (float)__builtin_popcount(pt);
    return a * (hs * inv_n);                                // the scan kernels' expression, bit for bit (pattern 0: 0 * inf = NaN)
}

__device__ __forceinline__ float repair_score_planned(const float alpha_scan, const float b, const float low_over_n)
{
    const float q = b * low_over_n;                         // (no contraction in this file)
    return alpha_scan + q;
}

__device__ __forceinline__ bool repair_ahead(float s, int32_t i, float t, int32_t j)
{
    return i >= 0 && (j < 0 || s > t || (s == t && i < j));       // (score desc, id asc); id < 0: no entry
}

// =====================================================================================================
// Pattern-grouped retrieval (binary category masks, no ingredient table).
//
// With m in {0,1}^C a dish's mask is one of 2^C - 1 patterns P, and Model_Recommender.py:67-96 collapses to
//     score(u, d) = alpha_P[u] + < w_P[u], RE[d] >,
//     alpha_P[u] = (a / n_P) sum_{c in P} <U_high[u], CE_c>,   w_P[u] = ((1-a) / n_P) sum_{c in P} U_low,c[u]
// so after sorting the dishes by pattern the contraction runs over K = E instead of (C+1)*E: 5x fewer
// MFMAs, 5x fewer bytes through the LDS ring, and a user operand of E/2 registers instead of 5E/2.
// alpha_P rides in as the initial accumulator.  Dishes with an empty mask (0/0 -> NaN) are left out and
// appended by m2d_topk_fill_absent.  Masks with other weights use the dense kernel above.
// =====================================================================================================
constexpr int GRP_MAXPAT = 16;
constexpr int GRP_NB = 16;                      // row-norm buckets inside a pattern group (bucket 0 = largest norms)
constexpr int GRP_KEYS = GRP_MAXPAT * GRP_NB;   // sort key = pattern * GRP_NB + bucket
// layout of the small `grp` table behind the block histograms (int32 words):
//   [0..15] first slot of each pattern's group   [16] tiles  [17] slots  [32] flags   [40..55] rows per pattern
//   [64..64+GRP_KEYS) first slot of each (pattern, bucket) key        [GRP_STAT..+4) row-norm statistics (floats)
//   [GRP_RMAX..+16) largest row norm of each pattern (float bits; scan-start threshold, grouped_threshold_seed)
constexpr int GRP_KEYOFF = 64, GRP_STAT = 64 + GRP_KEYS, GRP_RMAX = GRP_STAT + 8, GRP_WORDS = GRP_RMAX + 16;

// float <-> int32 with the same order (an involution): thresholds of a user's dish ranges meet in one atomicMax word
__device__ __forceinline__ int32_t thr_key(const float f)
{
    const int32_t b = __float_as_int(f);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float thr_unkey(const int32_t k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// A threshold to start the scan from, known before any dish is scored.  With 0/1 masks score(u, d) = alpha_P[u] +
// <w_P[u], RE[d]> >= alpha_P[u] - |w_P[u]| max_{d in P} |RE[d]| (Cauchy-Schwarz), so a pattern that holds at least k dishes
// puts k scores at or above that bound, and the user's final k-th score cannot be below the largest such bound.  The
// model blends 0.99 high level + 0.01 low level (Model_Recommender.py:95-96), so alpha_P dominates: for most users the
// bound of their best pattern is above every score of the other fourteen, and the scan inserts half as often (110 -> 60
// insertions per lane at 100 k dishes, scripts/diag/topk_scan_sim.py).  A pattern whose UPPER bound is below the scan-start
// bound cannot reach the user's top-k and is not scanned at all.
//
// What the bounds have to hold for is the score AS A SCAN KERNEL COMPUTES IT (pruned and unpruned calls must return the
// same bits), so they are widened by what f32 / split-bf16 arithmetic can do -- measured against the sums of ABSOLUTE
// terms, not against results that may have cancelled (u = 2^-24; gamma_n = n u bounds any summation order of n terms):
//   * alpha_P = (a / n) sum_{c in P} <U_high, CE_c>: each dot product is off by at most gamma_(E+8) ha_c, ha_c = sum_e |U_high,e
//     CE_c,e| (plan and scan kernels may sum in different orders), the few operations after it by 7 u: |alpha~ - alpha| <=
//     (gamma_(E+8) + 7 u) A,  A = (a / n) sum_{c in P} ha_c;
//   * |w_P|^2 = (b / n)^2 |sum_{c in P} U_low,c|^2 from the f32 Gram matrix G: every G_cd is off by at most gamma_(E+8)
//     sum_e |U_c,e U_d,e| <= gamma |U_c| |U_d|, the ten-term sum by gamma_10 more, so the true value is at most
//     w2 + gam S^2,  S = sum_{c in P} sqrt(G_cc)  (when the rows cancel, w2 itself can come out 0 or negative while the
//     true |w_P| is sqrt(gam) S -- the term restores it);
//   * the scan's operand w~ = fl(beta fl(sum_c U_low,c)) is off by at most 5 u (b / n) sum_c |U_c,e| per element: at most
//     5 u W in a score,  W = (b / n) S max|r|  (again the absolute sum, |w_P| may have cancelled);
//   * the contraction: exact-f32 MFMA chains starting from alpha, gamma_(E+1) (|alpha| + |w~||r|); split bf16, 3 x 2^-18
//     |w~||r| for the dropped lo x lo products and the splits' own rounding, gamma_E |w~||r| for the f32 accumulation, u
//     |score| for the final alpha + acc; the f32 row norms behind max|r| are off by gamma_(E/64+7) / 2.
// With gam = 2 (E + 32) u all of it fits in  slack = 1e-4 reach + gam (A + W):  every score a scan kernel computes for a dish
// of pattern P lies in [alpha~ - reach - slack, alpha~ + reach + slack], reach = (b / n) sqrt(w2 + gam S^2) max|r| (1 + gam).
// On the benchmark's tables the slack is 5e-4 of the reach (A ~ 0.6, W ~ 0.03, reach ~ 0.014): the bounds prune what they
// pruned before.
struct PatternBound {
    float alpha, reach, slack;                              // lo = alpha - reach - slack, hi = alpha + reach + slack
};

__device__ __forceinline__ PatternBound grouped_pattern_terms(const float (&hc)[4], const float (&ha)[4], const float (&G)[10], const int32_t *grp,
                                                              const int pt, const int k, const float a, const float b, const int E, float &lo,
                                                              float &hi)
{
    const int rows = grp[40 + pt];                          // wave-uniform
    const float rmax = __int_as_float(grp[GRP_RMAX + pt]);
    const float inv_n = 1.0f / (float)__builtin_popcount(pt);
    const float gam = (float)(E + 32) * 1.1920929e-7f;      // 2 (E + 32) 2^-24
    float hs = 0.f, as = 0.f, w2 = 0.f, S = 0.f;
    int i = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hs += ((pt >> c) & 1) ? hc[c] : 0.f;
        as += ((pt >> c) & 1) ? ha[c] : 0.f;
        S += ((pt >> c) & 1) ? sqrtf(G[i]) : 0.f;            // G[i] here: the diagonal entry G_cc (a sum of squares)
#pragma unroll
        for (int d = c; d < 4; ++d, ++i) w2 += (((pt >> c) & 1) && ((pt >> d) & 1)) ? (c == d ? G[i] : 2.f * G[i]) : 0.f;
    }
    PatternBound pb;
    pb.alpha = a * (hs * inv_n);                            // the scan kernels' own expression, bit for bit
    const float A = fabsf(a) * inv_n * as, W = fabsf(b) * inv_n * S * rmax;
    pb.reach = (fabsf(b) * inv_n) * sqrtf(fmaxf(w2, 0.f) + gam * (S * S)) * rmax * (1.0f + gam);
    pb.slack = 1e-4f * pb.reach + gam * (A + W) + 1e-30f;
    lo = pb.alpha - pb.reach - pb.slack;                    // k dishes at or above this ...
    hi = pb.alpha + pb.reach + pb.slack;                    // ... no dish of the pattern above this
    if (rows < k) lo = -INFINITY;                            // (branches, not selects: a wave-uniform select between a vector value and
    if (rows <= 0) hi = -INFINITY;                           //  a constant sent hipcc 7.2 into "Illegal instruction detected")
    return pb;
}

__device__ __forceinline__ void grouped_pattern_bounds(const float (&hc)[4], const float (&ha)[4], const float (&G)[10], const int32_t *grp,
                                                       const int k, const float a, const float b, const int E, float &seed, uint32_t &mask)
{
    seed = -INFINITY;
#pragma unroll 1
    for (int pt = 1; pt < 16; ++pt) {
        float lo, hi;
        grouped_pattern_terms(hc, ha, G, grp, pt, k, a, b, E, lo, hi);
        seed = fmaxf(seed, lo);                              // a NaN bound is ignored
    }
    mask = 0u;
#pragma unroll 1
    for (int pt = 1; pt < 16; ++pt) {
        float lo, hi;
        grouped_pattern_terms(hc, ha, G, grp, pt, k, a, b, E, lo, hi);
        mask |= !(hi < seed) ? (1u << pt) : 0u;              // NaN bounds keep their pattern
    }
}

// the same from 16 lanes that all hold hc, ha and G: lane j works out pattern j (lo / hi: its own bounds), the bound is the
// largest lo of the 16; grouped_mask_lanes: the patterns whose hi reaches a bound
__device__ __forceinline__ PatternBound grouped_pattern_bounds_lanes(const float (&hc)[4], const float (&ha)[4], const float (&G)[10],
                                                                     const int32_t *grp, const int k, const float a, const float b, const int E,
                                                                     const int j, float &seed, float &lo, float &hi)
{
    lo = -INFINITY;
    hi = -INFINITY;
    PatternBound pb{0.f, 0.f, 0.f};
    if (j >= 1) pb = grouped_pattern_terms(hc, ha, G, grp, j, k, a, b, E, lo, hi);
    lo = fmaxf(lo, -INFINITY);                              // a NaN bound is ignored
    seed = lo;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) seed = fmaxf(seed, __shfl_xor(seed, off, 64));
    return pb;
}

__device__ __forceinline__ uint32_t grouped_mask_lanes(const float hi, const float seed, const int j)
{
    uint32_t mask = (j >= 1 && !(hi < seed)) ? (1u << j) : 0u;      // NaN bounds keep their pattern
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) mask |= __shfl_xor(mask, off, 64);
    return mask;
}

__device__ __forceinline__ float grouped_threshold_seed(const v4f *pmu, const int Sr, const float (&hc)[4], const GroupedArgs &p)
{
    float G[10], ha[4] = {0.f, 0.f, 0.f, 0.f};
    const v4f *ce4 = reinterpret_cast<const v4f *>(p.ce);
#pragma unroll
    for (int i = 0; i < 10; ++i) G[i] = 0.f;
#pragma unroll 1
    for (int q = 0; q < Sr; ++q) {
        v4f u[4];
        const v4f uh = pmu[q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u[c] = pmu[(c + 1) * Sr + q];
            const v4f w = ce4[c * Sr + q];
            ha[c] += (fabsf(uh.x * w.x) + fabsf(uh.y * w.y)) + (fabsf(uh.z * w.z) + fabsf(uh.w * w.w));
        }
        int i = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = c; d < 4; ++d, ++i) G[i] += (u[c].x * u[d].x + u[c].y * u[d].y) + (u[c].z * u[d].z + u[c].w * u[d].w);
    }
    float seed;
    uint32_t mask;
    grouped_pattern_bounds(hc, ha, G, p.grp, p.k, p.a, p.b, Sr * 4, seed, mask);
    return seed;
}

// End of a pattern-grouped scan: the lane's register list goes to LDS with its slots translated to dish ids, the two
// lanes of a user (l, l + 32) are merged into this split's sorted list of k, and the split's tie value is written: the
// list's last score if a tie decides what the list holds -- a lane's tie event at that very value (tie_mask), or a
// score the merge leaves behind that equals it -- else NaN.
template <int KR>
__device__ __forceinline__ void grouped_publish(float *ls, int32_t *li, const float (&rs)[KR], const int32_t (&ri)[KR],
                                                const GroupedArgs &p, const int lane, const int64_t uidx, const bool uvalid,
                                                const unsigned long long tie_mask, const int split,
                                                const LeftOut lo = M2D_LEFTOUT_NONE)
{
    const int j = lane & 31, h = lane >> 5, k = p.k;
    // this lane's left-out scores, and the other lane's of the same user (l + 32), for the lane that merges the two lists
    const int32_t lo_id1 = (p.ex_out && lo.i1 >= 0) ? p.perm[lo.i1] : -1, lo_id2 = (p.ex_out && lo.i2 >= 0) ? p.perm[lo.i2] : -1;
    const float lob_s1 = __shfl(lo.s1, j + 32, 64), lob_s2 = __shfl(lo.s2, j + 32, 64), lob_s3 = __shfl(lo.s3, j + 32, 64);
    const int32_t lob_id1 = __shfl(lo_id1, j + 32, 64), lob_id2 = __shfl(lo_id2, j + 32, 64);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        ls[i * 64 + lane] = rs[i];
        li[i * 64 + lane] = ri[i] >= 0 ? p.perm[ri[i]] : -1;
        cnt += ri[i] >= 0 ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int cnt_hi = __shfl(cnt, j + 32, 64);
    const bool tie_a = ((tie_mask >> j) & 1ull) != 0ull, tie_b = ((tie_mask >> (j + 32)) & 1ull) != 0ull;
    if (h == 0 && uvalid) {
        const int ca = cnt, cb = cnt_hi;
        int pa = 0, pb = 0;
        float *os = p.out_scores + ((size_t)uidx * p.nsplit + split) * k;
        int32_t *oi = p.out_ids + ((size_t)uidx * p.nsplit + split) * k;
        float last = 0.f;
        bool full = true;
        for (int o = 0; o < k; ++o) {
            const bool ha = pa < ca, hb = pb < cb;
            if (!ha && !hb) {
                os[o] = __builtin_nanf("");
                oi[o] = -1;
                full = false;
                continue;
            }
            const float sa = ha ? ls[pa * 64 + lane] : 0.f, sb = hb ? ls[pb * 64 + lane + 32] : 0.f;
            const int32_t ia = ha ? li[pa * 64 + lane] : 0, ib = hb ? li[pb * 64 + lane + 32] : 0;
            bool take_a;
            if (!hb) take_a = true;
            else if (!ha) take_a = false;
            else take_a = sa > sb || (sa == sb && ia < ib);
            os[o] = last = take_a ? sa : sb;
            oi[o] = take_a ? ia : ib;
            pa += take_a ? 1 : 0;
            pb += take_a ? 0 : 1;
        }
        const bool tie = full && ((pa < ca && ls[pa * 64 + lane] == last) || (pb < cb && ls[pb * 64 + lane + 32] == last) ||
                                  (tie_a && ls[(KR - 1) * 64 + lane] == last) || (tie_b && ls[(KR - 1) * 64 + lane + 32] == last));
        p.tie_val[(size_t)uidx * p.nsplit + split] = tie ? last : __builtin_nanf("");
        if (p.ex_out) {
            // left out of this range's list: what the two lanes left out, and what the merge left behind in their lists (two
            // entries of each suffice for the best two)
            LeftOut o = lo;
            o.i1 = lo_id1; o.i2 = lo_id2;
            left_out_merge(o, lob_s1, lob_id1);
            left_out_merge(o, lob_s2, lob_id2);
            left_out_merge(o, lob_s3, -1);
            for (int q = 0; q < 3; ++q) {                    // three entries of each list suffice for the best three
                if (pa + q < ca) left_out_merge(o, ls[(pa + q) * 64 + lane], li[(pa + q) * 64 + lane]);
                if (pb + q < cb) left_out_merge(o, ls[(pb + q) * 64 + lane + 32], li[(pb + q) * 64 + lane + 32]);
            }
            float *ex = p.ex_out + ((size_t)uidx * p.nsplit + split) * 8;
            ex[0] = o.s1; ex[1] = __int_as_float(o.i1); ex[2] = o.s2; ex[3] = __int_as_float(o.i2); ex[4] = o.s3;
        }
    }
}

#ifndef M2D_TOPK_HALF_BLOCKS
#define M2D_TOPK_HALF_BLOCKS 1                             // the launcher's own choice of 128-user blocks (see launch_grouped)
#endif
constexpr int grouped_tiles_per_stage(int E) { return E <= 32 ? 16 : (E == 64 ? 8 : (E == 128 ? 4 : 2)); }   // 64 KiB stages

}  // namespace
