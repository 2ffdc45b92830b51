// Full-catalogue retrieval for gfx950: top-k dishes per user over all I dishes (m2d_topk_users).
//
// Model_Recommender.py:67-96 is linear in the user block: score(u, d) = <flatten(PM[u]), Dt[d]> with
//   Dt[d] = concat( a*(sum_c m_c CE_c)/n , (1-a)*m_0/n*RE[d], ..., (1-a)*m_{C-1}/n*RE[d] ),  n = sum_c m_c
// (SURVEY.md section 7).  Scoring every (user, dish) is therefore a dense [users x K] . [K x dishes]
// contraction, K = (C+1)*E, and it is MFMA-bound, not HBM-bound.  The parity bar is 1e-4 in float32,
// so the product runs on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (a k-ordered fmaf
// chain), never through bf16.
//
// Kernel shape.  A wave owns 32 users for the whole kernel: their K-vectors live in registers as
// the MFMA B operand (K/2 VGPRs).  Dish tiles of 32 rows stream through LDS as the A operand:
// LDS-DMA (global_load_lds_dwordx4) fills one stage while the previous one is multiplied; the
// image is XOR-swizzled on the source side so the ds_read_b128 fragment reads do not bank-conflict.
// The 32x32 accumulator leaves each lane holding 16 dishes of ONE user (column = lane & 31), so the
// running top-k needs no cross-lane traffic: a lane filters its 16 scores against its current k-th
// best and (rarely) inserts into a private sorted list in LDS.  Lanes l and l+32 share a user; their
// two lists are merged once at the end.  Dishes reach a lane in increasing id, and insertion is
// stable, so ties go to the lower dish id as heapq.nlargest does (evaluate.py:63).
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
static unsigned long long *g_m2d_diag_buffer = nullptr;   // set by scripts/diag only
#if M2D_DIAG & 16
#define STAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#else
#define STAMP(x)
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ---- dish vectors -------------------------------------------------------------------------------
// One wave per dish; Dt rows are padded to a multiple of 32 dishes (pad rows are zero and are never
// ranked: their id is >= I).
__global__ __launch_bounds__(256) void m2d_build_dish_vectors(const float *re, const float *ce,
                                                              const float *dish_cats, const float *hv, int64_t I,
                                                              int C, int E, float a, float b, float *dt,
                                                              int64_t rows)
{
    const int lane = threadIdx.x & 63;
    const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= rows) return;
    const int K = (C + 1) * E;
    float *o = dt + d * K;
    if (d >= I) {
        for (int k = lane; k < K; k += 64) o[k] = 0.f;
        return;
    }
    float n = 0.f;
    for (int c = 0; c < C; ++c) n += dish_cats[d * C + c];                 // :77
    for (int e = lane; e < E; e += 64) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(dish_cats[d * C + c], ce[(size_t)c * E + e], s);   // :67 summed over c
        o[e] = hv ? a * hv[d * E + e] : a * (s / n);                       // :79, :95 (hv: ingredient extension)
        const float r = re[d * E + e];
        for (int c = 0; c < C; ++c) o[(size_t)(c + 1) * E + e] = b * ((dish_cats[d * C + c] / n) * r);   // :82-96
    }
}

struct TopkArgs {
    const float *pm;       // [U, K]
    const float *dt;       // [rows, K]
    const int32_t *users;  // [nU] global ids
    int64_t nU;
    int64_t U;
    int64_t I;
    int64_t user_base;
    int32_t k;
    int32_t nsplit;        // dish-range splits (gridDim.y)
    int64_t tiles;         // dish tiles of 32 in total
    float *out_scores;     // [nU, nsplit, k]
    int32_t *out_ids;
    int32_t *err;
    unsigned long long *dbg;   // scripts/diag only (M2D_DIAG & 16): per-wave phase cycle sums
};

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

// NB = K / 8: float4 registers of the user operand per lane.  One stage = 32 dishes x KC floats.
// KR > 0: the lane's running list (KR >= k slots) lives in REGISTERS and an insertion is a branch-free
// compare-exchange sweep (about 8*KR VALU ops, no LDS latency chain), so every wave reaches the
// per-stage barrier at nearly the same time; KR == 0 keeps the list in LDS (k up to 64).
template <int NB, int WAVES, int KR>
__global__ __launch_bounds__(WAVES * 64) void m2d_topk_mfma(TopkArgs p)
{
    constexpr int KP = NB * 8;                   // K
    constexpr int KC8 = NB < 40 ? NB : 40;       // macro steps (8 k-values each) per stage
    constexpr int NKC = NB / KC8;                // stages per dish tile
    static_assert(NB % KC8 == 0, "K must split evenly into stages");
    constexpr int S = KC8 * 2;                   // 16-B slots per LDS row
    constexpr int G = (S % 16 == 0) ? 16 : 8;    // swizzle group
    constexpr int PIECES = S / 2;                // 1-KiB DMA pieces per stage (32 rows * S slots / 64)
    constexpr int STAGE_FLOATS = 32 * S * 4;

    extern __shared__ __align__(16) float smem[];
    float *stage0 = smem;
    float *lists = smem + 2 * STAGE_FLOATS;      // per wave: k x 64 scores then k x 64 ids
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int k = p.k;
    const int kl = KR > 0 ? KR : k;             // list slots per lane in the LDS image
    float *ls = lists + (size_t)wave * 2 * kl * 64;
    int32_t *li = reinterpret_cast<int32_t *>(ls + (size_t)kl * 64);
    float rs[KR > 0 ? KR : 1];                  // register-resident list (KR > 0)
    int32_t ri[KR > 0 ? KR : 1];
#pragma unroll
    for (int i = 0; i < (KR > 0 ? KR : 1); ++i) {
        rs[i] = -INFINITY;                      // empty slot: any finite score beats it
        ri[i] = -1;
    }

    // ---- this wave's 32 users -> B operand registers -------------------------------------------
    const int64_t utile = (int64_t)blockIdx.x * WAVES + wave;
    const int64_t uidx = utile * 32 + j;
    const bool uvalid = uidx < p.nU;
    int64_t ul = 0;
    if (uvalid) {
        const int32_t uid = p.users[uidx];
        ul = (int64_t)uid - p.user_base;
        if (ul < 0 || ul >= p.U) {
            if (atomicCAS(&p.err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
                p.err[1] = uid;
                p.err[2] = (int32_t)(uidx & 0xffffffff);
                p.err[3] = (int32_t)(uidx >> 32);
            }
            ul = 0;
        }
    }
    v4f breg[NB];
    {
        const v4f *pu = reinterpret_cast<const v4f *>(p.pm) + (size_t)ul * (KP / 4) + h;
#pragma unroll
        for (int t = 0; t < NB; ++t) breg[t] = pu[2 * t];
    }

    // ---- dish range of this block ----------------------------------------------------------------
    const int64_t per = (p.tiles + p.nsplit - 1) / p.nsplit;
    const int64_t t_begin = (int64_t)blockIdx.y * per;
    const int64_t t_end = min(p.tiles, t_begin + per);
    const int64_t nstages = (t_end > t_begin ? (t_end - t_begin) : 0) * NKC;

    auto issue_stage = [&](int64_t s, int buf) {
        const int64_t tile = t_begin + s / NKC;
        const int c = (int)(s % NKC);
        float *dst = stage0 + (size_t)buf * STAGE_FLOATS;
        for (int pc = wave; pc < PIECES; pc += WAVES) {
            const int ps = pc * 64 + lane;             // physical 16-B slot in the stage image
            const int r = ps / S, sl = ps - r * S;
            const int q = sl ^ (r & (G - 1));          // logical slot that must land here
            const float *src = p.dt + ((size_t)(tile * 32 + r) * KP + (size_t)c * (KC8 * 8) + q * 4);
            lds_dma16(src, dst + pc * 256);
        }
    };

    int cnt = 0;
    float thr = -INFINITY;
    v16f acc;

    if (nstages > 0) issue_stage(0, 0);
    wait_all_vmem();
    __syncthreads();

#if M2D_DIAG & 16
    unsigned long long t_mfma = 0, t_epi = 0, t_bar = 0, t_slow = 0, n_slow = 0, t0_, t1_;
#endif
    for (int64_t s = 0; s < nstages; ++s) {
        const int buf = (int)(s & 1);
#if M2D_DIAG & 16
        STAMP(t0_);
#endif
        if (!(M2D_DIAG & 2) && s + 1 < nstages) issue_stage(s + 1, buf ^ 1);
        const int c = (int)(s % NKC);
        if (c == 0) {
            const int64_t tile0 = (t_begin + s / NKC) * 32;
            if (KR > 0 && tile0 + 32 > p.I) {   // last, partial tile: pad rows start at -inf and never rank
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[r] = (tile0 + 4 * h + (r & 3) + 8 * (r >> 2) < p.I) ? 0.f : -INFINITY;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
        }
        const float *img = stage0 + (size_t)buf * STAGE_FLOATS + (size_t)j * (S * 4);
#pragma unroll
        for (int cc = 0; cc < NKC; ++cc) {
            if (cc == c) {                                   // static register indices per stage kind
#pragma unroll
                for (int T = 0; T < KC8; ++T) {
                    const int q = (2 * T + h) ^ (j & (G - 1));
                    const v4f av = *reinterpret_cast<const v4f *>(img + q * 4);
                    const v4f bv = breg[cc * KC8 + T];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
                }
            }
        }
#if M2D_DIAG & 16
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
        STAMP(t1_); t_mfma += t1_ - t0_; t0_ = t1_;
        bool was_slow = false;
#endif
        if (M2D_DIAG & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
        } else if (c == NKC - 1) {
            // ---- epilogue: lane holds user j, dishes base + (r&3) + 8*(r>>2) + 4*h, ascending in r
            const int64_t base = (t_begin + s / NKC) * 32 + 4 * h;
            if constexpr (KR > 0) {
                // Exact-f32 MFMA runs on the same FMA lanes as the VALU, so every VALU instruction here is
                // matrix time lost: one compare per score, and only where some lane beats its threshold a
                // branch-free sweep (5 VALU per slot).  Non-finite scores never beat a threshold; they are
                // appended after the scan (m2d_topk_fill_absent).  Dishes arrive in ascending id and the
                // sweep is stable, so ties keep the lower id first.
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[r];
                    const bool cand = v > thr;
                    if ((M2D_DIAG & 8) ? false : __any(cand)) {
#if M2D_DIAG & 16
                        was_slow = true;
#endif
                        sorted_insert<KR>(rs, ri, v, (int32_t)base + (r & 3) + 8 * (r >> 2));
                        thr = rs[KR - 1];
                    }
                }
            } else {
                float mx = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                const bool maybe = (cnt < k) || !(mx <= thr);
                if ((M2D_DIAG & 8) ? (mx == 12345.678f) : __any(maybe)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[r];
                        const int64_t dish = base + (r & 3) + 8 * (r >> 2);
                        const bool cand = dish < p.I && ((cnt < k) || ahead(v, thr));
                        if (cand) {
                            int pos = cnt < k ? cnt : k - 1;
                            while (pos > 0) {
                                const float w = ls[(pos - 1) * 64 + lane];
                                if (!ahead(v, w)) break;
                                ls[pos * 64 + lane] = w;
                                li[pos * 64 + lane] = li[(pos - 1) * 64 + lane];
                                --pos;
                            }
                            ls[pos * 64 + lane] = v;
                            li[pos * 64 + lane] = (int32_t)dish;
                            if (cnt < k) ++cnt;
                            if (cnt == k) thr = ls[(k - 1) * 64 + lane];
                        }
                    }
                }
            }
        }
#if M2D_DIAG & 16
        STAMP(t1_); t_epi += t1_ - t0_; if (was_slow) { t_slow += t1_ - t0_; ++n_slow; } t0_ = t1_;
#endif
        if (!(M2D_DIAG & 4)) {
            wait_all_vmem();
            __syncthreads();
        }
#if M2D_DIAG & 16
        STAMP(t1_); t_bar += t1_ - t0_;
#endif
    }

#if M2D_DIAG & 16
    if (lane == 0 && p.dbg) {
        unsigned long long *d = p.dbg + ((size_t)blockIdx.x * WAVES + wave) * 8;
        d[0] = t_mfma; d[1] = t_epi; d[2] = t_bar; d[3] = t_slow; d[4] = n_slow; d[5] = (unsigned long long)nstages;
    }
#endif
    if constexpr (KR > 0) {   // publish the register lists so the partner lane can be merged in
        cnt = 0;
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            ls[i * 64 + lane] = rs[i];
            li[i * 64 + lane] = ri[i];
            cnt += ri[i] >= 0 ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the two lanes of each user, write this split's sorted list ------------------------
    const int cnt_hi = __shfl(cnt, j + 32, 64);
    if (h == 0 && uvalid) {
        const int ca = cnt, cb = cnt_hi;
        int pa = 0, pb = 0;
        float *os = p.out_scores + ((size_t)uidx * p.nsplit + blockIdx.y) * k;
        int32_t *oi = p.out_ids + ((size_t)uidx * p.nsplit + blockIdx.y) * k;
        for (int o = 0; o < k; ++o) {
            const bool ha = pa < ca, hb = pb < cb;
            if (!ha && !hb) {
                os[o] = __builtin_nanf("");
                oi[o] = -1;
                continue;
            }
            const float sa = ha ? ls[pa * 64 + lane] : 0.f, sb = hb ? ls[pb * 64 + lane + 32] : 0.f;
            const int32_t ia = ha ? li[pa * 64 + lane] : 0, ib = hb ? li[pb * 64 + lane + 32] : 0;
            bool take_a;
            if (!hb) take_a = true;
            else if (!ha) take_a = false;
            else take_a = ahead(sa, sb) || (!ahead(sb, sa) && ia < ib);
            os[o] = take_a ? sa : sb;
            oi[o] = take_a ? ia : ib;
            pa += take_a ? 1 : 0;
            pb += take_a ? 0 : 1;
        }
    }
}

// Generic shapes (K not a multiple of 8*20, C != 4 ...): one block per user, waves stride over dishes,
// direct dot with the dish vector.  Correct for any C, E; not tuned.
__global__ __launch_bounds__(256) void m2d_topk_generic(TopkArgs p, int K)
{
    extern __shared__ __align__(16) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = p.k;
    float *ls = smem + (size_t)wave * 2 * k;
    int32_t *li = reinterpret_cast<int32_t *>(ls + k);
    const int64_t uidx = blockIdx.x;
    const int32_t uid = p.users[uidx];
    int64_t ul = (int64_t)uid - p.user_base;
    if (ul < 0 || ul >= p.U) {
        if (threadIdx.x == 0 && atomicCAS(&p.err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
            p.err[1] = uid;
            p.err[2] = (int32_t)(uidx & 0xffffffff);
            p.err[3] = (int32_t)(uidx >> 32);
        }
        ul = 0;
    }
    const float *um = p.pm + (size_t)ul * K;
    int cnt = 0;
    for (int64_t d = wave; d < p.I; d += 4) {
        const float *dv = p.dt + (size_t)d * K;
        float s = 0.f;
        for (int e = lane; e < K; e += 64) s = fmaf(um[e], dv[e], s);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) {
            const bool cand = cnt < k || ahead(s, ls[k - 1]);
            if (cand) {
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && ahead(s, ls[pos - 1])) {
                    ls[pos] = ls[pos - 1];
                    li[pos] = li[pos - 1];
                    --pos;
                }
                ls[pos] = s;
                li[pos] = (int32_t)d;
                if (cnt < k) ++cnt;
            }
        }
    }
    __shared__ int s_cnt[4];
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ptr[4] = {0, 0, 0, 0};
        for (int o = 0; o < k; ++o) {
            int best = -1;
            for (int w = 0; w < 4; ++w) {
                if (ptr[w] >= s_cnt[w]) continue;
                if (best < 0) { best = w; continue; }
                const float sv = smem[(size_t)w * 2 * k + ptr[w]], bv = smem[(size_t)best * 2 * k + ptr[best]];
                const int32_t si = reinterpret_cast<int32_t *>(smem + (size_t)w * 2 * k + k)[ptr[w]];
                const int32_t bi = reinterpret_cast<int32_t *>(smem + (size_t)best * 2 * k + k)[ptr[best]];
                if (ahead(sv, bv) || (!ahead(bv, sv) && si < bi)) best = w;
            }
            if (best < 0) {
                p.out_scores[uidx * k + o] = __builtin_nanf("");
                p.out_ids[uidx * k + o] = -1;
            } else {
                p.out_scores[uidx * k + o] = smem[(size_t)best * 2 * k + ptr[best]];
                p.out_ids[uidx * k + o] = reinterpret_cast<int32_t *>(smem + (size_t)best * 2 * k + k)[ptr[best]];
                ++ptr[best];
            }
        }
    }
}

// nsplit sorted partial lists per user -> final top-k.  LPU lanes per user (LPU = nsplit rounded up to a power of two,
// <= 64), lane w holding the head of split w's list; each of the k rounds is an argmax over the group by xor
// shuffles and the winning lane steps to its next entry.  (The first form of this kernel walked all nsplit * k
// entries from ONE thread through a scratch-memory pointer array: ~0.2 ms for a single user with 64 splits, most
// of that call's latency.)  Splits cover increasing dish ranges, so on equal scores the lower split (= lower lane)
// wins, which keeps ties in ascending-id order.
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

template <int LPU>
__global__ __launch_bounds__(256) void m2d_topk_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k,
                                                             float *out_scores, int32_t *out_ids, const float *tie_in, float *tie_out,
                                                             int32_t *tie_list, int64_t I, const float *ex_in = nullptr, float *ex_out = nullptr,
                                                             const float *plan = nullptr, int32_t *rcount = nullptr)
{
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
        if (rcount && tie_list && u < nU && w == 0) rcount[8 + u] = refine_ent;      // a word per user, no atomics (m2d_topk_refine compacts)
    }
}

// Users with fewer than k ranked dishes (every finite-scored dish is already in their list, the rest of the
// catalogue scored NaN -- an empty category mask, Model_Recommender.py:79 -- or -inf): append the dishes not
// in the list in ascending id with a NaN score, which is where heapq.nlargest-style "NaN last" puts them.
__global__ void m2d_topk_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I)
{
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < nU) fill_absent_user(scores + u * k, ids + u * k, k, I);
}

// Users whose final k-th score is tied with a score that was left out (tie_list: a count and their positions in the
// call, gathered from the final tie values by m2d_topk_tie_compact): the whole catalogue again in dish-id order, in plain
// f32 -- Model_Recommender.py:67-96 with the sums over a 0/1 mask's categories taken first -- and a strict insertion, so
// equal scores keep the lower id exactly as heapq.nlargest does (evaluate.py:63).  On N(0, 1/E) tables
// about one user in ten thousand (two f32 scores at the edge of a list are bit-equal); every user of an all-zero
// Personal_Memory table.
//   m2d_topk_repair_scan   block (dish range sp of REPAIR_SPLITS, listed user f): 64 groups of 16 lanes take a dish each,
//                          a float4 column per lane (coalesced 256-B row reads), and keep a private top-k in LDS; the
//                          block's 64 lists are merged into the partial list of (f, sp);
//   m2d_topk_repair_merge  one wave per listed user: its REPAIR_SPLITS partial lists -> the final list.
// Users beyond the REPAIR_CAP the scratch holds (degenerate tables) are done by m2d_topk_repair_rest, one block each.
constexpr int REPAIR_SPLITS = 64, REPAIR_CAP = 1024;

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

// UB listed users per pass over a dish range: a dish row is read once and scored for all of them (one user per pass read the
// whole f32 table per listed user -- 230 MB for nine users of a 100 k-dish catalogue, 73 us; eight per pass: two passes).
template <int UB, bool HVR>
__global__ __launch_bounds__(1024) void m2d_topk_repair_scan(RepairArgs p)
{
    extern __shared__ __align__(16) float rsm[];
    constexpr int C = 4, NG = 64, NP = 1 << C, ND = 2;   // C = 4 (as the pattern-grouped kernels); 64 groups of 16 lanes; ND dishes in flight per group
    const int E = p.E, E4 = E >> 2, k = p.k, W = (C + 1) * E;
    float *um = rsm;                                        // [UB][(C+1) E] the users' blocks
    float *wp = um + UB * W;                                // [UB][NP][E]   sum of the pattern's low-level rows (0/1 masks: :82 summed over c)
    float *ls = wp + UB * NP * E;                           // [UB][NG groups][k] scores
    int32_t *li = reinterpret_cast<int32_t *>(ls + UB * NG * k);
    __shared__ float hc[UB][C], alpha[UB][NP];              // <U_high, CE_c>; sum over the pattern's categories (:67-75)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, grp = t >> 4;
    const int count = min(p.tie_list[0], p.cap);
    for (int f0 = blockIdx.y * UB; f0 < count; f0 += gridDim.y * UB) {   // block-uniform
        const int nu = min(UB, count - f0);
        __syncthreads();
        for (int ub = 0; ub < nu; ++ub) {
            int64_t ul = (int64_t)p.users[p.tie_list[1 + f0 + ub]] - p.user_base;
            if (ul < 0 || ul >= p.U) ul = 0;                // latched by the scan kernel
            for (int i = t; i < W; i += 1024) um[ub * W + i] = p.pm[(size_t)ul * W + i];
        }
        // a group's running top-16 of a user lives in its 16 lanes, slot j in lane j (k <= 16): an insertion is two compares
        // and two selects per lane against the lane's own slot and its left neighbour's (DPP row_shr:1), no LDS, no serial walk
        float slot_s[UB];
        int32_t slot_i[UB];
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) { slot_s[ub] = -INFINITY; slot_i[ub] = -1; }
        __syncthreads();
        // the masks are 0/1 (the pattern-grouped kernels run for nothing else): a dish's terms depend on its pattern P only
        //   high = sum_{c in P} <U_high, CE_c> / n_P        low = < RE[d], sum_{c in P} U_low,c > / n_P
        for (int i = t; i < nu * NP * E; i += 1024) {
            const int ub = i / (NP * E), r = i - ub * (NP * E), pt = r / E, e = r - pt * E;
            float w = 0.f;
            for (int c = 0; c < C; ++c) w += ((pt >> c) & 1) ? um[ub * W + (c + 1) * E + e] : 0.f;
            wp[i] = w;
        }
        const bool planned = p.plan != nullptr && !HVR;
        for (int x = wave; x < nu * C; x += 16) {           // a wave per (user, category): <U_high, CE_c>
            const int ub = x / C, c = x - ub * C;
            float q = 0.f;
            if (planned) q = p.plan[(size_t)p.tie_list[1 + f0 + ub] * 8 + 1 + c];            // the scan kernels' own value
            else {
                for (int e = lane; e < E; e += 64) q = fmaf(um[ub * W + e], p.ce[(size_t)c * E + e], q);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
            }
            if (lane == 0) hc[ub][c] = q;
        }
        __syncthreads();
        if (t < nu * NP) {
            const int ub = t / NP, pt = t - ub * NP;
            if (planned) alpha[ub][pt] = repair_alpha(p.a, hc[ub], pt);
            else {
                float x = 0.f;
                for (int c = 0; c < C; ++c) x += ((pt >> c) & 1) ? hc[ub][c] : 0.f;
                alpha[ub][pt] = x / (float)__builtin_popcount(pt);                           // :79 (pattern 0: 0 / 0 = NaN)
            }
        }
        __syncthreads();
        const v4f *um4 = reinterpret_cast<const v4f *>(um), *wp4 = reinterpret_cast<const v4f *>(wp);
        // The dishes come from the pattern-sorted f32 table (rows: exact copies of Recipe_Embedding's, [H[d] | RE[d]] with the
        // ingredient extension; perm: slot -> dish id), group by group, and only the groups of patterns that can reach the top-k of
        // one of this pass's users (the union of their plan masks -- the bounds hold for this kernel's f32 arithmetic as for the
        // scan kernels', grouped_pattern_terms): a listed user has 1.3 relevant patterns on average, so a pass of four reads a
        // third of the table.  The relevant rows are dealt out evenly to the REPAIR_SPLITS blocks.  Slots are not in id order
        // across norm buckets, so an insertion compares (score desc, id asc) explicitly.
        uint32_t rel = 0xfffeu;
        if (p.plan && !p.all_patterns) {
            rel = 0u;
            for (int ub = 0; ub < nu; ++ub) rel |= __float_as_uint(p.plan[(size_t)p.tie_list[1 + f0 + ub] * 8 + 5]);
        }
        int64_t R = 0;
        for (int q = 1; q < NP; ++q) R += ((rel >> q) & 1u) ? p.grp[40 + q] : 0;
        const int64_t per = (R + REPAIR_SPLITS - 1) / REPAIR_SPLITS;
        const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(R, i0 + per);
        const int EW4 = p.ew >> 2;
        const v4f *rows4 = reinterpret_cast<const v4f *>(p.rows);
        int64_t cum = 0;                                    // relevant rows in front of pattern q's group
        for (int q = 1; q < NP; ++q) {                      // block-uniform
            const int64_t rows_q = ((rel >> q) & 1u) ? p.grp[40 + q] : 0;
            const int64_t lo_i = i0 > cum ? i0 : cum, hi_i = i1 < cum + rows_q ? i1 : cum + rows_q;
            const int64_t d0 = p.grp[q] + (lo_i - cum), d1 = p.grp[q] + (hi_i - cum);     // this block's slots of the group
            cum += rows_q;
            if (lo_i >= hi_i) continue;
            const float npat = (float)__builtin_popcount(q);                                 // :77
            // the ids and the first 16 float4 columns of the NEXT step's dishes are fetched while this step's are scored (a step
            // is one round trip to memory otherwise)
            v4f it_n[ND], hv_n[ND];
            int32_t id_n[ND];
            auto fetch = [&](const int64_t db) __attribute__((always_inline)) {
#pragma unroll
                for (int x = 0; x < ND; ++x) {
                    const int64_t d = db + x * NG + grp;
                    const int64_t da = d < d1 ? d : d0;
                    id_n[x] = p.perm[da];
                    if (j < E4) {
                        it_n[x] = rows4[(size_t)da * EW4 + (HVR ? E4 : 0) + j];
                        if (HVR) hv_n[x] = rows4[(size_t)da * EW4 + j];
                    }
                }
            };
            fetch(d0);
            for (int64_t db = d0; db < d1; db += ND * NG) { // wave-uniform trip count: the shuffles see a full EXEC
                int64_t dd[ND];
                int32_t did[ND];
                bool ok[ND];
                float hs[ND][UB], lo[ND][UB];
                v4f it0[ND], hv0[ND];
#pragma unroll
                for (int x = 0; x < ND; ++x) {
                    const int64_t d = db + x * NG + grp;
                    ok[x] = d < d1;
                    dd[x] = ok[x] ? d : d0;
                    did[x] = id_n[x];
                    it0[x] = it_n[x];
                    hv0[x] = hv_n[x];
#pragma unroll
                    for (int ub = 0; ub < UB; ++ub) hs[x][ub] = lo[x][ub] = 0.f;
                }
                if (db + ND * NG < d1) fetch(db + ND * NG);
                for (int c4 = j; c4 < E4; c4 += 16) {
                    v4f it[ND], hvv[ND];
#pragma unroll
                    for (int x = 0; x < ND; ++x) {
                        if (c4 == j) {
                            it[x] = it0[x];
                            hvv[x] = hv0[x];
                        } else {
                            it[x] = rows4[(size_t)dd[x] * EW4 + (HVR ? E4 : 0) + c4];
                            if (HVR) hvv[x] = rows4[(size_t)dd[x] * EW4 + c4];
                        }
                    }
#pragma unroll
                    for (int ub = 0; ub < UB; ++ub) {
                        if (ub < nu) {                      // block-uniform
                            const v4f w = wp4[(ub * NP + q) * E4 + c4];
#pragma unroll
                            for (int x = 0; x < ND; ++x) {
                                lo[x][ub] = fmaf(it[x].x, w.x, fmaf(it[x].y, w.y, fmaf(it[x].z, w.z, fmaf(it[x].w, w.w, lo[x][ub]))));
                                if (HVR) {
                                    const v4f uh = um4[ub * (W >> 2) + c4];
                                    hs[x][ub] = fmaf(uh.x, hvv[x].x, fmaf(uh.y, hvv[x].y, fmaf(uh.z, hvv[x].z, fmaf(uh.w, hvv[x].w, hs[x][ub]))));
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int ub = 0; ub < UB; ++ub) {
                    if (ub < nu) {                          // block-uniform
                        // sum over the 16 lanes of a row: four rotations (DPP row_ror 8, 4, 2, 1; the same pairs as an xor butterfly,
                        // so every lane ends with the same bits) -- no LDS round trip (ds_bpermute) per step
#pragma unroll
                        for (int x = 0; x < ND; ++x) {
                            lo[x][ub] = row16_sum(lo[x][ub]);
                            if (HVR) hs[x][ub] = row16_sum(hs[x][ub]);
                        }
#pragma unroll
                        for (int x = 0; x < ND; ++x) {
                            float sc = planned ? repair_score_planned(alpha[ub][q], p.b, lo[x][ub] / npat)
                                               : m2d_blend_unfused(p.a, HVR ? hs[x][ub] : alpha[ub][q], p.b, lo[x][ub] / npat);   // :79 (done above), :92, :95-96
                            sc = ok[x] ? fmaxf(sc, -INFINITY) : -INFINITY;                   // NaN -> -inf: never enters
                            const int32_t id = ok[x] ? did[x] : 0x7fffffff;
                            // left neighbour's slot (lane j - 1 of the same 16-lane row; lane 0 sees +inf / -1)
                            const float left_s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, INFINITY),
                                                                    __builtin_bit_cast(int, slot_s[ub]), 0x111, 0xf, 0xf, false));
                            const int32_t left_i = __builtin_amdgcn_update_dpp(-1, slot_i[ub], 0x111, 0xf, 0xf, false);
                            // (score desc, id asc); an empty slot holds (-inf, -1): any real score is above it, -inf never is
                            const bool above_left = sc > left_s || (sc == left_s && sc > -INFINITY && id < left_i);
                            const bool above_me = sc > slot_s[ub] || (sc == slot_s[ub] && sc > -INFINITY && id < slot_i[ub]);
                            slot_i[ub] = above_left ? left_i : (above_me ? id : slot_i[ub]);
                            slot_s[ub] = above_left ? left_s : (above_me ? sc : slot_s[ub]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
            if (j < k) { ls[(ub * NG + grp) * k + j] = slot_s[ub]; li[(ub * NG + grp) * k + j] = slot_s[ub] > -INFINITY ? slot_i[ub] : -1; }
        __syncthreads();
        for (int ub = wave; ub < nu; ub += 16) {            // wave ub: the user's NG lists -> this block's partial list, (score desc, id asc)
            int ptr = 0;
            float *os = p.part_s + ((size_t)(f0 + ub) * REPAIR_SPLITS + blockIdx.x) * k;
            int32_t *oi = p.part_i + ((size_t)(f0 + ub) * REPAIR_SPLITS + blockIdx.x) * k;
            for (int o = 0; o < k; ++o) {
                float bs = ptr < k ? ls[(ub * NG + lane) * k + ptr] : 0.f;
                int32_t bi = ptr < k ? li[(ub * NG + lane) * k + ptr] : -1;
                int bl = lane;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float xs = __shfl_xor(bs, off, 64);
                    const int32_t xi = __shfl_xor(bi, off, 64);
                    const int xl = __shfl_xor(bl, off, 64);
                    if (repair_ahead(xs, xi, bs, bi)) { bs = xs; bi = xi; bl = xl; }
                }
                if (lane == 0) { os[o] = bs; oi[o] = bi; }
                if (bi >= 0 && bl == lane) ++ptr;
            }
        }
    }
}

// the partial lists of listed user f < cap (one wave per user) -> the user's final list
__device__ __forceinline__ void repair_merge_phase(const RepairArgs &p)
{
    const int lane = threadIdx.x & 63, k = p.k;
    const int count = min(p.tie_list[0], p.cap);
    for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < count; f += gridDim.x * 4) {     // wave-uniform
        const int64_t u = p.tie_list[1 + f];
        const bool live = lane < REPAIR_SPLITS;
        const float *s = p.part_s + ((size_t)f * REPAIR_SPLITS + (live ? lane : 0)) * k;
        const int32_t *id = p.part_i + ((size_t)f * REPAIR_SPLITS + (live ? lane : 0)) * k;
        int ptr = 0;
        for (int o = 0; o < k; ++o) {
            float bs = (live && ptr < k) ? s[ptr] : 0.f;
            int32_t bi = (live && ptr < k) ? id[ptr] : -1;
            int bl = lane;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float xs = __shfl_xor(bs, off, 64);
                const int32_t xi = __shfl_xor(bi, off, 64);
                const int xl = __shfl_xor(bl, off, 64);
                if (repair_ahead(xs, xi, bs, bi)) { bs = xs; bi = xi; bl = xl; }
            }
            if (lane == 0) {
                p.out_scores[u * k + o] = bi >= 0 ? bs : __builtin_nanf("");
                p.out_ids[u * k + o] = bi;
            }
            if (live && bi >= 0 && bl == lane) ++ptr;
        }
        if (lane == 0) fill_absent_user(p.out_scores + u * k, p.out_ids + u * k, k, p.I);     // (its own stores, in program order)
    }
}

// listed users p.cap, p.cap + 1, ...: one block each over the whole catalogue (slow; degenerate tables only).  The same
// arithmetic as m2d_topk_repair_scan, step for step -- pattern sums, a float4 column per lane, the 16-lane rotation sum --
// so a user's re-ranked scores do not depend on which of the two kernels its place in the list sent it to (the list's order
// is the order of the compaction's atomics).
// The kernel also does the first tier's last step (m2d_topk_repair_merge's work, before its own loop): one launch less per call.
template <bool HVR>
__global__ __launch_bounds__(256) void m2d_topk_repair_finish(RepairArgs p)
{
    extern __shared__ __align__(16) float rsm[];
    constexpr int C = 4, NG = 16, NP = 1 << C;              // 16 groups of 16 lanes, a dish each
    repair_merge_phase(p);
    const int E = p.E, E4 = E >> 2, k = p.k, W = (C + 1) * E;
    float *um = rsm;                                        // [(C+1) E] this user's block
    float *wp = um + W;                                     // [NP][E]
    float *ls = wp + NP * E;                                // [NG][k]
    int32_t *li = reinterpret_cast<int32_t *>(ls + NG * k);
    __shared__ float hc[C], alpha[NP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 15, grp = t >> 4;
    const int count = p.tie_list[0];
    for (int f = p.cap + blockIdx.x; f < count; f += gridDim.x) {     // block-uniform
        const int64_t u = p.tie_list[1 + f];
        int64_t ul = (int64_t)p.users[u] - p.user_base;
        if (ul < 0 || ul >= p.U) ul = 0;                    // latched by the scan kernel
        __syncthreads();
        for (int i = t; i < W; i += 256) um[i] = p.pm[(size_t)ul * W + i];
        float slot_s = -INFINITY;
        int32_t slot_i = -1;
        __syncthreads();
        for (int i = t; i < NP * E; i += 256) {
            const int pt = i / E, e = i - pt * E;
            float w = 0.f;
            for (int c = 0; c < C; ++c) w += ((pt >> c) & 1) ? um[(c + 1) * E + e] : 0.f;
            wp[i] = w;
        }
        const bool planned = p.plan != nullptr && !HVR;
        {                                                   // wave c: <U_high, CE_c>
            const int c = wave;
            float q = 0.f;
            if (planned) q = p.plan[(size_t)u * 8 + 1 + c];
            else {
                for (int e = lane; e < E; e += 64) q = fmaf(um[e], p.ce[(size_t)c * E + e], q);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
            }
            if (lane == 0) hc[c] = q;
        }
        __syncthreads();
        if (t < NP) {
            if (planned) alpha[t] = repair_alpha(p.a, hc, t);
            else {
                float x = 0.f;
                for (int c = 0; c < C; ++c) x += ((t >> c) & 1) ? hc[c] : 0.f;
                alpha[t] = x / (float)__builtin_popcount(t);                                 // :79 (pattern 0: 0 / 0 = NaN)
            }
        }
        __syncthreads();
        const v4f *um4 = reinterpret_cast<const v4f *>(um), *wp4 = reinterpret_cast<const v4f *>(wp);
        for (int64_t db = 0; db < p.I; db += NG) {          // wave-uniform trip count
            const int64_t d = db + grp;
            const bool ok = d < p.I;
            const int64_t dd = ok ? d : 0;
            const v4f m = *reinterpret_cast<const v4f *>(p.cats + (size_t)dd * C);
            const int pt = (m.x != 0.f ? 1 : 0) | (m.y != 0.f ? 2 : 0) | (m.z != 0.f ? 4 : 0) | (m.w != 0.f ? 8 : 0);
            float hs = 0.f, lo = 0.f;
            for (int q = j; q < E4; q += 16) {
                const v4f it = reinterpret_cast<const v4f *>(p.re)[(size_t)dd * E4 + q];
                const v4f w = wp4[pt * E4 + q];
                lo = fmaf(it.x, w.x, fmaf(it.y, w.y, fmaf(it.z, w.z, fmaf(it.w, w.w, lo))));
                if (HVR) {
                    const v4f hvv = reinterpret_cast<const v4f *>(p.hv)[(size_t)dd * E4 + q];
                    const v4f uh = um4[q];
                    hs = fmaf(uh.x, hvv.x, fmaf(uh.y, hvv.y, fmaf(uh.z, hvv.z, fmaf(uh.w, hvv.w, hs))));
                }
            }
            lo = row16_sum(lo);
            if (HVR) hs = row16_sum(hs);
            const float n = (float)__builtin_popcount(pt);                                    // :77
            float sc = planned ? repair_score_planned(alpha[pt], p.b, lo / n)
                               : m2d_blend_unfused(p.a, HVR ? hs : alpha[pt], p.b, lo / n);   // :79, :92, :95-96
            sc = ok ? fmaxf(sc, -INFINITY) : -INFINITY;                                       // NaN -> -inf: never enters
            const float left_s = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, INFINITY),
                                                    __builtin_bit_cast(int, slot_s), 0x111, 0xf, 0xf, false));
            const int32_t left_i = __builtin_amdgcn_update_dpp(-1, slot_i, 0x111, 0xf, 0xf, false);
            const bool above_left = sc > left_s, above_me = sc > slot_s;                      // strict: ascending ids keep the lower id first
            slot_i = above_left ? left_i : (above_me ? (int32_t)dd : slot_i);
            slot_s = above_left ? left_s : (above_me ? sc : slot_s);
        }
        if (j < k) { ls[grp * k + j] = slot_s; li[grp * k + j] = slot_s > -INFINITY ? slot_i : -1; }
        __syncthreads();
        if (wave == 0) {                                    // the NG lists -> the user's final list, (score desc, id asc)
            const bool live = lane < NG;
            int ptr = 0;
            for (int o = 0; o < k; ++o) {
                float bs = (live && ptr < k) ? ls[lane * k + ptr] : 0.f;
                int32_t bi = (live && ptr < k) ? li[lane * k + ptr] : -1;
                int bl = lane;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float xs = __shfl_xor(bs, off, 64);
                    const int32_t xi = __shfl_xor(bi, off, 64);
                    const int xl = __shfl_xor(bl, off, 64);
                    if (repair_ahead(xs, xi, bs, bi)) { bs = xs; bi = xi; bl = xl; }
                }
                if (lane == 0) {
                    p.out_scores[u * k + o] = bi >= 0 ? bs : __builtin_nanf("");
                    p.out_ids[u * k + o] = bi;
                }
                if (live && bi >= 0 && bl == lane) ++ptr;
            }
            if (lane == 0) fill_absent_user(p.out_scores + u * k, p.out_ids + u * k, k, p.I);
        }
    }
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
template <int CH>                                           // float4 columns per emulated lane: E <= 64 CH
__global__ __launch_bounds__(256) void m2d_topk_refine(RefineArgs p)
{
    constexpr int C = 4;
    const int lane = threadIdx.x & 63, half = lane >> 5, i = lane & 31;
    const int k = p.k, E = p.E, E4 = E >> 2, W = (C + 1) * E;
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
        const bool mine = fvalid && i < nc;
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

}  // namespace

int m2d_ensure_dish_vectors(m2d_engine *h, hipStream_t st)
{
    if (h->dish_vec_valid) return M2D_OK;
    const int64_t rows = (h->I + 31) / 32 * 32;
    const size_t K = (size_t)(h->C + 1) * h->E;
    if (h->dish_vec_rows != rows || !h->dish_vec) {
        if (h->dish_vec) M2D_HIP_TRY(h, hipFree(h->dish_vec));
        h->dish_vec = nullptr;
        M2D_HIP_TRY(h, hipMalloc((void **)&h->dish_vec, (size_t)rows * K * sizeof(float)));
        h->dish_vec_rows = rows;
    }
    hipLaunchKernelGGL(m2d_build_dish_vectors, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, h->re, h->ce,
                       h->dish_cats, h->dish_high, h->I, h->C, h->E, h->a, h->b, h->dish_vec, rows);
    M2D_HIP_TRY(h, hipGetLastError());
    h->dish_vec_valid = true;
    return M2D_OK;
}

namespace {

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

// Inside a pattern group the dishes are scanned in descending order of their row norm, coarsely: 16 buckets of a
// quarter standard deviation between mean + 2 sigma and mean - 2 sigma, dish id order inside a bucket.  A dish's score
// is |w| |r| cos(theta), so the large scores sit among the large-norm rows: met first, they raise the running k-th best
// early and the small-norm rest of the group rarely beats it -- on N(0, 1/E) rows at 100 k dishes the fraction of
// tiles in which some lane of a wave has a candidate falls from 39 % to 20 % (simulated; every candidate tile costs
// an insertion round and staggers the waves at the next barrier).  Duplicate rows share a norm, hence a bucket, and
// keep their id order.
__global__ __launch_bounds__(256) void m2d_grp_norm_stats(const float *re, int64_t I, int E, float *norm, double *acc)
{
    // a wave takes 64 consecutive rows of the block's 256, one row at a time (coalesced); one pair of atomics per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double ssum[2][4];
    double s1 = 0.0, s2 = 0.0;
    const int64_t d0 = (int64_t)blockIdx.x * 256 + wave * 64;
    for (int r = 0; r < 64 && d0 + r < I; ++r) {
        const int64_t d = d0 + r;
        float q = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float x = re[d * E + e];
            q = fmaf(x, x, q);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
        const float nr = sqrtf(q);
        if (lane == 0) norm[d] = nr;
        if (nr == nr && nr < INFINITY) { s1 += nr; s2 += (double)nr * nr; }
    }
    if (lane == 0) { ssum[0][wave] = s1; ssum[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, ssum[0][0] + ssum[0][1] + ssum[0][2] + ssum[0][3]);
        atomicAdd(acc + 1, ssum[1][0] + ssum[1][1] + ssum[1][2] + ssum[1][3]);
    }
}

__global__ void m2d_grp_norm_params(const double *acc, int64_t I, float *stat)
{
    const double mean = acc[0] / (double)I;
    double var = acc[1] / (double)I - mean * mean;
    if (!(var > 0.0)) var = 0.0;
    const double sd = sqrt(var);
    stat[0] = (float)(mean + 2.0 * sd);                                  // upper edge of bucket 0
    stat[1] = sd > 1e-30 * (mean > 1.0 ? mean : 1.0) ? (float)(GRP_NB / (4.0 * sd)) : 0.f;   // buckets per unit of norm
}

__device__ __forceinline__ int grp_bucket(float nr, const float *stat)
{
    const float t = (stat[0] - nr) * stat[1];
    return t >= (float)(GRP_NB - 1) ? GRP_NB - 1 : (t > 0.f ? (int)t : 0);   // NaN norms land in bucket 0
}

__global__ __launch_bounds__(256) void m2d_grp_hist(const float *cats, const float *norm, const float *stat, int64_t I, int C,
                                                    int32_t *blk_hist, int32_t *flags, int32_t *rmax_bits)
{
    __shared__ int sh[GRP_KEYS];
    __shared__ int srmax[GRP_MAXPAT];
    sh[threadIdx.x] = 0;
    if (threadIdx.x < GRP_MAXPAT) srmax[threadIdx.x] = 0;
    __syncthreads();
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d < I) {
        int pat = 0;
        for (int c = 0; c < C; ++c) {
            const float m = cats[d * C + c];
            if (m != 0.f) {
                pat |= 1 << c;
                if (m != 1.f) atomicOr(flags, 1);      // not a 0/1 mask: the grouped form does not apply
            }
        }
        atomicAdd(&sh[pat * GRP_NB + grp_bucket(norm[d], stat)], 1);
        // the pattern's largest row norm (non-negative floats order like their bit patterns; a NaN norm counts as +inf)
        const float nr = norm[d];
        atomicMax(&srmax[pat], __float_as_int(nr == nr ? nr : INFINITY));
    }
    __syncthreads();
    blk_hist[(size_t)blockIdx.x * GRP_KEYS + threadIdx.x] = sh[threadIdx.x];
    if (threadIdx.x < GRP_MAXPAT && srmax[threadIdx.x] != 0) atomicMax(&rmax_bits[threadIdx.x], srmax[threadIdx.x]);   // one per pattern and block
}

// one block of 4 GRP_KEYS threads: per-key exclusive scan over the blocks (in place; four threads share a key, each
// owning a contiguous quarter of the blocks), padded group offsets, key offsets inside the groups (no padding between
// buckets), and the tile table  info = pattern | (valid rows << 8)
constexpr int GRP_SCAN_SPLIT = 4;
__global__ __launch_bounds__(GRP_KEYS * GRP_SCAN_SPLIT) void m2d_grp_scan(int32_t *blk_hist, int nblk, int32_t *grp,
                                                                           int32_t *tile_info)
{
    __shared__ int part[GRP_SCAN_SPLIT][GRP_KEYS];
    __shared__ int total[GRP_KEYS];
    __shared__ int tile0[GRP_MAXPAT], prow[GRP_MAXPAT];
    const int key = threadIdx.x % GRP_KEYS, qt = threadIdx.x / GRP_KEYS;
    const int per = (nblk + GRP_SCAN_SPLIT - 1) / GRP_SCAN_SPLIT;
    const int b0 = min(nblk, qt * per), b1 = min(nblk, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += blk_hist[(size_t)b * GRP_KEYS + key];
    part[qt][key] = sum;
    __syncthreads();
    int run = 0;
    for (int j = 0; j < qt; ++j) run += part[j][key];
    if (qt == GRP_SCAN_SPLIT - 1) total[key] = run + sum;
    for (int b = b0; b < b1; b += 8) {          // loads of a batch before its stores: the in-place update keeps them in order
        int c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = b + j < b1 ? blk_hist[(size_t)(b + j) * GRP_KEYS + key] : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (b + j < b1) blk_hist[(size_t)(b + j) * GRP_KEYS + key] = run;
            run += c[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int off = 0, t = 0;
        grp[0] = -1;
        grp[40] = 0;
        tile0[0] = 0;
        prow[0] = 0;
        for (int b = 0; b < GRP_NB; ++b) grp[GRP_KEYOFF + b] = 0;                 // pattern 0 (empty mask) is not ranked
        for (int q = 1; q < GRP_MAXPAT; ++q) {
            grp[q] = off;
            int rows = 0;
            for (int b = 0; b < GRP_NB; ++b) {
                grp[GRP_KEYOFF + q * GRP_NB + b] = off + rows;
                rows += total[q * GRP_NB + b];
            }
            grp[40 + q] = rows;                                                  // rows per pattern (pipelined kernel)
            const int nt = (rows + 31) / 32;
            tile0[q] = t;
            prow[q] = rows;
            t += nt;
            off += nt * 32;
        }
        grp[16] = t;
        grp[17] = off;
    }
    __syncthreads();
    for (int q = 1; q < GRP_MAXPAT; ++q) {
        const int rows = prow[q], nt = (rows + 31) / 32;
        for (int i = threadIdx.x; i < nt; i += GRP_KEYS * GRP_SCAN_SPLIT) tile_info[tile0[q] + i] = q | (min(32, rows - 32 * i) << 8);
    }
}

__global__ __launch_bounds__(256) void m2d_grp_scatter(const float *cats, const float *norm, const float *stat, int64_t I, int C,
                                                       const int32_t *blk_base, const int32_t *grp, int32_t *perm)
{
    __shared__ unsigned short sp[256];
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int key = 0xffff, pat = 0;
    if (d < I) {
        for (int c = 0; c < C; ++c) pat |= (cats[d * C + c] != 0.f) ? (1 << c) : 0;
        key = pat * GRP_NB + grp_bucket(norm[d], stat);
    }
    sp[threadIdx.x] = (unsigned short)key;
    __syncthreads();
    if (d < I && pat != 0) {
        int rank = 0;
        for (int q = 0; q < (int)threadIdx.x; ++q) rank += sp[q] == key ? 1 : 0;   // stable: ascending dish id
        perm[grp[GRP_KEYOFF + key] + blk_base[(size_t)blockIdx.x * GRP_KEYS + key] + rank] = (int32_t)d;
    }
}

// one wave per slot: RS[slot] = RE[perm[slot]] (zeros for padding), plus the split-bf16 image used by
// m2d_topk_grouped_bf16: per 32-row tile [hi: 32 x EW bf16][lo: 32 x EW bf16], x ~= hi + lo to 2^-17 |x|.
// With the ingredient extension (hv = H[d]) a slot's row is [H[d] | RE[d]], EW = 2 E: the high-level
// term <a U_high, H[d]> then rides in the same contraction as the low-level one (see GroupedArgs::hv).
__global__ __launch_bounds__(256) void m2d_grp_gather(const float *re, const float *hv, const int32_t *perm, int64_t slots,
                                                      int E, int EW, float *rs, __bf16 *rs16)
{
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= slots) return;
    const int32_t d = perm[slot];                 // EW = 2 E with hv, else E or E zero-padded to the kernel's width
    __bf16 *hi = rs16 + ((slot >> 5) * 64 + (slot & 31)) * (size_t)EW;
    __bf16 *lo = hi + 32 * (size_t)EW;
    for (int e = threadIdx.x & 63; e < EW; e += 64) {
        float x = 0.f;
        if (d >= 0) x = hv ? (e < E ? hv[(size_t)d * E + e] : re[(size_t)d * E + e - E]) : (e < E ? re[(size_t)d * E + e] : 0.f);
        rs[slot * EW + e] = x;
        const __bf16 xh = (__bf16)x;
        hi[e] = xh;
        lo[e] = (__bf16)(x - (float)xh);
    }
}

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

constexpr int PLAN_PROBES = 64;                            // most rows of the user's best pattern scored for the scan-start bound

// The plan of a retrieval call, one record of 8 floats per user: [0] scan-start bound, [1..4] <U_high, CE_c>, [5] the
// relevant-pattern mask (bits), for the pipelined kernel -- which takes its users in the order m2d_plan_* sort them into
// (by mask), so that the 256 users of a block share their relevant patterns and the block steps through those
// patterns' tiles only.  16 lanes per user, a float4 column each.
template <int CH>                                           // float4 columns a lane holds of a row: E <= 64 CH
__global__ __launch_bounds__(256) void m2d_topk_user_plan(const float *pm, const float *ce, const int32_t *users, int64_t nU, int64_t U,
                                                          int64_t user_base, int E, const int32_t *grp, int k, float a, float b,
                                                          int no_alpha, float *plan, int32_t *zero_tie, unsigned long long *zero_tiles,
                                                          int32_t *zero_hist, int nhist, const float *probe_rows, int probe_width, int nprobe,
                                                          int chain, int32_t *zero_refine)
{
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int E4 = E >> 2;
    // the call's counters start at zero (the kernels that count run after this one): the tie list's length, the tiles-scanned
    // diagnostic, the sort's histogram -- three memset launches less per call
    {
        const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (gid == 0) { *zero_tie = 0; *zero_tiles = 0ull; }
        if (gid < 4 && zero_refine) zero_refine[gid] = 0;   // m2d_topk_refine's counters
        if (zero_hist)
            for (int64_t i = gid; i < nhist; i += (int64_t)gridDim.x * 256) zero_hist[i] = 0;
    }
    int64_t ul = 0;
    if (u < nU) {
        ul = (int64_t)users[u] - user_base;
        if (ul < 0 || ul >= U) ul = 0;                      // reported by the scan kernel
    }
    const v4f *pmu = reinterpret_cast<const v4f *>(pm) + (size_t)ul * (5 * E4);
    const v4f *ce4 = reinterpret_cast<const v4f *>(ce);
    float hc[4] = {0.f, 0.f, 0.f, 0.f}, ha[4] = {0.f, 0.f, 0.f, 0.f}, G[10];     // ha: the same sums over |terms| (the bounds' rounding margin)
#pragma unroll
    for (int i = 0; i < 10; ++i) G[i] = 0.f;
    v4f r[4] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};     // (CH = 1: kept for the probes' operand)
    for (int q = j; q < E4; q += 16) {
        const v4f uh = pmu[q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f w = ce4[c * E4 + q];
            hc[c] += fmaf(uh.x, w.x, uh.y * w.y) + fmaf(uh.z, w.z, uh.w * w.w);
            ha[c] += (fabsf(uh.x * w.x) + fabsf(uh.y * w.y)) + (fabsf(uh.z * w.z) + fabsf(uh.w * w.w));
            r[c] = pmu[(c + 1) * E4 + q];
        }
        int i = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = c; d < 4; ++d, ++i) G[i] += fmaf(r[c].x, r[d].x, r[c].y * r[d].y) + fmaf(r[c].z, r[d].z, r[c].w * r[d].w);
    }
    // sums over the user's 16 lanes: row rotations by 8, 4, 2, 1 (DPP: one VALU each) -- the same bits in every lane as the xor
    // butterfly gave (each step adds the same two partial sums, and a + b = b + a), without its 72 ds_bpermute
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hc[c] = row16_sum(hc[c]);
        ha[c] = row16_sum(ha[c]);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) G[i] = row16_sum(G[i]);
    float seed, lo, hi;
    const PatternBound pb = grouped_pattern_bounds_lanes(hc, ha, G, grp, k, a, b, E, j, seed, lo, hi);
    // A better bound from a few dishes: the pattern whose lower bound IS the bound (the user's best) holds its largest-norm
    // rows first in the sorted table; the k-th largest exact score among the first 16 to 64 of them (more for larger catalogues) is a lower bound of
    // the pattern's k-th score -- about alpha_P + 0.1 |w||r| where Cauchy-Schwarz gives alpha_P - |w||r| -- and fewer other
    // patterns reach it (relevant patterns per user 1.9 -> 1.3 on the benchmark's tables; scripts/diag/two_phase_sim.py).
    if (probe_rows && no_alpha != 1) {
        const int g16 = lane & 48;                          // first lane of this user's 16
        const unsigned long long best = __ballot(j >= 1 && lo == seed && seed > -INFINITY);
        const int p1 = (int)((best >> g16) & 0xffffull) ? __builtin_ctz((unsigned)((best >> g16) & 0xffffull)) : 0;
        const int nrow = p1 ? (grp[40 + p1] < nprobe ? grp[40 + p1] : nprobe) : 0;      // >= k: the bound was finite
        const float alpha1 = __shfl(pb.alpha, g16 + p1, 64);                                  // alpha_P as the scan kernels compute it
        const float slack1 = __shfl(pb.slack, g16 + p1, 64);                                  // what arithmetic can move a score of P by
        // w_P1: this lane's float4 columns q = j, j + 16, ... (embedding sizes up to 256)
        v4f wv[CH];
        const float beta = b / (float)__builtin_popcount(p1 | (p1 == 0));
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            wv[i] = v4f{0.f, 0.f, 0.f, 0.f};
            const int q = j + 16 * i;
            if (CH == 1) {
                // the lane's float4 of the four rows is still in registers (the sums above read it): as conditional loads
                // they were four exec-masked blocks, each waiting for its own round trip
#pragma unroll
                for (int c = 0; c < 4; ++c) wv[0] += ((p1 >> c) & 1) ? r[c] : v4f{0.f, 0.f, 0.f, 0.f};
                wv[0] *= beta;
            } else if (q < E4) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if ((p1 >> c) & 1) wv[i] += pmu[(c + 1) * E4 + q];
                wv[i] *= beta;
            }
        }
        float slot = -INFINITY;                             // the probes' running top-16, entry j in lane j (k <= 16)
        const v4f *rows = reinterpret_cast<const v4f *>(probe_rows) + (size_t)grp[p1] * (probe_width >> 2);
        for (int i0 = 0; i0 < nprobe; i0 += 8) {            // uniform trip count (nprobe: a multiple of 8): the DPP rows see a full EXEC
            float part[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {                   // eight rows in flight
                const bool live = i0 + x < nrow;
                part[x] = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < CH; ++c4) {
                    const int q = j + 16 * c4;
                    if (live && q < E4) {
                        const v4f r = rows[(size_t)(i0 + x) * (probe_width >> 2) + q];
                        part[x] = fmaf(r.x, wv[c4].x, fmaf(r.y, wv[c4].y, fmaf(r.z, wv[c4].z, fmaf(r.w, wv[c4].w, part[x]))));
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                float sc = alpha1 + row16_sum(part[x]);
                sc = i0 + x < nrow ? fmaxf(sc, -INFINITY) : -INFINITY;
                const float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, INFINITY),
                                                      __builtin_bit_cast(int, slot), 0x111, 0xf, 0xf, false));
                slot = sc > left ? left : (sc > slot ? sc : slot);
            }
        }
        const float kth = __shfl(slot, g16 + k - 1, 64);
        // the k probe rows that scored kth or more here score kth - 2 slack or more in the scan: slack bounds the distance of a
        // computed score from the exact one for the scan kernels' arithmetic, and for this loop's (the same operand, f32 fma chains)
        const float probed = kth - 2.f * slack1;
        if (p1 && nrow >= k && probed > seed) seed = probed;
    }
    // How far a score of this user as a scan kernel computes it (split bf16: 3 x 2^-18 per product + the f32 accumulation; exact
    // f32: an MFMA chain from alpha) can lie from the same score in the tie repair's plain-f32 arithmetic -- the ranking the
    // lists are finished in (m2d_topk_refine): 2e-5 reach + gam (|alpha| + reach), the largest over ALL patterns with dishes,
    // so that it does not depend on the option form (seed and mask do).
    // (the repair takes alpha_P from this plan's words, bit for bit as the scan kernels form it: the high-level dot products'
    // own rounding is common to both and drops out.  Split bf16: 3 x 2^-18 |w||r| for the products and the splits; both: gam
    // |w||r| for the f32 accumulation orders; 4 u |score| for the final sums; an exact-f32 MFMA chain starts from alpha and
    // rounds its running sum E / 2 times at the score's magnitude -- `chain`.)
    const float smag = fabsf(pb.alpha) + pb.reach;
    // per unit of |w||r|: 1.2e-5 for the split (3 x 2^-18 = 1.144e-5), (E + 8) u for the split-bf16 kernel's f32 accumulation of E
    // products in whatever order the matrix pipe takes, (E / 16 + 12) u for the repair's own chain + rotation tree
    float delta = (j >= 1 && grp[40 + j] > 0)
                      ? (1.2e-5f + (float)(E + E / 16 + 20) * 5.9604645e-8f) * pb.reach + (4.f + (chain ? 0.5f * (float)E + 4.f : 0.f)) * 5.9604645e-8f * smag
                      : 0.f;
    // max over the user's 16 lanes (row rotations, as above)
    delta = fmaxf(delta, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, delta), 0x128, 0xf, 0xf, false)));
    delta = fmaxf(delta, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, delta), 0x124, 0xf, 0xf, false)));
    delta = fmaxf(delta, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, delta), 0x122, 0xf, 0xf, false)));
    delta = fmaxf(delta, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, delta), 0x121, 0xf, 0xf, false)));
    // a pattern is left out only if its dishes stay 2 delta under the bound: a score that close to the k-th must reach an
    // insertion in every launch shape (the refinement's candidates may not depend on the option form)
    uint32_t mask = grouped_mask_lanes(hi, seed - 2.f * delta, j);
    if (no_alpha == 1) { seed = -INFINITY; mask = 0xfffeu; }     // ingredient rows: the score has no alpha_P term to bound it with
    if (no_alpha == 2) mask = 0xfffeu;                        // option topk_prune = 2: the bound, but every pattern (A/B)
    if (no_alpha == 4) seed = -INFINITY;                      // option topk_prune = 4: the patterns, but no bound (A/B)
    if (u < nU && j == 0) {
        float *o = plan + (size_t)u * 8;
        o[0] = seed; o[1] = hc[0]; o[2] = hc[1]; o[3] = hc[2]; o[4] = hc[3]; o[5] = __uint_as_float(mask);
        o[6] = __int_as_float(thr_key(seed));               // the dish ranges' shared running threshold starts at the bound
        o[7] = delta;
    }
}

// counting sort of the call's users by their 15-bit pattern mask: histogram, scan (one block), scatter.  The order inside
// a mask does not matter -- a user's list does not depend on the block it is scored in.  Two thirds of the users share
// fifteen masks (one relevant pattern): one global atomic per user queued 65 536 of them on a few dozen addresses (190 us a
// pass; one per wave and distinct mask, found with ballots: 50 us).  Now a workgroup counts its users in a 128-KiB LDS table
// of all 32 768 keys and adds only the table's non-zero entries to the global counts; the scatter reserves a range per
// non-zero entry the same way and places its users inside the ranges with LDS atomics.
constexpr int PLAN_KEYS = 1 << 15;                         // a mask holds bits 1..15: key = mask >> 1
__device__ __forceinline__ int plan_key(const float *plan, const int64_t u)
{
    return (int)((__float_as_uint(plan[(size_t)u * 8 + 5]) >> 1) & (PLAN_KEYS - 1));
}

__global__ __launch_bounds__(1024) void m2d_plan_hist(const float *plan, int64_t nU, int32_t *hist)
{
    extern __shared__ __align__(16) int32_t plan_tab[];
    for (int i = threadIdx.x; i < PLAN_KEYS; i += 1024) plan_tab[i] = 0;
    __syncthreads();
    for (int64_t u = (int64_t)blockIdx.x * 1024 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 1024) atomicAdd(&plan_tab[plan_key(plan, u)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < PLAN_KEYS; i += 1024) {
        const int32_t c = plan_tab[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

__global__ __launch_bounds__(1024) void m2d_plan_scan(int32_t *hist)
{
    constexpr int PER4 = PLAN_KEYS / 1024 / 4;
    __shared__ int32_t wtot[16];
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i v[PER4];
    int32_t sum = 0;
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
        v[i] = reinterpret_cast<const v4i *>(hist)[threadIdx.x * PER4 + i];
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t t = __shfl_up(incl, off, 64);
        incl += lane >= off ? t : 0;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int32_t run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int i = 0; i < PER4; ++i) {
        v4i o;
        o.x = run; run += v[i].x;
        o.y = run; run += v[i].y;
        o.z = run; run += v[i].z;
        o.w = run; run += v[i].w;
        reinterpret_cast<v4i *>(hist)[threadIdx.x * PER4 + i] = o;
    }
}

__global__ __launch_bounds__(1024) void m2d_plan_scatter(const float *plan, int64_t nU, int32_t *cursor, int32_t *order)
{
    extern __shared__ __align__(16) int32_t plan_tab[];
    for (int i = threadIdx.x; i < PLAN_KEYS; i += 1024) plan_tab[i] = 0;
    __syncthreads();
    for (int64_t u = (int64_t)blockIdx.x * 1024 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 1024) atomicAdd(&plan_tab[plan_key(plan, u)], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < PLAN_KEYS; i += 1024) {       // a range of the key's positions for this workgroup's users
        const int32_t c = plan_tab[i];
        if (c) plan_tab[i] = atomicAdd(&cursor[i], c);
    }
    __syncthreads();
    for (int64_t u = (int64_t)blockIdx.x * 1024 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 1024)
        order[atomicAdd(&plan_tab[plan_key(plan, u)], 1)] = (int32_t)u;
}

// Launch order of a pruned scan.  Its workgroups -- (block of 256 sorted users, dish range) items -- are unequal: the
// tiles a block steps through are those of its users' patterns inside the dish range, anything from none to all of it.
// Handed out in grid order the long items that happen to come late leave most CUs idle at the end (a list-scheduling
// simulation of the benchmark's call, scripts/diag/pattern_prune_sim.py: 1.53 x the even share; longest first: 1.03 x).
// m2d_plan_items_work: one wave per user block ORs its users' masks and counts, per dish range, the tiles of those
// patterns.  m2d_plan_items_sort: one workgroup sorts the items by that count, descending (a counting sort over 1 024
// bins of the range's length; equal bins in any order -- the order changes when a list is computed, not what it holds).
__global__ __launch_bounds__(256) void m2d_plan_items_work(const float *plan, const int32_t *order, int64_t nU, const int32_t *grp,
                                                           int64_t tiles, int nsplit, int32_t *work, int upb)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // wave-uniform
    if (b * upb >= nU) return;                                          // upb: users per block of the scan kernel
    uint32_t m = 0u;
    for (int i = lane; i < upb; i += 64) {
        const int64_t pos = b * upb + i;
        if (pos < nU) m |= __float_as_uint(plan[(size_t)(order ? (int64_t)order[pos] : pos) * 8 + 5]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m |= __shfl_xor(m, off, 64);
    const int64_t per = (tiles + nsplit - 1) / nsplit;
    for (int s = lane; s < nsplit; s += 64) {
        const int64_t t0 = (int64_t)s * per, t1 = min(tiles, t0 + per);
        int64_t w = 0;
        for (int q = 1; q < GRP_MAXPAT; ++q) {
            const int rows = grp[40 + q];
            if (rows == 0 || !((m >> q) & 1u)) continue;
            const int64_t g0 = grp[q] >> 5, g1 = g0 + ((rows + 31) >> 5);
            const int64_t lo = g0 > t0 ? g0 : t0, hi = g1 < t1 ? g1 : t1;
            w += hi > lo ? hi - lo : 0;
        }
        work[b * nsplit + s] = (int32_t)w;
    }
}

__global__ __launch_bounds__(1024) void m2d_plan_items_sort(const int32_t *work, int64_t nitems, int64_t tiles, int nsplit, int32_t *items)
{
    constexpr int BINS = 1024;
    __shared__ int32_t cnt[BINS], base[BINS];
    const int64_t per = (tiles + nsplit - 1) / nsplit;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    auto bin_of = [&](const int32_t w) {
        const int64_t b = per > 0 ? (int64_t)w * (BINS - 1) / per : 0;
        return BINS - 1 - (int)(b > BINS - 1 ? BINS - 1 : b);            // bin 0 = the longest items
    };
    for (int64_t i = threadIdx.x; i < nitems; i += 1024) atomicAdd(&cnt[bin_of(work[i])], 1);
    __syncthreads();
    // exclusive scan of the 1 024 counts: inside each wave by shuffles, the 16 wave totals by the first wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int v = cnt[threadIdx.x], incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        incl += lane >= off ? t : 0;
    }
    __shared__ int32_t wtot[16];
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wtot[w];
    base[threadIdx.x] = woff + incl - v;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < nitems; i += 1024) items[atomicAdd(&base[bin_of(work[i])], 1)] = (int32_t)i;
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

// PAD: the tables' embedding size is p.e_real <= E (a multiple of 4): the sorted dish rows are zero-padded to E floats
// when the table is built, the user operand is zero beyond e_real -- the extra products are exact zeros, so the scores
// are those of an unpadded contraction.  This is what takes the reference's own size (embed_size 200,
// Train_recommender.py) off the one-block-per-user kernel.
template <int E8, int WAVES, int KR, bool PAD = false>
__global__ __launch_bounds__(WAVES * 64) void m2d_topk_grouped(GroupedArgs p)
{
    constexpr int E = E8 * 8, C = 4;
    constexpr int S = E / 4;                               // 16-B slots per row
    constexpr int TPS = grouped_tiles_per_stage(E);        // tiles per stage
    constexpr int STAGE_FLOATS = TPS * 32 * E;
    constexpr int PIECES = TPS * 32 * S / 64;              // 1-KiB DMA pieces per stage
    constexpr int SW = S < 16 ? S : 16;                    // XOR-swizzle modulus (bank row = 16 slots)

    extern __shared__ __align__(16) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int k = p.k;

    // (user block, dish range) of this workgroup, the user's place in the call, <U_high, CE_c>, the scan-start bound and the
    // relevant patterns: all from the call's plan, as in m2d_topk_grouped_bf16_pipe2 (m2d_topk_user_plan, m2d_plan_*)
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (p.items) {
        const int it = __builtin_amdgcn_readfirstlane(p.items[blockIdx.x]);
        bx = it / p.nsplit;
        by = it - bx * p.nsplit;
    }
    const int64_t pos = ((int64_t)bx * WAVES + wave) * 32 + j;
    const bool uvalid = pos < p.nU;
    // 32-bit on purpose (a call holds < 2^31 users, a shard < 2^31 rows): with 64-bit per-lane values live across the scan the
    // E = 256 instantiation spills one, and hipcc 7.2 reloads it into an odd register pair ("Subtarget requires even aligned
    // vector registers")
    const int uidx = uvalid ? (p.order ? p.order[pos] : (int)pos) : 0;
    int ul = 0;
    if (uvalid) {
        const int32_t uid = p.users[uidx];
        const int64_t ul64 = (int64_t)uid - p.user_base;
        ul = (int)ul64;
        if (ul64 < 0 || ul64 >= p.U) {
            if (atomicCAS(&p.err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
                p.err[1] = uid;
                p.err[2] = uidx;
                p.err[3] = 0;
            }
            ul = 0;
        }
    }
    const int Sr = PAD ? p.e_real / 4 : S;                 // 16-B slots per row of the tables
    float hc[C];                                           // <U_high, CE_c>   Model_Recommender.py:67-75
    const float *rec = p.plan + (size_t)uidx * 8;
#pragma unroll
    for (int c = 0; c < C; ++c) hc[c] = rec[1 + c];
    const float seed = uvalid ? rec[0] : INFINITY;         // a lane without a user never has a candidate
    constexpr bool EXT = !PAD && E8 <= 16;                 // what the lists leave out is kept for m2d_topk_refine (p.ex_out; E = 32 / 64 / 128)
    const float dlt2 = (EXT && p.ex_out && uvalid) ? 2.f * rec[7] : 0.f;     // scores this close under a threshold still reach the insertion
    LeftOut lout = M2D_LEFTOUT_NONE;
    uint32_t umask_lane = uvalid ? __float_as_uint(rec[5]) : 0u;
    __shared__ uint32_t s_umask;
    if (threadIdx.x == 0) s_umask = 0u;
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) umask_lane |= __shfl_xor(umask_lane, off, 64);
    if (lane == 0) atomicOr(&s_umask, umask_lane);
    __syncthreads();
    const uint32_t umask = __builtin_amdgcn_readfirstlane(s_umask);   // the block's patterns: the union over its users
    v4f wP[E8];
    float alpha = 0.f;
    int cur_pat = -1;

    float rs[KR];
    int32_t ri[KR];
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        rs[i] = -INFINITY;
        ri[i] = -1;
    }
    float thr = seed;

    const int64_t per = (p.tiles + p.nsplit - 1) / p.nsplit;
    const int64_t t_begin = (int64_t)by * per;
    const int64_t t_end = min(p.tiles, t_begin + per);
    // the stages to step through: those that hold a tile of a pattern in `umask`, as up to 15 ranges of stage numbers
    // (relative to t_begin) in lanes -- range i in lane i of r_first / r_cnt; inside a stage the tiles of other patterns
    // are passed over
    int r_first = 0, r_cnt = 0, nranges = 0, vstages = 0;
    {
        int last_end = -1;
        for (int q = 1; q < GRP_MAXPAT; ++q) {
            const int rows = p.grp[40 + q];
            if (rows == 0 || !((umask >> q) & 1u)) continue;
            const int64_t gt0 = p.grp[q] >> 5, gt1 = gt0 + ((rows + 31) >> 5);
            const int64_t lo = gt0 > t_begin ? gt0 : t_begin, hi = gt1 < t_end ? gt1 : t_end;
            if (lo >= hi) continue;
            int s0 = (int)((lo - t_begin) / TPS);
            const int s1 = (int)((hi - 1 - t_begin) / TPS);
            if (s0 <= last_end) s0 = last_end + 1;
            if (s0 > s1) continue;
            r_first = lane == nranges ? s0 : r_first;
            r_cnt = lane == nranges ? s1 - s0 + 1 : r_cnt;
            ++nranges;
            vstages += s1 - s0 + 1;
            last_end = s1;
        }
    }
    int w_idx = -1, w_stage = 0, w_left = 0;
    auto next_stage = [&]() __attribute__((always_inline)) {
        if (w_left == 0) {
            ++w_idx;
            w_stage = __builtin_amdgcn_readlane(r_first, w_idx < nranges ? w_idx : 0);
            w_left = __builtin_amdgcn_readlane(r_cnt, w_idx < nranges ? w_idx : 0);
        }
        --w_left;
        return w_stage++;
    };

    auto issue_stage = [&](int64_t s, int buf) {
        const float *src0 = p.rs + (size_t)(t_begin + s * TPS) * 32 * E;   // rows past the last tile are zero padding
        float *dst = smem + (size_t)buf * STAGE_FLOATS;
        for (int pc = wave; pc < PIECES; pc += WAVES) {
            const int ps = pc * 64 + lane;
            const int r = ps / S, sl = ps - r * S;
            const int q = sl ^ (r & (SW - 1));
            lds_dma16_b(src0 + (size_t)r * E + q * 4, dst + pc * 256);
        }
    };

    int st_cur = vstages > 0 ? next_stage() : 0, st_next = vstages > 1 ? next_stage() : 0;
    if (vstages > 0) issue_stage(st_cur, 0);
    wait_all_vmem();
    __syncthreads();

    v16f acc;
    unsigned long long tie_mask = 0ull;                    // lanes with a tie event at their list's present last value (tie_update)
    int tiles_done = 0;
    for (int v = 0; v < vstages; ++v) {
        const int buf = v & 1;
        if (v + 1 < vstages) issue_stage(st_next, buf ^ 1);
        for (int tl = 0; tl < TPS; ++tl) {
            const int64_t t = t_begin + (int64_t)st_cur * TPS + tl;
            if (t >= t_end) break;                                      // wave-uniform
            const int info = __builtin_amdgcn_readfirstlane(p.tile_info[t]);
            const int pat = info & 255, nvalid = info >> 8;
            if (!((umask >> pat) & 1u)) continue;                       // a neighbouring pattern's tile in a straddling stage
            ++tiles_done;
            if (pat != cur_pat) {                                       // at most 2^C - 1 times per block
                cur_pat = pat;
                const float inv_n = 1.0f / (float)__builtin_popcount(pat);
                float hs = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) hs += ((pat >> c) & 1) ? hc[c] : 0.f;
                alpha = p.a * (hs * inv_n);
                const float beta = p.b * inv_n;
#pragma unroll
                for (int T = 0; T < E8; ++T) wP[T] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int c = 0; c < C; ++c) {           // rolled: E8 loads in flight, not C * E8
                    if (!((pat >> c) & 1)) continue;
                    const v4f *row = reinterpret_cast<const v4f *>(p.pm) + (size_t)ul * ((C + 1) * Sr) + (c + 1) * Sr + h;
#pragma unroll
                    for (int T = 0; T < E8; ++T) {
                        if (!PAD || 2 * T + h < Sr) wP[T] += row[2 * T];
                    }
                }
#pragma unroll
                for (int T = 0; T < E8; ++T) wP[T] *= beta;
            }
            // alpha_P[u] is the initial accumulator; padding rows of a group's last tile start at -inf
            if (nvalid == 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = alpha;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (4 * h + (r & 3) + 8 * (r >> 2) < nvalid) ? alpha : -INFINITY;
            }
            const float *img = smem + (size_t)buf * STAGE_FLOATS + (size_t)(tl * 32 + j) * E;
#pragma unroll
            for (int T = 0; T < E8; ++T) {
                if (PAD && 8 * T >= p.e_real) continue;    // wave-uniform: the rest of the row is padding
                const int q = (2 * T + h) ^ (j & (SW - 1));
                const v4f av = *reinterpret_cast<const v4f *>(img + q * 4);
                const v4f bv = wP[T];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
            // epilogue: lane holds user j, slots t*32 + 4h + (r&3) + 8(r>>2), ascending in r (see m2d_topk_mfma)
            const int32_t sbase = (int32_t)(t * 32) + 4 * h;
            // one max tree + one branch settles the tiles in which no lane beats its threshold
            float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
            mx = fmaxf(mx, acc[15]);
            if (!__any(mx >= thr - dlt2)) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r];
                const bool cand = v >= thr - dlt2;
                if (__any(cand)) {
                    const float old_last = rs[KR - 1];
                    const int32_t old_id = ri[KR - 1];
                    sorted_insert_inplace<KR>(rs, ri, v, sbase + (r & 3) + 8 * (r >> 2));     // (the out-of-place form under this `if`: 18 v_mov per insertion to copy the new list over the old)
                    tie_mask = tie_update(tie_mask, v, old_last, rs[KR - 1]);
                    if (EXT) left_out_note(lout, fmaxf(v, -INFINITY), sbase + (r & 3) + 8 * (r >> 2), old_last, old_id, rs[KR - 1] - dlt2);
                    thr = fmaxf(rs[KR - 1], seed);
                }
            }
        }
        wait_all_vmem();
        __syncthreads();
        st_cur = st_next;
        if (v + 2 < vstages) st_next = next_stage();
    }
    if (p.tiles_scanned && threadIdx.x == 0) atomicAdd(p.tiles_scanned, (unsigned long long)tiles_done);

    // ---- publish (slot -> dish id), merge the two lanes of each user -------------------------------------
    float *ls = smem + (size_t)wave * 2 * KR * 64;       // aliases stage 0: every wave is past the last barrier
    int32_t *li = reinterpret_cast<int32_t *>(ls + (size_t)KR * 64);
    grouped_publish<KR>(ls, li, rs, ri, p, lane, uidx, uvalid, tie_mask, by, lout);
    (void)k;
}

// ---- split-bf16 ("bf16x3") variant of m2d_topk_grouped ---------------------------------------------------
// Exact-f32 MFMA runs at 1/16 of the bf16 matrix rate.  Here every operand is split x = hi + lo into two bf16
// and the product is a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
// (lo*lo, <= 2^-18 |a||b|, is dropped; bf16 x bf16 products are exact in fp32).  Per-product relative
// error <= ~1.2e-5, so the score error is ~1e-5 of sqrt(sum (a_k b_k)^2) -- inside the 1e-4 parity bar, and
// checked against the float64 restatement by the same tests as the exact kernel.  3 MFMAs of 16 k-values
// in 96 cycles replace 8 f32 MFMAs in 512.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int E, int WAVES, int KR>
__global__ __launch_bounds__(WAVES * 64) void m2d_topk_grouped_bf16(GroupedArgs p)
{
    constexpr int C = 4;
    constexpr int KS = E / 16;                             // k-steps (16 k-values) per tile
    constexpr int S8 = E / 8;                              // 16-B slots per bf16 row
    constexpr int RPB = 256 / (E * 2) > 0 ? 256 / (E * 2) : 1;   // rows per 256-B bank row
    constexpr int TPS = E == 64 ? 8 : 4;                   // tiles per stage: 64 KiB stages
    constexpr int ROW_BYTES = E * 2, TILE_BYTES = 64 * ROW_BYTES, STAGE_BYTES = TPS * TILE_BYTES;
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int S4 = E / 4;                              // float4 per f32 row of Personal_Memory

    extern __shared__ __align__(16) unsigned char smem8[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int k = p.k;

    const int64_t uidx = ((int64_t)blockIdx.x * WAVES + wave) * 32 + j;
    const bool uvalid = uidx < p.nU;
    int64_t ul = 0;
    if (uvalid) {
        const int32_t uid = p.users[uidx];
        ul = (int64_t)uid - p.user_base;
        if (ul < 0 || ul >= p.U) {
            if (atomicCAS(&p.err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
                p.err[1] = uid;
                p.err[2] = (int32_t)(uidx & 0xffffffff);
                p.err[3] = (int32_t)(uidx >> 32);
            }
            ul = 0;
        }
    }
    const v4f *pmu = reinterpret_cast<const v4f *>(p.pm) + (size_t)ul * ((C + 1) * S4);
    float hc[C];
    {
        const v4f *ce4 = reinterpret_cast<const v4f *>(p.ce);
#pragma unroll
        for (int c = 0; c < C; ++c) hc[c] = 0.f;
#pragma unroll 1
        for (int q = 0; q < S4; ++q) {
            const v4f u = pmu[q];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const v4f w = ce4[c * S4 + q];
                hc[c] += (u.x * w.x + u.y * w.y) + (u.z * w.z + u.w * w.w);
            }
        }
    }
    bf16x8 wh[KS], wl[KS];                                 // w_P[u] for k = 16 s + 8 h + (0..7), split hi / lo
    float alpha = 0.f;
    int cur_pat = -1;

    float rs[KR];
    int32_t ri[KR];
#pragma unroll
    for (int i = 0; i < KR; ++i) {
        rs[i] = -INFINITY;
        ri[i] = -1;
    }
    const float seed = grouped_threshold_seed(pmu, S4, hc, p);
    float thr = seed;

    const int64_t per = (p.tiles + p.nsplit - 1) / p.nsplit;
    const int64_t t_begin = (int64_t)blockIdx.y * per;
    const int64_t t_end = min(p.tiles, t_begin + per);
    const int64_t nstages = t_end > t_begin ? (t_end - t_begin + TPS - 1) / TPS : 0;

    auto issue_stage = [&](int64_t s, int buf) {
        const unsigned char *src0 = reinterpret_cast<const unsigned char *>(p.rs16) + (size_t)(t_begin + s * TPS) * TILE_BYTES;
        unsigned char *dst = smem8 + (size_t)buf * STAGE_BYTES;
        for (int pc = wave; pc < PIECES; pc += WAVES) {
            const int g = pc * 64 + lane;                  // physical 16-B slot in the stage image
            const int rw = g / S8, sl = g - rw * S8;       // stage row (hi and lo rows alike), slot in row
            const int q = sl ^ ((rw / RPB) & (S8 - 1));    // logical slot that must land here
            lds_dma16(src0 + (size_t)rw * ROW_BYTES + q * 16, dst + pc * 1024);
        }
    };

    if (nstages > 0) issue_stage(0, 0);
    wait_all_vmem();
    __syncthreads();

    const int key = (j / RPB) & (S8 - 1);                  // this lane's row swizzle (same for hi and lo rows)
    v16f acc;
    unsigned long long tie_mask = 0ull;                    // lanes with a tie event at their list's present last value (tie_update)
#if M2D_DIAG & 16
    unsigned long long t_mfma = 0, t_epi = 0, t_bar = 0, t_slow = 0, n_slow = 0, n_tile = 0, t0_, t1_;
    STAMP(t0_);
#endif
    for (int64_t s = 0; s < nstages; ++s) {
        const int buf = (int)(s & 1);
        if (s + 1 < nstages) issue_stage(s + 1, buf ^ 1);
        for (int tl = 0; tl < TPS; ++tl) {
            const int64_t t = t_begin + s * TPS + tl;
            if (t >= t_end) break;                                      // wave-uniform
            const int info = __builtin_amdgcn_readfirstlane(p.tile_info[t]);
            const int pat = info & 255, nvalid = info >> 8;
            if (pat != cur_pat) {                                       // at most 2^C - 1 times per block
                cur_pat = pat;
                const float inv_n = 1.0f / (float)__builtin_popcount(pat);
                float hs = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) hs += ((pat >> c) & 1) ? hc[c] : 0.f;
                alpha = p.a * (hs * inv_n);
                const float beta = p.b * inv_n;
#pragma unroll 1
                for (int ks = 0; ks < KS; ++ks) {                       // rolled over k-steps: 2 x C loads in flight
                    v4f w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        if ((pat >> c) & 1) {
                            const v4f *row = pmu + (c + 1) * S4 + 4 * ks + 2 * h;
                            w0 += row[0];
                            w1 += row[1];
                        }
                    }
                    w0 *= beta;
                    w1 *= beta;
                    const float x[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    bf16x8 vh, vl;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const __bf16 xh = (__bf16)x[i];
                        vh[i] = xh;
                        vl[i] = (__bf16)(x[i] - (float)xh);
                    }
                    // static register indices: select by compare (KS <= 8)
#pragma unroll
                    for (int q = 0; q < KS; ++q) {
                        if (q == ks) {
                            wh[q] = vh;
                            wl[q] = vl;
                        }
                    }
                }
            }
            if (nvalid == 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = alpha;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (4 * h + (r & 3) + 8 * (r >> 2) < nvalid) ? alpha : -INFINITY;
            }
            const unsigned char *img = smem8 + (size_t)buf * STAGE_BYTES + (size_t)tl * TILE_BYTES + (size_t)j * ROW_BYTES;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int q = (2 * ks + h) ^ key;
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(img + q * 16);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(img + 32 * ROW_BYTES + q * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh[ks], acc, 0, 0, 0);
            }
            const int32_t sbase = (int32_t)(t * 32) + 4 * h;
#if M2D_DIAG & 16
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[r]));
            STAMP(t1_); t_mfma += t1_ - t0_; t0_ = t1_; ++n_tile;
#endif
            // 16 independent compares against the threshold as it stands -> 16 lane masks in SGPRs.  Their OR
            // settles most tiles with scalar work; a set mask says which scores to insert (sorted_insert is a
            // no-op for lanes whose score no longer beats a threshold raised earlier in this tile).
            unsigned long long m[16], any_mask = 0ull;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                m[r] = __ballot(acc[r] >= thr);
                any_mask |= m[r];
            }
            if (any_mask == 0ull) {
#if M2D_DIAG & 16
                STAMP(t1_); t_epi += t1_ - t0_; t0_ = t1_;
#endif
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (m[r] != 0ull) {
                    const float old_last = rs[KR - 1];
                    sorted_insert<KR>(rs, ri, acc[r], sbase + (r & 3) + 8 * (r >> 2));
                    tie_mask = tie_update(tie_mask, acc[r], old_last, rs[KR - 1]);
                }
            }
            thr = fmaxf(rs[KR - 1], seed);
#if M2D_DIAG & 16
            STAMP(t1_); t_slow += t1_ - t0_; ++n_slow; t0_ = t1_;
#endif
        }
        wait_all_vmem();
        __syncthreads();
#if M2D_DIAG & 16
        STAMP(t1_); t_bar += t1_ - t0_; t0_ = t1_;
#endif
    }
#if M2D_DIAG & 16
    if (lane == 0 && p.dbg) {
        unsigned long long *d = p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 8;
        d[0] = t_mfma; d[1] = t_epi; d[2] = t_bar; d[3] = t_slow; d[4] = n_slow; d[5] = n_tile;
    }
#endif

    float *ls = reinterpret_cast<float *>(smem8) + (size_t)wave * 2 * KR * 64;   // aliases stage 0
    int32_t *li = reinterpret_cast<int32_t *>(ls + (size_t)KR * 64);
    grouped_publish<KR>(ls, li, rs, ri, p, lane, uidx, uvalid, tie_mask, (int)blockIdx.y);
    (void)k;
}

// ---- pipelined form of m2d_topk_grouped_bf16: constant-shape steps, insertions under the MFMAs ------------------
// Same arithmetic, same lists, same LDS image as m2d_topk_grouped_bf16 above; what changes is the order in which a
// wave issues its work.  The first form runs each 32-dish tile as [tile_info load, wait] -> [8 x (ds_read, wait,
// MFMA)] -> [16 compares] -> [insertions]: the matrix pipe waits on a global load and on every LDS read, the compares
// wait on the last MFMA (PMC: SQ_VALU_MFMA_BUSY 23 %).  Here a wave keeps three tiles in flight -- step q issues the
// MFMAs of tile q-1, the LDS reads of tile q (each into the registers its k-step just freed) and the threshold
// compares of tile q-2 in the issue slots the MFMAs leave free -- tile meta data come from a 16-entry group table
// held in lanes and walked with scalar ALU, and tiles past the block's range run as dummies (no valid row) so the
// steady-state step has no tail cases (the catalogue image is padded for that).  A first pipelined version that kept
// the accumulator initialisation, the per-stage DMA burst and the branch into an insertion loop measured (stamps of
// scripts/diag/topk_diag.cpp, 100 k dishes, E = 64, cycles per step per wave): body 1 000, insertion path 970 (56 % of
// the steps, 1 730 each), stage wait + barrier 650; with the insertions compiled out still 1 074 + 307, of which 208
// was the accumulator initialisation in front of each chain and ~250 the stage's LDS-DMA issue (address arithmetic +
// 8 pieces per wave, every wave at once right after the barrier, matrix pipe idle).  Hence:
//   * the chain starts from a zero C operand; alpha_P is added where a score enters a list (thresholds are compared
//     as thr - alpha), padding rows are forced to -inf only in the rare partial tile;
//   * the next stage's LDS-DMA pieces are issued a few per step inside the first steps of a stage, from a per-lane
//     source offset computed once (consecutive pieces of a wave are a fixed stride apart in source and destination);
//   * a step never branches into an insertion loop: the 16 compares of tile q-2 run under the MFMAs of tile q-1 as
//     before and leave 16 lane masks in SGPRs; when some lane has a candidate the masks are folded into a per-lane
//     16-bit map of candidate rows (v_addc: map = 2 map + mask bit), the lane's single best score of the tile (the max
//     tree's result; its row is the map's set bit) is parked as (px, pid) and inserted by the NEXT step -- slot
//     ranges of an in-place sorted insert placed between that step's MFMA groups.  Only a tile in which one lane
//     holds two or more candidates (the first tiles of a scan, then rare) takes the immediate per-row path.
// G = groups of 32 users per wave (8 / G waves per block, always 256 users per block).  G = 1 is what is launched.
// G = 2 (one wave per SIMD, every A fragment feeding two independent MFMA chains) was measured slower (100 k dishes
// E = 64: 4.7 ms against 3.7) and is kept only as a template parameter.
// Thresholds are one insertion stale when tile q-2 is compared: more candidates, never fewer.
// HV = true: the ingredient extension.  Dish rows are [H[d] | RE[d]] (E = 2 x the embedding width) and the user operand
// is [a U_high | w_P]: score = <a U_high, H[d]> + <w_P, RE[d]>, no alpha_P term.
// WAVES: 8 / G (a block of 256 users, 128 KiB of LDS, one block per CU), or 4 with G = 1: a block of 128 users over stages of
// half the tiles (64 KiB of LDS), TWO blocks per CU -- a stage barrier then holds up four waves, not eight, and the CU's other
// block keeps the matrix pipes busy meanwhile (launch_grouped: pruned launches, where the waves of a block are unequal).
template <int E, int KR, int G, bool HV = false, int WAVES = 8 / G, bool KEEP = false>
__global__ __launch_bounds__(WAVES * 64, 2) void m2d_topk_grouped_bf16_pipe2(GroupedArgs p)
{
    constexpr int C = 4;
    constexpr int KS = E / 16;                             // k-steps (16 k-values) per tile
    constexpr int S8 = E / 8;                              // 16-B slots per bf16 row
    constexpr int RPB = 256 / (E * 2) > 0 ? 256 / (E * 2) : 1;   // rows per 256-B bank row
    constexpr int TPS = (E == 64 ? 8 : 4) * (WAVES * G) / 8;   // tiles per stage: 64 KiB stages (32 KiB for blocks of four waves)
    static_assert(TPS >= 4, "a stage holds at least four tiles (its pieces are issued in the steps before its last)");
    constexpr int ROW_BYTES = E * 2, TILE_BYTES = 64 * ROW_BYTES, STAGE_BYTES = TPS * TILE_BYTES;
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;                    // 1-KiB DMA pieces per wave per stage
    constexpr int PCNT = (PPW + TPS - 2) / (TPS - 1);      // pieces issued per step (none in a stage's last step)
    constexpr int PSTRIDE = WAVES * 1024;                  // a wave's consecutive pieces: this far apart, source and LDS
    constexpr int EU = HV ? E / 2 : E;                     // embedding width of the user tables
    constexpr int S4 = EU / 4;                             // float4 per f32 row of Personal_Memory
    constexpr int RPK = 16 / KS;                           // compares of the previous tile per k-step
    constexpr bool SHARE = !HV && E == 64;                   // thresholds shared between a user's dish ranges (p.shared_thr): compiled in
                                                           // for E = 64 only -- at E = 128 (240-254 VGPRs) the code alone cost 3.5 % of a
                                                           // pruned call and 6 % of an every-tile one and bought nothing (one instantiation
                                                           // spilled), with the ingredient table (no plan bound to start from) 3 %
    constexpr int AR = KS < 4 ? KS : 4;                    // A-fragment register sets: the LDS reads run AR k-steps ahead
    // (with EXT the tie bits are not kept: a tie at a list's end is a left-out score EQUAL to its last one, and the left-out
    //  scores say so -- the merge's decision reads them)
    constexpr bool EXT = KEEP && !HV && !(E == 128 && KR == 16);   // what the lists leave out is kept for m2d_topk_refine (p.ex_out): an
                                                           // instantiation of its own (the bookkeeping's registers and code cost the scan
                                                           // 6 % also when it is not asked for), and five registers the E = 128, k > 10
                                                           // instantiation does not have
    static_assert(PIECES % WAVES == 0 && (WAVES * 64) % S8 == 0 && ((WAVES * 64 / S8) / RPB) % S8 == 0, "piece layout");

    extern __shared__ __align__(16) unsigned char smem8[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int k = p.k;
    // (user block, dish range) of this workgroup: the grid's (x, y), or -- a pruned launch -- entry blockIdx.x of the
    // launch's item list, longest item first (m2d_plan_items_work / _sort)
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (p.items) {
        const int it = __builtin_amdgcn_readfirstlane(p.items[blockIdx.x]);
        bx = it / p.nsplit;
        by = it - bx * p.nsplit;
    }

    // The users of a launch come in the order the call's plan sorted them into (by relevant-pattern mask, p.order):
    // uidx = the user's index in the CALL (users, plan, outputs), wherever the launch placed it.
    int32_t uidx[G];                                       // 32-bit on purpose (a call holds < 2^31 users): a register less across the scan
    bool uvalid[G];
    const v4f *pmu[G];
    float hc[G][C];                                        // <U_high, CE_c>   Model_Recommender.py:67-75 (from the plan)
    float seed[G];                                         // scan-start bound of the user's final k-th score (from the plan)
    float dlt2[G];                                         // 2 delta: scores this close under a threshold still reach the insertion
    LeftOut lout[G];                                       // what this lane's list leaves out (EXT)
    uint32_t umask_lane = 0u;                              // patterns that can reach the top-k of this lane's user(s)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t pos = (((int64_t)bx * WAVES + wave) * G + g) * 32 + j;
        uvalid[g] = pos < p.nU;
        uidx[g] = uvalid[g] ? (p.order ? p.order[pos] : (int32_t)pos) : 0;
        // everything that hangs on the user's index is fetched in ONE round trip: the id, the plan record, the shared word
        // (read one after the other -- id, its range check, record, word -- they were four in a row at the head of every item)
        const float *rec = p.plan + (size_t)uidx[g] * 8;
        const int32_t uid = uvalid[g] ? p.users[uidx[g]] : 0;
        const float rec0 = rec[0], rec1 = rec[1], rec2 = rec[2], rec3 = rec[3], rec4 = rec[4], rec5 = rec[5], rec7 = rec[7];
        int32_t shared_key = thr_key(-INFINITY);
        if (SHARE && p.shared_thr && uvalid[g])             // what the user's other dish ranges have reached so far (see exchange_thresholds)
            shared_key = __hip_atomic_load(p.shared_thr + (size_t)uidx[g] * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int64_t ul = 0;
        if (uvalid[g]) {
            ul = (int64_t)uid - p.user_base;
            if (ul < 0 || ul >= p.U) {
                if (atomicCAS(&p.err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
                    p.err[1] = uid;
                    p.err[2] = uidx[g];
                    p.err[3] = 0;
                }
                ul = 0;
            }
        }
        pmu[g] = reinterpret_cast<const v4f *>(p.pm) + (size_t)ul * ((C + 1) * S4);
        hc[g][0] = rec1; hc[g][1] = rec2; hc[g][2] = rec3; hc[g][3] = rec4;
        seed[g] = uvalid[g] ? rec0 : INFINITY;              // a lane without a user never has a candidate
        dlt2[g] = (EXT && p.ex_out && uvalid[g]) ? 2.f * rec7 : 0.f;
        lout[g] = M2D_LEFTOUT_NONE;
        if (SHARE && p.shared_thr && uvalid[g]) seed[g] = fmaxf(seed[g], thr_unkey(shared_key));
        umask_lane |= uvalid[g] ? __float_as_uint(rec5) : 0u;
    }
    // the block's patterns: the union over its users.  Tiles of every other pattern are not even fetched.
    __shared__ uint32_t s_umask;
    if (threadIdx.x == 0) s_umask = 0u;
    __syncthreads();
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) umask_lane |= __shfl_xor(umask_lane, off, 64);
    if (lane == 0) atomicOr(&s_umask, umask_lane);
    __syncthreads();
    const uint32_t umask = __builtin_amdgcn_readfirstlane(s_umask);
    // group table: lane q holds the first tile and the row count of mask pattern q (groups are padded to whole tiles)
    int g_first = 0, g_rows = 0;
    if (lane >= 1 && lane < GRP_MAXPAT) {
        g_first = p.grp[lane] >> 5;
        g_rows = p.grp[40 + lane];
    }

    bf16x8 wh[G][KS], wl[G][KS];                           // w_P[u] for k = 16 s + 8 h + (0..7), split hi / lo
    float alpha[G], alpha_prev[G];                         // alpha_P of the tile being multiplied / being compared
    int cur_pat = -1;
    int gp = 0;                                            // group walk: pattern, its tile range and row count
    int64_t g_beg = 0, g_end = 0;
    int g_tot = 0;

    float rs[G][KR];
    int32_t ri[G][KR];
    float thr[G], px[G];                                   // px, pid: parked candidate = this lane's best score of one tile
    int32_t pid[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < KR; ++i) {
            rs[g][i] = -INFINITY;
            ri[g][i] = -1;
        }
        thr[g] = seed[g];
        px[g] = -INFINITY;
        pid[g] = -1;
        alpha[g] = alpha_prev[g] = 0.f;
    }
    bool pend = false;                                     // wave-uniform: some (px, pid) waits to be inserted
    unsigned long long tie_mask[G];                        // lanes with a tie event at their list's present last value (tie_update)
#pragma unroll
    for (int g = 0; g < G; ++g) tie_mask[g] = 0ull;

    const int64_t per = (p.tiles + p.nsplit - 1) / p.nsplit;
    const int64_t t_begin = (int64_t)by * per;
    const int64_t t_end = min(p.tiles, t_begin + per);
    const int n_phys = (int)(t_end > t_begin ? t_end - t_begin : 0);   // tiles of this block's dish range
    const int nst = n_phys > 0 ? (n_phys + 2) / TPS + 1 : 0;           // stages of that range (the image is padded for the overhang)
    // The stages to step through: those that hold a tile of a pattern in `umask`, in scan order, as up to 15 ranges of
    // stage numbers (relative to t_begin) kept in lanes -- range i in lane i of r_first / r_cnt.  A stage that straddles a
    // group boundary brings a few tiles of a neighbouring pattern along; they are scored like any other.
    int r_first = 0, r_cnt = 0, nranges = 0, vstages = 0;
    {
        int last_end = -1;
        for (int q = 1; q < GRP_MAXPAT; ++q) {
            const int rows = __builtin_amdgcn_readlane(g_rows, q);
            if (rows == 0 || !((umask >> q) & 1u)) continue;
            const int64_t gt0 = __builtin_amdgcn_readlane(g_first, q), gt1 = gt0 + ((rows + 31) >> 5);
            const int64_t lo = gt0 > t_begin ? gt0 : t_begin, hi = gt1 < t_end ? gt1 : t_end;
            if (lo >= hi) continue;
            int s0 = (int)((lo - t_begin) / TPS);
            const int s1 = (int)((hi - 1 - t_begin) / TPS);
            if (s0 <= last_end) s0 = last_end + 1;
            if (s0 > s1) continue;
            r_first = lane == nranges ? s0 : r_first;
            r_cnt = lane == nranges ? s1 - s0 + 1 : r_cnt;
            ++nranges;
            vstages += s1 - s0 + 1;
            last_end = s1;
        }
    }
    const int64_t n = (int64_t)vstages * TPS;                       // tiles the steps below go through ("virtual" tiles 0 .. n - 1)
    // walker over the ranges: the physical stage of the next virtual stage (beyond the last: a stage number no range holds)
    int w_idx = -1, w_stage = 0, w_left = 0;
    auto next_stage = [&]() __attribute__((always_inline)) {
        if (w_left == 0) {
            ++w_idx;
            if (w_idx < nranges) {
                w_stage = __builtin_amdgcn_readlane(r_first, w_idx);
                w_left = __builtin_amdgcn_readlane(r_cnt, w_idx);
            } else {
                w_stage = 0x20000000;
                w_left = 0x20000000;
            }
        }
        --w_left;
        return w_stage++;
    };
    int ps_m1 = 0x20000000, ps_0 = 0x20000000, ps_p1 = 0x20000000;   // physical stages of virtual stages v - 1, v, v + 1 (v = q / TPS)

    // LDS-DMA: this lane's source offset inside a stage for its wave's first piece
    const unsigned char *const src_base = reinterpret_cast<const unsigned char *>(p.rs16) + (size_t)t_begin * TILE_BYTES;
    int dma_off;
    {
        const int g = wave * 64 + lane;                    // physical 16-B slot in the stage image
        const int rw = g / S8, sl = g - rw * S8;           // stage row (hi and lo rows alike), slot in row
        const int q = sl ^ ((rw / RPB) & (S8 - 1));        // logical slot that must land here
        dma_off = rw * ROW_BYTES + q * 16;
    }
    // a wave's pieces of one stage: buffer_load ... lds with the stage's base in an SGPR descriptor, the per-lane
    // source offset (computed once, above) as the 32-bit VGPR offset and the piece's 8 KiB multiple as the scalar
    // offset -- no 64-bit per-lane address arithmetic per piece (1-2 % over global_load_lds with VGPR addresses)
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    auto issue_pieces = [&](const int stage, const int buf, const int first, const int count) __attribute__((always_inline)) {
        if (stage >= nst) return;                          // past the dish range (or no stage left): nothing to fetch
        const uint64_t b = (uint64_t)(uintptr_t)(src_base + (size_t)stage * STAGE_BYTES);
        v4i_ rsrc;                                         // raw buffer (stride 0) over this stage of the catalogue image
        rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        rsrc.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
        rsrc.z = STAGE_BYTES;
        rsrc.w = 0x00020000;
        unsigned char *dst = smem8 + (size_t)(buf & 1) * STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int c = 0; c < count; ++c) {
            const int pp = first + c;
            if (pp < PPW) {
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)(dst + pp * PSTRIDE));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             ::"s"(m0v), "v"(dma_off), "s"(rsrc), "s"(pp * PSTRIDE) : "memory", "m0");
            }
        }
    };

    // this lane's byte offset inside a tile image: row j, slot (2 ks + h) ^ key = (2 ks) ^ (h ^ key)
    const int key = (j / RPB) & (S8 - 1);
    const int lane_off = j * ROW_BYTES + ((h ^ key) << 4);

    // A fragments, AR k-steps deep: set ks % AR holds k-step ks of the tile being multiplied and is refilled, as soon as
    // its three MFMAs are issued, with the k-step AR further on in the (tile, k) stream -- the same tile's at E = 128
    // (8 k-steps, 4 sets: 32 VGPRs instead of 64, which is what keeps this form under 256 registers there), the next
    // tile's at E = 64
    bf16x8 ah[AR], al[AR];
    v16f acc0[G], acc1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[g][r] = acc1[g][r] = -INFINITY;
    }
    const v16f zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (n > 0) {
        ps_0 = next_stage();
        ps_p1 = next_stage();
        issue_pieces(ps_0, 0, 0, PPW);
        wait_all_vmem();
        __syncthreads();
        issue_pieces(ps_p1, 1, 0, PCNT);                   // what step "0" of the first stage would have issued
#pragma unroll
        for (int ks = 0; ks < AR; ++ks) {                  // the first AR k-steps of tile 0
            const unsigned char *a = smem8 + (lane_off ^ (ks << 5));
            ah[ks] = *reinterpret_cast<const bf16x8 *>(a);
            al[ks] = *reinterpret_cast<const bf16x8 *>(a + 32 * ROW_BYTES);
        }
    }

#if M2D_DIAG & 16
    unsigned long long t_body = 0, t_slow = 0, t_bar = 0, n_slow = 0, n_step = 0, n_ins = 0, t0_, t1_, t_slow_first = 0, n_slow_first = 0;
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    auto share_threshold = [&](const int g) __attribute__((always_inline)) {
        // the user's other lane (l ^ 32): a score below the larger of the two lists' last entries cannot be in the user's
        // merged top-KR, nor can one below the smaller of their MIDDLE entries (KR / 2 entries of each list are at or above
        // it: KR scores in all) -- with the dishes dealt evenly to the two lanes that is about the merged list's last
        // entry itself, where each lane's own last is about its 2 KR-th (candidate tiles 17 % -> 15 %, insertions per
        // lane 110 -> 85 in scripts/diag/topk_scan_sim.py); and never below the scan-start bound
        const float t = rs[g][KR - 1], m = rs[g][KR / 2 - 1];
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
        const auto sm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
        thr[g] = fmaxf(fmaxf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])),
                             fminf(__uint_as_float(sm[0]), __uint_as_float(sm[1]))), seed[g]);
    };

    // A launch cut into dish ranges runs a user's ranges as separate workgroups, each with lists of its own -- and each
    // used to climb from the scan-start bound on its own, re-inserting what the others had long outgrown (8 ranges: about
    // five times the insertions of one scan).  Any threshold of any range is a lower bound of the user's FINAL k-th score
    // (k dishes at or above it exist), so the ranges meet in one word per user, plan record word 6: once per stage a lane
    // sends its threshold there (agent-scope atomic max on the ordered key) and takes what comes back -- the largest any
    // range had sent -- as its floor from the NEXT stage on: the answer has a whole stage to arrive and is never waited
    // for.  What the word holds when a lane looks depends on timing; the lists do not: a dish of the final top-k scores at or
    // above every lower bound of the k-th score, reaches its range's insertion whatever the floor, and stays in that
    // range's list; tie events at the final k-th value likewise involve scores at or above every floor.
    int32_t pend_key[G];
#pragma unroll
    for (int g = 0; g < G; ++g) pend_key[g] = thr_key(-INFINITY);
    auto exchange_thresholds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            seed[g] = fmaxf(seed[g], thr_unkey(pend_key[g]));
            thr[g] = fmaxf(thr[g], seed[g]);
            if (uvalid[g]) {
                int32_t *word = p.shared_thr + (size_t)uidx[g] * 8;
                if (thr[g] > seed[g]) {                     // news: above everything this lane has heard or said (seed = that floor)
                    pend_key[g] = __hip_atomic_fetch_max(word, thr_key(thr[g]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    seed[g] = thr[g];
                } else                                      // nothing to say: a read leaves the line shared between the XCDs' L2s
                    pend_key[g] = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    // the interleaved body: M(q-1) into accN, L(q), the compares + max tree of tile q-2 (accP), and -- INS -- the
    // parked insertion, slot ranges from the end of the list up, one range per k-step group
    auto body = [&](auto ins_tag, v16f (&accN)[G], const v16f (&accP)[G], const int img_prev, const int img_off,
                    const float (&thr_rel)[G], unsigned long long (&m)[G][16], float (&mx)[G]) __attribute__((always_inline)) {
        constexpr bool INS = decltype(ins_tag)::value;
        constexpr bool PIN = WAVES == 4;
        float x[G], old_last[G];
        int32_t old_id[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            x[g] = fmaxf(px[g], -INFINITY);
            mx[g] = -INFINITY;
            old_last[g] = rs[g][KR - 1];
            old_id[g] = ri[g][KR - 1];
        }
        // As common code of the two bodies (with / without the parked insertion) the sixteen compares are hoisted in front
        // of the branch, ahead of the first MFMA.  Blocks of four waves: an opaque copy of the threshold per body keeps them
        // where they are written, in the issue slots between this body's MFMAs (pruned call at 100 k dishes 0.591 -> 0.581 ms).
        // Blocks of eight waves are better off with the hoisted form (every-tile scan 2.516 against 2.527 ms, 1 M dishes
        // 20.30 against 20.51).
        float tr[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tr[g] = thr_rel[g];
            if constexpr (WAVES == 4) asm volatile("" : "+v"(tr[g]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = ks % AR;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[st], wh[g][ks], ks == 0 ? zero16 : accN[g], 0, 0, 0);
                accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st], wl[g][ks], accN[g], 0, 0, 0);
                accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st], wh[g][ks], accN[g], 0, 0, 0);
            }
            // k-step ks + AR of this tile (q-1), or k-step ks + AR - KS of the next (q)
            const unsigned char *a = smem8 + (ks + AR < KS ? (img_prev ^ ((ks + AR) << 5)) : (img_off ^ ((ks + AR - KS) << 5)));
            ah[st] = *reinterpret_cast<const bf16x8 *>(a);
            al[st] = *reinterpret_cast<const bf16x8 *>(a + 32 * ROW_BYTES);
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int r = ks * RPK; r < (ks + 1) * RPK; ++r) {
                    m[g][r] = __ballot(accP[g][r] >= tr[g]);           // one v_cmp into an SGPR pair; folded into a
                    mx[g] = fmaxf(mx[g], accP[g][r]);                  // per-lane row map only if some lane has a candidate
                }
                if (PIN && INS) asm volatile("" : "+v"(mx[g]));
                if constexpr (INS) {
                    // slots [lo, hi) of the list, highest ranges first
                    constexpr int PER = (KR + KS - 1) / KS;
                    constexpr int L0 = KR - PER > 0 ? KR - PER : 0, L1 = KR - 2 * PER > 0 ? KR - 2 * PER : 0,
                                  L2 = KR - 3 * PER > 0 ? KR - 3 * PER : 0, L3 = KS == 4 ? 0 : (KR - 4 * PER > 0 ? KR - 4 * PER : 0);
                    if (ks == 0) { sorted_insert_range<KR, L0, KR>(rs[g], ri[g], x[g], pid[g]); if (PIN) pin_range<KR, L0, KR>(rs[g]); }
                    if (ks == 1) { sorted_insert_range<KR, L1, L0>(rs[g], ri[g], x[g], pid[g]); if (PIN) pin_range<KR, L1, L0>(rs[g]); }
                    if (ks == 2) { sorted_insert_range<KR, L2, L1>(rs[g], ri[g], x[g], pid[g]); if (PIN) pin_range<KR, L2, L1>(rs[g]); }
                    if (ks == 3) { sorted_insert_range<KR, L3, L2>(rs[g], ri[g], x[g], pid[g]); if (PIN) pin_range<KR, L3, L2>(rs[g]); }
                    if (KS > 4 && ks == 4) sorted_insert_range<KR, 0, L3>(rs[g], ri[g], x[g], pid[g]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (INS) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (!EXT) tie_mask[g] = tie_update(tie_mask[g], x[g], old_last[g], rs[g][KR - 1]);
                if (EXT) left_out_note(lout[g], x[g], pid[g], old_last[g], old_id[g], rs[g][KR - 1] - dlt2[g]);
                share_threshold(g);
            }
        }
    };

    auto step = [&](v16f (&accN)[G], v16f (&accP)[G], const int64_t q) __attribute__((always_inline)) {
#if M2D_DIAG & 16
        STAMP(t0_);
#endif
        const int sub = (int)(q & (TPS - 1));
        if (sub == 0) {                                    // tile q opens stage q / TPS: it must have landed, for every wave
            wait_all_vmem();
            __syncthreads();                               // also: every wave is done reading the buffer refilled next
            ps_m1 = ps_0;
            ps_0 = ps_p1;
            ps_p1 = next_stage();
            if (SHARE && p.shared_thr) exchange_thresholds();
        }
        if (KS > AR && sub == 1) {
            // With more k-steps than fragment sets (E = 128: KS = 8, AR = 4) a tile's last KS - AR k-steps are read one step
            // after its first ones: the LAST tile of the previous stage was still being read during step "sub 0", after that
            // stage's barrier.  Its LDS region is refilled by the pieces issued at sub = TPS - 2; nothing kept a wave that
            // runs two steps ahead (no insertions, while another wave rebuilds its operand at a pattern switch or works
            // through a tile of candidates) from issuing them under the reader: wrong scores for that one tile, seen once
            // the scan-start thresholds made some waves that much faster than others.  So: every wave is past step "sub 0"
            // before any wave goes on to the steps that refill that region.
            asm volatile("s_barrier" ::: "memory");
        }
        if (sub < TPS - 1) issue_pieces(ps_p1, (int)(q / TPS + 1), sub * PCNT, PCNT);
#if M2D_DIAG & 16
        STAMP(t1_); t_bar += t1_ - t0_; t0_ = t1_;
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) alpha_prev[g] = alpha[g];   // tile q-2 was multiplied under the previous step's alpha
        int nvalid = 0;
        // physical tile of virtual tile q - 1 (the one being multiplied): its stage is v or v - 1
        const int pt1 = ((q - 1) / TPS == q / TPS ? ps_0 : ps_m1) * TPS + (int)((q - 1) & (TPS - 1));
        if (q - 1 < n && pt1 < n_phys) {
            const int64_t t = t_begin + pt1;
            while (t >= g_end) {                            // next non-empty group (scalar; at most 15 times per block)
                ++gp;
                g_tot = __builtin_amdgcn_readlane(g_rows, gp);
                g_beg = __builtin_amdgcn_readlane(g_first, gp);
                g_end = g_beg + ((g_tot + 31) >> 5);
            }
            const int left = g_tot - (int)(t - g_beg) * 32;
            nvalid = left < 32 ? left : 32;
            // a stage that straddles a group boundary brings a few tiles of a neighbouring pattern along: if none of the
            // block's users can rank a dish of that pattern, its tiles are dummies (no row valid) and the operands stay
            if (!((umask >> gp) & 1u)) nvalid = 0;
            else if (gp != cur_pat) {
                cur_pat = gp;
                const int pat = gp;
                const float inv_n = 1.0f / (float)__builtin_popcount(pat);
                const float beta = p.b * inv_n;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float hs = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) hs += ((pat >> c) & 1) ? hc[g][c] : 0.f;
                    alpha[g] = HV ? 0.f : p.a * (hs * inv_n);
                    // E = 64: the pattern's rows category by category -- a category's eight float4 in flight together, one wait,
                    // then the adds (the same sums in the same order).  Written k-step by k-step with the category test inside,
                    // the loads came out as sixteen exec-masked pairs, each waited for before the next was issued: sixteen
                    // round trips in a row at every pattern switch, with the block's matrix pipe idle
                    constexpr bool BYCAT = !HV && KS == 4;
                    v4f wacc[BYCAT ? KS : 1][2];
                    if constexpr (BYCAT) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) wacc[ks][0] = wacc[ks][1] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            if ((pat >> c) & 1) {
                                v4f ld[KS][2];
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks) {
                                    const v4f *row = pmu[g] + (c + 1) * S4 + 4 * ks + 2 * h;
                                    ld[ks][0] = row[0];
                                    ld[ks][1] = row[1];
                                }
#pragma unroll
                                for (int ks = 0; ks < KS; ++ks) {
                                    wacc[ks][0] += ld[ks][0];
                                    wacc[ks][1] += ld[ks][1];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {                   // unrolled: every register index is static
                        v4f w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
                        if constexpr (BYCAT) {
                            w0 = wacc[ks][0] * beta;
                            w1 = wacc[ks][1] * beta;
                        } else
                        if (HV && ks < KS / 2) {                        // k < EU: a U_high against H[d]
                            const v4f *row = pmu[g] + 4 * ks + 2 * h;
                            w0 = row[0] * p.a;
                            w1 = row[1] * p.a;
                        } else {
                            const int kk = HV ? ks - KS / 2 : ks;       // k - EU: w_P against RE[d]
#pragma unroll
                            for (int c = 0; c < C; ++c) {
                                if ((pat >> c) & 1) {
                                    const v4f *row = pmu[g] + (c + 1) * S4 + 4 * kk + 2 * h;
                                    w0 += row[0];
                                    w1 += row[1];
                                }
                            }
                            w0 *= beta;
                            w1 *= beta;
                        }
                        const float xx[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                        bf16x8 vh, vl;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const __bf16 xh = (__bf16)xx[i];
                            vh[i] = xh;
                            vl[i] = (__bf16)(xx[i] - (float)xh);
                        }
                        wh[g][ks] = vh;
                        wl[g][ks] = vl;
                    }
                }
            }
        }
        // accumulators carry no alpha: compare against thr - alpha -- less the two roundings between `acc >= thr - alpha` and
        // `acc + alpha >= thr` (the sum is quantised at ulp(alpha), many ulps of acc under the 0.99 : 0.01 blend), so that every
        // score whose TOTAL reaches thr gets to the insertion, which compares totals: an equal total refused here would be a
        // tie nobody records (seen as a tie-listed user that one split count listed and another did not)
        float thr_rel[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float d = thr[g] - alpha_prev[g];
            thr_rel[g] = d - 2.4e-7f * (fabsf(thr[g]) + fabsf(alpha_prev[g])) - dlt2[g];   // 2^-22 (|thr| + |alpha|); thr = +inf: NaN, no candidate
        }
        const int img_off = (int)(((q / TPS) & 1) * STAGE_BYTES + sub * TILE_BYTES) + lane_off;   // tile q
        const int img_prev = (int)((((q - 1) / TPS) & 1) * STAGE_BYTES + ((q - 1) & (TPS - 1)) * TILE_BYTES) + lane_off;
        unsigned long long m[G][16];                       // lane masks: row r of tile q-2 beats the lane's threshold
        float mx[G];
        if (pend) body(std::true_type{}, accN, accP, img_prev, img_off, thr_rel, m, mx);
        else body(std::false_type{}, accN, accP, img_prev, img_off, thr_rel, m, mx);
        if (nvalid < 32) {                                 // a group's last tile, or a dummy tile: padding rows never rank
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    accN[g][r] = (4 * h + (r & 3) + 8 * (r >> 2) < nvalid) ? accN[g][r] : -INFINITY;
            }
        }
        unsigned long long anyc = 0ull;
#pragma unroll
        for (int g = 0; g < G; ++g) anyc |= __ballot(mx[g] >= thr_rel[g]);
#if M2D_DIAG & 16
        STAMP(t1_); t_body += t1_ - t0_; t0_ = t1_; ++n_step;
#endif
        pend = false;
#if M2D_DIAG & 8
        asm volatile("" ::"s"(anyc), "s"(m[0][0] | m[0][5] | m[0][10] | m[0][15]), "v"(mx[0]), "v"(mx[G - 1]));
#endif
        if ((M2D_DIAG & 8) ? false : anyc != 0ull) {       // some lane of tile q-2 beat its threshold
            const int pt2 = ((q - 2) / TPS == q / TPS ? ps_0 : ps_m1) * TPS + (int)((q - 2) & (TPS - 1));   // physical tile of tile q - 2
            const int32_t sbase = (int32_t)((t_begin + pt2) * 32) + 4 * h;
            // per-lane 16-bit map of candidate rows (bit 15 - r), built here -- in the quarter of the steps that have a
            // candidate -- from the sixteen lane masks: map = 2 map + mask bit, one v_addc each
            uint32_t rowmap[G];
            bool cand[G];
            unsigned long long multi = 0ull;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                rowmap[g] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(rowmap[g]) : "v"(rowmap[g]), "s"(m[g][r]) : "vcc");
                cand[g] = rowmap[g] != 0u;
                multi |= __ballot((rowmap[g] & (rowmap[g] - 1u)) != 0u);                      // two or more bits set
            }
            // one candidate per lane at most: it is the lane's maximum, its row is the map's only set bit, and it is parked
            // for the next step's body (written ahead of the branch, not as its else-arm: as the two arms of a diamond the
            // compiler gave the lists other registers in the multi-candidate arm and paid for it in THIS arm -- 26 v_mov into
            // those registers and 26 back at the merge, in every step with a candidate)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int r = __builtin_clz(rowmap[g] | 1u) - 16;          // bit 15 - r  ->  r  (lanes without a candidate: any)
                px[g] = cand[g] ? mx[g] + alpha_prev[g] : -INFINITY;
                pid[g] = sbase + (r & 3) + 8 * (r >> 2);
            }
            pend = (M2D_DIAG & 128) ? (__ballot(px[0] == 12345.678f) != 0ull) : true;
            if ((M2D_DIAG & 64) ? false : __builtin_expect(multi != 0ull, 0)) { // immediate path: some lane holds two or more candidates of this tile
                pend = false;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // Two ways to get them in.  Row by row: every row in which SOME lane has a candidate is inserted by the
                    // whole wave (about 55 VALU a row) -- the first tiles of a scan, where every lane wants most rows.
                    // Lane by lane: each lane takes its own next candidate row (its map's highest bit), the value is
                    // picked out of the sixteen accumulators by sixteen compares and selects, one insertion per pass
                    // (about 85 VALU a pass, passes = the most candidates any lane holds).  In the tiles of a pattern the
                    // block's users all want -- the only tiles a pruned scan still visits -- nearly every row has a taker
                    // but a lane has two or three: 16 x 55 against 3 x 85.
                    // (Which is cheaper was worked out per tile from the number of rows with a taker -- sixteen scalar
                    //  compares the compiler turned into 32 VALU + 32 SALU, more than the choice ever saved: row by row wins
                    //  only when at most 3 / 4 / 6 rows have takers while a lane holds 2 / 3 / 4.  Now: lane by lane up to
                    //  four candidates per lane.)
                    const uint32_t pc = (uint32_t)__builtin_popcount(rowmap[g]);
                    const bool b5 = __ballot(pc >= 5u) != 0ull;
                    // (No loop and no else-arm below: plain ifs.  As a `while` beside an else-arm the compiler moved the lists
                    //  into other registers on the way in and back on the way out, 50 v_mov per multi-candidate tile.)
                    uint32_t rm = rowmap[g];
                    if (b5) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (m[g][r] != 0ull) {
                                const float ol = rs[g][KR - 1], xv = accP[g][r] + alpha_prev[g];
                                const int32_t oi = ri[g][KR - 1];
                                sorted_insert_inplace<KR>(rs[g], ri[g], xv, sbase + (r & 3) + 8 * (r >> 2));
                                if (!EXT) tie_mask[g] = tie_update(tie_mask[g], xv, ol, rs[g][KR - 1]);
                                if (EXT) left_out_note(lout[g], fmaxf(xv, -INFINITY), sbase + (r & 3) + 8 * (r >> 2), ol, oi, rs[g][KR - 1] - dlt2[g]);
                            }
                        }
                        rm = 0u;
                    }
                    auto pass = [&]() __attribute__((always_inline)) {     // a lane's rows in ascending order, one per pass
                        const bool has = rm != 0u;
                        const int r = __builtin_clz(rm | 1u) - 16;
                        // the lane's row r out of its sixteen accumulators: a binary tree of selects on the bits of r (four
                        // lane masks + fifteen v_cndmask; sixteen compares + sixteen selects before)
                        const unsigned long long b0 = __ballot((r & 1) != 0), b1 = __ballot((r & 2) != 0), b2 = __ballot((r & 4) != 0),
                                                 b3 = __ballot((r & 8) != 0);
                        float t8[8], t4[4], t2[2];
#pragma unroll
                        for (int q = 0; q < 8; ++q) t8[q] = __int_as_float(lane_select(b0, __float_as_int(accP[g][2 * q + 1]), __float_as_int(accP[g][2 * q])));
#pragma unroll
                        for (int q = 0; q < 4; ++q) t4[q] = __int_as_float(lane_select(b1, __float_as_int(t8[2 * q + 1]), __float_as_int(t8[2 * q])));
#pragma unroll
                        for (int q = 0; q < 2; ++q) t2[q] = __int_as_float(lane_select(b2, __float_as_int(t4[2 * q + 1]), __float_as_int(t4[2 * q])));
                        float xv = __int_as_float(lane_select(b3, __float_as_int(t2[1]), __float_as_int(t2[0])));
                        xv = has ? xv + alpha_prev[g] : -INFINITY;    // a lane without a candidate inserts nothing
                        const float ol = rs[g][KR - 1];
                        const int32_t oi = ri[g][KR - 1];
                        sorted_insert_inplace<KR>(rs[g], ri[g], xv, sbase + (r & 3) + 8 * (r >> 2));
                        if (!EXT) tie_mask[g] = tie_update(tie_mask[g], xv, ol, rs[g][KR - 1]);
                        if (EXT) left_out_note(lout[g], xv, sbase + (r & 3) + 8 * (r >> 2), ol, oi, rs[g][KR - 1] - dlt2[g]);
                        rm &= ~(0x8000u >> r);
                    };
                    if (__ballot(rm != 0u) != 0ull) {               // (not b5: some lane holds two to four)
                        pass();
                        pass();
                        if (__ballot(rm != 0u) != 0ull) {
                            pass();
                            if (__ballot(rm != 0u) != 0ull) pass();
                        }
                    }
                    share_threshold(g);
                }
#if M2D_DIAG & 16
                ++n_ins;
#endif
            }
#if M2D_DIAG & 16
            asm volatile("" ::"v"(thr[0]), "v"(px[0]));
            STAMP(t1_); t_slow += t1_ - t0_; ++n_slow;
            if (q <= TPS + 2) { t_slow_first += t1_ - t0_; ++n_slow_first; }
#endif
        }
    };

    for (int64_t q = 1; q <= n + 1 && n > 0; q += 2) {
        step(acc0, acc1, q);
        step(acc1, acc0, q + 1);
    }
    if (pend) {                                            // the last parked candidates
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float ol = rs[g][KR - 1];
            const int32_t oi = ri[g][KR - 1];
            sorted_insert_inplace<KR>(rs[g], ri[g], px[g], pid[g]);
            if (!EXT) tie_mask[g] = tie_update(tie_mask[g], fmaxf(px[g], -INFINITY), ol, rs[g][KR - 1]);
            if (EXT) left_out_note(lout[g], fmaxf(px[g], -INFINITY), pid[g], ol, oi, rs[g][KR - 1] - dlt2[g]);
        }
    }
    if (SHARE && p.shared_thr && n > 0) {                  // what this range ends with: ranges of the user that start later begin there
#pragma unroll
        for (int g = 0; g < G; ++g) {
            share_threshold(g);
            if (uvalid[g])
                __hip_atomic_fetch_max(p.shared_thr + (size_t)uidx[g] * 8, thr_key(thr[g]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    wait_all_vmem();                                       // no LDS-DMA may land after the lists are published below
    __syncthreads();
    if (p.tiles_scanned && threadIdx.x == 0) atomicAdd(p.tiles_scanned, (unsigned long long)n);
#if M2D_DIAG & 16
    if (lane == 0 && p.dbg) {
        unsigned long long *d = p.dbg + ((size_t)(by * ((p.nU + 255) / 256) + bx) * WAVES + wave) * 8;
        d[0] = t_body; d[1] = n_ins; d[2] = t_bar; d[3] = t_slow; d[4] = n_slow; d[5] = n_step;
        d[6] = __builtin_amdgcn_s_memtime() - clk0; d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
#if M2D_DIAG & 32
        d[7] = t_slow_first; d[1] = n_slow_first;          // candidate handling of an item's first stage (steps 1 .. TPS + 2)
#endif
    }
#endif

#pragma unroll
    for (int g = 0; g < G; ++g) {
        float *ls = reinterpret_cast<float *>(smem8) + (size_t)(wave * G + g) * 2 * KR * 64;   // aliases stage 0
        int32_t *li = reinterpret_cast<int32_t *>(ls + (size_t)KR * 64);
        grouped_publish<KR>(ls, li, rs[g], ri[g], p, lane, uidx[g], uvalid[g], tie_mask[g], by, lout[g]);
    }
    (void)k;
}

// row width of the sorted dish table: E itself where a kernel is instantiated for it, else the next such width
// (multiples of 4 up to 256 only; 0 = no grouped kernel serves this E)
int grouped_row_width(int E)
{
    if (E == 32 || E == 64 || E == 128) return E;
    if (E < 4 || E > 256 || E % 4 != 0) return 0;
    return E < 32 ? 32 : (E < 64 ? 64 : (E < 128 ? 128 : 256));
}

int ensure_grouped(m2d_engine *h, hipStream_t st)
{
    if (h->grp_valid) return M2D_OK;
    const int64_t I = h->I;
    const int nblk = (int)((I + 255) / 256);
    const int64_t max_tiles = (I + 31) / 32 + GRP_MAXPAT;
    const int64_t cap_rows = (max_tiles + 16) * 32;          // + one stage of zero rows past the last tile
    const int EW = h->dish_high ? 2 * h->E : grouped_row_width(h->E);   // ingredient extension: rows are [H[d] | RE[d]]
    if (h->grp_cap_rows != cap_rows || h->grp_ew != EW || !h->grp_rs) {
        for (void *q : {(void *)h->grp_rs, (void *)h->grp_rs16, (void *)h->grp_perm, (void *)h->grp_tile_info, (void *)h->grp_work})
            if (q) M2D_HIP_TRY(h, hipFree(q));
        h->grp_rs = nullptr; h->grp_rs16 = nullptr; h->grp_perm = nullptr; h->grp_tile_info = nullptr; h->grp_work = nullptr;
        M2D_HIP_TRY(h, hipMalloc((void **)&h->grp_rs, (size_t)cap_rows * EW * sizeof(float)));
        M2D_HIP_TRY(h, hipMalloc((void **)&h->grp_rs16, (size_t)cap_rows * EW * 4));
        M2D_HIP_TRY(h, hipMalloc((void **)&h->grp_perm, (size_t)cap_rows * sizeof(int32_t)));
        M2D_HIP_TRY(h, hipMalloc((void **)&h->grp_tile_info, (size_t)max_tiles * sizeof(int32_t)));
        M2D_HIP_TRY(h, hipMalloc((void **)&h->grp_work, ((size_t)nblk * GRP_KEYS + GRP_WORDS + 2 * 2) * sizeof(int32_t) +
                                                        (size_t)I * sizeof(float)));
        h->grp_cap_rows = cap_rows;
        h->grp_ew = EW;
    }
    int32_t *blk_hist = h->grp_work, *grp = h->grp_work + (size_t)nblk * GRP_KEYS, *flags = grp + 32;
    float *stat = reinterpret_cast<float *>(grp + GRP_STAT);
    double *acc = reinterpret_cast<double *>(grp + GRP_WORDS);           // 8-byte aligned: nblk * GRP_KEYS and GRP_WORDS are even
    float *norm = reinterpret_cast<float *>(grp + GRP_WORDS + 4);
    M2D_HIP_TRY(h, hipMemsetAsync(flags, 0, sizeof(int32_t), st));
    M2D_HIP_TRY(h, hipMemsetAsync(acc, 0, 2 * sizeof(double), st));
    M2D_HIP_TRY(h, hipMemsetAsync(h->grp_perm, 0xFF, (size_t)cap_rows * sizeof(int32_t), st));
    // scan order by the norm of the row that carries the larger term: H[d] (weight a) when the ingredient table is set
    hipLaunchKernelGGL(m2d_grp_norm_stats, dim3(nblk), dim3(256), 0, st, h->dish_high ? h->dish_high : h->re, I, h->E, norm, acc);
    hipLaunchKernelGGL(m2d_grp_norm_params, dim3(1), dim3(1), 0, st, acc, I, stat);
    M2D_HIP_TRY(h, hipMemsetAsync(grp + GRP_RMAX, 0, 16 * sizeof(int32_t), st));
    hipLaunchKernelGGL(m2d_grp_hist, dim3(nblk), dim3(256), 0, st, h->dish_cats, norm, stat, I, h->C, blk_hist, flags, grp + GRP_RMAX);
    hipLaunchKernelGGL(m2d_grp_scan, dim3(1), dim3(GRP_KEYS * GRP_SCAN_SPLIT), 0, st, blk_hist, nblk, grp, h->grp_tile_info);
    hipLaunchKernelGGL(m2d_grp_scatter, dim3(nblk), dim3(256), 0, st, h->dish_cats, norm, stat, I, h->C, blk_hist, grp, h->grp_perm);
    hipLaunchKernelGGL(m2d_grp_gather, dim3((unsigned)((cap_rows + 3) / 4)), dim3(256), 0, st, h->re, h->dish_high,
                       h->grp_perm, cap_rows, h->E, EW, h->grp_rs, reinterpret_cast<__bf16 *>(h->grp_rs16));
    M2D_HIP_TRY(h, hipGetLastError());
    int32_t host[4] = {0, 0, 0, 0};   // tiles, slots, flags, "a table value is not finite"  (a table build may synchronise)
    M2D_HIP_TRY(h, hipMemcpyAsync(host, grp + 16, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipMemcpyAsync(host + 2, flags, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipMemcpyAsync(host + 3, h->nonfinite_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    h->grp_tiles = host[0];
    h->grp_binary = host[2] == 0;
    // w_P = sum of the pattern's U_low rows leaves out the 0 * U_low[c] products of the other categories: with inf / NaN in
    // a table those are NaN in the reference formula, and the dense kernel (which multiplies them) serves the call
    h->grp_nonfinite = host[3] != 0;
    h->grp_nonfinite_known = true;
    h->grp_valid = true;
    return M2D_OK;
}

// m2d_write_memory adds into Personal_Memory: the sorted dish rows stay valid, but the device word "a table value is
// inf / NaN" may have been set by its row check -- read it again before choosing between the pattern-grouped kernels
// (which leave out the 0 * U_low[c] products) and the dense one
int refresh_grouped_nonfinite(m2d_engine *h, hipStream_t st)
{
    if (h->grp_nonfinite_known) return M2D_OK;
    int32_t word = 0;
    M2D_HIP_TRY(h, hipMemcpyAsync(&word, h->nonfinite_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    h->grp_nonfinite = word != 0;
    h->grp_nonfinite_known = true;
    return M2D_OK;
}

void m2d_launch_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *out_s, int32_t *out_i,
                             hipStream_t st, const float *tie_in = nullptr, float *tie_out = nullptr, int32_t *tie_list = nullptr,
                             int64_t I = 0, const float *ex_in = nullptr, float *ex_out = nullptr, const float *plan = nullptr,
                             int32_t *rcount = nullptr)
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
                              float *ex = nullptr, float *ex_final = nullptr, const float *plan = nullptr, int32_t *rcount = nullptr)
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

// shared tail of every MFMA retrieval launch: dish-range splits -> partial lists in scratch.  The grouped kernels run
// ONE block per CU (128 KiB of LDS), so the grid is ublocks x nsplit blocks dealt out in rounds of num_cu: the fewest
// splits whose last round is at least 90 % full are taken (each split repeats the early, insertion-heavy part of a
// scan and adds a merge pass: 65 536 users x 100 k dishes ran 3.69 ms with 2 splits and 2.91 ms with 1).  Few users:
// up to max_splits blocks per user block, so that a single query still uses the whole chip.
int pick_splits(m2d_engine *h, int64_t ublocks, int64_t tiles, int64_t min_tiles_per_split, int max_splits = 64)
{
    const int64_t cus = h->num_cu;
    int64_t cap = tiles / min_tiles_per_split > 1 ? tiles / min_tiles_per_split : 1;
    if (cap > max_splits) cap = max_splits;
    int64_t lim = (2 * cus + ublocks - 1) / ublocks;        // beyond two rounds' worth of blocks nothing is gained
    if (lim < 1) lim = 1;
    if (lim > cap) lim = cap;
    int nsplit = 1;
    double best = 0.0;
    for (int64_t ns = 1; ns <= lim; ++ns) {
        const int64_t blocks = ublocks * ns, rounds = (blocks + cus - 1) / cus;
        const double fill = (double)blocks / (double)(rounds * cus);
        if (fill > best + 1e-9) { best = fill; nsplit = (int)ns; }
        if (fill >= 0.9) break;
    }
    if (h->opt_variant >= 100) {   // test hook: force the number of dish-range splits
        nsplit = h->opt_variant - 100;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > max_splits) nsplit = max_splits;
    }
    if (nsplit > 64) nsplit &= ~63;    // two-pass merge: whole groups of 64
    return nsplit;
}

template <int E8, int WAVES, int KR, bool BF16X3, bool HV = false, bool PAD = false>
int launch_grouped(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *final_s, int32_t *final_i,
                   hipStream_t st)
{
    static_assert(!HV || BF16X3, "the ingredient form exists for the pipelined split-bf16 kernel only");
    static_assert(!PAD || !BF16X3, "zero-padded rows are served by the exact-f32 kernel");
    constexpr int E = E8 * 8;
    const bool pipe = HV || h->opt_topk_form != 1;           // "topk_form", split-bf16 kernels: see below
    // Blocks of 128 users (four waves, half-size stages, two blocks per CU) for pruned launches of the pipelined kernel at
    // E = 64: a stage barrier holds up four waves instead of eight, the CU's other block runs meanwhile, and 128 users share
    // fewer patterns than 256 (tiles stepped through 0.109 -> 0.100 of the catalogue).  Measured, k = 10: 65 536 users x 100 k
    // dishes 0.629 -> 0.584 ms, 262 144 users 2.10 -> 1.99 ms, 16 384 users 0.384 -> 0.376 ms; but every block streams its own
    // copy of the tiles, and once the catalogue image (8 KiB a tile) no longer sits in the Infinity Cache that costs more than
    // the barriers did -- 1 M dishes: 65 536 users 3.19 -> 3.24 ms, 262 144 users 10.7 -> 11.4 ms -- so: catalogues up to 8 192
    // tiles (64 MiB of image).  "topk_block" = 128 / 256 forces either.
    const bool half_ok = BF16X3 && !HV && E == 64 && pipe;
    // (k > 10 with the left-out bookkeeping: 232 registers since the launch bounds say two waves per SIMD -- it took 261 and one
    //  wave per SIMD before, and the launcher kept eight-wave blocks for it: 65 536 users x 100 k dishes, k = 16: 0.764 -> 0.666 ms)
    const bool half = half_ok && (h->opt_topk_block == 128 || (h->opt_topk_block == 0 && M2D_TOPK_HALF_BLOCKS && h->opt_topk_prune != 0 &&
                                                              h->opt_variant < 100 && nU >= 16384 && h->grp_tiles <= 8192));
    const int WV = half ? 4 : WAVES;                         // waves per block
    const int TPS = grouped_tiles_per_stage(E) * WV / WAVES;
    const size_t lds = (size_t)2 * TPS * 32 * E * sizeof(float);
    h->topk_block_users = 32 * WV;
    GroupedArgs a;
    a.pm = h->pm; a.ce = h->ce; a.rs = h->grp_rs; a.rs16 = reinterpret_cast<const __bf16 *>(h->grp_rs16);
    a.perm = h->grp_perm; a.tile_info = h->grp_tile_info;
    a.grp = h->grp_work + (size_t)((h->I + 255) / 256) * GRP_KEYS;
    a.users = users; a.nU = nU; a.U = h->U; a.user_base = h->user_base; a.k = k; a.tiles = h->grp_tiles;
    a.a = h->a; a.b = h->b; a.err = h->err_dev; a.dbg = g_m2d_diag_buffer; a.e_real = h->E;
    a.plan = nullptr; a.order = nullptr; a.tiles_scanned = nullptr; a.items = nullptr; a.shared_thr = nullptr; a.ex_out = nullptr;
    const int64_t ublocks = (nU + 32 * WV - 1) / (32 * WV);
    int nsplit = pick_splits(h, ublocks, a.tiles, 2 * TPS, 512);
    if ((!BF16X3 || (!HV && h->opt_topk_form != 1)) && h->opt_topk_prune != 0 && h->opt_variant < 100) {
        // Pattern pruning makes the blocks unequal -- a block of users with one relevant pattern steps through a fifteenth of
        // the catalogue, one whose users need most patterns through all of it -- so a launch with many user blocks is cut into
        // dish ranges and the (block, range) items are handed out longest first (m2d_plan_items_*).  The longest item bounds
        // the launch, every piece starts its lists from the scan-start bound again (more pieces re-insert more): measured
        // best at 100 k dishes, E = 64 -- 16 / 32 user blocks: 32 ranges (0.30 / 0.33 ms; 8 ranges 0.56), 64 blocks: 16
        // (0.42 ms; 8: 0.59, 32: 0.49), 128 blocks: 12 (0.58 ms; 8: 0.62, 24: 0.68), 256 blocks: 8 (0.85 ms; 16: 1.0).
        // 8 blocks: 64 (0.21 ms; 32: 0.25).  A handful of blocks (serving): most ranges hold no tile of the users' patterns and
        // return at once, the others are short -- 1 user: 512 ranges 0.056 ms (192, the unpruned launch's count: 0.066), 32
        // users 0.068 (0.091), 256 users: 256 ranges 0.101 (0.113), 1 024 users: 128 ranges 0.167 (0.185).
        if (ublocks >= 6) {
            nsplit = ublocks >= 192 ? 8 : (ublocks >= 96 ? 12 : (ublocks >= 48 ? 16 : (ublocks >= 24 ? 32 : (int)(512 / ublocks))));
            // long catalogues: since a user's ranges share their thresholds, twice the ranges cost little and balance better
            // (65 536 users x 1 M dishes: 8 ranges 3.41 ms, 16 ranges 3.19 ms, 24: 3.32; 262 144 users: 8 ranges 10.7, 16: 11.2)
            if (BF16X3 && E == 64 && a.tiles >= 16384 && ublocks >= 192 && ublocks < 768) nsplit = 16;
            // blocks of 128 users, many of them: about 4 096 items is what balances (8 rounds of the 512 block slots); more only
            // adds item prologues, partial stages and merge work -- 131 072 users x 100 k dishes: 4 ranges 0.87 ms (3: 0.95, 6:
            // 0.90, 8: 0.95); 262 144 users: 2 ranges 1.59 (1: 2.11, 3: 1.65, 4: 1.66, 8: 1.85); 524 288 users: 1 range 2.86
            // (2: 2.99, 4: 3.24, 8: 3.67).  (Blocks of 256 users over 1 M dishes: 8 stays -- 262 144 users 10.2 ms against 10.6
            // with 3 ... 6; 524 288 users 19.3 ... 20.2 for 2 ... 8, within the noise.)
            if (half && ublocks >= 768) nsplit = ublocks >= 3072 ? 1 : (ublocks >= 1536 ? 2 : 4);
            const int64_t most = a.tiles / (4 * TPS);        // at least four stages per range
            if (most < nsplit) nsplit = most > 1 ? (int)most : 1;
        } else {
            nsplit = ublocks == 1 ? (nU <= 64 ? 512 : 256) : (int)(512 / ublocks);
            const int64_t most = a.tiles / 4;                // at least four tiles per range
            if (most < nsplit) nsplit = most > 1 ? (int)most : 1;
        }
        if (nsplit > 64) nsplit &= ~63;                      // two-pass merge: whole groups of 64 (6 or 7 blocks: 85 / 73 -> 64)
    }
    a.nsplit = nsplit;
    // tie values (floats): per (user, split), per (user, group of 64 splits) when the merge takes two passes, per user;
    // behind them the repair list (int32: count, users) and the repair's partial lists
    const size_t tie_vals = (size_t)nU * (nsplit > 1 ? nsplit + (nsplit > 64 ? nsplit / 64 : 0) + 1 : 1);
    const size_t tie_need = tie_vals + 1 + (size_t)nU + (size_t)2 * REPAIR_CAP * REPAIR_SPLITS * k;
    if (h->topk_flags_cap < tie_need) {
        if (h->topk_flags) M2D_HIP_TRY(h, hipFree(h->topk_flags));
        h->topk_flags = nullptr; h->topk_flags_cap = 0;
        M2D_HIP_TRY(h, hipMalloc((void **)&h->topk_flags, tie_need * sizeof(float)));
        h->topk_flags_cap = tie_need;
    }
    float *tie_final = h->topk_flags + (nsplit > 1 ? (size_t)nU * (nsplit + (nsplit > 64 ? nsplit / 64 : 0)) : 0);
    a.tie_val = h->topk_flags;
    int32_t *tie_list = reinterpret_cast<int32_t *>(h->topk_flags + tie_vals);
    const bool planned = !BF16X3 || pipe;                    // (the first-form bf16 kernel takes no plan)
    if (!planned) M2D_HIP_TRY(h, hipMemsetAsync(tie_list, 0, sizeof(int32_t), st));
    h->topk_tie_final = tie_final;
    h->topk_tie_list = tie_list;
    h->topk_flags_used = nU;
    // what the lists leave out, for m2d_topk_refine (kernels that keep it: see EXT in the scan kernels)
    const bool ext = planned && h->opt_topk_refine != 0 && !HV && !PAD &&
                     (BF16X3 ? (pipe && !(E == 128 && KR == 16)) : E8 <= 16);
    float *ex_final = nullptr;
    if (ext) {
        const size_t ex_need = ((size_t)nU * (nsplit > 1 ? nsplit + (nsplit > 64 ? nsplit / 64 : 0) + 1 : 1)) * 8 + (size_t)nU + 8;
        if (h->topk_ex_cap < ex_need) {
            if (h->topk_ex) M2D_HIP_TRY(h, hipFree(h->topk_ex));
            h->topk_ex = nullptr; h->topk_ex_cap = 0;
            M2D_HIP_TRY(h, hipMalloc((void **)&h->topk_ex, ex_need * sizeof(float)));
            h->topk_ex_cap = ex_need;
        }
        a.ex_out = h->topk_ex;
        ex_final = h->topk_ex + (nsplit > 1 ? (size_t)nU * (nsplit + (nsplit > 64 ? nsplit / 64 : 0)) * 8 : 0);
        h->topk_refine_counter = reinterpret_cast<int32_t *>(h->topk_ex + ex_need - (size_t)nU - 8);     // [0] refined [1] sent to the repair; [8 + u] user u's word
                                                                                                       // (zeroed by the plan kernel: no memset launch)
    } else {
        h->topk_refine_counter = nullptr;
    }
    const size_t tmp_entries = nsplit > 64 ? (size_t)nU * (nsplit / 64) * k : 0;
    if (nsplit > 1) {
        const size_t need = ((size_t)nU * nsplit * k + tmp_entries) * 8 + 256;
        if (h->scratch_bytes < need) {
            if (h->scratch) M2D_HIP_TRY(h, hipFree(h->scratch));
            h->scratch = nullptr; h->scratch_bytes = 0;
            M2D_HIP_TRY(h, hipMalloc((void **)&h->scratch, need));
            h->scratch_bytes = need;
        }
        a.out_scores = h->scratch;
        a.out_ids = reinterpret_cast<int32_t *>(h->scratch + (size_t)nU * nsplit * k);
    } else {
        a.out_scores = final_s;
        a.out_ids = final_i;
    }
    float *tmp_s = h->scratch ? h->scratch + (size_t)2 * nU * nsplit * k : nullptr;
    int32_t *tmp_i = reinterpret_cast<int32_t *>(tmp_s ? tmp_s + tmp_entries : nullptr);
    // "topk_form" (split-bf16 kernels): 0 or 2 = pipelined form (E = 64: 2.55 ms against 3.3 at 100 k dishes; E = 128:
    // 38.3 ms against 45.4 at 1 M dishes), 1 = first form (kept as the A/B reference; it takes no plan)
    if (planned) {
        // the call's plan: per user the scan-start bound, <U_high, CE_c> and the mask of patterns that can reach the
        // top-k; users sorted by mask so that a block's 256 users share their patterns (a single block: no sort)
        const size_t nitems = (size_t)ublocks * nsplit;
        const size_t need = (size_t)nU * 8 + (size_t)nU + PLAN_KEYS + 8 + 2 * nitems;
        if (h->topk_plan_cap < need) {
            if (h->topk_plan) M2D_HIP_TRY(h, hipFree(h->topk_plan));
            h->topk_plan = nullptr; h->topk_plan_cap = 0;
            M2D_HIP_TRY(h, hipMalloc((void **)&h->topk_plan, need * sizeof(float)));
            h->topk_plan_cap = need;
        }
        float *plan = h->topk_plan;
        int32_t *order = reinterpret_cast<int32_t *>(plan + (size_t)nU * 8), *hist = order + ((nU + 3) & ~(int64_t)3);      // hist: 16-B aligned
        unsigned long long *counter = reinterpret_cast<unsigned long long *>(hist + PLAN_KEYS);
        const bool prune = h->opt_topk_prune != 0;
        const bool sorted = prune && !HV && nU > 32 * WV && h->opt_topk_prune != 3;      // 3: pruning without the sort (A/B)
        {
            const int pmode = (HV || !prune) ? 1 : (h->opt_topk_prune == 2 ? 2 : (h->opt_topk_prune == 4 ? 4 : 0));
            const float *probes = (HV || !prune || h->opt_topk_prune == 6) ? nullptr : h->grp_rs;      // 6: Cauchy-Schwarz bounds only (A/B)
            const dim3 pgrid((unsigned)((nU * 16 + 255) / 256));
            // probe rows per user: each costs a row read per user (16: +18 us for 65 536 users) and buys a tighter bound -- 16 rows
            // at 100 k dishes (0.71 ms; 32: 0.73), 32 at 1 M (3.83 ms; 16: 4.02)
            const int nprobe = h->opt_topk_probes ? h->opt_topk_probes : (a.tiles < 8192 ? 16 : (a.tiles < 65536 ? 32 : PLAN_PROBES));
            auto pk = h->E <= 64 ? m2d_topk_user_plan<1> : (h->E <= 128 ? m2d_topk_user_plan<2> : m2d_topk_user_plan<4>);
            hipLaunchKernelGGL(pk, pgrid, dim3(256), 0, st, h->pm, h->ce, users, nU, h->U, h->user_base, h->E, a.grp, (int)k, h->a, h->b, pmode,
                               plan, tie_list, counter, sorted ? hist : nullptr, PLAN_KEYS, probes, h->grp_ew, nprobe, BF16X3 ? 0 : 1, ext ? h->topk_refine_counter : nullptr);
        }
        a.plan = plan;
        // dish ranges of a user share their thresholds (pipelined kernel; "topk_prune" = 7 keeps them apart: A/B)
        if (BF16X3 && pipe && !HV && E == 64 && nsplit > 1 && h->opt_topk_prune != 7) a.shared_thr = reinterpret_cast<int32_t *>(plan) + 6;
        if (sorted) {
            const size_t tab = (size_t)PLAN_KEYS * sizeof(int32_t);
            const unsigned sblocks = (unsigned)((nU + 1023) / 1024 < 4 * h->num_cu ? (nU + 1023) / 1024 : 4 * h->num_cu);
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_plan_hist, (int)tab));
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_plan_scatter, (int)tab));
            hipLaunchKernelGGL(m2d_plan_hist, dim3(sblocks), dim3(1024), tab, st, plan, nU, hist);
            hipLaunchKernelGGL(m2d_plan_scan, dim3(1), dim3(1024), 0, st, hist);
            hipLaunchKernelGGL(m2d_plan_scatter, dim3(sblocks), dim3(1024), tab, st, plan, nU, hist, order);
            a.order = order;
        }
        if (a.order && nitems > (size_t)h->num_cu && h->opt_topk_prune != 5) {      // 5: grid order (A/B)
            int32_t *work = reinterpret_cast<int32_t *>(counter + 1), *items = work + nitems;
            hipLaunchKernelGGL(m2d_plan_items_work, dim3((unsigned)((ublocks + 3) / 4)), dim3(256), 0, st, plan, order, nU, a.grp, a.tiles,
                               nsplit, work, 32 * WV);
            hipLaunchKernelGGL(m2d_plan_items_sort, dim3(1), dim3(1024), 0, st, work, (int64_t)nitems, a.tiles, nsplit, items);
            a.items = items;
        }
        a.tiles_scanned = counter;
        h->topk_tiles_counter = counter;
        h->topk_tiles_full = (int64_t)ublocks * a.tiles;
        M2D_HIP_TRY(h, hipGetLastError());
    }
    const dim3 grid = a.items ? dim3((unsigned)(ublocks * nsplit)) : dim3((unsigned)ublocks, (unsigned)nsplit);
    if constexpr (BF16X3) {
        if constexpr (HV) {
            auto kern = m2d_topk_grouped_bf16_pipe2<E, KR, 1, true>;
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);
        } else if (!pipe) {
            auto kern = m2d_topk_grouped_bf16<E, WAVES, KR>;
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
            hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, a);
        } else {
            static_assert(WAVES == 8, "the pipelined kernel is written for 256 users per block");
            constexpr bool CAN_KEEP = !(E == 128 && KR == 16);
            if constexpr (E == 64) {
                if (half) {
                    auto kern4 = a.ex_out ? m2d_topk_grouped_bf16_pipe2<E, KR, 1, false, 4, true> : m2d_topk_grouped_bf16_pipe2<E, KR, 1, false, 4>;
                    M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern4, (int)lds));
                    hipLaunchKernelGGL(kern4, grid, dim3(256), lds, st, a);
                }
            }
            if (!half) {
                auto kern = (CAN_KEEP && a.ex_out) ? m2d_topk_grouped_bf16_pipe2<E, KR, 1, false, 8, CAN_KEEP> : m2d_topk_grouped_bf16_pipe2<E, KR, 1>;
                M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
                hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);
            }
        }
    } else {
        auto kern = m2d_topk_grouped<E8, WAVES, KR, PAD>;
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, st, a);
    }
    M2D_HIP_TRY(h, hipGetLastError());
    if (nsplit > 1) {
        m2d_launch_merge_splits2(a.out_scores, a.out_ids, nU, nsplit, k, tmp_s, tmp_i, final_s, final_i, st, h->topk_flags, tie_final, tie_list,
                                 h->I, a.ex_out, ex_final, ext ? a.plan : nullptr, ext ? h->topk_refine_counter : nullptr);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    {   // users whose final k-th score is tied with a score left out: re-ranked in dish-id order (none is the common case)
        RepairArgs r;
        r.pm = h->pm; r.re = h->re; r.ce = h->ce; r.cats = h->dish_cats; r.hv = HV ? h->dish_high : nullptr;
        r.users = users; r.tie_list = tie_list; r.nU = nU; r.U = h->U; r.I = h->I; r.user_base = h->user_base;
        r.C = h->C; r.E = h->E; r.k = k; r.a = h->a; r.b = h->b; r.out_scores = final_s; r.out_ids = final_i;
        r.rows = h->grp_rs; r.perm = h->grp_perm; r.grp = a.grp; r.ew = h->grp_ew;
        r.plan = a.plan;
        r.all_patterns = h->opt_topk_prune == 9 ? 1 : 0;                     // "topk_prune" = 9: the repair reads every pattern (A/B)
        r.cap = h->opt_variant == 13 ? 2 : REPAIR_CAP;      // test hook: send all but two listed users to the one-block-per-user kernel
        r.part_s = h->topk_flags + tie_vals + 1 + (size_t)nU;
        r.part_i = reinterpret_cast<int32_t *>(r.part_s + (size_t)REPAIR_CAP * REPAIR_SPLITS * k);
        const int ub = (!HV && h->E <= 128) ? 4 : 2;                  // listed users per pass of the repair scan (LDS: 21 E + 128 k floats each)
        const size_t slds = (size_t)ub * ((size_t)(h->C + 1 + 16) * h->E + (size_t)2 * 64 * k) * sizeof(float);
        const size_t rlds = ((size_t)(h->C + 1 + 16) * h->E + (size_t)2 * 16 * k) * sizeof(float);
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_topk_repair_finish<HV>, (int)rlds));
        if (nsplit == 1)                                     // (with dish ranges the last merge pass has listed the tied users)
            hipLaunchKernelGGL(m2d_topk_tie_compact, dim3((unsigned)((nU + 255) / 256)), dim3(256), 0, st, tie_final, nU, tie_list, final_s,
                               final_i, (int)k, h->I, ext ? 1 : 0);
        if (ext) {                                           // near-tied lists: finished in the repair's arithmetic (may add to the repair's list)
            RefineArgs f;
            f.pm = h->pm; f.re = h->re; f.ce = h->ce; f.cats = h->dish_cats; f.plan = a.plan; f.tie_final = tie_final; f.ex = ex_final;
            f.users = users; f.tie_list = tie_list; f.counter = h->topk_refine_counter; f.nU = nU; f.U = h->U; f.I = h->I;
            f.user_base = h->user_base; f.E = h->E; f.k = k; f.a = h->a; f.b = h->b; f.out_scores = final_s; f.out_ids = final_i;
            if (nsplit == 1)                                 // (with dish ranges the last merge pass has listed the near-tied users)
                hipLaunchKernelGGL(m2d_topk_refine_flag, dim3((unsigned)((nU + 255) / 256)), dim3(256), 0, st, f);
            if (h->E <= 64) hipLaunchKernelGGL(m2d_topk_refine<1>, dim3((unsigned)((nU + 63) / 64)), dim3(256), 0, st, f);
            else hipLaunchKernelGGL(m2d_topk_refine<2>, dim3((unsigned)((nU + 63) / 64)), dim3(256), 0, st, f);
        }
        if (ub == 4) {
            auto rk = m2d_topk_repair_scan<4, HV>;
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)rk, (int)slds));
            hipLaunchKernelGGL(rk, dim3(REPAIR_SPLITS, 8), dim3(1024), slds, st, r);
        } else {
            auto rk = m2d_topk_repair_scan<2, HV>;
            M2D_HIP_TRY(h, m2d_lds_limit((const void *)rk, (int)slds));
            hipLaunchKernelGGL(rk, dim3(REPAIR_SPLITS, 8), dim3(1024), slds, st, r);
        }
        hipLaunchKernelGGL(m2d_topk_repair_finish<HV>, dim3((unsigned)(h->num_cu * 2)), dim3(256), rlds, st, r);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    h->last_kernel = BF16X3 ? "m2d_topk_grouped_bf16x3" : "m2d_topk_grouped";      // both bf16 forms report this name
    return M2D_OK;
}

template <int NB, int WAVES, int KR>
int launch_mfma(m2d_engine *h, TopkArgs &a, float *final_s, int32_t *final_i, hipStream_t st)
{
    constexpr int KC8 = NB < 40 ? NB : 40;
    const int kl = KR > 0 ? KR : a.k;
    const size_t lds = (size_t)2 * 32 * (KC8 * 2) * 4 * sizeof(float) + (size_t)WAVES * 2 * kl * 64 * sizeof(float);
    const int64_t ublocks = (a.nU + 32 * WAVES - 1) / (32 * WAVES);
    // split the dish range when there are too few user blocks to fill the chip
    int nsplit = 1;
    const int64_t want = 2 * (int64_t)h->num_cu;
    if (ublocks < want) {
        int64_t ns = (want + ublocks - 1) / ublocks;
        const int64_t cap = a.tiles / 8 > 1 ? a.tiles / 8 : 1;
        if (ns > 64) ns = 64;
        if (ns > cap) ns = cap;
        nsplit = (int)ns;
    }
    if (h->opt_variant >= 100) {   // test hook: force the number of dish-range splits
        nsplit = h->opt_variant - 100;
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 64) nsplit = 64;
    }
    a.nsplit = nsplit;
    if (nsplit > 1) {
        const size_t need = (size_t)a.nU * nsplit * a.k * 8 + 256;
        if (h->scratch_bytes < need) {
            if (h->scratch) M2D_HIP_TRY(h, hipFree(h->scratch));
            h->scratch = nullptr; h->scratch_bytes = 0;
            M2D_HIP_TRY(h, hipMalloc((void **)&h->scratch, need));
            h->scratch_bytes = need;
        }
        a.out_scores = h->scratch;
        a.out_ids = reinterpret_cast<int32_t *>(h->scratch + (size_t)a.nU * nsplit * a.k);
    } else {
        a.out_scores = final_s;
        a.out_ids = final_i;
    }
    h->topk_tie_list = nullptr;                             // (no tie repair on this path: "topk_repaired" answers 0)
    auto kern = m2d_topk_mfma<NB, WAVES, KR>;
    M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)ublocks, (unsigned)nsplit), dim3(WAVES * 64), lds, st, a);
    M2D_HIP_TRY(h, hipGetLastError());
    if (nsplit > 1) {
        m2d_launch_merge_splits(a.out_scores, a.out_ids, a.nU, nsplit, a.k, final_s, final_i, st);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(m2d_topk_fill_absent, dim3((unsigned)((a.nU + 127) / 128)), dim3(128), 0, st, final_s, final_i,
                       a.nU, a.k, a.I);
    M2D_HIP_TRY(h, hipGetLastError());
    h->last_kernel = "m2d_topk_mfma";
    return M2D_OK;
}

}  // namespace

int m2d_launch_topk_users(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores,
                          int32_t *out_ids, hipStream_t stream)
{
    int rc;
    if ((rc = m2d_ensure_finite_scan(h, stream)) != M2D_OK) return rc;
    // 0/1 category masks, no ingredient table: contraction over E after sorting dishes by mask pattern
    // ("topk_grouped" = 0 keeps the dense kernel: dishes then arrive in id order whatever their masks, so exactly tied
    // scores of dishes with DIFFERENT mask patterns also resolve to the lower id -- see include/m2d.h)
    // ingredient extension: rows [H[d] | RE[d]] of width 2 E on the pipelined split-bf16 kernel (E = 32 / 64)
    const bool hv_ok = h->dish_high && h->opt_topk_bf16x3 != 0 && (h->E == 32 || h->E == 64);
    const int roww = grouped_row_width(h->E);
    const bool padded = !(h->E == 32 || h->E == 64 || h->E == 128);   // e.g. the reference's embed_size 200: rows padded to 256
    if (h->C == 4 && (!h->dish_high || (hv_ok && !padded)) && k <= 16 && roww != 0 &&
        h->opt_topk_grouped != 0 && h->opt_variant != 7 && h->opt_variant != 8 && h->opt_variant != 9) {
        if ((rc = ensure_grouped(h, stream)) != M2D_OK) return rc;
        if ((rc = refresh_grouped_nonfinite(h, stream)) != M2D_OK) return rc;
        if (hv_ok && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite) {
            if (h->E == 32)
                return k <= 10 ? launch_grouped<8, 8, 10, true, true>(h, users, nU, k, out_scores, out_ids, stream)
                               : launch_grouped<8, 8, 16, true, true>(h, users, nU, k, out_scores, out_ids, stream);
            return k <= 10 ? launch_grouped<16, 8, 10, true, true>(h, users, nU, k, out_scores, out_ids, stream)
                           : launch_grouped<16, 8, 16, true, true>(h, users, nU, k, out_scores, out_ids, stream);
        }
        if (!h->dish_high && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite) {
            // "topk_bf16x3" option: 1 = split-bf16 MFMA (E = 64 / 128), 0 = exact-f32 MFMA
            const bool x3 = h->opt_topk_bf16x3 != 0 && (h->E == 64 || h->E == 128);
#define M2D_GRP(EV, X3)                                                                                       \
    if (h->E == EV && x3 == X3)                                                                               \
        return k <= 10 ? launch_grouped<EV / 8, 8, 10, X3>(h, users, nU, k, out_scores, out_ids, stream)         \
                       : launch_grouped<EV / 8, 8, 16, X3>(h, users, nU, k, out_scores, out_ids, stream);
            M2D_GRP(32, false) M2D_GRP(64, false) M2D_GRP(128, false) M2D_GRP(64, true) M2D_GRP(128, true)
#undef M2D_GRP
#define M2D_GRP_PAD(EV)                                                                                                   \
    if (padded && roww == EV)                                                                                             \
        return k <= 10 ? launch_grouped<EV / 8, 8, 10, false, false, true>(h, users, nU, k, out_scores, out_ids, stream)  \
                       : launch_grouped<EV / 8, 8, 16, false, false, true>(h, users, nU, k, out_scores, out_ids, stream);
            M2D_GRP_PAD(32) M2D_GRP_PAD(64) M2D_GRP_PAD(128) M2D_GRP_PAD(256)
#undef M2D_GRP_PAD
        }
    }
    rc = m2d_ensure_dish_vectors(h, stream);
    if (rc != M2D_OK) return rc;
    const int K = (h->C + 1) * h->E;
    TopkArgs a;
    a.pm = h->pm; a.dt = h->dish_vec; a.users = users; a.nU = nU; a.U = h->U; a.I = h->I;
    a.user_base = h->user_base; a.k = k; a.nsplit = 1; a.tiles = (h->I + 31) / 32;
    a.out_scores = out_scores; a.out_ids = out_ids; a.err = h->err_dev;
    a.dbg = g_m2d_diag_buffer;
    const bool force_generic = h->opt_variant == 9;
    if (!force_generic && K % 8 == 0) {
        const int NB = K / 8;
        // list space: 8 waves up to k = 16, 2 waves beyond (LDS: 2 stages + waves * k * 512 B)
        // k <= 16: register-resident lists (10 or 16 slots); beyond: LDS lists, 2 waves (LDS: 2 stages + waves*k*512 B)
        const bool lds_lists = k > 16 || h->opt_variant == 8;
#define M2D_TOPK(NBV, WV)                                                                              \
    if (NB == NBV) {                                                                                  \
        if (lds_lists) return launch_mfma<NBV, 2, 0>(h, a, out_scores, out_ids, stream);               \
        return k <= 10 ? launch_mfma<NBV, WV, 10>(h, a, out_scores, out_ids, stream)                   \
                       : launch_mfma<NBV, WV, 16>(h, a, out_scores, out_ids, stream);                  \
    }
        M2D_TOPK(20, 8) M2D_TOPK(40, 8) M2D_TOPK(80, 4)
#undef M2D_TOPK
    }
    const size_t lds = (size_t)4 * 2 * k * sizeof(float);
    hipLaunchKernelGGL(m2d_topk_generic, dim3((unsigned)nU), dim3(256), lds, stream, a, K);
    M2D_HIP_TRY(h, hipGetLastError());
    h->last_kernel = "m2d_topk_generic";
    return M2D_OK;
}
