// The tie repair of pattern-grouped retrieval: users whose final k-th score is tied with three or more dishes left out (copies of
// dishes, all-zero users; with "topk_refine" = 0 every user tied at its list's end) are re-ranked over their relevant patterns.
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

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

}  // namespace

// r.cap, r.part_s / part_i and the lists are the caller's (launch_grouped); hv: the ingredient table's rows
int m2d_topk_launch_repair(m2d_engine *h, const RepairArgs &r, bool hv, hipStream_t st)
{
    const int ub = (!hv && r.E <= 128) ? 4 : 2;                  // listed users per pass of the repair scan (LDS: 21 E + 128 k floats each)
    const size_t slds = (size_t)ub * ((size_t)(r.C + 1 + 16) * r.E + (size_t)2 * 64 * r.k) * sizeof(float);
    const size_t rlds = ((size_t)(r.C + 1 + 16) * r.E + (size_t)2 * 16 * r.k) * sizeof(float);
#define M2D_REPAIR(UBV, HVV)                                                                          \
    if (ub == UBV && hv == HVV) {                                                                     \
        auto rk = m2d_topk_repair_scan<UBV, HVV>;                                                     \
        auto fk = m2d_topk_repair_finish<HVV>;                                                        \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)fk, (int)rlds));                                   \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)rk, (int)slds));                                   \
        hipLaunchKernelGGL(rk, dim3(REPAIR_SPLITS, 8), dim3(1024), slds, st, r);                      \
        hipLaunchKernelGGL(fk, dim3((unsigned)(h->num_cu * 2)), dim3(256), rlds, st, r);              \
    }
    M2D_REPAIR(4, false) M2D_REPAIR(2, false) M2D_REPAIR(2, true)
#undef M2D_REPAIR
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}
