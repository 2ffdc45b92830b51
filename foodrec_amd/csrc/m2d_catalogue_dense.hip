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
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

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
    int slot_lo[G / 2];                          // float offset of the lane's low slot bits, by T mod G / 2 (see the MFMA loop)
#pragma unroll
    for (int t = 0; t < G / 2; ++t) slot_lo[t] = (((2 * t + h) ^ (j & (G - 1))) & (G - 1)) * 4;

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
                // Fragment T of the lane's row sits in slot (2 T + h) ^ key of the swizzled image: the XOR touches the slot's low bits
                // only, so the lane-dependent part takes G / 2 values (by T mod G / 2) and the rest of the offset is a constant that
                // rides in the read's immediate.  Written as one expression per T, all 40 lane offsets were hoisted out of the stage
                // loop as loop invariants -- 40 VGPRs beside the 160 of the user operand: 33 spilled at k = 16, 7 at k = 10.
#pragma unroll
                for (int T = 0; T < KC8; ++T) {
                    const v4f av = *reinterpret_cast<const v4f *>(img + slot_lo[T % (G / 2)] + ((2 * T) & ~(G - 1)) * 4);
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
                        // (in place, from the last slot up: the out-of-place form's second copy of the list -- 2 KR registers beside
                        //  the 160 of the user operand -- was what spilled at NB = 40: 33 VGPRs at k = 16, 7 at k = 10)
                        sorted_insert_inplace<KR>(rs, ri, v, (int32_t)base + (r & 3) + 8 * (r >> 2));
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
    h->topk_tie_list = nullptr;                             // (no tie repair, refinement or pruning on this path: "topk_repaired",
    h->topk_refine_counter = nullptr;                       //  "topk_refined" and "topk_tiles_scanned" answer 0, not the last
    h->topk_tiles_counter = nullptr; h->topk_tiles_full = 0;      //  pattern-grouped call's numbers)
    auto kern = m2d_topk_mfma<NB, WAVES, KR>;
    M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)ublocks, (unsigned)nsplit), dim3(WAVES * 64), lds, st, a);
    M2D_HIP_TRY(h, hipGetLastError());
    if (nsplit > 1) {
        m2d_launch_merge_splits(a.out_scores, a.out_ids, a.nU, nsplit, a.k, final_s, final_i, st);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    m2d_topk_launch_fill_absent(final_s, final_i, a.nU, a.k, a.I, st);
    M2D_HIP_TRY(h, hipGetLastError());
    h->last_kernel = "m2d_topk_mfma";
    return M2D_OK;
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
    ++h->dish_vec_gen;
    return M2D_OK;
}

// Dispatch of the kernels of this file: weighted masks, k > 16, category counts other than 4, embedding sizes without a
// pattern-grouped kernel -- and 0/1 masks under option "topk_grouped" = 0.
int m2d_topk_dense_launch(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores, int32_t *out_ids,
                          hipStream_t stream)
{
    int rc = m2d_ensure_dish_vectors(h, stream);
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
        const bool lds_lists = k > 16;
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
    h->topk_tie_list = nullptr; h->topk_refine_counter = nullptr; h->topk_tiles_counter = nullptr; h->topk_tiles_full = 0;
    hipLaunchKernelGGL(m2d_topk_generic, dim3((unsigned)nU), dim3(256), lds, stream, a, K);
    M2D_HIP_TRY(h, hipGetLastError());
    h->last_kernel = "m2d_topk_generic";
    return M2D_OK;
}
