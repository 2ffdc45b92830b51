// Pattern-grouped scan on split-bf16 MFMA (the default for 0/1 masks at E = 64 / 128, and E = 32 / 64 with the ingredient table):
// the first form (m2d_topk_grouped_bf16, kept as the A/B reference) and the pipelined form that is launched
// (m2d_topk_grouped_bf16_pipe2).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

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
#if M2D_DIAG & 256
// (timing only) quarter 2 (ks & 1) + Q of a 32 x 32 accumulator as the C / D of one v_mfma_f32_16x16x32_bf16
template <int Q>
__device__ __forceinline__ void diag_mfma16(v16f &acc, const int ks, const bf16x8 a, const bf16x8 b, const bool from_zero)
{
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    if (ks & 1) {
        v4f t = {acc[8 + 4 * Q], acc[9 + 4 * Q], acc[10 + 4 * Q], acc[11 + 4 * Q]};
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, from_zero ? z : t, 0, 0, 0);
        acc[8 + 4 * Q] = t.x; acc[9 + 4 * Q] = t.y; acc[10 + 4 * Q] = t.z; acc[11 + 4 * Q] = t.w;
    } else {
        v4f t = {acc[4 * Q], acc[1 + 4 * Q], acc[2 + 4 * Q], acc[3 + 4 * Q]};
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, from_zero ? z : t, 0, 0, 0);
        acc[4 * Q] = t.x; acc[1 + 4 * Q] = t.y; acc[2 + 4 * Q] = t.z; acc[3 + 4 * Q] = t.w;
    }
}
#endif

// APX = true (blocks of 256 users; launch_grouped takes it for the large catalogues, where a tile with a candidate is the
// exception): the body multiplies only the hi x hi product of a tile -- a third of the matrix work, half the LDS
// reads -- and compares against the threshold LESS a bound of what the two cross products can add (eps = 2^-7 * 1.02 |w_P[u]| *
// the pattern's largest row norm: |x - hi(x)| <= 2^-8 |x| for both operands, Cauchy-Schwarz); a tile that still has a candidate
// gets its cross products then, from its rows still in LDS, and is handled from exact scores as before.  A score is the
// hi x hi sums of all k-steps with the cross products added behind them: not the bits of the three-product kernels, so the launcher's
// choice depends on the catalogue alone -- every launch shape of one problem takes the same arithmetic (lists bit for bit).
template <int E, int KR, int G, bool HV = false, int WAVES = 8 / G, bool KEEP = false, bool APX = false>
__global__ __launch_bounds__(WAVES * 64, 2) void m2d_topk_grouped_bf16_pipe2(GroupedArgs p)
{
    static_assert(!APX || ((E == 64 || E == 128) && WAVES == 8 && G == 1), "the hi x hi first form: blocks of eight waves");
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
    __shared__ __align__(16) uint32_t s_prog[8];           // (APX) stages whose step "sub 1" wave w has finished, see issue of pieces 6 and 7
    if (threadIdx.x == 0) s_umask = 0u;
    if (APX && threadIdx.x < 8) s_prog[threadIdx.x] = 0u;
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
    float eps[G], eps_prev[G];                             // (APX) what the cross products can add to a score of that tile, at most
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
        eps[g] = eps_prev[g] = 0.f;
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
    int first_pat = 0;                                     // the first pattern of the scan that the block needs (its operands: see the prologue)
    {
        int last_end = -1;
        for (int q = 1; q < GRP_MAXPAT; ++q) {
            const int rows = __builtin_amdgcn_readlane(g_rows, q);
            if (rows == 0 || !((umask >> q) & 1u)) continue;
            const int64_t gt0 = __builtin_amdgcn_readlane(g_first, q), gt1 = gt0 + ((rows + 31) >> 5);
            const int64_t lo = gt0 > t_begin ? gt0 : t_begin, hi = gt1 < t_end ? gt1 : t_end;
            if (lo >= hi) continue;
            if (first_pat == 0) first_pat = q;
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
    const int n = vstages * TPS;                                    // tiles the steps below go through ("virtual" tiles 0 .. n - 1)
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

    // a pattern's operands: alpha_P[u], w_P[u] split hi / lo, (APX) the cross products' bound -- built where a scan reaches a pattern its
    // block needs, and for the first such pattern ahead of the first stage's arrival (the rows' loads and the stage's DMA in flight
    // together: two memory round trips at the head of every (user block, dish range) item were one after the other)
    auto build_operands = [&](const int pat) __attribute__((always_inline)) {
        const float inv_n = 1.0f / (float)__builtin_popcount(pat);
        const float beta = p.b * inv_n;
        const float rmax_pat = APX ? __int_as_float(p.grp[GRP_RMAX + pat]) : 0.f;   // the pattern's largest row norm (NaN rows: +inf)
        const float rmax_re = (APX && HV) ? __int_as_float(p.grp[GRP_STAT + 2]) : 0.f;   // ingredient form: rmax_pat is of H[d], this the largest |RE[d]|
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
            float ww = 0.f, wmax = 0.f;                          // (APX) sum of squares and largest magnitude of this lane's half of w_P[u]
            float wwh = 0.f, wmaxh = 0.f;                        // (APX, ingredient form) the same of a U_high, whose k-values meet H[d]
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
                    if (APX) {
                        if (HV && ks < KS / 2) {
                            wwh = fmaf(xx[i], xx[i], wwh);
                            wmaxh = fmaxf(wmaxh, fabsf(xx[i]));
                        } else {
                            ww = fmaf(xx[i], xx[i], ww);
                            wmax = fmaxf(wmax, fabsf(xx[i]));
                        }
                    }
                }
                wh[g][ks] = vh;
                wl[g][ks] = vl;
            }
            if constexpr (APX) {
                // |x - hi| <= 2^-8 |x| for a dish value and for a w value: |sum (lo_d hi_w + hi_d lo_w)| <= 2^-7 (1 + 2^-8)
                // sum |d_k| |w_k| <= that times |d| |w|; 1.02 covers the norms' and the accumulations' own roundings.
                // |w|: the root of the sum of squares where that sum is a normal number (squares lost to underflow are then
                // far below what the 1.02 covers), else sqrt(E) times the largest |w_k| -- never less than |w|
                ww += __shfl_xor(ww, 32, 64);
                wmax = fmaxf(wmax, __shfl_xor(wmax, 32, 64));
                const float nw = (ww >= 1e-30f && ww < INFINITY) ? sqrtf(ww) : sqrtf((float)EU) * wmax;
                if constexpr (HV) {                         // two operand halves, two row halves: |a U_high| against |H[d]|, |w_P| against |RE[d]|
                    wwh += __shfl_xor(wwh, 32, 64);
                    wmaxh = fmaxf(wmaxh, __shfl_xor(wmaxh, 32, 64));
                    const float nh = (wwh >= 1e-30f && wwh < INFINITY) ? sqrtf(wwh) : sqrtf((float)EU) * wmaxh;
                    eps[g] = 1.02f * 0.0078125f * (nh * rmax_pat + nw * rmax_re);
                } else
                    eps[g] = 1.02f * 0.0078125f * nw * rmax_pat;
            }
        }
    };

    if (n > 0) {
        ps_0 = next_stage();
        ps_p1 = next_stage();
        issue_pieces(ps_0, 0, 0, PPW);
        if (first_pat > 0) {                               // (the scan's first needed pattern: its step finds the operands in place;
                                                           //  0.502 -> 0.500 ms per serving-size call)
            cur_pat = first_pat;
            build_operands(first_pat);
        }
        wait_all_vmem();
        __syncthreads();
        issue_pieces(ps_p1, 1, 0, (APX && KS > AR) ? 4 : PCNT);   // what step "0" of the first stage would have issued
#pragma unroll
        for (int ks = 0; ks < AR; ++ks) {                  // the first AR k-steps of tile 0
            const unsigned char *a = smem8 + (lane_off ^ (ks << 5));
            ah[ks] = *reinterpret_cast<const bf16x8 *>(a);
            if (!APX) al[ks] = *reinterpret_cast<const bf16x8 *>(a + 32 * ROW_BYTES);
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

    // the interleaved body: M(q-1) into accN, L(q), the max tree of tile q-2 (accP), and -- INS -- the parked insertion,
    // slot ranges from the end of the list up, one range per k-step group
    auto body = [&](auto ins_tag, v16f (&accN)[G], const v16f (&accP)[G], const int img_prev, const int img_off,
                    float (&mx)[G]) __attribute__((always_inline)) {
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
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = ks % AR;
#pragma unroll
            for (int g = 0; g < G; ++g) {
#if M2D_DIAG & 256
                // TIMING ONLY (wrong scores; scripts/diag/topk_diag.cpp): the k-step's flops as six v_mfma_f32_16x16x32_bf16 on the
                // quarters of the same accumulator registers -- what the 2 x 2 arrangement of 16 x 16 tiles over the wave's 32 users
                // x 32 dishes would issue (each A and each B fragment feeding two MFMAs: the same LDS reads, the same registers) --
                // to read the clock the chip holds on that shape before anything is built for it (MI355X_MICROARCH "DVFS give-back" 7)
                diag_mfma16<0>(accN[g], ks, al[st], wh[g][ks], ks < 2);
                diag_mfma16<0>(accN[g], ks, ah[st], wl[g][ks], false);
                diag_mfma16<0>(accN[g], ks, ah[st], wh[g][ks], false);
                diag_mfma16<1>(accN[g], ks, ah[st], wl[g][ks], ks < 2);      // (another order: identical chains would be merged)
                diag_mfma16<1>(accN[g], ks, ah[st], wh[g][ks], false);
                diag_mfma16<1>(accN[g], ks, al[st], wh[g][ks], false);
#else
                if constexpr (APX) {
                    accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st], wh[g][ks], ks == 0 ? zero16 : accN[g], 0, 0, 0);
                } else {
                    accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[st], wh[g][ks], ks == 0 ? zero16 : accN[g], 0, 0, 0);
                    accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st], wl[g][ks], accN[g], 0, 0, 0);
                    accN[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[st], wh[g][ks], accN[g], 0, 0, 0);
                }
#endif
            }
            // k-step ks + AR of this tile (q-1), or k-step ks + AR - KS of the next (q)
            const unsigned char *a = smem8 + (ks + AR < KS ? (img_prev ^ ((ks + AR) << 5)) : (img_off ^ ((ks + AR - KS) << 5)));
            ah[st] = *reinterpret_cast<const bf16x8 *>(a);
            if (!APX) al[st] = *reinterpret_cast<const bf16x8 *>(a + 32 * ROW_BYTES);
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int r = ks * RPK; r < (ks + 1) * RPK; ++r) mx[g] = fmaxf(mx[g], accP[g][r]);      // the tile's maximum per lane: one
                                                                       // compare of it against the threshold settles most tiles (below)
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

    // (APX) the cross products of tile qt, from its rows in LDS (hi and lo fragments read again), under the operands in registers --
    // so before a pattern switch rebuilds those: a step that switches completes tile q - 2 first, candidate or not (`accP_exact`).
    // Accumulated onto the tile's hi x hi sums, k-step by k-step: one order for every tile, the same bits wherever it sits in a launch.
    bool accP_exact = false;                               // wave-uniform
    unsigned n_completed = 0;                              // (diagnostic) tiles this wave completed
    auto complete = [&](v16f (&acc)[G], const int qt) __attribute__((always_inline)) {
        if (qt < 0) return;
        ++n_completed;
        const int img2 = (((qt / TPS) & 1) * STAGE_BYTES + (qt & (TPS - 1)) * TILE_BYTES) + lane_off;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned char *a2 = smem8 + (img2 ^ (ks << 5));
            const bf16x8 h2 = *reinterpret_cast<const bf16x8 *>(a2), l2 = *reinterpret_cast<const bf16x8 *>(a2 + 32 * ROW_BYTES);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(l2, wh[g][ks], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h2, wl[g][ks], acc[g], 0, 0, 0);
            }
        }
    };
    // (APX) Tile q - 2's rows must still be in LDS when a step completes it.  In a stage's steps "sub 0" and "sub 1" that tile is
    // tile 6 / 7 of the PREVIOUS stage's buffer, which this stage's steps refill for the next one -- region r by piece r, pieces 6
    // and 7 in step "sub 5".  A wave that is that far ahead of another would overwrite what that one is about to read.  So a
    // wave says when it has finished its step "sub 1" (at the head of "sub 2": its LDS reads are carried out in order, the word
    // after them), and reads everybody's word before it issues pieces 6 and 7 -- one wide LDS read per stage, met at once unless the
    // block's waves are more than a step apart.  A wait that is never met gives up after 2^20 polls (error latched) rather than hang.
    auto prog_signal = [&](const uint32_t v) __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        if (lane == 0) *((volatile __attribute__((address_space(3))) uint32_t *)s_prog + wave) = v;   // (LDS said so: a generic
                                                                  // volatile pointer compiles to flat_*, which counts on vmcnt too)
        asm volatile("" ::: "memory");
    };
    typedef int v4i_pg __attribute__((ext_vector_type(4)));
    auto prog_wait = [&](const int32_t target) __attribute__((always_inline)) {
        volatile __attribute__((address_space(3))) v4i_pg *w = (volatile __attribute__((address_space(3))) v4i_pg *)s_prog;
        for (int spin = 0;; ++spin) {
            const v4i_pg v0 = w[0], v1 = w[1];
            const int d = __builtin_amdgcn_readfirstlane(min(min(min(v0.x, v0.y), min(v0.z, v0.w)), min(min(v1.x, v1.y), min(v1.z, v1.w))) - target);
            if (d >= 0) break;
            // A wave that never shows up would hang the launch: after p.prog_limit polls (2^24 by default: about a second of
            // s_sleep -- a wave is a few microseconds behind its mates, never this) the call is latched M2D_ERR_KERNEL_TIMEOUT and
            // the wave goes on; the call's lists are INVALID then (pieces 6 and 7 may land in rows a slower wave still reads) and
            // m2d_check says so.  tests/test_gpu_catalogue.py forces it with "variant" = 15 (limit 0).
            if (spin >= p.prog_limit) {
                if (lane == 0 && atomicCAS(&p.err[0], 0, M2D_ERR_KERNEL_TIMEOUT) == 0) { p.err[1] = target; p.err[2] = (int32_t)blockIdx.x; p.err[3] = 0; }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };

    // Plain steps.  Most steps of a scan multiply a full tile of the pattern the operands were built for, well inside the block's
    // range: for those, what a step does between two bodies besides the stage's DMA pieces -- the group walk, the pattern test,
    // the range checks, the threshold arithmetic -- repeats the previous step's answers.  A general step therefore also counts
    // how many of its stage's following steps are plain (`plain_left`); those go from the DMA issue straight to the body, and the
    // relative thresholds are kept and redone only after something moved a threshold or alpha (`thr_dirty`).  Round 5, one GPU,
    // versions alternating: every tile 2.253 -> 2.205 ms, pruned 0.506 -> 0.500 ms, 1 M dishes 19.35 -> 19.0 / 2.77 -> 2.74 ms.
    // That is all there is in that section: with the compare and the thresholds moved into the body's gaps and the stage's
    // pieces issued in one burst (or inside the body) the every-tile form gained another 1 % and the pruned one lost 3 % (not
    // kept; DESIGN_LABBOOK.md section C).  The timing-only build without the section (-DM2D_DIAG=1024: 1.36 ms) is fast for other
    // reasons: no candidates, and operands that are constants -- the part holds 2.39 GHz on those and 2.0-2.15 GHz on real rows.
    int plain_left = 0;
    bool thr_dirty = true;                                 // wave-uniform
    float thr_rel[G];
#pragma unroll
    for (int g = 0; g < G; ++g) thr_rel[g] = INFINITY;

#if M2D_DIAG & 4096
    // (diagnostic) time line of workgroup (0, 0)'s first steps: p.dbg + 7 000 000 + ((wave * 512 + q) * 4 + {0: step start, 1: body start,
    // 2: body end, 3: step end}), s_memtime; scripts/diag/topk_diag.cpp prints two SIMD-mates side by side
    const bool trace_me = p.dbg && bx == 0 && by == 0;
#define TRACE(q_, k_) do { if (trace_me && (q_) < 512) { unsigned long long tt_; STAMP(tt_); if (lane == 0) p.dbg[7000000 + (((size_t)wave * 512 + (q_)) * 4 + (k_))] = tt_; } } while (0)
#else
#define TRACE(q_, k_)
#endif
    auto step = [&](v16f (&accN)[G], v16f (&accP)[G], const int q) __attribute__((always_inline)) {
#if M2D_DIAG & 16
        STAMP(t0_);
#endif
        TRACE(q, 0);
        const int sub = q & (TPS - 1);
        if (sub == 0) {                                    // tile q opens stage q / TPS: it must have landed, for every wave
#if (M2D_DIAG & 528) == 528
            unsigned long long s0_, s1_, s2_;              // (diagnostic 512: the stage wait apart from the barrier, in the slots of body / slow path)
            STAMP(s0_);
            wait_all_vmem();
            STAMP(s1_);
            __syncthreads();
            STAMP(s2_);
            t_body += s1_ - s0_; t_slow += s2_ - s1_; ++n_slow;
#else
            wait_all_vmem();
            __syncthreads();                               // also: every wave is done reading the buffer refilled next
#endif
            ps_m1 = ps_0;
            ps_0 = ps_p1;
            ps_p1 = next_stage();
            if (SHARE && p.shared_thr) exchange_thresholds();
        }
        if (KS > AR && !APX && sub == 1) {
            // With more k-steps than fragment sets (E = 128: KS = 8, AR = 4) a tile's last KS - AR k-steps are read one step
            // after its first ones: the LAST tile of the previous stage was still being read during step "sub 0", after that
            // stage's barrier.  Its LDS region is refilled by the pieces issued at sub = TPS - 2; nothing kept a wave that
            // runs two steps ahead (no insertions, while another wave rebuilds its operand at a pattern switch or works
            // through a tile of candidates) from issuing them under the reader: wrong scores for that one tile, seen once
            // the scan-start thresholds made some waves that much faster than others.  So: every wave is past step "sub 0"
            // before any wave goes on to the steps that refill that region.
            asm volatile("s_barrier" ::: "memory");
        }
        if constexpr (APX && KS > AR) {
            // E = 128, the hi x hi first form: a stage is four tiles of two pieces; beside the last tile's late k-steps (above) the
            // steps "sub 0" and "sub 1" may read tiles 2 and 3 of the previous stage's buffer again (a tile with a candidate gets
            // its cross products two steps after it was multiplied).  Pieces 0 .. 3 (tiles 0, 1) at the stage's first step, the
            // barrier at the head of step "sub 2" -- every wave is past "sub 1" -- and pieces 4 .. 7 behind it.
            static_assert(!(APX && KS > AR) || (TPS == 4 && PPW == 8), "two pieces per tile, four tiles per stage");
            if (sub == 0) issue_pieces(ps_p1, q / TPS + 1, 0, 4);
            if (sub == 2) {
                // the LDS reads that complete() issued in steps "sub 0 / 1" have RETURNED before any wave refills their region: the
                // wait is in the barrier's own asm, not left to where the compiler puts it before the consuming MFMAs
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_pieces(ps_p1, q / TPS + 1, 4, 4);
            }
        } else if constexpr (APX) {
            static_assert(!APX || KS > AR || (TPS == 8 && PCNT == 2 && PPW == 8), "pieces 6 and 7 = regions 6 and 7");
            // pieces 0 .. 5 in steps sub 0 .. 2 as ever; 6 and 7 in step sub 5, behind the look at the others' words: by then a wave
            // may be three steps ahead of the slowest before it has to wait (in step sub 3 the fast wave of a SIMD pair waited in
            // most stages: 1 400 cycles per stage), and the pieces still have three steps to land
            if (sub == 2) prog_signal((uint32_t)(q / TPS + 1));
            if (sub < 3) issue_pieces(ps_p1, q / TPS + 1, sub * PCNT, PCNT);
            if (sub == 5) {                                  // (4, 5 or 6: the same times)
                prog_wait(q / TPS + 1);
                issue_pieces(ps_p1, q / TPS + 1, 6, 2);
            }
        } else if (sub < TPS - 1) issue_pieces(ps_p1, q / TPS + 1, sub * PCNT, PCNT);
#if M2D_DIAG & 16
        STAMP(t1_); t_bar += t1_ - t0_; t0_ = t1_;
#endif
#if M2D_DIAG & 1024
        {   // TIMING ONLY (wrong lists): a step reduced to the stage barrier, the DMA pieces and the interleaved body -- no group walk,
            // no threshold arithmetic, no candidate test -- to read what everything between two bodies costs a scan
            const int img_off_ = (int)(((q / TPS) & 1) * STAGE_BYTES + sub * TILE_BYTES) + lane_off;
            const int img_prev_ = (int)((((q - 1) / TPS) & 1) * STAGE_BYTES + ((q - 1) & (TPS - 1)) * TILE_BYTES) + lane_off;
            float mx_[G];
            body(std::false_type{}, accN, accP, img_prev_, img_off_, mx_);
#pragma unroll
            for (int g = 0; g < G; ++g) asm volatile("" ::"v"(mx_[g]));
            return;
        }
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) {
            alpha_prev[g] = alpha[g];                          // tile q-2 was multiplied under the previous step's alpha
            eps_prev[g] = eps[g];
        }
        accP_exact = false;
        int nvalid = 32;
        const bool plain = (M2D_DIAG & 2048) ? false : plain_left > 0;
        if (plain) --plain_left;
        else {
        nvalid = 0;
        // physical tile of virtual tile q - 1 (the one being multiplied): its stage is v or v - 1
        const int pt1 = ((q - 1) / TPS == q / TPS ? ps_0 : ps_m1) * TPS + ((q - 1) & (TPS - 1));
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
                if constexpr (APX) {                            // tile q - 2 belongs to the operands that are about to go
                    complete(accP, q - 2);
                    accP_exact = true;
                }
                cur_pat = gp;
                build_operands(gp);
            }
            if (nvalid == 32) {
                // the steps of this stage after this one multiply tiles pt_next, pt_next + 1, ... of the stage ps_0: plain as long
                // as they are full tiles of this group, inside the dish range and inside the steps' count
                const int pt_next = sub == 0 ? ps_0 * TPS : pt1 + 1;
                const int64_t full = g_beg + (g_tot >> 5) - (t_begin + pt_next);
                int c = TPS - 1 - sub;
                c = full < c ? (int)full : c;
                c = n_phys - pt_next < c ? n_phys - pt_next : c;
                c = n - q < c ? n - q : c;
                plain_left = c > 0 ? c : 0;
            }
        }
        }
        // accumulators carry no alpha: compare against thr - alpha -- less the two roundings between `acc >= thr - alpha` and
        // `acc + alpha >= thr` (the sum is quantised at ulp(alpha), many ulps of acc under the 0.99 : 0.01 blend), so that every
        // score whose TOTAL reaches thr gets to the insertion, which compares totals: an equal total refused here would be a
        // tie nobody records (seen as a tie-listed user that one split count listed and another did not)
        // (kept from step to step; redone after a general step -- alpha_prev follows alpha one step late, so also in the step after
        //  one -- and after anything that moved a threshold: an insertion, the exchange at a stage's start)
        if (thr_dirty || !plain) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float d = thr[g] - alpha_prev[g];
                thr_rel[g] = d - 2.4e-7f * (fabsf(thr[g]) + fabsf(alpha_prev[g])) - dlt2[g];   // 2^-22 (|thr| + |alpha|); thr = +inf: NaN, no candidate
            }
        }
        thr_dirty = !plain || pend;                        // pend: this step's body inserts, and ends on share_threshold
        const int img_off = (((q / TPS) & 1) * STAGE_BYTES + sub * TILE_BYTES) + lane_off;   // tile q
        const int img_prev = ((((q - 1) / TPS) & 1) * STAGE_BYTES + ((q - 1) & (TPS - 1)) * TILE_BYTES) + lane_off;
        unsigned long long m[G][16];                       // lane masks: row r of tile q-2 beats the lane's threshold
        float mx[G];
        TRACE(q, 1);
        if (pend) body(std::true_type{}, accN, accP, img_prev, img_off, mx);
        else body(std::false_type{}, accN, accP, img_prev, img_off, mx);
        TRACE(q, 2);
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
        for (int g = 0; g < G; ++g) anyc |= __ballot(mx[g] >= (APX ? thr_rel[g] - eps_prev[g] : thr_rel[g]));   // (APX: mx is of hi x hi scores)
#if M2D_DIAG & 16
        STAMP(t1_); if (!(M2D_DIAG & 512)) t_body += t1_ - t0_; t0_ = t1_; ++n_step;
#endif
        pend = false;
#if M2D_DIAG & 8
        asm volatile("" ::"s"(anyc), "v"(mx[0]), "v"(mx[G - 1]));
#endif
        bool cand_step = (M2D_DIAG & 8) ? false : anyc != 0ull;
        if constexpr (APX) {
            if (cand_step) {                                // a hi x hi score came within eps of a threshold: the tile's exact scores
#if M2D_DIAG & 16
                ++n_ins;                                    // (APX builds: d[1] counts the tiles completed, not the multi-candidate tiles)
#endif
                if (!accP_exact) complete(accP, q - 2);
                unsigned long long any2 = 0ull;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float m2 = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) m2 = fmaxf(m2, accP[g][r]);
                    mx[g] = m2;
                    any2 |= __ballot(m2 >= thr_rel[g]);
                }
                cand_step = any2 != 0ull;
            }
        }
        if (cand_step) {                                    // some lane of tile q-2 beat its threshold
            const int pt2 = ((q - 2) / TPS == q / TPS ? ps_0 : ps_m1) * TPS + ((q - 2) & (TPS - 1));   // physical tile of tile q - 2
            const int32_t sbase = (int32_t)((t_begin + pt2) * 32) + 4 * h;
            // per-lane 16-bit map of candidate rows (bit 15 - r), built here -- in the quarter of the steps that have a
            // candidate -- from the sixteen lane masks: map = 2 map + mask bit, one v_addc each
            // The sixteen lane masks "row r of the tile beats the lane's threshold" are worked out HERE, in the steps that have a
            // candidate.  Through round 4 they were part of every step's body, four compares between the MFMAs of each k-step group:
            // an MFMA leaves vector issue free three quarters of its time, but the body was short of exactly that -- 1 009 cycles per
            // step and wave for 768 of the SIMD's two waves' matrix work -- and 85 % of an every-tile scan's steps have no candidate.
            // Same masks, same lists; body 1 009 -> 910 cycles, every tile 2.38 -> 2.27 ms, pruned 0.508 -> 0.500 ms (round 5).
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) m[g][r] = __ballot(accP[g][r] >= thr_rel[g]);
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
                thr_dirty = true;
#if M2D_DIAG & 16
                if (!APX) ++n_ins;
#endif
            }
#if M2D_DIAG & 16
            asm volatile("" ::"v"(thr[0]), "v"(px[0]));
            STAMP(t1_); if (!(M2D_DIAG & 512)) { t_slow += t1_ - t0_; ++n_slow; }
            if (q <= TPS + 2) { t_slow_first += t1_ - t0_; ++n_slow_first; }
#endif
        }
        TRACE(q, 3);
    };

    for (int q = 1; q <= n + 1 && n > 0; q += 2) {
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
    // diagnostic: tiles stepped through (per block; word 0) and, APX, (wave, tile) pairs whose cross products were multiplied (word 1:
    // a word of its own since round 6 -- packed above bit 36 of word 0 the count wrapped past 2.7e8, which a 500 k x 1 M every-tile launch exceeds)
    if (p.tiles_scanned && lane == 0) {
        if (wave == 0) atomicAdd(p.tiles_scanned, (unsigned long long)n);
        if (APX && n_completed) atomicAdd(p.tiles_scanned + 1, (unsigned long long)n_completed);
    }
#if M2D_DIAG & 16
    if (lane == 0 && p.dbg) {
        // one record per (workgroup, wave) in launch order (a 128-user launch has more user blocks than (nU + 255) / 256: the
        // round-5 index by (range, user block) made its records overwrite one another)
        unsigned long long *d = p.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave) * 8;
        d[0] = t_body; d[1] = n_ins; d[2] = t_bar; d[3] = t_slow; d[4] = n_slow; d[5] = n_step;
        d[6] = __builtin_amdgcn_s_memtime() - clk0; d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
#if M2D_DIAG & 8192
        // the launch's time line (topk_diag M2D_DIAG_DUMP): when the wave started (100 MHz, low 40 bits) beside its duration, and
        // which (user block, dish range) it ran beside its insert count
        d[7] = ((rt0 & 0xffffffffffull) << 24) | (d[7] & 0xffffffull);
        d[1] = (n_ins & 0xfffffull) | ((unsigned long long)(unsigned)bx << 20) | ((unsigned long long)(unsigned)by << 52);
#endif
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

}  // namespace

// The instantiations are compiled in two translation units, by kernel width (m2d_catalogue_scan_bf16_e64.hip / _e128.hip define
// M2D_SCAN_E and include this file); a build that includes this file as it is (scripts/diag) gets them all in one.
#if !defined(M2D_SCAN_E)
#define M2D_SCAN_LAUNCH m2d_topk_scan_bf16_launch
#elif M2D_SCAN_E == 64
#define M2D_SCAN_LAUNCH m2d_topk_scan_bf16_launch_e64
#else
#define M2D_SCAN_LAUNCH m2d_topk_scan_bf16_launch_e128
#endif
M2D_INTERNAL int M2D_SCAN_LAUNCH(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);
int M2D_SCAN_LAUNCH(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st)
{
#define M2D_SCAN_GO(KERN, THREADS)                                                      \
    {                                                                                   \
        auto kern = KERN;                                                               \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));                    \
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, st, a);                      \
        return M2D_OK;                                                                  \
    }
#define M2D_SCAN_BF16(EV, KRV)                                                                                                   \
    if (s.E == EV && s.KR == KRV) {                                                                                               \
        constexpr bool CAN_KEEP = !(EV == 128 && KRV == 16);                                                                      \
        if (s.hv && s.apx) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, true, 8, false, true>), 512)                      \
        if (s.hv) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, true>), 512)                                               \
        if (!s.pipe) M2D_SCAN_GO((m2d_topk_grouped_bf16<EV, 8, KRV>), 512)                                                        \
        if constexpr (EV == 64) {                                                                                                 \
            if (s.apx && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, true, true>), 512)                \
            if (s.apx) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, false, true>), 512)                         \
            if (s.waves == 4 && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 4, true>), 256)               \
            if (s.waves == 4) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 4>), 256)                               \
        }                                                                                                                         \
        if constexpr (EV == 128) {                                                                                                \
            if (CAN_KEEP && s.apx && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, CAN_KEEP, true>), 512)    \
            if (s.apx && !s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, false, true>), 512)              \
        }                                                                                                                         \
        if (CAN_KEEP && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, CAN_KEEP>), 512)                   \
        M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1>), 512)                                                               \
    }
#if !defined(M2D_SCAN_E) || M2D_SCAN_E == 64
    M2D_SCAN_BF16(64, 10) M2D_SCAN_BF16(64, 16)
#endif
#if !defined(M2D_SCAN_E) || M2D_SCAN_E == 128
    M2D_SCAN_BF16(128, 10) M2D_SCAN_BF16(128, 16)
#endif
#undef M2D_SCAN_BF16
#undef M2D_SCAN_GO
    h->last_error = "m2d_topk_scan_bf16_launch: no such instantiation";
    return M2D_ERR_UNSUPPORTED;
}
