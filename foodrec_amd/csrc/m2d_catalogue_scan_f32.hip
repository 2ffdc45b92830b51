// Pattern-grouped scan on exact-f32 MFMA (v_mfma_f32_32x32x2_f32): 0/1 masks, contraction over E instead of (C+1) E.
// Serves "topk_bf16x3" = 0, E = 32, and every embedding size without a kernel of its own (rows zero-padded to 32 / 64 / 128 / 256
// floats -- the reference's default embed_size 200, Train_recommender.py:51-58, among them).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

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
    int slot_lo[SW / 2];                                   // float offset of the lane's low slot bits, by T mod SW / 2 (see the MFMA loop)
#pragma unroll
    for (int t = 0; t < SW / 2; ++t) slot_lo[t] = (((2 * t + h) ^ (j & (SW - 1))) & (SW - 1)) * 4;
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
                // slot (2 T + h) ^ key: the XOR moves the low bits only -- SW / 2 lane-dependent values by T mod SW / 2, the rest in
                // the read's immediate (as one expression per T the E8 lane offsets were all hoisted out of the tile loop: 32 VGPRs
                // at E = 256, where the k = 16 instantiation spilled)
                const v4f av = *reinterpret_cast<const v4f *>(img + slot_lo[T % (SW / 2)] + ((2 * T) & ~(SW - 1)) * 4);
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

}  // namespace

int m2d_topk_scan_f32_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st)
{
#define M2D_SCAN_F32(EV, KRV, PADV)                                                     \
    if (s.E == EV && s.KR == KRV && s.pad == PADV) {                                    \
        auto kern = m2d_topk_grouped<EV / 8, 8, KRV, PADV>;                             \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));                    \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                          \
        return M2D_OK;                                                                  \
    }
    M2D_SCAN_F32(32, 10, false) M2D_SCAN_F32(32, 16, false) M2D_SCAN_F32(64, 10, false) M2D_SCAN_F32(64, 16, false)
    M2D_SCAN_F32(128, 10, false) M2D_SCAN_F32(128, 16, false)
    M2D_SCAN_F32(32, 10, true) M2D_SCAN_F32(32, 16, true) M2D_SCAN_F32(64, 10, true) M2D_SCAN_F32(64, 16, true)
    M2D_SCAN_F32(128, 10, true) M2D_SCAN_F32(128, 16, true) M2D_SCAN_F32(256, 10, true) M2D_SCAN_F32(256, 16, true)
#undef M2D_SCAN_F32
    h->last_error = "m2d_topk_scan_f32_launch: no such instantiation";
    return M2D_ERR_UNSUPPORTED;
}
