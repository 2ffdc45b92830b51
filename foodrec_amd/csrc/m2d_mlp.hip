// Build-defined extension (NO reference counterpart; BASELINE.json configs[2]):
// a 3-layer scoring head on top of the reference score.
//
//   z[k]  = flatten(PM[u])[k] * Dt[d][k]            k < K = (C+1)*E; sum_k z[k] IS the reference score
//                                                   (factored form of Model_Recommender.py:67-96)
//   h1    = relu(W1^T z + b1)   W1 [K, H1]
//   h2    = relu(W2^T h1 + b2)  W2 [H1, H2]
//   score = sum_k z[k] + (w3 . h2 + b3)
//
// With the head absent (or w3 = 0, b3 = 0) the score is the reference formula.  The closest reference
// material for a dense head is in the OTHER model (Code/WIRCNN/Model_WIRCNN.py:174, :208); it is not the
// Recommender's scorer and nothing pins this head except the build's own CPU restatement.
//
// MFMA kernel (H1 = 256, H2 = 64, K % 64 == 0), exact-f32 v_mfma_f32_32x32x2_f32:
//   * a wave owns 32 pairs; pairs sit on the MFMA *column* (B operand = z, 32 k-values per lane per
//     64-wide K chunk, built in registers from two gathered rows), hidden units on the rows, so the
//     layer-1 accumulators (8 tiles x 16 VGPRs) are already the B operand of layer 2 -- no transpose,
//     no LDS round trip between layers -- and layer 3 is a per-lane dot over the layer-2 accumulators;
//   * W1 streams through a 2 x 64 KiB LDS ring by LDS-DMA (one 1-KiB piece = one k-row), one barrier per
//     256 MFMAs per wave; W2 rides the same ring as the (K/64 + 1)-th stage;
//   * 8 waves (2 per SIMD) per workgroup = 256 pairs per tile, persistent grid-stride over tiles.
// Roofline: MFMA (2*(K*H1 + H1*H2 + H2) flop per pair = 360 704 at E = 128) vs 157.3 TFLOP/s f32.
#include "m2d_engine.h"

// Timing-only instrumentation for scripts/diag/mlp_diag.cpp (never defined in the product build).
#ifndef M2D_MLP_DIAG
#define M2D_MLP_DIAG 0
#endif
static unsigned long long *g_m2d_mlp_diag_buffer = nullptr;
#if M2D_MLP_DIAG
#define MSTAMP(x) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#define MACC(dst) do { MSTAMP(t1_); dst += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define MSTAMP(x)
#define MACC(dst)
#endif
// diag bit 6 (the barrier-arrival trace): block 0's first gatherer also logs, per step, when it started, had built z, had issued
// its row requests, had done the tile's last-step sums, and reached the barrier (5 s_memtime words per step, first 96 steps)
#if M2D_MLP_DIAG & 64
#define GSTAMP(k) do { if (blockIdx.x == 0 && wave == 4 && (threadIdx.x & 63) == 0 && nbar_ < 96 && p.dbg) { unsigned long long ts_; \
        MSTAMP(ts_); p.dbg[4096 * 8 + 8 * 128 * 2 + nbar_ * 5 + (k)] = ts_; } } while (0)
#else
#define GSTAMP(k)
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct MlpArgs {
    const float *pm;   // [U, K]
    const float *dt;   // [rows, K]
    const float *w1;   // [K, H1]
    const float *b1;   // [H1]
    const float *w2;   // [H1, H2]
    const float *b2;   // [H2]
    const float *w3;   // [H2]
    float b3;
    const int32_t *users;
    const int32_t *items;
    float *out;
    int64_t B, U, I, user_base;
    int32_t K, H1, H2;
    int32_t *err;
    const __bf16 *w1x3;        // split-bf16 image of W1 (m2d_mlp_split_w1), or null
    const __bf16 *w2x3;        // split-bf16 fragment image of W2 (m2d_mlp_split_w2), or null
    unsigned long long *dbg;   // scripts/diag only
    // producer / consumer kernel: pairs grouped by dish mask pattern (m2d_mlp_pg_* below)
    const int32_t *perm;         // [ntiles * 128] slot -> pair index, -1 = padding
    const uint32_t *tile_blocks; // [ntiles] the E-wide k-blocks a tile's pattern keeps: nibble j = j-th block, bits 28-31 = count
    const int32_t *ntiles_dev;   // [1]
    int32_t pshift;              // log2(periods of 32 k-values per block)
};

__device__ __forceinline__ void latch(int32_t *err, int code, int64_t value, int64_t index)
{
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int32_t)value;
        err[2] = (int32_t)(index & 0xffffffff);
        err[3] = (int32_t)(index >> 32);
    }
}

constexpr int MH1 = 256, MH2 = 64, MWAVES = 8;
constexpr int RING_FLOATS = 64 * MH1;   // one stage: 64 k-rows of W1 (= all of W2: 256 x 64)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// hidden unit held by accumulator tile nt, register r, lane half h (see the two layer-1 forms below)
template <bool X3>
__device__ __forceinline__ int hidden_unit(int nt, int r, int h)
{
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;           // row of the 32x32 tile
    return X3 ? 32 * nt + i : 128 * (nt >> 2) + 4 * i + (nt & 3);
}

// W1 [K, 256] f32 -> per 64-row chunk a transposed split-bf16 image [hi: 256 n x 64 k][lo: 256 n x 64 k]
// (64 KiB per chunk, the size of one LDS stage): the A fragment of v_mfma_f32_32x32x16_bf16 is 8 consecutive k
// of one hidden unit, i.e. one 16-byte read from this layout.
__global__ __launch_bounds__(256) void m2d_mlp_split_w1(const float *w1, int K, __bf16 *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one (k, n) element
    if (t >= (int64_t)K * MH1) return;
    const int k = (int)(t / MH1), n = (int)(t % MH1);
    const float x = w1[t];
    const __bf16 hi = (__bf16)x;
    __bf16 *chunk = out + (size_t)(k / 64) * (2 * MH1 * 64);
    chunk[(size_t)n * 64 + (k & 63)] = hi;
    chunk[(size_t)MH1 * 64 + (size_t)n * 64 + (k & 63)] = (__bf16)(x - (float)hi);
}

// W2 [256, 64] f32 -> split-bf16 fragment image [hi | lo], each [mt 2][ks 16][h 2][m 32][j 8]: the 16 bytes at
// (mt, ks, h, m) are W2[n][32 mt + m] for the 8 hidden units n = 32 (ks >> 1) + 16 (ks & 1) + (j & 3) + 8 (j >> 2) + 4 h
// that lane half h of the layer-1 accumulator tile ks >> 1 holds in registers 8 (ks & 1) + j -- so the layer-1
// result is the layer-2 B operand as it stands.  64 KiB = one LDS stage, conflict-free b128 reads.
__global__ __launch_bounds__(256) void m2d_mlp_split_w2(const float *w2, __bf16 *out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;                   // one output element, [mt][ks][h][m][j]
    if (t >= MH1 * MH2) return;
    const int j = t & 7, m = (t >> 3) & 31, hh = (t >> 8) & 1, ks = (t >> 9) & 15, mt = t >> 13;
    const int n = 32 * (ks >> 1) + 16 * (ks & 1) + (j & 3) + 8 * (j >> 2) + 4 * hh;
    const float x = w2[n * MH2 + 32 * mt + m];
    const __bf16 hi = (__bf16)x;
    out[t] = hi;
    out[MH1 * MH2 + t] = (__bf16)(x - (float)hi);
}

// X3 = false: layer 1 on exact-f32 MFMA.  X3 = true: layer 1 on split-bf16 MFMA (x = hi + lo, three bf16
// products, fp32 accumulation; per-product relative error <= ~1.2e-5, see m2d_topk_grouped_bf16) -- layer 1 is
// 91 % of the head's flops and f32 MFMA runs at 1/16 of the bf16 rate.  Layer 2 takes the same form (its f32
// MFMAs would otherwise cost as much as all of split-bf16 layer 1); layer 3 and the reference score stay f32.
// PADK: the tables' K = (C + 1) E = p.K is not a multiple of 64 (the reference's embed_size 200 gives 1000): the kernel runs
// KCH = ceil(K / 64) chunks on a W1 copy zero-padded to 64 KCH rows and reads zeros for the k-values past the end of a row.
template <int KCH /* K / 64 */, bool X3, bool PADK = false>
__global__ __launch_bounds__(MWAVES * 64) void m2d_mlp_mfma(MlpArgs p)
{
    extern __shared__ __align__(16) float smem[];
    float *ring = smem;                       // [2][RING_FLOATS]
    float *sb1 = smem + 2 * RING_FLOATS;      // [256]
    float *sb2 = sb1 + MH1;                   // [64]
    float *sw3 = sb2 + MH2;                   // [64]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pl = lane & 31, h = lane >> 5;
    constexpr int K = KCH * 64;
    constexpr int NST = KCH + 1;              // stages per tile: KCH chunks of W1, then W2
    const int K4 = PADK ? p.K / 4 : K / 4;    // 16-byte units per table row

    for (int i = threadIdx.x; i < MH1; i += MWAVES * 64) sb1[i] = p.b1[i];
    if (threadIdx.x < MH2) {
        sb2[threadIdx.x] = p.b2[threadIdx.x];
        sw3[threadIdx.x] = p.w3[threadIdx.x];
    }

    // stage s (mod NST): s < KCH -> rows [64 s, 64 s + 64) of W1; s == KCH -> W2.  64 pieces of 1 KiB.
    auto issue_stage = [&](int s, int buf) {
        float *dst = ring + (size_t)buf * RING_FLOATS;
        if (X3 && s < KCH) {
            // split-bf16 image: 512 rows (256 hi + 256 lo) of 128 B; 16-B slots XOR-swizzled by (row >> 1) & 7 so
            // the b128 fragment reads of 16 consecutive rows hit 16 different slots of the 256-B bank row
            const unsigned char *src = reinterpret_cast<const unsigned char *>(p.w1x3) + (size_t)s * (RING_FLOATS * 4);
            for (int pc = wave; pc < 64; pc += MWAVES) {
                const int rw = pc * 8 + (lane >> 3), sl = lane & 7;
                const int q = sl ^ ((rw >> 1) & 7);
                lds_dma16(src + (size_t)rw * 128 + q * 16, dst + pc * 256);
            }
            return;
        }
        const float *src = s < KCH ? p.w1 + (size_t)s * 64 * MH1 : X3 ? reinterpret_cast<const float *>(p.w2x3) : p.w2;
        for (int pc = wave; pc < 64; pc += MWAVES)
            lds_dma16(src + pc * 256 + lane * 4, dst + pc * 256);
    };

    const int64_t ntiles = (p.B + 32 * MWAVES - 1) / (32 * MWAVES);
    int ringpos = 0;                          // parity of the stage being consumed
    if ((int64_t)blockIdx.x < ntiles) issue_stage(0, 0);
    wait_all_vmem();
    __syncthreads();

#if M2D_MLP_DIAG
    unsigned long long t_pro = 0, t_l1 = 0, t_bar = 0, t_l23 = 0, t_vm = 0, t_gw = 0, t_mz = 0, n_tiles = 0, t0_, t1_;
    MSTAMP(t0_);
#endif
    // this lane's pair of a tile: ids -> the two row pointers (out-of-range ids are latched and clamped)
    int64_t pi = 0;
    bool valid = false, bad = false;
    const v4f *pu = nullptr, *pd = nullptr;
    auto locate = [&](int64_t tile) {
        pi = tile * (32 * MWAVES) + wave * 32 + pl;
        valid = pi < p.B;
        const int32_t uid = valid ? p.users[pi] : (int32_t)p.user_base;
        int32_t did = valid ? p.items[pi] : 0;
        int64_t ul = (int64_t)uid - p.user_base;
        bad = false;
        if (ul < 0 || ul >= p.U) { latch(p.err, M2D_ERR_BAD_USER_ID, uid, pi); ul = 0; bad = true; }
        if (did < 0 || (int64_t)did >= p.I) { latch(p.err, M2D_ERR_BAD_ITEM_ID, did, pi); did = 0; bad = true; }
        if (M2D_MLP_DIAG & 8) { ul = 0; did = 0; }          // diag bit 3: every pair reads row 0 (loads issue, no HBM traffic)
        // f32 form: the lane owns k = 64 kc + 32 h + t; bf16 form: k = 64 kc + 16 ks + 8 h + j (fragment order)
        pu = reinterpret_cast<const v4f *>(p.pm) + (size_t)ul * K4 + (X3 ? 2 : 8) * h;
        pd = reinterpret_cast<const v4f *>(p.dt) + (size_t)did * K4 + (X3 ? 2 : 8) * h;
    };
    v4f ra[4], rb[4];
    auto load_raw = [&](int g) {                            // g = 2 kc + half
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f4 = X3 ? (g >> 1) * 16 + (g & 1) * 8 + 4 * (i >> 1) + (i & 1) : (g >> 1) * 16 + (g & 1) * 4 + i;
            if (!PADK || f4 + (X3 ? 2 : 8) * h < K4) {
                ra[i] = pu[f4];
                rb[i] = pd[f4];
            } else {
                ra[i] = v4f{0.f, 0.f, 0.f, 0.f};
                rb[i] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if ((int64_t)blockIdx.x < ntiles) {
        locate(blockIdx.x);
        load_raw(0);
    }

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // biases ride in as the initial accumulators: layer 2 / 3 then need no add and no LDS read per step
        v16f acc1[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc1[nt][r] = sb1[hidden_unit<X3>(nt, r, h)];
        float base = 0.f;

        // ---- layer 1: K in chunks of 64 (one LDS stage); lane (pair pl, half h) owns k = 64 kc + 32 h + t.
        // Each chunk is multiplied in two halves of 16 k-values per lane so that the NEXT half's two gathered
        // rows (2 x 4 float4) are in flight under the current half's 128 MFMAs: the random-row latency is off
        // the critical path at 176 live registers (acc 128 + z 16 + raw 32), inside the 2-waves/SIMD budget.
        float z[16];
        bf16x8 zh[2], zl[2];                                // X3: two k-steps of 8 k-values per lane, split hi / lo
        auto make_z = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const v4f zz = ra[i] * rb[i];
                z[4 * i + 0] = zz.x; z[4 * i + 1] = zz.y; z[4 * i + 2] = zz.z; z[4 * i + 3] = zz.w;
                base += (zz.x + zz.y) + (zz.z + zz.w);
            }
            if constexpr (X3) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const __bf16 hi = (__bf16)z[e];
                    zh[e >> 3][e & 7] = hi;
                    zl[e >> 3][e & 7] = (__bf16)(z[e] - (float)hi);
                }
            }
        };
        make_z();                                           // rows of half 0 were requested a tile ago
#if M2D_MLP_DIAG
        asm volatile("" ::"v"(z[0]), "v"(z[15]));
        MACC(t_pro); ++n_tiles;
#endif
        const int xkey = (pl >> 1) & 7;                     // X3: this lane's row swizzle (rows 32 nt + pl)
#pragma unroll 1
        for (int kc = 0; kc < KCH; ++kc) {
            const int buf = ringpos & 1;
            if (!(M2D_MLP_DIAG & 4)) issue_stage((kc + 1) % NST, buf ^ 1);   // kc + 1 == KCH -> W2 (diag bit 2: no DMA)
            // A operand: one ds_read_b128 gives W1[k][128 g + 4 pl + q], q = 0..3, i.e. four hidden-unit tiles
            // at once; tile (g, q) row i is hidden unit n = 128 g + 4 i + q.  Reads run one step ahead.
            const v4f *wrow = reinterpret_cast<const v4f *>(ring + (size_t)buf * RING_FLOATS + (size_t)(32 * h) * MH1) + pl;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int g = 2 * kc + half;
                if (!(M2D_MLP_DIAG & 2) && g + 1 < 2 * KCH) load_raw(g + 1);   // diag bit 1: no in-tile gather
                if constexpr (X3) {
                    // rows n = 32 nt + pl of the [n][k] image; k-step ks = 2 half + ksl is logical slot 2 ks + h
                    const unsigned char *img = reinterpret_cast<const unsigned char *>(ring + (size_t)buf * RING_FLOATS) + pl * 128;
                    // fragments are read one (k-step, tile) ahead of the three MFMAs that consume them
                    auto frag = [&](int it, bf16x8 &ah, bf16x8 &al) {
                        const int slot = ((2 * (2 * half + (it >> 3)) + h) ^ xkey) * 16;
                        ah = *reinterpret_cast<const bf16x8 *>(img + (it & 7) * (32 * 128) + slot);
                        al = *reinterpret_cast<const bf16x8 *>(img + MH1 * 128 + (it & 7) * (32 * 128) + slot);
                    };
                    bf16x8 ah[2], al[2];
                    frag(0, ah[0], al[0]);
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        if (it + 1 < 16) frag(it + 1, ah[(it + 1) & 1], al[(it + 1) & 1]);
                        const int nt = it & 7, ksl = it >> 3;
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[it & 1], zh[ksl], acc1[nt], 0, 0, 0);
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[it & 1], zl[ksl], acc1[nt], 0, 0, 0);
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[it & 1], zh[ksl], acc1[nt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#if M2D_MLP_DIAG
                    asm volatile("" ::"v"(acc1[7][0]));
                    MACC(t_l1);
                    wait_all_vmem();
                    MACC(t_gw);
#endif
                    if (g + 1 < 2 * KCH) make_z();
#if M2D_MLP_DIAG
                    asm volatile("" ::"v"(zh[0][0]), "v"(zl[1][7]));
                    MACC(t_mz);
#endif
                    continue;
                }
                v4f a_cur[2], a_nxt[2];
                a_cur[0] = wrow[(16 * half) * (MH1 / 4)];
                a_cur[1] = wrow[(16 * half) * (MH1 / 4) + 32];
#pragma unroll
                for (int tt = 0; tt < 16; ++tt) {
                    if (tt + 1 < 16) {
                        a_nxt[0] = wrow[(16 * half + tt + 1) * (MH1 / 4)];
                        a_nxt[1] = wrow[(16 * half + tt + 1) * (MH1 / 4) + 32];
                    }
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        acc1[4 * gq + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[gq].x, z[tt], acc1[4 * gq + 0], 0, 0, 0);
                        acc1[4 * gq + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[gq].y, z[tt], acc1[4 * gq + 1], 0, 0, 0);
                        acc1[4 * gq + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[gq].z, z[tt], acc1[4 * gq + 2], 0, 0, 0);
                        acc1[4 * gq + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[gq].w, z[tt], acc1[4 * gq + 3], 0, 0, 0);
                    }
                    a_cur[0] = a_nxt[0];
                    a_cur[1] = a_nxt[1];
                }
                if (g + 1 < 2 * KCH) make_z();
            }
#if M2D_MLP_DIAG
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) asm volatile("" ::"v"(acc1[nt][0]));
            MACC(t_l1);
#endif
            wait_all_vmem();
            MACC(t_vm);
            __syncthreads();
            MACC(t_bar);
            ++ringpos;
        }

        // ---- layer 2: acc1[4 g + q] (row i = (r&3) + 8 (r>>2) + 4 h is hidden unit 128 g + 4 i + q; column =
        // pair) is the B operand as it stands ------------------------------------------------------------
        {
            const int buf = ringpos & 1;
            issue_stage(0, buf ^ 1);                       // next tile's first W1 chunk (harmless if none)
            const float *w2s = ring + (size_t)buf * RING_FLOATS + pl;
            // the finished tile's bookkeeping, then start the NEXT tile's gather under layers 2-3
            const int64_t cur_pi = pi;
            const bool cur_valid = valid, cur_bad = bad;
            if (tile + gridDim.x < ntiles) {
                locate(tile + gridDim.x);
                load_raw(0);
            }
            v16f acc2[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] = sb2[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h];
            if constexpr (X3) {
                const unsigned char *img = reinterpret_cast<const unsigned char *>(ring + (size_t)buf * RING_FLOATS) + (h * 32 + pl) * 16;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    bf16x8 bh, bl;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float hv = fmaxf(acc1[ks >> 1][8 * (ks & 1) + j], 0.f);
                        const __bf16 hi = (__bf16)hv;
                        bh[j] = hi;
                        bl[j] = (__bf16)(hv - (float)hi);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(img + (mt * 16 + ks) * 1024);
                        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(img + MH1 * MH2 * 2 + (mt * 16 + ks) * 1024);
                        acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2[mt], 0, 0, 0);
                        acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2[mt], 0, 0, 0);
                        acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2[mt], 0, 0, 0);
                    }
                }
            } else
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = hidden_unit<X3>(nt, r, h);
                    const float hv = fmaxf(acc1[nt][r], 0.f);
                    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2s[n * MH2], hv, acc2[0], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2s[n * MH2 + 32], hv, acc2[1], 0, 0, 0);
                }
            }
            // ---- layer 3: per-lane dot over the layer-2 accumulators, halves combined by one shuffle ----
            float o = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    o = fmaf(sw3[m], fmaxf(acc2[mt][r], 0.f), o);
                }
            o += __shfl_xor(o, 32, 64);
            base += __shfl_xor(base, 32, 64);
            if (h == 0 && cur_valid) p.out[cur_pi] = cur_bad ? __builtin_nanf("") : (base + (o + p.b3));
            MACC(t_l23);
            wait_all_vmem();
            __syncthreads();
            MACC(t_bar);
            ++ringpos;
        }
    }
#if M2D_MLP_DIAG
    if (lane == 0 && p.dbg) {
        unsigned long long *d = p.dbg + ((size_t)blockIdx.x * MWAVES + wave) * 8;
        d[0] = t_pro; d[1] = t_l1; d[2] = t_bar; d[3] = t_l23; d[4] = n_tiles; d[5] = t_vm; d[6] = t_gw; d[7] = t_mz;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 head, producer / consumer form (the default for the shapes it covers).  The kernel above makes every
// wave gather its own rows with one 16-byte load per lane and row -- 32 cache lines per wave-instruction, which blocks
// the wave's in-order issue (and with it its MFMAs) while the cache has misses outstanding: layer 1 took 93 k cycles
// per tile and wave from a 0.5 GB table against 77 k from a table that sits in L2, for 31 k cycles of MFMAs.  Here the
// roles are split by wave, one matrix wave per SIMD beside a memory wave:
//   waves 0-3  consumers   32 pairs each (tile = 128 pairs): ds_read + MFMA, and in the MFMA shadows the hi / lo
//                          split of the next period's z.  Layer 1 in periods of 32 k-values (48 MFMAs), layer 2 in
//                          two periods (W2 halves), layer 3 per lane.
//   waves 4, 6 gatherers   64 pairs each: whole 128-byte lines of the two gathered rows (8 rows per wave-instruction,
//                          16 B per lane), two periods ahead of use; z = u * d and the reference score's partial
//                          sums; ds_write_b128 of z into the consumer's B-operand order (the transpose the per-lane
//                          loads of the kernel above were avoiding).
//   waves 5, 7 loaders     LDS-DMA of the W1 / W2 image, half a ring stage each per period (16 pieces of 1 KiB), two
//                          periods ahead.  Their vmcnt queue holds nothing else: the DMA is inline asm the compiler's
//                          wait bookkeeping does not see, and in a wave that also had row loads in flight the waits
//                          it places for those came out one to three periods too strict.
// Hand-off: one s_barrier per period, placed in the MIDDLE of the consumers' period P.  Passing it means: period P + 1
// (ring stage and z set) is complete in LDS, and the consumers are done with period P - 1 and the first half of P.
// The consumers therefore read the first fragments of period P + 1 under the last MFMAs of period P and never wait
// for LDS at a period boundary.  The ring holds 7 half-stages of 16 KiB (3.5 periods: the stage two periods ahead can
// be written while the current one and the next are live); z needs 2 sets (a consumer copies its z into registers
// half a period before it uses it).
// LDS: ring 7 x 16 KiB | z sets 2 x (4 consumers x 4 KiB f32) | reference-score sums 2 x 512 B (by tile parity: a
// two-period tile's sums are written while the consumers may still be reading the previous tile's) | biases.
constexpr int PC_HALF = 16384, PC_STAGE = 2 * PC_HALF, PC_NHB = 7, PC_ZSET = 16384;
constexpr int PC_Z_OFF = PC_NHB * PC_HALF, PC_BASE_OFF = PC_Z_OFF + 2 * PC_ZSET, PC_B1_OFF = PC_BASE_OFF + 1024;
constexpr int PC_LDS_BYTES = PC_B1_OFF + (MH1 + 2 * MH2) * 4;
constexpr int PC_PAIRS = 128;             // pairs per tile
typedef int v4i_pc __attribute__((ext_vector_type(4)));

// W1 [K, 256] f32 -> per 32 k-values (one step of v_mfma_f32_16x16x32_bf16) one ring stage of two halves, a half = the 128
// hidden units n = 128 hf .. 128 hf + 127 as [hi: 8 tiles x 1 KiB | lo: the same], a tile = 16 units x 64 B (k-values 0 .. 31 of the
// period); the four 16-byte slots of a row (k-values 8 g .. 8 g + 7: lane group g's fragment) sit at slot g ^ 2 ((n >> 3) & 1), so
// that the four 16-lane groups a ds_read_b128 is served in -- {0-3, 12-15, 20-27}, ... -- each touch sixteen different 16-byte
// slots of the 256-byte bank row (rows are 64 B: four rows per bank row).  Halves by hidden units, not by k: the consumers are done
// with a period's first half at its middle barrier, which is what lets the ring do with 3.5 stages.
__global__ __launch_bounds__(256) void m2d_mlp_image_pc_w1(const float *w1, int K, __bf16 *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one (k, n) element
    if (t >= (int64_t)K * MH1) return;
    const int k = (int)(t / MH1), n = (int)(t % MH1);
    const int kk = k & 31, g = kk >> 3, j = kk & 7;
    const float x = w1[t];
    const __bf16 hi = (__bf16)x;
    __bf16 *half = out + ((size_t)(k >> 5) * 2 + (n >> 7)) * (PC_HALF / 2);
    const int e = ((n & 127) >> 4) * 512 + (n & 15) * 32 + ((g ^ (((n >> 3) & 1) << 1)) << 3) + j;
    half[e] = hi;
    half[PC_HALF / 4 + e] = (__bf16)(x - (float)hi);
}

// W2 [256, 64] f32 -> two ring stages = four halves; half q holds layer-2 steps s = 2 q, 2 q + 1 (32 hidden units each) as
// [s & 1: 8 KiB][h2 tile mt (16 units): 2 KiB][hi 1 KiB | lo 1 KiB], a 1-KiB tile = 16 units x 64 B with the slots swizzled as
// above.  The k-values of a step come in the order the layer-1 accumulators hold them: lane group g's fragment element j is
// hidden unit 32 s + 16 (j >> 2) + 4 g + (j & 3) -- rows 4 g .. 4 g + 3 of accumulator tiles 2 s and 2 s + 1 -- so relu(acc1) is the
// B operand with no lane movement.
__global__ __launch_bounds__(256) void m2d_mlp_image_pc_w2(const float *w2, __bf16 *out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;                   // [s 8][mt 4][m 16][g 4][j 8]
    if (t >= MH1 * MH2) return;
    const int j = t & 7, g = (t >> 3) & 3, m = (t >> 5) & 15, mt = (t >> 9) & 3, s = t >> 11;
    const int n = 32 * s + 16 * (j >> 2) + 4 * g + (j & 3);
    const float x = w2[n * MH2 + 16 * mt + m];
    const __bf16 hi = (__bf16)x;
    __bf16 *half = out + (size_t)(s >> 1) * (PC_HALF / 2);
    const int e = (s & 1) * 4096 + mt * 1024 + m * 32 + ((g ^ (((m >> 3) & 1) << 1)) << 3) + j;
    half[e] = hi;
    half[512 + e] = (__bf16)(x - (float)hi);
}

// ---- pairs grouped by dish mask pattern ------------------------------------------------------------------------
// The E k-values of a category whose mask weight is 0 are zeros in the dish vector: z = 0 there whatever the user row
// holds, and a tile whose 128 pairs all lack that category can skip those periods altogether -- their MFMAs, their
// share of the W1 stream and their row requests.  So a launch first buckets its pairs by the dish's pattern of
// non-zero weights (a histogram, a scan that pads every bucket to whole tiles, a scatter of pair indices) and the
// kernel walks tiles of one pattern each; scores go back to out[pair].  With uniform non-empty subsets of 4
// categories a tile runs (1 + 2.13) / 5 of the periods on average.  Without grouping (option skip_masked = 0, or blocks
// that are not a power-of-two number of periods): the pairs as they come, every block.
constexpr int PG_MAXPAT = 64;             // C <= 6: up to 7 blocks fit the nibbles of a word beside their count

// `group` points at the engine's "a table value is not finite" word: while it is set nothing is left out (0 * inf = NaN in
// the literal z), every pair goes to the all-blocks bucket.  So does a dish whose weights sum to 0 or NaN: its dish-vector
// blocks are (1 - a) m_c RE[d] / n = NaN even where m_c = 0 (with an ingredient table block 0 is finite, so the score
// is not NaN for some other reason), and the ungrouped kernels return NaN for it.
__device__ __forceinline__ int pg_pattern(const float *cats, int C, int64_t I, int32_t did, const int32_t *group)
{
    if (*group != 0 || did < 0 || (int64_t)did >= I) return (1 << C) - 1;      // bad ids: any bucket (the gatherer reports them)
    int pat = 0;
    float n = 0.f;
    for (int c = 0; c < C; ++c) {
        const float m = cats[(size_t)did * C + c];
        n += m;
        pat |= (m != 0.f ? 1 : 0) << c;
    }
    return (n == 0.f || n != n) ? (1 << C) - 1 : pat;
}

// the per-dish pattern byte, once per mask table: what m2d_mlp_pg_hist / _scatter read per pair instead of the dish's C mask weights
__global__ __launch_bounds__(256) void m2d_mlp_pg_pat8(const float *cats, int C, int64_t I, const int32_t *nogroup, uint8_t *pat8)
{
    const int32_t zero = 0;
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d < I) pat8[d] = (uint8_t)pg_pattern(cats, C, I, (int32_t)d, &zero);
    (void)nogroup;
}

// a pair's bucket: the dish's pattern byte; every block while a table value is not finite (`group`, see pg_pattern) or for a bad id
__device__ __forceinline__ int pg_bucket(const uint8_t *pat8, int C, int64_t I, int32_t did, int nogroup)
{
    return (nogroup || did < 0 || (int64_t)did >= I) ? (1 << C) - 1 : (int)pat8[did];
}

__global__ __launch_bounds__(256) void m2d_mlp_pg_hist(const int32_t *items, int64_t B, int64_t I, const uint8_t *pat8, int C,
                                                       const int32_t *group, int32_t *hist)
{
    // lane p of a wave counts pattern p in a register (one ballot + popcount per pattern and 64 pairs: no contended
    // atomics on a dozen addresses), then one LDS add and one global add per pattern and block
    __shared__ int32_t sh[PG_MAXPAT];
    const int lane = threadIdx.x & 63, npat = 1 << C;
    const int nogroup = *group != 0;
    if (threadIdx.x < PG_MAXPAT) sh[threadIdx.x] = 0;
    __syncthreads();
    int32_t cnt = 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < B; i0 += step) {   // block-uniform trip count: every ballot is full-wave
        const int64_t i = i0 + threadIdx.x;
        const int pat = i < B ? pg_bucket(pat8, C, I, items[i], nogroup) : -1;
        for (int q = 1; q < npat; ++q) {                                 // (pattern 0 does not occur: an empty mask takes every block)
            const unsigned long long b = __ballot(pat == q);
            if (lane == q) cnt += __popcll(b);
        }
    }
    if (lane < npat && cnt) atomicAdd(&sh[lane], cnt);
    __syncthreads();
    if (threadIdx.x < PG_MAXPAT && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}

// one block: bucket starts (whole tiles), each tile's block word, the tile count; hist becomes the scatter cursors
__global__ __launch_bounds__(256) void m2d_mlp_pg_scan(int32_t *hist, int C, uint32_t *tile_blocks, int32_t *ntiles)
{
    __shared__ int32_t start[PG_MAXPAT + 1];
    const int npat = 1 << C;
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int q = 0; q < npat; ++q) {
            start[q] = t;
            t += (hist[q] + PC_PAIRS - 1) / PC_PAIRS;
        }
        start[npat] = t;
        ntiles[0] = t;
    }
    __syncthreads();
    for (int q = 0; q < npat; ++q) {
        uint32_t w = 0;                                                   // block 0 (the high-level E values) always
        int nb = 1;
        for (int c = 0; c < C; ++c)
            if ((q >> c) & 1) w |= (uint32_t)(c + 1) << (4 * nb++);
        w |= (uint32_t)nb << 28;
        for (int t = start[q] + threadIdx.x; t < start[q + 1]; t += 256) tile_blocks[t] = w;
    }
    __syncthreads();
    if (threadIdx.x < npat) hist[threadIdx.x] = start[threadIdx.x] * PC_PAIRS;
}

// chunks of 4096 pairs: count per pattern in LDS, reserve each pattern's range with ONE global add per chunk, place.  (Round 6
// tried ranking by ballot instead of the per-pair LDS atomics -- a wave's lanes of one pattern behind one add of their leader:
// 97 us against 88 for 4 M pairs, the 240 dependent ballot / add / readlane rounds per wave cost more than the contention.)
__global__ __launch_bounds__(256) void m2d_mlp_pg_scatter(const int32_t *items, int64_t B, int64_t I, const uint8_t *pat8, int C,
                                                          const int32_t *group, int32_t *cursor, int32_t *perm)
{
    __shared__ int32_t cnt[PG_MAXPAT], base[PG_MAXPAT];
    constexpr int CH = 4096;
    const int nogroup = *group != 0;
    for (int64_t c0 = (int64_t)blockIdx.x * CH; c0 < B; c0 += (int64_t)gridDim.x * CH) {
        if (threadIdx.x < PG_MAXPAT) cnt[threadIdx.x] = 0;
        __syncthreads();
        int pat[CH / 256];
#pragma unroll
        for (int r = 0; r < CH / 256; ++r) {
            const int64_t i = c0 + r * 256 + threadIdx.x;
            pat[r] = i < B ? pg_bucket(pat8, C, I, items[i], nogroup) : -1;
            if (pat[r] >= 0) atomicAdd(&cnt[pat[r]], 1);
        }
        __syncthreads();
        if (threadIdx.x < PG_MAXPAT) {
            const int n = cnt[threadIdx.x];
            base[threadIdx.x] = n ? atomicAdd(&cursor[threadIdx.x], n) : 0;
            cnt[threadIdx.x] = 0;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CH / 256; ++r) {
            const int64_t i = c0 + r * 256 + threadIdx.x;
            if (pat[r] >= 0) perm[base[pat[r]] + atomicAdd(&cnt[pat[r]], 1)] = (int32_t)i;
        }
        __syncthreads();
    }
}

#if M2D_MLP_DIAG & 64
// diag bit 6: block 0 logs every wave's arrival at and release from its first 128 barriers
#define pc_barrier() do { unsigned long long ta_, tr_; MSTAMP(ta_); asm volatile("s_barrier" ::: "memory"); MSTAMP(tr_); \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && nbar_ < 128 && p.dbg) { unsigned long long *d_ = p.dbg + 4096 * 8 + ((threadIdx.x >> 6) * 128 + nbar_) * 2; d_[0] = ta_; d_[1] = tr_; } ++nbar_; } while (0)
#else
__device__ __forceinline__ void pc_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}
#endif

// Cross-lane sums without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): an LDS round trip
// per step, queued behind the consumers' ds_read_b128 streams.  The gatherers' 24 of them (8 rows x 3 steps, serial) were the
// 5 000-cycle barrier interval of every tile in round 6's arrival trace (profiles/r06_mlp_slack.txt, bars 7 / 17 / 27 ...) --
// tables in L2 or not.  DPP / permlane swaps stay in the VALU; the sums associate as before (same bits).
template <int CTRL>
__device__ __forceinline__ float pc_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the 8 lanes 8 r .. 8 r + 7 of a row, in every one of them: lane ^ 1, lane ^ 2 (quad_perm), then lane 7 - i of the
// half row (row_half_mirror: the other quad's sum, which is the same value in each of its lanes)
__device__ __forceinline__ float pc_sum8(float b)
{
    b += pc_dpp<0xB1>(b);                                   // quad_perm [1, 0, 3, 2]
    b += pc_dpp<0x4E>(b);                                   // quad_perm [2, 3, 0, 1]
    b += pc_dpp<0x141>(b);                                  // row_half_mirror
    return b;
}
// x[lane] + x[lane ^ 16], x[lane] + x[lane ^ 32]: v_permlane16_swap / v_permlane32_swap of a value with itself leave the row's
// (half's) own value in one result and its partner's in the other
__device__ __forceinline__ float pc_add_xor16(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ float pc_add_xor32(float x)
{
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

template <int N>
__device__ __forceinline__ void pc_wait_vmem()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef float v2f_pc __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_pc __attribute__((ext_vector_type(2)));

// OFF32: both gathered tables are under 4 GiB -- a row's offset is a 32-bit BYTE offset and the row loads take the form
// global_load_dwordx4 v, v_off, s[base:base+1] (the period's k-block in the scalar base): no address arithmetic in the gatherers,
// whose every vector instruction waits for an issue slot beside a consumer's MFMAs (40 of ~130 per period were address moves
// and 64-bit shift-adds).  Larger tables (up to 64 GiB): offsets in units of 16 B, one 64-bit shift-add per load.
template <int KCH, bool OFF32>
__global__ __launch_bounds__(512) void m2d_mlp_pc(MlpArgs p)
{
    extern __shared__ __align__(16) unsigned char pcs[];
    float *sbase = reinterpret_cast<float *>(pcs + PC_BASE_OFF);        // [tile parity 2][4 consumers][32]
    float *sb1 = reinterpret_cast<float *>(pcs + PC_B1_OFF);            // [256]
    float *sb2 = sb1 + MH1, *sw3 = sb2 + MH2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int K = KCH * 64, NH = 2 * KCH;
    // tiles of one mask pattern each (m2d_mlp_pg_scan); without grouping (perm == null): the pairs as they come, every block
    const int64_t ntiles = p.perm ? (int64_t)__builtin_amdgcn_readfirstlane(p.ntiles_dev[0]) : (p.B + PC_PAIRS - 1) / PC_PAIRS;
    if ((int64_t)blockIdx.x >= ntiles) return;                            // the grid is sized for the most tiles B pairs can make
    const int pshift = p.pshift;
    // a tile keeps the k-blocks of its pattern: its periods are kmap(0 .. nper - 1), then the two W2 stages NH, NH + 1
    auto tile_word = [&](int64_t tile) {
        return p.perm ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.tile_blocks[tile])
                      : (0x06543210u | ((uint32_t)(NH >> pshift) << 28));             // every block: nibble j = j, count = K / E
    };
    // The NEXT tile's word, requested a tile ahead (consumers, gatherers): as tile_word() reads it -- a scalar load at the head of
    // the tile, waited for on the spot -- every tile began with a dependent round trip to L2 / HBM in each role (round 6's arrival
    // trace: the barrier intervals after a tile boundary).  A VECTOR load (the index made opaque, so that it is not scalarised: a
    // pending scalar load would turn every LDS wait into lgkmcnt(0)), kept raw in a VGPR and made uniform where it is used.
    int pc_zero_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(pc_zero_));
    auto tile_word_raw = [&](int64_t tile) -> uint32_t {
        return p.perm ? p.tile_blocks[tile + pc_zero_] : (0x06543210u | ((uint32_t)(NH >> pshift) << 28));
    };
    auto tile_word_ready = [&](uint32_t raw) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)raw); };
    auto nper_of = [&](uint32_t w) { return (int)(w >> 28) << pshift; };
    auto kmap = [&](uint32_t w, int q) { return (int)(((w >> (4 * (q >> pshift))) & 15u) << pshift) | (q & ((1 << pshift) - 1)); };
#if M2D_MLP_DIAG & 64
    int nbar_ = 0;
#endif

    for (int i = threadIdx.x; i < MH1; i += 512) sb1[i] = p.b1[i];
    if (threadIdx.x < MH2) {
        sb2[threadIdx.x] = p.b2[threadIdx.x];
        sw3[threadIdx.x] = p.w3[threadIdx.x];
    }
    __syncthreads();

    if (wave < 4) {
        // ======================================= consumer =======================================
        // v_mfma_f32_16x16x32_bf16: the wave's 256 hidden units x 32 pairs as 16 x 2 tiles of 16 x 16 (lane: column c = lane & 15,
        // row group g = lane >> 4 -- an A / B fragment holds k = 8 g .. 8 g + 7 of a 32-deep step, a C / D tile rows 4 g .. 4 g + 3).
        // Round 5: the head is the one kernel here that the chip clocks down (1.98 GHz on v_mfma_f32_32x32x16_bf16); this shape
        // holds 2.13 GHz at the same output tile per wave, LDS reads and registers (MI355X_MICROARCH.md "DVFS give-back" 7).
        const int c16 = lane & 15, g4 = lane >> 4;
#if M2D_MLP_DIAG
        unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, n_t = 0, t0_ = 0, t1_;
#endif
        // this lane's 16 B of a W1 / W2 image row (16 rows x 64 B per 1-KiB tile; slot g swizzled by the row: see the image kernels)
        const unsigned a_off = c16 * 64 + ((g4 ^ ((c16 >> 3) << 1)) << 4);
        // z set: [blk = k-block of 8 = g][k-values 0-3 | 4-7 of the lane's 8][pair slot ^ (2 blk + part)][16 B]
        unsigned z_off[2][2];                                             // [pair tile ct][part]
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int part = 0; part < 2; ++part)
                z_off[ct][part] = wave * 4096u + g4 * 1024u + part * 512u + (((16 * ct + c16) ^ (2 * g4 + part)) << 4);
        unsigned hb = 0, zb = 0;                                          // half-buffer of this period's first half; z set
        bf16x8 ah[4], al[4], bA[2][2], bB[2][2];                          // b?[pair tile][hi, lo]
        v4f zraw[4];
        auto hb_add = [](unsigned x, unsigned d) { const unsigned y = x + d; return y >= PC_NHB ? y - PC_NHB : y; };
        // hidden tile nt (16 units) of a half (128 units): [hi: 8 tiles x 1 KiB | lo: the same]
        auto frag = [&](unsigned half_off, int nt, bf16x8 &fh, bf16x8 &fl) __attribute__((always_inline)) {
            const unsigned char *q = pcs + half_off + nt * 1024 + a_off;
            fh = *reinterpret_cast<const bf16x8 *>(q);
            fl = *reinterpret_cast<const bf16x8 *>(q + PC_HALF / 2);
        };
        // The reference score's part of the output, sum_k z[k] (= the score Model_Recommender.py:95-96 returns), is summed HERE since round 6:
        // every z value passes through exactly one consumer lane on its way into the B operand, four adds per chunk in the MFMAs'
        // shadow.  The gatherers used to keep it (16 packed adds per period and, at a tile's last step, eight 8-lane sums and LDS
        // writes): a memory wave gets a vector issue slot about every 16 cycles beside its SIMD-mate's MFMAs, the last step's
        // sums alone held every tile's barrier for 4 200 cycles (round 6's per-step stamps, profiles/r06_mlp_slack.txt).
        float bs[2] = {0.f, 0.f};                                         // this lane's k-values of pairs c16 and 16 + c16
        // chunk ch = 2 ct + part: 4 of the next period's z values -> their place in the split B operand of pair tile ct; `acc`
        // (wave-uniform): they belong to a period of a tile (after a tile's last layer-1 period the z set read is a stale one)
        auto split = [&](int ch, bf16x8 (&nb)[2][2], const bool acc) __attribute__((always_inline)) {
            const v4f zz = zraw[ch];
            const float zs4 = (zz.x + zz.y) + (zz.z + zz.w);
            bs[ch >> 1] += acc ? zs4 : 0.f;
            const v2f_pc z01 = {zz.x, zz.y}, z23 = {zz.z, zz.w};
            const uint32_t h01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(z01, bf16x2_pc));
            const uint32_t h23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(z23, bf16x2_pc));
            const v2f_pc f01 = {__builtin_bit_cast(float, h01 << 16), __builtin_bit_cast(float, h01 & 0xffff0000u)};
            const v2f_pc f23 = {__builtin_bit_cast(float, h23 << 16), __builtin_bit_cast(float, h23 & 0xffff0000u)};
            const bf16x2_pc l01 = __builtin_convertvector(z01 - f01, bf16x2_pc), l23 = __builtin_convertvector(z23 - f23, bf16x2_pc);
            const bf16x2_pc g01 = __builtin_bit_cast(bf16x2_pc, h01), g23 = __builtin_bit_cast(bf16x2_pc, h23);
            const int ct = ch >> 1, o = 4 * (ch & 1);
            nb[ct][0][o] = g01.x; nb[ct][0][o + 1] = g01.y; nb[ct][0][o + 2] = g23.x; nb[ct][0][o + 3] = g23.y;
            nb[ct][1][o] = l01.x; nb[ct][1][o + 1] = l01.y; nb[ct][1][o + 2] = l23.x; nb[ct][1][o + 3] = l23.y;
        };
        auto zread = [&](unsigned zset) __attribute__((always_inline)) {
            const unsigned char *q = pcs + PC_Z_OFF + zset * PC_ZSET;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) zraw[ch] = *reinterpret_cast<const v4f *>(q + z_off[ch >> 1][ch & 1]);
        };
        v4f acc1[16][2];                                                  // [hidden tile][pair tile]
        // one layer-1 period (32 k-values = one MFMA step): 16 hidden tiles x 2 pair tiles x 3 MFMAs; fragments two tiles ahead
        // (4 register sets), the barrier after the first half's 8 tiles (hidden units 0 .. 127: that half-stage is then done with),
        // the NEXT period's operands read and split under the last tiles
        auto period = [&](bf16x8 (&b)[2][2], bf16x8 (&nb)[2][2], const bool acc_next) __attribute__((always_inline)) {
            const unsigned h0 = hb * PC_HALF, h1 = hb_add(hb, 1) * PC_HALF, h2 = hb_add(hb, 2) * PC_HALF;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                if (it == 8) { MACC(t_a); pc_barrier(); MACC(t_b); }
                if (M2D_MLP_DIAG & 256) continue;                         // diag bit 8 (timing only): the memory waves alone
                const int jt = it + 2;
                frag(jt < 8 ? h0 : jt < 16 ? h1 : h2, jt & 7, ah[jt & 3], al[jt & 3]);
                if (it == 10) zread(zb ^ 1);
                if (it >= 12) split(it - 12, nb, acc_next);
                if (M2D_MLP_DIAG & 512) {                                  // diag bit 9 (timing only): LDS reads and the split, no MFMA
                    asm volatile("" ::"v"(ah[it & 3]), "v"(al[it & 3]), "v"(b[0][0]), "v"(b[0][1]), "v"(b[1][0]), "v"(b[1][1]));
                    if (M2D_MLP_DIAG & 1024) {                             // + bit 10: idle for about the six MFMAs' wall time
                        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc1[it][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[it & 3], b[ct][0], acc1[it][ct], 0, 0, 0);
                    acc1[it][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[it & 3], b[ct][1], acc1[it][ct], 0, 0, 0);
                    acc1[it][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[it & 3], b[ct][0], acc1[it][ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            hb = hb_add(hb, 2);
            zb ^= 1;
        };
        unsigned tpar = 0;                                                // parity of this block's tile count
        pc_barrier();                                                     // period 0 is published
        MSTAMP(t0_);
#if M2D_MLP_DIAG
        const unsigned long long clk0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
        frag(0, 0, ah[0], al[0]);
        frag(0, 1, ah[1], al[1]);
        zread(0);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) split(ch, bA, true);
        uint32_t tw_next = tile_word_raw(blockIdx.x);
        for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const uint32_t tw_c = tile_word_ready(tw_next);
            if (tile + gridDim.x < ntiles) tw_next = tile_word_raw(tile + gridDim.x);
            // where this lane's scores go (lanes 0-15: pairs c and 16 + c of the wave's 32): requested now, needed after layer 3
            // (-1: a padding slot)
            int32_t pi[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int64_t slot = tile * PC_PAIRS + wave * 32 + 16 * ct + c16;
                pi[ct] = p.perm ? p.perm[slot] : (slot < p.B ? (int32_t)slot : -1);
            }
#pragma unroll
            for (int nt = 0; nt < 16; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[nt][0][r] = acc1[nt][1][r] = sb1[16 * nt + 4 * g4 + r];
            const int nper = nper_of(tw_c);                               // even (blocks of >= 2 periods, or every block)
#pragma unroll 1
            for (int kc = 0; kc < nper / 2; ++kc) {
                period(bA, bB, true);
                period(bB, bA, kc + 1 < nper / 2);                        // the tile's last period: no next z of this tile
            }
            // ---- layer 2: relu(acc1) split hi / lo is the B operand -- a step's 32 hidden units are accumulator tiles 2 s and
            // 2 s + 1, rows 4 g .. 4 g + 3 of each (the W2 image holds its k-values in that order); W2 is ring stages NH, NH + 1 ----
            MACC(t_a);
            v4f acc2[4][2];                                               // [h2 tile][pair tile]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[mt][0][r] = acc2[mt][1][r] = sb2[16 * mt + 4 * g4 + r];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if ((s & 3) == 2) { MACC(t_c); pc_barrier(); MACC(t_b); }
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float hv = fmaxf(acc1[2 * s + (j >> 2)][ct][j & 3], 0.f);
                        const __bf16 hi = (__bf16)hv;
                        bh[ct][j] = hi;
                        bl[ct][j] = (__bf16)(hv - (float)hi);
                    }
                // half (s >> 1) & 1 of the stage: [step s & 1: 8 KiB][h2 tile mt: 2 KiB][hi 1 KiB | lo 1 KiB]
                const unsigned char *img = pcs + hb_add(hb, (s >> 1) & 1) * PC_HALF + (s & 1) * 8192 + a_off;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16x8 wh = *reinterpret_cast<const bf16x8 *>(img + mt * 2048);
                    const bf16x8 wl = *reinterpret_cast<const bf16x8 *>(img + mt * 2048 + 1024);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        acc2[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh[ct], acc2[mt][ct], 0, 0, 0);
                        acc2[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl[ct], acc2[mt][ct], 0, 0, 0);
                        acc2[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh[ct], acc2[mt][ct], 0, 0, 0);
                    }
                }
                if ((s & 3) == 3) { hb = hb_add(hb, 2); zb ^= 1; }
            }
            // the next tile's period 0 was published by the last barrier
            MACC(t_c);
            frag(hb * PC_HALF, 0, ah[0], al[0]);
            frag(hb * PC_HALF, 1, ah[1], al[1]);
            zread(zb);
            const float bfin[2] = {bs[0], bs[1]};                         // this tile's sums; the next tile's start below
            bs[0] = bs[1] = 0.f;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) split(ch, bA, true);
            // ---- layer 3 and the reference score (its sum over this lane's k-values: bfin; the gatherers' word is 0, or NaN = an
            // id was out of range) ----
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float o = 0.f;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = fmaf(sw3[16 * mt + 4 * g4 + r], fmaxf(acc2[mt][ct][r], 0.f), o);
                const float base = pc_add_xor32(pc_add_xor16(bfin[ct]));
                o = pc_add_xor32(pc_add_xor16(o));
                if (g4 == 0 && pi[ct] >= 0) p.out[pi[ct]] = (sbase[tpar * 128 + wave * 32 + 16 * ct + c16] + base) + (o + p.b3);
            }
            tpar ^= 1;
#if M2D_MLP_DIAG
            MACC(t_d); ++n_t;
#endif
        }
#if M2D_MLP_DIAG
        if (lane == 0 && p.dbg) {
            unsigned long long *d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = t_d; d[4] = n_t;
            d[5] = __builtin_amdgcn_s_memtime() - clk0_; d[6] = __builtin_amdgcn_s_memrealtime() - rt0_;
        }
#endif
        return;
    }

    if (wave & 1) {
        // ======================================= loader (waves 5, 7) =======================================
        // half li of every ring stage: 16 pieces of 1 KiB (hi 8, lo 8), two periods ahead of its use.  The stage's
        // base sits in an SGPR descriptor, the lane offset in one VGPR, the piece in the scalar offset.
        const int li = (wave - 5) >> 1;
        const unsigned char *img = reinterpret_cast<const unsigned char *>(p.w1x3) + li * PC_HALF;
        const int voff = lane * 16;
        unsigned hbn = li;                                                // half-buffer of this wave's half of the stage being fetched
#if M2D_MLP_DIAG
        unsigned long long t_a = 0, t_b = 0, t_c = 0, n_t = 0, t0_, t1_;
#endif
        auto dma = [&](int q) __attribute__((always_inline)) {
            if (!(M2D_MLP_DIAG & 4)) {                                    // diag bit 2: no DMA
                const uint64_t sb = (uint64_t)(uintptr_t)(img + (size_t)q * PC_STAGE);
                v4i_pc rsrc;
                rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)sb);
                rsrc.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(sb >> 32) & 0xffffu));
                rsrc.z = PC_HALF;
                rsrc.w = 0x00020000;
                unsigned char *dst = pcs + hbn * PC_HALF;
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) {
                    const uint32_t m0v = __builtin_amdgcn_readfirstlane(
                        (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)(dst + pc * 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 ::"s"(m0v), "v"(voff), "s"(rsrc), "s"(pc * 1024) : "memory", "m0");
                }
            }
            hbn = hbn + 2 >= PC_NHB ? hbn + 2 - PC_NHB : hbn + 2;
        };
        dma(0);
        MSTAMP(t0_);
        for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#if M2D_MLP_DIAG
            ++n_t;
#endif
            const uint32_t tw = tile_word(tile);
            const int nper = nper_of(tw);
#pragma unroll 1
            for (int q = 0; q < nper + 2; ++q) {
                // the stage after the one about to be published: this tile's next period, W2's two halves, the next
                // tile's first period (block 0 of every pattern)
                dma(q + 1 < nper ? kmap(tw, q + 1) : q + 1 == nper ? NH : q + 1 == nper + 1 ? NH + 1 : 0);
                MACC(t_a);
                pc_wait_vmem<16>();                                       // stage q landed; the one just requested stays in flight
                MACC(t_b);
                pc_barrier();
                MACC(t_c);
            }
        }
        pc_wait_vmem<0>();
        pc_barrier();
#if M2D_MLP_DIAG
        if (lane == 0 && p.dbg) {
            unsigned long long *d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = t_a; d[1] = t_b; d[2] = t_c; d[4] = n_t;
        }
#endif
        return;
    }

    // ======================================= gatherer (waves 4, 6) =======================================
    {
        const int g = (wave - 4) >> 1;                                    // serves consumers 2 g, 2 g + 1: pairs [64 g, 64 g + 64)
        const int r8 = lane >> 3, s = lane & 7;
        // this lane's 16 bytes of z: block s >> 1, part s & 1, pair slot swizzled by s
        const unsigned z_w = (2 * g) * 4096 + (s >> 1) * 1024 + (s & 1) * 512 + ((r8 ^ s) << 4);
        const unsigned char *pm = reinterpret_cast<const unsigned char *>(p.pm);
        const unsigned char *dt = reinterpret_cast<const unsigned char *>(p.dt);
        // rows as 32-bit offsets in units of 16 B, this lane's slot included: (row * K / 4 + s) -- the launcher takes
        // this kernel only for tables under 64 GiB.  One 64-bit shift-add per load turns one into an address.
        uint32_t cu[8], cd[8], nu[8], nd[8];                              // this tile / the next one
        unsigned badmask = 0, nbadmask = 0;
        v4f ra[2][8], rb_[2][8];
        int32_t npi[8];                                                   // pair index of each slot of the tile being converted
        auto load_ids = [&](int64_t tile) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t slot = tile * PC_PAIRS + 64 * g + 8 * i + r8;
                npi[i] = p.perm ? p.perm[slot] : (slot < p.B ? (int32_t)slot : -1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int32_t pi = npi[i] >= 0 ? npi[i] : 0;              // padding slot: any valid pair; masked when converted
                nu[i] = (uint32_t)p.users[pi];
                nd[i] = (uint32_t)p.items[pi];
            }
        };
        auto convert_ids = [&](int64_t) __attribute__((always_inline)) {
            nbadmask = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t pi = npi[i];
                const bool valid = pi >= 0;
                const int32_t uid = (int32_t)nu[i], did = (int32_t)nd[i];
                int64_t ul = (int64_t)uid - p.user_base;
                int32_t dl = did;
                if (valid && (ul < 0 || ul >= p.U)) { if (s == 0) latch(p.err, M2D_ERR_BAD_USER_ID, uid, pi); nbadmask |= 1u << i; }
                if (valid && (did < 0 || (int64_t)did >= p.I)) { if (s == 0) latch(p.err, M2D_ERR_BAD_ITEM_ID, did, pi); nbadmask |= 1u << i; }
                if (!valid || ul < 0 || ul >= p.U) ul = 0;
                if (!valid || did < 0 || (int64_t)did >= p.I) dl = 0;
                nu[i] = ((uint32_t)ul * (uint32_t)(K / 4) + (uint32_t)s) * (OFF32 ? 16u : 1u);
                nd[i] = ((uint32_t)dl * (uint32_t)(K / 4) + (uint32_t)s) * (OFF32 ? 16u : 1u);
            }
        };
        auto gather = [&](int set, const uint32_t (&iu)[8], const uint32_t (&id)[8], int half) __attribute__((always_inline)) {
            if (M2D_MLP_DIAG & 8) return;                                 // diag bit 3: no row requests
            const unsigned char *bu = pm + half * 128, *bd = dt + half * 128;       // uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t ou = iu[i], od = id[i];
                if (!OFF32) asm volatile("" : "+v"(ou), "+v"(od));        // keep the 64-bit products out of registers
                // user rows are read about once per batch: non-temporal, so they do not push the dish vectors (each
                // read ~10 times) out of L2 / the Infinity Cache
                ra[set][i] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(bu + (uint64_t)ou * (OFF32 ? 1 : 16)));
                rb_[set][i] = *reinterpret_cast<const v4f *>(bd + (uint64_t)od * (OFF32 ? 1 : 16));
                __builtin_amdgcn_sched_barrier(0);                        // one pair of addresses live at a time
            }
        };
        unsigned zb = 0, tpar = 0;                                        // z set being written; parity of this block's tile count
        // per 4 k-values of a pair: 2 packed multiplies, one 16-byte store (the reference score's sum: in the consumers)
        auto build = [&](int set) __attribute__((always_inline)) {
            if (M2D_MLP_DIAG & 2) return;                                 // diag bit 1: no z build
            unsigned char *zs = pcs + PC_Z_OFF + zb * PC_ZSET + z_w;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<v4f *>(zs + (i >> 2) * 4096 + (i & 3) * 128) = ra[set][i] * rb_[set][i];
        };
        // prologue: offsets of this block's first two tiles; the first two periods' rows
        const int64_t tile0 = blockIdx.x;
        auto next_of = [&](int64_t t) { return t + gridDim.x < ntiles ? t + gridDim.x : t; };   // none left: any valid rows
        load_ids(tile0);
        pc_wait_vmem<0>();
        convert_ids(tile0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { cu[i] = nu[i]; cd[i] = nd[i]; }
        badmask = nbadmask;
        load_ids(next_of(tile0));
        pc_wait_vmem<0>();
        convert_ids(next_of(tile0));
        gather(0, cu, cd, 0);
        gather(1, cu, cd, 1);
#if M2D_MLP_DIAG
        unsigned long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, n_t = 0, t0_, t1_;
        MSTAMP(t0_);
#endif
        uint32_t tw_next = tile_word_raw(tile0);
        for (int64_t tile = tile0; tile < ntiles; tile += gridDim.x) {
#if M2D_MLP_DIAG
            ++n_t;
#endif
            const uint32_t tw = tile_word_ready(tw_next);
            if (tile + gridDim.x < ntiles) tw_next = tile_word_raw(tile + gridDim.x);
            // producing z of period q: build it from the rows requested two steps ago (ordinary loads: the compiler
            // places the vmcnt waits, and in this wave it sees every outstanding request), request the rows two
            // periods on, publish
            auto step = [&](int set, const uint32_t (&iu)[8], const uint32_t (&id)[8], int half, bool last)
                            __attribute__((always_inline)) {
                MACC(t_d);
                GSTAMP(0);
                build(set);
                MACC(t_a);
                GSTAMP(1);
                gather(set, iu, id, half);
                asm volatile("" ::: "memory");
                MACC(t_b);
                GSTAMP(2);
                if (last) {
                    // the pair's word beside the consumers' sum: 0, or NaN when one of its ids was out of range.  One store per lane:
                    // lane (r8, s) writes the word of slot i = s (a row's eight lanes hold the same badmask)
                    sbase[tpar * 128 + (2 * g + (s >> 2)) * 32 + 8 * (s & 3) + r8] = ((badmask >> s) & 1) ? __builtin_nanf("") : 0.f;
                }
                GSTAMP(3);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MACC(t_c);
                GSTAMP(4);
                pc_barrier();
                zb ^= 1;
            };
#if M2D_MLP_DIAG & 16
            // diag bit 4 (timing only, wrong arithmetic): rows requested ONE period ahead
#pragma unroll 1
            for (int q = 0; q < nper_of(tw) - 2; q += 2) {
                step(0, cu, cd, q + 1, false);
                step(0, cu, cd, q + 2, false);
            }
#else
            const int nper = nper_of(tw);
#pragma unroll 1
            for (int q = 0; q < nper - 2; q += 2) {
                step(0, cu, cd, kmap(tw, q + 2), false);
                step(1, cu, cd, kmap(tw, q + 3), false);
            }
#endif
            step(0, nu, nd, 0, false);                                    // the last two request the next tile's rows
            step(1, nu, nd, 1, true);
            // the two W2 periods: nothing to build.  The offsets move up one tile; the ids of the tile after the next
            // are requested before the first of the two barriers and converted before the second.
#pragma unroll
            for (int i = 0; i < 8; ++i) { cu[i] = nu[i]; cd[i] = nd[i]; }
            badmask = nbadmask;
            tpar ^= 1;
            const int64_t t2 = next_of(next_of(tile));
            GSTAMP(0);
            load_ids(t2);
            GSTAMP(1);
            GSTAMP(4);
            pc_barrier();
            zb ^= 1;
            GSTAMP(0);
            pc_wait_vmem<0>();
            GSTAMP(1);
            convert_ids(t2);
            GSTAMP(2);
            GSTAMP(4);
            pc_barrier();
            zb ^= 1;
        }
        pc_wait_vmem<0>();
        pc_barrier();
#if M2D_MLP_DIAG
        if (lane == 0 && p.dbg) {
            unsigned long long *d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 8;
            d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = t_d; d[4] = n_t;
        }
#endif
    }
}

// (diag epilogue is emitted by the macro below, inside the kernel)
// Any K / H1 / H2: one wave per pair, activations in LDS.  Slow; for shapes the MFMA kernel does not cover.
__global__ __launch_bounds__(256) void m2d_mlp_generic(MlpArgs p)
{
    extern __shared__ __align__(16) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float *z = smem + (size_t)wave * (p.K + p.H1 + p.H2);
    float *h1 = z + p.K, *h2 = h1 + p.H1;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t pi = (int64_t)blockIdx.x * 4 + wave; pi < p.B; pi += nw) {
        int32_t uid = p.users[pi], did = p.items[pi];
        int64_t ul = (int64_t)uid - p.user_base;
        bool bad = false;
        if (ul < 0 || ul >= p.U) { if (lane == 0) latch(p.err, M2D_ERR_BAD_USER_ID, uid, pi); ul = 0; bad = true; }
        if (did < 0 || (int64_t)did >= p.I) { if (lane == 0) latch(p.err, M2D_ERR_BAD_ITEM_ID, did, pi); did = 0; bad = true; }
        float base = 0.f;
        for (int k = lane; k < p.K; k += 64) {
            const float v = p.pm[(size_t)ul * p.K + k] * p.dt[(size_t)did * p.K + k];
            z[k] = v;
            base += v;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) base += __shfl_xor(base, off, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int n = lane; n < p.H1; n += 64) {
            float s = 0.f;
            for (int k = 0; k < p.K; ++k) s = fmaf(z[k], p.w1[(size_t)k * p.H1 + n], s);
            h1[n] = fmaxf(s + p.b1[n], 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int m = lane; m < p.H2; m += 64) {
            float s = 0.f;
            for (int n = 0; n < p.H1; ++n) s = fmaf(h1[n], p.w2[(size_t)n * p.H2 + m], s);
            h2[m] = fmaxf(s + p.b2[m], 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float o = 0.f;
        for (int m = lane; m < p.H2; m += 64) o = fmaf(p.w3[m], h2[m], o);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) o += __shfl_xor(o, off, 64);
        if (lane == 0) p.out[pi] = bad ? __builtin_nanf("") : (base + (o + p.b3));
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

int m2d_launch_score_pairs_mlp(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B, float *out,
                               hipStream_t stream)
{
    if (B == 0) return M2D_OK;
    int rc = m2d_ensure_finite_scan(h, stream);
    if (rc != M2D_OK) return rc;
    if ((rc = m2d_ensure_dish_vectors(h, stream)) != M2D_OK) return rc;
    MlpArgs a;
    a.pm = h->pm; a.dt = h->dish_vec; a.w1 = h->mlp_w1; a.b1 = h->mlp_b1; a.w2 = h->mlp_w2; a.b2 = h->mlp_b2;
    a.w3 = h->mlp_w3; a.b3 = h->mlp_b3; a.users = users; a.items = items; a.out = out;
    a.B = B; a.U = h->U; a.I = h->I; a.user_base = h->user_base;
    a.K = (h->C + 1) * h->E; a.H1 = h->mlp_h1; a.H2 = h->mlp_h2; a.err = h->err_dev;
    a.dbg = g_m2d_mlp_diag_buffer;
    a.w1x3 = a.w2x3 = nullptr;
    a.perm = nullptr; a.tile_blocks = nullptr; a.ntiles_dev = nullptr; a.pshift = 0;
    const bool mfma_ok = a.H1 == MH1 && a.H2 == MH2 && a.K % 64 == 0 && h->opt_variant != 9;
    const int kch = a.K / 64;
    int pad_kch = 0;                          // K % 64 != 0: chunks of the zero-padded form (0 = none instantiated)
    if (a.H1 == MH1 && a.H2 == MH2 && a.K % 64 != 0 && a.K % 4 == 0 && h->opt_variant != 9) {
        const int need = (a.K + 63) / 64;
        pad_kch = need <= 5 ? 5 : (need <= 10 ? 10 : (need <= 16 ? 16 : (need <= 20 ? 20 : 0)));
    }
    const bool pc_ok = (uint64_t)h->U * a.K * 4 < (1ull << 36) && (uint64_t)h->I * a.K * 4 < (1ull << 36) &&   // 32-bit row offsets in 16-B units
                       h->E % 32 == 0 && ((h->E / 32) & (h->E / 32 - 1)) == 0 && h->C <= 6 && h->dish_cats &&      // k-blocks of whole periods
                       B < (1ll << 31) - (1 << 16);                                                             // 32-bit pair slots
    if (mfma_ok && (kch == 5 || kch == 10 || kch == 20 || kch == 3) && h->opt_mlp_bf16x3 != 0 && h->opt_mlp_form == 0 && pc_ok) {
        // producer / consumer form: its own image of W1 | W2, (2 kch + 2) ring stages of 32 KiB, built once per head
        if (!h->mlp_w1pc) {
            M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_w1pc, (size_t)(2 * kch + 2) * PC_STAGE));
            hipLaunchKernelGGL(m2d_mlp_image_pc_w1, dim3((unsigned)(((int64_t)a.K * MH1 + 255) / 256)), dim3(256), 0, stream,
                               h->mlp_w1, a.K, reinterpret_cast<__bf16 *>(h->mlp_w1pc));
            hipLaunchKernelGGL(m2d_mlp_image_pc_w2, dim3(MH1 * MH2 / 256), dim3(256), 0, stream, h->mlp_w2,
                               reinterpret_cast<__bf16 *>(h->mlp_w1pc) + (size_t)(2 * kch) * (PC_STAGE / 2));
            M2D_HIP_TRY(h, hipGetLastError());
        }
        a.w1x3 = reinterpret_cast<const __bf16 *>(h->mlp_w1pc);
        // pairs bucketed by the dish's pattern of non-zero mask weights; every bucket padded to whole tiles
        const int npat = 1 << h->C;
        a.pshift = __builtin_ctz((unsigned)(h->E / 32));
        // E = 32: a block is one period, keep every block; small batches: three more launches and mostly-empty tiles
        // cost more than the skipped periods save
        const bool group = h->opt_skip_masked != 0 && a.pshift >= 1 && B >= 16384;
        int64_t tiles_max = (B + PC_PAIRS - 1) / PC_PAIRS;
        if (group) {
            tiles_max += npat;
            const size_t need = (size_t)(PG_MAXPAT + 4) + (size_t)tiles_max + (size_t)tiles_max * PC_PAIRS;
            if (need > h->mlp_pg_cap) {
                if (h->mlp_pg) M2D_HIP_TRY(h, hipFree(h->mlp_pg));
                h->mlp_pg = nullptr;
                h->mlp_pg_cap = 0;
                M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_pg, need * sizeof(int32_t)));
                h->mlp_pg_cap = need;
            }
            if (!h->mlp_pat8 || h->mlp_pat8_gen != h->dish_vec_gen || h->mlp_pat8_rows != h->I) {      // once per mask table
                if (h->mlp_pat8 && h->mlp_pat8_rows != h->I) { M2D_HIP_TRY(h, hipFree(h->mlp_pat8)); h->mlp_pat8 = nullptr; }
                if (!h->mlp_pat8) M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_pat8, (size_t)h->I));
                h->mlp_pat8_rows = h->I;
                hipLaunchKernelGGL(m2d_mlp_pg_pat8, dim3((unsigned)((h->I + 255) / 256)), dim3(256), 0, stream, h->dish_cats, h->C, h->I,
                                   h->nonfinite_dev, h->mlp_pat8);
                M2D_HIP_TRY(h, hipGetLastError());
                h->mlp_pat8_gen = h->dish_vec_gen;
            }
            int32_t *hist = h->mlp_pg, *ntl = hist + PG_MAXPAT, *perm = ntl + 4 + tiles_max;
            uint32_t *tblocks = reinterpret_cast<uint32_t *>(ntl + 4);
            M2D_HIP_TRY(h, hipMemsetAsync(hist, 0, (PG_MAXPAT + 4) * sizeof(int32_t), stream));
            M2D_HIP_TRY(h, hipMemsetAsync(perm, 0xFF, (size_t)tiles_max * PC_PAIRS * sizeof(int32_t), stream));
            const unsigned gcap = (unsigned)h->num_cu * 8;
            const int64_t hb = (B + 255) / 256, sb = (B + 4095) / 4096;
            hipLaunchKernelGGL(m2d_mlp_pg_hist, dim3((unsigned)(hb < gcap ? hb : gcap)), dim3(256), 0, stream, items, B, h->I,
                               h->mlp_pat8, h->C, h->nonfinite_dev, hist);
            hipLaunchKernelGGL(m2d_mlp_pg_scan, dim3(1), dim3(256), 0, stream, hist, h->C, tblocks, ntl);
            hipLaunchKernelGGL(m2d_mlp_pg_scatter, dim3((unsigned)(sb < gcap ? sb : gcap)), dim3(256), 0, stream, items, B, h->I,
                               h->mlp_pat8, h->C, h->nonfinite_dev, hist, perm);
            M2D_HIP_TRY(h, hipGetLastError());
            a.perm = perm; a.tile_blocks = tblocks; a.ntiles_dev = ntl;
        }
        const unsigned grid = (unsigned)(tiles_max < h->num_cu ? tiles_max : h->num_cu);
        // row offsets as 32-bit byte offsets where both gathered tables allow it (see the kernel's OFF32)
        // ("variant" 16: the 16-byte-unit offsets whatever the sizes -- the tests' way to run the large-table instantiation)
        const bool off32 = (uint64_t)h->U * a.K * 4 <= (1ull << 32) && (uint64_t)h->I * a.K * 4 <= (1ull << 32) && h->opt_variant != 16;
#define M2D_MLP_PC_CASE(N)                                                                                  \
    if (kch == N && off32) {                                                                                \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_mlp_pc<N, true>, PC_LDS_BYTES));                     \
        hipLaunchKernelGGL((m2d_mlp_pc<N, true>), dim3(grid), dim3(512), PC_LDS_BYTES, stream, a);          \
    } else if (kch == N) {                                                                                  \
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_mlp_pc<N, false>, PC_LDS_BYTES));                    \
        hipLaunchKernelGGL((m2d_mlp_pc<N, false>), dim3(grid), dim3(512), PC_LDS_BYTES, stream, a);         \
    }
        M2D_MLP_PC_CASE(3) M2D_MLP_PC_CASE(5) M2D_MLP_PC_CASE(10) M2D_MLP_PC_CASE(20)
#undef M2D_MLP_PC_CASE
        h->last_kernel = "m2d_mlp_pc_bf16x3";
    } else if (mfma_ok && (kch == 5 || kch == 10 || kch == 20 || kch == 3)) {
        const size_t lds = (size_t)(2 * RING_FLOATS + MH1 + 2 * MH2) * sizeof(float);
        const int64_t ntiles = (B + 32 * MWAVES - 1) / (32 * MWAVES);
        const unsigned grid = (unsigned)(ntiles < h->num_cu ? ntiles : h->num_cu);
        const bool x3 = h->opt_mlp_bf16x3 != 0;
        if (x3 && !h->mlp_w1x3) {          // built once per head (m2d_set_mlp_head resets it)
            M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_w1x3, (size_t)a.K * MH1 * 4 + (size_t)MH1 * MH2 * 4));
            hipLaunchKernelGGL(m2d_mlp_split_w1, dim3((unsigned)(((int64_t)a.K * MH1 + 255) / 256)), dim3(256), 0, stream,
                               h->mlp_w1, a.K, reinterpret_cast<__bf16 *>(h->mlp_w1x3));
            hipLaunchKernelGGL(m2d_mlp_split_w2, dim3(MH1 * MH2 / 256), dim3(256), 0, stream, h->mlp_w2,
                               reinterpret_cast<__bf16 *>(h->mlp_w1x3) + (size_t)a.K * MH1 * 2);
            M2D_HIP_TRY(h, hipGetLastError());
        }
        a.w1x3 = reinterpret_cast<const __bf16 *>(h->mlp_w1x3);
        a.w2x3 = a.w1x3 ? a.w1x3 + (size_t)a.K * MH1 * 2 : nullptr;
#define M2D_MLP_CASE(N)                                                                                     \
    if (kch == N) {                                                                                         \
        if (x3) {                                                                                           \
            M2D_HIP_TRY(h, hipFuncSetAttribute((const void *)m2d_mlp_mfma<N, true>,                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
            hipLaunchKernelGGL((m2d_mlp_mfma<N, true>), dim3(grid), dim3(MWAVES * 64), lds, stream, a);     \
        } else {                                                                                            \
            M2D_HIP_TRY(h, hipFuncSetAttribute((const void *)m2d_mlp_mfma<N, false>,                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
            hipLaunchKernelGGL((m2d_mlp_mfma<N, false>), dim3(grid), dim3(MWAVES * 64), lds, stream, a);    \
        }                                                                                                   \
    }
        M2D_MLP_CASE(3) M2D_MLP_CASE(5) M2D_MLP_CASE(10) M2D_MLP_CASE(20)
#undef M2D_MLP_CASE
        h->last_kernel = x3 ? "m2d_mlp_mfma_bf16x3" : "m2d_mlp_mfma";
    } else if (pad_kch != 0) {
        // K not a multiple of 64 (e.g. the reference's embed_size 200: K = 1000): the every-wave-gathers kernel on a W1 copy
        // zero-padded to 64 pad_kch rows; a lane reads zeros for the k-values past the end of its rows
        const int Kp = pad_kch * 64;
        const size_t lds = (size_t)(2 * RING_FLOATS + MH1 + 2 * MH2) * sizeof(float);
        const int64_t ntiles = (B + 32 * MWAVES - 1) / (32 * MWAVES);
        const unsigned grid = (unsigned)(ntiles < h->num_cu ? ntiles : h->num_cu);
        const bool x3 = h->opt_mlp_bf16x3 != 0;
        if (!h->mlp_w1pad) {               // built once per head (m2d_set_mlp_head resets it)
            M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_w1pad, (size_t)Kp * MH1 * sizeof(float)));
            M2D_HIP_TRY(h, hipMemsetAsync(h->mlp_w1pad, 0, (size_t)Kp * MH1 * sizeof(float), stream));
            M2D_HIP_TRY(h, hipMemcpyAsync(h->mlp_w1pad, h->mlp_w1, (size_t)a.K * MH1 * sizeof(float), hipMemcpyDeviceToDevice, stream));
        }
        a.w1 = h->mlp_w1pad;
        if (x3 && !h->mlp_w1x3) {
            M2D_HIP_TRY(h, hipMalloc((void **)&h->mlp_w1x3, (size_t)Kp * MH1 * 4 + (size_t)MH1 * MH2 * 4));
            hipLaunchKernelGGL(m2d_mlp_split_w1, dim3((unsigned)(((int64_t)Kp * MH1 + 255) / 256)), dim3(256), 0, stream,
                               h->mlp_w1pad, Kp, reinterpret_cast<__bf16 *>(h->mlp_w1x3));
            hipLaunchKernelGGL(m2d_mlp_split_w2, dim3(MH1 * MH2 / 256), dim3(256), 0, stream, h->mlp_w2,
                               reinterpret_cast<__bf16 *>(h->mlp_w1x3) + (size_t)Kp * MH1 * 2);
            M2D_HIP_TRY(h, hipGetLastError());
        }
        a.w1x3 = reinterpret_cast<const __bf16 *>(h->mlp_w1x3);
        a.w2x3 = a.w1x3 ? a.w1x3 + (size_t)Kp * MH1 * 2 : nullptr;
#define M2D_MLP_PAD_CASE(N)                                                                                 \
    if (pad_kch == N) {                                                                                     \
        if (x3) {                                                                                           \
            M2D_HIP_TRY(h, hipFuncSetAttribute((const void *)m2d_mlp_mfma<N, true, true>,                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
            hipLaunchKernelGGL((m2d_mlp_mfma<N, true, true>), dim3(grid), dim3(MWAVES * 64), lds, stream, a); \
        } else {                                                                                            \
            M2D_HIP_TRY(h, hipFuncSetAttribute((const void *)m2d_mlp_mfma<N, false, true>,                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
            hipLaunchKernelGGL((m2d_mlp_mfma<N, false, true>), dim3(grid), dim3(MWAVES * 64), lds, stream, a); \
        }                                                                                                   \
    }
        M2D_MLP_PAD_CASE(5) M2D_MLP_PAD_CASE(10) M2D_MLP_PAD_CASE(16) M2D_MLP_PAD_CASE(20)
#undef M2D_MLP_PAD_CASE
        h->last_kernel = x3 ? "m2d_mlp_mfma_bf16x3" : "m2d_mlp_mfma";
    } else {
        const size_t lds = (size_t)4 * (a.K + a.H1 + a.H2) * sizeof(float);
        if (lds > 160 * 1024) {
            h->last_error = "m2d_score_pairs_mlp: K + H1 + H2 too large for the generic kernel";
            return M2D_ERR_UNSUPPORTED;
        }
        int64_t blocks = (B + 3) / 4;
        if (blocks > (int64_t)h->num_cu * 4) blocks = (int64_t)h->num_cu * 4;
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)m2d_mlp_generic, (int)lds));
        hipLaunchKernelGGL(m2d_mlp_generic, dim3((unsigned)blocks), dim3(256), lds, stream, a);
        h->last_kernel = "m2d_mlp_generic";
    }
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}
