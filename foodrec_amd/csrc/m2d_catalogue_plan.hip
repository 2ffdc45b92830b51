// Pattern-grouped retrieval, host side and planning kernels: the pattern-sorted dish table (built once per mask table), the plan of
// a call (per-user bounds and relevant patterns, users sorted by pattern mask, (user block, dish range) items longest first), the
// launcher that strings a call's kernels together, and m2d_launch_topk_users' choice between this path and the dense kernels.
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

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

// ingredient form: the largest |RE[d]| of the catalogue (the groups' norms above are H[d]'s), for the hi x hi first form's bound; float
// bits order like the values for non-negative floats, a NaN norm counts as +inf
__global__ __launch_bounds__(256) void m2d_grp_max_norm(const float *re, int64_t I, int E, int32_t *out_bits)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t d0 = (int64_t)blockIdx.x * 256 + wave * 64;
    float mx = 0.f;
    for (int r = 0; r < 64 && d0 + r < I; ++r) {
        float q = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float x = re[(d0 + r) * E + e];
            q = fmaf(x, x, q);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) q += __shfl_xor(q, off, 64);
        const float nr = sqrtf(q);
        mx = fmaxf(mx, nr == nr ? nr : INFINITY);
    }
    if (lane == 0) atomicMax(out_bits, __float_as_int(mx));
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
        if (gid == 0) { *zero_tie = 0; zero_tiles[0] = 0ull; zero_tiles[1] = 0ull; }
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

// ---- high_level_score_coefficient = 1 ------------------------------------------------------------------------------------------
// `1 - coef` is then an exact float32 zero (Model_Recommender.py:17, :96): score(u, d) = alpha_P[u] for EVERY dish of mask pattern
// P, and heapq.nlargest (evaluate.py:63: ties to the lower id) returns the best pattern's lowest ids, then the next pattern's.
// Through the scan every user would be one tie per group -- three or more dishes equal to the list's end, the tie repair's case:
// a pass over the user's patterns for each of them (65 536 users x 100 k dishes: seconds where the scan takes half a millisecond).
// So the list is read off instead: the patterns' first GRP_FIRST ids in id order (m2d_grp_first_ids, with the table build) and
// fifteen alpha_P per user -- no dish row is touched.
constexpr int GRP_FIRST = 16;                              // ids kept per pattern: the pattern-grouped path serves k <= 16

// slot q of pattern P: its q-th lowest dish id, or -1.  One block per pattern walks the mask table in id order and stops at
// GRP_FIRST matches (patterns are spread over the catalogue: a few hundred dishes in).
__global__ __launch_bounds__(256) void m2d_grp_first_ids(const float *cats, int64_t I, int C, int32_t *first)
{
    __shared__ int s_wave[4];
    const int pat = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < GRP_FIRST) first[pat * GRP_FIRST + threadIdx.x] = -1;
    if (pat == 0) return;                                   // empty masks score NaN: never ranked
    int count = 0;
    for (int64_t base = 0; base < I && count < GRP_FIRST; base += 256) {     // block-uniform
        const int64_t d = base + threadIdx.x;
        int pt = -1;
        if (d < I) {
            pt = 0;
            for (int c = 0; c < C; ++c) pt |= (cats[d * C + c] != 0.f) ? (1 << c) : 0;
        }
        const unsigned long long m = __ballot(pt == pat);
        __syncthreads();                                    // (the previous trip's s_wave has been read)
        if (lane == 0) s_wave[wave] = __builtin_popcountll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? s_wave[w] : 0;
            total += s_wave[w];
        }
        const int pos = count + before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (pt == pat && pos < GRP_FIRST) first[pat * GRP_FIRST + pos] = (int32_t)d;
        count += total;
    }
}

int32_t *grouped_first_ids(m2d_engine *h)                  // behind the row norms, in the table build's work area
{
    const size_t nblk = (size_t)((h->I + 255) / 256);
    return h->grp_work + nblk * GRP_KEYS + GRP_WORDS + 4 + (size_t)h->I;
}

// 16 lanes per user: <U_high, CE_c> exactly as m2d_topk_user_plan sums it (a float4 column per lane, row rotations), alpha_P as
// the scan kernels and the repair form it (repair_alpha), then the user's first lane merges the patterns' id lists by
// (alpha desc, id asc) -- k rounds over at most fifteen heads.  Fewer rankable dishes than k: the NaN dishes follow in id order.
__global__ __launch_bounds__(256) void m2d_topk_high_level_only(const float *pm, const float *ce, const int32_t *users, int64_t nU, int64_t U,
                                                                int64_t user_base, int E, const int32_t *first, int k, float a, int64_t I,
                                                                float *out_scores, int32_t *out_ids, int32_t *err)
{
    const int lane = threadIdx.x & 63, j = lane & 15;
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int E4 = E >> 2;
    int64_t ul = 0;
    if (u < nU) {
        const int32_t uid = users[u];
        ul = (int64_t)uid - user_base;
        if (ul < 0 || ul >= U) {
            if (j == 0 && atomicCAS(&err[0], 0, M2D_ERR_BAD_USER_ID) == 0) {
                err[1] = uid;
                err[2] = (int32_t)(u & 0xffffffff);
                err[3] = (int32_t)(u >> 32);
            }
            ul = 0;
        }
    }
    const v4f *pmu = reinterpret_cast<const v4f *>(pm) + (size_t)ul * (5 * E4);
    const v4f *ce4 = reinterpret_cast<const v4f *>(ce);
    float hc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = j; q < E4; q += 16) {
        const v4f uh = pmu[q];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f w = ce4[c * E4 + q];
            hc[c] += fmaf(uh.x, w.x, uh.y * w.y) + fmaf(uh.z, w.z, uh.w * w.w);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) hc[c] = row16_sum(hc[c]);
    if (u >= nU || j != 0) return;
    float alpha[GRP_MAXPAT];
    int head[GRP_MAXPAT];                                   // next slot of each pattern's id list
#pragma unroll
    for (int pt = 1; pt < GRP_MAXPAT; ++pt) {
        alpha[pt] = fmaxf(repair_alpha(a, hc, pt), -INFINITY);
        head[pt] = 0;
    }
    float *os = out_scores + (size_t)u * k;
    int32_t *oi = out_ids + (size_t)u * k;
    for (int o = 0; o < k; ++o) {
        int best = 0;
        int32_t bid = -1;
        float bs = 0.f;
#pragma unroll
        for (int pt = 1; pt < GRP_MAXPAT; ++pt) {
            const int32_t id = head[pt] < GRP_FIRST ? first[pt * GRP_FIRST + head[pt]] : -1;
            const bool take = id >= 0 && (bid < 0 || alpha[pt] > bs || (alpha[pt] == bs && id < bid));
            best = take ? pt : best;
            bid = take ? id : bid;
            bs = take ? alpha[pt] : bs;
        }
#pragma unroll
        for (int pt = 1; pt < GRP_MAXPAT; ++pt) head[pt] += pt == best ? 1 : 0;
        os[o] = bid >= 0 ? bs : __builtin_nanf("");
        oi[o] = bid;
    }
    fill_absent_user(os, oi, k, I);                         // (its own stores, in program order)
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
                                                        (size_t)I * sizeof(float) + (size_t)GRP_MAXPAT * GRP_FIRST * sizeof(int32_t)));
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
    M2D_HIP_TRY(h, hipMemsetAsync(stat + 2, 0, sizeof(float), st));            // stat[2]: the largest |RE[d]| (ingredient form only)
    if (h->dish_high) hipLaunchKernelGGL(m2d_grp_max_norm, dim3(nblk), dim3(256), 0, st, h->re, I, h->E, reinterpret_cast<int32_t *>(stat + 2));
    M2D_HIP_TRY(h, hipMemsetAsync(grp + GRP_RMAX, 0, 16 * sizeof(int32_t), st));
    hipLaunchKernelGGL(m2d_grp_hist, dim3(nblk), dim3(256), 0, st, h->dish_cats, norm, stat, I, h->C, blk_hist, flags, grp + GRP_RMAX);
    hipLaunchKernelGGL(m2d_grp_scan, dim3(1), dim3(GRP_KEYS * GRP_SCAN_SPLIT), 0, st, blk_hist, nblk, grp, h->grp_tile_info);
    hipLaunchKernelGGL(m2d_grp_scatter, dim3(nblk), dim3(256), 0, st, h->dish_cats, norm, stat, I, h->C, blk_hist, grp, h->grp_perm);
    hipLaunchKernelGGL(m2d_grp_first_ids, dim3(GRP_MAXPAT), dim3(256), 0, st, h->dish_cats, I, h->C, grouped_first_ids(h));
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
    // (the whole device, not `st` alone: the write whose row check sets the word may have been queued on another stream -- an
    //  engine moved between streams is the caller's to order, but a stale "finite" here would serve inf tables from the grouped
    //  kernels, and this read happens once after a m2d_write_memory, not per call)
    int32_t word = 0;
    M2D_HIP_TRY(h, hipDeviceSynchronize());
    M2D_HIP_TRY(h, hipMemcpyAsync(&word, h->nonfinite_dev, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    h->grp_nonfinite = word != 0;
    h->grp_nonfinite_known = true;
    return M2D_OK;
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

// One pattern-grouped call: plan -> scan -> merge of the dish ranges -> refinement / tie repair.
//   E8: the scan kernel's row width / 8 (the tables' E, or the next instantiated width when PAD; 2 E / 8 with HV)
//   KR: list slots per lane (10 or 16 >= k)      BF16X3: split-bf16 MFMA, else exact f32
//   HV: rows [H[d] | RE[d]] of the ingredient table (pipelined split-bf16 kernel only)      PAD: zero-padded rows (exact f32 only)
int launch_grouped(m2d_engine *h, const int E8, const int KR, const bool BF16X3, const bool HV, const bool PAD, const int32_t *users,
                   int64_t nU, int32_t k, float *final_s, int32_t *final_i, hipStream_t st)
{
    constexpr int WAVES = 8;                                 // waves per block unless the launcher takes blocks of 128 users (`half`)
    if ((HV && !BF16X3) || (PAD && BF16X3)) {
        h->last_error = "launch_grouped: the ingredient form is split bf16 only, zero-padded rows exact f32 only";
        return M2D_ERR_UNSUPPORTED;
    }
    const int E = E8 * 8;
    if (h->b == 0.f && !HV) {
        // high_level_score_coefficient = 1: every dish of a pattern scores alpha_P -- the lists are read off the patterns' first ids
        // (m2d_topk_high_level_only above), whatever kernel the options name
        hipLaunchKernelGGL(m2d_topk_high_level_only, dim3((unsigned)((nU * 16 + 255) / 256)), dim3(256), 0, st, h->pm, h->ce, users, nU, h->U,
                           h->user_base, h->E, grouped_first_ids(h), (int)k, h->a, h->I, final_s, final_i, h->err_dev);
        M2D_HIP_TRY(h, hipGetLastError());
        h->topk_tie_list = nullptr; h->topk_refine_counter = nullptr; h->topk_tiles_counter = nullptr;      // (diagnostics: nothing was scanned)
        h->topk_tiles_full = 0; h->topk_flags_used = nU;
        h->last_kernel = "m2d_topk_high_level_only";
        return M2D_OK;
    }
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
    // The hi x hi first form of that kernel (APX in m2d_catalogue_scan_bf16.hip) for the catalogues where a tile with a candidate is
    // the exception.  It pays once fewer than about a quarter (E = 64) / a half (E = 128) of the (wave, tile) pairs still need their
    // cross products -- 65 536 users, pruned, three-product form against it: E = 64 200 k dishes 0.73 / 0.88 ms, 500 k 1.51 / 1.63,
    // 1 M 2.81 / 2.63; E = 128 300 k 1.94 / 1.94, 500 k 2.96 / 2.75, 1 M 6.16 / 4.95 -- so: more than 24 576 tiles at E = 64, more
    // than 10 240 at E = 128.  The rule looks at the catalogue alone: that form's scores are not the three-product kernels' bits, and
    // every launch shape of one problem must return the same lists (blocks of 256 users there: "topk_block" = 128 is not
    // honoured).  "topk_form" 3 / 4 (diagnostic): that form for any catalogue / never.
    // (The ingredient form multiplies every tile -- no pattern can be pruned -- so tiles with a candidate are the exception from
    //  small catalogues on: 65 536 users x 20 k dishes 0.89 -> 0.64 ms, 100 k 3.79 -> 2.09 ms, 1 M 36.3 -> 18.5; more than 256 tiles.)
    const bool apx = BF16X3 && pipe && (E == 64 || E == 128) && h->opt_topk_form != 4 &&
                     (h->grp_tiles > (HV ? 256 : (E == 64 ? 24576 : 10240)) || h->opt_topk_form == 3);
    const bool half = half_ok && !apx && (h->opt_topk_block == 128 || (h->opt_topk_block == 0 && M2D_TOPK_HALF_BLOCKS && h->opt_topk_prune != 0 &&
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
    a.prog_limit = h->opt_variant == 15 ? 0 : (1 << 24);   // "variant" 15: test hook of the progress-word timeout
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
        h->topk_tie_list = nullptr; h->topk_tie_final = nullptr;     // (inside the buffer freed below)
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
    // (a user's word is its position in the call | left-out dishes to take along << 30: calls of 2^30 users or more go without)
    const bool ext = planned && h->opt_topk_refine != 0 && !HV && !PAD && nU < ((int64_t)1 << 30) &&
                     (BF16X3 ? (pipe && !(E == 128 && KR == 16)) : E8 <= 16);
    float *ex_final = nullptr;
    if (ext) {
        const size_t ex_need = ((size_t)nU * (nsplit > 1 ? nsplit + (nsplit > 64 ? nsplit / 64 : 0) + 1 : 1)) * 8 + (size_t)nU + 8;
        if (h->topk_ex_cap < ex_need) {
            h->topk_refine_counter = nullptr;                // (it points into the buffer freed below: nothing may read it if the allocation fails)
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
            h->topk_tiles_counter = nullptr;                 // (inside the buffer freed below)
            if (h->topk_plan) M2D_HIP_TRY(h, hipFree(h->topk_plan));
            h->topk_plan = nullptr; h->topk_plan_cap = 0;
            M2D_HIP_TRY(h, hipMalloc((void **)&h->topk_plan, need * sizeof(float)));
            h->topk_plan_cap = need;
        }
        float *plan = h->topk_plan;
        int32_t *order = reinterpret_cast<int32_t *>(plan + (size_t)nU * 8), *hist = order + ((nU + 3) & ~(int64_t)3);      // hist: 16-B aligned
        unsigned long long *counter = reinterpret_cast<unsigned long long *>(hist + PLAN_KEYS);
        const bool prune = h->opt_topk_prune != 0;
        const bool sorted = prune && !HV && nU > 32 * WV;
        {
            const int pmode = (HV || !prune) ? 1 : (h->opt_topk_prune == 2 ? 2 : (h->opt_topk_prune == 4 ? 4 : 0));
            const float *probes = (HV || !prune) ? nullptr : h->grp_rs;
            const dim3 pgrid((unsigned)((nU * 16 + 255) / 256));
            // probe rows per user: each costs a row read per user (16: +18 us for 65 536 users) and buys a tighter bound -- 16 rows
            // at 100 k dishes (0.71 ms; 32: 0.73), 32 at 1 M (3.83 ms; 16: 4.02)
            const int nprobe = a.tiles < 8192 ? 16 : (a.tiles < 65536 ? 32 : PLAN_PROBES);
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
            int32_t *work = reinterpret_cast<int32_t *>(counter + 2), *items = work + nitems;      // counter: two words (tiles, completed)
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
    {
        const ScanShape shape{E, KR, BF16X3, HV, PAD, pipe, WV, a.ex_out != nullptr, apx};
        h->topk_apx_last = apx;
        const int rc = BF16X3 ? m2d_topk_scan_bf16_launch(h, a, shape, grid, lds, st) : m2d_topk_scan_f32_launch(h, a, shape, grid, lds, st);
        if (rc != M2D_OK) return rc;
    }
    M2D_HIP_TRY(h, hipGetLastError());
    RefineArgs f;                                            // near-tied lists: finished in the repair's arithmetic (m2d_topk_refine)
    f.pm = h->pm; f.re = h->re; f.ce = h->ce; f.cats = h->dish_cats; f.plan = a.plan; f.tie_final = tie_final; f.ex = ex_final;
    f.users = users; f.tie_list = tie_list; f.counter = h->topk_refine_counter; f.nU = nU; f.U = h->U; f.I = h->I;
    f.user_base = h->user_base; f.E = h->E; f.k = k; f.a = h->a; f.b = h->b; f.out_scores = final_s; f.out_ids = final_i;
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
        if (nsplit == 1)                                     // (with dish ranges the last merge pass has listed the tied users)
            m2d_topk_launch_tie_compact(tie_final, nU, tie_list, final_s, final_i, (int)k, h->I, ext ? 1 : 0, st);
        if (ext)                                             // (may add to the repair's list; with dish ranges the last merge pass has
            m2d_topk_launch_refine(f, nsplit == 1, st);      //  flagged the near-tied users)
        const int rc = m2d_topk_launch_repair(h, r, HV, st);
        if (rc != M2D_OK) return rc;
    }
    h->last_kernel = BF16X3 ? "m2d_topk_grouped_bf16x3" : "m2d_topk_grouped";      // both bf16 forms report this name
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
        const int KR = k <= 10 ? 10 : 16;                    // list slots per lane
        if (hv_ok && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite)       // rows [H[d] | RE[d]]: width 2 E
            return launch_grouped(h, 2 * h->E / 8, KR, true, true, false, users, nU, k, out_scores, out_ids, stream);
        if (!h->dish_high && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite) {
            // "topk_bf16x3" option: 1 = split-bf16 MFMA (E = 64 / 128), 0 = exact-f32 MFMA; embedding sizes without a kernel of
            // their own (`padded`) run exact f32 on rows zero-padded to `roww` floats
            const bool x3 = h->opt_topk_bf16x3 != 0 && (h->E == 64 || h->E == 128);
            return launch_grouped(h, roww / 8, KR, x3, false, padded, users, nU, k, out_scores, out_ids, stream);
        }
    }
    return m2d_topk_dense_launch(h, users, nU, k, out_scores, out_ids, stream);
}
