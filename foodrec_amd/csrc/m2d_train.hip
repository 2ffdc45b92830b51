// Training step: Model.loss + Model.train, Model_Recommender.py:99-104, :223-241,
// as the call site runs it -- sess.run([model.loss_value, model.learning_rate, ..., model.train_op], feed_dict),
// Train_recommender.py:180-199.
//
//   loss   = mean_b( max(s_b, 0) - s_b y_b + log(1 + exp(-|s_b|)) ),  s = Model.inference (:56-97)        :101-103
//   grads  = d loss / d {Personal_Memory, Recipe_Embedding, Category_Embedding}   (General_Memory: none)    :236
//            per-pair rows for the two gathered tables (TF: IndexedSlices), dense for Category_Embedding:
//              q_h = g_b a / n_b,  q_l = g_b (1 - a) / n_b,  g_b = (sigmoid(s_b) - y_b) / B
//              dPM[u_b][0]   += q_h sum_c m_bc CE_c          dPM[u_b][1 + c] += q_l m_bc RE[d_b]
//              dRE[d_b]      += q_l sum_c m_bc PM[u_b][1+c]  dCE[c]          += q_h m_bc PM[u_b][0]
//   clip   = tf.clip_by_global_norm(grads, 5.0): the norm runs over the PER-PAIR rows (duplicate ids not yet
//            summed) and the dense dCE; scale = clip * min(1 / norm, 1 / clip)                              :237
//   update = optimizer.apply_gradients: duplicate rows are summed, then adam / adagrad / rmsprop / sgd     :228-240
//
// The optimizer rules restate TF 1.x's published behaviour (oracle/train_oracle.py lists them; PARITY UNPINNED,
// TensorFlow is not in the image).  The one that shapes the kernels: TF 1.x Adam's sparse path is not lazy -- it
// decays m and v of EVERY row and moves EVERY row by lr_t m / (sqrt(v) + eps) each step -- so Adam is a dense
// streaming pass over var / m / v (HBM-bound, 6 x table bytes per step) that picks a row's summed gradient out
// of a compact buffer when the row was touched; adagrad / rmsprop / sgd touch only the batch's rows.
//
// Data flow of one step (all on the caller's stream, no host synchronisation):
//   claim     one thread per pair: the first pair to see a user (dish) claims the next compact slot for it in a
//             row -> slot map (atomicCAS), so duplicate ids share one gradient row;
//   grad      one wave per pair: forward (two wave reductions), loss term, then the pair's gradient rows are
//             float-atomic-added into the compact buffers; dCE (reduced per wave in LDS), the loss sum and the sum
//             of squares of the per-pair values leave each block as plain per-block partials;
//   reduce    the per-block dCE partials -> dCE;
//   finalize  one block: per-block loss / square sums in a fixed order, norm over dCE joins in, writes
//             {loss, global norm, scale, lr};
//   apply     rows (Adam: all rows; others: claimed rows) and the dense dCE;
//   cleanup   claimed rows: map entry back to -1, compact gradient row back to zero.
// Batches of up to 1024 pairs (the reference's own: 128 per step, 8 in the memory-write branch) run the same steps as
// TWO launches, with no memset and no device-to-device copy in between (nine launches took 57 us per SGD step):
//   m2d_train_grad_fused   one wave per pair: claims its two slots itself (a pair that finds a slot being claimed by
//                          another wave waits for the number), forward, loss, gradient rows; the LAST block to finish
//                          (a ticket counter) adds up the per-block partials, writes {loss, norm, scale, lr}, Adam's
//                          lr_t, advances the step count / beta powers, and resets the other parity's slot counters;
//   m2d_train_apply_fused  every table in one grid: Adam walks all rows, the others the claimed rows; a row's slot and
//                          gradient are released by the wave that applied it.
// An out-of-range id is latched by the claim pass (m2d_check reports it) and the apply pass then leaves every
// table and slot as it was -- TF raises InvalidArgumentError from the gather before anything is assigned.
// Sums are float atomics: results are order-dependent in the last bits, like m2d_write_memory.
#include "m2d_engine.h"

namespace {

struct TrainArgs {
    float *pm, *re, *ce;
    const int32_t *users, *items;
    const float *cats;      // [B, C]
    const float *labels;    // [B]
    int64_t B, U, I, user_base;
    int32_t C, E;
    float a, b;
    int32_t *map_u, *map_d;     // row -> compact slot, -1 = untouched
    int32_t *slot_u, *slot_d;   // compact slot -> row
    int32_t *cnt;               // [0] users claimed, [1] dishes claimed
    float *gu, *gd;             // [cap, (C+1) E], [cap, E]
    float *gce;                 // [C, E]
    float *part_ce;             // [grid, C, E] per-block partial dCE (summed by the finalize kernel), or null
    double *part_acc;           // [grid, 2]    per-block partial {loss sum, square sum}
    int32_t nblocks;            // grid of the grad kernel
    float *scal;                // [0] loss, [1] global norm, [2] scale, [3] learning rate
    int32_t *err;
    int32_t accumulate;         // 0: loss + norm only (no slots are claimed, nothing is added to gu / gd)
    float clip, lr;
    // fused form (m2d_train_grad_fused): slot counters alternate between steps (cnt + 2 * parity), the last block finishes the step
    int32_t *done;              // ticket counter
    float *out;                 // caller's f32[4] {loss, norm, scale, lr}, or null
    struct OptState *opt;
    int32_t learner, parity;
};

__device__ __forceinline__ void train_latch(int32_t *err, int code, int64_t value, int64_t index)
{
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int32_t)value;
        err[2] = (int32_t)(index & 0xffffffff);
        err[3] = (int32_t)(index >> 32);
    }
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

__global__ __launch_bounds__(256) void m2d_train_claim(TrainArgs p)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= p.B) return;
    const int32_t uid = p.users[b], did = p.items[b];
    const int64_t ul = (int64_t)uid - p.user_base;
    if (ul < 0 || ul >= p.U) { train_latch(p.err, M2D_ERR_BAD_USER_ID, uid, b); return; }
    if (did < 0 || (int64_t)did >= p.I) { train_latch(p.err, M2D_ERR_BAD_ITEM_ID, did, b); return; }
    if (atomicCAS(&p.map_u[ul], -1, -2) == -1) {
        const int s = atomicAdd(&p.cnt[0], 1);
        p.slot_u[s] = (int32_t)ul;
        p.map_u[ul] = s;            // read by later launches only
    }
    if (atomicCAS(&p.map_d[did], -1, -2) == -1) {
        const int s = atomicAdd(&p.cnt[1], 1);
        p.slot_d[s] = did;
        p.map_d[did] = s;
    }
}

// One wave per pair.  LDS: this wave's partial dCE [C, E] when it fits (ce_lds != 0), else atomics to global.
// Nothing every wave would add to the same few addresses goes through atomics: dCE (C*E floats), the loss sum and
// the square sum leave a block as plain stores into per-block partial rows that the finalize kernel adds up in a
// fixed order (8192 waves each adding 256 floats into one 1-KiB dCE was 250 us of a 330-us kernel).
__global__ __launch_bounds__(256) void m2d_train_grad(TrainArgs p, int ce_lds)
{
    extern __shared__ float dce_all[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int C = p.C, E = p.E;
    float *dce = dce_all + (size_t)wv * C * E;
    if (ce_lds)
        for (int i = lane; i < C * E; i += 64) dce[i] = 0.f;
    __syncthreads();                                        // the zeroing lane is not the lane that owns column e below
    double loss_acc = 0.0, sq_acc = 0.0;
    const float invB = 1.0f / (float)p.B;
    for (int64_t b = wave0; b < p.B; b += nwaves) {
        const int32_t uid = p.users[b], did = p.items[b];
        const int64_t ul = (int64_t)uid - p.user_base;
        if (ul < 0 || ul >= p.U || did < 0 || (int64_t)did >= p.I) {      // latched by the claim pass / below
            if (!p.accumulate && lane == 0)
                train_latch(p.err, (ul < 0 || ul >= p.U) ? M2D_ERR_BAD_USER_ID : M2D_ERR_BAD_ITEM_ID,
                            (ul < 0 || ul >= p.U) ? uid : did, b);
            continue;
        }
        const float *m = p.cats + (size_t)b * C;
        const float *urow = p.pm + (size_t)ul * (C + 1) * E;
        const float *drow = p.re + (size_t)did * E;
        float n = 0.f;
        for (int c = 0; c < C; ++c) n += m[c];                                      // :77
        float hi = 0.f, lo = 0.f;
        for (int e = lane; e < E; e += 64) {
            float H = 0.f, L = 0.f;
            for (int c = 0; c < C; ++c) {
                H = fmaf(m[c], p.ce[(size_t)c * E + e], H);                         // :67-75
                L = fmaf(m[c], urow[(size_t)(1 + c) * E + e], L);                   // :82-90
            }
            hi = fmaf(urow[e], H, hi);
            lo = fmaf(drow[e], L, lo);
        }
        hi = wave_sum(hi);
        lo = wave_sum(lo);
        const float s = m2d_blend_unfused(p.a, hi / n, p.b, lo / n);  // :79, :93, :95-96
        const float y = p.labels[b];
        const float loss_b = fmaxf(s, 0.f) - s * y + log1pf(expf(-fabsf(s)));       // :101
        const float gs = (1.0f / (1.0f + expf(-s)) - y) * invB;                     // d mean / d s_b
        const float qh = gs * p.a / n, ql = gs * p.b / n;
        const int su = p.accumulate ? p.map_u[ul] : 0, sd = p.accumulate ? p.map_d[did] : 0;
        float *gu = p.gu + (size_t)su * (C + 1) * E;
        float *gd = p.gd + (size_t)sd * E;
        float sq = 0.f;
        for (int e = lane; e < E; e += 64) {
            float H = 0.f, L = 0.f;
            for (int c = 0; c < C; ++c) {
                H = fmaf(m[c], p.ce[(size_t)c * E + e], H);
                L = fmaf(m[c], urow[(size_t)(1 + c) * E + e], L);
            }
            const float uh = urow[e], it = drow[e];
            const float v0 = qh * H, vd = ql * L;
            sq = fmaf(v0, v0, sq);
            sq = fmaf(vd, vd, sq);
            if (p.accumulate) {
                atomicAdd(gu + e, v0);
                atomicAdd(gd + e, vd);
            }
            for (int c = 0; c < C; ++c) {
                const float vc = (ql * m[c]) * it;
                sq = fmaf(vc, vc, sq);
                // a zero mask entry contributes an exact zero (unless RE holds inf / NaN -- not reproduced)
                if (p.accumulate && m[c] != 0.f) atomicAdd(gu + (size_t)(1 + c) * E + e, vc);
                const float w = (qh * m[c]) * uh;
                if (ce_lds) dce[c * E + e] += w;                                    // this lane owns column e
                else if (m[c] != 0.f) atomicAdd(p.gce + (size_t)c * E + e, w);
            }
        }
        sq_acc += (double)wave_sum(sq);
        loss_acc += (double)loss_b;
    }
    __shared__ double wave_acc[4][2];
    if (lane == 0) { wave_acc[wv][0] = loss_acc; wave_acc[wv][1] = sq_acc; }
    __syncthreads();
    if (ce_lds) {
        float *row = p.part_ce + (size_t)blockIdx.x * C * E;
        for (int i = threadIdx.x; i < C * E; i += 256)
            row[i] = (dce_all[i] + dce_all[C * E + i]) + (dce_all[2 * C * E + i] + dce_all[3 * C * E + i]);
    }
    if (threadIdx.x < 2)
        p.part_acc[(size_t)blockIdx.x * 2 + threadIdx.x] =
            (wave_acc[0][threadIdx.x] + wave_acc[1][threadIdx.x]) + (wave_acc[2][threadIdx.x] + wave_acc[3][threadIdx.x]);
}

// dCE[i] = sum over the grad kernel's blocks of their partial rows.  A block owns 64 columns and every gridDim.y-th
// slice of the rows; its 16 waves split those, lane l reads column l of each (256-B coalesced); the gridDim.y
// block sums meet in dCE through one float atomic each (dCE is zeroed at the start of the step).
__global__ __launch_bounds__(1024) void m2d_train_reduce_ce(TrainArgs p)
{
    __shared__ float wsum[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = p.C * p.E, i = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (i < n)
        for (int b = blockIdx.y * 16 + wv; b < p.nblocks; b += 16 * gridDim.y) a += p.part_ce[(size_t)b * n + i];
    wsum[wv][lane] = a;
    __syncthreads();
    if (wv == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += wsum[w][lane];
        atomicAdd(p.gce + i, t);
    }
}

__global__ __launch_bounds__(256) void m2d_train_finalize(TrainArgs p)
{
    __shared__ double part[4];
    __shared__ double tot[2][4];
    {   // per-block partial loss / square sums, fixed order
        double l = 0.0, q = 0.0;
        for (int b = threadIdx.x; b < p.nblocks; b += 256) {
            l += p.part_acc[(size_t)b * 2];
            q += p.part_acc[(size_t)b * 2 + 1];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            l += __shfl_xor(l, off, 64);
            q += __shfl_xor(q, off, 64);
        }
        if ((threadIdx.x & 63) == 0) { tot[0][threadIdx.x >> 6] = l; tot[1][threadIdx.x >> 6] = q; }
    }
    __syncthreads();
    double s = 0.0;
    for (int i = threadIdx.x; i < p.C * p.E; i += 256) s += (double)p.gce[i] * (double)p.gce[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n2 = ((tot[1][0] + tot[1][1]) + (tot[1][2] + tot[1][3])) + part[0] + part[1] + part[2] + part[3];
        const float norm = (float)sqrt(n2);
        p.scal[0] = (float)(((tot[0][0] + tot[0][1]) + (tot[0][2] + tot[0][3])) / (double)p.B);   // reduce_mean, :103
        p.scal[1] = norm;
        p.scal[2] = p.clip * fminf(1.0f / norm, 1.0f / p.clip);                     // clip_by_global_norm
        p.scal[3] = p.lr;
    }
}

// Optimizer state that must move only when a step really applied: kept on the device and advanced by the step's last
// kernel under the same `err[0] == 0` condition as the apply kernels, so a step refused for an out-of-range id leaves
// the step count and Adam's beta powers where they were (TF raises before any assign, the beta-power assigns included).
struct OptState {
    float b1p, b2p;         // Adam's beta1_power / beta2_power variables: start at beta, times beta per applied step
    int64_t steps;          // optimizer steps applied since m2d_train_begin
};

struct RuleArgs {
    int32_t rule;           // M2D_LEARNER_*
    float lr;               // args.lr; adam scales it by sqrt(1 - b2^t) / (1 - b1^t) from the device-side powers
    float b1, b2, eps;      // adam: betas, epsilon; rmsprop: b1 = decay, b2 = momentum, eps
};

__device__ __forceinline__ void apply_one(const RuleArgs &r, float g, float &var, float &s0, float &s1)
{
    if (r.rule == M2D_LEARNER_ADAM) {
        s0 = s0 * r.b1 + g * (1.0f - r.b1);
        s1 = s1 * r.b2 + (g * g) * (1.0f - r.b2);
        var -= r.lr * s0 / (sqrtf(s1) + r.eps);
    } else if (r.rule == M2D_LEARNER_ADAGRAD) {
        s0 += g * g;
        var -= r.lr * g / sqrtf(s0);
    } else if (r.rule == M2D_LEARNER_RMSPROP) {
        s0 += (g * g - s0) * (1.0f - r.b1);
        s1 = s1 * r.b2 + r.lr * g / sqrtf(s0 + r.eps);
        var -= s1;
    } else {
        var -= r.lr * g;
    }
}

// Rows of one table.  ALL = true (Adam): every row r < R, gradient row map[r] when >= 0, else zero.
// ALL = false: claimed rows only, r = slot_row[s] for s < *count.  map == nullptr: dense gradient (row r of G).
// W floats per row; a wave walks a row 64 (x4 when W % 4 == 0) floats at a time.
typedef float v4f __attribute__((ext_vector_type(4)));
template <int VEC> struct RowVec;
template <> struct RowVec<1> { typedef float T; };
template <> struct RowVec<4> { typedef v4f T; };
__device__ __forceinline__ float lane_get(const v4f &v, int j) { return v[j]; }
__device__ __forceinline__ float lane_get(const float &v, int) { return v; }
__device__ __forceinline__ void lane_set(v4f &v, int j, float x) { v[j] = x; }
__device__ __forceinline__ void lane_set(float &v, int, float x) { v = x; }

template <bool ALL, int VEC>
__global__ __launch_bounds__(256) void m2d_train_apply(float *var, float *s0, float *s1, const float *G, const int32_t *map,
                                                       const int32_t *slot_row, const int32_t *count, int64_t R, int32_t W,
                                                       const float *scal, RuleArgs r, const int32_t *err, const OptState *st,
                                                       int32_t *nonfinite)
{
    typedef typename RowVec<VEC>::T vf;
    if (err[0] != 0) return;    // an id was out of range: like TF's InvalidArgumentError, the step applies nothing
    if (r.rule == M2D_LEARNER_ADAM) {                                     // AdamOptimizer._apply_dense; kept wave-uniform (SGPRs)
        const float b1p = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st->b1p)));
        const float b2p = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, st->b2p)));
        r.lr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r.lr * sqrtf(1.0f - b2p) / (1.0f - b1p))));
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const float scale = scal[2];
    const int64_t n = ALL ? R : (int64_t)*count;
    const bool two = r.rule == M2D_LEARNER_ADAM || r.rule == M2D_LEARNER_RMSPROP, one = two || r.rule == M2D_LEARNER_ADAGRAD;
    float fin = 0.f;                // x * 0 summed over the values written: NaN iff one of them is inf / NaN
    for (int64_t i = wave0; i < n; i += nwaves) {
        int64_t row, grow;
        if (ALL) { row = i; grow = map ? (int64_t)map[i] : i; }
        else { row = slot_row[i]; grow = i; }
        const size_t base = (size_t)row * W;
        const float *g = grow >= 0 ? G + (size_t)grow * W : nullptr;
        for (int e = lane * VEC; e < W; e += 64 * VEC) {
            vf v = *reinterpret_cast<const vf *>(var + base + e);
            vf a = vf(0.f), b = vf(0.f), gg = vf(0.f);
            if (one) a = *reinterpret_cast<const vf *>(s0 + base + e);
            if (two) b = *reinterpret_cast<const vf *>(s1 + base + e);
            if (g) gg = *reinterpret_cast<const vf *>(g + e) * scale;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float vv = lane_get(v, j), aa = lane_get(a, j), bb = lane_get(b, j);
                apply_one(r, lane_get(gg, j), vv, aa, bb);
                fin = fmaf(vv, 0.f, fin);
                lane_set(v, j, vv); lane_set(a, j, aa); lane_set(b, j, bb);
            }
            *reinterpret_cast<vf *>(var + base + e) = v;
            if (one) *reinterpret_cast<vf *>(s0 + base + e) = a;
            if (two) *reinterpret_cast<vf *>(s1 + base + e) = b;
        }
    }
    // a diverged run: the forward kernels stop leaving out the rows of weight-0 categories (0 * inf = NaN, :82)
    if (fin != fin) *nonfinite = 1;
}

__global__ __launch_bounds__(256) void m2d_train_cleanup(int32_t *map, const int32_t *slot_row, const int32_t *count, float *G, int32_t W,
                                                         const int32_t *err, OptState *advance)
{
    if (advance && blockIdx.x == 0 && threadIdx.x == 0 && err[0] == 0) {     // the step applied: AdamOptimizer._finish
        advance->b1p *= 0.9f;
        advance->b2p *= 0.999f;
        advance->steps += 1;
    }
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t n = *count;
    for (int64_t s = wave0; s < n; s += (int64_t)gridDim.x * 4) {
        if (lane == 0) map[slot_row[s]] = -1;
        for (int e = lane; e < W; e += 64) G[(size_t)s * W + e] = 0.f;
    }
}

// ---- the fused form for small batches (see the header comment) ---------------------------------------------------------
// A row's slot in the compact gradient buffer is claimed by the first wave that meets the row (map entry -1 free, -2 being
// claimed, else the slot number); a wave that finds -2 waits for the number.
__global__ __launch_bounds__(256) void m2d_train_grad_fused(TrainArgs p)
{
    extern __shared__ float dce_all[];                      // [4 waves][C, E] partial dCE
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int C = p.C, E = p.E;
    float *dce = dce_all + (size_t)wv * C * E;
    int32_t *cnt = p.cnt + 2 * p.parity;
    for (int i = lane; i < C * E; i += 64) dce[i] = 0.f;
    __syncthreads();
    double loss_acc = 0.0, sq_acc = 0.0;
    const float invB = 1.0f / (float)p.B;
    for (int64_t b = wave0; b < p.B; b += nwaves) {
        const int32_t uid = p.users[b], did = p.items[b];
        const int64_t ul = (int64_t)uid - p.user_base;
        if (ul < 0 || ul >= p.U || did < 0 || (int64_t)did >= p.I) {      // TF raises from the gather: the step applies nothing
            if (lane == 0)
                train_latch(p.err, (ul < 0 || ul >= p.U) ? M2D_ERR_BAD_USER_ID : M2D_ERR_BAD_ITEM_ID,
                            (ul < 0 || ul >= p.U) ? uid : did, b);
            continue;
        }
        // The pair's two claims go out together (lane 0: the user's slot, lane 1: the dish's) and are looked at only after the
        // forward pass, whose row reads do not depend on them: one round trip to memory instead of four in front of the rows.
        const bool claimer = p.accumulate && lane < 2;
        int32_t *const cmap = lane == 0 ? p.map_u : p.map_d;
        const int64_t crow = lane == 0 ? ul : (int64_t)did;
        int seen = 0;
        if (claimer) seen = atomicCAS(&cmap[crow], -1, -2);
        const float y = p.labels[b];
        const float *m = p.cats + (size_t)b * C;
        const float *urow = p.pm + (size_t)ul * (C + 1) * E;
        const float *drow = p.re + (size_t)did * E;
        float n = 0.f;
        for (int c = 0; c < C; ++c) n += m[c];                                      // :77
        float hi = 0.f, lo = 0.f;
        for (int e = lane; e < E; e += 64) {
            float H = 0.f, L = 0.f;
            for (int c = 0; c < C; ++c) {
                H = fmaf(m[c], p.ce[(size_t)c * E + e], H);                         // :67-75
                L = fmaf(m[c], urow[(size_t)(1 + c) * E + e], L);                   // :82-90
            }
            hi = fmaf(urow[e], H, hi);
            lo = fmaf(drow[e], L, lo);
        }
        hi = wave_sum(hi);
        lo = wave_sum(lo);
        const float s = m2d_blend_unfused(p.a, hi / n, p.b, lo / n);  // :79, :93, :95-96
        // -1: the row was free and this lane numbers it; anything else: the number, once it is there.  Two statements, in this
        // order, not the two arms of one `if`: a wave whose lane 0 numbers a row while its lane 1 waits for another wave's
        // number must publish BEFORE it waits -- with the arms in the other order two such waves, each holding the row the
        // other waits for, would spin for ever (the arms of a divergent branch run one after the other).
        int slot = seen;
        if (claimer && seen == -1) {
            slot = atomicAdd(cnt + lane, 1);
            (lane == 0 ? p.slot_u : p.slot_d)[slot] = (int32_t)crow;
            // relaxed, agent scope: the number itself is all a waiting wave takes from this one (slot_row is read by the apply
            // launch).  Release / acquire here cost an L2 write-back per claim and an invalidate per look -- the XCDs' L2s
            // are not coherent with each other -- for an ordering nobody uses.
            __hip_atomic_store(&cmap[crow], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);            // (the compiler keeps the order; the wave issues its stores before the loads below)
        if (claimer && seen != -1)
            while (slot < 0) slot = __hip_atomic_load(&cmap[crow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int su = __builtin_amdgcn_readlane(slot, 0), sd = __builtin_amdgcn_readlane(slot, 1);
        const float loss_b = fmaxf(s, 0.f) - s * y + log1pf(expf(-fabsf(s)));       // :101
        const float gs = (1.0f / (1.0f + expf(-s)) - y) * invB;                     // d mean / d s_b
        const float qh = gs * p.a / n, ql = gs * p.b / n;
        float *gu = p.gu + (size_t)su * (C + 1) * E;
        float *gd = p.gd + (size_t)sd * E;
        float sq = 0.f;
        for (int e = lane; e < E; e += 64) {
            float H = 0.f, L = 0.f;
            for (int c = 0; c < C; ++c) {
                H = fmaf(m[c], p.ce[(size_t)c * E + e], H);
                L = fmaf(m[c], urow[(size_t)(1 + c) * E + e], L);
            }
            const float uh = urow[e], it = drow[e];
            const float v0 = qh * H, vd = ql * L;
            sq = fmaf(v0, v0, sq);
            sq = fmaf(vd, vd, sq);
            if (p.accumulate) {
                atomicAdd(gu + e, v0);
                atomicAdd(gd + e, vd);
            }
            for (int c = 0; c < C; ++c) {
                const float vc = (ql * m[c]) * it;
                sq = fmaf(vc, vc, sq);
                if (p.accumulate && m[c] != 0.f) atomicAdd(gu + (size_t)(1 + c) * E + e, vc);
                dce[c * E + e] += (qh * m[c]) * uh;                                 // this lane owns column e
            }
        }
        sq_acc += (double)wave_sum(sq);
        loss_acc += (double)loss_b;
    }
    __shared__ double wave_acc[4][2];
    __shared__ int s_ticket;
    if (lane == 0) { wave_acc[wv][0] = loss_acc; wave_acc[wv][1] = sq_acc; }
    __syncthreads();
    // dCE: a few dozen blocks at most -- each adds its C x E sums into the (zeroed) dense gradient; the apply pass zeroes it again
    for (int i = threadIdx.x; i < C * E; i += 256) {
        const float t = (dce_all[i] + dce_all[C * E + i]) + (dce_all[2 * C * E + i] + dce_all[3 * C * E + i]);
        if (t != 0.f) atomicAdd(p.gce + i, t);
    }
    if (threadIdx.x < 2)
        p.part_acc[(size_t)blockIdx.x * 2 + threadIdx.x] =
            (wave_acc[0][threadIdx.x] + wave_acc[1][threadIdx.x]) + (wave_acc[2][threadIdx.x] + wave_acc[3][threadIdx.x]);
    // the last block to get here finishes the step (every other block's partials and gradient rows are then in memory)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(p.done, 1);
    __syncthreads();
    if (s_ticket != (int)gridDim.x - 1) return;
    __threadfence();
    __shared__ double red[4][3];
    double s2 = 0.0, l = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < C * E; i += 256) {        // |dCE|^2 (values other blocks added: read past this CU's L1)
        const float t = __builtin_nontemporal_load(p.gce + i);
        s2 += (double)t * (double)t;
        if (!p.accumulate) p.gce[i] = 0.f;                  // a loss-only call has no apply pass to zero it
    }
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) {   // the blocks' loss / square sums (at most 256 blocks: one each)
        l += __builtin_nontemporal_load(p.part_acc + (size_t)b * 2);
        q += __builtin_nontemporal_load(p.part_acc + (size_t)b * 2 + 1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s2 += __shfl_xor(s2, off, 64);
        l += __shfl_xor(l, off, 64);
        q += __shfl_xor(q, off, 64);
    }
    if (lane == 0) { red[wv][0] = s2; red[wv][1] = l; red[wv][2] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        l = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        q = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
        const float norm = (float)sqrt(q + ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])));
        const float loss = (float)(l / (double)p.B), scale = p.clip * fminf(1.0f / norm, 1.0f / p.clip);
        p.scal[0] = loss; p.scal[1] = norm; p.scal[2] = scale; p.scal[3] = p.lr;
        if (p.out) { p.out[0] = loss; p.out[1] = norm; p.out[2] = scale; p.out[3] = p.lr; }
        float lr_t = p.lr;
        if (p.learner == M2D_LEARNER_ADAM) lr_t = p.lr * sqrtf(1.0f - p.opt->b2p) / (1.0f - p.opt->b1p);   // AdamOptimizer._apply_dense
        p.scal[4] = lr_t;
        if (p.accumulate && __hip_atomic_load(&p.err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {   // the step applies: AdamOptimizer._finish
            p.opt->b1p *= 0.9f;
            p.opt->b2p *= 0.999f;
            p.opt->steps += 1;
        }
        p.cnt[2 * (p.parity ^ 1)] = 0;                       // the next step's slot counters (its own were read by this step's apply)
        p.cnt[2 * (p.parity ^ 1) + 1] = 0;
        *p.done = 0;
    }
}

// One grid over the three tables.  Table t holds R[t] rows of W[t] floats; its work items are all rows (ALL: Adam, and the
// dense Category_Embedding gradient) or the claimed rows slot_row[0 .. *count).  After a row is applied (or found not to
// be -- an id error was latched: nothing is assigned) its slot and gradient row are released by the same wave.
struct ApplyTab {
    float *var, *s0, *s1, *G;
    int32_t *map, *slot_row;
    const int32_t *count;
    int64_t R;
    int32_t W, all;
};
struct ApplyArgs {
    ApplyTab t[3];
    const float *scal;
    const int32_t *err;
    int32_t *nonfinite;
    RuleArgs r;
};

__global__ __launch_bounds__(256) void m2d_train_apply_fused(ApplyArgs p)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const bool ok = __builtin_amdgcn_readfirstlane(p.err[0]) == 0;
    RuleArgs r = p.r;
    const float scale = p.scal[2];
    r.lr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.scal[4])));   // Adam: lr_t of this step
    const bool two = r.rule == M2D_LEARNER_ADAM || r.rule == M2D_LEARNER_RMSPROP, one = two || r.rule == M2D_LEARNER_ADAGRAD;
    float fin = 0.f;
    int64_t n[3], tot = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        n[t] = p.t[t].all ? p.t[t].R : (int64_t)*p.t[t].count;
        tot += n[t];
    }
    for (int64_t i = wave0; i < tot; i += nwaves) {
        const int t = i < n[0] ? 0 : (i < n[0] + n[1] ? 1 : 2);
        const ApplyTab &tb = p.t[t];
        const int64_t k = i - (t == 0 ? 0 : (t == 1 ? n[0] : n[0] + n[1]));
        int64_t row, grow;
        if (tb.all) { row = k; grow = tb.map ? (int64_t)tb.map[k] : k; }
        else { row = tb.slot_row[k]; grow = k; }
        const int W = tb.W;
        const size_t base = (size_t)row * W;
        float *g = grow >= 0 ? tb.G + (size_t)grow * W : nullptr;
        // rows of whole, 16-B aligned float4 (every embedding size that is a multiple of 4; the caller's table may start anywhere)
        const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(tb.var) & 15) == 0;
        if (ok && (g || r.rule == M2D_LEARNER_ADAM)) {     // Adam decays and moves every row; the others touch rows with a gradient
            if (vec) {                                      // 16 B per lane: the dense Adam pass is a stream over var / m / v
                for (int e = lane * 4; e < W; e += 256) {
                    v4f v = *reinterpret_cast<const v4f *>(tb.var + base + e), a = v4f(0.f), b = v4f(0.f), gg = v4f(0.f);
                    if (one) a = *reinterpret_cast<const v4f *>(tb.s0 + base + e);
                    if (two) b = *reinterpret_cast<const v4f *>(tb.s1 + base + e);
                    if (g) gg = *reinterpret_cast<const v4f *>(g + e) * scale;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float vv = v[j], aa = a[j], bb = b[j];
                        apply_one(r, gg[j], vv, aa, bb);
                        fin = fmaf(vv, 0.f, fin);
                        v[j] = vv; a[j] = aa; b[j] = bb;
                    }
                    *reinterpret_cast<v4f *>(tb.var + base + e) = v;
                    if (one) *reinterpret_cast<v4f *>(tb.s0 + base + e) = a;
                    if (two) *reinterpret_cast<v4f *>(tb.s1 + base + e) = b;
                }
            } else {
                for (int e = lane; e < W; e += 64) {
                    float v = tb.var[base + e], a = one ? tb.s0[base + e] : 0.f, b = two ? tb.s1[base + e] : 0.f;
                    apply_one(r, g ? g[e] * scale : 0.f, v, a, b);
                    fin = fmaf(v, 0.f, fin);
                    tb.var[base + e] = v;
                    if (one) tb.s0[base + e] = a;
                    if (two) tb.s1[base + e] = b;
                }
            }
        }
        if (g && (tb.map || t == 2)) {                      // release: the gradient row back to zero, the slot back to free
            if (vec) { for (int e = lane * 4; e < W; e += 256) *reinterpret_cast<v4f *>(g + e) = v4f(0.f); }
            else { for (int e = lane; e < W; e += 64) g[e] = 0.f; }
            if (tb.map && lane == 0) tb.map[row] = -1;
        }
    }
    if (fin != fin) *p.nonfinite = 1;                       // a diverged run: the forward kernels stop leaving rows out (0 * inf = NaN, :82)
}

__global__ __launch_bounds__(256) void m2d_fill_kernel(float *x, int64_t n, float v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

__global__ __launch_bounds__(256) void m2d_fill_i32_kernel(int32_t *x, int64_t n, int32_t v)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = v;
}

unsigned blocks_for(const m2d_engine *h, int64_t waves)
{
    int64_t b = (waves + 3) / 4;
    const int64_t cap = (int64_t)h->num_cu * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

struct m2d_train_state {
    int32_t learner = M2D_LEARNER_ADAM;
    float lr = 0.001f, clip = 5.0f;
    OptState *opt = nullptr;            // device: step count and Adam's beta powers (see OptState)
    float *slot[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};   // [PM, RE, CE][slot]
    int32_t *map_u = nullptr, *map_d = nullptr, *slot_u = nullptr, *slot_d = nullptr, *cnt = nullptr;
    float *gu = nullptr, *gd = nullptr, *gce = nullptr, *scal = nullptr;
    float *part_ce = nullptr;           // [num_cu * 8, C, E]
    double *part_acc = nullptr;         // [num_cu * 8, 2]
    int64_t cap = 0;                    // pairs the compact buffers hold
    int32_t *done = nullptr;            // fused form: ticket counter of m2d_train_grad_fused
    int parity = 0;                     // fused form: which pair of slot counters the next step uses
};

void m2d_train_release(m2d_engine *h)
{
    m2d_train_state *t = h->train;
    if (!t) return;
    for (auto &tb : t->slot)
        for (float *q : tb)
            if (q) (void)hipFree(q);
    for (void *q : {(void *)t->map_u, (void *)t->map_d, (void *)t->slot_u, (void *)t->slot_d, (void *)t->cnt, (void *)t->gu,
                    (void *)t->gd, (void *)t->gce, (void *)t->scal, (void *)t->part_ce, (void *)t->part_acc, (void *)t->opt, (void *)t->done})
        if (q) (void)hipFree(q);
    delete t;
    h->train = nullptr;
}

int m2d_train_setup(m2d_engine *h, int32_t learner, float lr, float clip_norm, hipStream_t stream)
{
    m2d_train_release(h);
    m2d_train_state *t = new m2d_train_state;
    h->train = t;
    t->learner = learner; t->lr = lr; t->clip = clip_norm;
    const int64_t n[3] = {h->U * (int64_t)(h->C + 1) * h->E, h->I * (int64_t)h->E, (int64_t)h->C * h->E};
    const int nslots = (learner == M2D_LEARNER_ADAM || learner == M2D_LEARNER_RMSPROP) ? 2 : learner == M2D_LEARNER_ADAGRAD ? 1 : 0;
    for (int tb = 0; tb < 3; ++tb)
        for (int s = 0; s < nslots; ++s) {
            M2D_HIP_TRY(h, hipMalloc((void **)&t->slot[tb][s], (size_t)n[tb] * 4));
            // slot initial values: adam m = v = 0; adagrad accumulator 0.1; rmsprop rms = 1, momentum = 0
            const float v0 = learner == M2D_LEARNER_ADAGRAD ? 0.1f : (learner == M2D_LEARNER_RMSPROP && s == 0) ? 1.0f : 0.0f;
            hipLaunchKernelGGL(m2d_fill_kernel, dim3(blocks_for(h, n[tb] / 64 + 1)), dim3(256), 0, stream, t->slot[tb][s], n[tb], v0);
        }
    M2D_HIP_TRY(h, hipMalloc((void **)&t->map_u, (size_t)h->U * 4));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->map_d, (size_t)h->I * 4));
    hipLaunchKernelGGL(m2d_fill_i32_kernel, dim3(blocks_for(h, h->U / 64 + 1)), dim3(256), 0, stream, t->map_u, h->U, -1);
    hipLaunchKernelGGL(m2d_fill_i32_kernel, dim3(blocks_for(h, h->I / 64 + 1)), dim3(256), 0, stream, t->map_d, h->I, -1);
    M2D_HIP_TRY(h, hipMalloc((void **)&t->cnt, 4 * 4));                 // two pairs: the fused form alternates between them
    M2D_HIP_TRY(h, hipMemsetAsync(t->cnt, 0, 16, stream));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->done, 4));
    M2D_HIP_TRY(h, hipMemsetAsync(t->done, 0, 4, stream));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->gce, (size_t)n[2] * 4));
    M2D_HIP_TRY(h, hipMemsetAsync(t->gce, 0, (size_t)n[2] * 4, stream));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->scal, 8 * 4));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->part_ce, (size_t)h->num_cu * 8 * n[2] * 4));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->part_acc, (size_t)h->num_cu * 8 * 2 * 8));
    M2D_HIP_TRY(h, hipMemsetAsync(t->scal, 0, 32, stream));
    M2D_HIP_TRY(h, hipMalloc((void **)&t->opt, sizeof(OptState)));
    const OptState st0 = {0.9f, 0.999f, 0};
    M2D_HIP_TRY(h, hipMemcpyAsync(t->opt, &st0, sizeof st0, hipMemcpyHostToDevice, stream));
    M2D_HIP_TRY(h, hipStreamSynchronize(stream));      // st0 is on this frame
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}

int m2d_launch_train_step(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, const float *labels,
                          int64_t B, int32_t apply, float *out, hipStream_t stream)
{
    m2d_train_state *t = h->train;
    const int C = h->C, E = h->E, W = (C + 1) * E;
    if (apply && B > t->cap) {          // compact gradient rows: at most one per pair
        M2D_HIP_TRY(h, hipStreamSynchronize(stream));
        for (void *q : {(void *)t->slot_u, (void *)t->slot_d, (void *)t->gu, (void *)t->gd})
            if (q) (void)hipFree(q);
        t->slot_u = t->slot_d = nullptr; t->gu = t->gd = nullptr;
        M2D_HIP_TRY(h, hipMalloc((void **)&t->slot_u, (size_t)B * 4));
        M2D_HIP_TRY(h, hipMalloc((void **)&t->slot_d, (size_t)B * 4));
        M2D_HIP_TRY(h, hipMalloc((void **)&t->gu, (size_t)B * W * 4));
        M2D_HIP_TRY(h, hipMalloc((void **)&t->gd, (size_t)B * E * 4));
        M2D_HIP_TRY(h, hipMemsetAsync(t->gu, 0, (size_t)B * W * 4, stream));
        M2D_HIP_TRY(h, hipMemsetAsync(t->gd, 0, (size_t)B * E * 4, stream));
        t->cap = B;
    }
    TrainArgs a;
    a.pm = const_cast<float *>(h->pm); a.re = const_cast<float *>(h->re); a.ce = const_cast<float *>(h->ce);
    a.users = users; a.items = items; a.cats = cats; a.labels = labels;
    a.B = B; a.U = h->U; a.I = h->I; a.user_base = h->user_base; a.C = C; a.E = E; a.a = h->a; a.b = h->b;
    a.map_u = t->map_u; a.map_d = t->map_d; a.slot_u = t->slot_u; a.slot_d = t->slot_d; a.cnt = t->cnt;
    a.gu = t->gu; a.gd = t->gd; a.gce = t->gce; a.scal = t->scal; a.err = h->err_dev;
    a.accumulate = apply ? 1 : 0; a.clip = t->clip; a.lr = t->lr;      // Global_Step never moves (:240): lr is constant
    a.done = t->done; a.out = out; a.opt = t->opt; a.learner = t->learner; a.parity = t->parity;
    a.part_ce = t->part_ce; a.part_acc = t->part_acc;
    const size_t lds_f = (size_t)4 * C * E * 4;
    if (B <= 1024 && lds_f <= 48 * 1024 && h->opt_variant != 14) {     // the fused form ("variant" = 14: the nine-launch form, A/B)
        a.nblocks = (int32_t)blocks_for(h, B);
        hipLaunchKernelGGL(m2d_train_grad_fused, dim3((unsigned)a.nblocks), dim3(256), lds_f, stream, a);
        M2D_HIP_TRY(h, hipGetLastError());
        h->last_kernel = "m2d_train_grad_fused";
        if (!apply) return M2D_OK;
        ApplyArgs ap;
        const bool adam = t->learner == M2D_LEARNER_ADAM;
        int32_t *cnt = t->cnt + 2 * t->parity;
        ap.t[0] = {a.pm, t->slot[0][0], t->slot[0][1], t->gu, t->map_u, t->slot_u, cnt + 0, h->U, W, adam ? 1 : 0};
        ap.t[1] = {a.re, t->slot[1][0], t->slot[1][1], t->gd, t->map_d, t->slot_d, cnt + 1, h->I, E, adam ? 1 : 0};
        ap.t[2] = {a.ce, t->slot[2][0], t->slot[2][1], t->gce, nullptr, nullptr, nullptr, C, E, 1};
        ap.scal = t->scal; ap.err = h->err_dev; ap.nonfinite = h->nonfinite_dev;
        ap.r.rule = t->learner; ap.r.lr = t->lr; ap.r.b1 = ap.r.b2 = ap.r.eps = 0.f;
        if (adam) { ap.r.b1 = 0.9f; ap.r.b2 = 0.999f; ap.r.eps = 1e-8f; }
        else if (t->learner == M2D_LEARNER_RMSPROP) { ap.r.b1 = 0.9f; ap.r.b2 = 0.0f; ap.r.eps = 1e-10f; }
        const int64_t rows = adam ? h->U + h->I + C : 2 * B + C;
        hipLaunchKernelGGL(m2d_train_apply_fused, dim3(blocks_for(h, rows)), dim3(256), 0, stream, ap);
        M2D_HIP_TRY(h, hipGetLastError());
        t->parity ^= 1;
        h->dish_vec_valid = false;      // everything derived from the tables is stale now
        h->grp_valid = false;
        h->user_high_valid = false;
        return M2D_OK;
    }
    int32_t *cnt_now = t->cnt + 2 * t->parity;              // (the fused form may have left the other pair in use)
    a.cnt = cnt_now;
    M2D_HIP_TRY(h, hipMemsetAsync(cnt_now, 0, 8, stream));
    M2D_HIP_TRY(h, hipMemsetAsync(t->gce, 0, (size_t)C * E * 4, stream));
    if (apply) {
        hipLaunchKernelGGL(m2d_train_claim, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, a);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    const size_t lds = (size_t)4 * C * E * 4;
    const int ce_lds = lds <= 48 * 1024;
    a.part_ce = t->part_ce; a.part_acc = t->part_acc;
    a.nblocks = (int32_t)blocks_for(h, B);
    hipLaunchKernelGGL(m2d_train_grad, dim3((unsigned)a.nblocks), dim3(256), ce_lds ? lds : 0, stream, a, ce_lds);
    M2D_HIP_TRY(h, hipGetLastError());
    if (ce_lds) {
        const unsigned ry = a.nblocks >= 512 ? 8 : 1;
        hipLaunchKernelGGL(m2d_train_reduce_ce, dim3((unsigned)((C * E + 63) / 64), ry), dim3(1024), 0, stream, a);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(m2d_train_finalize, dim3(1), dim3(256), 0, stream, a);
    M2D_HIP_TRY(h, hipGetLastError());
    if (out) M2D_HIP_TRY(h, hipMemcpyAsync(out, t->scal, 16, hipMemcpyDeviceToDevice, stream));
    h->last_kernel = "m2d_train_grad";
    if (!apply) {
        M2D_HIP_TRY(h, hipMemsetAsync(t->gce, 0, (size_t)C * E * 4, stream));   // (the fused form expects the dense gradient at zero)
        return M2D_OK;
    }

    RuleArgs r;
    r.rule = t->learner; r.lr = t->lr; r.b1 = r.b2 = r.eps = 0.f;
    if (t->learner == M2D_LEARNER_ADAM) {
        r.b1 = 0.9f; r.b2 = 0.999f; r.eps = 1e-8f;
    } else if (t->learner == M2D_LEARNER_RMSPROP) {
        r.b1 = 0.9f; r.b2 = 0.0f; r.eps = 1e-10f;
    }
    struct Tab { float *var; float *G; int32_t *map; int32_t *slot_row; int32_t *count; int64_t R; int32_t W; int idx; };
    const Tab tabs[3] = {{a.pm, t->gu, t->map_u, t->slot_u, a.cnt + 0, h->U, W, 0},
                         {a.re, t->gd, t->map_d, t->slot_d, a.cnt + 1, h->I, E, 1},
                         {a.ce, t->gce, nullptr, nullptr, nullptr, C, E, 2}};
    for (const Tab &tb : tabs) {
        float *s0 = t->slot[tb.idx][0], *s1 = t->slot[tb.idx][1];
        const bool all = t->learner == M2D_LEARNER_ADAM || !tb.map;       // dense gradient: every row
        const int64_t rows = all ? tb.R : B;
        const unsigned grid = blocks_for(h, rows);
        const bool v4 = tb.W % 4 == 0;
#define M2D_APPLY(ALL, VEC)                                                                                               \
    hipLaunchKernelGGL((m2d_train_apply<ALL, VEC>), dim3(grid), dim3(256), 0, stream, tb.var, s0, s1, tb.G, tb.map, tb.slot_row, \
                       tb.count, tb.R, tb.W, t->scal, r, h->err_dev, t->opt, h->nonfinite_dev)
        if (all) { if (v4) M2D_APPLY(true, 4); else M2D_APPLY(true, 1); }
        else { if (v4) M2D_APPLY(false, 4); else M2D_APPLY(false, 1); }
#undef M2D_APPLY
        M2D_HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(m2d_train_cleanup, dim3(blocks_for(h, B)), dim3(256), 0, stream, t->map_u, t->slot_u, a.cnt + 0, t->gu, W,
                       h->err_dev, (OptState *)nullptr);
    hipLaunchKernelGGL(m2d_train_cleanup, dim3(blocks_for(h, B)), dim3(256), 0, stream, t->map_d, t->slot_d, a.cnt + 1, t->gd, E,
                       h->err_dev, t->opt);               // also advances the step count / beta powers if the step applied
    M2D_HIP_TRY(h, hipGetLastError());
    M2D_HIP_TRY(h, hipMemsetAsync(cnt_now, 0, 8, stream));   // the fused form expects its slot counters ...
    M2D_HIP_TRY(h, hipMemsetAsync(t->gce, 0, (size_t)C * E * 4, stream));   // ... and the dense gradient at zero
    // everything derived from Recipe_Embedding / Category_Embedding is stale now
    h->dish_vec_valid = false;
    h->grp_valid = false;
    h->user_high_valid = false;
    return M2D_OK;
}

// steps applied so far; setting it (checkpoint resume) also replays Adam's beta-power products, which TF keeps as
// float32 variables multiplied by beta once per step
int m2d_train_step_count(m2d_engine *h, int64_t *steps, int32_t set)
{
    m2d_train_state *t = h->train;
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    M2D_HIP_TRY(h, hipDeviceSynchronize());            // the count lives on the device; steps in flight finish first
    OptState st = {0.9f, 0.999f, 0};
    if (set) {
        st.steps = *steps;
        for (int64_t i = 0; i < st.steps; ++i) { st.b1p *= 0.9f; st.b2p *= 0.999f; }
        M2D_HIP_TRY(h, hipMemcpy(t->opt, &st, sizeof st, hipMemcpyHostToDevice));
    } else {
        M2D_HIP_TRY(h, hipMemcpy(&st, t->opt, sizeof st, hipMemcpyDeviceToHost));
        *steps = st.steps;
    }
    return M2D_OK;
}

int m2d_train_get_slot(m2d_engine *h, int32_t table, int32_t slot, float **dev, int64_t *count)
{
    m2d_train_state *t = h->train;
    const int64_t n[3] = {h->U * (int64_t)(h->C + 1) * h->E, h->I * (int64_t)h->E, (int64_t)h->C * h->E};
    *dev = t->slot[table][slot];
    *count = *dev ? n[table] : 0;
    return M2D_OK;
}
