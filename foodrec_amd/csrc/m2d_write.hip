// Memory write (training side): Model.Write_Memory, Model_Recommender.py:106-220, as a scatter-add.
//
// The reference scatters through dense one-hot batched matmuls -- one_hot(user, num_users) [B, U, 1] times
// a [B, 1, (C+1)E] row (:151-158, :190, :201, :207) -- O(B * U * (C+1) * E) work and memory, which is why
// its driver feeds this branch 8 pairs at a time (Train_recommender.py:170-186).  Semantically it is
//
//   n_b   = sum_c m_bc                                                          :130
//   v_b   = [ beta_2 s_b (sum_c m_bc CE_c) / n_b ;  beta_1 s_b m_bc RE[d_b] , c = 0..C-1 ]   :108-147, :160
//   g_b   = (sum_l y_bl GM[l]) / (sum_l y_bl)          (GM = General_Memory before this call)  :166-184
//   PM[u] += sum_{b: u_b = u} ( v_b + alpha * g_b )                                            :162, :186-198
//   GM[l] += sum_b y_bl v_b                                                                    :200-215
//
// with s = write_sign, y = user_one_hot_label, m = categories.  Here it is O(B * (C+1) * E): one wave per
// pair adds its row into PM[u_b] (and into every GM[l] with y_bl != 0) with float atomics, 256 contiguous
// bytes per wave-instruction (the shape MI355X's memory-side atomics run fastest at).  Duplicate users in
// a batch accumulate, as reduce_sum(..., 0) does.  Sums are order-dependent in the last bits.
//
// Two launches keep the reference's data flow: every g_b reads General_Memory as it was before the call,
// so the PM pass (which reads GM) completes before the GM pass writes it.
#include "m2d_engine.h"

namespace {

struct WriteArgs {
    float *pm;           // [U, C+1, E]  (written)
    const float *re;     // [I, E]
    const float *ce;     // [C, E]
    float *gm;           // [L, C+1, E]  (read by pass 0, written by pass 1)
    const int32_t *users;
    const int32_t *items;
    const float *cats;   // [B, C]
    const float *sign;   // [B]
    const float *labels; // [B, L]
    int64_t B, U, I, user_base;
    int32_t C, E, L;
    float beta_1, beta_2, alpha;
    int32_t *err;
};

__device__ __forceinline__ void latch(int32_t *err, int code, int64_t value, int64_t index)
{
    if (atomicCAS(&err[0], 0, code) == 0) {
        err[1] = (int32_t)value;
        err[2] = (int32_t)(index & 0xffffffff);
        err[3] = (int32_t)(index >> 32);
    }
}

// PASS 0: PM[u_b] += v_b + alpha g_b.   PASS 1: GM[l] += y_bl v_b (after pass 0, which latched any id error).
// PASS 2: the GM pass on its own (the `general`-only fetch): latches id errors itself.
template <int PASS>
__global__ __launch_bounds__(256) void m2d_write_memory_kernel(WriteArgs p)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int C = p.C, E = p.E, L = p.L;
    for (int64_t b = wave0; b < p.B; b += nwaves) {
        const int32_t uid = p.users[b], did = p.items[b];
        const int64_t ul = (int64_t)uid - p.user_base;
        if (ul < 0 || ul >= p.U) {
            if (PASS != 1 && lane == 0) latch(p.err, M2D_ERR_BAD_USER_ID, uid, b);
            continue;   // wave-uniform: nothing is written for a bad pair
        }
        if (did < 0 || (int64_t)did >= p.I) {
            if (PASS != 1 && lane == 0) latch(p.err, M2D_ERR_BAD_ITEM_ID, did, b);
            continue;
        }
        const float s = p.sign[b];
        const float *m = p.cats + (size_t)b * C;
        const float *y = p.labels + (size_t)b * L;
        float n = 0.f;
        for (int c = 0; c < C; ++c) n += m[c];                               // :130
        // the labels of this pair with a non-zero weight, as lane masks (a user has a handful of its L = 95 labels):
        // every loop over labels below visits those only.  Walking all L for each of the (C + 1) E elements -- L
        // dependent loads per element -- was 200 us of latency for a single pair.
        constexpr int MAXM = 4;                                              // masks for L <= 256; beyond that: every label
        unsigned long long am[MAXM];
#pragma unroll
        for (int q = 0; q < MAXM; ++q) {
            const int l = q * 64 + lane;
            am[q] = (q * 64 < L && L <= 64 * MAXM) ? __ballot(l < L && y[l < L ? l : 0] != 0.f) : 0ull;
        }
        auto each_label = [&](auto &&f) __attribute__((always_inline)) {     // in label order, wave-uniform
            if (L <= 64 * MAXM) {
#pragma unroll
                for (int q = 0; q < MAXM; ++q)
                    for (unsigned long long bits = am[q]; bits; bits &= bits - 1) f(q * 64 + __builtin_ctzll(bits));
            } else {
                for (int l = 0; l < L; ++l)
                    if (y[l] != 0.f) f(l);
            }
        };
        float ysum = 0.f;
        if (PASS == 0) each_label([&](int l) { ysum += y[l]; });             // :180 -- the zero weights add nothing
        const float lo = p.beta_1 * s, hi = p.beta_2 * s;                    // :115, :141
        for (int r = 0; r <= C; ++r) {
            // General_Memory passes: the row of a category whose weight is 0 is a row of zeros -- nothing to add
            if (PASS != 0 && r > 0 && m[r - 1] == 0.f) continue;
            for (int e = lane; e < E; e += 64) {
                float v;
                if (r == 0) {
                    float dc = 0.f;
                    for (int c = 0; c < C; ++c) dc += m[c] * p.ce[(size_t)c * E + e];   // :124-128
                    v = (dc / n) * hi;                                       // :134, :145
                } else {
                    v = (m[r - 1] * p.re[(size_t)did * E + e]) * lo;         // :111, :119
                }
                const size_t k = (size_t)r * E + e;
                if (PASS == 0) {
                    float g = 0.f;
                    each_label([&](int l) { g = fmaf(y[l], p.gm[(size_t)l * (C + 1) * E + k], g); });   // :172-176
                    atomicAdd(p.pm + (size_t)ul * (C + 1) * E + k, v + p.alpha * (g / ysum));   // :162, :184-198
                } else {
                    each_label([&](int l) { atomicAdd(p.gm + (size_t)l * (C + 1) * E + k, y[l] * v); });   // :200-215
                }
            }
        }
    }
}

// The General_Memory assign for small batches, without atomics: GM[l][k] += sum_b y_bl v_b[k] with one wave per (label,
// 64 elements) walking the batch in order -- each (l, k) has one owner, so the sum has a fixed order and a few hundred
// pairs adding into the same 95 rows do not queue up at the memory-side atomic units (256 pairs: 78 us with atomics).
// LATCH: this launch also reports out-of-range ids (the `general`-only fetch has no other pass to do it).
template <bool LATCH>
__global__ __launch_bounds__(256) void m2d_write_gm_gather(WriteArgs p)
{
    const int lane = threadIdx.x & 63;
    const int C = p.C, E = p.E, L = p.L;
    const int W = (C + 1) * E, chunks = (W + 63) / 64;
    const int64_t wv = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv >= (int64_t)L * chunks) return;
    const int l = (int)(wv / chunks), k = (int)(wv % chunks) * 64 + lane;
    const bool kok = k < W;
    const int r = kok ? k / E : 0, e = kok ? k - r * E : 0;
    float acc = 0.f;
    for (int64_t b0 = 0; b0 < p.B; b0 += 64) {               // 64 pairs at a time: lane i looks at pair b0 + i
        const int64_t bi = b0 + lane;
        const bool in = bi < p.B;
        const int32_t uid = in ? p.users[bi] : (int32_t)p.user_base, did = in ? p.items[bi] : 0;
        const int64_t ul = (int64_t)uid - p.user_base;
        const bool bad = ul < 0 || ul >= p.U || did < 0 || (int64_t)did >= p.I;
        if (LATCH && wv == 0 && in && bad)
            latch(p.err, (ul < 0 || ul >= p.U) ? M2D_ERR_BAD_USER_ID : M2D_ERR_BAD_ITEM_ID, (ul < 0 || ul >= p.U) ? uid : did, bi);
        const float yw = (in && !bad) ? p.labels[(size_t)bi * L + l] : 0.f;   // nothing is written for a bad pair
        for (unsigned long long bits = __ballot(yw != 0.f); bits; bits &= bits - 1) {            // in batch order
            const int i = __builtin_ctzll(bits);
            const int64_t b = b0 + i;
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yw), i));
            const int32_t db = __builtin_amdgcn_readlane(did, i);
            const float *m = p.cats + (size_t)b * C;
            const float s = p.sign[b];
            float v;
            if (r == 0) {
                float n = 0.f, dc = 0.f;
                for (int c = 0; c < C; ++c) n += m[c];                                           // :130
                for (int c = 0; c < C; ++c) dc += m[c] * p.ce[(size_t)c * E + e];                // :124-128
                v = (dc / n) * (p.beta_2 * s);                                                   // :134, :145
            } else {
                v = (m[r - 1] * p.re[(size_t)db * E + e]) * (p.beta_1 * s);                      // :111, :119
            }
            acc += w * v;                                                                        // :200-215
        }
    }
    if (kok) p.gm[(size_t)l * W + k] += acc;
}

// sum of n floats into acc[0] (double), for the `personal` / `general` fetches (reduce_mean, :217-218)
__global__ __launch_bounds__(256) void m2d_sum_kernel(const float *x, int64_t n, double *acc)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, s);
}

}  // namespace

int m2d_launch_write_memory(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                            const float *sign, const float *labels, int64_t B, int32_t L, float *gm, float beta_1,
                            float beta_2, float alpha, int32_t which, double *out_sums, hipStream_t stream)
{
    if (which & M2D_WRITE_PERSONAL) {
        h->user_high_valid = false;                             // Personal_Memory is about to change
        h->grp_nonfinite_known = false;                         // ... and may receive inf / NaN: retrieval reads the device word again
    }
    WriteArgs a;
    a.pm = const_cast<float *>(h->pm); a.re = h->re; a.ce = h->ce; a.gm = gm;
    a.users = users; a.items = items; a.cats = cats; a.sign = sign; a.labels = labels;
    a.B = B; a.U = h->U; a.I = h->I; a.user_base = h->user_base; a.C = h->C; a.E = h->E; a.L = L;
    a.beta_1 = beta_1; a.beta_2 = beta_2; a.alpha = alpha; a.err = h->err_dev;
    if (B > 0) {
        int64_t blocks = (B + 3) / 4;
        if (blocks > (int64_t)h->num_cu * 8) blocks = (int64_t)h->num_cu * 8;
        // each pass runs only when its assign is fetched (`personal` -> :167/:198, `general` -> :215); a GM-only call
        // still validates the ids (the gathers at :107 / one_hot at :149 are shared by both branches)
        if (which & M2D_WRITE_PERSONAL) {
            hipLaunchKernelGGL(m2d_write_memory_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            M2D_HIP_TRY(h, hipGetLastError());
            // the blocks just added into: an inf / NaN there ends the forward kernels' row skipping (0 * inf = NaN, :82)
            const int rc = m2d_launch_rows_finite_check(h, users, B, stream);
            if (rc != M2D_OK) return rc;
        }
        if (which & M2D_WRITE_GENERAL) {
            if (B <= 2048) {      // few pairs, all of them adding into the same L rows: owner-computes instead of atomics
                const int64_t waves = (int64_t)L * (((h->C + 1) * h->E + 63) / 64);
                const dim3 g2((unsigned)((waves + 3) / 4));
                if (which & M2D_WRITE_PERSONAL) hipLaunchKernelGGL(m2d_write_gm_gather<false>, g2, dim3(256), 0, stream, a);
                else hipLaunchKernelGGL(m2d_write_gm_gather<true>, g2, dim3(256), 0, stream, a);
            } else if (which & M2D_WRITE_PERSONAL)
                hipLaunchKernelGGL(m2d_write_memory_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL(m2d_write_memory_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
            M2D_HIP_TRY(h, hipGetLastError());
        }
    }
    if (out_sums) {
        M2D_HIP_TRY(h, hipMemsetAsync(out_sums, 0, 2 * sizeof(double), stream));
        if (which & M2D_WRITE_PERSONAL)
            hipLaunchKernelGGL(m2d_sum_kernel, dim3((unsigned)(h->num_cu * 4)), dim3(256), 0, stream, h->pm,
                               (int64_t)h->U * (h->C + 1) * h->E, out_sums);
        if (which & M2D_WRITE_GENERAL)
            hipLaunchKernelGGL(m2d_sum_kernel, dim3(64), dim3(256), 0, stream, gm, (int64_t)L * (h->C + 1) * h->E,
                               out_sums + 1);
        M2D_HIP_TRY(h, hipGetLastError());
    }
    return M2D_OK;
}
