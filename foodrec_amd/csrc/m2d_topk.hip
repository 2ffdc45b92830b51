// Ranking kernels for gfx950: the evaluator's per-user candidate ranking (evaluate.py:35-63) and
// full-catalogue retrieval.  Scores always come from the fused pair-score kernels (m2d_score.hip).
#include <math.h>

#include "m2d_engine.h"

namespace {

// ---- scratch ------------------------------------------------------------------------------------
int ensure_scratch(m2d_engine *h, size_t bytes)
{
    if (h->scratch_bytes >= bytes) return M2D_OK;
    if (h->scratch) M2D_HIP_TRY(h, hipFree(h->scratch));
    h->scratch = nullptr;
    h->scratch_bytes = 0;
    size_t want = bytes + (bytes >> 2);
    M2D_HIP_TRY(h, hipMalloc((void **)&h->scratch, want));
    h->scratch_bytes = want;
    return M2D_OK;
}

// users[s] -> one id per candidate slot; padded slots (pos >= len) score dish 0 and are ignored later
__global__ void m2d_expand_segments(const int32_t *users, const int32_t *items, const int32_t *lens,
                                    int64_t nseg, int32_t L, int32_t *users_x, int32_t *items_x)
{
    const int64_t total = nseg * (int64_t)L;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = t / L;
        const int32_t pos = (int32_t)(t - s * L);
        const int32_t len = lens ? min(max(lens[s], 0), L) : L;
        users_x[t] = users[s];
        items_x[t] = pos < len ? items[t] : 0;
    }
}

// One wave per segment.  Reproduces
//     for i: table[item_i] = score_i          (evaluate.py:60-61; a key keeps its FIRST position,
//                                              its value is the LAST score written)
//     heapq.nlargest(K, table, key=table.get) (evaluate.py:63; descending, ties -> earlier key)
// by computing, for every first-occurrence candidate, its rank among the first occurrences.
constexpr int RANK_WAVES = 4;

__global__ __launch_bounds__(RANK_WAVES * 64) void m2d_rank_segments(
    const float *scores, const int32_t *items, const int32_t *lens, int64_t nseg, int32_t L, int32_t k,
    float *out_scores, int32_t *out_items, int32_t *out_flags)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int32_t *s_item = reinterpret_cast<int32_t *>(smem) + (size_t)wave * 3 * L;  // as fed
    int32_t *s_key = s_item + L;                                 // item if first occurrence, else -1
    float *s_val = reinterpret_cast<float *>(s_key + L);         // collapsed value of that key
    const int64_t seg = (int64_t)blockIdx.x * RANK_WAVES + wave;
    const bool live = seg < nseg;
    const int32_t len = live ? (lens ? min(max(lens[seg], 0), L) : L) : 0;
    const float *sc = scores + (live ? seg : 0) * (int64_t)L;
    const int32_t *it = items + (live ? seg : 0) * (int64_t)L;

    for (int i = lane; i < len; i += 64) s_item[i] = it[i];
    __syncthreads();

    bool any_nan = false;
    int nkeys = 0;
    for (int i = lane; i < len; i += 64) {
        const int32_t me = s_item[i];
        bool first = true;
        int last = i;
        for (int jj = 0; jj < len; ++jj) {
            if (s_item[jj] == me) {
                first = first && (jj >= i);
                last = max(last, jj);
            }
        }
        const float v = sc[last];
        s_val[i] = v;
        s_key[i] = first ? me : -1;
        any_nan = any_nan || (first && v != v);
        nkeys += first ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nkeys += __shfl_xor(nkeys, off, 64);
    __syncthreads();

    for (int i = lane; i < len; i += 64) {
        const int32_t key = s_key[i];
        if (key < 0) continue;
        const float v = s_val[i];
        int rank = 0;
        for (int jj = 0; jj < len; ++jj) {
            const float w = s_val[jj];
            const bool ahead = (w > v) || (w == v && jj < i);
            rank += (s_key[jj] >= 0 && ahead) ? 1 : 0;
        }
        if (rank < k) {
            out_scores[seg * k + rank] = v;
            out_items[seg * k + rank] = key;
        }
    }
    // the ranks of the distinct keys are a permutation of 0..nkeys-1: slots past them were not written
    if (live)
        for (int i = nkeys + lane; i < k; i += 64) {
            out_scores[seg * k + i] = __builtin_nanf("");
            out_items[seg * k + i] = -1;
        }
    const bool wave_nan = __any(any_nan);
    if (live && lane == 0) out_flags[seg] = wave_nan ? 1 : 0;
}

}  // namespace

int m2d_launch_rank_candidates(m2d_engine *h, const int32_t *users, const int32_t *items,
                               const int32_t *lens, int64_t nseg, int32_t L, int32_t k, float *out_scores,
                               int32_t *out_items, int32_t *out_flags, hipStream_t stream)
{
    if (nseg == 0) return M2D_OK;
    const int64_t total = nseg * (int64_t)L;
    // scratch: users_x i32[total] | items_x i32[total] | scores f32[total]
    int rc = ensure_scratch(h, (size_t)total * 12 + 256);
    if (rc != M2D_OK) return rc;
    int32_t *users_x = reinterpret_cast<int32_t *>(h->scratch);
    int32_t *items_x = users_x + total;
    float *scores = reinterpret_cast<float *>(items_x + total);

    int64_t eb = (total + 255) / 256;
    if (eb > 4096) eb = 4096;
    hipLaunchKernelGGL(m2d_expand_segments, dim3((unsigned)eb), dim3(256), 0, stream, users, items, lens, nseg,
                       L, users_x, items_x);
    M2D_HIP_TRY(h, hipGetLastError());
    rc = m2d_launch_score_pairs(h, users_x, items_x, h->dish_cats, /*by_dish=*/true, total, scores, stream);
    if (rc != M2D_OK) return rc;
    const int64_t rb = (nseg + RANK_WAVES - 1) / RANK_WAVES;
    const size_t lds = (size_t)RANK_WAVES * 3 * L * 4;
    hipLaunchKernelGGL(m2d_rank_segments, dim3((unsigned)rb), dim3(RANK_WAVES * 64), lds, stream, scores,
                       items_x, lens, nseg, L, k, out_scores, out_items, out_flags);
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}
