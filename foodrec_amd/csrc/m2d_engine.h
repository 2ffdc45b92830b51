// Engine state shared by the ABI layer and the kernel launchers (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <map>
#include <unordered_map>
#include <utility>

#include "../../include/m2d.h"

struct m2d_train_state;   // csrc/m2d_train.hip

struct m2d_engine {
    // tables in HBM, row-major, float32 (Model_Recommender.py:45-53)
    const float *pm = nullptr;  // [U, C+1, E]
    const float *re = nullptr;  // [I, E]
    const float *ce = nullptr;  // [C, E]
    const float *dish_cats = nullptr;  // [I, C] or null
    bool own_pm = false, own_re = false, own_ce = false, own_dish_cats = false;
    int64_t U = 0, I = 0;
    int32_t C = 0, E = 0;
    int64_t user_base = 0;
    float a = 0.f;  // float32(coef)             Model_Recommender.py:17
    float b = 0.f;  // 1.0f - a, taken in float32 Model_Recommender.py:96
    int device = 0;
    int num_cu = 256;

    // id-error latch: {code, bad value, index lo, index hi}; cleared by m2d_check
    int32_t *err_dev = nullptr;
    int32_t *err_host = nullptr;  // pinned

    // "some value of Personal_Memory / Recipe_Embedding / Category_Embedding is not finite" (device word, sticky until the
    // next full scan).  Model_Recommender.py:82-90 multiplies the row of a category a dish does not have by 0, and
    // 0 * inf = NaN: every kernel that leaves such rows out (option skip_masked, the pattern-grouped forms) reads this
    // word and fetches / multiplies everything when it is set.  Set by m2d_scan_tables (queued by m2d_create and
    // m2d_tables_updated, run by the next scoring call on its stream) and by the engine's own writers (training step,
    // Write_Memory) on the values they write.
    int32_t *nonfinite_dev = nullptr;
    bool finite_scan_pending = true;

    // build-defined extension: multi-hot ingredient table
    const float *ing = nullptr;        // [R, E]
    const int32_t *ing_off = nullptr;  // [I+1] CSR offsets per dish
    const int32_t *ing_ids = nullptr;  // [nnz]
    const float *ing_w = nullptr;      // [nnz] or null (all ones)
    bool own_ing = false;
    int64_t ing_rows = 0, ing_nnz = 0;
    float *dish_high = nullptr;        // [I, E]  H[d] = sum_j w_j ING[id_j] / sum_j w_j

    // build-defined extension: 3-layer scoring head
    const float *mlp_w1 = nullptr, *mlp_b1 = nullptr, *mlp_w2 = nullptr, *mlp_b2 = nullptr, *mlp_w3 = nullptr;
    float mlp_b3 = 0.f;
    int32_t mlp_h1 = 0, mlp_h2 = 0;
    bool own_mlp = false;
    void *mlp_w1x3 = nullptr;           // split-bf16 image of W1 for the bf16x3 layer-1 path (built lazily)
    float *mlp_w1pad = nullptr;         // W1 zero-padded to a multiple of 64 rows, for K = (C + 1) E that is not one (built lazily)
    void *mlp_w1pc = nullptr;           // W1 | W2 image of the producer / consumer kernel (built lazily)
    int32_t *mlp_pg = nullptr;          // per-launch pair grouping of that kernel: histogram | tile count | tile blocks | slot -> pair
    uint8_t *mlp_pat8 = nullptr;        // [I] a dish's pattern of non-zero mask weights (all blocks for n = 0 / NaN): what the grouping reads per pair
    uint64_t mlp_pat8_gen = 0;          // the dish-vector build it belongs to (dish_vec_gen)
    int64_t mlp_pat8_rows = 0;
    size_t mlp_pg_cap = 0;              // ints

    // derived table for pair scoring: <U_high[u], CE_c> per user and category (built lazily by large m2d_score_pairs calls;
    // stale after any write to Personal_Memory / Category_Embedding: the engine's own writers and m2d_tables_updated reset it)
    float *user_high = nullptr;  // [U, 4]
    bool user_high_valid = false;

    // factored dish vectors for catalogue retrieval (built lazily by m2d_topk_users)
    float *dish_vec = nullptr;  // [I_pad, (C+1)*E]
    int64_t dish_vec_rows = 0;
    bool dish_vec_valid = false;
    uint64_t dish_vec_gen = 0;          // counts m2d_ensure_dish_vectors' rebuilds (what hangs on the dish masks rebuilds with it)

    // pattern-grouped retrieval tables (0/1 masks only; built lazily by m2d_topk_users)
    float *grp_rs = nullptr;            // [grp_cap_rows, E] Recipe_Embedding rows sorted by (mask pattern, dish id)
    void *grp_rs16 = nullptr;           // the same rows split into bf16 hi | lo blocks per 32-row tile
    int32_t *grp_perm = nullptr;        // [grp_cap_rows]    slot -> dish id, -1 = padding
    int32_t *grp_tile_info = nullptr;   // [tiles]           pattern | valid rows << 8
    int32_t *grp_work = nullptr;        // block histograms / group offsets / flags
    int64_t grp_tiles = 0, grp_cap_rows = 0;
    int grp_ew = 0;                     // row width of grp_rs: E, or 2 E with the ingredient extension ([H[d] | RE[d]])
    bool grp_valid = false, grp_binary = false;
    bool grp_nonfinite = false;         // *nonfinite_dev as last read by the retrieval launcher (with the table build, or again
    bool grp_nonfinite_known = false;   //  after a writer that leaves the sorted dish rows alone: m2d_write_memory on Personal_Memory)

    // training step: optimizer slots and gradient scratch, created by m2d_train_begin
    m2d_train_state *train = nullptr;

    // staging for m2d_score_pairs_host: one pinned block and its device twin
    unsigned char *stage_host = nullptr, *stage_dev = nullptr;
    size_t stage_bytes = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};   // chunked host feeds: a block's chunk has been delivered
    uint32_t stage_ticket = 0;          // completion word of the last m2d_score_pairs_host call (wraps)

    // scratch for rank_candidates
    float *scratch = nullptr;
    size_t scratch_bytes = 0;
    float *topk_flags = nullptr;        // pattern-grouped retrieval: tie values per (user, split) / per user (NaN: no tie at the k-th score)
    size_t topk_flags_cap = 0;          // floats
    float *topk_tie_final = nullptr;    // the per-user values of the last call, inside topk_flags
    float *topk_plan = nullptr;         // pipelined retrieval kernel: per-user plan records | launch order | sort histogram | tile counter
    size_t topk_plan_cap = 0;           // floats
    unsigned long long *topk_tiles_counter = nullptr;   // tiles the blocks of the last pipelined launch stepped through (inside topk_plan)
    bool topk_apx_last = false;         // the last pipelined retrieval launch took the hi x hi first form (see "topk_tiles_completed")
    int64_t topk_tiles_full = 0;        // ... and what they would have stepped through without pattern pruning
    int32_t *topk_tie_list = nullptr;   // [0] users the tie repair re-ranked in the last pattern-grouped call (get_option "topk_repaired")
    int64_t topk_flags_used = 0;        // users of the last pattern-grouped call (get_option "topk_repaired" counts the non-NaN values)

    // benchmarking knobs
    int opt_prefetch = 2;
    int opt_nt = 1;
    int opt_blocks_per_cu = 8;
    int opt_variant = 0;
    int opt_user_high = 0;              // opt-in: batches of >= 2^18 pairs take the high-level sum from the derived <U_high, CE_c> table
    int opt_host_zero_copy = 2;         // m2d_score_pairs_host, <= 65536 pairs: 1 = the kernel reads / writes the pinned staging block itself,
                                        // 2 = and the host spins on a completion word in that block before falling back to a stream wait; 0 = staged copies
    int opt_skip_masked = 1;            // pair kernels: rows of categories with mask weight 0 are not fetched (their products are 0)
    int opt_mlp_form = 0;               // split-bf16 MLP head: 0 = matrix waves fed by gather / DMA waves (m2d_mlp_pc), 1 = every wave gathers its own rows
    int opt_mlp_bf16x3 = 1;             // MLP head layer 1 (build-defined) on split-bf16 MFMA; 0 = exact-f32 MFMA
    int opt_topk_bf16x3 = 1;            // retrieval (build-defined) on split-bf16 MFMA; 0 = exact-f32 MFMA
    int opt_topk_grouped = 1;           // 0/1-mask catalogues: pattern-grouped retrieval (contraction over E); 0 = dense kernel
    int opt_topk_form = 0;              // split-bf16 retrieval kernel: 0 / 2 = pipelined form, 1 = first form; 3 / 4 = hi x hi first form always / never
    int opt_topk_refine = 1;            // near-tied lists are finished in the tie repair's plain-f32 arithmetic (m2d_topk_refine): 0 = off (A/B)
    int64_t topk_refined = 0;           // diagnostics of the last call (device counters, read on request)
    float *topk_ex = nullptr;           // what the lists leave out, per (user, dish range) / (user, group) / user: 4 floats each
    size_t topk_ex_cap = 0;             // floats
    int32_t *topk_refine_counter = nullptr;   // [2] users refined, users sent on to the repair (inside topk_ex's allocation)
    int opt_topk_block = 0;             // users per block of a pruned split-bf16 launch: 0 = the launcher's choice, 128 / 256 forced (A/B)
    int topk_block_users = 256;         // what the last pattern-grouped launch used (the tile counters count tiles of blocks this size)
    int opt_topk_prune = 1;             // pipelined form: blocks step through the tiles of their users' relevant mask patterns only (0 = every tile)

    std::string last_error;
    const char *last_kernel = "";
};

#define M2D_HIP_TRY(h, expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (h)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);               \
            return M2D_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size: a retrieval call makes six such calls, each a
// trip into the runtime, for a value that does not change (8 us of a 50 us host-side call)
// (The attribute belongs to the function ON A DEVICE, and a process may hold engines on several -- m2d_create(device): the
// cache is keyed by the current device as well; every entry point has already made the engine's device current.)
static inline hipError_t m2d_lds_limit(const void *fn, int bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> seen;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(dev, fn);
    auto it = seen.find(key);
    if (it != seen.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) seen[key] = bytes;
    return e;
}

// a * x + b * y as TF's Mul, Mul, Add compute Model_Recommender.py:95-96: two rounded products, one rounded sum.
// __fadd_rn(__fmul_rn(a, x), __fmul_rn(b, y)) does not guarantee that -- the device library's bodies carry the `contract` flag,
// and hipcc 7.2 fused the second product into the add (v_fmac, one rounding fewer) in SOME copies of an unrolled loop: the same
// (user, dish) then scored one ulp apart depending on the slot it was computed in (m2d_catalogue.hip, repair scan).  Plain
// operators under `fp contract(off)` carry no such flag, also after inlining.
__device__ __forceinline__ float m2d_blend_unfused(const float a, const float x, const float b, const float y)
{
#pragma clang fp contract(off)
    const float p = a * x;
    const float q = b * y;
    return p + q;
}

// ---- device helpers shared by the LDS-DMA kernels (m2d_catalogue.hip, m2d_mlp.hip) ----
// vmcnt(0) twice over: the builtin is an s_waitcnt the compiler's own counter model sees (so it stops assuming that
// loads from a previous loop trip are still in flight), the asm one cannot be optimised away on the grounds that
// the compiler knows of nothing outstanding (the DMA ops below are hidden from it).
static __device__ __forceinline__ void wait_all_vmem()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS-DMA of 64 x 16 B: lane l's 16 bytes at `src` land at lds_base + 16 l.  Written as asm on purpose: after the
// builtin the compiler puts s_waitcnt vmcnt(0) in front of the next ds_read (it must assume the read aliases the
// DMA'd bytes), which serialises the NEXT stage's fill with the CURRENT stage's multiply.  Here a stage is
// published by an explicit vmcnt(0) + barrier before anyone reads it, so that wait is never needed.  The hidden
// VMEM op only makes the compiler's own vmcnt(N) waits more conservative (returns are in order), never less.
static __device__ __forceinline__ void lds_dma16(const void *src, const void *lds_base_uniform)
{
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)lds_base_uniform);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m0v) : "memory", "m0");
}

// the same through the compiler's builtin (the inline-asm form takes its 64-bit address in any VGPR pair; where the
// allocator picks an odd one hipcc 7.2 stops with "Subtarget requires even aligned vector registers")
static __device__ __forceinline__ void lds_dma16_b(const void *src, void *lds_base_uniform)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_base_uniform, 16, 0, 0);
}

// launchers (m2d_score.hip / m2d_topk.hip); all enqueue on `stream` and return a status
int m2d_launch_score_pairs(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                           bool by_dish, int64_t B, float *out, hipStream_t stream,
                           bool use_ingredients = false);
int m2d_ensure_finite_scan(m2d_engine *h, hipStream_t stream);   // m2d_abi.hip: runs the queued table scan, if any
int m2d_launch_rows_finite_check(m2d_engine *h, const int32_t *users, int64_t B, hipStream_t stream);   // Personal_Memory rows of a batch
int m2d_launch_build_dish_high(m2d_engine *h, hipStream_t stream);
int m2d_ensure_user_high(m2d_engine *h, hipStream_t stream);
int m2d_launch_check_csr(m2d_engine *h, hipStream_t stream);
int m2d_ensure_dish_vectors(m2d_engine *h, hipStream_t stream);
int m2d_launch_write_memory(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                            const float *sign, const float *labels, int64_t B, int32_t L, float *gm, float beta_1,
                            float beta_2, float alpha, int32_t which, double *out_sums, hipStream_t stream);
int m2d_train_setup(m2d_engine *h, int32_t learner, float lr, float clip_norm, hipStream_t stream);
void m2d_train_release(m2d_engine *h);
int m2d_launch_train_step(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, const float *labels,
                          int64_t B, int32_t apply, float *out, hipStream_t stream);
int m2d_train_get_slot(m2d_engine *h, int32_t table, int32_t slot, float **dev, int64_t *count);
int m2d_train_step_count(m2d_engine *h, int64_t *steps, int32_t set);
int m2d_launch_score_pairs_mlp(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B, float *out,
                               hipStream_t stream);
int m2d_launch_rank_candidates(m2d_engine *h, const int32_t *users, const int32_t *items,
                               const int32_t *lens, int64_t nseg, int32_t L, int32_t k, float *out_scores,
                               int32_t *out_items, int32_t *out_flags, hipStream_t stream);
int m2d_launch_topk_users(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores,
                          int32_t *out_ids, hipStream_t stream);
