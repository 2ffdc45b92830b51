/* m2d.h -- C ABI of the MI355X (gfx950) Market2Dish scoring engine (libm2d.so).
 *
 * The reference (WenjieWWJ/FoodRec) has no FFI: its boundary is a Python class plus TF1 session
 * fetches.  Each entry point below names the reference interface it stands in for (paths relative
 * to the reference root, Code/Recommender/...).  The Python layer (foodrec_amd/) is the only
 * caller; it registers these as torch.library custom ops (foodrec_amd/ops.py).
 *
 * Conventions
 *   - every function returns an int status: M2D_OK (0) or a negative M2D_ERR_* code; nothing throws;
 *   - data pointers are DEVICE pointers (HBM) unless a parameter says "host"; the caller owns every
 *     buffer; the engine owns only what it allocates itself (tables passed with M2D_TABLES_HOST);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are enqueued and
 *     the call returns without synchronising.  Id errors found by a kernel are latched on the
 *     device and reported by m2d_check(), which does synchronise;
 *   - an engine is not thread-safe; distinct engines / streams are independent;
 *   - all arithmetic is float32; ids are int32 (Model_Recommender.py:26-32).
 */
#ifndef M2D_H_
#define M2D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct m2d_engine m2d_engine;

#define M2D_OK 0
#define M2D_ERR_INVALID_ARG (-1)    /* null pointer, non-positive size, k out of range ...          */
#define M2D_ERR_HIP (-2)            /* a HIP runtime call failed; text in m2d_last_error()           */
#define M2D_ERR_BAD_USER_ID (-3)    /* a user id outside [user_base, user_base + U)                  */
#define M2D_ERR_BAD_ITEM_ID (-4)    /* a dish id outside [0, I)                                      */
#define M2D_ERR_NOT_CONFIGURED (-5) /* call needs m2d_set_dish_categories / m2d_set_* first          */
#define M2D_ERR_UNSUPPORTED (-6)    /* shape outside what the kernels cover (message says which)     */
#define M2D_ERR_NO_DEVICE (-7)      /* no HIP device visible: there is no CPU fallback, by design    */
#define M2D_ERR_BAD_INGREDIENT (-8) /* extension: ingredient id outside [0, R) or a malformed CSR     */
#define M2D_ERR_KERNEL_TIMEOUT (-9) /* m2d_topk_users: a wave gave up waiting for its workgroup (about 1 s); that call's
                                       lists are INVALID.  Latched on the device, reported by m2d_check */

#define M2D_TABLES_HOST 0   /* table pointers are host memory: copied to HBM once, engine-owned      */
#define M2D_TABLES_DEVICE 1 /* table pointers are device memory: borrowed, caller keeps them alive   */

/* ABI version: bumped on any signature change. */
int m2d_abi_version(void);

/* Model build.  Replaces Model.__init__ + instantiate_weights (Model_Recommender.py:5-37, :43-54):
 *   pm [U, C+1, E]  Personal_Memory  (row 0 high-level, rows 1..C low-level per category)
 *   re [I, E]       Recipe_Embedding
 *   ce [C, E]       Category_Embedding
 *   coef            high_level_score_coefficient, held as float32; (1 - coef) is taken in float32
 *                   (Model_Recommender.py:17, :96)
 * General_Memory is not an argument: the forward never reads it (Model_Recommender.py:56-97). */
int m2d_create(const float *pm, const float *re, const float *ce, int64_t U, int64_t I, int32_t C,
               int32_t E, float coef, int device, int table_flags, m2d_engine **out);
int m2d_destroy(m2d_engine *h);

/* Text of the last failure on this engine (or of the last failed m2d_create when h == NULL). */
const char *m2d_last_error(const m2d_engine *h);

/* User-axis sharding (no reference counterpart): this engine holds users
 * [user_base, user_base + U); ids in every call stay GLOBAL. */
int m2d_set_user_base(m2d_engine *h, int64_t user_base);

/* Resident per-dish category masks [I, C] -- the device-side form of dish_to_category.json
 * (Train_recommender.py:132, evaluate.py:43,50).  Needed by the *_bydish / rank / topk calls. */
int m2d_set_dish_categories(m2d_engine *h, const float *cats, int table_flags);

/* Predict.  Replaces sess.run([model.logits], feed_dict) (evaluate.py:55-59) = Model.inference
 * (Model_Recommender.py:56-97) for B pairs:
 *   users i32[B], items i32[B], cats f32[B, C] (the [B, C, 1] placeholder flattened; any float
 *   weight is accepted, a row summing to 0 yields NaN as 0/0 does at :79/:92), out f32[B]. */
int m2d_score_pairs(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                    int64_t B, float *out, void *stream);

/* m2d_score_pairs for HOST buffers -- what the reference's call site actually hands over (lists / numpy, 51 pairs
 * per call, evaluate.py:39-59).  users, items, cats and out are host pointers; the call puts them in one pinned
 * block, WAITS for the work it enqueued on `stream` and returns the scores in `out`.  Up to 65 536 pairs the kernel
 * reads and writes that block over the host link and the call spins on a completion word behind the scores before it
 * falls back to a stream wait; larger feeds (or option "host_zero_copy" = 0) take one copy in and one copy out
 * (1 = pinned block without the spin); feeds of more than 262 144 pairs go in chunks of that size through two
 * pinned blocks, the host's copy of one chunk overlapping the transfer and kernel of the one before (0.66 -> 1.0 G
 * pairs/s at 4 M pairs).  Same kernels, same bits either way.  An out-of-range
 * id is returned directly as M2D_ERR_BAD_USER_ID / M2D_ERR_BAD_ITEM_ID (text in m2d_last_error, position counted in
 * the whole feed), `out` untouched -- or, for a chunked feed, holding the scores of the chunks before the offending one.
 * This is the latency path; its rate includes PCIe and is never what bench.py reports as `value`. */
int m2d_score_pairs_host(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                         int64_t B, float *out, void *stream);

/* Same, with cats looked up from the resident dish table: cats[b] = dish_categories[items[b]]. */
int m2d_score_pairs_bydish(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B,
                           float *out, void *stream);

/* Evaluator inner step.  Replaces eval_one_rating's scoring + ranking (evaluate.py:35-63) for
 * nseg users in ONE launch: segment s scores users[s] against items[s*L .. s*L + lens[s]) and
 * returns the first k of heapq.nlargest order -- a repeated item keeps its first position and its
 * last score (dict collapse, :60-61), ties go to the earlier position (:63).
 *   lens may be NULL (every segment has L candidates); 1 <= L <= 1024; 1 <= k <= 64
 *   out_items i32[nseg, k] (-1 past the number of distinct candidates), out_scores f32[nseg, k],
 *   out_flags i32[nseg]: bit 0 set when a candidate score is NaN -- the caller must then rank that
 *   segment on the host, where the reference's NaN behaviour can be reproduced exactly. */
int m2d_rank_candidates(m2d_engine *h, const int32_t *users, const int32_t *items, const int32_t *lens,
                        int64_t nseg, int32_t L, int32_t k, float *out_scores, int32_t *out_items,
                        int32_t *out_flags, void *stream);

/* Full-catalogue retrieval (build-defined generalisation of evaluate.py:39-63 to every dish; BASELINE
 * config 5): for each of nU users the k best dishes over all I, descending score, ties to the lower
 * dish id, NaN scores last.  out_scores f32[nU, k], out_ids i32[nU, k].  1 <= k <= 64.
 * Scores agree with m2d_score_pairs_bydish within the 1e-4 bar, not bit for bit (factored form; see the
 * "topk_bf16x3" option below).
 * Kernels: 0/1 masks, C = 4, k <= 16 and E a multiple of 4 up to 256 run the pattern-grouped MFMA kernels (E = 64 / 128
 * on split bf16 by default; other sizes, e.g. the reference's embed_size 200, exact f32 on dish rows zero-padded to
 * 32 / 64 / 128 / 256 floats); other masks or k run the dense MFMA kernel where (C + 1) E / 8 is 20, 40 or 80, and a
 * one-block-per-user kernel otherwise.
 * With the ingredient table set (m2d_set_ingredients) the high-level term uses H[d]; E = 32 / 64 stay on the
 * pattern-grouped split-bf16 kernel (rows [H[d] | RE[d]]), other shapes use the dense kernel.
 * Tie rule: bit-equal scores resolve to the lower dish id, whatever kernel runs -- heapq.nlargest's rule
 * (evaluate.py:63).  The pattern-grouped kernels scan the dishes grouped by mask pattern and, inside a pattern, by
 * descending row norm; they notice when a tie decides what a list holds (a score equal to a list's last entry falls off
 * or is refused), and such users -- an all-zero Personal_Memory block, a user vector that scores whole groups
 * identically -- are re-ranked over the catalogue in id order with the formula in plain f32 (their scores are then
 * exact-f32 even under "topk_bf16x3" = 1).  Option "topk_grouped" = 0 selects the dense kernel (about 5x the matrix
 * work), which scans in id order.
 * coef = 1 (Train_recommender.py:61-62; `1 - coef` is then an exact float32 zero, Model_Recommender.py:96): every dish of a
 * mask pattern scores alpha_P[u], so the pattern-grouped path reads each list off the patterns' lowest dish ids, best
 * alpha_P first (m2d_topk_high_level_only: no scan, no repair; what heapq.nlargest returns for whole groups of equal scores). */
int m2d_topk_users(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores,
                   int32_t *out_ids, void *stream);

/* Memory write (training side).  Replaces Model.Write_Memory
 * (Model_Recommender.py:106-220) -- the `personal` / `general` fetches of Train_recommender.py:180-199 --
 * with an O(B (C+1) E) atomic scatter-add in place of the reference's dense one-hot matmuls:
 *     PM[u_b] += v_b + alpha * g_b ;   GM[l] += sum_b y_bl v_b          (v_b, g_b: see csrc/m2d_write.hip)
 *   users/items i32[B], cats f32[B, C], write_sign f32[B] (the [B, 1] placeholder flattened),
 *   labels f32[B, L] (user_one_hot_label), general_memory f32[L, C+1, E] (device, updated in place).
 * Personal_Memory is updated IN PLACE in the buffer given to m2d_create, which must therefore be
 * writable device memory (M2D_TABLES_DEVICE) or engine-owned (M2D_TABLES_HOST).
 * `which` selects the assigns that run, as the fetch list does in the reference graph: `personal` depends on the two
 * chained Personal_Memory assigns only (:167, :198), `general` on the General_Memory assign only (:215) -- the
 * driver's ordinary batch fetches `general` alone (Train_recommender.py:195-199), so it never writes Personal_Memory.
 *   M2D_WRITE_PERSONAL  PM[u_b] += v_b + alpha * g_b      (g_b reads General_Memory as it stands before this call)
 *   M2D_WRITE_GENERAL   GM[l]   += sum_b y_bl v_b
 * out_sums (device f64[2], may be NULL) receives sum(PM) in [0] when M2D_WRITE_PERSONAL is set and sum(GM) in [1]
 * when M2D_WRITE_GENERAL is set, after the write (the reference returns their means); the other entry is 0. */
#define M2D_WRITE_PERSONAL 1
#define M2D_WRITE_GENERAL 2
int m2d_write_memory(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                     const float *write_sign, const float *labels, int64_t B, int32_t L, float *general_memory,
                     float beta_1, float beta_2, float alpha, int32_t which, double *out_sums, void *stream);

/* Training step.  Replaces Model.loss + Model.train (Model_Recommender.py:99-104,
 * :223-241) as the driver runs them: sess.run([model.loss_value, model.learning_rate, ..., model.train_op], feed_dict)
 * (Train_recommender.py:180-199).  All three tables given to m2d_create are updated IN PLACE and must be writable
 * device memory (M2D_TABLES_DEVICE) or engine-owned; General_Memory gets no gradient and is not involved.
 *   m2d_train_begin  `learner` as the reference's --learner flag (:228-235: adagrad / rmsprop / adam, anything else
 *                    is gradient descent), `lr` = args.lr (the decayed rate equals it for ever: apply_gradients is
 *                    called without global_step, :240), `clip_norm` = 5.0 in the reference (:237).  Allocates the
 *                    optimizer slots (TF 1.x initial values) and scratch; calling it again resets the optimizer.
 *   m2d_train_step   users/items i32[B], cats f32[B, C], labels f32[B].  apply = 1: loss, gradients, global-norm
 *                    clip and the update; apply = 0: loss and gradient norm only (the loss_value fetch alone).
 *                    out (device f32[4], may be NULL) receives {loss, global gradient norm, clip scale, lr}.
 *                    An out-of-range id blocks the whole update (TF raises from the gather before any assign): the
 *                    tables, the slots, the step count and Adam's beta powers all stay as they were.  The id error
 *                    stays latched until m2d_check reports it, and while it is latched every later step is blocked
 *                    too -- call m2d_check after a step whose ids are not known to be valid.
 *   m2d_train_slot   copies optimizer slot `slot` of table 0 = Personal_Memory, 1 = Recipe_Embedding,
 *                    2 = Category_Embedding (adam: m, v; adagrad: accumulator; rmsprop: rms, momentum) into `buf`
 *                    (restore = 0) or from it (restore = 1); `buf` is device memory shaped like the table.  For
 *                    checkpoint / resume (the reference's tf.train.Saver covers the slots too).
 *                    M2D_ERR_INVALID_ARG when the learner has no such slot.
 * The update rules restate TF 1.x's published behaviour (oracle/train_oracle.py); parity is unpinned. */
#define M2D_LEARNER_SGD 0
#define M2D_LEARNER_ADAGRAD 1
#define M2D_LEARNER_RMSPROP 2
#define M2D_LEARNER_ADAM 3
int m2d_train_begin(m2d_engine *h, int32_t learner, float lr, float clip_norm, void *stream);
int m2d_train_step(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, const float *labels,
                   int64_t B, int32_t apply, float *out, void *stream);
int m2d_train_slot(m2d_engine *h, int32_t table, int32_t slot, float *buf, int32_t restore, void *stream);
/* Tell the engine that borrowed table memory (M2D_TABLES_DEVICE) was written from outside -- a checkpoint restore,
 * an optimizer of the caller's own: everything the engine derives from Recipe_Embedding / Category_Embedding
 * (retrieval's dish vectors and pattern-grouped tables) is rebuilt on next use.  m2d_train_step and
 * m2d_write_memory do this themselves. */
int m2d_tables_updated(m2d_engine *h);

/* Optimizer steps applied since m2d_train_begin: read (*steps receives it, set = 0) or restored (set = 1, for a
 * checkpoint resume; Adam's bias-correction powers are re-derived from it). */
int m2d_train_steps(m2d_engine *h, int64_t *steps, int32_t set);
int m2d_train_end(m2d_engine *h);

/* ---- build-defined extension, NO reference counterpart (BASELINE.json configs 2-5) ----
 * Multi-hot ingredient table for the high-level path:
 *     H[d]  = sum_j w_j ING[id_j] / sum_j w_j     over dish d's list ids[off[d] .. off[d+1])
 *     high  = <U_high[u], H[d]>                    in place of Model_Recommender.py:67-79
 *     score = a*high + (1-a)*low                   low-level path and blend unchanged (:82-96)
 * With ING = Category_Embedding, ids = 0..C-1 and w = the dish's category mask this is the reference's
 * formula.  ing f32[R, E]; off i32[I+1] (CSR, off[0] = 0); ids i32[nnz]; w f32[nnz] or NULL (all 1).
 * The call gathers and segment-sums the rows into H once (it synchronises and reports a malformed CSR /
 * bad id as M2D_ERR_BAD_INGREDIENT; an id error still latched from an earlier launch is returned first, under its own
 * code, exactly as m2d_check would); all four arrays use `table_flags`.  A dish with an empty list
 * scores NaN, like an empty category mask. */
int m2d_set_ingredients(m2d_engine *h, const float *ing, int64_t R, const int32_t *off, const int32_t *ids,
                        const float *w, int64_t nnz, int table_flags);
int m2d_clear_ingredients(m2d_engine *h);

/* m2d_score_pairs with the ingredient high-level path.  cats f32[B, C] feeds the low-level path;
 * cats == NULL uses the resident dish masks (m2d_set_dish_categories). */
int m2d_score_pairs_ingredients(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                                int64_t B, float *out, void *stream);

/* ---- build-defined extension, NO reference counterpart (BASELINE.json configs[2]) ----
 * 3-layer scoring head on the interaction vector z[k] = flatten(PM[u])[k] * Dt[d][k], k < K = (C+1)*E
 * (sum_k z[k] is the reference score, Model_Recommender.py:67-96 in factored form):
 *     score = sum_k z[k] + w3 . relu(W2^T relu(W1^T z + b1) + b2) + b3
 * W1 f32[K, H1], b1 f32[H1], W2 f32[H1, H2], b2 f32[H2], w3 f32[H2].  H1 = 256, H2 = 64 with K % 64 == 0
 * run on the MFMA kernels (split-bf16 layers 1-2 by default, exact f32 with option "mlp_bf16x3" = 0), and so do
 * K % 4 == 0, K <= 1280 on a W1 copy zero-padded to whole 64-row chunks (the reference's embed_size 200 gives
 * K = 1000); other sizes run a generic kernel.  Dish masks must be resident
 * (m2d_set_dish_categories); the ingredient table, when set, feeds Dt's high-level part. */
int m2d_set_mlp_head(m2d_engine *h, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *w3, float b3, int32_t H1, int32_t H2, int table_flags);
int m2d_clear_mlp_head(m2d_engine *h);
int m2d_score_pairs_mlp(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B, float *out,
                        void *stream);

/* Synchronise `stream` and report (then clear) the first id error latched by kernels since the
 * previous check: M2D_OK, M2D_ERR_BAD_USER_ID or M2D_ERR_BAD_ITEM_ID.  TF-CPU GatherV2 raises
 * InvalidArgument for such ids; the kernels never read out of bounds and write NaN for the pair.
 * bad_value / bad_index (host pointers, may be NULL) receive the offending id and its position. */
int m2d_check(m2d_engine *h, void *stream, int64_t *bad_value, int64_t *bad_index);

/* Options.  m2d_score_pairs* -- the reference path (Model_Recommender.py:56-97) -- is exact float32 whatever is set here.
 * Unknown names: M2D_ERR_INVALID_ARG.  "Results" = the scores / ids a call returns.
 *
 * ---- product switches -------------------------------------------------------------------------------------------------------------
 * name             default  values  changes results?                          what it does
 * skip_masked      1        0 / 1   no while every table value is finite      Pair kernels do not fetch the Personal_Memory row of a category whose
 *                                   (same bits); with inf / NaN in a table    mask weight is exactly 0 -- the graph multiplies it by 0 (:82) -- so a pair
 *                                   the engine ignores it and multiplies      moves (2 + active categories) E 4 B of rows, not (C + 2) E 4 B; a device word
 *                                   everything (the graph's NaNs)             "a table value is not finite" (table scan queued by m2d_create /
 *                                                                             m2d_tables_updated, the engine's writers on what they write) switches it
 *                                                                             off.  m2d_score_pairs_mlp: pairs grouped by the dish's pattern of non-zero
 *                                                                             weights, k-blocks a pattern lacks not multiplied.  0 = literal fetch-and-multiply.
 * user_high_table  0        0 / 1   scores within 1e-6 (another summation     Calls of >= 2^18 pairs read sum_c m_c <U_high[u], CE_c> / n from a derived
 *                                   order)                                    [U, 4] table (16 B per user; rebuilt after the engine's writers or
 *                                                                             m2d_tables_updated) instead of the U_high row.  A serving option.
 * host_zero_copy   2        0 1 2   no                                        m2d_score_pairs_host up to 65 536 pairs: 2 = the kernel works on the pinned
 *                                                                             block and the call spins on a completion word, 1 = without the spin, 0 = staged copies.
 * topk_bf16x3      1        0 / 1   scores within ~1e-5 relative; dish ids    m2d_topk_users, 0/1 masks, E = 64 / 128: contraction on split-bf16 MFMA
 *                                   identical while topk_refine = 1           (x = hi + lo, three bf16 products, f32 accumulation).  0 = exact-f32 MFMA.
 * topk_refine      1        0 / 1   ids of near-tied lists (E = 32 / 64 /     Pattern-grouped kernels finish lists whose neighbouring scores -- or last
 *                                   128; k <= 10 at E = 128 split bf16)       entry and best score left out -- lie within twice the kernel's rounding
 *                                                                             margin in ONE plain-f32 arithmetic, so split-bf16 and exact-f32 return the
 *                                                                             same ids (a few per cent of the users re-scored; no measurable cost).
 *                                                                             0 = lists as the scan leaves them, ties at a list's end repaired.
 * topk_grouped     1        0 / 1   scores within the 1e-4 bar; both forms    0/1 masks, C = 4, k <= 16: dishes sorted by mask pattern, contraction over E
 *                                   resolve ties to the lower dish id         (m2d_topk_users above).  0 = the dense kernel over (C + 1) E (about 5x the work).
 * mlp_bf16x3       1        0 / 1   scores within ~1e-5 relative              Layers 1-2 of m2d_score_pairs_mlp on split-bf16 MFMA.  0 = exact-f32 MFMA.
 *
 * ---- diagnostic / A-B values: kernel selection for benchmarking and tests; results never depend on them (same bits) unless noted ------
 * name             default  values
 * prefetch         2        1 / 2 / 4    row-load steps issued ahead in the pair kernels
 * nt_loads         1        0 / 1        non-temporal loads of Personal_Memory rows
 * blocks_per_cu    8        1 ... 16     grid size of the pair kernels
 * mlp_form         0        0 / 1        split-bf16 head kernel: 0 = matrix waves fed by gather / DMA waves, 1 = every wave gathers its own rows
 *                                        (same results within the split's rounding)
 * topk_form        0        0 ... 4      split-bf16 retrieval kernel: 0 / 2 = pipelined form, 1 = first form (same lists).  Large catalogues (more than
 *                                        24 576 tiles of 32 dishes at E = 64, more than 10 240 at E = 128; with the ingredient table, where every tile
 *                                        is multiplied, more than 256): the pipelined form multiplies the hi x hi
 *                                        product of every tile and the two cross products only for tiles that can still hold a candidate -- scores
 *                                        are within the split's rounding of the three-product kernels' but not their bits, which is why the rule looks
 *                                        at the catalogue alone (blocks of 256 users; "topk_block" = 128 is not honoured there).  3 = that form for
 *                                        any catalogue, 4 = never (both diagnostic)
 * topk_prune       1        0 1 2 4 5 7 9  pattern-grouped retrieval: 1 = scan starts from a lower bound of the user's k-th score and steps through the
 *                                        tiles of the mask patterns that can reach its top-k only (bounds: alpha_P[u] +- |w_P[u]| max|RE[d]|, widened by
 *                                        what the arithmetic can move a computed score by; users sorted by pattern mask; (user block, dish range) items
 *                                        longest first); 0 = every tile.  A/B forms the tests compare with it: 2 the bound only, 4 the patterns only, 5 the
 *                                        grid's launch order, 7 a user's dish ranges keep their thresholds apart (by default they meet in one atomic-max
 *                                        word per user, E = 64), 9 the tie repair reads every pattern's dishes.  Same lists, bit for bit.
 * topk_block       0        0 128 256    users per block of a pruned pipelined launch (0 = the launcher's choice)
 * variant          0        7 9 11 12 13 14 15 16, 100 + n
 *                                        7 / 9: retrieval on the dense MFMA kernel / on the one-block-per-user kernel; 9 also forces the generic pair
 *                                        and head kernels; 11 / 12: the pair kernel's throughput / latency form whatever the batch size; 13: tie
 *                                        repair's one-block-per-user tier from the third listed user on; 14: the nine-launch training step;
 *                                        15: the retrieval scan's progress-word wait gives up at once (M2D_ERR_KERNEL_TIMEOUT: test hook);
 *                                        16: the MLP head's row offsets in 16-byte units (the instantiation tables of 4 GiB and more take);
 *                                        100 + n: n dish-range splits in retrieval
 * (round 6 removed the values no test or profile script used: topk_prune 3 / 6, topk_probes, variant 8)
 *
 * ---- read-only diagnostics of the last m2d_topk_users call (m2d_get_option; they synchronise the device) ---------------------------------
 * topk_repaired          users the tie repair re-ranked over their patterns
 * topk_refined           users whose near-tied list m2d_topk_refine finished;  topk_refine_repaired: of those, sent on to the tie repair
 * topk_tiles_scanned     32-dish tiles the blocks stepped through;  topk_tiles_full: what they would have without pruning
 * topk_tiles_completed   the hi x hi first form (see topk_form): (wave, tile) pairs whose cross products were multiplied; -1 if another form ran
 * topk_block_users       users per block of that launch (128 or 256);  num_cu: compute units of the device
 */
int m2d_set_option(m2d_engine *h, const char *name, int64_t value);
int m2d_get_option(const m2d_engine *h, const char *name, int64_t *value);

/* Calibration probe, not part of the scoring path: a plain 16-B-per-lane streaming read of `bytes`
 * bytes (multiple of 16) that folds everything into sink[0]; bench.py times it to report the
 * achievable HBM read rate of the box beside the 8 TB/s spec peak. */
int m2d_stream_read_probe(m2d_engine *h, const void *buf, int64_t bytes, float *sink, void *stream);

/* Name of the kernel the last m2d_score_pairs* call launched (for matching rocprofv3 traces). */
const char *m2d_last_kernel(const m2d_engine *h);

#ifdef __cplusplus
}
#endif
#endif /* M2D_H_ */
