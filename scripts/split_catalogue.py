#!/usr/bin/env python3
"""One-off (round 5): cut foodrec_amd/csrc/m2d_catalogue.hip into translation units over one shared header.
Kept for the record of what moved where; the kernels' text is taken line for line from the old file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import subprocess
# the file as round 4 left it
L = subprocess.check_output(["git", "-C", ROOT, "show", "6ff440a:foodrec_amd/csrc/m2d_catalogue.hip"]).decode().split("\n")


def cut(a, b):
    """lines a..b of the old file, 1-based inclusive"""
    return "\n".join(L[a - 1:b]) + "\n"


CONTRACT = cut(25, 31)
DIAG = cut(33, 44)

HEADER = '''// Full-catalogue retrieval (m2d_topk_users) for gfx950: what its translation units share.
//
//   m2d_catalogue_dense.hip       dish vectors Dt[d]; m2d_topk_mfma (dense [users x (C+1)E] . [(C+1)E x dishes], exact f32: weighted
//                            masks, k > 16, the ingredient table beyond E = 64), m2d_topk_generic (any shape, one block per user)
//   m2d_catalogue_plan.hip        0/1 masks: the pattern-sorted dish table, the call's plan (per-user bounds, relevant patterns, the sort,
//                            the launch order), the launcher of a pattern-grouped call, m2d_launch_topk_users' dispatch
//   m2d_catalogue_scan_f32.hip    m2d_topk_grouped: the pattern-grouped scan on v_mfma_f32_32x32x2_f32 (exact f32; zero-padded widths)
//   m2d_catalogue_scan_bf16.hip   m2d_topk_grouped_bf16 / _bf16_pipe2: the same scan on split-bf16 MFMA (the default, E = 64 / 128)
//   m2d_catalogue_merge.hip       dish ranges' partial lists -> a user's list; near-tied lists finished in plain f32 (m2d_topk_refine)
//   m2d_catalogue_repair.hip      users whose k-th score is tied three ways or more: re-ranked over their patterns in id order
//
// Reference behaviour all of it reproduces: score = Model_Recommender.py:67-96 per (user, dish), ranking = heapq.nlargest
// (evaluate.py:63: score descending, ties to the lower dish id, NaN last).
#pragma once

#include <math.h>

#include <type_traits>

#include "m2d_engine.h"

''' + CONTRACT + '''
''' + DIAG + '''
#define M2D_INTERNAL __attribute__((visibility("hidden")))

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// ---- argument blocks that cross translation units (global namespace: one type for every unit) ------------------------------
''' + cut(1560, 1588) + '''
''' + cut(747, 765) + '''
''' + cut(1164, 1173) + '''
// a pattern-grouped scan launch: which instantiation (m2d_catalogue_scan_f32.hip / m2d_catalogue_scan_bf16.hip)
struct ScanShape {
    int E;           // kernel width: 32 / 64 / 128 / 256 floats per dish row (with the ingredient table: [H[d] | RE[d]], twice the embedding)
    int KR;          // list slots per lane: 10 or 16
    bool bf16x3;     // split-bf16 MFMA (E = 64 / 128); else exact f32
    bool hv;         // ingredient rows (pipelined split-bf16 kernel only)
    bool pad;        // dish rows zero-padded to E floats (exact f32 only)
    bool pipe;       // split bf16: the pipelined form (else the first form)
    int waves;       // waves per block: 8 (256 users) or 4 (128 users; pipelined split bf16, E = 64)
    bool keep;       // the lists' left-out scores are kept for m2d_topk_refine (GroupedArgs::ex_out)
};

// ---- launchers other units call (all enqueue on `st`; int results are M2D_* codes) ---------------------------------------------
M2D_INTERNAL int m2d_topk_dense_launch(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores, int32_t *out_ids,
                                       hipStream_t st);
M2D_INTERNAL int m2d_topk_scan_f32_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);
M2D_INTERNAL int m2d_topk_scan_bf16_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st);
M2D_INTERNAL void m2d_launch_merge_splits(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *out_s, int32_t *out_i,
                                          hipStream_t st, const float *tie_in = nullptr, float *tie_out = nullptr, int32_t *tie_list = nullptr,
                                          int64_t I = 0, const float *ex_in = nullptr, float *ex_out = nullptr, const float *plan = nullptr,
                                          int32_t *rcount = nullptr);
M2D_INTERNAL void m2d_launch_merge_splits2(const float *ps, const int32_t *pi, int64_t nU, int nsplit, int k, float *tmp_s, int32_t *tmp_i,
                                           float *out_s, int32_t *out_i, hipStream_t st, float *tie, float *tie_final, int32_t *tie_list,
                                           int64_t I, float *ex = nullptr, float *ex_final = nullptr, const float *plan = nullptr,
                                           int32_t *rcount = nullptr);
M2D_INTERNAL void m2d_topk_launch_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I, hipStream_t st);
M2D_INTERNAL void m2d_topk_launch_tie_compact(const float *tie_final, int64_t nU, int32_t *tie_list, float *scores, int32_t *ids, int k,
                                              int64_t I, int refined, hipStream_t st);
M2D_INTERNAL void m2d_topk_launch_refine(const RefineArgs &f, bool flag_pass, hipStream_t st);
M2D_INTERNAL int m2d_topk_launch_repair(m2d_engine *h, const RepairArgs &r, bool hv, hipStream_t st);

namespace {

''' + cut(96, 100) + '''
''' + cut(102, 176) + '''
''' + cut(178, 196) + '''
''' + cut(198, 234) + '''
''' + cut(550, 573) + '''
// The tie repair's scratch: REPAIR_SPLITS partial lists for each of up to REPAIR_CAP listed users (m2d_catalogue_repair.hip)
constexpr int REPAIR_SPLITS = 64, REPAIR_CAP = 1024;

''' + cut(767, 774) + '''
''' + cut(776, 800) + '''
''' + cut(1358, 1376) + '''
''' + cut(1590, 1596) + '''
''' + cut(1598, 1729) + '''
''' + cut(2030, 2101) + '''
''' + cut(2103, 2106) + '''
}  // namespace
'''

DENSE = cut(1, 18) + '''#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(51, 77) + '''
''' + cut(79, 94) + '''
''' + cut(236, 472) + '''
''' + cut(474, 542) + '''
''' + cut(3639, 3690) + '''
}  // namespace

''' + cut(1338, 1354) + '''
// Dispatch of the kernels of this file: weighted masks, k > 16, category counts other than 4, embedding sizes without a
// pattern-grouped kernel -- and 0/1 masks under option "topk_grouped" = 0.
int m2d_topk_dense_launch(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores, int32_t *out_ids,
                          hipStream_t stream)
{
    int rc = m2d_ensure_dish_vectors(h, stream);
''' + cut(3735, 3762)

PLAN = '''// Pattern-grouped retrieval, host side and planning kernels: the pattern-sorted dish table (built once per mask table), the plan of
// a call (per-user bounds and relevant patterns, users sorted by pattern mask, (user block, dish range) items longest first), the
// launcher that strings a call's kernels together, and m2d_launch_topk_users' choice between this path and the dense kernels.
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(1378, 1558) + '''
''' + cut(1731, 1887) + '''
''' + cut(1889, 2028) + '''
''' + cut(3245, 3252) + '''
''' + cut(3254, 3319) + '''
''' + cut(3356, 3384) + '''
''' + cut(3386, 3637) + '''
}  // namespace

''' + cut(3694, 3733) + '''    return m2d_topk_dense_launch(h, users, nU, k, out_scores, out_ids, stream);
}
'''

SCAN_F32 = '''// Pattern-grouped scan on exact-f32 MFMA (v_mfma_f32_32x32x2_f32): 0/1 masks, contraction over E instead of (C+1) E.
// Serves "topk_bf16x3" = 0, E = 32, and every embedding size without a kernel of its own (rows zero-padded to 32 / 64 / 128 / 256
// floats -- the reference's default embed_size 200, Train_recommender.py:51-58, among them).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(2108, 2328) + '''
}  // namespace

int m2d_topk_scan_f32_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st)
{
#define M2D_SCAN_F32(EV, KRV, PADV)                                                     \\
    if (s.E == EV && s.KR == KRV && s.pad == PADV) {                                    \\
        auto kern = m2d_topk_grouped<EV / 8, 8, KRV, PADV>;                             \\
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));                    \\
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, a);                          \\
        return M2D_OK;                                                                  \\
    }
    M2D_SCAN_F32(32, 10, false) M2D_SCAN_F32(32, 16, false) M2D_SCAN_F32(64, 10, false) M2D_SCAN_F32(64, 16, false)
    M2D_SCAN_F32(128, 10, false) M2D_SCAN_F32(128, 16, false)
    M2D_SCAN_F32(32, 10, true) M2D_SCAN_F32(32, 16, true) M2D_SCAN_F32(64, 10, true) M2D_SCAN_F32(64, 16, true)
    M2D_SCAN_F32(128, 10, true) M2D_SCAN_F32(128, 16, true) M2D_SCAN_F32(256, 10, true) M2D_SCAN_F32(256, 16, true)
#undef M2D_SCAN_F32
    h->last_error = "m2d_topk_scan_f32_launch: no such instantiation";
    return M2D_ERR_UNSUPPORTED;
}
'''

SCAN_BF16 = '''// Pattern-grouped scan on split-bf16 MFMA (the default for 0/1 masks at E = 64 / 128, and E = 32 / 64 with the ingredient table):
// the first form (m2d_topk_grouped_bf16, kept as the A/B reference) and the pipelined form that is launched
// (m2d_topk_grouped_bf16_pipe2).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(2330, 2543) + '''
''' + cut(2545, 3243) + '''
}  // namespace

int m2d_topk_scan_bf16_launch(m2d_engine *h, const GroupedArgs &a, const ScanShape &s, dim3 grid, size_t lds, hipStream_t st)
{
#define M2D_SCAN_GO(KERN, THREADS)                                                      \\
    {                                                                                   \\
        auto kern = KERN;                                                               \\
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)kern, (int)lds));                    \\
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, st, a);                      \\
        return M2D_OK;                                                                  \\
    }
#define M2D_SCAN_BF16(EV, KRV)                                                                                                   \\
    if (s.E == EV && s.KR == KRV) {                                                                                               \\
        constexpr bool CAN_KEEP = !(EV == 128 && KRV == 16);                                                                      \\
        if (s.hv) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, true>), 512)                                               \\
        if (!s.pipe) M2D_SCAN_GO((m2d_topk_grouped_bf16<EV, 8, KRV>), 512)                                                        \\
        if constexpr (EV == 64) {                                                                                                 \\
            if (s.waves == 4 && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 4, true>), 256)               \\
            if (s.waves == 4) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 4>), 256)                               \\
        }                                                                                                                         \\
        if (CAN_KEEP && s.keep) M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1, false, 8, CAN_KEEP>), 512)                   \\
        M2D_SCAN_GO((m2d_topk_grouped_bf16_pipe2<EV, KRV, 1>), 512)                                                               \\
    }
    M2D_SCAN_BF16(64, 10) M2D_SCAN_BF16(64, 16) M2D_SCAN_BF16(128, 10) M2D_SCAN_BF16(128, 16)
#undef M2D_SCAN_BF16
#undef M2D_SCAN_GO
    h->last_error = "m2d_topk_scan_bf16_launch: no such instantiation";
    return M2D_ERR_UNSUPPORTED;
}
'''

MERGE = '''// What follows a pattern-grouped (or dense) scan: the dish ranges' partial lists merged into a user's list, lists shorter than k
// completed, tied users listed for the repair, and near-tied lists finished in the repair's plain-f32 arithmetic (m2d_topk_refine).
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(544, 549) + cut(575, 710) + '''
''' + cut(712, 719) + '''
''' + cut(734, 745) + '''
''' + cut(1154, 1163) + cut(1175, 1334) + '''
}  // namespace

''' + cut(3321, 3354).replace(' = nullptr', '').replace('int64_t I = 0', 'int64_t I') + '''
void m2d_topk_launch_fill_absent(float *scores, int32_t *ids, int64_t nU, int k, int64_t I, hipStream_t st)
{
    hipLaunchKernelGGL(m2d_topk_fill_absent, dim3((unsigned)((nU + 127) / 128)), dim3(128), 0, st, scores, ids, nU, k, I);
}

void m2d_topk_launch_tie_compact(const float *tie_final, int64_t nU, int32_t *tie_list, float *scores, int32_t *ids, int k, int64_t I,
                                 int refined, hipStream_t st)
{
    hipLaunchKernelGGL(m2d_topk_tie_compact, dim3((unsigned)((nU + 255) / 256)), dim3(256), 0, st, tie_final, nU, tie_list, scores, ids, k, I,
                       refined);
}

// near-tied lists: finished in the repair's arithmetic (may add to the repair's list).  flag_pass: a launch with ONE dish range has
// no merge pass to decide who is near-tied -- m2d_topk_refine_flag does
void m2d_topk_launch_refine(const RefineArgs &f, bool flag_pass, hipStream_t st)
{
    if (flag_pass) hipLaunchKernelGGL(m2d_topk_refine_flag, dim3((unsigned)((f.nU + 255) / 256)), dim3(256), 0, st, f);
    if (f.E <= 64) hipLaunchKernelGGL(m2d_topk_refine<1>, dim3((unsigned)((f.nU + 63) / 64)), dim3(256), 0, st, f);
    else hipLaunchKernelGGL(m2d_topk_refine<2>, dim3((unsigned)((f.nU + 63) / 64)), dim3(256), 0, st, f);
}
'''

REPAIR = '''// The tie repair of pattern-grouped retrieval: users whose final k-th score is tied with three or more dishes left out (copies of
// dishes, all-zero users; with "topk_refine" = 0 every user tied at its list's end) are re-ranked over their relevant patterns.
#include "m2d_catalogue.h"

#pragma clang fp contract(off)

namespace {

''' + cut(721, 731) + '''
''' + cut(802, 1151) + '''
}  // namespace

// r.cap, r.part_s / part_i and the lists are the caller's (launch_grouped); hv: the ingredient table's rows
int m2d_topk_launch_repair(m2d_engine *h, const RepairArgs &r, bool hv, hipStream_t st)
{
    const int ub = (!hv && r.E <= 128) ? 4 : 2;                  // listed users per pass of the repair scan (LDS: 21 E + 128 k floats each)
    const size_t slds = (size_t)ub * ((size_t)(r.C + 1 + 16) * r.E + (size_t)2 * 64 * r.k) * sizeof(float);
    const size_t rlds = ((size_t)(r.C + 1 + 16) * r.E + (size_t)2 * 16 * r.k) * sizeof(float);
#define M2D_REPAIR(UBV, HVV)                                                                          \\
    if (ub == UBV && hv == HVV) {                                                                     \\
        auto rk = m2d_topk_repair_scan<UBV, HVV>;                                                     \\
        auto fk = m2d_topk_repair_finish<HVV>;                                                        \\
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)fk, (int)rlds));                                   \\
        M2D_HIP_TRY(h, m2d_lds_limit((const void *)rk, (int)slds));                                   \\
        hipLaunchKernelGGL(rk, dim3(REPAIR_SPLITS, 8), dim3(1024), slds, st, r);                      \\
        hipLaunchKernelGGL(fk, dim3((unsigned)(h->num_cu * 2)), dim3(256), rlds, st, r);              \\
    }
    M2D_REPAIR(4, false) M2D_REPAIR(2, false) M2D_REPAIR(2, true)
#undef M2D_REPAIR
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}
'''

def sub(text, old, new, count=1):
    assert text.count(old) >= 1, old[:80]
    return text.replace(old, new, count)


# ---- hand edits on the cut text ---------------------------------------------------------------------------------------------
HEADER = sub(HEADER, "static unsigned long long *g_m2d_diag_buffer = nullptr;", "__attribute__((unused)) static unsigned long long *g_m2d_diag_buffer = nullptr;")
DENSE = sub(DENSE, """    hipLaunchKernelGGL(m2d_topk_fill_absent, dim3((unsigned)((a.nU + 127) / 128)), dim3(128), 0, st, final_s, final_i,
                       a.nU, a.k, a.I);
""", """    m2d_topk_launch_fill_absent(final_s, final_i, a.nU, a.k, a.I, st);
""")
# launch_grouped: the instantiation is a run-time description now (ScanShape); the kernels live in other units
PLAN = sub(PLAN, """template <int E8, int WAVES, int KR, bool BF16X3, bool HV = false, bool PAD = false>
int launch_grouped(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *final_s, int32_t *final_i,
                   hipStream_t st)
{
    static_assert(!HV || BF16X3, "the ingredient form exists for the pipelined split-bf16 kernel only");
    static_assert(!PAD || !BF16X3, "zero-padded rows are served by the exact-f32 kernel");
    constexpr int E = E8 * 8;
""", """// One pattern-grouped call: plan -> scan -> merge of the dish ranges -> refinement / tie repair.
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
""")
i0 = PLAN.index("    const dim3 grid = a.items ? dim3((unsigned)(ublocks * nsplit)) : dim3((unsigned)ublocks, (unsigned)nsplit);")
i1 = PLAN.index("    M2D_HIP_TRY(h, hipGetLastError());\n    if (nsplit > 1) {\n        m2d_launch_merge_splits2(")
PLAN = PLAN[:i0] + """    const dim3 grid = a.items ? dim3((unsigned)(ublocks * nsplit)) : dim3((unsigned)ublocks, (unsigned)nsplit);
    {
        const ScanShape shape{E, KR, BF16X3, HV, PAD, pipe, WV, a.ex_out != nullptr};
        const int rc = BF16X3 ? m2d_topk_scan_bf16_launch(h, a, shape, grid, lds, st) : m2d_topk_scan_f32_launch(h, a, shape, grid, lds, st);
        if (rc != M2D_OK) return rc;
    }
""" + PLAN[i1:]
i0 = PLAN.index("        const int ub = (!HV && h->E <= 128) ? 4 : 2;")
i1 = PLAN.index("    h->last_kernel = BF16X3 ? \"m2d_topk_grouped_bf16x3\" : \"m2d_topk_grouped\";")
PLAN = PLAN[:i0] + """        if (nsplit == 1)                                     // (with dish ranges the last merge pass has listed the tied users)
            m2d_topk_launch_tie_compact(tie_final, nU, tie_list, final_s, final_i, (int)k, h->I, ext ? 1 : 0, st);
        if (ext) {                                           // near-tied lists: finished in the repair's arithmetic (may add to the repair's list)
            RefineArgs f;
            f.pm = h->pm; f.re = h->re; f.ce = h->ce; f.cats = h->dish_cats; f.plan = a.plan; f.tie_final = tie_final; f.ex = ex_final;
            f.users = users; f.tie_list = tie_list; f.counter = h->topk_refine_counter; f.nU = nU; f.U = h->U; f.I = h->I;
            f.user_base = h->user_base; f.E = h->E; f.k = k; f.a = h->a; f.b = h->b; f.out_scores = final_s; f.out_ids = final_i;
            m2d_topk_launch_refine(f, nsplit == 1, st);      // (with dish ranges the last merge pass has flagged the near-tied users)
        }
        const int rc = m2d_topk_launch_repair(h, r, HV, st);
        if (rc != M2D_OK) return rc;
    }
""" + PLAN[i1:]
# the dispatcher's calls
i0 = PLAN.index("        if (hv_ok && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite) {")
i1 = PLAN.index("    return m2d_topk_dense_launch(h, users, nU, k, out_scores, out_ids, stream);")
PLAN = PLAN[:i0] + """        const int KR = k <= 10 ? 10 : 16;                    // list slots per lane
        if (hv_ok && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite)       // rows [H[d] | RE[d]]: width 2 E
            return launch_grouped(h, 2 * h->E / 8, KR, true, true, false, users, nU, k, out_scores, out_ids, stream);
        if (!h->dish_high && h->grp_binary && h->grp_tiles > 0 && !h->grp_nonfinite) {
            // "topk_bf16x3" option: 1 = split-bf16 MFMA (E = 64 / 128), 0 = exact-f32 MFMA; embedding sizes without a kernel of
            // their own (`padded`) run exact f32 on rows zero-padded to `roww` floats
            const bool x3 = h->opt_topk_bf16x3 != 0 && (h->E == 64 || h->E == 128);
            return launch_grouped(h, roww / 8, KR, x3, false, padded, users, nU, k, out_scores, out_ids, stream);
        }
    }
""" + PLAN[i1:]

out = {"m2d_catalogue.h": HEADER, "m2d_catalogue_dense.hip": DENSE, "m2d_catalogue_plan.hip": PLAN, "m2d_catalogue_scan_f32.hip": SCAN_F32,
       "m2d_catalogue_scan_bf16.hip": SCAN_BF16, "m2d_catalogue_merge.hip": MERGE, "m2d_catalogue_repair.hip": REPAIR}
for name, text in out.items():
    with open(os.path.join(ROOT, "foodrec_amd", "csrc", name), "w") as f:
        f.write(text)
    print(name, text.count("\n"))
