// extern "C" surface of libm2d.so -- see include/m2d.h for the contract and the reference
// interfaces each entry point replaces.  No exceptions cross this boundary.
#include <string.h>
#include <time.h>

#include <new>

#include "m2d_engine.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// plain streaming read: 4 x 16 B in flight per lane, non-temporal, the four loads a whole grid stride apart, TWO workgroups per CU.
// Round 6 sweep on MI355X over the 1.28 GB Personal_Memory (profiles/r06_stream_probe_sweep.txt):
// this form 6 970-7 000 GB/s at 2-3 workgroups per CU against 5 560-5 840 at 4-16 (round 1-5's launch: 8 per CU), 6 250 at 32;
// plain loads 6 310 at 2 per CU; block-contiguous slices (4 or 8 x 16 B per lane) 6 610-6 740 at 2 per CU.
__global__ __launch_bounds__(256) void m2d_stream_read(const v4f *p, int64_t n4, float *sink)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const v4f a = __builtin_nontemporal_load(p + i);
        const v4f b = __builtin_nontemporal_load(p + i + stride);
        const v4f c = __builtin_nontemporal_load(p + i + 2 * stride);
        const v4f d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += (a + b) + (c + d);
    }
    for (; i < n4; i += stride) acc += __builtin_nontemporal_load(p + i);
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 123456.789f) sink[0] = s;   // keeps the loads live; practically never true
}

// x * 0 is 0 for every finite x and NaN for inf / NaN: one fma per value, the lane's sum is NaN iff it met one
__global__ __launch_bounds__(256) void m2d_scan_nonfinite(const float *p, int64_t n, int32_t *flag)
{
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
    const v4f *p4 = reinterpret_cast<const v4f *>(p);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const v4f a = __builtin_nontemporal_load(p4 + i);
        const v4f b = __builtin_nontemporal_load(p4 + i + stride);
        const v4f c = __builtin_nontemporal_load(p4 + i + 2 * stride);
        const v4f d = __builtin_nontemporal_load(p4 + i + 3 * stride);
        acc += (a * 0.f + b * 0.f) + (c * 0.f + d * 0.f);
    }
    for (; i < n4; i += stride) acc += __builtin_nontemporal_load(p4 + i) * 0.f;
    float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += p[(n4 << 2) + threadIdx.x] * 0.f;
    if (s != s) *flag = 1;
}

// the Personal_Memory blocks of a batch's users (after a writer that adds into them with atomics)
__global__ __launch_bounds__(256) void m2d_rows_nonfinite(const float *pm, const int32_t *users, int64_t B, int64_t U, int64_t user_base,
                                                          int32_t W, int32_t *flag)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    float s = 0.f;
    for (int64_t b = wave0; b < B; b += nwaves) {
        const int64_t ul = (int64_t)users[b] - user_base;
        if (ul < 0 || ul >= U) continue;                     // reported by the writer itself
        const float *row = pm + (size_t)ul * W;
        for (int e = lane; e < W; e += 64) s += row[e] * 0.f;
    }
    if (s != s) *flag = 1;
}

std::string g_create_error;

int fail(m2d_engine *h, int code, const char *msg)
{
    if (h) h->last_error = msg; else g_create_error = msg;
    return code;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// copy a host table to HBM (engine-owned) or borrow a device pointer
int adopt_table(m2d_engine *h, const float *src, size_t count, int flags, const float **dst, bool *own)
{
    if (flags == M2D_TABLES_DEVICE) {
        *dst = src;
        *own = false;
        return M2D_OK;
    }
    float *d = nullptr;
    M2D_HIP_TRY(h, hipMalloc((void **)&d, count * sizeof(float)));
    hipError_t e = hipMemcpy(d, src, count * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        h->last_error = std::string("hipMemcpy(table): ") + hipGetErrorString(e);
        return M2D_ERR_HIP;
    }
    *dst = d;
    *own = true;
    return M2D_OK;
}

void release(m2d_engine *h)
{
    for (hipEvent_t e : h->stage_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    m2d_train_release(h);
    if (h->own_pm && h->pm) (void)hipFree((void *)h->pm);
    if (h->own_re && h->re) (void)hipFree((void *)h->re);
    if (h->own_ce && h->ce) (void)hipFree((void *)h->ce);
    if (h->own_dish_cats && h->dish_cats) (void)hipFree((void *)h->dish_cats);
    if (h->dish_vec) (void)hipFree(h->dish_vec);
    if (h->grp_rs) (void)hipFree(h->grp_rs);
    if (h->grp_rs16) (void)hipFree(h->grp_rs16);
    if (h->grp_perm) (void)hipFree(h->grp_perm);
    if (h->grp_tile_info) (void)hipFree(h->grp_tile_info);
    if (h->grp_work) (void)hipFree(h->grp_work);
    if (h->own_mlp) {
        for (const float *q : {h->mlp_w1, h->mlp_b1, h->mlp_w2, h->mlp_b2, h->mlp_w3})
            if (q) (void)hipFree((void *)q);
    }
    if (h->mlp_w1x3) (void)hipFree(h->mlp_w1x3);
    if (h->mlp_w1pad) (void)hipFree(h->mlp_w1pad);
    if (h->mlp_w1pc) (void)hipFree(h->mlp_w1pc);
    if (h->mlp_pg) (void)hipFree(h->mlp_pg);
    if (h->mlp_pat8) (void)hipFree(h->mlp_pat8);
    if (h->user_high) (void)hipFree(h->user_high);
    if (h->dish_high) (void)hipFree(h->dish_high);
    if (h->own_ing) {
        if (h->ing) (void)hipFree((void *)h->ing);
        if (h->ing_off) (void)hipFree((void *)h->ing_off);
        if (h->ing_ids) (void)hipFree((void *)h->ing_ids);
        if (h->ing_w) (void)hipFree((void *)h->ing_w);
    }
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->topk_flags) (void)hipFree(h->topk_flags);
    if (h->topk_plan) (void)hipFree(h->topk_plan);
    if (h->topk_ex) (void)hipFree(h->topk_ex);
    if (h->err_dev) (void)hipFree(h->err_dev);
    if (h->err_host) (void)hipHostFree(h->err_host);
}

}  // namespace

// The scan queued by m2d_create / m2d_tables_updated: one streaming pass over the three tables on the caller's stream, in
// front of the launch that needs its answer (1.28 GB of Personal_Memory: 0.2 ms, once per table change).
int m2d_ensure_finite_scan(m2d_engine *h, hipStream_t st)
{
    if (!h->finite_scan_pending) return M2D_OK;
    M2D_HIP_TRY(h, hipMemsetAsync(h->nonfinite_dev, 0, sizeof(int32_t), st));
    const int64_t n[3] = {h->U * (int64_t)(h->C + 1) * h->E, h->I * (int64_t)h->E, (int64_t)h->C * h->E};
    const float *t[3] = {h->pm, h->re, h->ce};
    for (int i = 0; i < 3; ++i) {
        int64_t blocks = (n[i] / 4 + 1023) / 1024;
        if (blocks > (int64_t)h->num_cu * 8) blocks = (int64_t)h->num_cu * 8;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(m2d_scan_nonfinite, dim3((unsigned)blocks), dim3(256), 0, st, t[i], n[i], h->nonfinite_dev);
    }
    M2D_HIP_TRY(h, hipGetLastError());
    h->finite_scan_pending = false;
    return M2D_OK;
}

int m2d_launch_rows_finite_check(m2d_engine *h, const int32_t *users, int64_t B, hipStream_t st)
{
    if (B <= 0) return M2D_OK;
    int64_t blocks = (B + 3) / 4;
    if (blocks > (int64_t)h->num_cu * 8) blocks = (int64_t)h->num_cu * 8;
    hipLaunchKernelGGL(m2d_rows_nonfinite, dim3((unsigned)blocks), dim3(256), 0, st, h->pm, users, B, h->U, h->user_base,
                       (h->C + 1) * h->E, h->nonfinite_dev);
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}

extern "C" {

int m2d_abi_version(void) { return 2; }

int m2d_create(const float *pm, const float *re, const float *ce, int64_t U, int64_t I, int32_t C, int32_t E,
               float coef, int device, int table_flags, m2d_engine **out)
{
    if (!out) return fail(nullptr, M2D_ERR_INVALID_ARG, "m2d_create: out is null");
    *out = nullptr;
    if (!pm || !re || !ce) return fail(nullptr, M2D_ERR_INVALID_ARG, "m2d_create: null table pointer");
    if (U <= 0 || I <= 0 || C <= 0 || E <= 0)
        return fail(nullptr, M2D_ERR_INVALID_ARG, "m2d_create: U, I, C, E must be positive");
    if (I > INT32_MAX || U > (int64_t)INT32_MAX)
        return fail(nullptr, M2D_ERR_UNSUPPORTED, "m2d_create: ids are int32 (Model_Recommender.py:26-29)");
    if (table_flags != M2D_TABLES_HOST && table_flags != M2D_TABLES_DEVICE)
        return fail(nullptr, M2D_ERR_INVALID_ARG, "m2d_create: bad table_flags");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, M2D_ERR_NO_DEVICE,
                    "m2d_create: no HIP device visible (this engine has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, M2D_ERR_INVALID_ARG, "m2d_create: bad device index");
    m2d_engine *h = new (std::nothrow) m2d_engine();
    if (!h) return fail(nullptr, M2D_ERR_HIP, "m2d_create: out of host memory");
    h->U = U; h->I = I; h->C = C; h->E = E; h->device = device;
    h->a = coef;            // float32(coef)               Model_Recommender.py:17
    h->b = 1.0f - h->a;     // evaluated in float32        Model_Recommender.py:96
    int rc = M2D_OK;
    auto bail = [&](int code) {
        g_create_error = h->last_error;
        release(h);
        delete h;
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) { h->last_error = "hipSetDevice failed"; return bail(M2D_ERR_HIP); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        h->num_cu = prop.multiProcessorCount;
    if ((rc = adopt_table(h, pm, (size_t)U * (C + 1) * E, table_flags, &h->pm, &h->own_pm)) != M2D_OK) return bail(rc);
    if ((rc = adopt_table(h, re, (size_t)I * E, table_flags, &h->re, &h->own_re)) != M2D_OK) return bail(rc);
    if ((rc = adopt_table(h, ce, (size_t)C * E, table_flags, &h->ce, &h->own_ce)) != M2D_OK) return bail(rc);
    if (!aligned16(h->pm) || !aligned16(h->re) || !aligned16(h->ce)) {
        h->last_error = "m2d_create: device tables must be 16-byte aligned";
        return bail(M2D_ERR_INVALID_ARG);
    }
    if (hipMalloc((void **)&h->err_dev, 8 * sizeof(int32_t)) != hipSuccess ||       // id-error latch [4] | non-finite word | pad
        hipMemset(h->err_dev, 0, 8 * sizeof(int32_t)) != hipSuccess ||
        hipHostMalloc((void **)&h->err_host, 4 * sizeof(int32_t)) != hipSuccess) {
        h->last_error = "m2d_create: could not allocate the error latch";
        return bail(M2D_ERR_HIP);
    }
    h->nonfinite_dev = h->err_dev + 4;
    h->finite_scan_pending = true;      // the first scoring call scans the tables on its stream
    *out = h;
    return M2D_OK;
}

int m2d_destroy(m2d_engine *h)
{
    if (!h) return M2D_OK;
    (void)hipSetDevice(h->device);
    release(h);
    delete h;
    return M2D_OK;
}

const char *m2d_last_error(const m2d_engine *h) { return h ? h->last_error.c_str() : g_create_error.c_str(); }

const char *m2d_last_kernel(const m2d_engine *h) { return h ? h->last_kernel : ""; }

int m2d_set_user_base(m2d_engine *h, int64_t user_base)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (user_base < 0 || user_base + h->U > (int64_t)INT32_MAX + 1)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_user_base: shard range leaves int32");
    h->user_base = user_base;
    return M2D_OK;
}

int m2d_set_dish_categories(m2d_engine *h, const float *cats, int table_flags)
{
    if (!h || !cats) return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_dish_categories: null argument");
    if (table_flags != M2D_TABLES_HOST && table_flags != M2D_TABLES_DEVICE)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_dish_categories: bad table_flags");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    if (h->own_dish_cats && h->dish_cats) (void)hipFree((void *)h->dish_cats);
    h->dish_cats = nullptr;
    h->own_dish_cats = false;
    h->dish_vec_valid = false;
    h->grp_valid = false;
    int rc = adopt_table(h, cats, (size_t)h->I * h->C, table_flags, &h->dish_cats, &h->own_dish_cats);
    if (rc != M2D_OK) return rc;
    if (!aligned16(h->dish_cats)) return fail(h, M2D_ERR_INVALID_ARG, "dish categories must be 16-byte aligned");
    return M2D_OK;
}

int m2d_score_pairs(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, int64_t B,
                    float *out, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs: negative batch");
    if (B == 0) return M2D_OK;
    if (!users || !items || !cats || !out) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs: null buffer");
    if (h->C == 4 && !aligned16(cats)) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs: cats must be 16-byte aligned");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_score_pairs(h, users, items, cats, false, B, out, (hipStream_t)stream);
}

// Host-buffer form of m2d_score_pairs: what the reference's own call site hands over (numpy / lists, 51 pairs per
// sess.run, evaluate.py:55-59).  One pinned staging block [users | items | cats | out | err], one H2D copy, the
// kernel, one D2H copy that brings the scores AND the id-error latch back, one synchronisation.
namespace {
__global__ void m2d_copy_latch(const int32_t *latch, int32_t *dst, int32_t *done, int32_t ticket)
{
    if (threadIdx.x < 4) dst[threadIdx.x] = latch[threadIdx.x];
    if (done) {                         // host-visible completion word, written after the latch (and after the scores,
        __threadfence_system();         // which the kernel before this one on the stream wrote)
        if (threadIdx.x == 0) __hip_atomic_store(done, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

namespace {
constexpr int64_t HOST_CHUNK = 1 << 18;

int ensure_stage(m2d_engine *h, size_t total, hipStream_t st)
{
    if (total <= h->stage_bytes) return M2D_OK;
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    h->stage_host = h->stage_dev = nullptr; h->stage_bytes = 0;
    const size_t cap = total * 2;
    // the kernel writes scores, the id-error latch and the completion word into this block and the host polls the word:
    // host-coherent (fine-grained) and mapped, stated rather than left to the runtime's default / HIP_HOST_COHERENT
    M2D_HIP_TRY(h, hipHostMalloc((void **)&h->stage_host, cap, hipHostMallocCoherent | hipHostMallocMapped));
    M2D_HIP_TRY(h, hipMalloc((void **)&h->stage_dev, cap));
    h->stage_bytes = cap;
    return M2D_OK;
}

// Feeds of more than HOST_CHUNK pairs: chunks through two pinned blocks, so that the host's copy of chunk k + 1 into its
// block runs while chunk k is on the link and in the kernel (one block: the five steps of a call run one after another
// and the host-side copies are the longest of them).  Chunks retire in order; an id error is reported with the position
// in the whole feed, and the scores of the chunks before the offending one have then been delivered.
int score_pairs_host_chunked(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, int64_t B,
                             float *out, hipStream_t st)
{
    const int C = h->C;
    const size_t in_b = (size_t)HOST_CHUNK * C * 4 + 2 * (size_t)HOST_CHUNK * 4, out_b = (size_t)HOST_CHUNK * 4 + 16;
    const size_t blk = in_b + out_b;
    // an id error latched by an earlier, unchecked launch is reported as what it is -- with that call's position -- before
    // anything of this feed is queued (the latch keeps the first error; a chunk's position arithmetic below must only
    // ever see an error of its own chunk)
    int rc = m2d_check(h, (void *)st, nullptr, nullptr);
    if (rc != M2D_OK) return rc;
    if ((rc = ensure_stage(h, 2 * blk, st)) != M2D_OK) return rc;
// a HIP error inside the loop: nothing may stay in flight on the two pinned blocks when the call returns
#define M2D_CHUNK_TRY(expr)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            h->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                 \
            (void)hipStreamSynchronize(st);                                                    \
            return M2D_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)
    for (hipEvent_t &e : h->stage_ev)
        if (!e) M2D_HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int64_t nch = (B + HOST_CHUNK - 1) / HOST_CHUNK;
    auto retire = [&](int64_t k) -> int {
        unsigned char *hs = h->stage_host + (size_t)(k & 1) * blk;
        M2D_CHUNK_TRY(hipEventSynchronize(h->stage_ev[k & 1]));
        const int32_t *err = reinterpret_cast<const int32_t *>(hs + in_b + (size_t)HOST_CHUNK * 4);
        if (err[0] != 0) {
            // the kernel latched a position inside its chunk: put the position in the feed there before reporting it
            const int64_t idx = (((int64_t)(uint32_t)err[3] << 32) | (uint32_t)err[2]) + k * HOST_CHUNK;
            int32_t *fixed = reinterpret_cast<int32_t *>(hs);                   // pinned; the block is not in use any more
            fixed[0] = err[0]; fixed[1] = err[1]; fixed[2] = (int32_t)(idx & 0xffffffff); fixed[3] = (int32_t)(idx >> 32);
            M2D_CHUNK_TRY(hipMemcpyAsync(h->err_dev, fixed, 16, hipMemcpyHostToDevice, st));
            return m2d_check(h, (void *)st, nullptr, nullptr);                  // synchronises, formats, clears the latch
        }
        const int64_t n = B - k * HOST_CHUNK < HOST_CHUNK ? B - k * HOST_CHUNK : HOST_CHUNK;
        memcpy(out + k * HOST_CHUNK, hs + in_b, (size_t)n * 4);
        return M2D_OK;
    };
    for (int64_t k = 0; k < nch; ++k) {
        if (k >= 2 && (rc = retire(k - 2)) != M2D_OK) {
            (void)hipStreamSynchronize(st);
            return rc;
        }
        unsigned char *hs = h->stage_host + (size_t)(k & 1) * blk, *ds = h->stage_dev + (size_t)(k & 1) * blk;
        const int64_t o = k * HOST_CHUNK, n = B - o < HOST_CHUNK ? B - o : HOST_CHUNK;
        const size_t o_u = (size_t)HOST_CHUNK * C * 4, o_i = o_u + (size_t)HOST_CHUNK * 4;
        memcpy(hs, cats + o * C, (size_t)n * C * 4);
        memcpy(hs + o_u, users + o, (size_t)n * 4);
        memcpy(hs + o_i, items + o, (size_t)n * 4);
        M2D_CHUNK_TRY(hipMemcpyAsync(ds, hs, (size_t)n * C * 4, hipMemcpyHostToDevice, st));
        M2D_CHUNK_TRY(hipMemcpyAsync(ds + o_u, hs + o_u, (size_t)n * 4, hipMemcpyHostToDevice, st));
        M2D_CHUNK_TRY(hipMemcpyAsync(ds + o_i, hs + o_i, (size_t)n * 4, hipMemcpyHostToDevice, st));
        rc = m2d_launch_score_pairs(h, reinterpret_cast<const int32_t *>(ds + o_u), reinterpret_cast<const int32_t *>(ds + o_i),
                                    reinterpret_cast<const float *>(ds), false, n, reinterpret_cast<float *>(ds + in_b), st);
        if (rc != M2D_OK) {
            (void)hipStreamSynchronize(st);
            return rc;
        }
        hipLaunchKernelGGL(m2d_copy_latch, dim3(1), dim3(64), 0, st, h->err_dev,
                           reinterpret_cast<int32_t *>(ds + in_b + (size_t)HOST_CHUNK * 4), (int32_t *)nullptr, 0);
        M2D_CHUNK_TRY(hipGetLastError());
        M2D_CHUNK_TRY(hipMemcpyAsync(hs + in_b, ds + in_b, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        M2D_CHUNK_TRY(hipMemcpyAsync(hs + in_b + (size_t)HOST_CHUNK * 4, ds + in_b + (size_t)HOST_CHUNK * 4, 16, hipMemcpyDeviceToHost, st));
        M2D_CHUNK_TRY(hipEventRecord(h->stage_ev[k & 1], st));
    }
    for (int64_t k = nch >= 2 ? nch - 2 : 0; k < nch; ++k)
        if ((rc = retire(k)) != M2D_OK) {
            (void)hipStreamSynchronize(st);
            return rc;
        }
    return M2D_OK;
#undef M2D_CHUNK_TRY
}
}  // namespace

int m2d_score_pairs_host(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, int64_t B,
                         float *out, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_host: negative batch");
    if (B == 0) return M2D_OK;
    if (!users || !items || !cats || !out) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_host: null buffer");
    hipStream_t st = (hipStream_t)stream;
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    if (B > HOST_CHUNK && h->opt_host_zero_copy != 0) return score_pairs_host_chunked(h, users, items, cats, B, out, st);
    const int C = h->C;
    // layout (16-byte aligned sections): cats [B, C] | users [B] | items [B] || out [B] | err [4] | completion word
    const size_t nb = ((size_t)B + 3) & ~(size_t)3;
    const size_t in_bytes = nb * C * 4 + 2 * nb * 4, out_bytes = nb * 4 + 16, total = in_bytes + out_bytes + 16;
    {
        const int rc_stage = ensure_stage(h, total, st);
        if (rc_stage != M2D_OK) return rc_stage;
    }
    unsigned char *hs = h->stage_host, *ds = h->stage_dev;
    memcpy(hs, cats, (size_t)B * C * 4);
    memcpy(hs + nb * C * 4, users, (size_t)B * 4);
    memcpy(hs + nb * C * 4 + nb * 4, items, (size_t)B * 4);
    // Small feeds (the reference's 51 pairs per call, evaluate.py:55-59): the kernel reads the pinned block and writes the
    // scores into it over the host link -- two launches and one synchronisation, no copy engine round trips.
    unsigned char *ws = ds;
    if (h->opt_host_zero_copy && B <= 65536) {
        void *mapped = nullptr;
        M2D_HIP_TRY(h, hipHostGetDevicePointer(&mapped, hs, 0));
        ws = static_cast<unsigned char *>(mapped);
    } else {
        M2D_HIP_TRY(h, hipMemcpyAsync(ds, hs, in_bytes, hipMemcpyHostToDevice, st));
    }
    const int rc = m2d_launch_score_pairs(h, reinterpret_cast<const int32_t *>(ws + nb * C * 4),
                                          reinterpret_cast<const int32_t *>(ws + nb * C * 4 + nb * 4),
                                          reinterpret_cast<const float *>(ws), false, B, reinterpret_cast<float *>(ws + in_bytes), st);
    if (rc != M2D_OK) return rc;
    // the latch rides back behind the scores: put it next to them first (16 B, same stream)
    const bool poll = ws != ds && h->opt_host_zero_copy >= 2;
    int32_t *done_dev = poll ? reinterpret_cast<int32_t *>(ws + in_bytes + nb * 4 + 16) : nullptr;
    volatile int32_t *done_host = reinterpret_cast<volatile int32_t *>(hs + in_bytes + nb * 4 + 16);
    const int32_t ticket = (int32_t)++h->stage_ticket;
    if (poll) *done_host = (int32_t)(h->stage_ticket - 1u);
    hipLaunchKernelGGL(m2d_copy_latch, dim3(1), dim3(64), 0, st, h->err_dev, reinterpret_cast<int32_t *>(ws + in_bytes + nb * 4),
                       done_dev, ticket);
    M2D_HIP_TRY(h, hipGetLastError());
    if (ws == ds) M2D_HIP_TRY(h, hipMemcpyAsync(hs + in_bytes, ds + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
    bool seen = false;
    if (poll) {
        // spin on the completion word (a stream synchronisation costs more than the two kernels), for at most 250 us of
        // wall time -- a 65 536-pair feed, the largest that takes this path, is done in ~0.1 ms -- then wait on the stream
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            for (int spin = 0; spin < 256 && !seen; ++spin)
                seen = __atomic_load_n(const_cast<const int32_t *>(done_host), __ATOMIC_ACQUIRE) == ticket;
            if (seen) break;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 250000ll) break;
        }
    }
    if (!seen) M2D_HIP_TRY(h, hipStreamSynchronize(st));
    const int32_t *err = reinterpret_cast<const int32_t *>(hs + in_bytes + nb * 4);
    if (err[0] != 0) return m2d_check(h, stream, nullptr, nullptr);     // formats the message, clears the latch
    memcpy(out, hs + in_bytes, (size_t)B * 4);
    return M2D_OK;
}

int m2d_score_pairs_bydish(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B, float *out,
                           void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_bydish: negative batch");
    if (B == 0) return M2D_OK;
    if (!users || !items || !out) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_bydish: null buffer");
    if (!h->dish_cats) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_dish_categories first");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_score_pairs(h, users, items, h->dish_cats, true, B, out, (hipStream_t)stream);
}

int m2d_rank_candidates(m2d_engine *h, const int32_t *users, const int32_t *items, const int32_t *lens,
                        int64_t nseg, int32_t L, int32_t k, float *out_scores, int32_t *out_items,
                        int32_t *out_flags, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (nseg < 0 || L < 1 || L > 1024 || k < 1 || k > 64)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_rank_candidates: need nseg >= 0, 1 <= L <= 1024, 1 <= k <= 64");
    if (nseg == 0) return M2D_OK;
    if (!users || !items || !out_scores || !out_items || !out_flags)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_rank_candidates: null buffer");
    if (!h->dish_cats) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_dish_categories first");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_rank_candidates(h, users, items, lens, nseg, L, k, out_scores, out_items, out_flags,
                                      (hipStream_t)stream);
}

int m2d_topk_users(m2d_engine *h, const int32_t *users, int64_t nU, int32_t k, float *out_scores,
                   int32_t *out_ids, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (nU < 0 || k < 1 || k > 64 || (int64_t)k > h->I)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_topk_users: need nU >= 0 and 1 <= k <= min(64, I)");
    if (nU == 0) return M2D_OK;
    if (!users || !out_scores || !out_ids) return fail(h, M2D_ERR_INVALID_ARG, "m2d_topk_users: null buffer");
    if (!h->dish_cats) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_dish_categories first");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_topk_users(h, users, nU, k, out_scores, out_ids, (hipStream_t)stream);
}

int m2d_clear_ingredients(m2d_engine *h)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->own_ing) {
        if (h->ing) (void)hipFree((void *)h->ing);
        if (h->ing_off) (void)hipFree((void *)h->ing_off);
        if (h->ing_ids) (void)hipFree((void *)h->ing_ids);
        if (h->ing_w) (void)hipFree((void *)h->ing_w);
    }
    if (h->dish_high) (void)hipFree(h->dish_high);
    h->ing = nullptr; h->ing_off = nullptr; h->ing_ids = nullptr; h->ing_w = nullptr; h->dish_high = nullptr;
    h->own_ing = false; h->ing_rows = 0; h->ing_nnz = 0;
    h->dish_vec_valid = false;
    h->grp_valid = false;               // the pattern-grouped retrieval rows carry H[d] when it is set
    return M2D_OK;
}

int m2d_set_ingredients(m2d_engine *h, const float *ing, int64_t R, const int32_t *off, const int32_t *ids,
                        const float *w, int64_t nnz, int table_flags)
{
    if (!h || !ing || !off || (!ids && nnz > 0) || R <= 0 || nnz < 0 || nnz > INT32_MAX)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_ingredients: bad argument");
    if (table_flags != M2D_TABLES_HOST && table_flags != M2D_TABLES_DEVICE)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_ingredients: bad table_flags");
    // an id error latched by an earlier launch is reported as what it is, before the CSR check can mislabel it
    int rc = m2d_check(h, nullptr, nullptr, nullptr);
    if (rc != M2D_OK) return rc;
    if ((rc = m2d_clear_ingredients(h)) != M2D_OK) return rc;
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    if (table_flags == M2D_TABLES_DEVICE) {
        h->ing = ing; h->ing_off = off; h->ing_ids = ids; h->ing_w = w; h->own_ing = false;
    } else {
        bool own = false;
        const float *t = nullptr;
        if ((rc = adopt_table(h, ing, (size_t)R * h->E, table_flags, &h->ing, &own)) != M2D_OK) return rc;
        h->own_ing = true;
        if ((rc = adopt_table(h, reinterpret_cast<const float *>(off), (size_t)h->I + 1, table_flags, &t, &own)) != M2D_OK) return rc;
        h->ing_off = reinterpret_cast<const int32_t *>(t);
        if (nnz > 0) {
            if ((rc = adopt_table(h, reinterpret_cast<const float *>(ids), (size_t)nnz, table_flags, &t, &own)) != M2D_OK) return rc;
            h->ing_ids = reinterpret_cast<const int32_t *>(t);
            if (w) {
                if ((rc = adopt_table(h, w, (size_t)nnz, table_flags, &h->ing_w, &own)) != M2D_OK) return rc;
            }
        }
    }
    h->ing_rows = R;
    h->ing_nnz = nnz;
    M2D_HIP_TRY(h, hipMalloc((void **)&h->dish_high, (size_t)h->I * h->E * sizeof(float)));
    if (!aligned16(h->dish_high)) return fail(h, M2D_ERR_HIP, "dish_high not aligned");
    if ((rc = m2d_launch_check_csr(h, nullptr)) != M2D_OK) return rc;
    rc = m2d_check(h, nullptr, nullptr, nullptr);
    if (rc != M2D_OK) {
        (void)m2d_clear_ingredients(h);
        h->last_error = "m2d_set_ingredients: offsets are not a CSR row pointer (off[0] = 0, non-decreasing, off[I] = nnz)";
        return M2D_ERR_BAD_INGREDIENT;
    }
    if ((rc = m2d_launch_build_dish_high(h, nullptr)) != M2D_OK) return rc;
    rc = m2d_check(h, nullptr, nullptr, nullptr);
    if (rc != M2D_OK) {
        std::string msg = h->last_error;
        (void)m2d_clear_ingredients(h);
        h->last_error = "m2d_set_ingredients: " + msg;
        return M2D_ERR_BAD_INGREDIENT;
    }
    h->dish_vec_valid = false;
    h->grp_valid = false;
    return M2D_OK;
}

int m2d_score_pairs_ingredients(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                                int64_t B, float *out, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_ingredients: negative batch");
    if (B == 0) return M2D_OK;
    if (!users || !items || !out) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_ingredients: null buffer");
    if (!h->dish_high) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_ingredients first");
    if (!cats && !h->dish_cats) return fail(h, M2D_ERR_NOT_CONFIGURED, "cats == NULL needs m2d_set_dish_categories");
    if (cats && h->C == 4 && !aligned16(cats)) return fail(h, M2D_ERR_INVALID_ARG, "cats must be 16-byte aligned");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_score_pairs(h, users, items, cats ? cats : h->dish_cats, cats == nullptr, B, out,
                                  (hipStream_t)stream, /*use_ingredients=*/true);
}

int m2d_write_memory(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats,
                     const float *write_sign, const float *labels, int64_t B, int32_t L, float *general_memory,
                     float beta_1, float beta_2, float alpha, int32_t which, double *out_sums, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0 || L <= 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_write_memory: need B >= 0 and L > 0");
    if (which <= 0 || (which & ~(M2D_WRITE_PERSONAL | M2D_WRITE_GENERAL)))
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_write_memory: which must be M2D_WRITE_PERSONAL, M2D_WRITE_GENERAL or both");
    if (!general_memory || (B > 0 && (!users || !items || !cats || !write_sign || !labels)))
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_write_memory: null buffer");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_write_memory(h, users, items, cats, write_sign, labels, B, L, general_memory, beta_1, beta_2,
                                   alpha, which, out_sums, (hipStream_t)stream);
}

int m2d_clear_mlp_head(m2d_engine *h)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->own_mlp) {
        for (const float *q : {h->mlp_w1, h->mlp_b1, h->mlp_w2, h->mlp_b2, h->mlp_w3})
            if (q) (void)hipFree((void *)q);
    }
    if (h->mlp_w1x3) (void)hipFree(h->mlp_w1x3);
    h->mlp_w1x3 = nullptr;
    if (h->mlp_w1pad) (void)hipFree(h->mlp_w1pad);
    h->mlp_w1pad = nullptr;
    if (h->mlp_w1pc) (void)hipFree(h->mlp_w1pc);
    h->mlp_w1pc = nullptr;
    h->mlp_w1 = h->mlp_b1 = h->mlp_w2 = h->mlp_b2 = h->mlp_w3 = nullptr;
    h->own_mlp = false;
    h->mlp_h1 = h->mlp_h2 = 0;
    return M2D_OK;
}

int m2d_set_mlp_head(m2d_engine *h, const float *W1, const float *b1, const float *W2, const float *b2,
                     const float *w3, float b3, int32_t H1, int32_t H2, int table_flags)
{
    if (!h || !W1 || !b1 || !W2 || !b2 || !w3 || H1 <= 0 || H2 <= 0)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_mlp_head: bad argument");
    if (table_flags != M2D_TABLES_HOST && table_flags != M2D_TABLES_DEVICE)
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_mlp_head: bad table_flags");
    int rc = m2d_clear_mlp_head(h);
    if (rc != M2D_OK) return rc;
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    const size_t K = (size_t)(h->C + 1) * h->E;
    bool own = false;
    if ((rc = adopt_table(h, W1, K * H1, table_flags, &h->mlp_w1, &own)) != M2D_OK) return rc;
    h->own_mlp = own;
    if ((rc = adopt_table(h, b1, (size_t)H1, table_flags, &h->mlp_b1, &own)) != M2D_OK) return rc;
    if ((rc = adopt_table(h, W2, (size_t)H1 * H2, table_flags, &h->mlp_w2, &own)) != M2D_OK) return rc;
    if ((rc = adopt_table(h, b2, (size_t)H2, table_flags, &h->mlp_b2, &own)) != M2D_OK) return rc;
    if ((rc = adopt_table(h, w3, (size_t)H2, table_flags, &h->mlp_w3, &own)) != M2D_OK) return rc;
    if (!aligned16(h->mlp_w1) || !aligned16(h->mlp_w2)) return fail(h, M2D_ERR_INVALID_ARG, "MLP weights must be 16-byte aligned");
    h->mlp_b3 = b3;
    h->mlp_h1 = H1;
    h->mlp_h2 = H2;
    return M2D_OK;
}

int m2d_score_pairs_mlp(m2d_engine *h, const int32_t *users, const int32_t *items, int64_t B, float *out,
                        void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (B < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_mlp: negative batch");
    if (B == 0) return M2D_OK;
    if (!users || !items || !out) return fail(h, M2D_ERR_INVALID_ARG, "m2d_score_pairs_mlp: null buffer");
    if (!h->mlp_w1) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_mlp_head first");
    if (!h->dish_cats) return fail(h, M2D_ERR_NOT_CONFIGURED, "call m2d_set_dish_categories first");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_score_pairs_mlp(h, users, items, B, out, (hipStream_t)stream);
}

int m2d_train_begin(m2d_engine *h, int32_t learner, float lr, float clip_norm, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (learner < M2D_LEARNER_SGD || learner > M2D_LEARNER_ADAM) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_begin: unknown learner");
    if (!(clip_norm > 0.f)) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_begin: clip_norm must be positive");
    if (h->ing) return fail(h, M2D_ERR_UNSUPPORTED, "m2d_train_begin: the ingredient extension has no training step");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_train_setup(h, learner, lr, clip_norm, (hipStream_t)stream);
}

int m2d_train_step(m2d_engine *h, const int32_t *users, const int32_t *items, const float *cats, const float *labels,
                   int64_t B, int32_t apply, float *out, void *stream)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    if (!h->train) return fail(h, M2D_ERR_NOT_CONFIGURED, "m2d_train_step: call m2d_train_begin first");
    if (B <= 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_step: need B > 0 (the mean of an empty batch is NaN)");
    if (!users || !items || !cats || !labels) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_step: null buffer");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    return m2d_launch_train_step(h, users, items, cats, labels, B, apply, out, (hipStream_t)stream);
}

int m2d_train_slot(m2d_engine *h, int32_t table, int32_t slot, float *buf, int32_t restore, void *stream)
{
    if (!h || !buf) return M2D_ERR_INVALID_ARG;
    if (!h->train) return fail(h, M2D_ERR_NOT_CONFIGURED, "m2d_train_slot: call m2d_train_begin first");
    if (table < 0 || table > 2 || slot < 0 || slot > 1) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_slot: table 0..2, slot 0..1");
    float *dev = nullptr;
    int64_t count = 0;
    (void)m2d_train_get_slot(h, table, slot, &dev, &count);
    if (!dev) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_slot: this learner has no such slot");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    M2D_HIP_TRY(h, hipMemcpyAsync(restore ? dev : buf, restore ? buf : dev, (size_t)count * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return M2D_OK;
}

int m2d_tables_updated(m2d_engine *h)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    h->dish_vec_valid = false;      // factored dish vectors (Recipe_Embedding, Category_Embedding)
    h->user_high_valid = false;     // <U_high, CE_c> (Personal_Memory, Category_Embedding)
    h->grp_valid = false;           // pattern-grouped retrieval tables (Recipe_Embedding)
    h->finite_scan_pending = true;  // "every table value is finite" has to be established again
    return M2D_OK;
}

int m2d_train_steps(m2d_engine *h, int64_t *steps, int32_t set)
{
    if (!h || !steps) return M2D_ERR_INVALID_ARG;
    if (!h->train) return fail(h, M2D_ERR_NOT_CONFIGURED, "m2d_train_steps: call m2d_train_begin first");
    if (set && *steps < 0) return fail(h, M2D_ERR_INVALID_ARG, "m2d_train_steps: negative step count");
    return m2d_train_step_count(h, steps, set);
}

int m2d_train_end(m2d_engine *h)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    m2d_train_release(h);
    return M2D_OK;
}

int m2d_check(m2d_engine *h, void *stream, int64_t *bad_value, int64_t *bad_index)
{
    if (!h) return M2D_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    M2D_HIP_TRY(h, hipMemcpyAsync(h->err_host, h->err_dev, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    const int code = h->err_host[0];
    if (code == 0) return M2D_OK;
    const int64_t value = h->err_host[1];
    const int64_t index = ((int64_t)(uint32_t)h->err_host[3] << 32) | (uint32_t)h->err_host[2];
    if (bad_value) *bad_value = value;
    if (bad_index) *bad_index = index;
    M2D_HIP_TRY(h, hipMemsetAsync(h->err_dev, 0, 4 * sizeof(int32_t), st));
    M2D_HIP_TRY(h, hipStreamSynchronize(st));
    if (code == M2D_ERR_KERNEL_TIMEOUT) {
        h->last_error = "m2d_topk_users: a wave of workgroup " + std::to_string(h->err_host[2]) + " gave up waiting for its workgroup's progress "
                        "words (stage " + std::to_string(value) + "): that call's lists are invalid";
        return code;
    }
    const char *what = code == M2D_ERR_BAD_USER_ID ? "user" : code == M2D_ERR_BAD_ITEM_ID ? "item" : "ingredient";
    h->last_error = std::string(what) + " id " + std::to_string(value) + " at position " +
                    std::to_string(code == M2D_ERR_BAD_INGREDIENT ? (int64_t)h->err_host[2] : index) + " is out of range";
    return code;
}

int m2d_stream_read_probe(m2d_engine *h, const void *buf, int64_t bytes, float *sink, void *stream)
{
    if (!h || !buf || !sink || bytes <= 0 || (bytes & 15) || !aligned16(buf))
        return fail(h, M2D_ERR_INVALID_ARG, "m2d_stream_read_probe: need a 16-byte aligned buffer and size");
    M2D_HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(m2d_stream_read, dim3(h->num_cu * 2), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const v4f *>(buf), bytes / 16, sink);
    M2D_HIP_TRY(h, hipGetLastError());
    return M2D_OK;
}

int m2d_set_option(m2d_engine *h, const char *name, int64_t value)
{
    if (!h || !name) return M2D_ERR_INVALID_ARG;
    if (!strcmp(name, "prefetch")) h->opt_prefetch = (int)value;
    else if (!strcmp(name, "nt_loads")) h->opt_nt = (int)value;
    else if (!strcmp(name, "blocks_per_cu")) h->opt_blocks_per_cu = (int)value;
    else if (!strcmp(name, "variant")) h->opt_variant = (int)value;
    else if (!strcmp(name, "topk_bf16x3")) h->opt_topk_bf16x3 = (int)value;
    else if (!strcmp(name, "topk_form")) h->opt_topk_form = (int)value;
    else if (!strcmp(name, "topk_prune")) h->opt_topk_prune = (int)value;
    else if (!strcmp(name, "topk_block")) h->opt_topk_block = (int)value;
    else if (!strcmp(name, "topk_refine")) h->opt_topk_refine = (int)value;
    else if (!strcmp(name, "topk_grouped")) h->opt_topk_grouped = (int)value;
    else if (!strcmp(name, "mlp_bf16x3")) h->opt_mlp_bf16x3 = (int)value;
    else if (!strcmp(name, "mlp_form")) h->opt_mlp_form = (int)value;
    else if (!strcmp(name, "skip_masked")) h->opt_skip_masked = (int)value;
    else if (!strcmp(name, "user_high_table")) h->opt_user_high = (int)value;
    else if (!strcmp(name, "host_zero_copy")) h->opt_host_zero_copy = (int)value;
    else return fail(h, M2D_ERR_INVALID_ARG, "m2d_set_option: unknown option");
    return M2D_OK;
}

int m2d_get_option(const m2d_engine *h, const char *name, int64_t *value)
{
    if (!h || !name || !value) return M2D_ERR_INVALID_ARG;
    if (!strcmp(name, "prefetch")) *value = h->opt_prefetch;
    else if (!strcmp(name, "nt_loads")) *value = h->opt_nt;
    else if (!strcmp(name, "blocks_per_cu")) *value = h->opt_blocks_per_cu;
    else if (!strcmp(name, "variant")) *value = h->opt_variant;
    else if (!strcmp(name, "topk_bf16x3")) *value = h->opt_topk_bf16x3;
    else if (!strcmp(name, "topk_form")) *value = h->opt_topk_form;
    else if (!strcmp(name, "topk_prune")) *value = h->opt_topk_prune;
    else if (!strcmp(name, "topk_block")) *value = h->opt_topk_block;
    else if (!strcmp(name, "topk_block_users")) *value = h->topk_block_users;
    else if (!strcmp(name, "topk_refine")) *value = h->opt_topk_refine;
    else if (!strcmp(name, "topk_refined") || !strcmp(name, "topk_refine_repaired")) {
        int32_t c[2] = {0, 0};
        if (h->topk_refine_counter) {
            if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(c, h->topk_refine_counter, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess)
                return M2D_ERR_HIP;
        }
        *value = !strcmp(name, "topk_refined") ? c[0] : c[1];
    }
    else if (!strcmp(name, "topk_tiles_scanned") || !strcmp(name, "topk_tiles_full") || !strcmp(name, "topk_tiles_completed")) {
        // diagnostic (synchronises the device): 32-dish tiles the blocks of the last pipelined retrieval launch stepped
        // through, and what they would have stepped through without pattern pruning
        *value = 0;
        if (!strcmp(name, "topk_tiles_full")) *value = h->topk_tiles_full;
        else if (h->topk_tiles_counter) {
            unsigned long long v[2] = {0, 0};
            if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(v, h->topk_tiles_counter, sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
                return M2D_ERR_HIP;
            // (word 1: the hi x hi first form's count of (wave, tile) pairs whose cross products were multiplied; -1: another form ran)
            *value = !strcmp(name, "topk_tiles_completed") ? (h->topk_apx_last ? (int64_t)v[1] : -1) : (int64_t)v[0];
        }
        else if (!strcmp(name, "topk_tiles_completed")) *value = -1;
    }
    else if (!strcmp(name, "topk_grouped")) *value = h->opt_topk_grouped;
    else if (!strcmp(name, "mlp_bf16x3")) *value = h->opt_mlp_bf16x3;
    else if (!strcmp(name, "mlp_form")) *value = h->opt_mlp_form;
    else if (!strcmp(name, "skip_masked")) *value = h->opt_skip_masked;
    else if (!strcmp(name, "user_high_table")) *value = h->opt_user_high;
    else if (!strcmp(name, "host_zero_copy")) *value = h->opt_host_zero_copy;
    else if (!strcmp(name, "num_cu")) *value = h->num_cu;
    else if (!strcmp(name, "topk_repaired")) {
        // diagnostic (synchronises the device): users of the last pattern-grouped m2d_topk_users call that the tie repair re-ranked
        // over their patterns (a tie at the list's end; with the refinement on: three or more dishes that close)
        *value = 0;
        if (h->topk_tie_list) {
            int32_t c = 0;
            if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                hipMemcpy(&c, h->topk_tie_list, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess)
                return M2D_ERR_HIP;
            *value = c;
        }
    }
    else return M2D_ERR_INVALID_ARG;
    return M2D_OK;
}

}  // extern "C"
