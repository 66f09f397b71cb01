// ise_knn.hip -- brute-force L2 / inner-product kNN for MI355X (gfx950, CDNA4).
//
// Replaces the native work behind faiss.IndexFlatL2 / IndexFlatIP as the
// reference uses them (backend/engine.py:55, backend/utils.py:293-330,
// backend/kmeans_faiss.py:49, backend/siamese/test_index.py:54) and
// faiss.normalize_L2 (backend/utils.py:303).  C ABI: include/ise_knn.h.
//
// Data layout in HBM
//   xb     [cap][dp]  float32 or bf16 index rows, stored unshifted, zero padded to whole
//                     k-steps of 64 bytes; cap is a multiple of 16 rows
//   norms  [cap]      float32 |y - mu|^2 per row (mu = 0 for inner product / bf16)
//   mu     [dp]       float32 shift vector of float32 L2 indexes (column mean of the rows)
//   part   [nqt][nb][16 T][k] u64  per-block sorted candidate lists (workspace slot)
//
// Kernels
//   scan_kernel    ise_scan.hpp    one pass over the index per 16 T queries; HBM-bound
//   rerank / exact ise_exact.hpp   float32 L2: direct-difference re-rank + certificate (fused into
//                                  the merge), exact fallback scan for what it cannot prove
//   assign_kernel  ise_assign.hpp  k = 1 against a small index (centroids); MFMA-bound
//   merge_kernel   ise_merge.hpp   k-way merge of sorted per-block / per-rank lists
//   row helpers    ise_rows.hpp    norms, padding / bf16 conversion, normalize_L2, shift
//
// Candidate order: a 64-bit key = ord(score) << 32 | row id, where ord() is the
// order-preserving map float -> uint32 and score = squared L2 (or -inner product).
// Ascending key order is (score, id) order, which is the order Faiss reports (ties by
// ascending id), and keys are unique, so every selection is deterministic and
// independent of the grid shape.
//
// This file: the host side and the C ABI (include/ise_knn.h).

#include "ise_common.hpp"
#include "ise_scan_params.hpp"
#include "ise_assign.hpp"
#include "ise_exact.hpp"
#include "ise_merge.hpp"
#include "ise_exact_scan.hpp"
#include "ise_gemm_scan.hpp"
#include "ise_gemm_bf16.hpp"
#include "ise_rows.hpp"
#include "ise_short_scan.hpp"

// ---------------------------------------------------------------- host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int ise_fail_(int code, const std::string& msg) { return fail(code, msg); }  // for the other translation units
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? ISE_E_NOMEM : ISE_E_HIP,               \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

struct ise_index {
    int d = 0, dp = 0, metric = ISE_METRIC_L2, device = 0;
    int storage = ISE_STORE_F32;  // element type of xb
    long long n = 0, cap = 0;
    void* xb = nullptr;
    float* norms = nullptr;
    // float32 L2 indexes: shift vector mu [dp] = column mean of the rows, (re)computed lazily at the
    // first search after the index has grown by a quarter since the last time (or pinned by
    // ise_index_set_shift); norms [0, norms_rows) are |y - mu|^2 for the current mu.  mu only
    // decides how tight the scan's lower bounds are -- results are exact for any mu (ise_exact.hpp)
    float* mu = nullptr;
    bool shift_pinned = false;
    long long mu_rows = 0;     // rows mu was computed from (0 = not yet)
    long long norms_rows = 0;  // rows whose norm is valid
    unsigned long long* stats_dev = nullptr;  // [4]: reranked queries, exact-scan queries
    unsigned long long mu_updates = 0;
    unsigned long long gemm_chunks = 0;  // query chunks that took the large-batch path
    // workspaces (grown lazily, guarded by mu): NWS slots, so searches on different streams
    // may be in flight together.  A stream keeps the slot it used last (stream order is all
    // the ordering that needs); a stream without one takes a fresh slot, or the least
    // recently taken one behind an event wait.  Six slots on purpose: a server that issues
    // batches round-robin on more streams than that (bench.py: 16) gets at most six scans in
    // flight, chained slot to slot on the GPU, with the next ones already queued -- measured
    // best at 1M x 512 (303 us per batch against 328 with 4 streams and 319 with 16 slots)
    struct WorkSlot {
        u64* part = nullptr;
        size_t part_elems = 0;
        u64* keys_tmp = nullptr;  // multi-pass k scratch
        size_t keys_tmp_elems = 0;
        u64* xchg = nullptr;      // threshold-exchange entries of the scan kernel, tagged by xchg_seq
        size_t xchg_elems = 0;
        uint32_t xchg_seq = 0;    // bumped per scan launch: entries of older launches never match
        // large-batch path (ise_gemm_scan.hpp): one allocation holding qprep | xn | tau | ccnt | sample keys | cand
        char* gemm = nullptr;
        size_t gemm_bytes = 0;
        u64* fl_state = nullptr;  // exact path: launch seq << 32 | number of queries on the fallback list
        int* fl_list = nullptr;   // [fl_elems] the listed queries
        size_t fl_elems = 0;
        uint32_t fl_seq = 0;      // bumped per rerank launch, never reset while fl_state lives
        hipEvent_t done = nullptr;
        bool used = false;
        hipStream_t last_stream = nullptr;  // valid when used
    };
    static constexpr int NWS = 6;
    WorkSlot ws[NWS];
    unsigned ws_next = 0;
    hipStream_t stream = nullptr;  // add / reconstruct / shift maintenance
    // host-API search contexts: a caller owns one for the duration of its call
    struct HostCtx {
        hipStream_t stream = nullptr;
        float* q_dev = nullptr;  size_t q_elems = 0;
        float* D_dev = nullptr;  long long* I_dev = nullptr;  size_t out_elems = 0;
        // page-locked staging of a combined batch (queries in, results out)
        float* q_pin = nullptr;  size_t q_pin_elems = 0;
        float* D_pin = nullptr;  long long* I_pin = nullptr;  size_t out_pin_elems = 0;
        float* D_pin_dev = nullptr;  long long* I_pin_dev = nullptr;  // the same buffers as the device addresses them
        bool busy = false;
    };
    static constexpr int NHC = 4;
    HostCtx hc[NHC];
    std::mutex hc_mu;
    std::condition_variable hc_cv;
    // combining of concurrent host searches (ise_index_search_host): callers queue their request; one
    // of them -- at most CQ_LEADERS at a time -- takes the requests at the head of the queue that ask
    // for the same k, runs them as ONE batch and hands the results out
    struct HostReq {
        const float* q; long long nq; int k; float* D; long long* I;
        int rc = 0; std::string err; bool done = false;
        std::condition_variable cv;
    };
    static constexpr int CQ_LEADERS = 2;
    std::mutex cq_mu;
    std::deque<HostReq*> cq;
    int cq_leaders = 0;
    unsigned long long cq_batches = 0, cq_requests = 0;
    unsigned long long direct_queries = 0;  // queries answered by the direct small-batch scan (under mu_)
    unsigned long long short_batches = 0;   // batches whose scan was the short-index kernel (under mu_)
    int num_cu = 256;
    std::mutex mu_;
};

// rows are padded to whole k-steps of 64 bytes (16 floats / 32 bf16); rows longer than
// 4 steps to a multiple of 4 steps so that wider register chunks divide them
static int elem_size(int storage) { return storage == ISE_STORE_BF16 ? 2 : 4; }
static int pad_dim(int d, int storage) {
    const int per_step = 64 / elem_size(storage);
    const int steps = (d + per_step - 1) / per_step;
    return (steps > 4 ? (steps + 3) / 4 * 4 : steps) * per_step;
}
static size_t row_bytes(const ise_index* h) { return (size_t)h->dp * elem_size(h->storage); }
static int chunk_steps(const ise_index* h) {
    const int steps = (int)(row_bytes(h) / 64);
    for (int ch = 8; ch > 1; ch >>= 1)
        if (steps % ch == 0) return ch;
    return 1;
}
// LDS query row stride in 4-byte units: (stride/4) % 16 == 2 makes the 16 rows x 4 k-groups
// ds_read_b128 pattern bank-conflict-free
static int qs_stride_for(const ise_index* h) {
    const int units = (int)(row_bytes(h) / 4);
    const int pad = ((2 - (units / 4)) % 16 + 16) % 16 * 4;
    return units + pad;
}
#define KPASS_MAX 36 /* most keys per query one scan pass selects: k = 32 with the exact path's 4 spare candidates still is
                        ONE pass (k = 29..32 took two: 730 us instead of 355 at 1M x 512); more: floor-keyed passes */
#define XPASS_MAX 32 /* most results per query of one exact-scan pass, of the direct scan and of the large-batch paths */
static size_t scan_lds_bytes(const ise_index* h, int waves, int T, int kb) {
    return scan_lds_layout(qs_stride_for(h), waves, T, kb);
}

extern "C" int ise_version(void) { return 100; }
extern "C" const char* ise_last_error(void) { return g_err.c_str(); }

extern "C" int ise_device_count(int* count) {
    if (!count) return fail(ISE_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(ISE_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return ISE_OK;
}

extern "C" int ise_device_arch(int device, char* buf, int buflen) {
    if (!buf || buflen <= 0) return fail(ISE_E_INVALID, "buf is NULL");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, (size_t)buflen, "%s", prop.gcnArchName);
    return ISE_OK;
}

extern "C" int ise_index_create(ise_index_t** out, int d, int metric, int device) {
    return ise_index_create_ex(out, d, metric, device, ISE_STORE_F32);
}

extern "C" int ise_index_create_ex(ise_index_t** out, int d, int metric, int device, int storage) {
    if (!out) return fail(ISE_E_INVALID, "out is NULL");
    *out = nullptr;
    if (d <= 0) return fail(ISE_E_INVALID, "d must be positive");
    if (storage != ISE_STORE_F32 && storage != ISE_STORE_BF16)
        return fail(ISE_E_INVALID, "storage must be ISE_STORE_F32 or ISE_STORE_BF16");
    if (metric != ISE_METRIC_L2 && metric != ISE_METRIC_INNER_PRODUCT)
        return fail(ISE_E_INVALID, "metric must be ISE_METRIC_L2 or ISE_METRIC_INNER_PRODUCT");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ISE_E_NODEVICE, "no HIP device visible: the kNN path needs an MI355X (gfx950) GPU");
    if (device < 0 || device >= ndev) return fail(ISE_E_INVALID, "device out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ISE_E_NODEVICE, std::string("device is ") + prop.gcnArchName +
                                        ", this library is built for gfx950 only");
    ise_index* h = new (std::nothrow) ise_index();
    if (!h) return fail(ISE_E_NOMEM, "host allocation failed");
    h->d = d;
    h->storage = storage;
    h->dp = pad_dim(d, storage);
    h->metric = metric;
    h->device = device;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    DeviceGuard gd(device);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&h->mu, (size_t)h->dp * sizeof(float));
    if (e == hipSuccess) e = hipMemset(h->mu, 0, (size_t)h->dp * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&h->stats_dev, 32 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->stats_dev, 0, 32 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        if (h->stream) (void)hipStreamDestroy(h->stream);
        if (h->mu) (void)hipFree(h->mu);
        if (h->stats_dev) (void)hipFree(h->stats_dev);
        delete h;
        return fail(ISE_E_HIP, std::string("index setup: ") + hipGetErrorString(e));
    }
    *out = h;
    return ISE_OK;
}

static void free_all(ise_index* h) {
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    for (auto& w : h->ws) {
        if (w.part) (void)hipFree(w.part);
        if (w.keys_tmp) (void)hipFree(w.keys_tmp);
        if (w.xchg) (void)hipFree(w.xchg);
        if (w.gemm) (void)hipFree(w.gemm);
        if (w.fl_state) (void)hipFree(w.fl_state);
        if (w.fl_list) (void)hipFree(w.fl_list);
        if (w.done) (void)hipEventDestroy(w.done);
        w = ise_index::WorkSlot();
    }
    for (auto& c : h->hc) {
        if (c.q_dev) (void)hipFree(c.q_dev);
        if (c.D_dev) (void)hipFree(c.D_dev);
        if (c.I_dev) (void)hipFree(c.I_dev);
        if (c.q_pin) (void)hipHostFree(c.q_pin);
        if (c.D_pin) (void)hipHostFree(c.D_pin);
        if (c.I_pin) (void)hipHostFree(c.I_pin);
        if (c.stream) (void)hipStreamDestroy(c.stream);
        c = ise_index::HostCtx();
    }
    h->xb = h->norms = nullptr;
    h->n = h->cap = 0;
}

extern "C" int ise_index_destroy(ise_index_t* h) {
    if (!h) return ISE_OK;
    {
        DeviceGuard gd(h->device);
        (void)hipDeviceSynchronize();
        free_all(h);
        if (h->mu) (void)hipFree(h->mu);
        if (h->stats_dev) (void)hipFree(h->stats_dev);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return ISE_OK;
}

extern "C" int ise_index_reset(ise_index_t* h) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    HIP_TRY(hipDeviceSynchronize());
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    h->xb = h->norms = nullptr;
    h->n = h->cap = 0;
    h->shift_pinned = false;
    h->mu_rows = h->norms_rows = 0;
    HIP_TRY(hipMemset(h->mu, 0, (size_t)h->dp * sizeof(float)));
    return ISE_OK;
}

extern "C" int ise_index_info(const ise_index_t* h, int* d, int* metric, int64_t* ntotal, int* device) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (d) *d = h->d;
    if (metric) *metric = h->metric;
    if (ntotal) *ntotal = h->n;
    if (device) *device = h->device;
    return ISE_OK;
}

// grow storage to hold at least `need` rows (capacity a multiple of 16 rows,
// pad rows zeroed so a partial last tile reads zeros)
static int reserve_rows(ise_index* h, long long need, hipStream_t st) {
    if (need <= h->cap) return ISE_OK;
    long long cap = h->cap ? h->cap : 0;
    long long want = need;
    if (cap > 0 && want < cap + cap / 2) want = cap + cap / 2;  // geometric growth on re-add
    want = (want + 15) / 16 * 16;
    const size_t rb = row_bytes(h);
    char* nx = nullptr;
    float* nn = nullptr;
    HIP_TRY(hipMalloc(&nx, (size_t)want * rb));
    hipError_t e = hipMalloc(&nn, (size_t)want * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(nx);
        return fail(ISE_E_NOMEM, std::string("hipMalloc(norms): ") + hipGetErrorString(e));
    }
    if (h->n > 0) {
        HIP_TRY(hipMemcpyAsync(nx, h->xb, (size_t)h->n * rb, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(nn, h->norms, (size_t)h->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(nx + (size_t)h->n * rb, 0, (size_t)(want - h->n) * rb, st));
    HIP_TRY(hipMemsetAsync(nn + h->n, 0, (size_t)(want - h->n) * sizeof(float), st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h->xb) HIP_TRY(hipDeviceSynchronize());  // searches in flight on other streams still read the old storage
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    h->xb = nx;
    h->norms = nn;
    h->cap = want;
    return ISE_OK;
}

// true when float32 rows can be copied verbatim into the index layout
static bool rows_copy_verbatim(const ise_index* h) { return h->storage == ISE_STORE_F32 && h->dp == h->d; }

static bool uses_shift(const ise_index* h) { return h->storage == ISE_STORE_F32 && h->metric == ISE_METRIC_L2; }

// float32 L2 indexes: bring mu and the norms up to date with the rows (called with the handle
// locked, before a search / assignment reads them).  mu = column mean of ALL rows, recomputed when
// the index has grown by a quarter since it was last taken (amortised O(1) per row); a new mu
// means new norms for every row.  Rare and blocking: other streams' searches read mu and norms.
static int prepare_shift_locked(ise_index* h, hipStream_t st);

static void launch_norms(ise_index* h, long long row0, long long n, hipStream_t st) {
    const long long nblk = (n + 3) / 4;  // n < 2^32 so nblk fits the 32-bit grid
    if (h->storage == ISE_STORE_BF16)
        hipLaunchKernelGGL(norms_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const __bf16*)h->xb, row0, n,
                           h->dp, h->norms);
    else
        hipLaunchKernelGGL(norms_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)h->xb, row0, n, h->dp,
                           uses_shift(h) ? h->mu : (const float*)nullptr, h->norms);
}

static int prepare_shift_locked(ise_index* h, hipStream_t st) {
    if (!uses_shift(h) || h->n == 0) return ISE_OK;
    const bool need_mu = !h->shift_pinned && (h->mu_rows == 0 || h->n >= h->mu_rows + h->mu_rows / 4 + 1);
    if (!need_mu && h->norms_rows == h->n) return ISE_OK;
    HIP_TRY(hipDeviceSynchronize());  // nothing in flight reads mu / norms while they change
    if (need_mu) {
        const int groups = (int)std::max<long long>(1, std::min<long long>(COLMEAN_GROUPS_MAX, h->n / 64));
        float* partial = nullptr;
        HIP_TRY(hipMalloc(&partial, (size_t)groups * h->dp * sizeof(float)));
        hipLaunchKernelGGL(col_sum_kernel, dim3((h->dp + 255) / 256, groups), dim3(256), 0, st, (const float*)h->xb,
                           h->n, h->d, h->dp, groups, partial);
        hipLaunchKernelGGL(col_mean_kernel, dim3((h->dp + 255) / 256), dim3(256), 0, st, partial, h->n, h->d, h->dp,
                           groups, h->mu);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // partial is freed below
        (void)hipFree(partial);
        if (e != hipSuccess) return fail(ISE_E_HIP, std::string("shift vector: ") + hipGetErrorString(e));
        h->mu_rows = h->n;
        h->norms_rows = 0;
        h->mu_updates++;
    }
    launch_norms(h, h->norms_rows, h->n - h->norms_rows, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    h->norms_rows = h->n;
    return ISE_OK;
}

// n rows have just been written behind row h->n: their norms are taken now when they can be
// (no shift, or a shift that is already fixed), otherwise with the shift at the next search
static void norms_after_add(ise_index* h, long long n, hipStream_t st) {
    if (uses_shift(h) && !h->shift_pinned && h->mu_rows == 0) return;
    if (h->norms_rows != h->n) return;  // earlier rows are still waiting for the shift
    launch_norms(h, h->n, n, st);
    h->norms_rows = h->n + n;
}

static int add_device_locked(ise_index* h, const float* x_dev, long long n, hipStream_t st) {
    if (n == 0) return ISE_OK;
    if (h->n + n >= (1ll << 32)) return fail(ISE_E_INVALID, "index would exceed 2^32 - 1 rows");
    int rc = reserve_rows(h, h->n + n, st);
    if (rc) return rc;
    char* dst = static_cast<char*>(h->xb) + (size_t)h->n * row_bytes(h);
    if (rows_copy_verbatim(h)) {
        HIP_TRY(hipMemcpyAsync(dst, x_dev, (size_t)n * h->d * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        const long long total = n * h->dp;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
        if (h->storage == ISE_STORE_BF16)
            hipLaunchKernelGGL(pad_rows_bf16_kernel, dim3(blocks), dim3(256), 0, st, x_dev, n, h->d, (__bf16*)dst, h->dp);
        else
            hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, st, x_dev, n, h->d, (float*)dst, h->dp);
        HIP_TRY(hipGetLastError());
    }
    norms_after_add(h, n, st);
    HIP_TRY(hipGetLastError());
    h->n += n;
    return ISE_OK;
}

extern "C" int ise_index_set_shift(ise_index_t* h, const float* mu_host) {
    if (!h || !mu_host) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    if (!uses_shift(h)) return ISE_OK;  // only float32 L2 indexes are shifted
    DeviceGuard gd(h->device);
    HIP_TRY(hipDeviceSynchronize());  // searches in flight read mu
    HIP_TRY(hipMemset(h->mu, 0, (size_t)h->dp * sizeof(float)));
    HIP_TRY(hipMemcpy(h->mu, mu_host, (size_t)h->d * sizeof(float), hipMemcpyHostToDevice));
    h->shift_pinned = true;
    h->norms_rows = 0;  // every norm is retaken around the new shift at the next search
    return ISE_OK;
}

extern "C" int ise_index_get_shift(ise_index_t* h, float* mu_host) {
    if (!h || !mu_host) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    int rc = prepare_shift_locked(h, h->stream);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(mu_host, h->mu, (size_t)h->d * sizeof(float), hipMemcpyDeviceToHost));
    return ISE_OK;
}

extern "C" int ise_index_add_device(ise_index_t* h, const float* x_dev, int64_t n, void* stream) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && !x_dev)) return fail(ISE_E_INVALID, "bad rows argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    return add_device_locked(h, x_dev, n, (hipStream_t)stream);
}

extern "C" int ise_index_add_host(ise_index_t* h, const float* x, int64_t n) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && !x)) return fail(ISE_E_INVALID, "bad rows argument");
    if (n == 0) return ISE_OK;
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    if (h->n + n >= (1ll << 32)) return fail(ISE_E_INVALID, "index would exceed 2^32 - 1 rows");
    int rc = reserve_rows(h, h->n + n, h->stream);
    if (rc) return rc;
    // upload in slabs; a device staging buffer is needed when rows are padded or converted
    const long long slab = std::max<long long>(1, (256ll << 20) / ((long long)h->d * 4));
    float* tmp = nullptr;
    if (!rows_copy_verbatim(h)) HIP_TRY(hipMalloc(&tmp, (size_t)std::min<long long>(slab, n) * h->d * sizeof(float)));
    for (long long i0 = 0; i0 < n; i0 += slab) {
        const long long m = std::min<long long>(slab, n - i0);
        if (rows_copy_verbatim(h)) {
            char* dst = static_cast<char*>(h->xb) + (size_t)h->n * row_bytes(h);
            hipError_t e = hipMemcpyAsync(dst, x + (size_t)i0 * h->d, (size_t)m * h->d * sizeof(float),
                                          hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) return fail(ISE_E_HIP, std::string("H2D: ") + hipGetErrorString(e));
            norms_after_add(h, m, h->stream);
            h->n += m;
        } else {
            hipError_t e = hipMemcpyAsync(tmp, x + (size_t)i0 * h->d, (size_t)m * h->d * sizeof(float),
                                          hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) {
                (void)hipFree(tmp);
                return fail(ISE_E_HIP, std::string("H2D: ") + hipGetErrorString(e));
            }
            rc = add_device_locked(h, tmp, m, h->stream);
            if (rc) {
                (void)hipFree(tmp);
                return rc;
            }
        }
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            if (tmp) (void)hipFree(tmp);
            return fail(ISE_E_HIP, std::string("add sync: ") + hipGetErrorString(e));
        }
    }
    if (tmp) (void)hipFree(tmp);
    return ISE_OK;
}

extern "C" int ise_index_reconstruct_host(ise_index_t* h, int64_t i0, int64_t n, float* out) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu_);
    if (i0 < 0 || n < 0 || i0 + n > h->n || (n > 0 && !out)) return fail(ISE_E_INVALID, "row range out of bounds");
    if (n == 0) return ISE_OK;
    DeviceGuard gd(h->device);
    const char* src = static_cast<const char*>(h->xb) + (size_t)i0 * row_bytes(h);
    if (h->storage == ISE_STORE_F32) {
        HIP_TRY(hipMemcpy2DAsync(out, (size_t)h->d * 4, src, row_bytes(h), (size_t)h->d * 4, (size_t)n,
                                 hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return ISE_OK;
    }
    // bf16 rows come back as the float32 values they hold, in slabs through a device buffer
    const long long slab = std::max<long long>(1, (64ll << 20) / ((long long)h->d * 4));
    float* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)std::min<long long>(slab, n) * h->d * sizeof(float)));
    int rc = ISE_OK;
    for (long long r0 = 0; r0 < n && rc == ISE_OK; r0 += slab) {
        const long long m = std::min<long long>(slab, n - r0);
        const long long total = m * h->d;
        hipLaunchKernelGGL(unpack_rows_bf16_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)),
                           dim3(256), 0, h->stream, (const __bf16*)(src + (size_t)r0 * row_bytes(h)), m, h->d, h->dp, tmp);
        hipError_t e = hipMemcpyAsync(out + (size_t)r0 * h->d, tmp, (size_t)total * sizeof(float), hipMemcpyDeviceToHost,
                                      h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = fail(ISE_E_HIP, std::string("reconstruct: ") + hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    return rc;
}

// ---- search
static void launch_scan(const ise_index* h, int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st,
                        const ScanParams& sp) {
    if (h->storage == ISE_STORE_BF16) ise_launch_scan_bf16(ch, waves, T, grid, lds, st, sp);
    else if (uses_shift(h)) ise_launch_scan_f32_shift(ch, waves, T, grid, lds, st, sp);
    else ise_launch_scan_f32_plain(ch, waves, T, grid, lds, st, sp);
}

static void launch_short(const ise_index* h, int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st,
                         const ScanParams& sp, const ShortParams& tp) {
    if (h->storage == ISE_STORE_BF16) ise_launch_short_bf16(ch, waves, bpc, grid, lds, st, sp, tp);
    else if (uses_shift(h)) ise_launch_short_f32_shift(ch, waves, bpc, grid, lds, st, sp, tp);
    else ise_launch_short_f32_plain(ch, waves, bpc, grid, lds, st, sp, tp);
}

struct ScanPlan {
    int nblocks, tiles_total, tiles_per_block, nqt, ch, kpass, kb, waves, T;
    size_t lds;
    bool exact;  // float32 L2: the scan is the filter of the exact search (ise_exact.hpp)
    int kc;      // keys per query the scan + merge stage selects: k, or k + extra candidates when exact
    bool gemm;   // the batch takes the large-batch path (ise_gemm_scan.hpp): the slot also holds its buffers
    size_t gemm_bytes;
    bool short_;  // the batch's scan is the short-index kernel (ise_short_scan.hpp)
    int short_bpc;  // ... with this many blocks per CU
};

// candidates kept beyond k on the exact path: enough that the certificate holds on data whose
// neighbour spacing exceeds the bound's width.  4 keeps k + extra below the 16-slot block lists at
// k <= 11, so that the boot's windowed cut has a window ([kc, kb]) instead of one exact rank
// (overridable for experiments: $ISE_EXACT_EXTRA)
static int exact_extra(int k) {
    static const int forced = [] { const char* e = getenv("ISE_EXACT_EXTRA"); return e ? atoi(e) : 0; }();
    (void)k;
    return forced > 0 && forced <= 16 ? forced : 4;
}
// relative width of the scan's lower bound: every rounding between the stored floats and the keyed
// value, in units of u = 2^-24 times (|x-mu|^2 + |y-mu|^2) (derivation: DESIGN.md section 4.1)
static float exact_beta(const ise_index* h) { return (0.5625f * h->dp + 256.f) * 5.9604645e-8f * 1.02f; }
// Test / rehearsal knobs that may change while the process runs: read from the environment when the library
// is first used and again whenever ise_refresh_env_knobs() is called (the tests call it after changing the
// environment) -- never inside a search, where another thread's setenv would race with getenv.
struct EnvKnobs {
    std::atomic<int> force_exact{0};    // ISE_FORCE_EXACT=1: fail every certificate (exercises the exact scan)
    std::atomic<int> no_direct{0};      // ISE_NO_DIRECT=1: one-query batches take the filtered path as well
    std::atomic<int> no_short{0};       // ISE_NO_SHORT=1: short indexes take the streaming kernel + merge launches
    std::atomic<int> short_tpb_max{0};  // ISE_SHORT_TPB_MAX: most row tiles per block the short-index kernel takes
    std::atomic<int> direct_short_max_tiles{0};  // ISE_DIRECT_SHORT_MAX_TILES: longest SHORT index (16-row tiles) whose one-query batches take the direct scan
    void refresh() {
        auto flag = [](const char* name) { const char* e = getenv(name); return (e && e[0] == '1') ? 1 : 0; };
        auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
        force_exact.store(flag("ISE_FORCE_EXACT"));
        no_direct.store(flag("ISE_NO_DIRECT"));
        no_short.store(flag("ISE_NO_SHORT"));
        short_tpb_max.store(num("ISE_SHORT_TPB_MAX"));
        direct_short_max_tiles.store(num("ISE_DIRECT_SHORT_MAX_TILES"));
    }
};
static EnvKnobs& knobs() {
    static EnvKnobs k;
    static std::once_flag once;
    std::call_once(once, [] { k.refresh(); });
    return k;
}
extern "C" int ise_refresh_env_knobs(void) {
    knobs().refresh();
    return ISE_OK;
}
static bool force_exact() { return knobs().force_exact.load(std::memory_order_relaxed) != 0; }

static bool xchg_enabled() {  // dev knob: ISE_NO_XCHG=1 switches the threshold exchange off
    static const bool on = [] { const char* e = getenv("ISE_NO_XCHG"); return !(e && e[0] == '1'); }();
    return on;
}

// pick (query tiles per pass T, waves per block) for nq queries: the largest T <= 3
// that the batch can use and whose LDS image fits, preferring 8 waves
static int make_plan(const ise_index* h, long long nq, int k, ScanPlan* pl, bool allow_short = true) {
    pl->exact = uses_shift(h);
    pl->kc = pl->exact ? k + exact_extra(k) : k;
    pl->kpass = pl->kc < KPASS_MAX ? pl->kc : KPASS_MAX;
    pl->kb = pl->kpass <= 16 ? 16 : (pl->kpass <= 32 ? 32 : KB_MAX);
    pl->ch = chunk_steps(h);
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_CH")) {  // dev: force a smaller chunk (must divide dp/16)
        const int ch = atoi(e);
        if ((ch == 1 || ch == 2 || ch == 4 || ch == 8) && (int)(row_bytes(h) / 64) % ch == 0) pl->ch = ch;
    }
#endif
    // relative time of one pass over the index with T query tiles (measured, 1M x 512)
    // (fp32: T = 3 is MFMA-bound; bf16 rows stay HBM-bound, the growth is top-k bookkeeping)
    static const double pass_cost_f32[5] = {0.0, 1.0, 1.11, 1.45, 0.0};
    static const double pass_cost_bf16[5] = {0.0, 1.0, 1.07, 1.16, 1.25};
    const bool bf16 = h->storage == ISE_STORE_BF16;
    const double* pass_cost = bf16 ? pass_cost_bf16 : pass_cost_f32;
    int tmax = bf16 ? 4 : 3;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_TMAX")) tmax = std::min(tmax, std::max(1, atoi(e)));
#endif
    pl->T = 0;
    double best = 0;
    for (int t = 1; t <= tmax; t++) {
        int wv = 0;
        size_t lds = 0;
        // two tiles: 8 waves with the threshold exchange when the index is long enough for it to run and
        // pay (>= 6 row tiles per wave: 383 vs 399 us at 1M x 512); else one 16-wave block per CU, whose
        // extra waves hide the bookkeeping instead (71 vs 85 us at 125k rows)
        // (fp32 rows only: bf16 rows stay HBM-bound and run 208 vs 217 us with the 16-wave block)
        const bool xchg_pays = !bf16 && xchg_enabled() && (h->n + 15) / 16 >= 6ll * 8 * h->num_cu;
        if (t == 2 && !xchg_pays && scan_lds_bytes(h, 16, 2, pl->kb) <= LDS_LIMIT) {
            wv = 16;
            lds = scan_lds_bytes(h, 16, 2, pl->kb);
        }
        for (int cand_w = 8; cand_w >= (t == 1 ? 4 : 8) && !wv; cand_w -= 4) {  // 4 waves: one tile only
            lds = scan_lds_bytes(h, cand_w, t, pl->kb);
            if (lds <= LDS_LIMIT) wv = cand_w;
        }
        if (!wv) break;
        const double cost = (double)((nq + 16 * t - 1) / (16 * t)) * pass_cost[t];
        if (!pl->T || cost < best - 1e-9) {
            pl->T = t;
            pl->waves = wv;
            pl->lds = lds;
            best = cost;
        }
    }
    if (!pl->T) return fail(ISE_E_INVALID, "d too large: a 16-query tile must fit the 160 KiB LDS "
                                              "(float32 rows: d <= 2240, bf16 rows: d <= 4480)");
    // one query tile runs 16 waves per CU at <= 128 VGPRs: 4-step chunks (2 x 4 KB in flight per
    // wave) measured faster than 8-step ones there (no spills, more waves' worth of loads)
    if ((pl->T == 1 || pl->waves == 16) && pl->ch > 4) pl->ch = 4;  // both run at <= 128 VGPRs
    pl->tiles_total = (int)((h->n + 15) / 16);
    int blocks_per_cu = (pl->T == 1 && pl->waves <= 8 && pl->lds <= LDS_LIMIT / 2) ? 2 : 1;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_PLAN")) {  // dev: "waves,blocks_per_cu" (T = 1 only)
        int wv = 8, bpc = 2;
        if (pl->T == 1 && sscanf(e, "%d,%d", &wv, &bpc) == 2 && (wv == 4 || wv == 8) && bpc >= 1) {
            pl->waves = wv;
            pl->lds = scan_lds_bytes(h, wv, 1, pl->kb);
            blocks_per_cu = bpc;
        }
    }
#endif
    // Blocks: one per `waves` row tiles (a tile per wave) on long indexes.  A SMALL index is spread further -- down
    // to ~64 KB of rows per block (one tile of rows > 2 KB), as long as every block of the launch is resident at
    // once: the row stream is bound per CU (~30 GB/s), and eight one-tile waves of a 1000 x 2048 index on each of
    // 8 CUs streamed 33 us where 63 CUs take 8 (the reference's own index size and default metric: 59 -> 34 us
    // per one-query batch; 1000 x 512, 32 queries: 56 -> 45 us).  More blocks than CUs x blocks_per_cu would run
    // in rounds: batches of several query tiles (grid.y) divide the budget.
    const bool long_rows = row_bytes(h) > 2048;
    const int nqt_plan = (int)((nq + 16 * pl->T - 1) / (16 * pl->T));
    const int min_tiles = (int)std::max<size_t>(1, 65536 / (16 * row_bytes(h)));
    const int slots = h->num_cu * blocks_per_cu;
    const int nb_packed = std::min(slots, (pl->tiles_total + pl->waves - 1) / pl->waves);
    const int nb_spread = std::min(std::max(1, slots / nqt_plan), (pl->tiles_total + min_tiles - 1) / min_tiles);
    int nb = std::max(nb_packed, nb_spread);
    if (nb < 1) nb = 1;
    if (nb > MERGE_LISTS_MAX) nb = MERGE_LISTS_MAX;
    pl->tiles_per_block = (pl->tiles_total + nb - 1) / nb;
    if (pl->tiles_per_block < 1) pl->tiles_per_block = 1;
    pl->nblocks = (pl->tiles_total + pl->tiles_per_block - 1) / pl->tiles_per_block;
    if (pl->nblocks < 1) pl->nblocks = 1;
    // ... and a wave that owns ONE tile of long rows is latency-bound on its own loads: 8-step chunks (2 x 8 KB in
    // flight) where the register budget has them (8-wave blocks)
    if (long_rows && pl->T == 1 && pl->waves <= 8 && pl->tiles_per_block <= pl->waves && chunk_steps(h) == 8) pl->ch = 8;
    pl->nqt = (int)((nq + 16 * pl->T - 1) / (16 * pl->T));
    pl->gemm = false;
    pl->gemm_bytes = 0;
    pl->short_ = false;
    pl->short_bpc = 0;
    // Short indexes, one query tile, one pass: short_scan_kernel (scores dumped to LDS, one selection per block;
    // ise_short_scan.hpp) writes the per-block lists instead of scan_kernel.  Two 8-wave blocks per CU when their
    // LDS images fit side by side, else one; a block's rows must fit the selection (SHORT_TPB_MAX tiles).
    // 17 .. 64 queries: the same kernel with TWO query tiles per pass (twice the MFMA work per row tile, still under
    // the row stream for float32 at d = 512), 33 .. 64 as two such passes side by side (grid.y)
    static const bool short_t2 = [] { const char* e = getenv("ISE_SHORT_T2"); return !(e && e[0] == '0'); }();
    const int short_T = nq <= 16 ? 1 : 2;
    if (allow_short && (nq <= 16 || (short_t2 && nq <= 64)) && pl->kc <= pl->kpass && h->n > 0 &&
        !knobs().no_short.load(std::memory_order_relaxed)) {
        int tpb_max = knobs().short_tpb_max.load(std::memory_order_relaxed);
        if (tpb_max <= 0 || tpb_max > SHORT_TPB_MAX) tpb_max = SHORT_TPB_MAX;
        const int S = qs_stride_for(h);
        // shapes tried in order: one 16-wave block per CU (the queries are staged once per CU, one query per
        // wave in the selection, half the lists for the merge: 36.3 us per step at 100k x 512 against 42.0 with two
        // 8-wave blocks per CU and 44.2 with three), then two 8-wave blocks per CU (up to 262k rows).  The row tiles
        // are split evenly over the blocks: the stream is bound per CU, so every CU gets the same bytes (within one
        // tile).  Rows of more than 2 KB keep the streaming kernel (and the direct scan for one query): staging a
        // 16-query tile of such rows per block costs more than the bookkeeping it saves (30k x 1024: 77 against
        // 68 us per batch of 16; 100k x 2048, the reference's own descriptor size: 318 against 245, and 291 against
        // 166 us for one query, scripts/long_rows_probe.py).
        static const int shapes[2][3] = {{16, 1, 1}, {8, 2, 1}};  // waves, blocks per CU, rounds over the CUs
        int first = row_bytes(h) <= 2048 ? 0 : 2;
#ifdef ISE_ABLATE
        if (const char* e = getenv("ISE_SHORT_SHAPE")) first = std::max(first, std::min(1, atoi(e)));  // dev: skip shapes
#endif
        for (int si = first; si < 2 && !pl->short_; si++) {
            const int wv = shapes[si][0], bpc = shapes[si][1];
            const int short_nqt = (int)((nq + 16 * short_T - 1) / (16 * short_T));
            const int packed = (pl->tiles_total + wv - 1) / wv;
            // as above: >= ~64 KB of rows per block, the whole launch resident at once
            const int spread = std::min(std::max(1, h->num_cu * bpc / short_nqt), (pl->tiles_total + min_tiles - 1) / min_tiles);
            int nbs = std::max(1, std::min(h->num_cu * bpc * shapes[si][2], std::max(packed, spread)));
            nbs = std::min(nbs, MERGE_LISTS_MAX);
            const int tpb = (pl->tiles_total + nbs - 1) / nbs;
            const size_t lds = short_lds_layout(S, tpb, wv, short_T);
            if (tpb <= tpb_max && lds <= (size_t)LDS_LIMIT / bpc) {
                pl->short_ = true;
                pl->nblocks = nbs;
                pl->tiles_per_block = tpb;
                pl->lds = lds;
                pl->waves = wv;
                pl->short_bpc = bpc;
                pl->T = short_T;
                pl->nqt = short_nqt;
                pl->ch = std::min(chunk_steps(h), 4);
            }
        }
    }
    return ISE_OK;
}

// workspace of one slot: part [nqt][nb][16 T][kpass]; for kc > kpass additionally
// keys_tmp = keys_all [nq][kc] | floor [nq] | pass_keys [nq][kpass] | exact floor [nq] | exact pass keys [nq][32];
// exact path: the fallback list
static int ensure_workspace(ise_index::WorkSlot* w, const ScanPlan& pl, long long nq, bool* changed) {
    if (!w->done) HIP_TRY(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    // at least what the direct one-query scan can ask for (direct_applies): MERGE_LISTS_MAX lists of XPASS_MAX keys
    const size_t need = std::max<size_t>((size_t)pl.nqt * pl.nblocks * (16 * pl.T) * pl.kpass, (size_t)MERGE_LISTS_MAX * XPASS_MAX);
    if (need > w->part_elems) {
        if (w->part) (void)hipFree(w->part);  // hipFree waits for outstanding work
        w->part = nullptr;
        w->part_elems = 0;
        HIP_TRY(hipMalloc(&w->part, need * sizeof(u64)));
        w->part_elems = need;
        *changed = true;
    }
    const size_t needx = (size_t)pl.nqt * (16 * pl.T) * pl.nblocks;
    if (needx > w->xchg_elems) {
        if (w->xchg) (void)hipFree(w->xchg);
        w->xchg = nullptr;
        w->xchg_elems = 0;
        HIP_TRY(hipMalloc(&w->xchg, needx * sizeof(u64)));
        HIP_TRY(hipMemset(w->xchg, 0xFF, needx * sizeof(u64)));  // tag 0xFFFFFFFF is never issued
        w->xchg_elems = needx;
        w->xchg_seq = 0;
        if (const char* e = getenv("ISE_XCHG_SEQ_START"))  // test knob: start near the tag wrap
            w->xchg_seq = (uint32_t)strtoul(e, nullptr, 0);
        *changed = true;
    }
    if (pl.kc > pl.kpass) {
        const size_t need2 = (size_t)nq * ((size_t)pl.kc + 2 + pl.kpass + XPASS_MAX);
        if (need2 > w->keys_tmp_elems) {
            if (w->keys_tmp) (void)hipFree(w->keys_tmp);
            w->keys_tmp = nullptr;
            w->keys_tmp_elems = 0;
            HIP_TRY(hipMalloc(&w->keys_tmp, need2 * sizeof(u64)));
            w->keys_tmp_elems = need2;
            *changed = true;
        }
    }
    if (pl.gemm) {
        const size_t needg = pl.gemm_bytes;
        if (needg > w->gemm_bytes) {
            if (w->gemm) (void)hipFree(w->gemm);
            w->gemm = nullptr;
            w->gemm_bytes = 0;
            HIP_TRY(hipMalloc(&w->gemm, needg));
            w->gemm_bytes = needg;
            *changed = true;
        }
    }
    if (pl.exact) {
        if (!w->fl_state) {
            HIP_TRY(hipMalloc(&w->fl_state, 2 * sizeof(u64)));  // [0] the list's state, [1] the exact scan's arrival counter
            HIP_TRY(hipMemset(w->fl_state, 0, 2 * sizeof(u64)));  // tag 0 is never issued
            w->fl_seq = 0;
            if (const char* e = getenv("ISE_XCHG_SEQ_START")) w->fl_seq = (uint32_t)strtoul(e, nullptr, 0);
            *changed = true;
        }
        if ((size_t)nq > w->fl_elems) {
            if (w->fl_list) (void)hipFree(w->fl_list);
            w->fl_list = nullptr;
            w->fl_elems = 0;
            HIP_TRY(hipMalloc(&w->fl_list, (size_t)nq * sizeof(int)));
            w->fl_elems = (size_t)nq;
            *changed = true;
        }
    }
    return ISE_OK;
}

// Every slot is sized for the plan at once (nothing is allocated, filled or synchronised on a later
// call of the same shape: a serving loop's steady state is allocation-free from its second batch on;
// ise_index_reserve_workspaces does this ahead of the first batch).
static int ensure_workspaces(ise_index* h, const ScanPlan& pl, long long nq) {
    bool changed = false;
    for (auto& w : h->ws) {
        int rc = ensure_workspace(&w, pl, nq, &changed);
        if (rc) return rc;
    }
    if (changed) HIP_TRY(hipDeviceSynchronize());  // the fills are done before any stream's launch reads them
    return ISE_OK;
}

// last key of each query's pass -> floor[] for the next pass
__global__ void floor_from_keys_kernel(const u64* pass_keys, int nq, int kp, u64* floor_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) floor_out[q] = pass_keys[(size_t)q * kp + kp - 1];
}
// scatter one pass's keys [nq][kp] into keys_all[nq][k] at column off
__global__ void scatter_pass_kernel(const u64* pass_keys, int nq, int kp, u64* keys_all, int k, int off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq * kp) {
        const int q = i / kp, r = i - q * kp;
        if (off + r < k) keys_all[(size_t)q * k + off + r] = pass_keys[i];
    }
}
// decode final keys into D / I
__global__ void decode_keys_kernel(const u64* keys, long long total, int metric, float* D, long long* I) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const u64 key = keys[i];
        const bool pad = key == KEY_PAD;
        const float sc = unord_f32((uint32_t)(key >> 32));
        D[i] = pad ? (metric == ISE_METRIC_L2 ? FLT_MAX : -FLT_MAX) : (metric == ISE_METRIC_L2 ? sc : -sc);
        I[i] = pad ? -1ll : (long long)(uint32_t)key;
    }
}

struct TimedOut {
    hipEvent_t e0, e1, e2;  // scan start, scan end, end of the batch (merge / rerank / exact launches)
    bool on = false;
};

// a launch tag no older entry of the slot's exchange buffer carries
static int next_xchg_seq(ise_index::WorkSlot* w, hipStream_t st, uint32_t* seq) {
    if (w->xchg_seq >= 0xFFFFFFF0u) {  // wrap: wipe the tags (stream-ordered behind the slot's last use)
        HIP_TRY(hipMemsetAsync(w->xchg, 0xFF, w->xchg_elems * sizeof(u64), st));
        w->xchg_seq = 0;
    }
    *seq = ++w->xchg_seq;
    return ISE_OK;
}
static int next_fl_seq(ise_index::WorkSlot* w, hipStream_t st, uint32_t* seq) {
    if (w->fl_seq >= 0xFFFFFFF0u) {
        HIP_TRY(hipMemsetAsync(w->fl_state, 0, sizeof(u64), st));
        w->fl_seq = 0;
    }
    *seq = ++w->fl_seq;
    return ISE_OK;
}

template <bool RERANK>
static void launch_merge(unsigned grid, size_t lds, hipStream_t st, const MergeParams& mp, const ExactParams& xp) {
    hipLaunchKernelGGL((merge_kernel<RERANK>), dim3(grid), dim3(MERGE_THREADS), lds, st, mp, xp);
}

// The exact fallback scan for the queries the rerank put on the slot's list: launched behind every
// rerank and gated on the GPU (nothing is read back on the way), so a launch without failed
// certificates costs one kernel that exits at once.  k <= 32: one exact pass written straight to
// the outputs; larger k: one pass per 32 results, floor-keyed like the filter passes.
static int enqueue_exact_fallback(ise_index* h, ise_index::WorkSlot* w, const ScanPlan& pl, const ExactParams& xp,
                                  long long nq, hipStream_t st) {
    ExactScanParams xs;
    xs.xb = (const float*)h->xb; xs.q = xp.q; xs.n = h->n; xs.d = h->d; xs.dp = h->dp;
    xs.id_base = xp.id_base; xs.fl_state = w->fl_state; xs.fl_list = w->fl_list; xs.seq = xp.seq;
    xs.part = w->part;  // the filter's lists are dead: [position][nblocks][kp] fits (kp <= kpass, positions <= nq)
    xs.rows_per_block = (long long)pl.tiles_per_block * 16;
    xs.arrive = reinterpret_cast<unsigned int*>(w->fl_state + 1);
    xs.direct_n = 0;
    const size_t lds = (size_t)XQ * h->dp * 4 + (size_t)XQ * 4 * 32 * 8;
    MergeParams mp;  // the per-block lists are merged by the scan's last block
    mp.lists = w->part; mp.qt = 1; mp.n_lists = pl.nblocks; mp.nq = (int)nq; mp.metric = h->metric;
    mp.fl_state = w->fl_state; mp.fl_list = w->fl_list; mp.seq = xp.seq; mp.dbg = nullptr; mp.gate = nullptr;
    const int k = xp.k;
    if (k <= XPASS_MAX) {
        xs.kpass = k; xs.floor_keys = nullptr;
        mp.k = k; mp.stride_list = k; mp.stride_qtile = (long long)pl.nblocks * k;
        mp.D = xp.D; mp.I = xp.I; mp.keys_out = xp.keys_out; mp.out_by_pos = 0;
        hipLaunchKernelGGL(exact_scan_kernel<XQ>, dim3((unsigned)pl.nblocks), dim3(256), lds, st, xs, mp);
        HIP_TRY(hipGetLastError());
        return ISE_OK;
    }
    u64* fb_floor = w->keys_tmp + (size_t)nq * ((size_t)pl.kc + 1 + pl.kpass);  // [nq]
    u64* fb_pass = fb_floor + nq;                                             // [nq][XPASS_MAX]
    const int kp = XPASS_MAX;
    for (int off = 0; off < k; off += kp) {
        xs.kpass = kp; xs.floor_keys = off ? fb_floor : nullptr;
        mp.k = kp; mp.stride_list = kp; mp.stride_qtile = (long long)pl.nblocks * kp;
        mp.D = nullptr; mp.I = nullptr; mp.keys_out = fb_pass; mp.out_by_pos = 1;
        hipLaunchKernelGGL(exact_scan_kernel<XQ>, dim3((unsigned)pl.nblocks), dim3(256), lds, st, xs, mp);
        const long long tot = nq * kp;
        hipLaunchKernelGGL(exact_scatter_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, xp,
                           (const u64*)fb_pass, kp, off, fb_floor);
        HIP_TRY(hipGetLastError());
    }
    return ISE_OK;
}

// ---- small batches against float32 L2 rows: the direct-difference scan IS the search.
// The reference searches one query per request (backend/engine.py:50-55), which in Faiss is the nq < 20
// algorithm: fvec_L2sqr per (query, row) pair and a k-heap.  exact_scan_kernel is that algorithm (it is the
// fallback of the filtered search, with the re-rank's own d()), so for a one-query batch it is launched on its
// own: one kernel, no filter, no certificate, no merge kernel, and VALU work only -- the pass runs at the
// pace of the row stream instead of at the board's power cap.  Same bits as the filtered path.
static bool no_direct() { return knobs().no_direct.load(std::memory_order_relaxed) != 0; }
static bool direct_applies(const ise_index* h, const ise_index::WorkSlot* w, const ScanPlan& pl, long long nq, int k,
                           int* blocks_out) {
    // one query only: two to four queries are VALU-bound here (380-520 us) and faster through the filter (356 us)
    // (ISE_FORCE_EXACT asks for the filtered path's fallback to be exercised: it implies the filtered path)
    if (!pl.exact || nq != 1 || k > XPASS_MAX || h->n <= 0 || !w->fl_state || no_direct() || force_exact()) return false;
    // One query against a short index: up to ~2k rows the direct scan -- ONE launch, no merge, no gate -- has the
    // lower latency, by a microsecond (1000 x 512: 20.6 against 22.1 us per call; 3000: 22.2 against 22.5; 4000:
    // 23.6 against 22.3; 16000: 30.9 against 23.7, scripts/direct_crossover_probe.py): its time grows with the rows
    // a wave inserts and the lists the last block folds, the filtered search behind short_scan_kernel stays flat.
    // $ISE_DIRECT_SHORT_MAX_TILES moves the crossover.
    int max_tiles = knobs().direct_short_max_tiles.load(std::memory_order_relaxed);
    if (max_tiles <= 0) max_tiles = 128;
    if (pl.short_ && pl.tiles_total > max_tiles) return false;
    // rows per block: ~32 KB worth, between 16 (one step per wave) and 64 (4 waves x XR rows x 4 steps), and fewer
    // than 64 only as far as it takes to put a small index on 64 blocks: the scan of a block is latency-bound, so a
    // small index wants many blocks (1000 x 2048, the reference's own: 16 blocks 44.6 us, 63 blocks 30; 1000 x 512:
    // 35.5 -> 28.0 us per call; 1000 x 128: 24.4 -> 18.8), a longer one few lists for the last block to fold
    // (scripts/host_call_probe.py); at most the merge's list count
    long long min_rows = std::max<long long>(16, std::min<long long>(64, (32 * 1024) / ((long long)h->dp * 4)));
    if (h->n < min_rows * 64) min_rows = std::max<long long>(16, h->n / 64);
    long long blocks = std::min<long long>(MERGE_LISTS_MAX, (h->n + min_rows - 1) / min_rows);
    static const long long per_cu = [] { const char* e = getenv("ISE_DIRECT_BLOCKS_PER_CU"); const int v = e ? atoi(e) : 0; return (long long)(v > 0 ? v : 2); }();
    blocks = std::min<long long>(blocks, (long long)h->num_cu * per_cu);
    if ((size_t)nq * blocks * k > w->part_elems) return false;  // the slot's lists are sized for the filter's plan
    *blocks_out = (int)blocks;
    return true;
}
static int direct_small_enqueue(ise_index* h, ise_index::WorkSlot* w, int blocks, const float* q_dev, long long nq, int k,
                                uint32_t id_base, float* D_dev, long long* I_dev, u64* keys_out, hipStream_t st,
                                TimedOut* tm) {
    ExactScanParams xs;
    xs.xb = (const float*)h->xb; xs.q = q_dev; xs.n = h->n; xs.d = h->d; xs.dp = h->dp;
    xs.kpass = k; xs.id_base = id_base; xs.fl_state = w->fl_state; xs.fl_list = w->fl_list; xs.seq = 0;
    xs.floor_keys = nullptr; xs.part = w->part;
    xs.rows_per_block = (h->n + blocks - 1) / blocks;
    xs.arrive = reinterpret_cast<unsigned int*>(w->fl_state + 1);
    xs.direct_n = (int)nq;
    MergeParams mp;
    mp.lists = w->part; mp.qt = 1; mp.n_lists = blocks; mp.nq = (int)nq; mp.metric = h->metric;
    mp.fl_state = w->fl_state; mp.fl_list = w->fl_list; mp.seq = 0; mp.dbg = nullptr; mp.gate = nullptr;
    mp.k = k; mp.stride_list = k; mp.stride_qtile = (long long)blocks * k;
    mp.D = D_dev; mp.I = I_dev; mp.keys_out = keys_out; mp.out_by_pos = 0;
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
    const int qn = 1;
    const size_t lds = (size_t)qn * h->dp * 4 + (size_t)qn * 4 * 32 * 8;
    // non-temporal loads: every wave instruction covers whole lines here (1 KB contiguous), where they stream
    // faster than plain loads (336 against 359 us)
    hipLaunchKernelGGL((exact_scan_kernel<1, true>), dim3((unsigned)blocks), dim3(256), lds, st, xs, mp);
    HIP_TRY(hipGetLastError());
    if (tm && tm->on) {
        HIP_TRY(hipEventRecord(tm->e1, st));
        HIP_TRY(hipEventRecord(tm->e2, st));
    }
    h->direct_queries += (unsigned long long)nq;
    return ISE_OK;
}

// ---- large query batches against float32 L2 rows: sample pass (dump) -> thresholds -> GEMM-shaped
// pass -> select + exact re-rank (ise_gemm_scan.hpp)
#define GEMM_MIN_NQ 256
#define GEMM_CAPQ 4096        /* candidate slots per query (a power of two); expected fill ~ N kc / sample rows */
#define GEMM_SAMPLE_SLABS 128 /* 128-row slabs in the threshold sample (16384 rows), spread over the index */
static bool gemm_applies(const ise_index* h, long long nq, int k) {
    static const bool off = [] { const char* e = getenv("ISE_NO_GEMM"); return e && e[0] == '1'; }();
    static const long long min_nq = [] { const char* e = getenv("ISE_GEMM_MIN_NQ"); const int v = e ? atoi(e) : 0; return (long long)(v >= 17 ? v : GEMM_MIN_NQ); }();
    // measured crossover at 1M x 512: float32 L2 between 192 and 256 queries, bf16 rows (16x the MFMA rate) at 128
    const long long need = h->storage == ISE_STORE_BF16 ? std::min<long long>(min_nq, 128) : min_nq;
    if (off || nq < need || h->n < 128ll * 1024) return false;  // shorter indexes: the streaming passes are as fast
    if (h->dp > 512 || h->dp % 128 != 0) return false;            // the row tiles live in <= 128 VGPRs
    if (h->storage == ISE_STORE_BF16) return k <= XPASS_MAX;      // either metric (ise_gemm_bf16.hpp)
    if (!uses_shift(h)) return k <= XPASS_MAX;                    // float32 inner product: no re-rank behind the pass
    return k + exact_extra(k) <= XPASS_MAX;                       // the select stage hands at most 32 candidates to the re-rank
}
#define GEMM_CAPW 2048 /* entries of a wave's candidate buffer (expected fill: a few hundred) */
struct GemmLayout {
    size_t qprep, xn, tau, ccnt, dump, cand, wbuf, wcnt, total;
};
static GemmLayout gemm_layout(const ise_index* h) {
    GemmLayout g;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    g.qprep = take((size_t)GEMM_NQ_MAX * qs_stride_for(h) * 4);
    g.xn = take((size_t)GEMM_NQ_MAX * 4);
    g.tau = take((size_t)GEMM_NQ_MAX * 4);
    g.ccnt = take((size_t)(GEMM_NQ_MAX * GEMM_SUBS + 64) * 4);  // [GEMM_NQ_MAX][GEMM_SUBS] counters + the overflow flag
    g.dump = take((size_t)GEMM_NQ_MAX * GEMM_SAMPLE_SLABS * 128 * 4);
    g.cand = take((size_t)GEMM_NQ_MAX * GEMM_CAPQ * 8);
    g.wbuf = take((size_t)h->num_cu * 8 * GEMM_CAPW * 16);
    g.wcnt = take((size_t)h->num_cu * 8 * 4);
    g.total = o;
    return g;
}

// the plan a batch is enqueued with: large batches of float32 L2 queries go through the GEMM-shaped path
// in chunks of GEMM_NQ_MAX, so their streaming plan (the exact fallback's shape, the slot's lists) is a chunk's
static int plan_for_batch(const ise_index* h, long long nq, int k, ScanPlan* pl) {
    const bool big = gemm_applies(h, nq, k);
    int rc = make_plan(h, big ? std::min<long long>(nq, GEMM_NQ_MAX) : nq, k, pl);
    if (rc) return rc;
    pl->gemm = big;
    pl->gemm_bytes = big ? gemm_layout(h).total : 0;
    return ISE_OK;
}

template <int NS, bool DUMP, bool IPM>
static void launch_gemm_metric(int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&gemm_scan_kernel<NS, DUMP, IPM>), LDS_LIMIT);
    hipLaunchKernelGGL((gemm_scan_kernel<NS, DUMP, IPM>), dim3(grid), dim3(512), lds, st, gp);
}
template <int NS, bool DUMP>
static void launch_gemm_one(int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    if (gp.metric == ISE_METRIC_INNER_PRODUCT) launch_gemm_metric<NS, DUMP, true>(grid, lds, st, gp);
    else launch_gemm_metric<NS, DUMP, false>(grid, lds, st, gp);
}
template <bool DUMP>
static int launch_gemm(int ns, int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    switch (ns) {
        case 8: launch_gemm_one<8, DUMP>(grid, lds, st, gp); break;
        case 16: launch_gemm_one<16, DUMP>(grid, lds, st, gp); break;
        case 24: launch_gemm_one<24, DUMP>(grid, lds, st, gp); break;
        case 32: launch_gemm_one<32, DUMP>(grid, lds, st, gp); break;
        default: return fail(ISE_E_INVALID, "large-batch path: unsupported padded dimension");
    }
    return ISE_OK;
}

// one chunk of <= GEMM_NQ_MAX queries; the slot w is already this stream's
static int search_large_chunk(ise_index* h, ise_index::WorkSlot* w, const float* q_dev, long long nq, int k,
                              uint32_t id_base, float* D_dev, long long* I_dev, u64* keys_out, hipStream_t st,
                              TimedOut* tm) {
    const int kc = k + exact_extra(k);
    const int S = qs_stride_for(h);
    const GemmLayout gl = gemm_layout(h);
    h->gemm_chunks++;
    float* qprep = reinterpret_cast<float*>(w->gemm + gl.qprep);
    float* xn = reinterpret_cast<float*>(w->gemm + gl.xn);
    float* tau = reinterpret_cast<float*>(w->gemm + gl.tau);
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(w->gemm + gl.ccnt);
    float* dump = reinterpret_cast<float*>(w->gemm + gl.dump);
    u64* cand = reinterpret_cast<u64*>(w->gemm + gl.cand);
    const int nq_pad = (int)((nq + GQ - 1) / GQ * GQ);
    int rc;

    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
    hipLaunchKernelGGL(qprep_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, q_dev, (int)nq, nq_pad, h->d, S,
                       (const float*)h->mu, qprep, xn);
    HIP_TRY(hipGetLastError());

    GemmScanParams gp;
    gp.xb = (const float*)h->xb; gp.norms = h->norms; gp.mu = h->mu; gp.n = h->n; gp.rows16 = (h->n + 15) / 16 * 16;
    gp.dp = h->dp; gp.S = S; gp.qprep = qprep; gp.xn = xn; gp.tau = tau; gp.nq = (int)nq; gp.nq_pad = nq_pad;
    gp.beta = exact_beta(h); gp.id_base = id_base; gp.metric = ISE_METRIC_L2;
    gp.wbuf = reinterpret_cast<u32x4*>(w->gemm + gl.wbuf); gp.wcnt = reinterpret_cast<unsigned int*>(w->gemm + gl.wcnt);
    gp.capw = GEMM_CAPW;
    gp.ablate = 0;
    const int slabs_all = (int)((h->n + 127) / 128);
    const size_t lds = gemm_lds_bytes(S);
    const int ns = h->dp / 16;

    // ---- thresholds: GEMM_SAMPLE_SLABS slabs spread over the index, every score dumped, k-th selected per query.
    // One slab per block, the query stages split over qparts blocks per slab, so that the sample
    // keeps every CU busy for a fraction of a slab's time.
    gp.slabs = std::min(slabs_all, GEMM_SAMPLE_SLABS);
    gp.slab_stride = slabs_all / gp.slabs;
    const int nstages = nq_pad / GQ;
    gp.qparts = std::max(1, std::min(nstages, (2 * h->num_cu) / gp.slabs));
    gp.dump = dump;
    if ((rc = launch_gemm<true>(ns, gp.slabs * gp.qparts, lds, st, gp))) return rc;
    hipLaunchKernelGGL(kth_select_kernel, dim3((unsigned)nq_pad), dim3(256), 0, st, (const float*)dump, gp.slabs * 128, kc,
                       (int)nq, tau);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(ccnt, 0, (size_t)(GEMM_NQ_MAX * GEMM_SUBS + 64) * 4, st));  // the counters and, behind them, the overflow flag

    // ---- the GEMM-shaped pass over the whole index
    gp.slabs = slabs_all; gp.slab_stride = 1; gp.qparts = 1; gp.dump = nullptr;
    const int grid = std::min(slabs_all, h->num_cu);
    if ((rc = launch_gemm<false>(ns, grid, lds, st, gp))) return rc;
    unsigned int* overflow = ccnt + GEMM_NQ_MAX * GEMM_SUBS;
    hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)grid * 8), dim3(256), 0, st, (const u32x4*)gp.wbuf,
                       (const unsigned int*)gp.wcnt, GEMM_CAPW, cand, ccnt, GEMM_CAPQ, overflow);
    HIP_TRY(hipGetLastError());
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));

    // ---- select the kc best candidates, re-rank exactly, certify; the exact scan takes what fails
    ScanPlan pl;  // shapes the exact fallback scan (blocks, rows per block) and sized the slot's lists
    rc = make_plan(h, nq, k, &pl);
    if (rc) return rc;
    ExactParams xp;
    xp.xb = (const float*)h->xb; xp.q = q_dev; xp.n = h->n; xp.d = h->d; xp.dp = h->dp; xp.nq = (int)nq;
    xp.k = k; xp.kc = kc; xp.id_base = id_base; xp.D = D_dev; xp.I = I_dev; xp.keys_out = keys_out;
    xp.fl_state = w->fl_state; xp.fl_list = w->fl_list; xp.seq = 0; xp.stats = h->stats_dev;
    xp.force_fail = force_exact() ? 1 : 0;
    xp.tau_bound = tau;
    if ((rc = next_fl_seq(w, st, &xp.seq))) return rc;
    hipLaunchKernelGGL(gemm_select_kernel, dim3((unsigned)nq), dim3(256),
                       rerank_lds_bytes(h->dp, kc) + (size_t)(GEMM_CAPQ + 256) * 8, st, xp, (const u64*)cand,
                       (const unsigned int*)ccnt, GEMM_CAPQ, (const unsigned int*)overflow);
    HIP_TRY(hipGetLastError());
    if ((rc = enqueue_exact_fallback(h, w, pl, xp, nq, st))) return rc;
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
    return ISE_OK;
}

// The streaming path for one batch on slot w: scan pass(es) -> merge (-> exact re-rank -> gated exact scan
// for float32 L2).  gate: optional device flag -- when given, every kernel of the batch exits at once
// unless it is non-zero (the bf16 large-batch path queues this behind itself for the case that its
// candidate buffers overflow).
static int scan_path_enqueue(ise_index* h, ise_index::WorkSlot* w, const ScanPlan& pl, const float* q_dev, long long nq,
                             int k, uint32_t id_base, float* D_dev, long long* I_dev, u64* keys_out, hipStream_t st,
                             TimedOut* tm, const unsigned int* gate) {
    int rc;
    ScanParams sp;
    sp.xb = h->xb; sp.norms = h->norms; sp.q = q_dev; sp.mu = h->mu; sp.floor_keys = nullptr; sp.part = w->part;
    sp.n = h->n; sp.d = h->d; sp.dp = h->dp; sp.qs_stride = qs_stride_for(h);
    sp.row_slots = (int)(row_bytes(h) / 16);
    sp.nq = (int)nq; sp.k = pl.kpass; sp.kb = pl.kb; sp.metric = h->metric; sp.id_base = id_base;
    sp.beta = pl.exact ? exact_beta(h) : 0.f;
    sp.tiles_total = pl.tiles_total; sp.tiles_per_block = pl.tiles_per_block;
    sp.xchg = xchg_enabled() ? w->xchg : nullptr;
    sp.xchg_seq = 0;
    sp.gate = gate;
    sp.ablate = 0;
    sp.stamps = nullptr;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_ABLATE")) sp.ablate = atoi(e);
    if (const char* e = getenv("ISE_STAMPS")) sp.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
#endif

    MergeParams mp;
    const int NQ = 16 * pl.T;
    mp.lists = w->part; mp.stride_list = (long long)NQ * pl.kpass;
    mp.stride_qtile = (long long)pl.nblocks * NQ * pl.kpass; mp.qt = NQ;
    mp.n_lists = pl.nblocks; mp.nq = (int)nq; mp.k = pl.kpass; mp.metric = h->metric;
    mp.fl_state = nullptr; mp.fl_list = nullptr; mp.seq = 0; mp.out_by_pos = 0;
    mp.dbg = h->stats_dev + 8;
    mp.gate = gate;

    ExactParams xp;  // used on the exact path only
    xp.xb = (const float*)h->xb; xp.q = q_dev; xp.n = h->n; xp.d = h->d; xp.dp = h->dp; xp.nq = (int)nq;
    xp.k = k; xp.kc = pl.kc; xp.id_base = id_base; xp.D = D_dev; xp.I = I_dev; xp.keys_out = keys_out;
    xp.fl_state = w->fl_state; xp.fl_list = w->fl_list; xp.seq = 0; xp.stats = h->stats_dev;
    xp.force_fail = force_exact() ? 1 : 0;
    xp.tau_bound = nullptr;

    if (pl.short_) {  // short index: stream + per-block selection in one pass without boot or thresholds
        if (gate) return fail(ISE_E_INVALID, "internal: a gated rerun was planned for the short-index kernel");
        ShortParams shp;
        shp.even_split = 1;
        shp.T = pl.T;
        shp.nqt = pl.nqt;
        h->short_batches++;
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
        launch_short(h, pl.ch, pl.waves, pl.short_bpc, pl.nblocks, pl.lds, st, sp, shp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
        if (pl.exact) {
            if ((rc = next_fl_seq(w, st, &xp.seq))) return rc;
            mp.D = nullptr; mp.I = nullptr; mp.keys_out = nullptr;
            launch_merge<true>((unsigned)nq, rerank_lds_bytes(h->dp, pl.kc), st, mp, xp);
            HIP_TRY(hipGetLastError());
            if ((rc = enqueue_exact_fallback(h, w, pl, xp, nq, st))) return rc;
        } else {
            mp.D = D_dev; mp.I = I_dev; mp.keys_out = keys_out;
            launch_merge<false>((unsigned)nq, 0, st, mp, xp);
            HIP_TRY(hipGetLastError());
        }
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
        return ISE_OK;
    }
    const dim3 grid((unsigned)pl.nblocks, (unsigned)pl.nqt);
    if (pl.kc <= pl.kpass) {  // one scan pass selects everything the merge stage needs
        if ((rc = next_xchg_seq(w, st, &sp.xchg_seq))) return rc;
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
        launch_scan(h, pl.ch, pl.waves, pl.T, grid, pl.lds, st, sp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
        if (pl.exact) {  // merge the kc lower-bound keys, re-rank them exactly, certify or list for the exact scan
            if ((rc = next_fl_seq(w, st, &xp.seq))) return rc;
            mp.D = nullptr; mp.I = nullptr; mp.keys_out = nullptr;
            launch_merge<true>((unsigned)nq, rerank_lds_bytes(h->dp, pl.kc), st, mp, xp);
            HIP_TRY(hipGetLastError());
            if ((rc = enqueue_exact_fallback(h, w, pl, xp, nq, st))) return rc;
        } else {
            mp.D = D_dev; mp.I = I_dev; mp.keys_out = keys_out;
            launch_merge<false>((unsigned)nq, 0, st, mp, xp);
            HIP_TRY(hipGetLastError());
        }
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
        return ISE_OK;
    }
    // kc > KPASS_MAX: passes of KPASS_MAX; a pass only admits keys above the
    // previous pass's last key (keys are totally ordered and unique)
    const int kc = pl.kc;
    u64* keys_all = (!pl.exact && keys_out) ? keys_out : w->keys_tmp;  // [nq][kc]
    u64* floor_dev = w->keys_tmp + (size_t)nq * kc;                    // [nq]
    u64* pass_keys = floor_dev + nq;                                   // [nq][kpass]
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
    for (int off = 0; off < kc; off += pl.kpass) {
        sp.floor_keys = off ? floor_dev : nullptr;
        mp.D = nullptr; mp.I = nullptr; mp.keys_out = pass_keys;
        if ((rc = next_xchg_seq(w, st, &sp.xchg_seq))) return rc;
        launch_scan(h, pl.ch, pl.waves, pl.T, grid, pl.lds, st, sp);
        HIP_TRY(hipGetLastError());
        launch_merge<false>((unsigned)nq, 0, st, mp, xp);
        HIP_TRY(hipGetLastError());
        const int tot = (int)nq * pl.kpass;
        hipLaunchKernelGGL(scatter_pass_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, pass_keys, (int)nq,
                           pl.kpass, keys_all, kc, off);
        hipLaunchKernelGGL(floor_from_keys_kernel, dim3(((int)nq + 255) / 256), dim3(256), 0, st, pass_keys, (int)nq,
                           pl.kpass, floor_dev);
        HIP_TRY(hipGetLastError());
    }
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
    if (pl.exact) {
        if ((rc = next_fl_seq(w, st, &xp.seq))) return rc;
        hipLaunchKernelGGL(rerank_kernel, dim3((unsigned)nq), dim3(256), rerank_lds_bytes(h->dp, kc), st, xp,
                           (const u64*)keys_all);
        HIP_TRY(hipGetLastError());
        if ((rc = enqueue_exact_fallback(h, w, pl, xp, nq, st))) return rc;
    } else if (D_dev) {
        const long long total = nq * k;
        hipLaunchKernelGGL(decode_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, keys_all, total,
                           h->metric, D_dev, I_dev);
        HIP_TRY(hipGetLastError());
    }
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
    return ISE_OK;
}


template <int NS, bool DUMP, bool L2>
static void launch_gemm_bf16_metric(int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&gemm_scan_bf16_kernel<NS, DUMP, L2>), LDS_LIMIT);
    hipLaunchKernelGGL((gemm_scan_bf16_kernel<NS, DUMP, L2>), dim3(grid), dim3(512), lds, st, gp);
}
template <int NS, bool DUMP>
static void launch_gemm_bf16_one(int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    if (gp.metric == ISE_METRIC_L2) launch_gemm_bf16_metric<NS, DUMP, true>(grid, lds, st, gp);
    else launch_gemm_bf16_metric<NS, DUMP, false>(grid, lds, st, gp);
}
template <bool DUMP>
static int launch_gemm_bf16(int ns, int grid, size_t lds, hipStream_t st, const GemmScanParams& gp) {
    switch (ns) {
        case 4: launch_gemm_bf16_one<4, DUMP>(grid, lds, st, gp); break;
        case 8: launch_gemm_bf16_one<8, DUMP>(grid, lds, st, gp); break;
        case 12: launch_gemm_bf16_one<12, DUMP>(grid, lds, st, gp); break;
        case 16: launch_gemm_bf16_one<16, DUMP>(grid, lds, st, gp); break;
        default: return fail(ISE_E_INVALID, "large-batch path: unsupported padded dimension");
    }
    return ISE_OK;
}

// bf16 rows, one chunk of <= GEMM_NQ_MAX queries (ise_gemm_bf16.hpp): sample dump -> thresholds -> GEMM pass ->
// regroup -> select; then the streaming passes, gated on the rerun flag (set when a candidate buffer overflowed)
static int search_large_chunk_bf16(ise_index* h, ise_index::WorkSlot* w, const float* q_dev, long long nq, int k,
                                   uint32_t id_base, float* D_dev, long long* I_dev, u64* keys_out, hipStream_t st,
                                   TimedOut* tm) {
    const int S = qs_stride_for(h);
    const GemmLayout gl = gemm_layout(h);
    h->gemm_chunks++;
    uint32_t* qprep = reinterpret_cast<uint32_t*>(w->gemm + gl.qprep);
    float* xn = reinterpret_cast<float*>(w->gemm + gl.xn);
    float* tau = reinterpret_cast<float*>(w->gemm + gl.tau);
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(w->gemm + gl.ccnt);
    float* dump = reinterpret_cast<float*>(w->gemm + gl.dump);
    u64* cand = reinterpret_cast<u64*>(w->gemm + gl.cand);
    const int nq_pad = (int)((nq + GB_GQ - 1) / GB_GQ * GB_GQ);
    int rc;

    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
    hipLaunchKernelGGL(qprep_bf16_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, q_dev, (int)nq, nq_pad, h->d, S,
                       qprep, xn);
    HIP_TRY(hipGetLastError());

    constexpr int ROWS = 8 * GB_XT * 16;
    GemmScanParams gp;
    gp.xb = (const float*)h->xb; gp.norms = h->norms; gp.mu = nullptr; gp.n = h->n; gp.rows16 = (h->n + 15) / 16 * 16;
    gp.dp = h->dp; gp.S = S; gp.qprep = reinterpret_cast<const float*>(qprep); gp.xn = xn; gp.tau = tau; gp.nq = (int)nq;
    gp.nq_pad = nq_pad; gp.beta = 0.f; gp.metric = h->metric; gp.id_base = id_base;
    gp.wbuf = reinterpret_cast<u32x4*>(w->gemm + gl.wbuf); gp.wcnt = reinterpret_cast<unsigned int*>(w->gemm + gl.wcnt);
    gp.capw = GEMM_CAPW;
    gp.ablate = 0;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_GEMM_ABLATE")) gp.ablate = atoi(e);
#endif
    const int slabs_all = (int)((h->n + ROWS - 1) / ROWS);
    const size_t lds = (size_t)2 * GB_GQ * S * 4 + (size_t)2 * GEMM_NQ_MAX * 4;
    const int ns = h->dp / 32;

    gp.slabs = std::min(slabs_all, GEMM_SAMPLE_SLABS * 128 / ROWS);  // the same 16384 sample rows
    gp.slab_stride = slabs_all / gp.slabs;
    const int nstages = nq_pad / GB_GQ;
    gp.qparts = std::max(1, std::min(nstages, (2 * h->num_cu) / gp.slabs));
    gp.dump = dump;
    if ((rc = launch_gemm_bf16<true>(ns, gp.slabs * gp.qparts, lds, st, gp))) return rc;
    hipLaunchKernelGGL(kth_select_kernel, dim3((unsigned)nq_pad), dim3(256), 0, st, (const float*)dump, gp.slabs * ROWS, k,
                       (int)nq, tau);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(ccnt, 0, (size_t)(GEMM_NQ_MAX * GEMM_SUBS + 64) * 4, st));  // counters, overflow flag, rerun flag

    gp.slabs = slabs_all; gp.slab_stride = 1; gp.qparts = 1; gp.dump = nullptr;
    const int grid = std::min(slabs_all, h->num_cu);
    if ((rc = launch_gemm_bf16<false>(ns, grid, lds, st, gp))) return rc;
    unsigned int* overflow = ccnt + GEMM_NQ_MAX * GEMM_SUBS;
    unsigned int* rerun = overflow + 1;
    hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)grid * 8), dim3(256), 0, st, (const u32x4*)gp.wbuf,
                       (const unsigned int*)gp.wcnt, GEMM_CAPW, cand, ccnt, GEMM_CAPQ, overflow);
    HIP_TRY(hipGetLastError());
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
    hipLaunchKernelGGL(gemm_select_plain_kernel, dim3((unsigned)nq), dim3(256), (size_t)(GEMM_CAPQ + 320) * 8, st, (const u64*)cand,
                       (const unsigned int*)ccnt, GEMM_CAPQ, (const unsigned int*)overflow, rerun, k, h->metric, D_dev, I_dev,
                       keys_out);
    HIP_TRY(hipGetLastError());
    // incomplete candidates anywhere in the chunk: the streaming passes answer the whole chunk instead
    ScanPlan pl;
    rc = make_plan(h, nq, k, &pl, /*allow_short=*/false);  // a gated rerun: the streaming kernels carry the gate
    if (rc) return rc;
    rc = scan_path_enqueue(h, w, pl, q_dev, nq, k, id_base, D_dev, I_dev, keys_out, st, nullptr, rerun);
    if (rc) return rc;
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
    return ISE_OK;
}

// float32 INNER PRODUCT rows, one chunk of <= GEMM_NQ_MAX queries: the GEMM-shaped pass of ise_gemm_scan.hpp without
// shift, norms or re-rank (the reference's default index type is "cosine" = IndexFlatIP over normalised rows,
// backend/utils.py:293,300-303): sample dump -> thresholds (the k-th smallest sampled score) -> GEMM pass ->
// regroup -> select; then the streaming passes, gated on the rerun flag (set when a candidate buffer overflowed).
// Same bits as the streaming passes: the kernel sums a dot product in scan_kernel's order.
static int search_large_chunk_ip(ise_index* h, ise_index::WorkSlot* w, const float* q_dev, long long nq, int k,
                                 uint32_t id_base, float* D_dev, long long* I_dev, u64* keys_out, hipStream_t st,
                                 TimedOut* tm) {
    const int S = qs_stride_for(h);
    const GemmLayout gl = gemm_layout(h);
    h->gemm_chunks++;
    float* qprep = reinterpret_cast<float*>(w->gemm + gl.qprep);
    float* xn = reinterpret_cast<float*>(w->gemm + gl.xn);
    float* tau = reinterpret_cast<float*>(w->gemm + gl.tau);
    unsigned int* ccnt = reinterpret_cast<unsigned int*>(w->gemm + gl.ccnt);
    float* dump = reinterpret_cast<float*>(w->gemm + gl.dump);
    u64* cand = reinterpret_cast<u64*>(w->gemm + gl.cand);
    const int nq_pad = (int)((nq + GQ - 1) / GQ * GQ);
    int rc;

    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
    hipLaunchKernelGGL(qprep_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, q_dev, (int)nq, nq_pad, h->d, S,
                       (const float*)nullptr, qprep, xn);
    HIP_TRY(hipGetLastError());

    GemmScanParams gp;
    gp.xb = (const float*)h->xb; gp.norms = h->norms; gp.mu = nullptr; gp.n = h->n; gp.rows16 = (h->n + 15) / 16 * 16;
    gp.dp = h->dp; gp.S = S; gp.qprep = qprep; gp.xn = xn; gp.tau = tau; gp.nq = (int)nq; gp.nq_pad = nq_pad;
    gp.beta = 0.f; gp.id_base = id_base; gp.metric = ISE_METRIC_INNER_PRODUCT;
    gp.wbuf = reinterpret_cast<u32x4*>(w->gemm + gl.wbuf); gp.wcnt = reinterpret_cast<unsigned int*>(w->gemm + gl.wcnt);
    gp.capw = GEMM_CAPW;
    gp.ablate = 0;
    const int slabs_all = (int)((h->n + 127) / 128);
    const size_t lds = gemm_lds_bytes(S);
    const int ns = h->dp / 16;

    gp.slabs = std::min(slabs_all, GEMM_SAMPLE_SLABS);
    gp.slab_stride = slabs_all / gp.slabs;
    const int nstages = nq_pad / GQ;
    gp.qparts = std::max(1, std::min(nstages, (2 * h->num_cu) / gp.slabs));
    gp.dump = dump;
    if ((rc = launch_gemm<true>(ns, gp.slabs * gp.qparts, lds, st, gp))) return rc;
    hipLaunchKernelGGL(kth_select_kernel, dim3((unsigned)nq_pad), dim3(256), 0, st, (const float*)dump, gp.slabs * 128, k,
                       (int)nq, tau);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(ccnt, 0, (size_t)(GEMM_NQ_MAX * GEMM_SUBS + 64) * 4, st));  // counters, overflow flag, rerun flag

    gp.slabs = slabs_all; gp.slab_stride = 1; gp.qparts = 1; gp.dump = nullptr;
    const int grid = std::min(slabs_all, h->num_cu);
    if ((rc = launch_gemm<false>(ns, grid, lds, st, gp))) return rc;
    unsigned int* overflow = ccnt + GEMM_NQ_MAX * GEMM_SUBS;
    unsigned int* rerun = overflow + 1;
    hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)grid * 8), dim3(256), 0, st, (const u32x4*)gp.wbuf,
                       (const unsigned int*)gp.wcnt, GEMM_CAPW, cand, ccnt, GEMM_CAPQ, overflow);
    HIP_TRY(hipGetLastError());
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
    hipLaunchKernelGGL(gemm_select_plain_kernel, dim3((unsigned)nq), dim3(256), (size_t)(GEMM_CAPQ + 320) * 8, st, (const u64*)cand,
                       (const unsigned int*)ccnt, GEMM_CAPQ, (const unsigned int*)overflow, rerun, k, h->metric, D_dev, I_dev,
                       keys_out);
    HIP_TRY(hipGetLastError());
    ScanPlan pl;
    rc = make_plan(h, nq, k, &pl, /*allow_short=*/false);  // a gated rerun: the streaming kernels carry the gate
    if (rc) return rc;
    rc = scan_path_enqueue(h, w, pl, q_dev, nq, k, id_base, D_dev, I_dev, keys_out, st, nullptr, rerun);
    if (rc) return rc;
    if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
    return ISE_OK;
}

// enqueue one search batch; outputs (D, I) and/or keys.  Nothing here blocks once the slots are
// sized (first batch of a shape) and the shift is current (first batch after rows were added).
static int search_enqueue(ise_index* h, const float* q_dev, long long nq, int k, uint32_t id_base, float* D_dev,
                          long long* I_dev, u64* keys_out, hipStream_t st, TimedOut* tm) {
    int rc = prepare_shift_locked(h, st);
    if (rc) return rc;
    ScanPlan pl;
    rc = plan_for_batch(h, nq, k, &pl);
    if (rc) return rc;
    rc = ensure_workspaces(h, pl, pl.gemm ? std::min<long long>(nq, GEMM_NQ_MAX) : nq);
    if (rc) return rc;
    ise_index::WorkSlot* w = nullptr;
    bool same_stream = false;
    for (auto& s : h->ws)
        if (s.used && s.last_stream == st) { w = &s; same_stream = true; break; }
    if (!w)
        for (auto& s : h->ws)
            if (!s.used) { w = &s; break; }
    if (!w) w = &h->ws[h->ws_next++ % ise_index::NWS];
    if (w->used && !same_stream) HIP_TRY(hipStreamWaitEvent(st, w->done, 0));
    struct Release {  // whatever path returns, a later user on another stream waits for this call
        ise_index::WorkSlot* w;
        hipStream_t st;
        ~Release() {
            if (hipEventRecord(w->done, st) == hipSuccess) { w->used = true; w->last_stream = st; }
        }
    } release{w, st};

    if (pl.gemm) {  // float32 L2 or bf16 rows, nq >= 256: GEMM-shaped pass, GEMM_NQ_MAX queries at a time
        for (long long q0 = 0; q0 < nq; q0 += GEMM_NQ_MAX) {
            const long long m = std::min<long long>(GEMM_NQ_MAX, nq - q0);
            rc = (h->storage == ISE_STORE_BF16 ? search_large_chunk_bf16
                  : uses_shift(h)               ? search_large_chunk
                                                : search_large_chunk_ip)(
                h, w, q_dev + (size_t)q0 * h->d, m, k, id_base, D_dev ? D_dev + (size_t)q0 * k : nullptr,
                I_dev ? I_dev + (size_t)q0 * k : nullptr, keys_out ? keys_out + (size_t)q0 * k : nullptr, st, tm);
            if (rc) return rc;
        }
        return ISE_OK;
    }

    int dblocks = 0;
    if (direct_applies(h, w, pl, nq, k, &dblocks))
        return direct_small_enqueue(h, w, dblocks, q_dev, nq, k, id_base, D_dev, I_dev, keys_out, st, tm);
    return scan_path_enqueue(h, w, pl, q_dev, nq, k, id_base, D_dev, I_dev, keys_out, st, tm, nullptr);
}

static int check_search_args(const ise_index* h, const void* q, long long nq, int k) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (nq < 0 || (nq > 0 && !q)) return fail(ISE_E_INVALID, "bad query argument");
    if (k <= 0 || k > ISE_MAX_K) return fail(ISE_E_INVALID, "k must be in [1, 2048]");
    if (nq > (1ll << 20)) return fail(ISE_E_INVALID, "at most 2^20 queries per call");
    return ISE_OK;
}

extern "C" int ise_index_search_device(ise_index_t* h, const float* q_dev, int64_t nq, int k, float* D_dev,
                                       int64_t* I_dev, void* stream) {
    int rc = check_search_args(h, q_dev, nq, k);
    if (rc) return rc;
    if (nq == 0) return ISE_OK;
    if (!D_dev || !I_dev) return fail(ISE_E_INVALID, "output pointer is NULL");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    return search_enqueue(h, q_dev, nq, k, 0u, D_dev, (long long*)I_dev, nullptr, (hipStream_t)stream, nullptr);
}

extern "C" int ise_index_search_keys_device(ise_index_t* h, const float* q_dev, int64_t nq, int k, uint32_t id_base,
                                            uint64_t* keys_dev, void* stream) {
    int rc = check_search_args(h, q_dev, nq, k);
    if (rc) return rc;
    if (nq == 0) return ISE_OK;
    if (!keys_dev) return fail(ISE_E_INVALID, "output pointer is NULL");
    if ((long long)id_base + h->n > (1ll << 32)) return fail(ISE_E_INVALID, "id_base + ntotal exceeds 2^32");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    return search_enqueue(h, q_dev, nq, k, id_base, nullptr, nullptr, (u64*)keys_dev, (hipStream_t)stream, nullptr);
}

extern "C" int ise_index_search_timed_device(ise_index_t* h, const float* q_dev, int64_t nq, int k, float* D_dev,
                                             int64_t* I_dev, void* stream, int iters, float* scan_ms_avg,
                                             float* merge_ms_avg) {
    int rc = check_search_args(h, q_dev, nq, k);
    if (rc) return rc;
    if (nq == 0 || iters <= 0) return fail(ISE_E_INVALID, "timed search needs nq > 0, iters > 0");
    if (!D_dev || !I_dev) return fail(ISE_E_INVALID, "output pointer is NULL");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    hipStream_t st = (hipStream_t)stream;
    TimedOut tm;
    tm.on = true;
    HIP_TRY(hipEventCreate(&tm.e0));
    HIP_TRY(hipEventCreate(&tm.e1));
    HIP_TRY(hipEventCreate(&tm.e2));
    double s_scan = 0, s_merge = 0;
    for (int it = 0; it < iters; it++) {
        rc = search_enqueue(h, q_dev, nq, k, 0u, D_dev, (long long*)I_dev, nullptr, st, &tm);
        if (rc) break;
        hipError_t e = hipEventSynchronize(tm.e2);
        if (e != hipSuccess) { rc = fail(ISE_E_HIP, hipGetErrorString(e)); break; }
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, tm.e0, tm.e1);
        (void)hipEventElapsedTime(&b, tm.e1, tm.e2);
        s_scan += a;
        s_merge += b;
    }
    (void)hipEventDestroy(tm.e0);
    (void)hipEventDestroy(tm.e1);
    (void)hipEventDestroy(tm.e2);
    if (rc) return rc;
    if (scan_ms_avg) *scan_ms_avg = (float)(s_scan / iters);
    if (merge_ms_avg) *merge_ms_avg = (float)(s_merge / iters);
    return ISE_OK;
}

// Host-API searches run on a small pool of contexts (stream + staging buffers), and hold the handle
// lock only while their kernels are enqueued: concurrent callers (Flask request threads,
// backend/engine.py:137; joblib threads, backend/descriptors.py:125) overlap their copies and
// their scans instead of queueing behind one stream.
static ise_index::HostCtx* acquire_ctx(ise_index* h) {
    std::unique_lock<std::mutex> lk(h->hc_mu);
    for (;;) {
        for (auto& c : h->hc)
            if (!c.busy) { c.busy = true; return &c; }
        h->hc_cv.wait(lk);
    }
}
static void release_ctx(ise_index* h, ise_index::HostCtx* c) {
    { std::lock_guard<std::mutex> lk(h->hc_mu); c->busy = false; }
    h->hc_cv.notify_one();
}

// one caller, one scan: queries straight from the caller's memory, results straight into it
static int search_host_direct(ise_index* h, const float* q, long long nq, int k, float* D, long long* I) {
    DeviceGuard gd(h->device);
    ise_index::HostCtx* c = acquire_ctx(h);
    struct Rel { ise_index* h; ise_index::HostCtx* c; ~Rel() { release_ctx(h, c); } } rel{h, c};
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // bounds the workspace (part + multi-pass keys); larger calls loop
    const long long batch = k + 6 <= XPASS_MAX ? 4096 : 1024;
    const size_t qe = (size_t)std::min<long long>(nq, batch) * h->d;
    const size_t oe = (size_t)std::min<long long>(nq, batch) * k;
    if (qe > c->q_elems) {
        if (c->q_dev) (void)hipFree(c->q_dev);
        c->q_dev = nullptr; c->q_elems = 0;
        HIP_TRY(hipMalloc(&c->q_dev, qe * sizeof(float)));
        c->q_elems = qe;
    }
    if (oe > c->out_elems) {
        if (c->D_dev) (void)hipFree(c->D_dev);
        if (c->I_dev) (void)hipFree(c->I_dev);
        c->D_dev = nullptr; c->I_dev = nullptr; c->out_elems = 0;
        HIP_TRY(hipMalloc(&c->D_dev, oe * sizeof(float)));
        HIP_TRY(hipMalloc(&c->I_dev, oe * sizeof(long long)));
        c->out_elems = oe;
    }
    for (long long i0 = 0; i0 < nq; i0 += batch) {
        const long long m = std::min<long long>(batch, nq - i0);
        HIP_TRY(hipMemcpyAsync(c->q_dev, q + (size_t)i0 * h->d, (size_t)m * h->d * sizeof(float), hipMemcpyHostToDevice,
                               c->stream));
        int rc;
        {
            std::lock_guard<std::mutex> lk(h->mu_);
            rc = search_enqueue(h, c->q_dev, m, k, 0u, c->D_dev, c->I_dev, nullptr, c->stream, nullptr);
        }
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(D + (size_t)i0 * k, c->D_dev, (size_t)m * k * sizeof(float), hipMemcpyDeviceToHost,
                               c->stream));
        HIP_TRY(hipMemcpyAsync(I + (size_t)i0 * k, c->I_dev, (size_t)m * k * sizeof(long long), hipMemcpyDeviceToHost,
                               c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return ISE_OK;
}

// Largest number of queries a combined batch holds (0 = every caller runs its own scan).  A scan
// costs the same for 1 or 16 queries and little more for 64: concurrent one-query
// callers -- the reference's serving pattern, one search per HTTP request on a threaded Flask
// (backend/engine.py:55,137) -- share the pass over the index instead of queueing for one each.
static long long host_combine_max() {
    static const long long v = [] {
        const char* e = getenv("ISE_HOST_COMBINE_MAX");
        const long long x = e ? atoll(e) : 64;
        return x < 0 ? 0 : std::min<long long>(x, 1024);
    }();
    return v;
}

// the requests of one combined batch (same k): gather -> one upload -> one search -> one download -> scatter
static int run_combined(ise_index* h, const std::vector<ise_index::HostReq*>& batch, long long total, int k) {
    DeviceGuard gd(h->device);
    ise_index::HostCtx* c = acquire_ctx(h);
    struct Rel { ise_index* h; ise_index::HostCtx* c; ~Rel() { release_ctx(h, c); } } rel{h, c};
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t qe = (size_t)total * h->d, oe = (size_t)total * k;
    if (qe > c->q_elems) {
        if (c->q_dev) (void)hipFree(c->q_dev);
        c->q_dev = nullptr; c->q_elems = 0;
        HIP_TRY(hipMalloc(&c->q_dev, qe * sizeof(float)));
        c->q_elems = qe;
    }
    if (oe > c->out_elems) {
        if (c->D_dev) (void)hipFree(c->D_dev);
        if (c->I_dev) (void)hipFree(c->I_dev);
        c->D_dev = nullptr; c->I_dev = nullptr; c->out_elems = 0;
        HIP_TRY(hipMalloc(&c->D_dev, oe * sizeof(float)));
        HIP_TRY(hipMalloc(&c->I_dev, oe * sizeof(long long)));
        c->out_elems = oe;
    }
    if (qe > c->q_pin_elems) {
        if (c->q_pin) (void)hipHostFree(c->q_pin);
        c->q_pin = nullptr; c->q_pin_elems = 0;
        const size_t want = std::max<size_t>(qe, (size_t)std::max<long long>(host_combine_max(), 1) * h->d);
        HIP_TRY(hipHostMalloc(&c->q_pin, want * sizeof(float), hipHostMallocDefault));
        c->q_pin_elems = want;
    }
    if (oe > c->out_pin_elems) {
        if (c->D_pin) (void)hipHostFree(c->D_pin);
        if (c->I_pin) (void)hipHostFree(c->I_pin);
        c->D_pin = nullptr; c->I_pin = nullptr; c->out_pin_elems = 0;
        const size_t want = std::max<size_t>(oe, (size_t)std::max<long long>(host_combine_max(), 1) * k);
        HIP_TRY(hipHostMalloc(&c->D_pin, want * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostMalloc(&c->I_pin, want * sizeof(long long), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->D_pin_dev, c->D_pin, 0));
        HIP_TRY(hipHostGetDevicePointer((void**)&c->I_pin_dev, c->I_pin, 0));
        c->out_pin_elems = want;
    }
    size_t off = 0;
    for (const auto* r : batch) {
        memcpy(c->q_pin + off * h->d, r->q, (size_t)r->nq * h->d * sizeof(float));
        off += (size_t)r->nq;
    }
    HIP_TRY(hipMemcpyAsync(c->q_dev, c->q_pin, qe * sizeof(float), hipMemcpyHostToDevice, c->stream));
    int rc;
    {
        std::lock_guard<std::mutex> lk(h->mu_);
        // results go straight into the pinned host buffers (mapped, coherent: the last kernel's few hundred bytes
        // travel as posted writes and are visible when the stream has drained) -- two copy launches less per call
        rc = search_enqueue(h, c->q_dev, total, k, 0u, c->D_pin_dev, c->I_pin_dev, nullptr, c->stream, nullptr);
    }
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    off = 0;
    for (auto* r : batch) {
        memcpy(r->D, c->D_pin + off * k, (size_t)r->nq * k * sizeof(float));
        memcpy(r->I, c->I_pin + off * k, (size_t)r->nq * k * sizeof(long long));
        off += (size_t)r->nq;
    }
    return ISE_OK;
}

extern "C" int ise_index_search_host(ise_index_t* h, const float* q, int64_t nq, int k, float* D, int64_t* I) {
    int rc = check_search_args(h, q, nq, k);
    if (rc) return rc;
    if (nq == 0) return ISE_OK;
    if (!D || !I) return fail(ISE_E_INVALID, "output pointer is NULL");
    const long long cmax = host_combine_max();
    // large calls fill their own passes; a large k costs the others more than the shared pass saves
    if (nq > 16 || nq > cmax || k > XPASS_MAX) return search_host_direct(h, q, nq, k, D, (long long*)I);

    ise_index::HostReq r;
    r.q = q; r.nq = nq; r.k = k; r.D = D; r.I = (long long*)I;
    std::unique_lock<std::mutex> lk(h->cq_mu);
    h->cq.push_back(&r);
    while (!r.done) {
        if (h->cq_leaders < ise_index::CQ_LEADERS && !h->cq.empty()) {
            // lead: the head of the queue and whatever behind it asks for the same k, up to cmax queries
            h->cq_leaders++;
            std::vector<ise_index::HostReq*> batch;
            long long total = 0;
            const int bk = h->cq.front()->k;
            while (!h->cq.empty() && h->cq.front()->k == bk && total + h->cq.front()->nq <= cmax) {
                batch.push_back(h->cq.front());
                total += h->cq.front()->nq;
                h->cq.pop_front();
            }
            h->cq_batches++;
            h->cq_requests += batch.size();
            lk.unlock();
            const int brc = run_combined(h, batch, total, bk);
            const std::string berr = brc ? g_err : std::string();
            lk.lock();
            for (auto* b : batch) {
                b->rc = brc;
                if (brc) b->err = berr;
                b->done = true;
                if (b != &r) b->cv.notify_one();
            }
            h->cq_leaders--;
            if (!h->cq.empty()) h->cq.front()->cv.notify_one();  // the next head leads its own batch
        } else {
            r.cv.wait(lk);
        }
    }
    lk.unlock();
    if (r.rc) return fail(r.rc, r.err);  // the message travels to the caller's own thread
    return ISE_OK;
}

extern "C" int ise_index_short_stats(ise_index_t* h, uint64_t* out1) {
    if (!h || !out1) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    out1[0] = h->short_batches;
    return ISE_OK;
}

extern "C" int ise_index_host_stats(ise_index_t* h, uint64_t* out3) {
    if (!h || !out3) return fail(ISE_E_INVALID, "NULL argument");
    {
        std::lock_guard<std::mutex> lk(h->cq_mu);
        out3[0] = h->cq_batches;
        out3[1] = h->cq_requests;
    }
    std::lock_guard<std::mutex> lk(h->mu_);
    out3[2] = h->direct_queries;
    return ISE_OK;
}

extern "C" int ise_index_reserve_workspaces(ise_index_t* h, int64_t nq, int k) {
    int rc = check_search_args(h, h, nq, k);
    if (rc) return rc;
    if (nq == 0) return ISE_OK;
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    rc = prepare_shift_locked(h, h->stream);
    if (rc) return rc;
    ScanPlan pl;
    rc = plan_for_batch(h, nq, k, &pl);
    if (rc) return rc;
    return ensure_workspaces(h, pl, pl.gemm ? std::min<long long>(nq, GEMM_NQ_MAX) : nq);
}

extern "C" int ise_index_stats(ise_index_t* h, uint64_t* out4) {
    if (!h || !out4) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long tmp[4];
    HIP_TRY(hipMemcpy(tmp, h->stats_dev, sizeof(tmp), hipMemcpyDeviceToHost));
#ifdef ISE_ABLATE
    if (getenv("ISE_DEBUG_STAMPS")) {  // dev: merge + rerank phase stamps of block 0 (100 MHz ticks)
        unsigned long long st[16];
        if (hipMemcpy(st, h->stats_dev + 8, sizeof(st), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "merge/rerank stamps (us since kernel entry):");
            for (int i = 1; i < 8; i++) fprintf(stderr, " [%d] %.2f", i, (double)(long long)(st[i] - st[0]) / 100.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    out4[0] = tmp[0];
    out4[1] = tmp[1];
    out4[2] = h->mu_updates;
    out4[3] = h->gemm_chunks;
    return ISE_OK;
}

// centroids per LDS stage: as many as fit ~150 KB, a multiple of 16, at most ASSIGN_CS_MAX
static int assign_stage_rows(const ise_index* h) {
    const size_t per = (size_t)(qs_stride_for(h) + 1) * 4;
    int cs = (int)std::min<size_t>(ASSIGN_CS_MAX, (150 * 1024) / per) / 16 * 16;
    return cs;
}
static size_t assign_lds_bytes(const ise_index* h) {
    return (size_t)assign_stage_rows(h) * (qs_stride_for(h) + 1) * 4;
}

template <int NS, int XT>
static void launch_assign(int grid, size_t lds, hipStream_t st, const AssignParams& ap) {
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&assign_kernel<NS, XT>), LDS_LIMIT);
    hipLaunchKernelGGL((assign_kernel<NS, XT>), dim3(grid), dim3(512), lds, st, ap);
}

// true when the k = 1 assignment kernel applies to this index
static bool assign_supported(const ise_index* h) {
    return h->storage == ISE_STORE_F32 && h->dp <= 512 && h->n > 0 && assign_stage_rows(h) >= 16;
}

extern "C" int ise_index_assign_device(ise_index_t* h, const float* x_dev, int64_t n, float* D_dev, int64_t* I_dev,
                                       void* stream) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && (!x_dev || !I_dev))) return fail(ISE_E_INVALID, "bad argument");
    if (n == 0) return ISE_OK;
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    if (!assign_supported(h))
        return fail(ISE_E_INVALID, "assignment kernel needs a non-empty float32 index with d <= 512");
    {
        int rc = prepare_shift_locked(h, (hipStream_t)stream);
        if (rc) return rc;
    }
    AssignParams ap;
    ap.x = x_dev; ap.cb = (const float*)h->xb; ap.cnorm = h->norms; ap.mu = uses_shift(h) ? h->mu : nullptr; ap.n = n; ap.d = h->d; ap.dp = h->dp;
    ap.cs_stride = qs_stride_for(h); ap.K = (int)h->n; ap.metric = h->metric; ap.cs = assign_stage_rows(h);
    ap.I = (long long*)I_dev; ap.D = D_dev;
    const size_t lds = assign_lds_bytes(h);
    const int ns = h->dp / 16;
    // XT row tiles per wave keep the X fragments at <= 128 VGPRs
    const int xt = ns <= 8 ? 4 : (ns <= 16 ? 2 : 1);
    const long long rows_per_block = 8ll * xt * 16;
    const long long nslabs = (n + rows_per_block - 1) / rows_per_block;
    const int grid = (int)std::min<long long>(nslabs, (long long)h->num_cu);
    hipStream_t st = (hipStream_t)stream;
    switch (ns) {
        case 1: launch_assign<1, 4>(grid, lds, st, ap); break;
        case 2: launch_assign<2, 4>(grid, lds, st, ap); break;
        case 3: launch_assign<3, 4>(grid, lds, st, ap); break;
        case 4: launch_assign<4, 4>(grid, lds, st, ap); break;
        case 8: launch_assign<8, 4>(grid, lds, st, ap); break;
        case 12: launch_assign<12, 2>(grid, lds, st, ap); break;
        case 16: launch_assign<16, 2>(grid, lds, st, ap); break;
        case 20: launch_assign<20, 1>(grid, lds, st, ap); break;
        case 24: launch_assign<24, 1>(grid, lds, st, ap); break;
        case 28: launch_assign<28, 1>(grid, lds, st, ap); break;
        case 32: launch_assign<32, 1>(grid, lds, st, ap); break;
        default: return fail(ISE_E_INVALID, "unsupported padded dimension for the assignment kernel");
    }
    HIP_TRY(hipGetLastError());
    return ISE_OK;
}

extern "C" int ise_merge_keys_device(const uint64_t* keys_dev, int n_lists, int64_t nq, int k, int metric, float* D_dev,
                                     int64_t* I_dev, int device, void* stream) {
    if (!keys_dev || !D_dev || !I_dev) return fail(ISE_E_INVALID, "NULL pointer");
    if (n_lists <= 0 || n_lists > MERGE_LISTS_MAX) return fail(ISE_E_INVALID, "n_lists must be in [1, 1024]");
    if (k <= 0 || k > ISE_MAX_K || nq < 0 || nq > (1ll << 20)) return fail(ISE_E_INVALID, "bad nq / k");
    if (metric != ISE_METRIC_L2 && metric != ISE_METRIC_INNER_PRODUCT) return fail(ISE_E_INVALID, "bad metric");
    if (nq == 0) return ISE_OK;
    DeviceGuard gd(device);
    MergeParams mp;
    mp.lists = (const u64*)keys_dev;
    mp.stride_list = (long long)nq * k;
    mp.stride_qtile = (long long)k; mp.qt = 1;
    mp.n_lists = n_lists; mp.nq = (int)nq; mp.k = k; mp.metric = metric;
    mp.D = D_dev; mp.I = (long long*)I_dev; mp.keys_out = nullptr;
    mp.fl_state = nullptr; mp.fl_list = nullptr; mp.seq = 0; mp.out_by_pos = 0; mp.dbg = nullptr; mp.gate = nullptr;
    if (n_lists <= 64)
        hipLaunchKernelGGL(merge_small_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mp);
    else
        launch_merge<false>((unsigned)nq, 0, (hipStream_t)stream, mp, ExactParams());
    HIP_TRY(hipGetLastError());
    return ISE_OK;
}

extern "C" int ise_normalize_rows_device(float* x_dev, int64_t n, int d, int device, void* stream) {
    if (n < 0 || d <= 0 || (n > 0 && !x_dev)) return fail(ISE_E_INVALID, "bad argument");
    if (n == 0) return ISE_OK;
    if (n >= (1ll << 32)) return fail(ISE_E_INVALID, "too many rows");
    DeviceGuard gd(device);
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x_dev,
                       (long long)n, d);
    HIP_TRY(hipGetLastError());
    return ISE_OK;
}

extern "C" int ise_bovw_histogram_device(const int64_t* labels_dev, const int64_t* offsets_dev, int64_t n_images, int K,
                                         double* out_dev, int device, void* stream) {
    if (n_images < 0 || K <= 0 || (n_images > 0 && (!offsets_dev || !out_dev)))
        return fail(ISE_E_INVALID, "bad argument");
    if (K > HIST_K_MAX) return fail(ISE_E_INVALID, "histogram: at most 16384 bins");
    if (n_images == 0) return ISE_OK;
    if (n_images >= (1ll << 31)) return fail(ISE_E_INVALID, "too many images");
    DeviceGuard gd(device);
    hipLaunchKernelGGL(bovw_histogram_kernel, dim3((unsigned)n_images), dim3(256), (size_t)K * sizeof(unsigned int),
                       (hipStream_t)stream, (const long long*)labels_dev, (const long long*)offsets_dev, K, out_dev);
    HIP_TRY(hipGetLastError());
    return ISE_OK;
}

extern "C" int ise_normalize_rows_host(float* x, int64_t n, int d, int device) {
    if (n < 0 || d <= 0 || (n > 0 && !x)) return fail(ISE_E_INVALID, "bad argument");
    if (n == 0) return ISE_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(ISE_E_NODEVICE, "no HIP device visible: normalize_L2 runs on the GPU");
    DeviceGuard gd(device);
    const long long slab = std::max<long long>(1, (256ll << 20) / ((long long)d * 4));
    float* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)std::min<long long>(slab, n) * d * sizeof(float)));
    int rc = ISE_OK;
    for (long long i0 = 0; i0 < n && rc == ISE_OK; i0 += slab) {
        const long long m = std::min<long long>(slab, n - i0);
        hipError_t e = hipMemcpy(tmp, x + (size_t)i0 * d, (size_t)m * d * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, 0, tmp, m, d);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpy(x + (size_t)i0 * d, tmp, (size_t)m * d * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(ISE_E_HIP, std::string("normalize: ") + hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    return rc;
}
