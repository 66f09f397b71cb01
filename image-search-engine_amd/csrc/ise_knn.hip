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
//   mu     [dp]       float32 shift vector of float32 L2 indexes (mean of the first rows)
//   part   [nqt][nb][16 T][k] u64  per-block sorted candidate lists (workspace slot)
//
// Kernels
//   scan_kernel    ise_scan.hpp    one pass over the index per 16 T queries; HBM-bound
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
#include "ise_merge.hpp"
#include "ise_rows.hpp"

// ---------------------------------------------------------------- host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
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
    float* mu = nullptr;       // [dp] shift vector (fp32 L2 only), zero until shift_set
    bool shift_set = false;    // fixed once: at the first add, or by ise_index_set_shift before it
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
        hipEvent_t done = nullptr;
        bool used = false;
        hipStream_t last_stream = nullptr;  // valid when used
    };
    static constexpr int NWS = 6;
    WorkSlot ws[NWS];
    unsigned ws_next = 0;
    // host-API staging
    hipStream_t stream = nullptr;
    float* q_dev = nullptr;  size_t q_elems = 0;
    float* D_dev = nullptr;  long long* I_dev = nullptr;  size_t out_elems = 0;
    float* h_stage = nullptr;  size_t h_stage_bytes = 0;  // pinned
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
#define KPASS_MAX 32 /* largest k one scan pass selects; larger k runs floor-keyed passes */
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
    if (e != hipSuccess) {
        if (h->stream) (void)hipStreamDestroy(h->stream);
        if (h->mu) (void)hipFree(h->mu);
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
        if (w.done) (void)hipEventDestroy(w.done);
        w = ise_index::WorkSlot();
    }
    if (h->q_dev) (void)hipFree(h->q_dev);
    if (h->D_dev) (void)hipFree(h->D_dev);
    if (h->I_dev) (void)hipFree(h->I_dev);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    h->xb = h->norms = nullptr;
    h->q_dev = h->D_dev = nullptr;
    h->I_dev = nullptr;
    h->h_stage = nullptr;
    h->q_elems = h->out_elems = h->h_stage_bytes = 0;
    h->n = h->cap = 0;
}

extern "C" int ise_index_destroy(ise_index_t* h) {
    if (!h) return ISE_OK;
    {
        DeviceGuard gd(h->device);
        (void)hipDeviceSynchronize();
        free_all(h);
        if (h->mu) (void)hipFree(h->mu);
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
    h->shift_set = false;
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
#define SHIFT_SAMPLE_ROWS 4096 /* the shift vector is the mean of the first rows added (at most this many) */

// rows [0, n_new) have just been written at the start of an empty index: fix the shift
static int fix_shift_from_first_rows(ise_index* h, long long n_new, hipStream_t st) {
    if (!uses_shift(h) || h->shift_set) return ISE_OK;
    const long long rows = std::min<long long>(n_new, SHIFT_SAMPLE_ROWS);
    float* partial = nullptr;
    HIP_TRY(hipMalloc(&partial, (size_t)COLMEAN_GROUPS * h->dp * sizeof(float)));
    hipLaunchKernelGGL(col_sum_kernel, dim3((h->dp + 255) / 256, COLMEAN_GROUPS), dim3(256), 0, st,
                       (const float*)h->xb, rows, h->d, h->dp, partial);
    hipLaunchKernelGGL(col_mean_kernel, dim3((h->dp + 255) / 256), dim3(256), 0, st, partial, rows, h->d, h->dp, h->mu);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // partial is freed below
    (void)hipFree(partial);
    if (e != hipSuccess) return fail(ISE_E_HIP, std::string("shift vector: ") + hipGetErrorString(e));
    h->shift_set = true;
    return ISE_OK;
}

static void launch_norms(ise_index* h, long long row0, long long n, hipStream_t st) {
    const long long nblk = (n + 3) / 4;  // n < 2^32 so nblk fits the 32-bit grid
    if (h->storage == ISE_STORE_BF16)
        hipLaunchKernelGGL(norms_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const __bf16*)h->xb, row0, n,
                           h->dp, h->norms);
    else
        hipLaunchKernelGGL(norms_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)h->xb, row0, n, h->dp,
                           uses_shift(h) ? h->mu : (const float*)nullptr, h->norms);
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
    if (h->n == 0) {
        rc = fix_shift_from_first_rows(h, n, st);
        if (rc) return rc;
    }
    launch_norms(h, h->n, n, st);
    HIP_TRY(hipGetLastError());
    h->n += n;
    return ISE_OK;
}

extern "C" int ise_index_set_shift(ise_index_t* h, const float* mu_host) {
    if (!h || !mu_host) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    if (!uses_shift(h)) return ISE_OK;  // only float32 L2 indexes are shifted
    if (h->n > 0 || h->shift_set) return fail(ISE_E_INVALID, "the shift must be set before the first add");
    DeviceGuard gd(h->device);
    HIP_TRY(hipMemset(h->mu, 0, (size_t)h->dp * sizeof(float)));
    HIP_TRY(hipMemcpy(h->mu, mu_host, (size_t)h->d * sizeof(float), hipMemcpyHostToDevice));
    h->shift_set = true;
    return ISE_OK;
}

extern "C" int ise_index_get_shift(ise_index_t* h, float* mu_host) {
    if (!h || !mu_host) return fail(ISE_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
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
            if (h->n == 0) {
                rc = fix_shift_from_first_rows(h, m, h->stream);
                if (rc) return rc;
            }
            launch_norms(h, h->n, m, h->stream);
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
    if (i0 < 0 || n < 0 || i0 + n > h->n || (n > 0 && !out)) return fail(ISE_E_INVALID, "row range out of bounds");
    if (n == 0) return ISE_OK;
    std::lock_guard<std::mutex> lk(h->mu_);
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

struct ScanPlan {
    int nblocks, tiles_total, tiles_per_block, nqt, ch, kpass, kb, waves, T;
    size_t lds;
};

static bool xchg_enabled() {  // dev knob: ISE_NO_XCHG=1 switches the threshold exchange off
    static const bool on = [] { const char* e = getenv("ISE_NO_XCHG"); return !(e && e[0] == '1'); }();
    return on;
}

// pick (query tiles per pass T, waves per block) for nq queries: the largest T <= 3
// that the batch can use and whose LDS image fits, preferring 8 waves
static int make_plan(const ise_index* h, long long nq, int k, ScanPlan* pl) {
    pl->kpass = k < KPASS_MAX ? k : KPASS_MAX;
    pl->kb = pl->kpass <= 16 ? 16 : 32;
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
    if (!pl->T) return fail(ISE_E_INVALID, "d too large: a 16-query tile must fit the 160 KiB LDS (d <= ~2400)");
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
    int nb = h->num_cu * blocks_per_cu;
    const int max_useful = (pl->tiles_total + pl->waves - 1) / pl->waves;
    if (nb > max_useful) nb = max_useful;
    if (nb < 1) nb = 1;
    if (nb > MERGE_THREADS * MERGE_LPT) nb = MERGE_THREADS * MERGE_LPT;
    pl->tiles_per_block = (pl->tiles_total + nb - 1) / nb;
    if (pl->tiles_per_block < 1) pl->tiles_per_block = 1;
    pl->nblocks = (pl->tiles_total + pl->tiles_per_block - 1) / pl->tiles_per_block;
    if (pl->nblocks < 1) pl->nblocks = 1;
    pl->nqt = (int)((nq + 16 * pl->T - 1) / (16 * pl->T));
    return ISE_OK;
}

// workspace: part [nqt][nb][16][kpass]; for k > kpass additionally
// keys_tmp = keys_all [nq][k] | floor [nq] | pass_keys [nq][kpass]
static int ensure_workspace(ise_index::WorkSlot* w, const ScanPlan& pl, long long nq, int k) {
    if (!w->done) HIP_TRY(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    const size_t need = (size_t)pl.nqt * pl.nblocks * (16 * pl.T) * pl.kpass;
    if (need > w->part_elems) {
        if (w->part) (void)hipFree(w->part);  // hipFree waits for outstanding work
        w->part = nullptr;
        w->part_elems = 0;
        HIP_TRY(hipMalloc(&w->part, need * sizeof(u64)));
        w->part_elems = need;
    }
    const size_t needx = (size_t)pl.nqt * (16 * pl.T) * pl.nblocks;
    if (needx > w->xchg_elems) {
        if (w->xchg) (void)hipFree(w->xchg);
        w->xchg = nullptr;
        w->xchg_elems = 0;
        HIP_TRY(hipMalloc(&w->xchg, needx * sizeof(u64)));
        HIP_TRY(hipMemset(w->xchg, 0xFF, needx * sizeof(u64)));  // tag 0xFFFFFFFF is never issued
        HIP_TRY(hipStreamSynchronize(nullptr));  // the fill is done before any stream's launch reads it
        w->xchg_elems = needx;
        w->xchg_seq = 0;
        if (const char* e = getenv("ISE_XCHG_SEQ_START"))  // test knob: start near the tag wrap
            w->xchg_seq = (uint32_t)strtoul(e, nullptr, 0);
    }
    if (k > pl.kpass) {
        const size_t need2 = (size_t)nq * ((size_t)k + 1 + pl.kpass);
        if (need2 > w->keys_tmp_elems) {
            if (w->keys_tmp) (void)hipFree(w->keys_tmp);
            w->keys_tmp = nullptr;
            w->keys_tmp_elems = 0;
            HIP_TRY(hipMalloc(&w->keys_tmp, need2 * sizeof(u64)));
            w->keys_tmp_elems = need2;
        }
    }
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
    hipEvent_t e0, e1, e2;
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

// enqueue one search batch; outputs (D, I) and/or keys.  Nothing here blocks.
static int search_enqueue(ise_index* h, const float* q_dev, long long nq, int k, uint32_t id_base, float* D_dev,
                          long long* I_dev, u64* keys_out, hipStream_t st, TimedOut* tm) {
    ScanPlan pl;
    int rc = make_plan(h, nq, k, &pl);
    if (rc) return rc;
    ise_index::WorkSlot* w = nullptr;
    bool same_stream = false;
    for (auto& s : h->ws)
        if (s.used && s.last_stream == st) { w = &s; same_stream = true; break; }
    if (!w)
        for (auto& s : h->ws)
            if (!s.used) { w = &s; break; }
    if (!w) w = &h->ws[h->ws_next++ % ise_index::NWS];
    rc = ensure_workspace(w, pl, nq, k);
    if (rc) return rc;
    if (w->used && !same_stream) HIP_TRY(hipStreamWaitEvent(st, w->done, 0));
    struct Release {  // whatever path returns, a later user on another stream waits for this call
        ise_index::WorkSlot* w;
        hipStream_t st;
        ~Release() {
            if (hipEventRecord(w->done, st) == hipSuccess) { w->used = true; w->last_stream = st; }
        }
    } release{w, st};

    ScanParams sp;
    sp.xb = h->xb; sp.norms = h->norms; sp.q = q_dev; sp.mu = h->mu; sp.floor_keys = nullptr; sp.part = w->part;
    sp.n = h->n; sp.d = h->d; sp.dp = h->dp; sp.qs_stride = qs_stride_for(h);
    sp.row_slots = (int)(row_bytes(h) / 16);
    sp.nq = (int)nq; sp.k = pl.kpass; sp.kb = pl.kb; sp.metric = h->metric; sp.id_base = id_base;
    sp.tiles_total = pl.tiles_total; sp.tiles_per_block = pl.tiles_per_block;
    sp.xchg = xchg_enabled() ? w->xchg : nullptr;
    sp.xchg_seq = 0;
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

    const dim3 grid((unsigned)pl.nblocks, (unsigned)pl.nqt);
    if (k <= pl.kpass) {
        mp.D = D_dev; mp.I = I_dev; mp.keys_out = keys_out;
        if ((rc = next_xchg_seq(w, st, &sp.xchg_seq))) return rc;
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
        launch_scan(h, pl.ch, pl.waves, pl.T, grid, pl.lds, st, sp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(MERGE_THREADS), 0, st, mp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
        return ISE_OK;
    }
    // k > KPASS_MAX: passes of KPASS_MAX; a pass only admits keys above the
    // previous pass's last key (keys are totally ordered and unique)
    u64* keys_all = keys_out ? keys_out : w->keys_tmp;      // [nq][k]
    u64* floor_dev = w->keys_tmp + (size_t)nq * k;          // [nq]
    u64* pass_keys = floor_dev + nq;                        // [nq][kpass]
    for (int off = 0; off < k; off += pl.kpass) {
        sp.floor_keys = off ? floor_dev : nullptr;
        mp.D = nullptr; mp.I = nullptr; mp.keys_out = pass_keys;
        if ((rc = next_xchg_seq(w, st, &sp.xchg_seq))) return rc;
        launch_scan(h, pl.ch, pl.waves, pl.T, grid, pl.lds, st, sp);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(MERGE_THREADS), 0, st, mp);
        HIP_TRY(hipGetLastError());
        const int tot = (int)nq * pl.kpass;
        hipLaunchKernelGGL(scatter_pass_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, pass_keys, (int)nq,
                           pl.kpass, keys_all, k, off);
        hipLaunchKernelGGL(floor_from_keys_kernel, dim3(((int)nq + 255) / 256), dim3(256), 0, st, pass_keys, (int)nq,
                           pl.kpass, floor_dev);
        HIP_TRY(hipGetLastError());
    }
    if (D_dev) {
        const long long total = nq * k;
        hipLaunchKernelGGL(decode_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, keys_all, total,
                           h->metric, D_dev, I_dev);
        HIP_TRY(hipGetLastError());
    }
    return ISE_OK;
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
    if (nq == 0 || iters <= 0 || k > KPASS_MAX)
        return fail(ISE_E_INVALID, "timed search needs nq > 0, iters > 0, k <= 32");
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

extern "C" int ise_index_search_host(ise_index_t* h, const float* q, int64_t nq, int k, float* D, int64_t* I) {
    int rc = check_search_args(h, q, nq, k);
    if (rc) return rc;
    if (nq == 0) return ISE_OK;
    if (!D || !I) return fail(ISE_E_INVALID, "output pointer is NULL");
    std::lock_guard<std::mutex> lk(h->mu_);
    DeviceGuard gd(h->device);
    // bounds the workspace (part + multi-pass keys); larger calls loop
    const long long batch = k <= KPASS_MAX ? 4096 : 1024;
    const size_t qe = (size_t)std::min<long long>(nq, batch) * h->d;
    const size_t oe = (size_t)std::min<long long>(nq, batch) * k;
    if (qe > h->q_elems) {
        if (h->q_dev) (void)hipFree(h->q_dev);
        h->q_dev = nullptr; h->q_elems = 0;
        HIP_TRY(hipMalloc(&h->q_dev, qe * sizeof(float)));
        h->q_elems = qe;
    }
    if (oe > h->out_elems) {
        if (h->D_dev) (void)hipFree(h->D_dev);
        if (h->I_dev) (void)hipFree(h->I_dev);
        h->D_dev = nullptr; h->I_dev = nullptr; h->out_elems = 0;
        HIP_TRY(hipMalloc(&h->D_dev, oe * sizeof(float)));
        HIP_TRY(hipMalloc(&h->I_dev, oe * sizeof(long long)));
        h->out_elems = oe;
    }
    for (long long i0 = 0; i0 < nq; i0 += batch) {
        const long long m = std::min<long long>(batch, nq - i0);
        HIP_TRY(hipMemcpyAsync(h->q_dev, q + (size_t)i0 * h->d, (size_t)m * h->d * sizeof(float), hipMemcpyHostToDevice,
                               h->stream));
        rc = search_enqueue(h, h->q_dev, m, k, 0u, h->D_dev, h->I_dev, nullptr, h->stream, nullptr);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(D + (size_t)i0 * k, h->D_dev, (size_t)m * k * sizeof(float), hipMemcpyDeviceToHost,
                               h->stream));
        HIP_TRY(hipMemcpyAsync(I + (size_t)i0 * k, h->I_dev, (size_t)m * k * sizeof(long long), hipMemcpyDeviceToHost,
                               h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
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
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&assign_kernel<NS, XT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr_done = true;
    }
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
    if (n_lists <= 0 || n_lists > MERGE_THREADS * MERGE_LPT) return fail(ISE_E_INVALID, "n_lists must be in [1, 1024]");
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
    if (n_lists <= 64)
        hipLaunchKernelGGL(merge_small_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mp);
    else
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(MERGE_THREADS), 0, (hipStream_t)stream, mp);
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
