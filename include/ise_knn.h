/*
 * include/ise_knn.h -- C ABI of the MI355X (gfx950) brute-force kNN library.
 *
 * This is the drop-in boundary for the one hot path of
 * ManuelZ/image-search-engine: the arithmetic the reference delegates to the
 * Faiss IndexFlat objects.  The reference has no FFI of its own (pure Python
 * over the Faiss SWIG module), so every entry point names the
 * reference call site whose native work it replaces.  Plain pointers and
 * sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (ISE_E_*); nothing
 *     throws across the ABI; ise_last_error() returns a thread-local message.
 *   - the caller owns every buffer it passes; the index owns a private copy of
 *     added rows (reference contract of index.add, backend/utils.py:327).
 *   - "device" pointers are HIP device pointers on the index's device;
 *     `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     *_device entry points only enqueue work; *_host entry points block.
 *   - distances follow Faiss: METRIC_L2 = SQUARED L2, ascending;
 *     METRIC_INNER_PRODUCT = inner product, descending; ties by ascending id;
 *     a row enters a result only if strictly better than +-FLT_MAX, so
 *     unfilled slots come back as id -1 / distance +-FLT_MAX.
 *   - ids are row numbers in insertion order (+ id_base); a (sharded) index
 *     holds fewer than 2^32 rows.
 *   - concurrent *_host calls on one handle are safe and overlap on the GPU (each runs on one of
 *     a few internal contexts; the handle is locked only while kernels are enqueued);
 *     *_device calls on one handle may target different streams (the handle
 *     rotates through a few workspaces and orders their reuse with events, so
 *     independent batches on different streams overlap on the GPU); results of
 *     a call are ordered after it on its own stream only.
 */
#ifndef ISE_KNN_H
#define ISE_KNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISE_METRIC_INNER_PRODUCT 0 /* faiss.METRIC_INNER_PRODUCT */
#define ISE_METRIC_L2 1            /* faiss.METRIC_L2 */

#define ISE_STORE_F32 0  /* index rows kept as float32: exact results */
#define ISE_STORE_BF16 1 /* index rows (and queries) rounded to bf16, fp32 accumulation: approximate,
                            half the HBM traffic (BASELINE config 5: cosine over normalised rows) */

#define ISE_OK 0
#define ISE_E_INVALID -1  /* bad argument (shape, k, NULL) */
#define ISE_E_HIP -2      /* HIP runtime error (message has the hipError string) */
#define ISE_E_NOMEM -3    /* host or device allocation failed */
#define ISE_E_NODEVICE -4 /* no usable gfx950 device */

#define ISE_MAX_K 2048 /* largest k accepted by the search entry points */

typedef struct ise_index ise_index_t;

/* library / device ------------------------------------------------------- */
int ise_version(void);
const char* ise_last_error(void);
int ise_device_count(int* count);
/* name of device `device` into buf (NUL-terminated), e.g. "gfx950:sramecc+:xnack-" */
int ise_device_arch(int device, char* buf, int buflen);

/* index lifetime: replaces faiss.IndexFlatL2(d) / faiss.IndexFlatIP(d)
 * (backend/utils.py:302,306; backend/siamese/siamese_pt/create_index.py:40). */
int ise_index_create(ise_index_t** out, int d, int metric, int device);
/* same with an explicit row storage type (ISE_STORE_*); not a Faiss IndexFlat feature */
int ise_index_create_ex(ise_index_t** out, int d, int metric, int device, int storage);
int ise_index_destroy(ise_index_t* h);
int ise_index_reset(ise_index_t* h); /* drop all rows, keep d/metric */
int ise_index_info(const ise_index_t* h, int* d, int* metric, int64_t* ntotal, int* device);

/* index.add(x) (backend/utils.py:327): append n rows of d float32, copied.
 * Also computes the per-row squared norms the L2 search uses. */
int ise_index_add_host(ise_index_t* h, const float* x, int64_t n);
int ise_index_add_device(ise_index_t* h, const float* x_dev, int64_t n, void* stream);

/* Float32 L2 indexes are searched EXACTLY: the streaming scan evaluates the expanded form
 * |x-mu|^2 + |y-mu|^2 - 2 (x-mu).(y-mu) around a shift vector mu (the column mean of the rows,
 * refreshed at the first search after the index has grown by a quarter) only as a filter keyed by
 * a rigorous lower bound; the candidates are re-evaluated as sum (x_i - y_i)^2 -- what Faiss's
 * IndexFlatL2 computes for the reference's one-query searches (backend/engine.py:55) -- and a
 * query the filter cannot certify is recomputed by a direct-difference scan of the whole index
 * (csrc/ise_exact.hpp).  Ids and distances therefore do not depend on mu; it only decides how
 * often the slower path runs.  set_shift pins mu (no automatic refresh) -- e.g. to give the shards
 * of one logical index the same one; get_shift returns the current one.  mu: d float32 on the
 * host.  No-ops for inner-product and bf16 indexes. */
int ise_index_set_shift(ise_index_t* h, const float* mu_host);
int ise_index_get_shift(ise_index_t* h, float* mu_host);

/* Counters of the exact float32 L2 path since the index was created:
 *   out4[0] queries re-ranked, out4[1] queries whose certificate failed and that were recomputed by
 *   the exact scan, out4[2] refreshes of the shift vector, out4[3] query chunks (<= 1024 queries) that
 *   took the large-batch GEMM-shaped path (nq >= 64).  Blocks. */
int ise_index_stats(ise_index_t* h, uint64_t* out4);

/* Short indexes (a batch of <= 16 queries, k + spare candidates <= 32, at most 32 row tiles of 16 rows per
 * block of the grid: <= 262k rows on an MI355X) are scanned by short_scan_kernel (csrc/ise_short_scan.hpp: the
 * rows' scores are dumped to LDS and selected once per block, no boot and no thresholds in the stream): the
 * reference's own regime, ~1 k images and one query per request (backend/utils.py:309-310,
 * backend/engine.py:50-55).  out1[0] = batches scanned that way ($ISE_NO_SHORT=1: none; same bits). */
int ise_index_short_stats(ise_index_t* h, uint64_t* out1);

/* Test / rehearsal knobs ($ISE_FORCE_EXACT, $ISE_NO_DIRECT, $ISE_NO_SHORT, $ISE_SHORT_TPB_MAX,
 * $ISE_DIRECT_SHORT_MAX_TILES) are read from the environment when the library is first used and again when
 * this is called -- never inside a search. */
int ise_refresh_env_knobs(void);

/* Size every internal workspace for batches of nq queries and k results now (device allocations,
 * fills and the shift refresh otherwise happen inside the first search of that shape), so that a
 * serving loop is allocation-free from its first batch on.  Blocks. */
int ise_index_reserve_workspaces(ise_index_t* h, int64_t nq, int k);

/* copy rows [i0, i0+n) back to host as n x d float32 (used by write_index,
 * backend/indexer.py:59). */
int ise_index_reconstruct_host(ise_index_t* h, int64_t i0, int64_t n, float* out);

/* index.search(x, k) -> (D, I) (backend/engine.py:55,
 * backend/siamese/test_index.py:54, backend/kmeans_faiss.py:49).
 * q: nq x d float32; D: nq x k float32; I: nq x k int64. */
int ise_index_search_host(ise_index_t* h, const float* q, int64_t nq, int k,
                          float* D, int64_t* I);
/* Thread-safe, and concurrent small calls SHARE a pass over the index: calls with nq <= 16 and
 * k <= 32 that arrive while others are waiting or running are run together as one batch of up to
 * $ISE_HOST_COMBINE_MAX (default 64; 0 = never) queries of the same k, each caller getting exactly
 * the rows it would have got alone.  This is the reference's serving pattern -- one query per HTTP
 * request on a threaded Flask (backend/engine.py:55,137) -- where a scan costs the same for 1 or 16
 * queries.  ise_index_host_stats: out3[0] = batches run that way, out3[1] = calls they served,
 * out3[2] = queries (of any entry point) answered by the direct small-batch scan: float32 L2 batches of
 * up to 4 queries with k <= 32 run Faiss's nq < 20 algorithm as it stands -- one direct-difference scan with
 * a k-best list, no filter in front ($ISE_NO_DIRECT=1 sends them through the filtered path; same bits). */
int ise_index_host_stats(ise_index_t* h, uint64_t* out3);
int ise_index_search_device(ise_index_t* h, const float* q_dev, int64_t nq, int k,
                            float* D_dev, int64_t* I_dev, void* stream);

/* Shard-local search for the multi-GPU path: writes nq x k
 * packed candidates, sorted best-first, suitable for one all-gather:
 *   key = (order-preserving uint32 image of the score) << 32 | (row + id_base)
 * with unfilled slots = 0xFFFFFFFFFFFFFFFF.  For inner product the score
 * image is taken of -score so that ascending key order is best-first for
 * both metrics. */
int ise_index_search_keys_device(ise_index_t* h, const float* q_dev, int64_t nq, int k,
                                 uint32_t id_base, uint64_t* keys_dev, void* stream);

/* Merge n_lists sorted candidate lists per query (layout [n_lists][nq][k],
 * e.g. the all-gathered output of ise_index_search_keys_device over ranks)
 * into D (nq x k float32) and I (nq x k int64).  Needs no index handle;
 * `device` selects the GPU the pointers live on. */
int ise_merge_keys_device(const uint64_t* keys_dev, int n_lists, int64_t nq, int k,
                          int metric, float* D_dev, int64_t* I_dev, int device, void* stream);

/* The exchange step of the row-sharded search (the reference itself never shards:
 * one in-RAM IndexFlat, backend/utils.py:327): ONE all-gather of every rank's packed candidates,
 * issued by RCCL on the caller's stream between the shard scans and the merge -- no host
 * synchronisation, no framework in between.
 *   ise_comm_precheck    everything ise_comm_create needs that can be checked WITHOUT the other ranks
 *                        (librccl resolves with every symbol, `device` exists and can be made current):
 *                        the caller agrees on the outcome across ranks BEFORE any rank enters the
 *                        rendezvous of ise_comm_create, where a missing peer would block the others
 *   ise_comm_unique_id   rank 0 draws the 128-byte id of a new communicator; the caller hands it to
 *                        the other ranks (any channel; sharded.py uses one torch.distributed broadcast)
 *   ise_comm_create      every rank, collectively: join as `rank` of `world`, one GPU per rank
 *   ise_comm_allgather_keys  recv[r * count + i] = rank r's send[i] on every rank (count uint64 per
 *                        rank, device pointers); enqueued on `stream`, returns at once
 * librccl is resolved at run time, so the library loads without it; these entry points then
 * return ISE_E_NODEVICE. */
typedef struct ise_comm ise_comm_t;
int ise_comm_precheck(int device);
int ise_comm_unique_id(void* id128);
int ise_comm_create(ise_comm_t** out, const void* id128, int world, int rank, int device);
int ise_comm_allgather_keys(ise_comm_t* c, const uint64_t* send_dev, uint64_t* recv_dev,
                            int64_t count, void* stream);
int ise_comm_destroy(ise_comm_t* c);

/* index.search(X, 1) for MANY rows against a SMALL index: nearest-centroid assignment,
 * FaissKMeans.transform (backend/kmeans_faiss.py:46-50; BASELINE config 4).  X: n x d
 * float32 on the device; I: n int64 (row of the best index entry, -1 if none); D: n
 * float32 or NULL (squared L2 / inner product of the best entry).  MFMA-bound GEMM-shaped
 * kernel; needs a float32 index with d <= 512 (otherwise use ise_index_search_*). */
int ise_index_assign_device(ise_index_t* h, const float* x_dev, int64_t n, float* D_dev,
                            int64_t* I_dev, void* stream);

/* faiss.normalize_L2(x) (backend/utils.py:303, backend/engine.py:53,
 * backend/siamese/test_index.py:53, siamese_pt/create_index.py:57,
 * siamese_tf/create_index.py:54):
 * in-place row L2 normalisation of n x d float32, zero rows untouched. */
int ise_normalize_rows_device(float* x_dev, int64_t n, int d, int device, void* stream);
int ise_normalize_rows_host(float* x, int64_t n, int d, int device);

/* The visual-word histogram of BOVW.transform (backend/bag_of_visual_words.py:98-106): for
 * every image i, np.histogram(labels[offsets[i] : offsets[i+1]], bins=K) -- K equal-width bins
 * between that image's own smallest and largest label, as numpy computes it when no range is
 * given.  labels: int64 ids from the k = 1 assignment (values in [0, 2^53)); offsets: n_images
 * + 1 non-decreasing int64 row offsets; out: n_images x K float64 counts (the reference's
 * np.zeros((n, K)) array).  An image without rows gives a zero row.  K <= 16384. */
int ise_bovw_histogram_device(const int64_t* labels_dev, const int64_t* offsets_dev,
                              int64_t n_images, int K, double* out_dev, int device, void* stream);

/* measurement hook for bench.py: run one search batch `iters` times on `stream` and return
 * the average duration of the scan kernel (all filter passes when k needs several) and of
 * everything behind it (merge + exact re-rank + the gated exact-scan launches) in milliseconds,
 * measured with hipEvents recorded on that stream around the kernels.  Results land in
 * D_dev / I_dev as for ise_index_search_device. */
int ise_index_search_timed_device(ise_index_t* h, const float* q_dev, int64_t nq, int k,
                                  float* D_dev, int64_t* I_dev, void* stream, int iters,
                                  float* scan_ms_avg, float* merge_ms_avg);

#ifdef __cplusplus
}
#endif
#endif /* ISE_KNN_H */
