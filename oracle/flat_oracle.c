/*
 * oracle/flat_oracle.c -- CPU restatement of the Faiss IndexFlat small-batch
 * search the reference calls.  TEST INFRASTRUCTURE ONLY: it is built by
 * oracle/Makefile into oracle/_build/libflat_oracle.so and loaded only from
 * tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  The
 * product library (image-search-engine_amd/csrc) never links or loads it.
 *
 * PARITY UNPINNED: the reference holds no tests / golden vectors for this
 * path and the arithmetic lives in PyPI faiss-cpu (unpinned,
 * backend/siamese/requirements.txt:2; ~1.8.0/1.9.0 at the snapshot date),
 * whose source is not under /root/reference.  The published algorithm is
 * restated here [upstream-faiss]; call sites it must serve:
 *   backend/engine.py:55              index.search(x(1,d), k)
 *   backend/siamese/test_index.py:54  index.search(embedding, n_results) (IP)
 *   backend/kmeans_faiss.py:49        index.search(X, 1)
 *   backend/utils.py:303 etc.         faiss.normalize_L2(x)
 *
 * Algorithm restated (Faiss exhaustive_L2sqr_seq / exhaustive_inner_product_seq,
 * the path taken when nq < distance_compute_blas_threshold = 20):
 *   for each query: heap of k entries initialised to the neutral value
 *   (+FLT_MAX, id -1); for each index row j in order: dis = fvec_L2sqr(x, y_j)
 *   (direct sum of (x-y)^2, float32) or fvec_inner_product; the row replaces
 *   the heap top iff it is STRICTLY better than the current k-th best; at the
 *   end the heap is emitted in sorted order, ties by ascending id.
 * The float32 summation order of Faiss's SIMD kernels depends on its build
 * (AVX2: 8 lanes); this restatement accumulates in 8 interleaved float32 lanes
 * and adds them pairwise.  Tests compare with a tolerance, never bitwise.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_METRIC_IP 0
#define ORACLE_METRIC_L2 1

static inline float hsum8(const float* a) {
    return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

/* fvec_L2sqr [upstream-faiss]: sum_i (x_i - y_i)^2 in float32 */
float oracle_fvec_L2sqr(const float* x, const float* y, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int l = 0; l < 8; l++) {
            const float t = x[i + l] - y[i + l];
            acc[l] += t * t;
        }
    float res = hsum8(acc);
    for (; i < d; i++) {
        const float t = x[i] - y[i];
        res += t * t;
    }
    return res;
}

/* fvec_inner_product [upstream-faiss] */
float oracle_fvec_inner_product(const float* x, const float* y, size_t d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= d; i += 8)
        for (int l = 0; l < 8; l++) acc[l] += x[i + l] * y[i + l];
    float res = hsum8(acc);
    for (; i < d; i++) res += x[i] * y[i];
    return res;
}

/* fvec_norm_L2sqr + fvec_renorm_L2 [upstream-faiss]: in place, zero rows untouched,
 * scale = (float)(1.0 / sqrtf(nr)) */
void oracle_renorm_L2(size_t d, size_t nx, float* x) {
#pragma omp parallel for if (nx > 10000)
    for (int64_t i = 0; i < (int64_t)nx; i++) {
        float* xi = x + (size_t)i * d;
        const float nr = oracle_fvec_inner_product(xi, xi, d);
        if (nr > 0) {
            const float inv_nr = (float)(1.0 / sqrtf(nr));
            for (size_t j = 0; j < d; j++) xi[j] *= inv_nr;
        }
    }
}

/* ---- max-heap on (key, id): top is the WORST kept entry ------------------ */
typedef struct {
    float key;   /* L2: distance; IP: -score (negation is exact) */
    int64_t id;
} ent_t;

static inline int worse(ent_t a, ent_t b) { /* a after b in (key,id) order */
    return a.key > b.key || (a.key == b.key && a.id > b.id);
}

static void heap_init(ent_t* h, int k) {
    for (int i = 0; i < k; i++) {
        h[i].key = FLT_MAX;
        h[i].id = -1;
    }
}

static void heap_replace_top(ent_t* h, int k, ent_t e) {
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m;
        if (l >= k) break;
        m = (r < k && worse(h[r], h[l])) ? r : l;
        if (!worse(h[m], e)) break;
        h[i] = h[m];
        i = m;
    }
    h[i] = e;
}

static int ent_cmp(const void* pa, const void* pb) {
    const ent_t a = *(const ent_t*)pa, b = *(const ent_t*)pb;
    /* unfilled slots (id -1, key FLT_MAX) sort last */
    if (a.id < 0 || b.id < 0) return (a.id < 0) - (b.id < 0);
    return worse(a, b) - worse(b, a);
}

static void scan_rows(const float* xq_i, const float* xb, int64_t j0, int64_t j1,
                      int d, int metric, ent_t* h, int k) {
    float thr = h[0].key;
    for (int64_t j = j0; j < j1; j++) {
        const float* y = xb + (size_t)j * d;
        const float s = metric == ORACLE_METRIC_L2 ? oracle_fvec_L2sqr(xq_i, y, d)
                                                   : -oracle_fvec_inner_product(xq_i, y, d);
        if (thr > s) { /* strict: NaN and >= FLT_MAX never enter */
            ent_t e = {s, j};
            heap_replace_top(h, k, e);
            thr = h[0].key;
        }
    }
}

/*
 * IndexFlat{L2,IP}.search restated.  nthreads <= 0: Faiss's own scheme
 * (parallel over queries, min(nq, max threads)).  nthreads > 0: additionally
 * split the index rows into slabs so that every core is busy at small nq; the
 * per-slab heaps are merged under the same (key, id) order, which yields the
 * same result as the sequential scan.  Returns the number of threads used.
 */
int oracle_knn_flat(const float* xb, int64_t n, int d, const float* xq, int64_t nq,
                    int k, int metric, float* D, int64_t* I, int nthreads) {
    const float pad = metric == ORACLE_METRIC_L2 ? FLT_MAX : -FLT_MAX;
    for (int64_t i = 0; i < nq * (int64_t)k; i++) {
        D[i] = pad;
        I[i] = -1;
    }
    if (k <= 0 || nq <= 0) return 0;
    int maxt = 1;
#ifdef _OPENMP
    maxt = omp_get_max_threads();
#endif
    int slabs = 1, nt;
    if (nthreads <= 0) {
        nt = nq < maxt ? (int)nq : maxt;
    } else {
        nt = nthreads;
        slabs = (int)((nt + nq - 1) / nq);
        if (slabs < 1) slabs = 1;
    }
    if (nt < 1) nt = 1;
    const int64_t tasks = nq * slabs;
    ent_t* heaps = (ent_t*)malloc(sizeof(ent_t) * (size_t)tasks * (size_t)k);
    if (!heaps) return -1;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
    for (int64_t t = 0; t < tasks; t++) {
        const int64_t i = t / slabs;
        const int s = (int)(t % slabs);
        const int64_t j0 = n * s / slabs, j1 = n * (s + 1) / slabs;
        ent_t* h = heaps + (size_t)t * k;
        heap_init(h, k);
        scan_rows(xq + (size_t)i * d, xb, j0, j1, d, metric, h, k);
    }
    for (int64_t i = 0; i < nq; i++) {
        ent_t* all = heaps + (size_t)i * slabs * k;
        const int m = slabs * k;
        qsort(all, (size_t)m, sizeof(ent_t), ent_cmp);
        for (int r = 0; r < k && r < m; r++) {
            if (all[r].id < 0) break;
            D[i * k + r] = metric == ORACLE_METRIC_L2 ? all[r].key : -all[r].key;
            I[i * k + r] = all[r].id;
        }
    }
    free(heaps);
    return nt;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
