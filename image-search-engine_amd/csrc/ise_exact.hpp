// ise_exact.hpp -- the VERIFIER of the exact float32 L2 search: direct-difference re-rank of the
// scan's candidates with a certificate, and the exact fallback scan for the queries it cannot prove.
//
// What Faiss computes for the reference's query shape (one query per call, backend/engine.py:50,55;
// nq < 20 in IndexFlatL2::search [upstream-faiss]) is the per-pair sum (x_i - y_i)^2 -- no
// expansion, no cancellation; the reference's own numpy restatement does the same
// (backend/siamese/test_index.py:62-64).  The streaming scan (ise_scan.hpp) evaluates the expanded
// form on the matrix core, which is fast but loses digits, so for float32 L2 indexes it is used as
// a FILTER only:
//
//   scan    keys every row by lo = s~ - beta (|x-mu|^2 + |y-mu|^2), a rigorous lower bound of the
//           direct-difference value d(x, y) as THIS file computes it, and keeps the kc = k + extra
//           rows of smallest (lo, id) per query.
//   rerank  (rerank_block, fused into the merge kernel) evaluates d for those kc rows, sorts them by
//           (d, id) and checks the certificate  lo_(kc) > d_(k):  every row outside the candidate
//           list has lo >= lo_(kc), hence d >= lo > d_(k), hence cannot enter the top k.  A list
//           with fewer than kc entries holds every admissible row and needs no check.
//   exact   queries whose certificate fails (dense near-ties, far-apart clusters, massive
//           duplicates: wherever beta (|x-mu|^2 + |y-mu|^2) exceeds the neighbour spacing) are put
//           on a per-launch list; exact_scan_kernel -- launched behind every rerank, exiting at
//           once when the list is empty -- recomputes them as a direct-difference scan of the whole
//           index with the same d(), i.e. Faiss's small-batch algorithm itself.
//
// Either way the reported distances are d(x, y) and the ids are the exact top k by (d, id): the
// result is independent of mu, of the grid shape and of which path produced it.
#pragma once
#include "ise_common.hpp"
#include "ise_select.hpp"

// d(x, y) = sum (y_i - x_i)^2 in float32, one wave per row, fixed order: lane l accumulates the
// elements 4l..4l+3 of every 256-element segment in sequence (fmaf chain), then an xor butterfly.
// row: dp floats (zero padded, 16-byte aligned); qs: the query in LDS, zero padded to dp.
// Every lane returns the same value.
template <int R>
__device__ __forceinline__ void exact_l2_rows(const float* const (&rows)[R], const float* qs, int dp, int lane,
                                              float (&out)[R]) {
    float s[R];
#pragma unroll
    for (int r = 0; r < R; r++) s[r] = 0.f;
    for (int j = lane * 4; j < dp; j += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(qs + j);
        f32x4 y[R];
#pragma unroll
        for (int r = 0; r < R; r++) y[r] = *reinterpret_cast<const f32x4*>(rows[r] + j);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const f32x4 t = y[r] - x;
            s[r] = fmaf(t[0], t[0], s[r]);
            s[r] = fmaf(t[1], t[1], s[r]);
            s[r] = fmaf(t[2], t[2], s[r]);
            s[r] = fmaf(t[3], t[3], s[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) out[r] = wave_sum_f32(s[r]);
}

struct ExactParams {
    const float* xb;   // [cap][dp] float32 rows
    const float* q;    // [nq][d]
    long long n;
    int d, dp, nq;
    int k;             // results per query
    int kc;            // candidates per query handed to the rerank (k < kc unless the index is short)
    uint32_t id_base;
    float* D;          // [nq][k] or null
    long long* I;      // [nq][k] or null
    u64* keys_out;     // [nq][k] or null: exact keys ord(d) << 32 | id
    // per-launch list of the queries whose certificate failed: state = launch seq << 32 | count
    u64* fl_state;
    int* fl_list;      // [nq]
    uint32_t seq;
    unsigned long long* stats;  // [0] queries reranked, [1] queries sent to the exact scan
    int force_fail;    // test knob ($ISE_FORCE_EXACT=1): fail every certificate
    // large-batch path: rows with lo > tau[q] never became candidates; the certificate covers them too
    const float* tau_bound;  // [nq] or null
};

// append query q to the launch's fallback list (the counter is tagged with the launch sequence
// number, so it needs no reset between launches)
__device__ __forceinline__ void fallback_list_push(const ExactParams& p, int q) {
    u64 old = __hip_atomic_load(p.fl_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const u64 want = ((uint32_t)(old >> 32) == p.seq) ? old + 1 : (((u64)p.seq << 32) | 1ull);
        if (__hip_atomic_compare_exchange_strong(p.fl_state, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) {
            p.fl_list[(uint32_t)want - 1] = q;
            return;
        }
    }
}

__device__ __forceinline__ void emit_exact(const ExactParams& p, size_t o, u64 key) {
    if (p.keys_out) p.keys_out[o] = key;
    if (p.D) {
        const bool pad = key == KEY_PAD;
        p.D[o] = pad ? FLT_MAX : unord_f32((uint32_t)(key >> 32));
        p.I[o] = pad ? -1ll : (long long)(uint32_t)key;
    }
}

// block-wide bitonic sort of n2 (a power of two) keys in LDS, ascending
__device__ __forceinline__ void block_sort_u64(u64* a, int n2, int tid, int nthreads) {
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int i = tid; i < (n2 >> 1); i += nthreads) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const u64 x = a[lo], y = a[hi];
                if ((x > y) == up) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
    __syncthreads();
}

// LDS image of one rerank block: qs [dp] floats | kin [kc] | kex [n2] (u64, 8-byte aligned)
__host__ __device__ constexpr int rerank_pow2(int kc) {
    int n2 = 1;
    while (n2 < kc) n2 <<= 1;
    return n2;
}
__host__ __device__ constexpr size_t rerank_lds_bytes(int dp, int kc) {
    return (size_t)dp * 4 + (size_t)kc * 8 + (size_t)rerank_pow2(kc) * 8;
}

// The query of a rerank block into LDS (zero padded to dp).  Separate from rerank_block so that the
// fused merge kernel can issue it before the lists are merged.
template <int NT>
__device__ __forceinline__ void rerank_stage_query(const ExactParams& p, int q, unsigned char* smem) {
    float* qs = reinterpret_cast<float*>(smem);
    const float* src = p.q + (size_t)q * p.d;
    for (int j = threadIdx.x; j < p.dp; j += NT) qs[j] = j < p.d ? src[j] : 0.f;
}

// Re-rank the kc candidates of query q (kin: ascending lo-keys, KEY_PAD padded, in LDS behind the
// staged query) by their direct-difference distance; emit the top k; on a failed certificate put q
// on the fallback list.  Called by all NT threads of the block; smem is the block's dynamic LDS
// (rerank_lds_bytes).  keys_in_global: candidates to load first, or null when kin is already filled.
template <int NT>
__device__ __forceinline__ void rerank_block(const ExactParams& p, int q, unsigned char* smem,
                                             const u64* keys_in_global) {
    constexpr int NW = NT / 64;
    float* qs = reinterpret_cast<float*>(smem);
    u64* kin = reinterpret_cast<u64*>(qs + p.dp);
    u64* kex = kin + p.kc;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const int n2 = rerank_pow2(p.kc);
    DBG_STAMP(p.stats ? p.stats + 8 : nullptr, 3);
    if (keys_in_global)
        for (int c = tid; c < p.kc; c += NT) kin[c] = keys_in_global[(size_t)q * p.kc + c];
    __syncthreads();
    // real entries come first (the list is sorted and pads are the largest key)
    int count = 0;
    for (int c0 = 0; c0 < p.kc; c0 += 64) {
        const bool real = (c0 + lane) < p.kc && kin[c0 + lane] != KEY_PAD;
        count += __popcll(__ballot(real));
    }
    DBG_STAMP(p.stats ? p.stats + 8 : nullptr, 4);
    constexpr int R = NW >= 8 ? 2 : 4;  // rows in flight per wave
    for (int c0 = w * R; c0 < n2; c0 += NW * R) {
        if (c0 >= count) {  // nothing real from here on (and no row to read when the index is empty)
            if (lane < R && c0 + lane < n2) kex[c0 + lane] = KEY_PAD;
            continue;
        }
        const float* rows[R];
        u64 kc_[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int c = c0 + r;
            kc_[r] = c < count ? kin[c] : KEY_PAD;
            const long long row = kc_[r] != KEY_PAD ? (long long)((uint32_t)kc_[r] - p.id_base) : 0ll;
            rows[r] = p.xb + (size_t)row * p.dp;
        }
        float dd[R];
        exact_l2_rows<R>(rows, qs, p.dp, lane, dd);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < R; r++)
                if (c0 + r < n2)
                    kex[c0 + r] = kc_[r] != KEY_PAD ? (((u64)ord_f32(dd[r]) << 32) | (uint32_t)kc_[r]) : KEY_PAD;
        }
    }
    DBG_STAMP(p.stats ? p.stats + 8 : nullptr, 5);
    if (n2 <= 64) {  // one wave ranks all pairs: no barrier per stage
        __syncthreads();
        if (w == 0) {
            const u64 mine = lane < n2 ? kex[lane] : KEY_PAD;
            int rk = 0;
            for (int j = 0; j < n2; j++) rk += readlane_u64(mine, j) < mine ? 1 : 0;
            // pads are equal: they keep their slots' worth of the tail in any order
            const u64 padm = __ballot(mine == KEY_PAD) & ((lane < 63 ? (1ull << (lane + 1)) : 0ull) - 1ull);
            if (mine == KEY_PAD) rk += __popcll(padm) - 1;
            wave_lds_fence();
            if (lane < n2) kex[rk] = mine;
        }
        __syncthreads();
    } else {
        block_sort_u64(kex, n2, tid, NT);
    }
    DBG_STAMP(p.stats ? p.stats + 8 : nullptr, 6);
    // certificate: lo of the last candidate strictly above the k-th direct distance.  A list that is
    // not full holds every admissible row of the index.
    bool ok = true;
    if (count == p.kc && p.n > p.kc) {
        const float lo_last = unord_f32((uint32_t)(kin[p.kc - 1] >> 32));
        const float d_k = unord_f32((uint32_t)(kex[p.k - 1] >> 32));
        ok = lo_last > d_k;  // false on NaN
    }
    if (p.tau_bound) {  // rows cut by the admit threshold have lo > tau: tau >= d_(k) keeps them out of the top k
        if (count >= p.k) ok = ok && p.tau_bound[q] >= unord_f32((uint32_t)(kex[p.k - 1] >> 32));
        else ok = ok && (long long)count >= p.n;  // fewer than k candidates and rows left outside: not a result
    }
    if (p.force_fail) ok = false;
    for (int r = tid; r < p.k; r += NT) emit_exact(p, (size_t)q * p.k + r, r < count ? kex[r] : KEY_PAD);
    DBG_STAMP(p.stats ? p.stats + 8 : nullptr, 7);
    if (tid == 0) {
        if (p.stats) atomicAdd(&p.stats[0], 1ull);
        if (!ok) {
            if (p.stats) atomicAdd(&p.stats[1], 1ull);
            fallback_list_push(p, q);
        }
    }
}

// standalone form: candidates [nq][kc] in HBM (the multi-pass path, kc > 32)
static __global__ __launch_bounds__(256) void rerank_kernel(const ExactParams p, const u64* keys_in) {
    extern __shared__ __align__(16) unsigned char smem_rr[];
    rerank_stage_query<256>(p, (int)blockIdx.x, smem_rr);
    rerank_block<256>(p, (int)blockIdx.x, smem_rr, keys_in);
}

