// ise_merge.hpp -- k-way merge of sorted candidate lists.
#pragma once
#include "ise_common.hpp"
#include "ise_exact.hpp"

// ---------------------------------------------------------------- merge kernel
// One block per query; thread t owns lists t, t+256, ...; k rounds of a
// block-wide argmin over the list heads (keys are unique).
#define MERGE_THREADS 256
#define MERGE_LPT 4 /* lists per thread -> at most 1024 lists */
struct MergeParams {
    const u64* lists;
    long long stride_list;   // elements between consecutive lists
    long long stride_qtile;  // elements between consecutive query tiles
    int qt;                  // queries per tile
    int n_lists, nq, k, metric;
    float* D;        // [nq][k] or null
    long long* I;    // [nq][k] or null
    u64* keys_out;   // [nq][k] or null
    // gated form (the exact fallback path, ise_exact.hpp): block i serves position i of the launch's
    // fallback list -- lists are indexed by position, results by the listed query (or by position
    // when out_by_pos); blocks beyond the list, or of a launch without one, exit at once
    const u64* fl_state;
    const int* fl_list;
    uint32_t seq;
    int out_by_pos;
};

__device__ __forceinline__ void emit_result(const MergeParams& p, size_t o, u64 key) {
    if (p.keys_out) p.keys_out[o] = key;
    if (p.D) {
        const bool pad = key == KEY_PAD;
        const float sc = unord_f32((uint32_t)(key >> 32));
        const bool l2 = p.metric == ISE_METRIC_L2;
        p.D[o] = pad ? (l2 ? FLT_MAX : -FLT_MAX) : (l2 ? sc : -sc);
        p.I[o] = pad ? -1ll : (long long)(uint32_t)key;
    }
}

// RERANK (float32 L2 indexes): the k = kc merged keys are lower-bound keys of the scan; they stay in
// LDS and rerank_block (ise_exact.hpp) turns them into the exact top xp.k.
template <bool RERANK>
__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(const MergeParams p, const ExactParams xp) {
    __shared__ u64 wmin[2][MERGE_THREADS / 64];
    extern __shared__ __align__(16) unsigned char smem_mr[];  // RERANK only: rerank_lds_bytes(dp, kc)
    int q = blockIdx.x, lq = blockIdx.x;
    if (p.fl_state) {
        const u64 st = __hip_atomic_load(p.fl_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(st >> 32) != p.seq || lq >= (int)(uint32_t)st) return;
        q = p.out_by_pos ? lq : p.fl_list[lq];
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int k = p.k;
    u64* kin = RERANK ? reinterpret_cast<u64*>(smem_mr + (size_t)xp.dp * 4) : nullptr;
    const u64* base = p.lists + (size_t)(lq / p.qt) * p.stride_qtile + (size_t)(lq % p.qt) * k;
    const u64* lst[MERGE_LPT];
    int pos[MERGE_LPT];
    u64 cur[MERGE_LPT];
#pragma unroll
    for (int e = 0; e < MERGE_LPT; e++) {
        const int l = tid + e * MERGE_THREADS;
        lst[e] = base + (size_t)(l < p.n_lists ? l : 0) * p.stride_list;
        pos[e] = 0;
        cur[e] = l < p.n_lists ? lst[e][0] : KEY_PAD;
    }
    for (int r = 0; r < k; r++) {
        u64 m = cur[0];
#pragma unroll
        for (int e = 1; e < MERGE_LPT; e++) m = min_u64(m, cur[e]);
        m = wave_min_u64(m);
        if (lane == 0) wmin[r & 1][w] = m;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MERGE_THREADS / 64; i++) m = min_u64(m, wmin[r & 1][i]);
        const size_t o = (size_t)q * k + r;
        if (m == KEY_PAD) {
            if (tid == 0) {
                if (RERANK) kin[r] = KEY_PAD;
                else emit_result(p, o, KEY_PAD);
            }
        } else {
            // keys are unique: exactly one list head equals m
#pragma unroll
            for (int e = 0; e < MERGE_LPT; e++)
                if (cur[e] == m) {
                    if (RERANK) kin[r] = m;
                    else emit_result(p, o, m);
                    pos[e]++;
                    cur[e] = pos[e] < k ? lst[e][pos[e]] : KEY_PAD;
                }
        }
    }
    if (RERANK) rerank_block(xp, q, smem_mr, true, nullptr);  // starts with a block barrier
}

// Up to 64 lists (the all-gathered per-rank lists of the multi-GPU path): one wave per query,
// one list per lane, no LDS and no barriers.
__global__ __launch_bounds__(256) void merge_small_kernel(const MergeParams p) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= p.nq) return;
    const int k = p.k;
    const u64* lst = p.lists + (size_t)(q / p.qt) * p.stride_qtile + (size_t)(q % p.qt) * k +
                     (size_t)(lane < p.n_lists ? lane : 0) * p.stride_list;
    const bool live = lane < p.n_lists;
    int pos = 0;
    u64 cur = live ? lst[0] : KEY_PAD;
    u64 nxt = (live && k > 1) ? lst[1] : KEY_PAD;
    for (int r = 0; r < k; r++) {
        const u64 m = wave_min_u64(cur);
        const size_t o = (size_t)q * k + r;
        if (m == KEY_PAD) {
            if (lane == 0) emit_result(p, o, KEY_PAD);
        } else if (cur == m) {
            emit_result(p, o, m);
            pos++;
            cur = nxt;
            nxt = pos + 1 < k ? lst[pos + 1] : KEY_PAD;
        }
    }
}
