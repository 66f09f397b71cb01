// ise_merge.hpp -- k-way merge of sorted candidate lists.
#pragma once
#include "ise_common.hpp"

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

__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(const MergeParams p) {
    __shared__ u64 wmin[2][MERGE_THREADS / 64];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int k = p.k;
    const u64* base = p.lists + (size_t)(q / p.qt) * p.stride_qtile + (size_t)(q % p.qt) * k;
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
            if (tid == 0) emit_result(p, o, KEY_PAD);
        } else {
            // keys are unique: exactly one list head equals m
#pragma unroll
            for (int e = 0; e < MERGE_LPT; e++)
                if (cur[e] == m) {
                    emit_result(p, o, m);
                    pos[e]++;
                    cur[e] = pos[e] < k ? lst[e][pos[e]] : KEY_PAD;
                }
        }
    }
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
