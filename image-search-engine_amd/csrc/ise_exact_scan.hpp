// ise_exact_scan.hpp -- the exact fallback scan of the float32 L2 search (overview: ise_exact.hpp).
#pragma once
#include "ise_exact.hpp"
#include "ise_merge.hpp"

// ---------------------------------------------------------------- exact fallback scan
// Direct-difference scan of the whole index for the queries on the launch's fallback list -- the
// algorithm Faiss runs for nq < 20 (fvec_L2sqr per pair + a k-heap), restated for the GPU: a block
// owns a contiguous slab of rows, a wave scores XR rows against XQ listed queries at a time with the
// same d() as the rerank, and keeps its k best per query as a sorted list spread over its lanes
// (insertion by ballot rank).  Exits at once when the list is empty (the common case).
//   part: [list position][gridDim.x][kpass] sorted keys per block; the block that finishes LAST merges
//         them (merge_rounds) and emits the results -- the hand-off is the guide's counter form:
//         every wave drains its stores, block barrier, one lane releases at agent scope and adds to
//         the arrival counter; the last arriver acquires at agent scope before any wave of it loads.
//   floor_keys: optional [list position] -- only keys above it enter (k > 32: one exact pass per 32)
#define XQ 4   /* listed queries scored per pass over the rows (the fallback list is worked off in groups of XQ) */
#define XR 4   /* rows a wave scores per step (two when four queries are scored together: registers) */
#define XDEPTH 2 /* row-step loads in flight ahead of the one being scored */
#ifndef XDEPTH1
#define XDEPTH1 5 /* the same for the one-query instantiation (256 VGPRs to spend) */
#endif
struct ExactScanParams {
    const float* xb;
    const float* q;
    long long n;
    int d, dp;
    int kpass;  // <= 32
    uint32_t id_base;
    const u64* fl_state;
    const int* fl_list;
    uint32_t seq;
    const u64* floor_keys;  // [list position] or null
    u64* part;
    long long rows_per_block;
    unsigned int* arrive;   // zero between launches: blocks that have written their lists
    // > 0: the DIRECT search of small batches (nq <= XQ, ise_knn.hip): queries 0 .. direct_n - 1 are the list,
    // the launch is not gated -- this kernel alone is then the whole search, Faiss's nq < 20 algorithm as it stands
    int direct_n;
};

__device__ __forceinline__ u64 shfl_up1_u64(u64 v) {
    const int lo = __shfl_up((int)(uint32_t)v, 1), hi = __shfl_up((int)(uint32_t)(v >> 32), 1);
    return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}

// QN = queries scored together (1, 2 or XQ): the fallback uses XQ, the direct search the batch size.
// The arithmetic of one (row, query) pair does not depend on QN, XR or XDEPTH: per lane an fmaf chain over
// its elements in ascending column order, then wave_sum_f32 -- exact_l2_rows' d() of the re-rank, bit for bit.
template <int QN, bool NT = false>
__global__ __launch_bounds__(256, QN == 1 ? 2 : 4) void exact_scan_kernel(const ExactScanParams p, const MergeParams mp) {
    // one query (the direct search of a single-query batch): two blocks per CU with a deep ring stream fastest
    // (336 us at 1M x 512 against 360 with four blocks and a ring of two); several queries: four blocks per CU
    constexpr int DEPTH = QN == 1 ? XDEPTH1 : XDEPTH;
    constexpr int XRN = QN >= 2 ? 2 : XR;  // rows per step: several queries' partial sums leave room for two rows' loads
    int nfl = p.direct_n;
    if (nfl <= 0) {
        const u64 st = __hip_atomic_load(p.fl_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(st >> 32) != p.seq) return;  // no certificate failed in this launch
        nfl = (int)(uint32_t)st;
    }
    const bool direct = p.direct_n > 0;
    __shared__ u64 wmin[2][4];
    __shared__ int is_last;
    __shared__ u64 fold_stage[4 * 32];  // the last block's selection scratch (block_topk_u64) and its result
    __shared__ u64 fold_out[32];
    extern __shared__ __align__(16) unsigned char smem_xs[];
    float* qs = reinterpret_cast<float*>(smem_xs);            // [QN][dp]
    u64* wl = reinterpret_cast<u64*>(qs + (size_t)QN * p.dp);  // [QN][4 waves][32]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const long long r_begin = (long long)blockIdx.x * p.rows_per_block;
    const long long r_end = min(p.n, r_begin + p.rows_per_block);
    const int kp = p.kpass;
    const int nj = (p.dp + 255) >> 8;  // 256-float column blocks of a row: one 16-byte load per lane and block
    // this wave's row steps: rows r_begin + (w + 4 g) XRN .. + XRN - 1, g = 0 .. G - 1; a step has nj load units
    const long long span = r_end - r_begin - (long long)w * XRN;
    const long long G = span > 0 ? (span + 4 * XRN - 1) / (4 * XRN) : 0;
    const long long U = G * nj;
    for (int g0 = 0; g0 < nfl; g0 += QN) {
        const int ng = min(QN, nfl - g0);
        __syncthreads();  // the previous group's LDS is dead
        for (int i = tid; i < QN * p.dp; i += 256) {
            const int gq = i / p.dp, j = i - gq * p.dp;
            float v = 0.f;
            if (gq < ng && j < p.d) v = p.q[(size_t)(direct ? g0 + gq : p.fl_list[g0 + gq]) * p.d + j];
            qs[i] = v;
        }
        __syncthreads();
        u64 lst[QN], tau[QN], flo[QN];
#pragma unroll
        for (int gq = 0; gq < QN; gq++) {
            lst[gq] = KEY_PAD;
            tau[gq] = TAU0;
            flo[gq] = (p.floor_keys && gq < ng) ? p.floor_keys[g0 + gq] : 0ull;
        }
        // ---- the rows: a ring of DEPTH + 1 load units (XRN rows x one column block each), requested
        // unconditionally DEPTH units ahead (past the end the last unit is read again), so that no control
        // flow sits between a load and its use and the waits stay counted
        f32x4 ring[DEPTH + 1][XRN];
        long long lg = 0;  // position of the next unit to LOAD: row step lg, column block lj
        int lj = 0;
        auto load_unit = [&](f32x4(&y)[XRN]) {
            const long long r0 = r_begin + ((long long)w + 4 * lg) * XRN;
            const int j = (lj << 8) + lane * 4;
#pragma unroll
            for (int r = 0; r < XRN; r++) {
                const long long row = min(r0 + r, r_end - 1);
                y[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (j < p.dp) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(p.xb + (size_t)row * p.dp + j);
                    y[r] = NT ? __builtin_nontemporal_load(src) : *src;
                }
            }
            int nlj = lj + 1;
            long long nlg = lg;
            if (nlj == nj) { nlj = 0; nlg = lg + 1; }
            if (nlg < G) { lg = nlg; lj = nlj; }
        };
        if (U > 0) {
#pragma unroll
            for (int i = 0; i < DEPTH; i++) load_unit(ring[i]);
        }
        float s[QN][XRN];
#pragma unroll
        for (int gq = 0; gq < QN; gq++)
#pragma unroll
            for (int r = 0; r < XRN; r++) s[gq][r] = 0.f;
        long long cg = 0;  // position of the unit being SCORED
        int cj = 0;
        long long u = 0;
        while (u < U) {
#pragma unroll
            for (int slot = 0; slot <= DEPTH; slot++) {
                if (u < U) {
                    load_unit(ring[(slot + DEPTH) % (DEPTH + 1)]);
                    __builtin_amdgcn_sched_barrier(0);  // keep the request ahead of the arithmetic
                    const int j = (cj << 8) + lane * 4;
                    if (j < p.dp) {
#pragma unroll
                        for (int gq = 0; gq < QN; gq++) {
                            const f32x4 x = *reinterpret_cast<const f32x4*>(qs + (size_t)gq * p.dp + j);
#pragma unroll
                            for (int r = 0; r < XRN; r++) {
                                const f32x4 t = ring[slot][r] - x;
                                s[gq][r] = fmaf(t[0], t[0], s[gq][r]);
                                s[gq][r] = fmaf(t[1], t[1], s[gq][r]);
                                s[gq][r] = fmaf(t[2], t[2], s[gq][r]);
                                s[gq][r] = fmaf(t[3], t[3], s[gq][r]);
                            }
                        }
                    }
                    if (cj == nj - 1) {  // the row step is complete: reduce, insert
                        const long long r0 = r_begin + ((long long)w + 4 * cg) * XRN;
#pragma unroll
                        for (int gq = 0; gq < QN; gq++)
#pragma unroll
                            for (int r = 0; r < XRN; r++) {
                                const float dd = wave_sum_f32(s[gq][r]);  // the same on every lane
                                s[gq][r] = 0.f;
                                const u64 kj = ((u64)ord_f32(dd) << 32) | (uint32_t)((uint32_t)(r0 + r) + p.id_base);
                                const bool ok = (r0 + r < r_end) && gq < ng && dd < FLT_MAX && kj < tau[gq] && kj > flo[gq];
                                if (ok) {  // wave-uniform
                                    const int pos = __popcll(__ballot(lst[gq] < kj));
                                    const u64 up = shfl_up1_u64(lst[gq]);
                                    lst[gq] = lane < pos ? lst[gq] : (lane == pos ? kj : up);
                                    if (lane >= kp) lst[gq] = KEY_PAD;
                                    const u64 kth = readlane_u64(lst[gq], kp - 1);
                                    tau[gq] = kth == KEY_PAD ? TAU0 : kth;
                                }
                            }
                        cj = 0;
                        cg++;
                    } else {
                        cj++;
                    }
                    u++;
                }
            }
        }
#pragma unroll
        for (int gq = 0; gq < QN; gq++)
            if (lane < 32) wl[(gq * 4 + w) * 32 + lane] = lst[gq];
        __syncthreads();
        if (w < ng) {  // wave w folds the four wave lists of listed query g0 + w
            u64 kk[2];
            kk[0] = wl[(w * 4) * 32 + lane];        // waves 0, 1
            kk[1] = wl[(w * 4 + 2) * 32 + lane];    // waves 2, 3
            u64* out = p.part + ((size_t)(g0 + w) * gridDim.x + blockIdx.x) * kp;
            u64 kth_unused;
            const int nw = wave_select<2>(kk, 128, kp, out, &kth_unused);
            if (lane >= nw && lane < kp) out[lane] = KEY_PAD;
        }
    }
    // ---- arrive; the last block merges every listed query's per-block lists and emits
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have left
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(p.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = prev == gridDim.x - 1 ? 1 : 0;
        if (is_last) {
            __hip_atomic_store(p.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    // Up to 4096 keys in all (a small index: the reference's own ~1000 rows give 63 lists of k = 20): the lists
    // are contiguous, [blocks][kp] -- one quickselect over all of them (two barriers) instead of kp block-wide
    // argmin rounds of ~1 us each (1000 x 2048, k = 20: the call's device time 33 -> see DESIGN.md 4.2-5)
    const bool small_fold = (long long)gridDim.x * kp <= 4096 && mp.stride_list == kp && kp <= 32;
    for (int i = 0; i < nfl; i++) {
        const u64* base = mp.lists + (size_t)i * mp.stride_qtile;
        const int q = (mp.out_by_pos || direct) ? i : p.fl_list[i];
        if (small_fold) {
            block_topk_u64(base, (int)gridDim.x * kp, kp, fold_stage, fold_out);
            if (tid < kp) emit_result(mp, (size_t)q * kp + tid, fold_out[tid]);
        } else {
            merge_rounds<256, 4>(mp, base, wmin, [&](int r, u64 key) { emit_result(mp, (size_t)q * kp + r, key); });
        }
        __syncthreads();
    }
}

// k > 32 on the exact path: copy one exact pass's keys [list position][kp] into the final outputs at
// column `off` and keep each query's last key as the floor of the next pass
__global__ __launch_bounds__(256) void exact_scatter_kernel(const ExactParams p, const u64* pass_keys, int kp, int off,
                                                            u64* floor_out) {
    const u64 st = __hip_atomic_load(p.fl_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(st >> 32) != p.seq) return;
    const int nfl = (int)(uint32_t)st;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfl * kp) return;
    const int pos = i / kp, r = i - pos * kp;
    const u64 key = pass_keys[i];
    if (off + r < p.k) emit_exact(p, (size_t)p.fl_list[pos] * p.k + off + r, key);
    if (r == kp - 1) floor_out[pos] = key;  // KEY_PAD when the index ran out: later passes admit nothing
}
