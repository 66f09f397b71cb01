// ise_merge.hpp -- k-way merge of sorted candidate lists.
#pragma once
#include "ise_common.hpp"
#include "ise_exact.hpp"

// ---------------------------------------------------------------- merge kernel
// One block per query over up to 1024 sorted lists (keys are unique).
#define MERGE_THREADS 512   /* merge_kernel: 8 waves, lists t and t + 512 per thread */
#define MERGE_LISTS_MAX 1024
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
    unsigned long long* dbg;  // dev builds: stamp buffer or null
    const unsigned int* gate;  // optional: the launch does nothing unless *gate != 0
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

// The lists of one thread: an NT-thread block deals the lists ROUND-ROBIN over its waves -- lane i of wave w owns
// lists i NW + w, + NT, ... (LPT of them) -- so that however few lists there are, every wave holds its share of
// them: merge_waves below is fast when no wave holds more than MERGE_PRE of the k winners, and with list t on
// thread t the 188 lists of a 6000-row index sat on three waves (17 % of the queries took the slow path, the
// merge 11.8-13.9 us instead of 7.7).
// Every list keeps its head AND the element behind it in registers, and the element after that is
// requested the moment a list wins: the round trip to L2 / HBM stays off the round-to-round path.
template <int NT, int LPT>
struct ListHeads {
    const u64* lst[LPT];
    int pos[LPT];
    u64 cur[LPT], nxt[LPT];
    __device__ __forceinline__ void load(const MergeParams& p, const u64* base) {
#pragma unroll
        for (int e = 0; e < LPT; e++) {
            constexpr int NW = NT / 64;
            const int l = ((int)threadIdx.x & 63) * NW + ((int)threadIdx.x >> 6) + e * NT;
            const bool live = l < p.n_lists;
            lst[e] = base + (size_t)(live ? l : 0) * p.stride_list;
            pos[e] = 0;
            cur[e] = live ? lst[e][0] : KEY_PAD;
            nxt[e] = (live && p.k > 1) ? lst[e][1] : KEY_PAD;
        }
    }
    __device__ __forceinline__ u64 best() const {
        u64 m = cur[0];
#pragma unroll
        for (int e = 1; e < LPT; e++) m = min_u64(m, cur[e]);
        return m;
    }
    // the list whose head is m (if this thread owns it) moves on; true for the owner
    __device__ __forceinline__ bool advance(u64 m, int k) {
        bool mine = false;
#pragma unroll
        for (int e = 0; e < LPT; e++)
            if (cur[e] == m) {
                mine = true;
                pos[e]++;
                cur[e] = nxt[e];
                nxt[e] = pos[e] + 1 < k ? lst[e][pos[e] + 1] : KEY_PAD;
            }
        return mine;
    }
};

// The k smallest keys of the lists, in order: k rounds of a block-wide argmin over the list heads
// (any k).  emit(r, key) is called by exactly one thread per round (tid 0 for the KEY_PAD tail).
template <int NT, int LPT, typename Emit>
__device__ __forceinline__ void merge_rounds(const MergeParams& p, const u64* base, u64 (*wmin)[NT / 64], Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const int k = p.k;
    ListHeads<NT, LPT> h;
    h.load(p, base);
    for (int r = 0; r < k; r++) {
        u64 m = wave_min_u64(h.best());
        if (lane == 0) wmin[r & 1][w] = m;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NT / 64; i++) m = min_u64(m, wmin[r & 1][i]);
        if (m == KEY_PAD) {
            if (tid == 0) emit(r, KEY_PAD);
        } else if (h.advance(m, k)) {
            emit(r, m);
        }
    }
}

// k <= 32, the shape every search batch has: two block barriers instead of k.  Measured behind a
// 1M-row scan (dev-build stamps, scripts/merge_stamp_probe.py): k = 16 block-wide rounds cost ~1 us
// each; wave-local rounds (DPP argmin, no LDS, no barrier) ~0.4 us each at 4 lists per lane.  So:
//   A. every one of the 8 waves merges the lists of its lanes by only MERGE_PRE = 8 wave-local rounds;
//   B. the 64 collected keys are ranked by all waves together; the k smallest go to out in order
//      (T = the k-th of them);
//   C. a wave is DONE when its 8th key is >= T (everything it has not collected is larger still) or
//      its lists ran out.  With 500 lists of independent rows the k best are spread over the waves
//      (about 2 each) and every wave is done; otherwise -- rare -- all waves go on to k rounds and the
//      selection runs over 8 x 32 keys.
// out: LDS [k], sorted, KEY_PAD padded.
#define MERGE_FAST_K 40 /* >= the most keys one scan pass selects (36) */
#define MERGE_PRE 8
struct MergeFastScratch {
    u64 wl[MERGE_THREADS / 64][MERGE_FAST_K];
    int rank[64];
    int need_full;
};
__device__ __forceinline__ void merge_waves(const MergeParams& p, const u64* base, MergeFastScratch& s, u64* out) {
    constexpr int NW = MERGE_THREADS / 64, LPT = MERGE_LISTS_MAX / MERGE_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const int k = p.k;
    ListHeads<MERGE_THREADS, LPT> h;
    h.load(p, base);
    DBG_STAMP(p.dbg, 1);
    u64 mine = KEY_PAD;  // lane r keeps the wave's r-th smallest key
    const int pre = k < MERGE_PRE ? k : MERGE_PRE;
    for (int r = 0; r < pre; r++) {
        const u64 m = wave_min_u64(h.best());
        if (lane == r) mine = m;
        if (m != KEY_PAD) (void)h.advance(m, k);
    }
    if (lane < MERGE_PRE) s.wl[w][lane] = mine;
    if (tid < 64) s.rank[tid] = 0;
    if (tid == 0) s.need_full = 0;
    DBG_STAMP(p.dbg, 2);
    __syncthreads();
    // rank of each of the 64 collected keys among them, by all 8 waves: lane i holds key i, wave w
    // counts the keys 8w..8w+7 below it (a single wave ranking 64 keys costs ~3 us of serial issue)
    const u64 mykey = s.wl[lane >> 3][lane & 7];
    {
        int part = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) part += readlane_u64(mykey, 8 * w + j) < mykey ? 1 : 0;
        if (part) atomicAdd(&s.rank[lane], part);
    }
    __syncthreads();
    if (w == 0) {
        const bool real = mykey != KEY_PAD;  // real keys are unique: their ranks are exact
        const int rk = s.rank[lane];
        const int nreal = __popcll(__ballot(real));
        const int nw = nreal < k ? nreal : k;
        if (real && rk < k) out[rk] = mykey;
        if (lane >= nw && lane < k) out[lane] = KEY_PAD;
        const u64 hit = __ballot(real && rk == k - 1);
        const u64 T = hit ? readlane_u64(mykey, __ffsll((long long)hit) - 1) : KEY_PAD;
        // lanes 0..7: wave `lane`'s last collected key
        const u64 last = pre == MERGE_PRE ? s.wl[lane & 7][MERGE_PRE - 1] : KEY_PAD;
        const bool undone = lane < NW && pre < k && last < T;
        if (__ballot(undone) != 0ull && lane == 0) s.need_full = 1;
    }
    __syncthreads();
    if (s.need_full) {  // block-uniform, rare
        for (int r = pre; r < k; r++) {
            const u64 m = wave_min_u64(h.best());
            if (lane == r) mine = m;
            if (m != KEY_PAD) (void)h.advance(m, k);
        }
        __syncthreads();  // wave 0 has read wl
        if (lane < MERGE_FAST_K) s.wl[w][lane] = mine;  // lanes >= k hold KEY_PAD
        __syncthreads();
        if (w == 0) {
            constexpr int KPLM = (NW * MERGE_FAST_K + 63) / 64;
            static_assert(NW * MERGE_FAST_K % 64 == 0, "the waves' lists fill whole registers");
            u64 kk[KPLM];
#pragma unroll
            for (int e = 0; e < KPLM; e++) kk[e] = (&s.wl[0][0])[lane + 64 * e];
            u64 kth_unused;
            const int nw = wave_select<KPLM>(kk, NW * MERGE_FAST_K, k, out, &kth_unused);
            if (lane >= nw && lane < k) out[lane] = KEY_PAD;
        }
        __syncthreads();
    }
}

// RERANK (float32 L2 indexes): the k = kc merged keys are lower-bound keys of the scan; they stay in
// LDS and rerank_block (ise_exact.hpp) turns them into the exact top xp.k.
template <bool RERANK>
__global__ __launch_bounds__(MERGE_THREADS) void merge_kernel(const MergeParams p, const ExactParams xp) {
    constexpr int LPT = MERGE_LISTS_MAX / MERGE_THREADS;
    __shared__ u64 wmin[2][MERGE_THREADS / 64];
    __shared__ MergeFastScratch fast;
    __shared__ u64 res[MERGE_FAST_K];
    extern __shared__ __align__(16) unsigned char smem_mr[];  // RERANK only: rerank_lds_bytes(dp, kc)
    if (p.gate && *p.gate == 0u) return;
    int q = blockIdx.x, lq = blockIdx.x;
    if (p.fl_state) {
        const u64 st = __hip_atomic_load(p.fl_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(st >> 32) != p.seq || lq >= (int)(uint32_t)st) return;
        q = p.out_by_pos ? lq : p.fl_list[lq];
    }
    const int k = p.k;
    const u64* base = p.lists + (size_t)(lq / p.qt) * p.stride_qtile + (size_t)(lq % p.qt) * k;
    if (RERANK) {
        u64* kin = reinterpret_cast<u64*>(smem_mr + (size_t)xp.dp * 4);
        DBG_STAMP(xp.stats ? xp.stats + 8 : nullptr, 0);
        rerank_stage_query<MERGE_THREADS>(xp, q, smem_mr);  // its loads fly while the lists are merged
        if (k <= MERGE_FAST_K) merge_waves(p, base, fast, kin);
        else merge_rounds<MERGE_THREADS, LPT>(p, base, wmin, [&](int r, u64 key) { kin[r] = key; });
        rerank_block<MERGE_THREADS>(xp, q, smem_mr, nullptr);  // starts with a block barrier
    } else if (k <= MERGE_FAST_K) {
        merge_waves(p, base, fast, res);
        if ((int)threadIdx.x < k) emit_result(p, (size_t)q * k + threadIdx.x, res[threadIdx.x]);
    } else {
        merge_rounds<MERGE_THREADS, LPT>(p, base, wmin, [&](int r, u64 key) { emit_result(p, (size_t)q * k + r, key); });
    }
}

// Up to 64 lists (the all-gathered per-rank lists of the multi-GPU path): one wave per query,
// one list per lane, no LDS and no barriers.
static __global__ __launch_bounds__(256) void merge_small_kernel(const MergeParams p) {
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
