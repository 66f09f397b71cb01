// ise_knn.hip -- brute-force L2 / inner-product kNN for MI355X (gfx950, CDNA4).
//
// Replaces the native work behind faiss.IndexFlatL2 / IndexFlatIP as the
// reference uses them (backend/engine.py:55, backend/utils.py:293-330,
// backend/kmeans_faiss.py:49, backend/siamese/test_index.py:54) and
// faiss.normalize_L2 (backend/utils.py:303).  C ABI: include/ise_knn.h.
//
// Data layout in HBM
//   xb     [cap][dp]  float32  index rows, row stride dp = d rounded up to a
//                              multiple of 16 (64 if d > 64) floats, zero padded
//   norms  [cap]      float32  |y|^2 per row (written at add time)
//   part   [nqt][nb][16][k] u64  per-block sorted candidate lists (workspace)
//
// Kernels
//   scan_kernel   one pass over the index per tile of 16 queries.  A wave owns
//                 16 index rows x 16 queries at a time: the Q x I^T tile is an
//                 fp32 MFMA chain (v_mfma_f32_16x16x4_f32, bit-exact fmaf
//                 chain), index rows go HBM -> VGPR with 16-byte loads and are
//                 read exactly once, queries sit in LDS, distances never leave
//                 registers: each lane filters its 4 scores against the
//                 wave's running k-th best and only survivors touch LDS.
//                 HBM-bound: 4*N*d bytes per 16-query tile.
//   merge_kernel  k-way merge of the per-block (or per-rank) sorted lists.
//   norms_kernel / normalize_kernel / pad_rows_kernel  wave-per-row helpers.
//
// Candidate order: a 64-bit key = ord(score) << 32 | row id, where ord() is
// the order-preserving map float -> uint32 and score = squared L2 (or
// -inner product).  Ascending key order is (score, id) order, which is the
// order Faiss reports (ties by ascending id), and keys are unique, so every
// selection below is deterministic and independent of the grid shape.

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/ise_knn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define KEY_PAD (~0ull)
#define QT 16 /* queries per scan tile (one MFMA column block) */

// ---------------------------------------------------------------- device utils
__device__ __forceinline__ uint32_t ord_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, src);
    hi = (uint32_t)__builtin_amdgcn_readlane((int)hi, src);
    return ((u64)hi << 32) | lo;
}
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ u64 min_u64(u64 a, u64 b) { return a < b ? a : b; }
// min over each aligned group of 16 lanes (all lanes of the group get it)
__device__ __forceinline__ u64 row_min_u64(u64 v) {
    v = min_u64(v, dpp_u64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = min_u64(v, dpp_u64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = min_u64(v, dpp_u64<0x141>(v));  // row_half_mirror
    v = min_u64(v, dpp_u64<0x140>(v));  // row_mirror
    return v;
}
// min over the whole wave (all lanes get it)
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
    v = row_min_u64(v);
    const u64 r0 = readlane_u64(v, 0), r1 = readlane_u64(v, 16), r2 = readlane_u64(v, 32),
              r3 = readlane_u64(v, 48);
    return min_u64(min_u64(r0, r1), min_u64(r2, r3));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- scan kernel
struct ScanParams {
    const float* xb;     // [cap][dp]
    const float* norms;  // [cap]
    const float* q;      // [nq][d]
    const u64* floor_keys;  // optional [nq]: only keys > floor enter (k > 64 passes)
    u64* part;           // [nqt][nb][16][k]
    long long n;         // rows in the index
    int d, dp, qs_stride;
    int nq, k, metric;
    uint32_t id_base;
    int tiles_total, tiles_per_block;
    int ablate;  // dev builds (-DISE_ABLATE): bit mask of phases to skip, from $ISE_ABLATE
    unsigned long long* stamps;  // dev builds: [blocks][waves][8] s_memrealtime stamps (100 MHz), or null
};
#ifdef ISE_ABLATE
#define ABL(bit) (p.ablate & (bit))
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (p.stamps && lane == 0)                                                                 \
            p.stamps[((size_t)blockIdx.x * W + w) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();   \
    } while (0)
#else
#define ABL(bit) 0
#define STAMP(i) do {} while (0)
#endif

#define TAU0 ((u64)0xFF7FFFFFu << 32) /* ord(FLT_MAX) << 32: strict gate score < FLT_MAX */
#define KB 32                          /* boot-winner slots per query = largest k of one pass */

// Wave-level selection: the min(k, #real) smallest of the keys held as
// kk[e] = element (lane + 64 e), e < KPL (KEY_PAD = empty slot; elements with
// 64 e >= n must be empty), written SORTED to dst[0..).  Returns the number
// written; *kth = the k-th smallest key when k were written.
// Quickselect on the key value with wave-uniform control flow (ballot counts),
// then an all-pairs rank among the <= k winners only.  Real keys are unique and
// lie strictly between 0 and KEY_PAD.
template <int KPL>
__device__ __forceinline__ int wave_select(const u64 (&kk)[KPL], int n, int k, u64* dst, u64* kth) {
    int nreal = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n) nreal += __popcll(__ballot(kk[e] != KEY_PAD));
    u64 kstar = KEY_PAD - 1;  // fewer than k real keys: all of them win
    if (nreal > k) {
        u64 L = 0, H = KEY_PAD;  // the target lies in the open interval (L, H)
        int t = k - 1;           // its rank among the keys of that interval
        for (int round = 0;; round++) {
            // pivot: an element of the interval, position rotated per round so that
            // sorted input does not degrade the search
            u64 P = 0;
            bool found = false;
            const int rot = (round * 23 + 7) & 63;
#pragma unroll
            for (int e = 0; e < KPL; e++) {
                if (64 * e < n && !found) {
                    const u64 m = __ballot(kk[e] > L && kk[e] < H);
                    if (m) {
                        const u64 hi = (m >> rot) << rot;
                        const int pick = __ffsll((long long)(hi ? hi : m)) - 1;
                        P = readlane_u64(kk[e], pick);
                        found = true;
                    }
                }
            }
            int c_lt = 0;
#pragma unroll
            for (int e = 0; e < KPL; e++)
                if (64 * e < n) c_lt += __popcll(__ballot(kk[e] > L && kk[e] < P));
            if (c_lt == t) {
                kstar = P;
                break;
            }
            if (c_lt > t) {
                H = P;
            } else {
                L = P;
                t -= c_lt + 1;
            }
        }
    }
    // rank among the winners (keys <= kstar); at most k of them
    int rk[KPL];
#pragma unroll
    for (int e = 0; e < KPL; e++) rk[e] = 0;
#pragma unroll
    for (int es = 0; es < KPL; es++) {
        if (64 * es < n) {
            u64 m = __ballot(kk[es] <= kstar);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const u64 ki = readlane_u64(kk[es], l);
#pragma unroll
                for (int e = 0; e < KPL; e++) rk[e] += (ki < kk[e]) ? 1 : 0;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n && kk[e] <= kstar) dst[rk[e]] = kk[e];
    const int nw = min(nreal, k);
    if (nw == k) *kth = kstar;
    return nw;
}

// LDS bytes of one scan block (host mirror: scan_lds_bytes)
#define CAP 16 /* slots of a wave's private candidate list; merged out at MERGE_TRIG */
#define MERGE_TRIG 8
__host__ __device__ constexpr size_t scan_lds_layout(int S, int waves) {
    return (size_t)(QT * S + QT) * 4      /* qs, xn */
           + (size_t)QT * 8               /* tauS */
           + (size_t)QT * 4 * 2           /* bwc, lockS */
           + (size_t)waves * QT * 4       /* cntS */
           + (size_t)QT * KB * 8          /* bootw */
           + (size_t)waves * QT * CAP * 8 /* cand (boot staging aliases it: 16 x W*16 keys) */;
}

// CH : k-steps (16 floats each) per register chunk; dp/16 is a multiple of CH
// W  : waves per block
//
// Top-k bookkeeping (all off the streaming path):
//   boot   every wave scores its first tile and dumps all 16x16 keys; two block
//          barriers later each query has the sorted k best of those W*16 rows
//          (bootw) and a block-wide threshold tauS[q] = their k-th key.
//   steady a lane holds 4 scores per tile for one query and compares them with
//          its copy of the threshold; survivors (a few per wave over the whole
//          kernel) are appended to the wave's private list; at MERGE_TRIG entries
//          the wave takes the query's LDS lock, folds its list into bootw and
//          publishes the new k-th key, so tauS tracks the block's running k-th best.
//   final  per query, one wave selects the top-k of bootw + what is left in the W
//          private lists and writes the block's sorted list.
template <int CH, int W>
__global__ __launch_bounds__(W * 64, W / 2) void scan_kernel(const ScanParams p) {
    constexpr int BLOCK_THREADS = W * 64;
    constexpr int TPR = BLOCK_THREADS / QT;       // threads staging one query row
    constexpr int KPLB = (W * 16 + 63) / 64;      // boot: keys per lane
    constexpr int KPLF = (KB + W * CAP + 63) / 64;  // final: keys per lane (worst case)
    static_assert(KB + CAP <= 64 && MERGE_TRIG + 4 <= CAP && W * 16 * 16 <= W * QT * CAP, "list sizes");
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = p.qs_stride;
    float* qs = reinterpret_cast<float*>(smem);                 // [16][S]
    float* xn = qs + QT * S;                                    // [16]
    u64* tauS = reinterpret_cast<u64*>(xn + QT);                // [16]
    int* bwc = reinterpret_cast<int*>(tauS + QT);               // [16]
    int* lockS = bwc + QT;                                      // [16]
    int* cntS = lockS + QT;                                     // [W][16]
    u64* bootw = reinterpret_cast<u64*>(cntS + W * QT);         // [16][KB]
    u64* cand = bootw + QT * KB;                                // [W][16][CAP]
    u64* boot = cand;                                           // [16][W*16], dead before cand is used

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.y * QT;
    const int nqt = min(QT, p.nq - q0);
    const int k = p.k;
    const int nsteps = p.dp >> 4;
    const int t0 = blockIdx.x * p.tiles_per_block;
    const int t1 = min(t0 + p.tiles_per_block, p.tiles_total);
    const bool l2 = p.metric == ISE_METRIC_L2;

    auto load_chunk = [&](f32x4(&a)[CH], int tile, int s0) {
        const float* base = p.xb + ((size_t)tile * 16 + c) * p.dp + 4 * g + 16 * s0;
#pragma unroll
        for (int s = 0; s < CH; s++) a[s] = *reinterpret_cast<const f32x4*>(base + 16 * s);
    };
    auto load_norms = [&](int tile) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(p.norms + (size_t)tile * 16 + 4 * g);
    };

    STAMP(0);
    // ---- the first index chunk is requested before anything else: its HBM
    // latency overlaps the query staging below
    int tile = t0 + w;
    const bool has_work = tile < t1 && !ABL(16);
    f32x4 a0[CH], a1[CH];
    f32x4 yn_cur = {0.f, 0.f, 0.f, 0.f};
    if (has_work) {
        load_chunk(a0, tile, 0);
        yn_cur = load_norms(tile);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- stage the query tile (zero padded to 16 x S) and |x|^2: TPR threads per row
    {
        const int cc = tid / TPR, t = tid % TPR;
        const bool rowok = cc < nqt && !ABL(1);
        const float* src = p.q + (size_t)(q0 + (rowok ? cc : 0)) * p.d;
        float sn = 0.f;
        if ((p.d & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.q) & 15) == 0)) {
            const int d4 = p.d >> 2, S4 = S >> 2;
            for (int j4 = t; j4 < S4; j4 += TPR) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rowok && j4 < d4) v = *reinterpret_cast<const f32x4*>(src + 4 * j4);
                *reinterpret_cast<f32x4*>(qs + cc * S + 4 * j4) = v;
                sn = fmaf(v[0], v[0], sn);
                sn = fmaf(v[1], v[1], sn);
                sn = fmaf(v[2], v[2], sn);
                sn = fmaf(v[3], v[3], sn);
            }
        } else {
            for (int j = t; j < S; j += TPR) {
                const float v = (rowok && j < p.d) ? src[j] : 0.f;
                qs[cc * S + j] = v;
                sn = fmaf(v, v, sn);
            }
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sn += __shfl_xor(sn, o);
        if (t == 0) xn[cc] = sn;
        if (tid < QT) {
            tauS[tid] = TAU0;
            bwc[tid] = 0;
            lockS[tid] = 0;
        }
    }
    __syncthreads();
    STAMP(1);

    const float xq_n = xn[c];
    const float* qrow = qs + c * S + 4 * g;
    const u64 qmask = 0x0001000100010001ull << c;
    const u64 lt_mask = (1ull << lane) - 1ull;
    const u64 key_floor = (p.floor_keys && c < nqt) ? p.floor_keys[q0 + c] : 0ull;
    const bool use_floor = p.floor_keys != nullptr;

    u64 tau = TAU0;
    int cnt = 0;
    bool booted = false;
    u64* mybuf = cand + (size_t)(w * QT + c) * CAP;

    // fold the wave's private list of query qq into the block's sorted list bootw[qq]
    // (top-k of their union) under the query's LDS lock; publish the new k-th key
    auto merge_out = [&](int qq) {
        const int n_ = __builtin_amdgcn_readlane(cnt, qq);
        const u64* buf = cand + (size_t)(w * QT + qq) * CAP;
        if (lane == 0)
            while (atomicCAS(&lockS[qq], 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
        const int nb = bwc[qq];
        u64 kk[1];
        kk[0] = lane < nb ? bootw[qq * KB + lane] : (lane - nb < n_ ? buf[lane - nb] : KEY_PAD);
        wave_lds_fence();
        u64 ktau = KEY_PAD;
        const int nw = wave_select<1>(kk, nb + n_, k, bootw + qq * KB, &ktau);  // nb + n_ <= KB + CAP <= 64
        if (lane == 0) {
            bwc[qq] = nw;
            if (nw == k) tauS[qq] = ktau;  // <= the old value: the union only adds keys
        }
        wave_lds_fence();
        if (lane == 0) atomicExch(&lockS[qq], 0);
        if (c == qq) {
            cnt = 0;
            if (nw == k) tau = min_u64(tau, ktau);
        }
    };

    // boot: all W*16 first-tile keys of a query -> its k best + the block threshold
    auto boot_phase = [&](const u64(&key)[4]) {
        STAMP(2);
#pragma unroll
        for (int j = 0; j < 4; j++) boot[(size_t)c * (W * 16) + w * 16 + 4 * g + j] = key[j];
        __syncthreads();
        for (int qq = w; qq < QT; qq += W) {
            u64 kk[KPLB];
#pragma unroll
            for (int e = 0; e < KPLB; e++)
                kk[e] = (lane + 64 * e) < W * 16 ? boot[(size_t)qq * (W * 16) + lane + 64 * e] : KEY_PAD;
            u64 ktau = TAU0;
            const int nw = wave_select<KPLB>(kk, W * 16, k, bootw + qq * KB, &ktau);
            if (lane == 0) {
                bwc[qq] = nw;
                tauS[qq] = ktau;
            }
        }
        __syncthreads();
        tau = tauS[c];
        booted = true;
        STAMP(3);
    };

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};

    auto epilogue = [&](int etile, f32x4 yn) {
        const f32x4 dot = acc0 + acc1;
        acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        const long long row0 = (long long)etile * 16 + 4 * g;
        u64 key[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float sc;
            if (l2) {
                sc = (xq_n + yn[j]) - 2.f * dot[j];
                sc = sc < 0.f ? 0.f : sc;  // keeps NaN (Faiss: if (dis < 0) dis = 0)
            } else {
                sc = -dot[j];
            }
            bool ok = (row0 + j < p.n) && (sc < FLT_MAX) && (c < nqt);
            const u64 kj = ((u64)ord_f32(sc) << 32) | (uint32_t)((uint32_t)(row0 + j) + p.id_base);
            if (use_floor) ok = ok && (kj > key_floor);
            key[j] = ok ? kj : KEY_PAD;
        }
        if (!booted) {
            boot_phase(key);
            return;
        }
        tau = min_u64(tau, tauS[c]);
        const bool pend = key[0] < tau || key[1] < tau || key[2] < tau || key[3] < tau;
        if (__any(pend) && !ABL(2)) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool v = key[j] < tau;
                const u64 m = __ballot(v);
                if (m) {
                    const u64 mq = m & qmask;
                    if (v) mybuf[cnt + __popcll(mq & lt_mask)] = key[j];
                    cnt += __popcll(mq);
                    u64 nm = __ballot(cnt >= MERGE_TRIG) & 0xFFFFull;
                    while (nm) {
                        const int qq = __ffsll((long long)nm) - 1;
                        nm &= nm - 1;
                        merge_out(qq);
                    }
                }
            }
        }
    };

    auto compute_chunk = [&](const f32x4(&a)[CH], int s0) {
#pragma unroll
        for (int s = 0; s < CH; s++) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(qrow + 16 * (s0 + s));
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][2], b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][3], b[3], acc1, 0, 0, 0);
        }
    };

    // ---- main loop: two register chunks in flight, tiles interleaved by wave.
    // Loads are issued unconditionally (the last iteration re-reads its own
    // chunk) so that no control-flow join sits between a load and its use:
    // hipcc then emits counted vmcnt waits and the prefetch stays in flight.
    if (has_work) {
        int s0 = 0;
        for (;;) {
            int ns0 = s0 + CH, ntile = tile;
            if (ns0 >= nsteps) { ns0 = 0; ntile = tile + W; }
            bool has_next = ntile < t1;
            {
                const int lt = has_next ? ntile : tile, ls = has_next ? ns0 : s0;
                load_chunk(a1, lt, ls);
                const f32x4 yn_nx = load_norms(lt);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs
                compute_chunk(a0, s0);
                if (s0 + CH >= nsteps && !ABL(8)) epilogue(tile, yn_cur);
                yn_cur = yn_nx;
            }
            if (!has_next) break;
            tile = ntile; s0 = ns0;

            ns0 = s0 + CH; ntile = tile;
            if (ns0 >= nsteps) { ns0 = 0; ntile = tile + W; }
            has_next = ntile < t1;
            {
                const int lt = has_next ? ntile : tile, ls = has_next ? ns0 : s0;
                load_chunk(a0, lt, ls);
                const f32x4 yn_nx = load_norms(lt);
                __builtin_amdgcn_sched_barrier(0);
                compute_chunk(a1, s0);
                if (s0 + CH >= nsteps && !ABL(8)) epilogue(tile, yn_cur);
                yn_cur = yn_nx;
            }
            if (!has_next) break;
            tile = ntile; s0 = ns0;
        }
    }
    if (!booted) {  // a wave without a tile still takes part in the two boot barriers
        const u64 none[4] = {KEY_PAD, KEY_PAD, KEY_PAD, KEY_PAD};
        boot_phase(none);
    }

    // ---- final: per query, rank bootw + the W private lists, write the sorted top-k
    STAMP(4);
    if (g == 0) cntS[w * QT + c] = cnt;
    __syncthreads();
    STAMP(5);
    for (int qq = w; qq < QT && !ABL(4); qq += W) {
        int P[W + 2];
        P[0] = 0;
        P[1] = bwc[qq];
#pragma unroll
        for (int i = 0; i < W; i++) P[i + 2] = P[i + 1] + cntS[i * QT + qq];
        const int n = P[W + 1];
        u64 kk[KPLF];
#pragma unroll
        for (int e = 0; e < KPLF; e++) {
            kk[e] = KEY_PAD;
            const int idx = lane + 64 * e;
            if (64 * e < n) {
                if (idx < P[1]) kk[e] = bootw[qq * KB + idx];
#pragma unroll
                for (int i = 0; i < W; i++)
                    if (idx >= P[i + 1] && idx < P[i + 2])
                        kk[e] = cand[(size_t)(i * QT + qq) * CAP + idx - P[i + 1]];
            }
        }
        u64* out = p.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * QT + qq) * k;
        u64 kth_unused;
        const int nw = wave_select<KPLF>(kk, n, k, out, &kth_unused);
        if (lane >= nw && lane < k) out[lane] = KEY_PAD;  // k <= KB <= 64
    }
    STAMP(6);
}

// ---------------------------------------------------------------- merge kernel
// One block per query; thread t owns lists t, t+256, ...; k rounds of a
// block-wide argmin over the list heads (keys are unique).
#define MERGE_THREADS 256
#define MERGE_LPT 4 /* lists per thread -> at most 1024 lists */
struct MergeParams {
    const u64* lists;
    long long stride_list;   // elements between consecutive lists
    long long stride_qtile;  // elements between consecutive 16-query tiles
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
    const u64* base = p.lists + (size_t)(q >> 4) * p.stride_qtile + (size_t)(q & 15) * k;
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

// ---------------------------------------------------------------- row helpers
// |y|^2 per row, wave per row, fixed summation order (lanes stride float4, then
// an xor butterfly): deterministic for a given dp.
__global__ __launch_bounds__(256) void norms_kernel(const float* __restrict__ x, long long row0,
                                                    long long n, int dp, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = row0 + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= row0 + n) return;
    const float* xr = x + (size_t)r * dp;
    float s = 0.f;
    for (int j = lane * 4; j < dp; j += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + j);
        s = fmaf(v[0], v[0], s);
        s = fmaf(v[1], v[1], s);
        s = fmaf(v[2], v[2], s);
        s = fmaf(v[3], v[3], s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) out[r] = s;
}

// copy n rows of d floats (unpadded, src) into the padded index layout
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, long long n, int d,
                                                       float* __restrict__ dst, int dp) {
    const long long total = n * dp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dp;
        const int j = (int)(i - r * dp);
        dst[i] = j < d ? src[(size_t)r * d + j] : 0.f;
    }
}

// faiss.normalize_L2 [upstream-faiss fvec_renorm_L2]: nr = |x|^2 (float32);
// if nr > 0: x *= (float)(1.0 / sqrtf(nr)).  Wave per row.
__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ x, long long n, int d) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float* xr = x + (size_t)r * d;
    float s = 0.f;
    for (int j = lane; j < d; j += 64) {
        const float v = xr[j];
        s = fmaf(v, v, s);
    }
    s = wave_sum_f32(s);
    if (s > 0.f) {
        const float inv = (float)(1.0 / (double)sqrtf(s));
        for (int j = lane; j < d; j += 64) xr[j] *= inv;
    }
}

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
    long long n = 0, cap = 0;
    float* xb = nullptr;
    float* norms = nullptr;
    // workspace (grown lazily, guarded by mu)
    u64* part = nullptr;
    size_t part_elems = 0;
    u64* keys_tmp = nullptr;  // [nq][k] scratch for multi-pass k
    size_t keys_tmp_elems = 0;
    // host-API staging
    hipStream_t stream = nullptr;
    float* q_dev = nullptr;  size_t q_elems = 0;
    float* D_dev = nullptr;  long long* I_dev = nullptr;  size_t out_elems = 0;
    float* h_stage = nullptr;  size_t h_stage_bytes = 0;  // pinned
    int num_cu = 256;
    std::mutex mu;
};

static int pad_dim(int d) { return d > 64 ? (d + 63) / 64 * 64 : (d + 15) / 16 * 16; }
static int chunk_steps(int dp) {
    const int steps = dp / 16;
    for (int ch = 8; ch > 1; ch >>= 1)
        if (steps % ch == 0) return ch;
    return 1;
}
static int qs_stride_for(int dp) {
    // (stride/4) % 16 == 2 makes the 16x4 ds_read_b128 pattern conflict-free
    const int pad = ((2 - (dp / 4)) % 16 + 16) % 16 * 4;
    return dp + pad;
}
#define KPASS_MAX 32 /* largest k one scan pass selects; larger k runs floor-keyed passes */
static size_t scan_lds_bytes(int dp, int waves) { return scan_lds_layout(qs_stride_for(dp), waves); }

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
    if (!out) return fail(ISE_E_INVALID, "out is NULL");
    *out = nullptr;
    if (d <= 0) return fail(ISE_E_INVALID, "d must be positive");
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
    h->dp = pad_dim(d);
    h->metric = metric;
    h->device = device;
    h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    DeviceGuard gd(device);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(ISE_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    *out = h;
    return ISE_OK;
}

static void free_all(ise_index* h) {
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    if (h->part) (void)hipFree(h->part);
    if (h->keys_tmp) (void)hipFree(h->keys_tmp);
    if (h->q_dev) (void)hipFree(h->q_dev);
    if (h->D_dev) (void)hipFree(h->D_dev);
    if (h->I_dev) (void)hipFree(h->I_dev);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    h->xb = h->norms = nullptr;
    h->part = h->keys_tmp = nullptr;
    h->q_dev = h->D_dev = nullptr;
    h->I_dev = nullptr;
    h->h_stage = nullptr;
    h->part_elems = h->keys_tmp_elems = h->q_elems = h->out_elems = h->h_stage_bytes = 0;
    h->n = h->cap = 0;
}

extern "C" int ise_index_destroy(ise_index_t* h) {
    if (!h) return ISE_OK;
    {
        DeviceGuard gd(h->device);
        (void)hipDeviceSynchronize();
        free_all(h);
        if (h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
    return ISE_OK;
}

extern "C" int ise_index_reset(ise_index_t* h) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard gd(h->device);
    HIP_TRY(hipDeviceSynchronize());
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    h->xb = h->norms = nullptr;
    h->n = h->cap = 0;
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
    float* nx = nullptr;
    float* nn = nullptr;
    HIP_TRY(hipMalloc(&nx, (size_t)want * h->dp * sizeof(float)));
    hipError_t e = hipMalloc(&nn, (size_t)want * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(nx);
        return fail(ISE_E_NOMEM, std::string("hipMalloc(norms): ") + hipGetErrorString(e));
    }
    if (h->n > 0) {
        HIP_TRY(hipMemcpyAsync(nx, h->xb, (size_t)h->n * h->dp * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(nn, h->norms, (size_t)h->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipMemsetAsync(nx + (size_t)h->n * h->dp, 0, (size_t)(want - h->n) * h->dp * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(nn + h->n, 0, (size_t)(want - h->n) * sizeof(float), st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h->xb) (void)hipFree(h->xb);
    if (h->norms) (void)hipFree(h->norms);
    h->xb = nx;
    h->norms = nn;
    h->cap = want;
    return ISE_OK;
}

static int add_device_locked(ise_index* h, const float* x_dev, long long n, hipStream_t st) {
    if (n == 0) return ISE_OK;
    if (h->n + n >= (1ll << 32)) return fail(ISE_E_INVALID, "index would exceed 2^32 - 1 rows");
    int rc = reserve_rows(h, h->n + n, st);
    if (rc) return rc;
    float* dst = h->xb + (size_t)h->n * h->dp;
    if (h->dp == h->d) {
        HIP_TRY(hipMemcpyAsync(dst, x_dev, (size_t)n * h->d * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else {
        const long long total = n * h->dp;
        const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(blocks), dim3(256), 0, st, x_dev, n, h->d, dst, h->dp);
        HIP_TRY(hipGetLastError());
    }
    const long long nblk = (n + 3) / 4;
    // grid.x is a 32-bit quantity; n < 2^32 so nblk < 2^30
    hipLaunchKernelGGL(norms_kernel, dim3((unsigned)nblk), dim3(256), 0, st, h->xb, h->n, n, h->dp, h->norms);
    HIP_TRY(hipGetLastError());
    h->n += n;
    return ISE_OK;
}

extern "C" int ise_index_add_device(ise_index_t* h, const float* x_dev, int64_t n, void* stream) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && !x_dev)) return fail(ISE_E_INVALID, "bad rows argument");
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard gd(h->device);
    return add_device_locked(h, x_dev, n, (hipStream_t)stream);
}

extern "C" int ise_index_add_host(ise_index_t* h, const float* x, int64_t n) {
    if (!h) return fail(ISE_E_INVALID, "handle is NULL");
    if (n < 0 || (n > 0 && !x)) return fail(ISE_E_INVALID, "bad rows argument");
    if (n == 0) return ISE_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard gd(h->device);
    if (h->n + n >= (1ll << 32)) return fail(ISE_E_INVALID, "index would exceed 2^32 - 1 rows");
    int rc = reserve_rows(h, h->n + n, h->stream);
    if (rc) return rc;
    // upload in slabs through a device staging buffer (only needed when padding)
    const long long slab = std::max<long long>(1, (256ll << 20) / ((long long)h->d * 4));
    float* tmp = nullptr;
    if (h->dp != h->d) HIP_TRY(hipMalloc(&tmp, (size_t)std::min<long long>(slab, n) * h->d * sizeof(float)));
    for (long long i0 = 0; i0 < n; i0 += slab) {
        const long long m = std::min<long long>(slab, n - i0);
        if (h->dp == h->d) {
            float* dst = h->xb + (size_t)h->n * h->dp;
            hipError_t e = hipMemcpyAsync(dst, x + (size_t)i0 * h->d, (size_t)m * h->d * sizeof(float),
                                          hipMemcpyHostToDevice, h->stream);
            if (e != hipSuccess) return fail(ISE_E_HIP, std::string("H2D: ") + hipGetErrorString(e));
            const long long nblk = (m + 3) / 4;
            hipLaunchKernelGGL(norms_kernel, dim3((unsigned)nblk), dim3(256), 0, h->stream, h->xb, h->n, m,
                               h->dp, h->norms);
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
    std::lock_guard<std::mutex> lk(h->mu);
    DeviceGuard gd(h->device);
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)h->d * 4, h->xb + (size_t)i0 * h->dp, (size_t)h->dp * 4, (size_t)h->d * 4,
                             (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return ISE_OK;
}

// ---- search
#define LDS_LIMIT (160 * 1024)
template <int CH, int W>
static void launch_one(dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    static bool attr_done = false;  // benign race: the attribute is idempotent
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel<CH, W>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr_done = true;
    }
    hipLaunchKernelGGL((scan_kernel<CH, W>), grid, dim3(W * 64), lds, st, sp);
}
template <int W>
static void launch_scan_ch(int ch, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    switch (ch) {
        case 8: launch_one<8, W>(grid, lds, st, sp); break;
        case 4: launch_one<4, W>(grid, lds, st, sp); break;
        case 2: launch_one<2, W>(grid, lds, st, sp); break;
        default: launch_one<1, W>(grid, lds, st, sp); break;
    }
}
static void launch_scan(int ch, int waves, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    if (waves == 8) launch_scan_ch<8>(ch, grid, lds, st, sp);
    else launch_scan_ch<4>(ch, grid, lds, st, sp);
}

struct ScanPlan {
    int nblocks, tiles_total, tiles_per_block, nqt, ch, kpass, waves;
    size_t lds;
};

static int make_plan(const ise_index* h, long long nq, int k, ScanPlan* pl) {
    pl->kpass = k < KPASS_MAX ? k : KPASS_MAX;
    pl->ch = chunk_steps(h->dp);
    pl->waves = 8;
    pl->lds = scan_lds_bytes(h->dp, 8);
    if (pl->lds > LDS_LIMIT) {
        pl->waves = 4;
        pl->lds = scan_lds_bytes(h->dp, 4);
    }
    if (pl->lds > LDS_LIMIT)
        return fail(ISE_E_INVALID, "d too large: the 16-query tile must fit the 160 KiB LDS (d <= 2048)");
    pl->tiles_total = (int)((h->n + 15) / 16);
    int blocks_per_cu = pl->lds <= LDS_LIMIT / 2 ? 2 : 1;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_PLAN")) {  // dev: "waves,blocks_per_cu"
        int wv = 8, bpc = 2;
        if (sscanf(e, "%d,%d", &wv, &bpc) == 2 && (wv == 4 || wv == 8) && bpc >= 1) {
            pl->waves = wv;
            pl->lds = scan_lds_bytes(h->dp, wv);
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
    pl->nqt = (int)((nq + QT - 1) / QT);
    return ISE_OK;
}

// workspace: part [nqt][nb][16][kpass]; for k > kpass additionally
// keys_tmp = keys_all [nq][k] | floor [nq] | pass_keys [nq][kpass]
static int ensure_workspace(ise_index* h, const ScanPlan& pl, long long nq, int k) {
    const size_t need = (size_t)pl.nqt * pl.nblocks * QT * pl.kpass;
    if (need > h->part_elems) {
        if (h->part) (void)hipFree(h->part);
        h->part = nullptr;
        h->part_elems = 0;
        HIP_TRY(hipMalloc(&h->part, need * sizeof(u64)));
        h->part_elems = need;
    }
    if (k > pl.kpass) {
        const size_t need2 = (size_t)nq * ((size_t)k + 1 + pl.kpass);
        if (need2 > h->keys_tmp_elems) {
            if (h->keys_tmp) (void)hipFree(h->keys_tmp);
            h->keys_tmp = nullptr;
            h->keys_tmp_elems = 0;
            HIP_TRY(hipMalloc(&h->keys_tmp, need2 * sizeof(u64)));
            h->keys_tmp_elems = need2;
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

// enqueue one search batch; outputs (D, I) and/or keys.  Nothing here blocks.
static int search_enqueue(ise_index* h, const float* q_dev, long long nq, int k, uint32_t id_base, float* D_dev,
                          long long* I_dev, u64* keys_out, hipStream_t st, TimedOut* tm) {
    ScanPlan pl;
    int rc = make_plan(h, nq, k, &pl);
    if (rc) return rc;
    rc = ensure_workspace(h, pl, nq, k);
    if (rc) return rc;

    ScanParams sp;
    sp.xb = h->xb; sp.norms = h->norms; sp.q = q_dev; sp.floor_keys = nullptr; sp.part = h->part;
    sp.n = h->n; sp.d = h->d; sp.dp = h->dp; sp.qs_stride = qs_stride_for(h->dp);
    sp.nq = (int)nq; sp.k = pl.kpass; sp.metric = h->metric; sp.id_base = id_base;
    sp.tiles_total = pl.tiles_total; sp.tiles_per_block = pl.tiles_per_block;
    sp.ablate = 0;
    sp.stamps = nullptr;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_ABLATE")) sp.ablate = atoi(e);
    if (const char* e = getenv("ISE_STAMPS")) sp.stamps = (unsigned long long*)strtoull(e, nullptr, 0);
#endif

    MergeParams mp;
    mp.lists = h->part; mp.stride_list = (long long)QT * pl.kpass;
    mp.stride_qtile = (long long)pl.nblocks * QT * pl.kpass;
    mp.n_lists = pl.nblocks; mp.nq = (int)nq; mp.k = pl.kpass; mp.metric = h->metric;

    const dim3 grid((unsigned)pl.nblocks, (unsigned)pl.nqt);
    if (k <= pl.kpass) {
        mp.D = D_dev; mp.I = I_dev; mp.keys_out = keys_out;
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e0, st));
        launch_scan(pl.ch, pl.waves, grid, pl.lds, st, sp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e1, st));
        hipLaunchKernelGGL(merge_kernel, dim3((unsigned)nq), dim3(MERGE_THREADS), 0, st, mp);
        HIP_TRY(hipGetLastError());
        if (tm && tm->on) HIP_TRY(hipEventRecord(tm->e2, st));
        return ISE_OK;
    }
    // k > KPASS_MAX: passes of KPASS_MAX; a pass only admits keys above the
    // previous pass's last key (keys are totally ordered and unique)
    u64* keys_all = keys_out ? keys_out : h->keys_tmp;      // [nq][k]
    u64* floor_dev = h->keys_tmp + (size_t)nq * k;          // [nq]
    u64* pass_keys = floor_dev + nq;                        // [nq][kpass]
    for (int off = 0; off < k; off += pl.kpass) {
        sp.floor_keys = off ? floor_dev : nullptr;
        mp.D = nullptr; mp.I = nullptr; mp.keys_out = pass_keys;
        launch_scan(pl.ch, pl.waves, grid, pl.lds, st, sp);
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
    std::lock_guard<std::mutex> lk(h->mu);
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
    std::lock_guard<std::mutex> lk(h->mu);
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
    std::lock_guard<std::mutex> lk(h->mu);
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
    std::lock_guard<std::mutex> lk(h->mu);
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
    mp.stride_qtile = (long long)QT * k;
    mp.n_lists = n_lists; mp.nq = (int)nq; mp.k = k; mp.metric = metric;
    mp.D = D_dev; mp.I = (long long*)I_dev; mp.keys_out = nullptr;
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
