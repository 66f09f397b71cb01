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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define KEY_PAD (~0ull)

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
    const void* xb;      // [cap][dp] float32 or bf16 rows; 16-byte "slots": row_slots per row
    const float* norms;  // [cap]
    const float* q;      // [nq][d]
    const float* mu;     // [dp] shift vector (zero padded), used by SHIFT kernels
    const u64* floor_keys;  // optional [nq]: only keys > floor enter (k > 64 passes)
    u64* part;           // [nqt][nb][16 T][k]
    long long n;         // rows in the index
    int d, dp, qs_stride;  // qs_stride: LDS query row stride in 4-byte units (floats, or bf16 pairs)
    int row_slots;         // 16-byte slots per index row = dp * elem_size / 16; one k-step = 4 slots
    int nq, k, kb, metric;  // kb: block-list slots per query (16 or 32, >= k)
    uint32_t id_base;
    int tiles_total, tiles_per_block;
    int ablate;  // dev builds (-DISE_ABLATE): bit mask of phases to skip, from $ISE_ABLATE
    unsigned long long* stamps;  // dev builds: [blocks][waves][16] stamps (0-7 s_memrealtime 100 MHz, 8-9 s_memtime), or null
};
#ifdef ISE_ABLATE
#define ABL(bit) (p.ablate & (bit))
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        if (p.stamps && lane == 0)                                                                 \
            p.stamps[((size_t)blockIdx.x * W + w) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#define CSTAMP(i)                                                                                  \
    do {                                                                                           \
        if (p.stamps && lane == 0)                                                                 \
            p.stamps[((size_t)blockIdx.x * W + w) * 16 + (i)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#else
#define ABL(bit) 0
#define STAMP(i) do {} while (0)
#define CSTAMP(i) do {} while (0)
#endif

#define TAU0 ((u64)0xFF7FFFFFu << 32) /* ord(FLT_MAX) << 32: strict gate score < FLT_MAX */
#define CAP 16       /* slots of a wave's private candidate list; folded out at MERGE_TRIG */
#define MERGE_TRIG 12
#define KB_MAX 32    /* block-list slots per query (runtime kb = 16 or 32), >= k of one pass */

// ---- wave-level selection primitives on 64-bit keys.  Keys are held as
// kk[e] = element (lane + 64 e), e < KPL; KEY_PAD = empty slot; elements with
// 64 e >= n must be empty.  Real keys are unique and lie strictly between 0 and
// KEY_PAD.  Control flow is wave-uniform (ballot counts in SGPRs).

// Exact: the min(k, #real) smallest keys, written SORTED to dst[0..).  Returns
// the number written; *kth = the k-th smallest key when k were written.
// Quickselect on the key value, then an all-pairs rank among the <= k winners.
template <int KPL>
__device__ __forceinline__ int wave_select(const u64 (&kk)[KPL], int n, int k, u64* dst, u64* kth) {
    int nreal = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n) nreal += __popcll(__ballot(kk[e] != KEY_PAD));
    u64 kstar = KEY_PAD - 1;  // all real keys are ranked; right when few keys are held
    if (nreal > k && nreal > 40) {  // many keys: narrow to the k winners first (quickselect)
        u64 L = 0, H = KEY_PAD;  // the target lies in the open interval (L, H)
        int t = k - 1;           // its rank among the keys of that interval
        for (int round = 0;; round++) {
            u64 P = 0;  // pivot: an element of the interval, position rotated per round
            bool found = false;
            const int rot = (round * 23 + 7) & 63;
#pragma unroll
            for (int e = 0; e < KPL; e++) {
                if (64 * e < n && !found) {
                    const u64 m = __ballot(kk[e] > L && kk[e] < H);
                    if (m) {
                        const u64 hi = (m >> rot) << rot;
                        P = readlane_u64(kk[e], __ffsll((long long)(hi ? hi : m)) - 1);
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
    int rk[KPL];  // rank among the winners (keys <= kstar); at most k of them
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
    u64 kth_key = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++) {
        if (64 * e < n) {
            const bool win = kk[e] <= kstar && rk[e] < k;
            if (win) dst[rk[e]] = kk[e];
            const u64 hit = __ballot(win && rk[e] == k - 1);
            if (hit) kth_key = readlane_u64(kk[e], __ffsll((long long)hit) - 1);
        }
    }
    const int nw = min(nreal, k);
    if (nw == k) *kth = kth_key;
    return nw;
}

// Windowed cut: finds a key P with kmin <= #(keys <= P) <= kmax and writes those
// keys UNSORTED to dst (all real keys if there are at most kmax).  Returns the
// count; *cut = P when the count reached kmin.  A valid, cheap threshold: the
// kmin-th smallest key is <= P.  Window width makes this take a few rounds only.
template <int KPL>
__device__ __forceinline__ int wave_cut(const u64 (&kk)[KPL], int n, int kmin, int kmax, u64* dst, u64* cut) {
    const int lane = threadIdx.x & 63;
    const u64 lt_mask = (1ull << lane) - 1ull;
    int nreal = 0;
    u64 mx = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n) nreal += __popcll(__ballot(kk[e] != KEY_PAD));
    u64 P = KEY_PAD - 1;
    int cnt = nreal;
    if (nreal > kmax) {
        u64 L = 0, H = KEY_PAD;
        int base = 0;  // #(keys <= L)
        for (int round = 0;; round++) {
            u64 piv = 0;
            bool found = false;
            const int rot = (round * 23 + 7) & 63;
#pragma unroll
            for (int e = 0; e < KPL; e++) {
                if (64 * e < n && !found) {
                    const u64 m = __ballot(kk[e] > L && kk[e] < H);
                    if (m) {
                        const u64 hi = (m >> rot) << rot;
                        piv = readlane_u64(kk[e], __ffsll((long long)(hi ? hi : m)) - 1);
                        found = true;
                    }
                }
            }
            int c = base;
#pragma unroll
            for (int e = 0; e < KPL; e++)
                if (64 * e < n) c += __popcll(__ballot(kk[e] > L && kk[e] <= piv));
            if (c < kmin) {
                L = piv;
                base = c;
            } else if (c > kmax) {
                H = piv;
            } else {
                P = piv;
                cnt = c;
                break;
            }
        }
    } else if (nreal >= kmin) {
        // every real key stays: the cut is the largest of them
#pragma unroll
        for (int e = 0; e < KPL; e++)
            if (64 * e < n) {
                const u64 v = kk[e] == KEY_PAD ? 0ull : kk[e];
                mx = mx > v ? mx : v;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(mx, o);
            mx = mx > other ? mx : other;
        }
        P = mx;
    }
    int off = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++) {
        if (64 * e < n) {
            const bool keep = kk[e] <= P;  // P < KEY_PAD: empty slots never kept
            const u64 m = __ballot(keep);
            if (keep) dst[off + __popcll(m & lt_mask)] = kk[e];
            off += __popcll(m);
        }
    }
    if (cnt >= kmin) *cut = P;
    return cnt;
}

// LDS bytes of one scan block (host and device agree through this function)
__host__ __device__ constexpr size_t scan_lds_layout(int S, int waves, int T, int kb) {
    return (size_t)S * 4 /* mus: the shift vector, laid out like one query row */ +
           (size_t)(16 * T) * ((size_t)S * 4 + 4 /* qs, xn */ + 8 /* tauS */ + 8 /* bwc, lockS */ +
                               (size_t)waves * 4 /* cntS */ + (size_t)kb * 8 /* bootw */ +
                               (size_t)waves * CAP * 8 /* cand; boot staging aliases it */);
}

// CH : k-steps (16 floats each) per register chunk; dp/16 is a multiple of CH
// W  : waves per block
// T  : query tiles of 16 per pass: the index stream is read once for 16*T queries
//      (one MFMA column block and two accumulator chains per tile)
//
// Top-k bookkeeping (all off the streaming path, per query):
//   boot   every wave scores its first row tile and dumps all 16x16 keys; two block
//          barriers later the query has a block list bootw of between k and kb of
//          the best of those W*16 rows (windowed cut) and a threshold tauS = the cut.
//   steady a lane holds 4 scores per row tile for one query per query tile and
//          compares them with its copy of the threshold; survivors (a few per wave
//          over the whole kernel) are appended to the wave's private list; at
//          MERGE_TRIG entries the wave takes the query's LDS lock, folds its list
//          into bootw (exact top-k of the union, sorted) and publishes the new k-th
//          key, so tauS tracks the block's running k-th best.
//   final  one wave selects the exact sorted top-k of bootw + what is left in the W
//          private lists and writes the block's list to HBM.
//
// SHIFT (fp32 L2 only): distances are translation invariant, and the expanded form
// |x|^2 + |y|^2 - 2 x.y loses digits when the rows share a large common component
// (CNN embeddings: |y|^2 ~ 1e5, neighbour distances ~ 1e-1).  The index keeps a fixed
// shift vector mu (mean of the first rows added); norms are |y - mu|^2, queries are
// staged as x - mu and the row fragments are shifted in registers before the MFMAs
// (4 VALU subs per k-step), so every term is as small as the data's spread, not its
// offset.  Rows stay stored unshifted (reconstruct / write_index are exact).
template <int CH, int W, int T, bool BF16, bool SHIFT>
__global__ __launch_bounds__(W * 64, T == 1 ? W / 2 : (W / 4 > 0 ? W / 4 : 1)) void scan_kernel(const ScanParams p) {
    static_assert(!(BF16 && SHIFT), "the shift is applied to fp32 rows only");
    constexpr int BLOCK_THREADS = W * 64;
    constexpr int NQ = 16 * T;                        // queries per block pass
    constexpr int TPR = BLOCK_THREADS / 16;           // threads staging one query row (per tile)
    constexpr int KPLB = (W * 16 + 63) / 64;          // boot: keys per lane
    constexpr int KPLF = (KB_MAX + W * CAP + 63) / 64;  // final: keys per lane (worst case)
    static_assert(KB_MAX + CAP <= 64 && MERGE_TRIG + 4 <= CAP, "list sizes");
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = p.qs_stride;
    const int kb = p.kb;
    float* mus = reinterpret_cast<float*>(smem);                // [S] shift vector (SHIFT only)
    float* qs = mus + S;                                        // [NQ][S]
    float* xn = qs + NQ * S;                                    // [NQ]
    u64* tauS = reinterpret_cast<u64*>(xn + NQ);                // [NQ]
    int* bwc = reinterpret_cast<int*>(tauS + NQ);               // [NQ]
    int* lockS = bwc + NQ;                                      // [NQ]
    int* cntS = lockS + NQ;                                     // [W][NQ]
    u64* bootw = reinterpret_cast<u64*>(cntS + W * NQ);         // [NQ][kb]
    u64* cand = bootw + NQ * kb;                                // [W][NQ][CAP]
    u64* boot = cand;                                           // [NQ][W*16], dead before cand is used

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.y * NQ;
    const int nqt = min(NQ, p.nq - q0);  // valid queries of this pass
    const int k = p.k;
    const int nsteps = p.row_slots >> 2;  // one k-step = 64 bytes of a row: 16 floats or 32 bf16
    const int t0 = blockIdx.x * p.tiles_per_block;
    const int t1 = min(t0 + p.tiles_per_block, p.tiles_total);
    const bool l2 = p.metric == ISE_METRIC_L2;
    STAMP(0);

    auto load_chunk = [&](f32x4(&a)[CH], int tile, int s0) {
        const char* base = static_cast<const char*>(p.xb) +
                           ((((size_t)tile * 16 + c) * p.row_slots + 4 * s0 + g) << 4);
        if (ABL(32)) base = static_cast<const char*>(p.xb) + (((size_t)c * p.row_slots + g) << 4);  // dev: L1-hot
#pragma unroll
        for (int s = 0; s < CH; s++) a[s] = *reinterpret_cast<const f32x4*>(base + 64 * s);
    };
    auto load_norms = [&](int tile) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(p.norms + (size_t)tile * 16 + 4 * g);
    };

    // ---- query staging, step 1: REQUEST the query tiles first (small, L2-resident after
    // the first block): issued behind the index prefetch they would queue for microseconds.
    // TPR threads per query row, QV 16-byte pieces per thread and tile.
    // A thread handles 16-byte LDS slots j4 = t, t + TPR, ...: 4 floats (f32 store) or 8
    // bf16 converted from 8 floats (bf16 store), i.e. FPS float4 loads per slot.
    constexpr int FPS = BF16 ? 2 : 1;
    constexpr int QV = 8;              // float4 registers per thread and tile
    constexpr int QVS = QV / FPS;      // slots per thread and tile
    const int S4 = S >> 2;             // 16-byte slots per LDS query row
    const int dslots = BF16 ? (p.d >> 3) : (p.d >> 2);  // slots that carry data (vector path only)
    const bool vec_q = (p.d & (BF16 ? 7 : 3)) == 0 && ((reinterpret_cast<uintptr_t>(p.q) & 15) == 0) &&
                       S4 <= TPR * QVS;
    f32x4 qv[T][QV];
    f32x4 muv[SHIFT ? QVS : 1];
    if (SHIFT && vec_q) {
#pragma unroll
        for (int i = 0; i < QVS; i++) {
            const int j4 = tid % TPR + i * TPR;
            muv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (j4 < dslots) muv[i] = *reinterpret_cast<const f32x4*>(p.mu + 4 * j4);
        }
    }
    if (vec_q) {
#pragma unroll
        for (int tq = 0; tq < T; tq++) {
            const int cc = tq * 16 + tid / TPR, t = tid % TPR;
            const bool rowok = cc < nqt && !ABL(1);
            const float* src = p.q + (size_t)(q0 + (rowok ? cc : 0)) * p.d;
#pragma unroll
            for (int i = 0; i < QVS; i++) {
                const int j4 = t + i * TPR;
#pragma unroll
                for (int f = 0; f < FPS; f++) {
                    qv[tq][i * FPS + f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (rowok && j4 < dslots)
                        qv[tq][i * FPS + f] = *reinterpret_cast<const f32x4*>(src + 4 * (j4 * FPS + f));
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- register ring of R index chunks (CH k-steps x 16 rows each).  Chunk positions
    // run tile-major over this wave's row tiles (t0 + w, + W, ...).  The first R - 1
    // chunks are requested now: their HBM latency overlaps the rest of the staging.
    constexpr int R = T == 1 ? 2 : (T == 2 ? 4 : 3);
    const bool has_work = (t0 + w) < t1 && !ABL(16);
    f32x4 A[R][CH];
    int ltile = t0 + w, ls0 = 0;  // position of the next chunk to LOAD (clamped at the end)
    auto advance_load = [&]() {
        int ns = ls0 + CH, nt = ltile;
        if (ns >= nsteps) { ns = 0; nt = ltile + W; }
        if (nt < t1) { ltile = nt; ls0 = ns; }  // past the end: keep re-reading the last chunk
    };
    if (has_work) {
#pragma unroll
        for (int j = 0; j < R - 1; j++) {
            load_chunk(A[j], ltile, ls0);
            advance_load();
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- query staging, step 2: into LDS (zero padded to NQ x S units) with |x|^2.  With bf16
    // storage the queries are rounded to bf16 as well and |x|^2 is taken of the rounded values.
    auto to_bf16_pair = [](float lo, float hi) -> uint32_t {
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;
        return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    };
    auto bf16_round = [](float v) -> float { return (float)(__bf16)v; };
#pragma unroll
    for (int tq = 0; tq < T; tq++) {
        const int cc = tq * 16 + tid / TPR, t = tid % TPR;
        float sn = 0.f;
        if (vec_q) {
#pragma unroll
            for (int i = 0; i < QVS; i++) {
                const int j4 = t + i * TPR;
                if (j4 < S4) {
                    if (BF16) {
                        const f32x4 v0 = qv[tq][i * FPS], v1 = qv[tq][i * FPS + FPS - 1];
                        u32x4 o;
                        o[0] = to_bf16_pair(v0[0], v0[1]);
                        o[1] = to_bf16_pair(v0[2], v0[3]);
                        o[2] = to_bf16_pair(v1[0], v1[1]);
                        o[3] = to_bf16_pair(v1[2], v1[3]);
                        *reinterpret_cast<u32x4*>(qs + cc * S + 4 * j4) = o;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float r0 = bf16_round(v0[e]), r1 = bf16_round(v1[e]);
                            sn = fmaf(r0, r0, sn);
                            sn = fmaf(r1, r1, sn);
                        }
                    } else {
                        f32x4 v = qv[tq][i];
                        if (SHIFT) {
                            if (cc < nqt) v = v - muv[i];  // padding rows stay zero
                            if (cc == 0) *reinterpret_cast<f32x4*>(mus + 4 * j4) = muv[i];
                        }
                        *reinterpret_cast<f32x4*>(qs + cc * S + 4 * j4) = v;
                        sn = fmaf(v[0], v[0], sn);
                        sn = fmaf(v[1], v[1], sn);
                        sn = fmaf(v[2], v[2], sn);
                        sn = fmaf(v[3], v[3], sn);
                    }
                }
            }
        } else {  // odd d, unaligned queries or very long rows: scalar path, one 4-byte unit at a time
            const bool rowok = cc < nqt && !ABL(1);
            const float* src = p.q + (size_t)(q0 + (rowok ? cc : 0)) * p.d;
            for (int j = t; j < S; j += TPR) {
                if (BF16) {
                    const float lo = (rowok && 2 * j < p.d) ? src[2 * j] : 0.f;
                    const float hi = (rowok && 2 * j + 1 < p.d) ? src[2 * j + 1] : 0.f;
                    reinterpret_cast<uint32_t*>(qs)[cc * S + j] = to_bf16_pair(lo, hi);
                    const float r0 = bf16_round(lo), r1 = bf16_round(hi);
                    sn = fmaf(r0, r0, sn);
                    sn = fmaf(r1, r1, sn);
                } else {
                    const float m = (SHIFT && j < p.d) ? p.mu[j] : 0.f;
                    const float v = (rowok && j < p.d) ? src[j] - m : 0.f;
                    qs[cc * S + j] = v;
                    if (SHIFT && cc == 0) mus[j] = m;
                    sn = fmaf(v, v, sn);
                }
            }
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sn += __shfl_xor(sn, o);
        if (t == 0) xn[cc] = sn;
    }
    for (int i = tid; i < NQ; i += BLOCK_THREADS) {
        tauS[i] = TAU0;
        bwc[i] = 0;
        lockS[i] = 0;
    }
    __syncthreads();
    STAMP(1);
    CSTAMP(8);

    const float* qrow = qs + c * S + 4 * g;
    const bool use_floor = p.floor_keys != nullptr;

    float xq_n[T];
    u64 tau[T];
    int cnt[T];
    f32x4 acc0[T], acc1[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        xq_n[t] = xn[t * 16 + c];
        tau[t] = TAU0;
        cnt[t] = 0;
        acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    bool booted = false;

    // fold the wave's private list of query qq = 16 t + cq into the block list bootw[qq]
    // (exact sorted top-k of their union) under the query's LDS lock; publish the k-th key
    auto merge_out = [&](int t, int cq) {
        const int qq = t * 16 + cq;
        const int n_ = __builtin_amdgcn_readlane(cnt[t], cq);
        const u64* buf = cand + (size_t)(w * NQ + qq) * CAP;
        if (lane == 0)
            while (atomicCAS(&lockS[qq], 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
        const int nb = bwc[qq];
        u64 kk[1];
        kk[0] = lane < nb ? bootw[qq * kb + lane] : (lane - nb < n_ ? buf[lane - nb] : KEY_PAD);
        wave_lds_fence();
        u64 ktau = KEY_PAD;
        const int nw = wave_select<1>(kk, nb + n_, k, bootw + qq * kb, &ktau);  // nb + n_ <= kb + CAP <= 64
        if (lane == 0) {
            bwc[qq] = nw;
            if (nw == k) tauS[qq] = ktau;  // never above the old value: the union only adds keys
        }
        wave_lds_fence();
        if (lane == 0) atomicExch(&lockS[qq], 0);
        if (c == cq) {
            cnt[t] = 0;
            if (nw == k) tau[t] = min_u64(tau[t], ktau);
        }
    };

    // boot: all W*16 first-tile keys of a query -> block list + block threshold
    auto boot_phase = [&](const u64(&key)[T][4]) {
        STAMP(2);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) boot[(size_t)(t * 16 + c) * (W * 16) + w * 16 + 4 * g + j] = key[t][j];
        __syncthreads();
        STAMP(7);
        for (int qq = w; qq < NQ; qq += W) {
            u64 kk[KPLB];
#pragma unroll
            for (int e = 0; e < KPLB; e++)
                kk[e] = (lane + 64 * e) < W * 16 ? boot[(size_t)qq * (W * 16) + lane + 64 * e] : KEY_PAD;
            u64 ktau = TAU0;
            const int nw = wave_cut<KPLB>(kk, W * 16, k, kb, bootw + qq * kb, &ktau);
            if (lane == 0) {
                bwc[qq] = nw;
                tauS[qq] = ktau;  // TAU0 when fewer than k real keys were seen
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; t++) tau[t] = tauS[t * 16 + c];
        booted = true;
        STAMP(3);
    };

    // score of (query tile t, row slot j) exactly as it is keyed: squared L2 in the
    // |x|^2 + |y|^2 - 2 x.y form clamped at 0 (NaN kept), or minus the inner product
    auto score = [&](int t, float dotj, float ynj) -> float {
        if (l2) {
            const float sc = (xq_n[t] + ynj) - 2.f * dotj;
            return sc < 0.f ? 0.f : sc;  // keeps NaN (Faiss: if (dis < 0) dis = 0)
        }
        return -dotj;
    };
    auto make_keys = [&](int etile, const f32x4(&sc)[T], u64(&key)[T][4]) {
        const long long row0 = (long long)etile * 16 + 4 * g;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bool ok = (row0 + j < p.n) && (sc[t][j] < FLT_MAX) && (t * 16 + c < nqt);
                const u64 kj = ((u64)ord_f32(sc[t][j]) << 32) | (uint32_t)((uint32_t)(row0 + j) + p.id_base);
                if (use_floor) ok = ok && (kj > p.floor_keys[q0 + min(t * 16 + c, nqt - 1)]);  // multi-pass k only
                key[t][j] = ok ? kj : KEY_PAD;
            }
    };

    float tau_sc[T];  // float image of tau's score part: a conservative pre-filter
#pragma unroll
    for (int t = 0; t < T; t++) tau_sc[t] = FLT_MAX;

    auto epilogue = [&](int etile, f32x4 yn) {
        f32x4 sc[T];
#pragma unroll
        for (int t = 0; t < T; t++) {
            const f32x4 dot = acc0[t] + acc1[t];
            acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) sc[t][j] = score(t, dot[j], yn[j]);
        }
        if (!booted) {
            u64 key[T][4];
            make_keys(etile, sc, key);
            boot_phase(key);
#pragma unroll
            for (int t = 0; t < T; t++) tau_sc[t] = unord_f32((uint32_t)(tau[t] >> 32));
            return;
        }
        // fast path: a row can only enter if its score does not exceed the threshold's
        // score (ties on the score are settled by the exact key compare below)
        bool pend = false;
#pragma unroll
        for (int t = 0; t < T; t++) {
            tau[t] = min_u64(tau[t], tauS[t * 16 + c]);  // other waves' merges tighten it
            tau_sc[t] = unord_f32((uint32_t)(tau[t] >> 32));
        }
#pragma unroll
        for (int t = 0; t < T; t++)
            pend = pend || sc[t][0] <= tau_sc[t] || sc[t][1] <= tau_sc[t] || sc[t][2] <= tau_sc[t] ||
                   sc[t][3] <= tau_sc[t];
        if (__any(pend) && !ABL(2)) {
            u64 key[T][4];
            make_keys(etile, sc, key);
            const u64 qmask = 0x0001000100010001ull << c;  // lanes holding the same query
            const u64 lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int t = 0; t < T; t++) {
                u64* mybuf = cand + (size_t)(w * NQ + t * 16 + c) * CAP;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool v = key[t][j] < tau[t];
                    const u64 m = __ballot(v);
                    if (m) {
                        const u64 mq = m & qmask;
                        if (v) mybuf[cnt[t] + __popcll(mq & lt_mask)] = key[t][j];
                        cnt[t] += __popcll(mq);
                        u64 nm = __ballot(cnt[t] >= MERGE_TRIG) & 0xFFFFull;
                        while (nm) {
                            const int cq = __ffsll((long long)nm) - 1;
                            nm &= nm - 1;
                            merge_out(t, cq);
                        }
                    }
                }
                tau_sc[t] = unord_f32((uint32_t)(tau[t] >> 32));
            }
        }
    };

    // B operand (queries, from LDS) is software-pipelined one k-step ahead of the MFMAs
    // that consume it, across chunk boundaries too: bcur holds the B fragments of the
    // next step to be computed.
    f32x4 bcur[T];
    auto load_b = [&](f32x4(&b)[T], int step) {
#pragma unroll
        for (int t = 0; t < T; t++) b[t] = *reinterpret_cast<const f32x4*>(qrow + (size_t)t * 16 * S + 16 * step);
    };
    auto compute_chunk = [&](const f32x4(&a)[CH], int s0, int next_first_step) {
#pragma unroll
        for (int s = 0; s < CH; s++) {
            f32x4 bnext[T];
            load_b(bnext, s + 1 < CH ? s0 + s + 1 : next_first_step);
            f32x4 as = a[s];
            if (SHIFT) as = as - *reinterpret_cast<const f32x4*>(mus + 4 * g + 16 * (s0 + s));
#pragma unroll
            for (int t = 0; t < T; t++) {
                if (BF16) {  // one 16x16x32 bf16 MFMA per k-step (8 bf16 per lane and operand)
                    const bf16x8 av = __builtin_bit_cast(bf16x8, a[s]), bv = __builtin_bit_cast(bf16x8, bcur[t]);
                    if (s & 1) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc1[t], 0, 0, 0);
                    else acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc0[t], 0, 0, 0);
                } else {     // four 16x16x4 fp32 MFMAs per k-step (exact fp32 fmaf chains)
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[0], bcur[t][0], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[1], bcur[t][1], acc1[t], 0, 0, 0);
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[2], bcur[t][2], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[3], bcur[t][3], acc1[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < T; t++) bcur[t] = bnext[t];
        }
    };

    // ---- main loop.  Each step: request the chunk R - 1 positions ahead (unconditionally:
    // no control-flow join between a load and its use, so hipcc emits counted vmcnt waits
    // and the ring stays in flight), then run the MFMAs of the oldest chunk.  The row
    // norms of the tile being computed are requested first, so the epilogue's wait on them
    // never drains the younger index loads.
    if (has_work) {
        int tile = t0 + w, s0 = 0;  // position of the chunk being COMPUTED
        load_b(bcur, 0);
        bool done = false;
        while (!done) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (!done) {
                    const f32x4 yn = load_norms(tile);
                    load_chunk(A[(j + R - 1) % R], ltile, ls0);
                    advance_load();
                    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs
                    int ns0 = s0 + CH, ntile = tile;
                    if (ns0 >= nsteps) { ns0 = 0; ntile = tile + W; }
                    compute_chunk(A[j], s0, ns0);
                    if (ns0 == 0 && !ABL(8)) epilogue(tile, yn);
                    done = ntile >= t1;
                    tile = ntile; s0 = ns0;
                }
            }
        }
    }
    if (!booted) {  // a wave without a row tile still takes part in the two boot barriers
        u64 none[T][4];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) none[t][j] = KEY_PAD;
        boot_phase(none);
    }

    // ---- final: per query, exact sorted top-k of bootw + the W private lists
    STAMP(4);
    CSTAMP(9);
#pragma unroll
    for (int t = 0; t < T; t++)
        if (g == 0) cntS[w * NQ + t * 16 + c] = cnt[t];
    __syncthreads();
    STAMP(5);
    for (int qq = w; qq < NQ && !ABL(4); qq += W) {
        int P[W + 2];
        P[0] = 0;
        P[1] = bwc[qq];
#pragma unroll
        for (int i = 0; i < W; i++) P[i + 2] = P[i + 1] + cntS[i * NQ + qq];
        const int n = P[W + 1];
        u64 kk[KPLF];
#pragma unroll
        for (int e = 0; e < KPLF; e++) {
            kk[e] = KEY_PAD;
            const int idx = lane + 64 * e;
            if (64 * e < n) {
                if (idx < P[1]) kk[e] = bootw[qq * kb + idx];
#pragma unroll
                for (int i = 0; i < W; i++)
                    if (idx >= P[i + 1] && idx < P[i + 2])
                        kk[e] = cand[(size_t)(i * NQ + qq) * CAP + idx - P[i + 1]];
            }
        }
        u64* out = p.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NQ + qq) * k;
        u64 kth_unused;
        const int nw = wave_select<KPLF>(kk, n, k, out, &kth_unused);
        if (lane >= nw && lane < k) out[lane] = KEY_PAD;  // k <= KB_MAX <= 64
    }
    STAMP(6);
}

// ---------------------------------------------------------------- assign kernel
// k = 1 search of MANY rows against a SMALL index: nearest-centroid assignment
// (FaissKMeans.transform, backend/kmeans_faiss.py:46-50; BASELINE config 4: 50M x 128
// SIFT-like rows vs 4096 centroids).  GEMM-shaped and MFMA-bound (2*n*K*d flop against
// 4*n*d bytes), so the roles flip with respect to scan_kernel: the big operand X is read
// once from HBM into registers (XT row tiles of 16 per wave), the centroids stream from
// L2 through LDS in stages of ASSIGN_CS, and each lane keeps a running best (score, id)
// per row slot -- no top-k machinery at all.  fp32 MFMA 16x16x4, exact fmaf chains.
//   score = x.c              (inner product)    -> arg max
//   score = x.c - |c|^2 / 2  (L2; |c|^2/2 rides in as the accumulator's initial value)
// Ties go to the lowest centroid id.
#define ASSIGN_CS_MAX 128 /* centroids per LDS stage (fewer when rows are long) */
struct AssignParams {
    const float* x;      // [n][d] rows to assign (unpadded)
    const float* cb;     // [K][dp] centroids, padded rows (the index's xb)
    const float* cnorm;  // [K] |c|^2, or |c - mu|^2 when the index is shifted
    const float* mu;     // [dp] the index's shift vector or null: x and c are both shifted by it
    long long n;
    int d, dp, cs_stride, K, metric;
    int cs;        // centroids per LDS stage, a multiple of 16
    long long* I;  // [n]
    float* D;      // [n] or null
};

template <int NS, int XT>
__global__ __launch_bounds__(512, 2) void assign_kernel(const AssignParams p) {
    constexpr int W = 8;
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = p.cs_stride;
    const int CS = p.cs;
    float* cs = reinterpret_cast<float*>(smem);  // [CS][S]
    float* cn = cs + CS * S;                     // [CS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const bool l2 = p.metric == ISE_METRIC_L2;
    const long long rows_per_block = (long long)W * XT * 16;
    const long long nslabs = (p.n + rows_per_block - 1) / rows_per_block;
    const int nstages = (p.K + CS - 1) / CS;
    const bool vec_x = (p.d & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
    const float* brow = cs + c * S + 4 * g;

    for (long long slab = blockIdx.x; slab < nslabs; slab += gridDim.x) {
        const long long wrow0 = slab * rows_per_block + (long long)w * XT * 16;
        // A fragments of this wave's XT row tiles: lane (r = c, g) holds X[r][16 s + 4 g .. + 3]
        f32x4 A[XT][NS];
        float xnp[XT];  // partial |x|^2 of row r over this lane's k-slices
#pragma unroll
        for (int xt = 0; xt < XT; xt++) {
            const long long r = min(wrow0 + xt * 16 + c, p.n - 1);
            const float* xr = p.x + (size_t)r * p.d;
            xnp[xt] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int col = 16 * s + 4 * g;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (vec_x) {
                    if (col < p.d) v = *reinterpret_cast<const f32x4*>(xr + col);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (col + e < p.d) v[e] = xr[col + e];
                }
                if (p.mu) v = v - *reinterpret_cast<const f32x4*>(p.mu + col);  // mu is zero beyond d, like v
                A[xt][s] = v;
                xnp[xt] = fmaf(v[0], v[0], xnp[xt]);
                xnp[xt] = fmaf(v[1], v[1], xnp[xt]);
                xnp[xt] = fmaf(v[2], v[2], xnp[xt]);
                xnp[xt] = fmaf(v[3], v[3], xnp[xt]);
            }
        }
        float best[XT][4];
        int bidx[XT][4];
#pragma unroll
        for (int xt = 0; xt < XT; xt++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                best[xt][j] = -FLT_MAX;
                bidx[xt][j] = -1;
            }

        for (int stage = 0; stage < nstages; stage++) {
            __syncthreads();  // everyone is done with the previous stage's centroids
            const int c0 = stage * CS;
            for (int i = tid; i < CS * (p.dp >> 2); i += 512) {
                const int cr = i / (p.dp >> 2), j4 = i - cr * (p.dp >> 2);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c0 + cr < p.K) {
                    v = *reinterpret_cast<const f32x4*>(p.cb + (size_t)(c0 + cr) * p.dp + 4 * j4);
                    if (p.mu) v = v - *reinterpret_cast<const f32x4*>(p.mu + 4 * j4);
                }
                *reinterpret_cast<f32x4*>(cs + cr * S + 4 * j4) = v;
            }
            if (tid < CS) cn[tid] = c0 + tid < p.K ? p.cnorm[c0 + tid] : 0.f;
            __syncthreads();
#pragma unroll 1
            for (int ct = 0; ct < CS / 16; ct++) {
                const int cid = c0 + ct * 16 + c;
                const float init = l2 ? -0.5f * cn[ct * 16 + c] : 0.f;
                f32x4 acc0[XT], acc1[XT];
#pragma unroll
                for (int xt = 0; xt < XT; xt++) {
                    acc0[xt] = (f32x4){init, init, init, init};
                    acc1[xt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                const float* bp = brow + (size_t)ct * 16 * S;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(bp + 16 * s);
#pragma unroll
                    for (int xt = 0; xt < XT; xt++) {
                        acc0[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xt][s][0], b[0], acc0[xt], 0, 0, 0);
                        acc1[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xt][s][1], b[1], acc1[xt], 0, 0, 0);
                        acc0[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xt][s][2], b[2], acc0[xt], 0, 0, 0);
                        acc1[xt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xt][s][3], b[3], acc1[xt], 0, 0, 0);
                    }
                }
                const bool cvalid = cid < p.K;
#pragma unroll
                for (int xt = 0; xt < XT; xt++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float sc = acc0[xt][j] + acc1[xt][j];
                        const bool better = cvalid && sc > best[xt][j];  // strict: earlier (lower) id wins ties; NaN never
                        best[xt][j] = better ? sc : best[xt][j];
                        bidx[xt][j] = better ? cid : bidx[xt][j];
                    }
            }
        }

        // ---- per row: best over the 16 lanes of its DPP row (same g), then write
#pragma unroll
        for (int xt = 0; xt < XT; xt++) {
            // |x|^2 of row r: sum of the 4 g-lanes holding row r (lanes r, r+16, r+32, r+48)
            float xn = xnp[xt];
            xn += __shfl_xor(xn, 16);
            xn += __shfl_xor(xn, 32);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float bs = best[xt][j];
                int bi = bidx[xt][j];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    const float os = __shfl_xor(bs, o);
                    const int oi = __shfl_xor(bi, o);
                    const bool take = oi >= 0 && (bi < 0 || os > bs || (os == bs && oi < bi));
                    bs = take ? os : bs;
                    bi = take ? oi : bi;
                }
                const int rr = 4 * g + j;                 // row slot this lane group reports
                const float xr2 = __shfl(xn, rr);         // lane rr (g = 0 copy) holds |x_rr|^2
                const long long row = wrow0 + xt * 16 + rr;
                if (c == 0 && row < p.n) {
                    float dist;
                    bool ok = bi >= 0;
                    if (l2) {
                        dist = xr2 - 2.f * bs;
                        dist = dist < 0.f ? 0.f : dist;
                        ok = ok && dist < FLT_MAX;
                        if (!ok) dist = FLT_MAX;
                    } else {
                        dist = bs;
                        ok = ok && bs > -FLT_MAX;
                        if (!ok) dist = -FLT_MAX;
                    }
                    p.I[row] = ok ? (long long)bi : -1ll;
                    if (p.D) p.D[row] = dist;
                }
            }
        }
    }
}

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

// ---------------------------------------------------------------- row helpers
// |y|^2 per row, wave per row, fixed summation order (lanes stride float4, then
// an xor butterfly): deterministic for a given dp.
__global__ __launch_bounds__(256) void norms_kernel(const float* __restrict__ x, long long row0,
                                                    long long n, int dp, const float* __restrict__ mu,
                                                    float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = row0 + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= row0 + n) return;
    const float* xr = x + (size_t)r * dp;
    float s = 0.f;
    for (int j = lane * 4; j < dp; j += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + j);
        if (mu) v = v - *reinterpret_cast<const f32x4*>(mu + j);  // |y - mu|^2 (padding columns: 0 - 0)
        s = fmaf(v[0], v[0], s);
        s = fmaf(v[1], v[1], s);
        s = fmaf(v[2], v[2], s);
        s = fmaf(v[3], v[3], s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) out[r] = s;
}

// mean of the first `rows` rows per column (d columns of a padded row).  Two levels, both in a
// fixed order (COLMEAN_GROUPS row groups summed in row order, then the groups in group order):
// deterministic, so every index built from the same leading rows gets the same shift vector.
#define COLMEAN_GROUPS 64
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, long long rows, int d, int dp,
                                                      float* __restrict__ partial /* [GROUPS][dp] */) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int gidx = blockIdx.y;
    if (j >= dp) return;
    const long long per = (rows + COLMEAN_GROUPS - 1) / COLMEAN_GROUPS;
    const long long r0 = gidx * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (j < d)
        for (long long r = r0; r < r1; r++) {
            const float v = x[(size_t)r * dp + j];
            if (fabsf(v) <= FLT_MAX) s += v;  // NaN / inf entries must not poison every distance
        }
    partial[(size_t)gidx * dp + j] = s;
}
__global__ __launch_bounds__(256) void col_mean_kernel(const float* __restrict__ partial, long long rows, int d, int dp,
                                                       float* __restrict__ mu) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dp) return;
    float s = 0.f;
    for (int gi = 0; gi < COLMEAN_GROUPS; gi++) s += partial[(size_t)gi * dp + j];
    const float m = s / (float)rows;
    mu[j] = (j < d && fabsf(m) <= FLT_MAX) ? m : 0.f;
}

// |y|^2 of bf16 rows (the values the bf16 scan multiplies), fp32 accumulation
__global__ __launch_bounds__(256) void norms_bf16_kernel(const __bf16* __restrict__ x, long long row0, long long n,
                                                         int dp, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = row0 + (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= row0 + n) return;
    const __bf16* xr = x + (size_t)r * dp;
    float s = 0.f;
    for (int j = lane * 8; j < dp; j += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(xr + j);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float f = (float)v[e];
            s = fmaf(f, f, s);
        }
    }
    s = wave_sum_f32(s);
    if (lane == 0) out[r] = s;
}

// float32 rows (unpadded) -> padded bf16 index rows (round to nearest even)
__global__ __launch_bounds__(256) void pad_rows_bf16_kernel(const float* __restrict__ src, long long n, int d,
                                                            __bf16* __restrict__ dst, int dp) {
    const long long total = n * dp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / dp;
        const int j = (int)(i - r * dp);
        dst[i] = (__bf16)(j < d ? src[(size_t)r * d + j] : 0.f);
    }
}

// padded bf16 rows -> float32 rows of d (reconstruct / write_index)
__global__ __launch_bounds__(256) void unpack_rows_bf16_kernel(const __bf16* __restrict__ src, long long n, int d,
                                                               int dp, float* __restrict__ dst) {
    const long long total = n * d;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / d;
        const int j = (int)(i - r * d);
        dst[i] = (float)src[(size_t)r * dp + j];
    }
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
    constexpr int VMAX = 8;  // up to 8 float4 per lane: rows of <= 2048 floats stay in registers
    const bool vec = (d & 3) == 0 && d <= 64 * 4 * VMAX && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) {  // one read, one write
        f32x4 v[VMAX];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VMAX; i++) {
            const int j = (lane + 64 * i) * 4;
            v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (j < d) v[i] = *reinterpret_cast<const f32x4*>(xr + j);
            s = fmaf(v[i][0], v[i][0], s);
            s = fmaf(v[i][1], v[i][1], s);
            s = fmaf(v[i][2], v[i][2], s);
            s = fmaf(v[i][3], v[i][3], s);
        }
        s = wave_sum_f32(s);
        if (s > 0.f) {
            const float inv = (float)(1.0 / (double)sqrtf(s));
#pragma unroll
            for (int i = 0; i < VMAX; i++) {
                const int j = (lane + 64 * i) * 4;
                if (j < d) *reinterpret_cast<f32x4*>(xr + j) = v[i] * inv;
            }
        }
        return;
    }
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
    int storage = ISE_STORE_F32;  // element type of xb
    long long n = 0, cap = 0;
    void* xb = nullptr;
    float* norms = nullptr;
    float* mu = nullptr;       // [dp] shift vector (fp32 L2 only), zero until shift_set
    bool shift_set = false;    // fixed once: at the first add, or by ise_index_set_shift before it
    // workspaces (grown lazily, guarded by mu): calls rotate through NWS slots and a
    // slot's reuse is ordered behind its previous use with an event, so searches on
    // different streams may be in flight together
    struct WorkSlot {
        u64* part = nullptr;
        size_t part_elems = 0;
        u64* keys_tmp = nullptr;  // multi-pass k scratch
        size_t keys_tmp_elems = 0;
        hipEvent_t done = nullptr;
        bool used = false;
    };
    static constexpr int NWS = 4;
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
#define LDS_LIMIT (160 * 1024)
template <int CH, int W, int T, bool BF16, bool SHIFT>
static void launch_one(dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    static bool attr_done = false;  // benign race: the attribute is idempotent
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&scan_kernel<CH, W, T, BF16, SHIFT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        attr_done = true;
    }
    hipLaunchKernelGGL((scan_kernel<CH, W, T, BF16, SHIFT>), grid, dim3(W * 64), lds, st, sp);
}
template <int W, int T, bool BF16, bool SHIFT>
static void launch_scan_ch(int ch, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    switch (ch) {
        case 8: launch_one<8, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
        case 4: launch_one<4, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
        case 2: launch_one<2, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
        default: launch_one<1, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
    }
}
// variants built: (waves, T) in {(8,1), (4,1), (8,2), (8,3)}; 4 waves exist for one query
// tile only (long rows, where 8 waves' lists no longer fit beside the tile)
template <bool BF16, bool SHIFT>
static void launch_scan_v(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    if (T == 1 && waves == 4) launch_scan_ch<4, 1, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 1) launch_scan_ch<8, 1, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 2) launch_scan_ch<8, 2, BF16, SHIFT>(ch, grid, lds, st, sp);
    else launch_scan_ch<8, 3, BF16, SHIFT>(ch, grid, lds, st, sp);
}
static void launch_scan(const ise_index* h, int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st,
                        const ScanParams& sp) {
    if (h->storage == ISE_STORE_BF16) launch_scan_v<true, false>(ch, waves, T, grid, lds, st, sp);
    else if (uses_shift(h)) launch_scan_v<false, true>(ch, waves, T, grid, lds, st, sp);
    else launch_scan_v<false, false>(ch, waves, T, grid, lds, st, sp);
}

struct ScanPlan {
    int nblocks, tiles_total, tiles_per_block, nqt, ch, kpass, kb, waves, T;
    size_t lds;
};

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
    static const double pass_cost[4] = {0.0, 1.0, 1.08, 1.25};
    int tmax = 3;
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_TMAX")) tmax = std::min(3, std::max(1, atoi(e)));
#endif
    pl->T = 0;
    double best = 0;
    for (int t = 1; t <= tmax; t++) {
        int wv = 0;
        size_t lds = 0;
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
    if (pl->T == 1 && pl->ch > 4) pl->ch = 4;
    pl->tiles_total = (int)((h->n + 15) / 16);
    int blocks_per_cu = (pl->T == 1 && pl->lds <= LDS_LIMIT / 2) ? 2 : 1;
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

// enqueue one search batch; outputs (D, I) and/or keys.  Nothing here blocks.
static int search_enqueue(ise_index* h, const float* q_dev, long long nq, int k, uint32_t id_base, float* D_dev,
                          long long* I_dev, u64* keys_out, hipStream_t st, TimedOut* tm) {
    ScanPlan pl;
    int rc = make_plan(h, nq, k, &pl);
    if (rc) return rc;
    ise_index::WorkSlot* w = &h->ws[h->ws_next++ % ise_index::NWS];
    rc = ensure_workspace(w, pl, nq, k);
    if (rc) return rc;
    if (w->used) HIP_TRY(hipStreamWaitEvent(st, w->done, 0));
    struct Release {  // whatever path returns, later users of the slot wait for this call
        ise_index::WorkSlot* w;
        hipStream_t st;
        ~Release() {
            if (hipEventRecord(w->done, st) == hipSuccess) w->used = true;
        }
    } release{w, st};

    ScanParams sp;
    sp.xb = h->xb; sp.norms = h->norms; sp.q = q_dev; sp.mu = h->mu; sp.floor_keys = nullptr; sp.part = w->part;
    sp.n = h->n; sp.d = h->d; sp.dp = h->dp; sp.qs_stride = qs_stride_for(h);
    sp.row_slots = (int)(row_bytes(h) / 16);
    sp.nq = (int)nq; sp.k = pl.kpass; sp.kb = pl.kb; sp.metric = h->metric; sp.id_base = id_base;
    sp.tiles_total = pl.tiles_total; sp.tiles_per_block = pl.tiles_per_block;
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
