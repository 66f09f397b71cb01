// ise_scan.hpp -- the streaming distance + top-k kernel (see ise_knn.hip for the overview).
#pragma once
#include "ise_common.hpp"
#include "ise_scan_params.hpp"
#include "ise_select.hpp"

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
//   exchange (8-wave kernels with T >= 2, once per block): at boot a block publishes, per query, the score of its best boot
//          row (one 8-byte store: launch seq << 32 | ord(score)).  After its first row tile a
//          wave reads the entries of all blocks for its queries: the k-th smallest of those scores
//          is the k-th best of nblocks DISTINCT rows, hence an upper bound of the final k-th
//          distance -- as tight as the k-th best of a W*16*nblocks-row sample (32k-64k rows)
//          instead of the block's own W*16.  It only lowers tauS; entries not yet written (or of
//          an older launch: seq mismatch) just count as absent.  No polling, no ordering needed.
//
// SHIFT (fp32 L2 only): distances are translation invariant, and the expanded form
// |x|^2 + |y|^2 - 2 x.y loses digits when the rows share a large common component
// (CNN embeddings: |y|^2 ~ 1e5, neighbour distances ~ 1e-1).  The index keeps a shift
// vector mu (the column mean of its rows); norms are |y - mu|^2, queries are staged as
// x - mu and the row fragments are shifted in registers before the MFMAs (4 VALU subs per
// k-step), so every term is as small as the data's spread, not its offset.  Rows stay
// stored unshifted (reconstruct / write_index are exact).  What the expanded form still
// loses is bounded rigorously (beta) and the rows are keyed by that lower bound: this
// kernel is the FILTER of the exact search, ise_exact.hpp holds the verifier.
template <int CH, int W, int T, bool BF16, bool SHIFT>
__global__ __launch_bounds__(W * 64, T == 1 ? W / 2 : (W / 4 > 0 ? W / 4 : 1)) void scan_kernel(const ScanParams p) {
    static_assert(!(BF16 && SHIFT), "the shift is applied to fp32 rows only");
    if (p.gate && *p.gate == 0u) return;  // a queued rerun that is not needed
    constexpr int BLOCK_THREADS = W * 64;
    // threshold exchange: compiled into the 8-wave kernels with two or more query tiles, where the candidate
    // bookkeeping is what it saves (nq = 48: 560 -> 508 us; nq = 32: 399 (16 waves) -> 383 us).  Not into the
    // 16-wave two-tile kernel (the read costs more than it saves there: 399 -> 406 us; the host picks it
    // for short indexes, where the exchange would not run anyway) nor at T = 1 (neutral).
    constexpr bool XCHG = T >= 3 || (T == 2 && W == 8);
    constexpr int NQ = 16 * T;                        // queries per block pass
    // threads staging one query row (per tile).  16-wave blocks stage with their first 8 waves: |x|^2 then has the
    // summation order of the 8-wave kernels (and of short_scan_kernel), and the bf16 L2 distances, which carry it,
    // come out the same bits whichever kernel scans (4-wave blocks exist for rows no other kernel takes)
    constexpr int TPR = W >= 16 ? 32 : BLOCK_THREADS / 16;
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

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
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
        // plain loads on purpose: non-temporal loads stream a pure read faster (7.1 vs 6.2 TB/s,
        // scripts/microbench/read_bw.hip) but lose here (369 vs 343 us) -- a wave load covers 16 rows x
        // 64 B, half lines whose other half the next load wants from L2
        if (ABL(128)) return;  // dev: no index loads at all (registers keep whatever they hold)
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
    const bool stager = tid < 16 * TPR;  // whole waves
    if (SHIFT && vec_q && stager) {
#pragma unroll
        for (int i = 0; i < QVS; i++) {
            const int j4 = tid % TPR + i * TPR;
            muv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (j4 < dslots) muv[i] = *reinterpret_cast<const f32x4*>(p.mu + 4 * j4);
        }
    }
    if (vec_q && stager) {
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
    constexpr int R = (T == 1 || W >= 16) ? 2 : (T == 2 ? 4 : (T == 3 ? 3 : 2));
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
        if (!stager) continue;
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
            // (the loads of SB units are requested together, then consumed in the same ascending order as a
            // plain loop would: |x|^2 keeps its summation order, the staging of a 2048-float row its round trips
            // cut by SB -- 8.7 -> see DESIGN.md 4.1)
            const bool rowok = cc < nqt && !ABL(1);
            const float* src = p.q + (size_t)(q0 + (rowok ? cc : 0)) * p.d;
            constexpr int SB = 8;
            for (int j0 = t; j0 < S; j0 += SB * TPR) {
                float lo[SB], hi[SB], m[SB];
#pragma unroll
                for (int b = 0; b < SB; b++) {
                    const int j = j0 + b * TPR;
                    lo[b] = hi[b] = m[b] = 0.f;
                    if (BF16) {
                        if (rowok && j < S && 2 * j < p.d) lo[b] = src[2 * j];
                        if (rowok && j < S && 2 * j + 1 < p.d) hi[b] = src[2 * j + 1];
                    } else {
                        if (SHIFT && j < S && j < p.d) m[b] = p.mu[j];
                        if (rowok && j < S && j < p.d) lo[b] = src[j];
                    }
                }
#pragma unroll
                for (int b = 0; b < SB; b++) {
                    const int j = j0 + b * TPR;
                    if (j < S) {
                        if (BF16) {
                            reinterpret_cast<uint32_t*>(qs)[cc * S + j] = to_bf16_pair(lo[b], hi[b]);
                            const float r0 = bf16_round(lo[b]), r1 = bf16_round(hi[b]);
                            sn = fmaf(r0, r0, sn);
                            sn = fmaf(r1, r1, sn);
                        } else {
                            const float v = (rowok && j < p.d) ? lo[b] - m[b] : 0.f;
                            qs[cc * S + j] = v;
                            if (SHIFT && cc == 0) mus[j] = m[b];
                            sn = fmaf(v, v, sn);
                        }
                    }
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
            if (nw == k) tauS[qq] = min_u64(tauS[qq], ktau);  // the exchange may have set it lower
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
            u64 best = kk[0];
#pragma unroll
            for (int e = 1; e < KPLB; e++) best = min_u64(best, kk[e]);
            best = wave_min_u64(best);
            if (lane == 0) {
                bwc[qq] = nw;
                tauS[qq] = ktau;  // TAU0 when fewer than k real keys were seen
                if (XCHG && p.xchg)
                    __hip_atomic_store(p.xchg + ((size_t)blockIdx.y * NQ + qq) * gridDim.x + blockIdx.x,
                                       ((u64)p.xchg_seq << 32) | (best >> 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; t++) tau[t] = tauS[t * 16 + c];
        booted = true;
        STAMP(3);
    };

    // score of (query tile t, row slot j) exactly as it is keyed.
    //   float32 rows, L2 (SHIFT): a rigorous LOWER BOUND of the direct-difference distance,
    //       lo = (|x-mu|^2 + |y-mu|^2 - 2 (x-mu).(y-mu)) - beta (|x-mu|^2 + |y-mu|^2),
    //     beta covering every rounding between the stored floats and this value.
    //     The scan selects the K' = k + extra rows of smallest lo; the merge stage re-evaluates
    //     them as sum (x-y)^2 and proves (or sends to the exact fallback scan) that no other
    //     row can enter the top k (ise_exact.hpp).  Not clamped: the order is what matters.
    //   bf16 rows, L2: the expanded form clamped at 0 (NaN kept), approximate by construction.
    //   inner product: minus the dot product.
    auto score = [&](int t, float dotj, float ynj) -> float {
        if (l2) {
            const float tt = xq_n[t] + ynj;
            const float sc = tt - 2.f * dotj;
            if (SHIFT) return fmaf(-p.beta, tt, sc);
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
        // fast path: a row can only enter if its score does not exceed the threshold's score
        // (ties on the score are settled by the exact key compare below).  One ballot per
        // (query tile, row slot); keys are built only for the slots some lane may fill.
#pragma unroll
        for (int t = 0; t < T; t++) {
            tau[t] = min_u64(tau[t], tauS[t * 16 + c]);  // other waves' merges tighten it
            tau_sc[t] = unord_f32((uint32_t)(tau[t] >> 32));
        }
        u64 hit[T][4];
        u64 any_hit = 0;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                hit[t][j] = __ballot(sc[t][j] <= tau_sc[t]);
                any_hit |= hit[t][j];
            }
        if (any_hit && !ABL(2)) {
            const long long row0 = (long long)etile * 16 + 4 * g;
            const u64 qmask = 0x0001000100010001ull << c;  // lanes holding the same query
            const u64 lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int t = 0; t < T; t++) {
                u64* mybuf = cand + (size_t)(w * NQ + t * 16 + c) * CAP;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (hit[t][j]) {  // wave-uniform
                        const float s_ = sc[t][j];
                        bool ok = (row0 + j < p.n) && (s_ < FLT_MAX) && (t * 16 + c < nqt);
                        const u64 kj = ((u64)ord_f32(s_) << 32) | (uint32_t)((uint32_t)(row0 + j) + p.id_base);
                        if (use_floor) ok = ok && (kj > p.floor_keys[q0 + min(t * 16 + c, nqt - 1)]);  // multi-pass k only
                        const bool v = ok && kj < tau[t];
                        const u64 m = __ballot(v);
                        if (m) {
                            const u64 mq = m & qmask;
                            if (v) mybuf[cnt[t] + __popcll(mq & lt_mask)] = kj;
                            cnt[t] += __popcll(mq);
                            u64 nm = __ballot(cnt[t] >= MERGE_TRIG) & 0xFFFFull;
                            while (nm) {
                                const int cq = __ffsll((long long)nm) - 1;
                                nm &= nm - 1;
                                merge_out(t, cq);
                            }
                        }
                    }
                }
                tau_sc[t] = unord_f32((uint32_t)(tau[t] >> 32));
            }
        }
    };

    // one-shot threshold exchange (see the header comment): queries w, w + W, ... of this wave.
    // 256 slots (4 per lane) cover the grid, a slot folding G entries by their minimum; the k-th
    // smallest slot value is found bit by bit with ballot counts (absent entries = 0xFFFFFFFF).
    auto exchange = [&]() {
        const int nb = (int)gridDim.x;
        const int G = (nb + 255) >> 8;
        for (int qq = w; qq < NQ; qq += W) {
            const u64* src = p.xchg + ((size_t)blockIdx.y * NQ + qq) * nb;
            auto slot_value = [&](int slot) -> uint32_t {
                uint32_t best = 0xFFFFFFFFu;
                for (int j = 0; j < G; j++) {
                    const int i = slot * G + j;
                    if (i < nb) {
                        const u64 v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((uint32_t)(v >> 32) == p.xchg_seq) best = min(best, (uint32_t)v);
                    }
                }
                return best;
            };
            const uint32_t v0 = slot_value(lane), v1 = slot_value(lane + 64), v2 = slot_value(lane + 128),
                           v3 = slot_value(lane + 192);
            uint32_t prefix = 0;
            int rank = k;  // 1-based rank of the wanted value among those matching the prefix
            for (int b = 31; b >= 0; b--) {
                const uint32_t hm = b == 31 ? 0u : 0xFFFFFFFFu << (b + 1);
                auto zero_here = [&](uint32_t v) { return ((v ^ prefix) & hm) == 0u && ((v >> b) & 1u) == 0u; };
                const int c0 = __popcll(__ballot(zero_here(v0))) + __popcll(__ballot(zero_here(v1))) +
                               __popcll(__ballot(zero_here(v2))) + __popcll(__ballot(zero_here(v3)));
                if (rank > c0) {
                    rank -= c0;
                    prefix |= 1u << b;
                }
            }
            if (prefix != 0xFFFFFFFFu) {  // at least k entries of this launch were there
                const u64 bound = ((u64)prefix << 32) | 0xFFFFFFFFull;  // every id at that score stays admissible
                if (lane == 0) {
                    while (atomicCAS(&lockS[qq], 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
                }
                wave_lds_fence();
                if (lane == 0 && bound < tauS[qq]) tauS[qq] = bound;
                wave_lds_fence();
                if (lane == 0) atomicExch(&lockS[qq], 0);
            }
        }
    };

    // B operand (queries, from LDS) is software-pipelined one k-step ahead of the MFMAs
    // that consume it, across chunk boundaries too: bcur holds the B fragments of the
    // next step to be computed.
    f32x4 bcur[T];
    auto load_b = [&](f32x4(&b)[T], int step) {
        if (ABL(512)) return;  // dev: no LDS reads of the query operand (registers keep whatever they hold)
#pragma unroll
        for (int t = 0; t < T; t++) b[t] = *reinterpret_cast<const f32x4*>(qrow + (size_t)t * 16 * S + 16 * step);
    };
    auto compute_chunk = [&](const f32x4(&a)[CH], int s0, int next_first_step) {
#pragma unroll
        for (int s = 0; s < CH; s++) {
            f32x4 bnext[T];
            if (ABL(64)) {  // dev: no LDS reads, no MFMA -- the loaded data is only summed
#pragma unroll
                for (int t = 0; t < T; t++) acc0[t] += a[s];
                continue;
            }
            load_b(bnext, s + 1 < CH ? s0 + s + 1 : next_first_step);
            f32x4 as = a[s];
            if (SHIFT && !ABL(1024)) as = as - *reinterpret_cast<const f32x4*>(mus + 4 * g + 16 * (s0 + s));
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
        int tiles_done = 0;
        bool exchanged = !XCHG || p.xchg == nullptr || ABL(256);
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
                    if (ns0 == 0) tiles_done++;
                    done = ntile >= t1;
                    tile = ntile; s0 = ns0;
                }
            }
            // outside the unrolled ring steps (one copy of the code): after the wave's second row tile
            // -- and only when enough row tiles remain for the tighter threshold to pay for the read
            if constexpr (XCHG) {
                if (!exchanged && tiles_done >= 1) {
                    if (t1 - tile >= 4 * W) exchange();
                    exchanged = true;
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
