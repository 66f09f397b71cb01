// ise_short_scan.hpp -- the whole search of one batch of <= 16 queries against a SHORT index in ONE launch.
//
// The reference's own regime is short indexes (about 1 k images, backend/utils.py:309-310; one query per
// request, backend/engine.py:50-55); BASELINE config 2 is 100k x 512 and every rank of the 8-GPU run scans a
// 125k-row shard.  There a wave of scan_kernel (ise_scan.hpp) owns one or two row tiles: its boot (dump, two
// barriers, windowed cut -- with nothing in flight meanwhile), its final selection and the two launches behind
// it (merge + re-rank, the exact scan's gate) ARE the search: 45.8 + 10.5 + 3.9 us at 100k x 512 where the
// rows stream in 29 us (profiles/r03/README.md).  This kernel keeps the row stream, the MFMA tile and the
// arithmetic of scan_kernel -- the keys it selects are the same keys, bit for bit -- and replaces the rest:
//
//   stream   a wave runs its row tiles back to back; a tile's 16 x 16 scores go to LDS as plain floats
//            (one 16-byte store per lane and tile: the row id is the slot index).  No threshold, no lists,
//            no barrier until the block's rows are done, and R - 1 chunks are requested BEFORE the queries
//            are staged, so the stream also covers the staging.
//   select   one barrier; per query one wave rebuilds the keys of the block's rows and selects the exact
//            sorted top k (wave_select) -> the block's list in `part`, as scan_kernel leaves it.
//   tail     every block takes a ticket (one agent-scope add behind its drained stores and a release).
//            The holders of the LAST min(nq, blocks) tickets are the tail workers, one query each: a worker
//            waits until all blocks have arrived (a relaxed poll; the holder of the last ticket does not
//            wait at all), acquires, and runs what merge_kernel runs: the 8-wave merge of the per-block lists
//            and, for float32 L2, the direct-difference re-rank with its certificate (ise_exact.hpp).
//
// Progress: a waiting worker holds one block slot; at most 16 per launch wait and at most NWS launches are
// in flight (one per workspace slot), so waiters hold at most 96 of the chip's >= 512 slots and the blocks
// they wait for are always scheduled.  The wait is bounded all the same: a worker that gives up puts its
// query on the exact scan's fallback list (float32 L2: the gated exact scan then answers it, the result stays
// exact), or emits an empty result and counts the event (inner product / bf16: no fallback exists).
//
// The arrival counter is monotonic (never reset): the host passes its value before the launch.
#pragma once
#include "ise_common.hpp"
#include "ise_exact.hpp"
#include "ise_merge.hpp"
#include "ise_scan_params.hpp"
#include "ise_select.hpp"

#define SHORT_W 8          /* waves per block: the tail is merge_kernel's 8-wave merge */
#define SHORT_KPL 8        /* keys per lane of the block selection: <= 512 rows (32 tiles) per block */
#define SHORT_TPB_MAX (SHORT_KPL * 64 / 16)
#define SHORT_WAIT_TICKS 200000000ull /* 2 s of the 100 MHz clock: a worker gives up waiting for the other blocks */

struct ShortTailParams {
    MergeParams mp;            // lists = the per-block lists this launch writes (n_lists = gridDim.x, k = keys per list)
    ExactParams xp;            // RERANK instantiations only
    unsigned int* arrive;      // the slot's arrival counter (monotonic)
    unsigned int arrive_base;  // its value before this launch
    unsigned long long* gave_up;  // counts tail workers that stopped waiting (never, in practice)
};

// dump row stride in floats: whole 32-float groups + 4, so that the 8 lanes of a 16-byte store group
// (8 queries, same rows) fall into different banks
__host__ __device__ constexpr int short_dump_stride(int tiles_per_block) { return (tiles_per_block * 16 + 31) / 32 * 32 + 4; }
// bytes of the tail's LDS image: merge scratch | merged keys | rerank image (query, candidates, exact keys)
__host__ __device__ constexpr size_t short_tail_bytes(int dp, int kc) {
    return sizeof(MergeFastScratch) + (size_t)MERGE_FAST_K * 8 + rerank_lds_bytes(dp, kc) + 16;
}
__host__ __device__ constexpr size_t short_lds_layout(int S, int tiles_per_block, int dp, int kc) {
    const size_t head = (size_t)S * 4 + 16 * ((size_t)S * 4) + 16 * 4;  // mus | qs | xn
    const size_t dump = (size_t)16 * short_dump_stride(tiles_per_block) * 4;
    const size_t tail = short_tail_bytes(dp, kc);
    return head + (dump > tail ? dump : tail);
}

#ifdef ISE_ABLATE
#define SSTAMP(i)                                                                                      \
    do {                                                                                               \
        if (p.stamps && lane == 0)                                                                     \
            p.stamps[((size_t)blockIdx.x * SHORT_W + w) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif

// CH: k-steps per register chunk; R: chunks in the ring (R - 1 requested ahead of the one computed)
template <int CH, int R, bool BF16, bool SHIFT, bool RERANK>
__global__ __launch_bounds__(SHORT_W * 64, 4) void short_scan_kernel(const ScanParams p, const ShortTailParams tp) {
    static_assert(!(BF16 && SHIFT), "the shift is applied to fp32 rows only");
    static_assert(!RERANK || SHIFT, "the re-rank belongs to the float32 L2 search");
    constexpr int W = SHORT_W;
    constexpr int BLOCK_THREADS = W * 64;
    constexpr int NQ = 16;
    constexpr int TPR = BLOCK_THREADS / 16;  // threads staging one query row
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = p.qs_stride;
    float* mus = reinterpret_cast<float*>(smem);   // [S] shift vector (SHIFT only)
    float* qs = mus + S;                           // [NQ][S]
    float* xn = qs + NQ * S;                       // [NQ]
    float* dump = xn + NQ;                         // [NQ][DS] scores of the block's rows; the tail's image later
    unsigned char* tail_mem = reinterpret_cast<unsigned char*>(dump);
    __shared__ unsigned int s_ticket;
    __shared__ int s_ok;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nqt = min(NQ, p.nq);
    const int k = p.k;
    const int nsteps = p.row_slots >> 2;
    // even split of the row tiles over the blocks
    const int t0 = (int)((long long)blockIdx.x * p.tiles_total / gridDim.x);
    const int t1 = (int)((long long)(blockIdx.x + 1) * p.tiles_total / gridDim.x);
    const int DS = short_dump_stride(p.tiles_per_block);
    const bool l2 = p.metric == ISE_METRIC_L2;
    SSTAMP(0);

    auto load_chunk = [&](f32x4(&a)[CH], int tile, int s0) {
        const char* base = static_cast<const char*>(p.xb) +
                           ((((size_t)tile * 16 + c) * p.row_slots + 4 * s0 + g) << 4);
#pragma unroll
        for (int s = 0; s < CH; s++) a[s] = *reinterpret_cast<const f32x4*>(base + 64 * s);
    };
    auto load_norms = [&](int tile) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(p.norms + (size_t)tile * 16 + 4 * g);
    };

    // ---- query staging, step 1: request the query rows (as scan_kernel does: TPR threads per row)
    constexpr int FPS = BF16 ? 2 : 1;
    constexpr int QV = 8;
    constexpr int QVS = QV / FPS;
    const int S4 = S >> 2;
    const int dslots = BF16 ? (p.d >> 3) : (p.d >> 2);
    const bool vec_q = (p.d & (BF16 ? 7 : 3)) == 0 && ((reinterpret_cast<uintptr_t>(p.q) & 15) == 0) &&
                       S4 <= TPR * QVS;
    f32x4 qv[QV];
    {
        const int cc = tid / TPR, t = tid % TPR;
        const bool rowok = cc < nqt;
        const float* src = p.q + (size_t)(rowok ? cc : 0) * p.d;
        if (vec_q) {
#pragma unroll
            for (int i = 0; i < QVS; i++) {
                const int j4 = t + i * TPR;
#pragma unroll
                for (int f = 0; f < FPS; f++) {
                    qv[i * FPS + f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (rowok && j4 < dslots) qv[i * FPS + f] = *reinterpret_cast<const f32x4*>(src + 4 * (j4 * FPS + f));
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the ring: R - 1 chunks requested now, so the row stream runs while the queries are staged
    const bool has_work = (t0 + w) < t1;
    f32x4 A[R][CH];
    int ltile = t0 + w, ls0 = 0;
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

    // ---- query staging, step 2: into LDS (zero padded to NQ x S units) with |x|^2; the shift vector is read
    // slot by slot here (L2-resident), not held in registers beside the ring
    auto to_bf16_pair = [](float lo, float hi) -> uint32_t {
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;
        return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    };
    auto bf16_round = [](float v) -> float { return (float)(__bf16)v; };
    {
        const int cc = tid / TPR, t = tid % TPR;
        float sn = 0.f;
        if (vec_q) {
#pragma unroll
            for (int i = 0; i < QVS; i++) {
                const int j4 = t + i * TPR;
                if (j4 < S4) {
                    if (BF16) {
                        const f32x4 v0 = qv[i * FPS], v1 = qv[i * FPS + FPS - 1];
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
                        f32x4 v = qv[i];
                        if (SHIFT) {
                            f32x4 m = (f32x4){0.f, 0.f, 0.f, 0.f};
                            if (j4 < dslots) m = *reinterpret_cast<const f32x4*>(p.mu + 4 * j4);
                            if (cc < nqt) v = v - m;  // padding rows stay zero
                            if (cc == 0) *reinterpret_cast<f32x4*>(mus + 4 * j4) = m;
                        }
                        *reinterpret_cast<f32x4*>(qs + cc * S + 4 * j4) = v;
                        sn = fmaf(v[0], v[0], sn);
                        sn = fmaf(v[1], v[1], sn);
                        sn = fmaf(v[2], v[2], sn);
                        sn = fmaf(v[3], v[3], sn);
                    }
                }
            }
        } else {  // odd d, unaligned queries or very long rows: one 4-byte unit at a time
            const bool rowok = cc < nqt;
            const float* src = p.q + (size_t)(rowok ? cc : 0) * p.d;
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
    __syncthreads();
    SSTAMP(1);

    // ---- the row tiles of this wave, back to back.  Scores exactly as scan_kernel keys them (ise_scan.hpp).
    const float* qrow = qs + c * S + 4 * g;
    const float xq_n = xn[c];
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto score = [&](float dotj, float ynj) -> float {
        if (l2) {
            const float tt = xq_n + ynj;
            const float sc = tt - 2.f * dotj;
            if (SHIFT) return fmaf(-p.beta, tt, sc);
            return sc < 0.f ? 0.f : sc;  // keeps NaN (Faiss: if (dis < 0) dis = 0)
        }
        return -dotj;
    };
    f32x4 bcur;
    auto load_b = [&](f32x4& b, int step) { b = *reinterpret_cast<const f32x4*>(qrow + 16 * step); };
    auto compute_chunk = [&](const f32x4(&a)[CH], int s0, int next_first_step) {
#pragma unroll
        for (int s = 0; s < CH; s++) {
            f32x4 bnext;
            load_b(bnext, s + 1 < CH ? s0 + s + 1 : next_first_step);
            f32x4 as = a[s];
            if (SHIFT) as = as - *reinterpret_cast<const f32x4*>(mus + 4 * g + 16 * (s0 + s));
            if (BF16) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, a[s]), bv = __builtin_bit_cast(bf16x8, bcur);
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc0, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[0], bcur[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[1], bcur[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[2], bcur[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(as[3], bcur[3], acc1, 0, 0, 0);
            }
            bcur = bnext;
        }
    };
    if (has_work) {
        int tile = t0 + w, s0 = 0;
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
                    if (ns0 == 0) {  // the tile is complete: its 16 x 16 scores go to the dump
                        const f32x4 dot = acc0 + acc1;
                        acc0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                        acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
                        f32x4 sc;
#pragma unroll
                        for (int jj = 0; jj < 4; jj++) sc[jj] = score(dot[jj], yn[jj]);
                        *reinterpret_cast<f32x4*>(dump + (size_t)c * DS + (tile - t0) * 16 + 4 * g) = sc;
                    }
                    done = ntile >= t1;
                    tile = ntile; s0 = ns0;
                }
            }
        }
    }
    SSTAMP(2);
    __syncthreads();
    SSTAMP(3);

    // ---- select: per query the exact sorted top k of the block's rows -> the block's list
    const int nkeys = (t1 - t0) * 16;
    for (int qq = w; qq < NQ; qq += W) {
        u64 kk[SHORT_KPL];
#pragma unroll
        for (int e = 0; e < SHORT_KPL; e++) {
            kk[e] = KEY_PAD;
            const int idx = lane + 64 * e;
            if (idx < nkeys) {
                const float sc = dump[(size_t)qq * DS + idx];
                const long long row = (long long)t0 * 16 + idx;
                const bool ok = row < p.n && sc < FLT_MAX && qq < nqt;
                if (ok) kk[e] = ((u64)ord_f32(sc) << 32) | (uint32_t)((uint32_t)row + p.id_base);
            }
        }
        u64* out = p.part + ((size_t)blockIdx.x * NQ + qq) * k;
        u64 kth_unused;
        const int nw = wave_select<SHORT_KPL>(kk, nkeys, k, out, &kth_unused);
        if (lane >= nw && lane < k) out[lane] = KEY_PAD;  // k <= KB_MAX <= 64
    }
    SSTAMP(4);

    // ---- arrive: this block's lists are visible at agent scope before its ticket is
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's list stores have left
    __syncthreads();                                    // ... and every wave's; the dump is dead from here on
    const unsigned int nb = gridDim.x;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(tp.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tp.arrive_base;
    }
    __syncthreads();
    const unsigned int ticket = s_ticket;
    const unsigned int nwork = (unsigned int)p.nq < nb ? (unsigned int)p.nq : nb;
    if (ticket < nb - nwork) return;
    SSTAMP(5);

    // ---- tail worker: wait for every block's lists, then merge (+ re-rank) the queries worker, worker + nwork, ...
    if (tid == 0) {
        int ok = 1;
        if (ticket != nb - 1) {  // the holder of the last ticket knows that everybody has arrived
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(tp.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tp.arrive_base < nb) {
                __builtin_amdgcn_s_sleep(4);
                if (__builtin_amdgcn_s_memrealtime() - t_start > SHORT_WAIT_TICKS) { ok = 0; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ok = ok;
    }
    __syncthreads();
    SSTAMP(6);
    MergeFastScratch& fast = *reinterpret_cast<MergeFastScratch*>(tail_mem);
    u64* res = reinterpret_cast<u64*>(tail_mem + sizeof(MergeFastScratch));                   // [MERGE_FAST_K]
    unsigned char* rr = tail_mem + ((sizeof(MergeFastScratch) + MERGE_FAST_K * 8 + 15) & ~(size_t)15);  // rerank image
    for (int q = (int)(ticket - (nb - nwork)); q < p.nq; q += (int)nwork) {
        if (!s_ok) {  // gave up waiting (see the header): never a silently wrong answer
            if (tid == 0) {
                if (tp.gave_up) atomicAdd(tp.gave_up, 1ull);
                if (RERANK) fallback_list_push(tp.xp, q);
            }
            if (!RERANK && tid < tp.mp.k) emit_result(tp.mp, (size_t)q * tp.mp.k + tid, KEY_PAD);
            continue;
        }
        const u64* base = tp.mp.lists + (size_t)q * tp.mp.k;
        if (RERANK) {
            u64* kin = reinterpret_cast<u64*>(rr + (size_t)tp.xp.dp * 4);
            rerank_stage_query<BLOCK_THREADS>(tp.xp, q, rr);  // its loads fly while the lists are merged
            merge_waves(tp.mp, base, fast, kin);
            rerank_block<BLOCK_THREADS>(tp.xp, q, rr, nullptr);  // starts with a block barrier
        } else {
            merge_waves(tp.mp, base, fast, res);
            if (tid < tp.mp.k) emit_result(tp.mp, (size_t)q * tp.mp.k + tid, res[tid]);
        }
        __syncthreads();  // the image is reused by the worker's next query
    }
    SSTAMP(7);
}

template <int CH, bool BF16, bool SHIFT, bool RERANK>
static void launch_short_one(int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortTailParams& tp) {
    constexpr int R = CH >= 4 ? 3 : (CH == 2 ? 5 : 8);
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&short_scan_kernel<CH, R, BF16, SHIFT, RERANK>), LDS_LIMIT);
    hipLaunchKernelGGL((short_scan_kernel<CH, R, BF16, SHIFT, RERANK>), dim3(grid), dim3(SHORT_W * 64), lds, st, sp, tp);
}
template <bool BF16, bool SHIFT, bool RERANK>
static void launch_short_v(int ch, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortTailParams& tp) {
    if (ch >= 4) launch_short_one<4, BF16, SHIFT, RERANK>(grid, lds, st, sp, tp);
    else if (ch == 2) launch_short_one<2, BF16, SHIFT, RERANK>(grid, lds, st, sp, tp);
    else launch_short_one<1, BF16, SHIFT, RERANK>(grid, lds, st, sp, tp);
}

// one launch entry point per kernel family, defined in the family's translation unit (ise_scan_*.hip)
void ise_launch_short_f32_shift(int ch, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortTailParams& tp);
void ise_launch_short_f32_plain(int ch, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortTailParams& tp);
void ise_launch_short_bf16(int ch, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortTailParams& tp);
