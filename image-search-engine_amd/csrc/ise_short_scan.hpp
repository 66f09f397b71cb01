// ise_short_scan.hpp -- the scan of one batch of <= 64 queries (one or two 16-query tiles per pass) against a SHORT index.
//
// The reference's own regime is short indexes (about 1 k images, backend/utils.py:309-310; one query per
// request, backend/engine.py:50-55); BASELINE config 2 is 100k x 512 and every rank of the 8-GPU run scans a
// 125k-row shard.  There a wave of scan_kernel (ise_scan.hpp) owns one or two row tiles, and its top-k
// bookkeeping IS the kernel: the boot (dump, two barriers, a windowed cut -- nothing in flight meanwhile) sits
// in the middle of the stream and the final selection behind it: 45.8 us at 100k x 512, where a kernel that
// only reads the rows in the same shape takes 30 (scripts/microbench/tile_read_bw.hip).  This kernel keeps
// the row stream, the MFMA tile and the arithmetic of scan_kernel -- the per-block lists it writes hold the
// same keys, bit for bit, and go to the same merge_kernel -- and replaces the bookkeeping:
//
//   stream   a wave runs its row tiles back to back; a tile's 16 x 16 scores go to LDS as plain floats
//            (one 16-byte store per lane and tile: the row id is the slot index).  No threshold, no lists,
//            no barrier until the block's rows are done; R - 1 chunks are requested BEFORE the queries are
//            staged, so the stream also covers the staging.
//   select   one barrier; per query one wave finds a threshold on the 32-bit scores (the k-th smallest of the
//            64 lane minima bounds the k-th smallest score), turns the rows at or below it -- about k + a few
//            -- into 64-bit keys and ranks those: the exact sorted top k -> the block's list in `part`.
//
// Tried and NOT adopted (profiles/r03/README.md): the merge + exact re-rank folded into this launch (every
// block takes a ticket behind a release; the holders of the last nq tickets wait for the others, acquire, and
// each run merge_kernel's body for one query).  Correct (every parity test passed with it), but the tail cost
// 11 us inside the kernel against 10.5 us + a 1.5 us launch boundary as its own launch, the arrive sequence
// another 3 us for every block, and the launch held one of the four hardware queues for its whole 62 us while
// only 16 blocks worked: 43.8 us per step against 39.0 for scan + merge launches.
#pragma once
#include "ise_common.hpp"
#include "ise_scan_params.hpp"
#include "ise_select.hpp"

#define SHORT_W_MAX 16     /* waves per block: 16 (one block per CU) or 8 (two) */
#define SHORT_KPL 8        /* keys per lane of the block selection: <= 512 rows (32 tiles) per block */
#define SHORT_TPB_MAX (SHORT_KPL * 64 / 16)

struct ShortParams {
    int even_split;  // 1 = the row tiles are split evenly over the blocks instead of tiles_per_block each
    int T = 1;       // query tiles per pass (1 or 2)
    int nqt = 1;     // passes side by side (grid.y): ceil(nq / (16 T))
};

// dump row stride in floats: whole 32-float groups + 4, so that the 8 lanes of a 16-byte store group
// (8 queries, same rows) fall into different banks
__host__ __device__ constexpr int short_dump_stride(int tiles_per_block) { return (tiles_per_block * 16 + 31) / 32 * 32 + 4; }
__host__ __device__ constexpr size_t short_lds_layout(int S, int tiles_per_block, int waves, int T = 1) {
    const size_t nq = (size_t)16 * T;
    const size_t head = (size_t)S * 4 + nq * ((size_t)S * 4) + nq * 4;  // mus | qs | xn
    const size_t dump = nq * short_dump_stride(tiles_per_block) * 4 + (size_t)waves * 64 * 8;  // + selection scratch
    return head + dump;
}

#ifdef ISE_ABLATE
#define SABL(bit) (p.ablate & (bit))
#else
#define SABL(bit) 0
#endif
#ifdef ISE_ABLATE
#define SSTAMP(i)                                                                                      \
    do {                                                                                               \
        if (p.stamps && lane == 0)                                                                     \
            p.stamps[((size_t)blockIdx.x * W + w) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif

// CH: k-steps per register chunk; R: chunks in the ring (R - 1 requested ahead of the one computed)
// WPS: waves per SIMD the register budget is sized for (blocks per CU * W / 4)
// T: 16-query tiles scored per pass over the rows (1: batches of up to 16 queries; 2: of 17 .. 32, and 33 .. 64 as
//    two such passes side by side in grid.y) -- the same MFMA tile per query tile, the same bits
template <int CH, int R, int W, int WPS, int T, bool BF16, bool SHIFT>
__global__ __launch_bounds__(W * 64, WPS) void short_scan_kernel(const ScanParams p, const ShortParams tp) {
    static_assert(!(BF16 && SHIFT), "the shift is applied to fp32 rows only");
    constexpr int NQ = 16 * T;
    // 32 threads stage one query row, whatever the block width (a 16-wave block stages with its first 8 waves):
    // the sum of squares |x|^2 then has ONE summation order -- the one of scan_kernel's 8-wave blocks -- and
    // the bf16 L2 distances, which carry it, come out the same bits from every kernel
    constexpr int TPR = 32;
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = p.qs_stride;
    float* mus = reinterpret_cast<float*>(smem);   // [S] shift vector (SHIFT only)
    float* qs = mus + S;                           // [NQ][S]
    float* xn = qs + NQ * S;                       // [NQ]
    float* dump = xn + NQ;                         // [NQ][DS] scores of the block's rows

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as a scalar: the per-query loops below then branch on SGPRs (hipcc otherwise treats
    // tid >> 6 as divergent and compiles them with exec masks and spilled scalar registers)
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool stager = tid < 16 * TPR;
    const int c = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.y * NQ;  // this block's first query
    const int nqt = min(NQ, p.nq - q0);
    const int k = p.k;
    const int nsteps = p.row_slots >> 2;
    // a block owns tiles_per_block = 8 r consecutive row tiles, its waves interleave: r tiles per wave, every
    // wave of the grid (but the last block's) the same number -- a wave that owns one tile more than the others
    // finishes a whole latency-bound tile time after them
    const int t0 = tp.even_split ? (int)((long long)blockIdx.x * p.tiles_total / gridDim.x) : blockIdx.x * p.tiles_per_block;
    const int t1 = tp.even_split ? (int)((long long)(blockIdx.x + 1) * p.tiles_total / gridDim.x)
                                 : min(t0 + p.tiles_per_block, p.tiles_total);
    // (even: every block -- every CU -- the same bytes, within a tile; the stream is bound per CU)
    const int DS = short_dump_stride(p.tiles_per_block);
    const bool l2 = p.metric == ISE_METRIC_L2;
    SSTAMP(0);

    auto load_chunk = [&](f32x4(&a)[CH], int tile, int s0) {
        const char* base = static_cast<const char*>(p.xb) +
                           ((((size_t)tile * 16 + c) * p.row_slots + 4 * s0 + g) << 4);
        if (SABL(128)) return;  // dev: no index loads
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
    f32x4 mu1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the shift vector goes through LDS (one slot per thread, S4 <= 256 here) and the queries read it from
    // there: held in registers per thread (8 slots) it would cost the ring a chunk.  Requested first: waits
    // are counted in issue order
    if (stager && SHIFT && vec_q && tid < dslots) mu1 = *reinterpret_cast<const f32x4*>(p.mu + 4 * tid);
    auto request_queries = [&](int tq) {  // step 1 for query tile tq: TPR threads per row, QV 16-byte pieces each
        if (stager && vec_q) {
            const int cc = tq * 16 + tid / TPR, t = tid % TPR;
            const bool rowok = cc < nqt;
            const float* src = p.q + (size_t)(q0 + (rowok ? cc : 0)) * p.d;
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
    };
    request_queries(0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- the ring: R - 1 chunks requested now, so the row stream runs while the queries are staged
    const bool has_work = (t0 + w) < t1 && !SABL(16);
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

    // ---- query staging, step 2: into LDS (zero padded to NQ x S units) with |x|^2
    if (SHIFT && vec_q) {
        if (tid < S4) *reinterpret_cast<f32x4*>(mus + 4 * tid) = mu1;
        __syncthreads();
    }
    auto to_bf16_pair = [](float lo, float hi) -> uint32_t {
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;
        return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
    };
    auto bf16_round = [](float v) -> float { return (float)(__bf16)v; };
    auto store_queries = [&](int tq) {
      if (stager) {
        const int cc = tq * 16 + tid / TPR, t = tid % TPR;
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
                        if (SHIFT && cc < nqt) v = v - *reinterpret_cast<const f32x4*>(mus + 4 * j4);  // padding rows stay zero
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
    };
    store_queries(0);
#pragma unroll
    for (int tq = 1; tq < T; tq++) {  // further query tiles: behind the first, the row stream already running
        request_queries(tq);
        store_queries(tq);
    }
    __syncthreads();
    SSTAMP(1);

    // ---- the row tiles of this wave, back to back.  Scores exactly as scan_kernel keys them (ise_scan.hpp).
    const float* qrow = qs + c * S + 4 * g;  // query tile t: + t * 16 * S
    float xq_n[T];
    f32x4 acc0[T], acc1[T];
#pragma unroll
    for (int t = 0; t < T; t++) {
        xq_n[t] = xn[t * 16 + c];
        acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto score = [&](float dotj, float ynj, float xqn) -> float {
        if (l2) {
            const float tt = xqn + ynj;
            const float sc = tt - 2.f * dotj;
            if (SHIFT) return fmaf(-p.beta, tt, sc);
            return sc < 0.f ? 0.f : sc;  // keeps NaN (Faiss: if (dis < 0) dis = 0)
        }
        return -dotj;
    };
    f32x4 bcur[T];
    auto load_b = [&](f32x4(&b)[T], int step) {
#pragma unroll
        for (int t = 0; t < T; t++) b[t] = *reinterpret_cast<const f32x4*>(qrow + (size_t)t * 16 * S + 16 * step);
    };
    auto compute_chunk = [&](const f32x4(&a)[CH], int s0, int next_first_step) {
#pragma unroll
        for (int s = 0; s < CH; s++) {
            if (SABL(64)) {  // dev: no LDS reads, no MFMA -- the loaded data is only summed
                acc0[0] += a[s];
                continue;
            }
            f32x4 bnext[T];
            load_b(bnext, s + 1 < CH ? s0 + s + 1 : next_first_step);
            f32x4 as = a[s];
            if (SHIFT) as = as - *reinterpret_cast<const f32x4*>(mus + 4 * g + 16 * (s0 + s));
#pragma unroll
            for (int t = 0; t < T; t++) {
                if (BF16) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, a[s]), bv = __builtin_bit_cast(bf16x8, bcur[t]);
                    if (s & 1) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc1[t], 0, 0, 0);
                    else acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc0[t], 0, 0, 0);
                } else {
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[0], bcur[t][0], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[1], bcur[t][1], acc1[t], 0, 0, 0);
                    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[2], bcur[t][2], acc0[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[3], bcur[t][3], acc1[t], 0, 0, 0);
                }
                bcur[t] = bnext[t];
            }
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
                    if (ns0 == 0) {  // the tile is complete: its 16 x 16 scores per query tile go to the dump
                        const long long row0 = (long long)tile * 16 + 4 * g;
#pragma unroll
                        for (int t = 0; t < T; t++) {
                            const f32x4 dot = acc0[t] + acc1[t];
                            acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                            // rows past the end of the index and scores that may not enter (>= FLT_MAX, NaN: Faiss's
                            // strict gate) are dumped as +inf: the selection then compares plain floats
                            f32x4 sc;
#pragma unroll
                            for (int jj = 0; jj < 4; jj++) {
                                const float v = score(dot[jj], yn[jj], xq_n[t]);
                                sc[jj] = (row0 + jj < p.n && v < FLT_MAX) ? v : __builtin_inff();
                            }
                            *reinterpret_cast<f32x4*>(dump + (size_t)(t * 16 + c) * DS + (tile - t0) * 16 + 4 * g) = sc;
                        }
                        if (tile == t0 + w) SSTAMP(8);
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

    // ---- select: per query the exact sorted top k of the block's rows -> the block's list.
    // All 16 waves of a CU reach this point together and share the SIMDs' issue slots, so what counts is the
    // instruction count (a 64-bit quickselect over all rows: 6 us per query; a 64-step rank of the lane minima
    // in front of wave_select: 2.7 us).  Threshold first, on plain floats: a lane holds up to SHORT_KPL scores
    // of the query; the k-th smallest of the 64 LANE MINIMA (64 distinct rows; found by a ballot quickselect)
    // bounds the k-th smallest score from above, so only the rows at or below it -- about k + a few -- become
    // 64-bit keys (ord(score) << 32 | row id) and are ranked, all pairs, one key per lane.  More than 64
    // survivors (hundreds of equal rows): the general selection.
    const int nkeys = (t1 - t0) * 16;
    const int ne = (nkeys + 63) >> 6;
    u64* sel = reinterpret_cast<u64*>(dump + (size_t)NQ * DS) + w * 64;  // [W][64] compaction scratch behind the dump
    const u64 lt_mask = (1ull << lane) - 1ull;
    const float INF = __builtin_inff();
    for (int qq = w; qq < NQ && !SABL(4); qq += W) {
        u64* out = p.part + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NQ + qq) * k;
        if (qq >= nqt) {  // a padding query of the tile: an empty list (the merge never reads it)
            if (lane < k) out[lane] = KEY_PAD;
            continue;
        }
        float v[SHORT_KPL];
        float mn = INF;
#pragma unroll
        for (int e = 0; e < SHORT_KPL; e++) {
            v[e] = INF;
            if (e < ne) {
                const int idx = lane + 64 * e;
                if (idx < nkeys) v[e] = dump[(size_t)qq * DS + idx];
                mn = fminf(mn, v[e]);
            }
        }
        if (qq == w) SSTAMP(9);
        // TH = the k-th smallest lane minimum (by value); fewer than k lanes with a row: every row stays
        float TH = FLT_MAX;
        if (__popcll(__ballot(mn < INF)) >= k) {
            float L = -INF, H = INF;  // the target lies in [L, H]; lanes strictly inside are pivot candidates
            for (int round = 0;; round++) {
                const u64 m = __ballot(mn > L && mn < H);
                if (!m) {  // nothing strictly inside: the target is one of the bounds
                    TH = __popcll(__ballot(mn <= L)) >= k ? L : H;
                    break;
                }
                const int rot = (round * 23 + 7) & 63;
                const u64 hi = (m >> rot) << rot;
                const float P = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mn),
                                                                                    __ffsll((long long)(hi ? hi : m)) - 1));
                const int c_lt = __popcll(__ballot(mn < P)), c_le = __popcll(__ballot(mn <= P));
                if (c_lt < k && c_le >= k) {
                    TH = P;
                    break;
                }
                if (c_lt >= k) H = P;
                else L = P;
            }
        }
        if (qq == w) SSTAMP(10);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < SHORT_KPL; e++) {
            if (e < ne) {
                const bool keep = v[e] <= TH;  // +inf (no row) never: TH <= FLT_MAX
                const u64 m = __ballot(keep);
                const int pos = cnt + __popcll(m & lt_mask);
                if (keep && pos < 64)
                    sel[pos] = ((u64)ord_f32(v[e]) << 32) | (uint32_t)((uint32_t)(t0 * 16 + lane + 64 * e) + p.id_base);
                cnt += __popcll(m);
            }
        }
        if (qq == w) SSTAMP(11);
        int nw;
        if (cnt <= 64) {  // rank the survivors: lane i holds key i
            wave_lds_fence();
            const u64 mine = lane < cnt ? sel[lane] : KEY_PAD;
            wave_lds_fence();
            int rk = 0;
            for (int j = 0; j < cnt; j++) rk += readlane_u64(mine, j) < mine ? 1 : 0;
            if (lane < cnt && rk < k) out[rk] = mine;
            nw = cnt < k ? cnt : k;
        } else {
            u64 kk[SHORT_KPL];
#pragma unroll
            for (int e = 0; e < SHORT_KPL; e++)
                kk[e] = v[e] == INF ? KEY_PAD
                                    : (((u64)ord_f32(v[e]) << 32) | (uint32_t)((uint32_t)(t0 * 16 + lane + 64 * e) + p.id_base));
            u64 kth_unused;
            nw = wave_select<SHORT_KPL>(kk, nkeys, k, out, &kth_unused);
        }
        if (lane >= nw && lane < k) out[lane] = KEY_PAD;  // k <= KB_MAX <= 64
        if (qq == w) SSTAMP(12);
    }
    // the ring's last R - 1 requests (issued unconditionally, past the wave's rows: a re-read of its last chunk)
    // may still be in flight: their registers stay reserved up to here, so that the selection above does not
    // begin by waiting for them
#pragma unroll
    for (int j = 0; j < R; j++)
#pragma unroll
        for (int s = 0; s < CH; s++) asm volatile("" ::"v"(A[j][s]));
    SSTAMP(4);
}

template <int CH, int R, int W, int WPS, bool BF16, bool SHIFT>
static void launch_short_r(int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp) {
    if (tp.T == 2) {  // 17 .. 64 queries: two query tiles per pass, grid.y passes side by side
        static LdsAttrOnce attr2;
        attr2.ensure(reinterpret_cast<const void*>(&short_scan_kernel<CH, R, W, WPS, 2, BF16, SHIFT>), LDS_LIMIT);
        hipLaunchKernelGGL((short_scan_kernel<CH, R, W, WPS, 2, BF16, SHIFT>), dim3(grid, tp.nqt), dim3(W * 64), lds, st, sp, tp);
        return;
    }
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&short_scan_kernel<CH, R, W, WPS, 1, BF16, SHIFT>), LDS_LIMIT);
    hipLaunchKernelGGL((short_scan_kernel<CH, R, W, WPS, 1, BF16, SHIFT>), dim3(grid), dim3(W * 64), lds, st, sp, tp);
}
template <int CH, int W, int WPS, bool BF16, bool SHIFT>
static void launch_short_one(int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp) {
    // chunks in the ring: ONE chunk (4 KB of a wave's rows at CH = 4) requested ahead of the chunk computed.
    // Measured at 100k x 512 (profiles/r03/README.md): a deeper ring is slower here (43.9 us with one chunk
    // ahead, 45.6 with two, per batch of 16; 36.3 against 41.4 us per step with 16 batches in flight): sixteen
    // waves per CU keep 64 KB in flight as it is, and more requests only queue in the CU's memory pipeline
    constexpr int R = CH >= 4 ? 2 : (CH == 2 ? 3 : 5);
#ifdef ISE_ABLATE
    if (SHIFT) {  // dev: ring depth A/B
        const char* e = getenv("ISE_SHORT_RING");
        const int r = e ? atoi(e) : 0;
        if (r == 2) return launch_short_r<CH, 2, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
        if (r == 3) return launch_short_r<CH, 3, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
        if (r == 4) return launch_short_r<CH, 4, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
        if (r == 6) return launch_short_r<CH, 6, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
    }
#endif
    launch_short_r<CH, R, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
}
template <int W, int WPS, bool BF16, bool SHIFT>
static void launch_short_ch(int ch, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp) {
#ifdef ISE_ABLATE
    if (const char* e = getenv("ISE_SHORT_CH")) {  // dev: smaller chunks (must divide the row's k-steps)
        const int c = atoi(e);
        if ((c == 1 || c == 2) && c <= ch) ch = c;
    }
#endif
    if (ch >= 4) launch_short_one<4, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
    else if (ch == 2) launch_short_one<2, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
    else launch_short_one<1, W, WPS, BF16, SHIFT>(grid, lds, st, sp, tp);
}
// block shapes: (waves, blocks per CU) = (16, 1), (8, 2), (8, 1)
template <bool BF16, bool SHIFT>
static void launch_short_v(int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st, const ScanParams& sp,
                           const ShortParams& tp) {
    (void)bpc;
    if (waves == 16) launch_short_ch<16, 4, BF16, SHIFT>(ch, grid, lds, st, sp, tp);
    else launch_short_ch<8, 4, BF16, SHIFT>(ch, grid, lds, st, sp, tp);
}

// one launch entry point per kernel family, defined in the family's translation unit (ise_scan_*.hip)
void ise_launch_short_f32_shift(int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp);
void ise_launch_short_f32_plain(int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp);
void ise_launch_short_bf16(int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp);
