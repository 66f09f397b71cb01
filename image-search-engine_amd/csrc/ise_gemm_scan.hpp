// ise_gemm_scan.hpp -- large query batches against float32 L2 rows: the filter as ONE GEMM-shaped pass.
//
// What Faiss does for nq >= 20 (IndexFlatL2::search -> blocked SGEMM of 4096 queries x 1024 rows with
// |x|^2 + |y|^2 - 2 x.y [upstream-faiss]; reached from backend/engine.py:55 only through batched
// callers) is a dense contraction: at nq = 1024 the streaming kernel (ise_scan.hpp) re-reads the
// 2 GB index 22 times and sits at ~62 % of the fp32 MFMA peak, limited by its per-tile top-k
// bookkeeping and by two waves per SIMD having to overlap a 344 us stream with a 417 us MFMA
// phase.  Here the roles flip, as in assign_kernel:
//
//   index rows  HBM -> VGPR once: a wave holds one 16-row tile (16 x d floats, shifted by mu) for a
//               whole sweep over the queries.
//   queries     pre-shifted and padded once (qprep_kernel), then streamed L2 -> LDS in stages of
//               GQ = 32 by LDS-DMA (global_load_lds_dwordx4, no registers), two buffers, one
//               barrier per stage; every wave reads the stage as its MFMA B operand (the
//               conflict-free ds_read_b128 pattern of the scan kernel).
//   top-k       none in this kernel.  A strided SAMPLE of the index (128 slabs of 128 rows spread over
//               the index) goes through the same kernel first in DUMP mode -- every lower-bound
//               score is written out, 64 MB at nq = 1024 -- and kth_select_kernel gives each query
//               an admit threshold tau = its kc-th smallest score in the sample (>= the kc-th
//               smallest overall).  The GEMM pass proper appends every (row, query) with lo <= tau
//               to the query's candidate array in HBM (one atomic per candidate; ~N kc / 16384 of
//               them per query).
//   then        gemm_select_kernel sorts a query's candidates, keeps the kc best and hands them to the
//               exact re-rank (ise_exact.hpp).  Its certificate also covers the rows the threshold
//               cut: they all have lo > tau, so min(lo_(kc), tau) > d_(k) proves the result; a
//               query whose candidate array overflowed, or that fails, goes to the exact scan.
//
// MFMA-bound: 2 nq N d flops per launch against 4 N d bytes of HBM traffic.
#pragma once
#include "ise_common.hpp"
#include "ise_exact.hpp"
#include "ise_scan_params.hpp"
#include "ise_select.hpp"

#define GQ 32          /* queries per LDS stage */
#define GEMM_NQ_MAX 1024 /* queries per launch (thresholds and norms live in LDS) */

// End of a query stage: the LDS-DMA requests of the NEXT stage were issued at the start of this one, the
// candidate appends (plain stores to HBM, acknowledged a microsecond later) behind them.  Vector-memory
// operations retire in issue order, so waiting until at most `younger` of them are outstanding waits for the
// DMA and for nothing younger: a plain vmcnt(0) here made every wave sit out its last store's round trip at
// every barrier (SQ_WAIT_ANY 41 % of the wave cycles in the bf16 kernel, MFMA pipe 44 % busy).
// `younger` = stores this wave issued since the stage's DMA (wave-uniform; under-counting is safe).
__device__ __forceinline__ void wait_stage_dma(unsigned younger) {
    if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (younger < 8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct GemmScanParams {
    const float* xb;      // [cap][dp]
    const float* norms;   // [cap] |y - mu|^2
    const float* mu;      // [dp]
    long long n;
    long long rows16;     // n rounded up to whole 16-row tiles (readable; pad rows are zero)
    int dp, S;            // S: row stride of qprep / the LDS stage in floats ((S / 4) mod 16 == 2)
    const float* qprep;   // [nq_pad][S] x - mu, zero padded
    const float* xn;      // [nq_pad] |x - mu|^2
    const float* tau;     // [nq_pad] admit score per query (+FLT_MAX admits everything finite)
    int nq, nq_pad;       // nq_pad: multiple of GQ, <= GEMM_NQ_MAX
    float beta;
    int metric;           // bf16 rows only (ise_gemm_bf16.hpp); the float32 kernel is L2
    uint32_t id_base;
    // candidates leave the GEMM pass without atomics: every wave appends (key, query) entries to a
    // buffer of its own; regroup_kernel sorts them out per query afterwards
    u32x4* wbuf;          // [waves][capw] entries {key lo, key hi, query, 0}
    unsigned int* wcnt;   // [waves] entries written (may exceed capw: overflow)
    int capw;
    int slabs;            // 128-row slabs this launch covers; slab i starts at row i * slab_stride * 128
    int slab_stride;
    int qparts;           // the query stages are split over this many blocks per slab (DUMP launches: 1 slab per block)
    float* dump;          // DUMP: [nq_pad][slabs * 128] lower-bound scores (FLT_MAX for rows past the end)
    int ablate;           // dev builds: 1 = no epilogue, 2 = no query staging, 4 = no MFMA ($ISE_GEMM_ABLATE)
};

__host__ __device__ constexpr size_t gemm_lds_bytes(int S) {
    return (size_t)2 * GQ * S * 4 + (size_t)2 * GEMM_NQ_MAX * 4 + (size_t)S * 4;
}

// queries -> shifted, padded rows + their norms (wave per query, fixed summation order)
__global__ __launch_bounds__(256) void qprep_kernel(const float* __restrict__ q, int nq, int nq_pad, int d, int S,
                                                    const float* __restrict__ mu, float* __restrict__ qprep,
                                                    float* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq_pad) return;
    float s = 0.f;
    for (int j = lane; j < S; j += 64) {
        float v = 0.f;
        if (i < nq && j < d) v = q[(size_t)i * d + j] - (mu ? mu[j] : 0.f);  // mu = null: inner product, no shift
        qprep[(size_t)i * S + j] = v;
        s = fmaf(v, v, s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) xn[i] = s;
}

// kc-th smallest of each query's `rows` dumped scores (one block per query; the scores are read ONCE, into
// registers, when rows <= 16384):
//   1. U = the kc-th smallest of the 256 per-thread minima: at least kc values lie at or below it;
//   2. the values <= U (about kc * rows / 256 of them... at most KTH_LIST) are collected in LDS (one atomic per
//      wave and step); block_topk_u64 finds the kc-th of them.  A list that overflows (heavily tied scores) keeps U.
#define KTH_LIST 2048
#define KTH_REG 64 /* scores per thread held in registers */
__global__ __launch_bounds__(256) void kth_select_kernel(const float* __restrict__ dump, int rows, int kc, int nq,
                                                         float* __restrict__ tau) {
    __shared__ u64 mins[256];
    __shared__ u64 lst[KTH_LIST];
    __shared__ u64 stage[4 * 64];
    __shared__ u64 topk[64];
    __shared__ int cnt;
    __shared__ u64 ubound;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (q >= nq) {  // padding queries admit nothing
        if (tid == 0) tau[q] = -FLT_MAX;
        return;
    }
    const float* src = dump + (size_t)q * rows;
    const bool in_regs = rows <= KTH_REG * 256;  // block-uniform
    // keys: order-preserving score image in the high word, position in the low word (unique)
    float v[KTH_REG];
    u64 mn = KEY_PAD;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < KTH_REG; j++) {
            const int i = tid + 256 * j;
            v[j] = i < rows ? src[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < KTH_REG; j++) {
            const int i = tid + 256 * j;
            if (i < rows) mn = min_u64(mn, ((u64)ord_f32(v[j]) << 32) | (uint32_t)i);
        }
    } else {
        for (int i = tid; i < rows; i += 256) mn = min_u64(mn, ((u64)ord_f32(src[i]) << 32) | (uint32_t)i);
    }
    mins[tid] = mn;
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (w == 0) {
        u64 kk[4];
#pragma unroll
        for (int e = 0; e < 4; e++) kk[e] = mins[lane + 64 * e];
        u64 kth = KEY_PAD;
        const int nw = wave_select<4>(kk, 256, kc, topk, &kth);
        if (lane == 0) ubound = nw == kc ? kth : KEY_PAD - 1;
    }
    __syncthreads();
    const u64 U = ubound;
    const u64 lt_mask = (1ull << lane) - 1ull;
    auto collect = [&](u64 key, bool valid) {  // wave-uniform call: one LDS atomic per wave and step
        const bool hit = valid && key <= U;
        const u64 m = __ballot(hit);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&cnt, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int at = base + __popcll(m & lt_mask);
            if (hit && at < KTH_LIST) lst[at] = key;
        }
    };
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < KTH_REG; j++) {
            const int i = tid + 256 * j;
            collect(((u64)ord_f32(v[j]) << 32) | (uint32_t)i, i < rows);
        }
    } else {
        for (int i0 = 0; i0 < rows; i0 += 256) {
            const int i = i0 + tid;
            collect(i < rows ? ((u64)ord_f32(src[i]) << 32) | (uint32_t)i : KEY_PAD, i < rows);
        }
    }
    __syncthreads();
    const int n = cnt;
    if (n > KTH_LIST) {  // block-uniform: too many values share the bound
        if (tid == 0) tau[q] = unord_f32((uint32_t)(U >> 32));
        return;
    }
    block_topk_u64(lst, n, kc, stage, topk);  // n >= kc: the kc minima below U are in the list
    if (tid == 0) tau[q] = unord_f32((uint32_t)(topk[kc - 1] >> 32));
}

// IPM: float32 INNER PRODUCT rows (the reference's default "cosine" index, backend/utils.py:293,300-303): no
// shift, no norms, the score is -x.y summed exactly as scan_kernel sums it (elements 0, 2 of every k-step in
// one accumulator chain, 1, 3 in the other, then their sum), so that the large-batch path returns the bits of
// the streaming passes; no re-rank behind it (gemm_select_plain_kernel emits the k best candidates).
template <int NS, bool DUMP, bool IPM = false>
__global__ __launch_bounds__(512, 2) void gemm_scan_kernel(const GemmScanParams p) {
    extern __shared__ __align__(16) unsigned char smem_g[];
    const int S = p.S;
    float* qbuf0 = reinterpret_cast<float*>(smem_g);  // [GQ][S]
    float* qbuf1 = qbuf0 + (size_t)GQ * S;
    float* tauL = qbuf1 + (size_t)GQ * S;             // [GEMM_NQ_MAX]
    float* xnL = tauL + GEMM_NQ_MAX;                  // [GEMM_NQ_MAX]
    float* muL = xnL + GEMM_NQ_MAX;                   // [S]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const int c = lane & 15, g = lane >> 4;
    const int nstages = p.nq_pad / GQ;

    // one stage = GQ * S floats, contiguous in qprep and in LDS: 1 KB per wave instruction
    auto stage_load = [&](int st, float* dst) {
        const char* src = reinterpret_cast<const char*>(p.qprep + (size_t)st * GQ * S);
        const int bytes = GQ * S * 4;
        for (int off = w * 1024; off < bytes; off += 8 * 1024)
            if (off + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off + lane * 16),
                                                 (lds_ptr_t)(reinterpret_cast<char*>(dst) + off), 16, 0, 0);
    };

    // this block's share: slabs first, first + step, ... and the query stages [st0, st1)
    const int part = blockIdx.x % p.qparts;
    const int first = blockIdx.x / p.qparts, step = gridDim.x / p.qparts;
    const int st0 = part * nstages / p.qparts, st1 = (part + 1) * nstages / p.qparts;
    for (int i = tid; i < p.nq_pad; i += 512) {
        // thresholds strictly below FLT_MAX: "score <= tau" then also says "score < FLT_MAX" (and NaN fails it)
        tauL[i] = DUMP ? 0.f : fminf(p.tau[i], 3.4028233e38f);
        xnL[i] = p.xn[i];
    }
    for (int i = tid; i < S; i += 512) muL[i] = (!IPM && i < p.dp) ? p.mu[i] : 0.f;
    stage_load(st0, qbuf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    unsigned int wfill = 0;  // wave-uniform: entries this wave has appended
    u32x4* const wmine = DUMP ? nullptr : p.wbuf + (size_t)(blockIdx.x * 8 + w) * p.capw;
    const u64 lt_mask = (1ull << lane) - 1ull;
    for (int slab = first; slab < p.slabs; slab += step) {
        // ---- this wave's 16-row tile: HBM -> registers, shifted by mu (rows past the end read the last tile's pad)
        const long long row_base = (long long)slab * p.slab_stride * 128 + w * 16;
        const long long rr = min(row_base + c, p.rows16 - 1);
        f32x4 a[NS];
        {
            const float* rp = p.xb + (size_t)rr * p.dp + 4 * g;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                a[s] = *reinterpret_cast<const f32x4*>(rp + 16 * s);
                if (!IPM) a[s] = a[s] - *reinterpret_cast<const f32x4*>(muL + 16 * s + 4 * g);
            }
        }
        f32x4 yn = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!IPM && row_base + 4 * g + 3 < p.rows16) yn = *reinterpret_cast<const f32x4*>(p.norms + row_base + 4 * g);
        const long long left = p.n - (row_base + 4 * g);
        const int nv = left >= 4 ? 4 : (left > 0 ? (int)left : 0);  // valid rows among this lane's four: checked once per slab

        for (int sq = st0; sq < st1; sq++) {
            float* cur = buf ? qbuf1 : qbuf0;
            float* nxt = buf ? qbuf0 : qbuf1;
            stage_load(sq + 1 == st1 ? st0 : sq + 1, nxt);  // cyclic: the next slab starts at the first stage again
            unsigned younger = 0;  // stores issued behind that DMA

            const float* q0 = cur + (size_t)c * S + 4 * g;
            const float* q1 = q0 + (size_t)16 * S;
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 odd0 = (f32x4){0.f, 0.f, 0.f, 0.f}, odd1 = (f32x4){0.f, 0.f, 0.f, 0.f};  // IPM: the chains of elements 1, 3
            f32x4 b0 = *reinterpret_cast<const f32x4*>(q0), b1 = *reinterpret_cast<const f32x4*>(q1);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                f32x4 nb0 = b0, nb1 = b1;
                if (s + 1 < NS) {
                    nb0 = *reinterpret_cast<const f32x4*>(q0 + 16 * (s + 1));
                    nb1 = *reinterpret_cast<const f32x4*>(q1 + 16 * (s + 1));
                }
                // two independent accumulator chains, interleaved: the 40-cycle dependent latency never shows
                if constexpr (IPM) {  // scan_kernel's summation order: four chains, none waits for another
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b0[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b1[0], acc1, 0, 0, 0);
                    odd0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b0[1], odd0, 0, 0, 0);
                    odd1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b1[1], odd1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][2], b0[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][2], b1[2], acc1, 0, 0, 0);
                    odd0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][3], b0[3], odd0, 0, 0, 0);
                    odd1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][3], b1[3], odd1, 0, 0, 0);
                } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b0[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][0], b1[0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b0[1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][1], b1[1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][2], b0[2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][2], b1[2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][3], b0[3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][3], b1[3], acc1, 0, 0, 0);
                }
                b0 = nb0;
                b1 = nb1;
                __builtin_amdgcn_sched_barrier(0);  // B fragments are requested one k-step ahead, not all up front (registers)
            }
            // ---- epilogue: lane (c, g) holds rows 4g..4g+3 of its tile against query c of either query tile
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f32x4 dot = IPM ? (t ? acc1 + odd1 : acc0 + odd0) : (t ? acc1 : acc0);
                const int q = sq * GQ + t * 16 + c;
                const float xq_n = xnL[q], tq = tauL[q];
                float lo[4];
                bool pass[4];
                bool any = false;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float tt = xq_n + yn[j];
                    lo[j] = IPM ? -dot[j] : fmaf(-p.beta, tt, tt - 2.f * dot[j]);
                    pass[j] = lo[j] <= tq && j < nv;
                    any |= pass[j];
                }
                if constexpr (DUMP) {  // the threshold sample: every score goes out (rows past the end and NaN as FLT_MAX)
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        o[j] = (j < nv && lo[j] < FLT_MAX) ? lo[j] : FLT_MAX;
                    *reinterpret_cast<f32x4*>(p.dump + (size_t)q * ((size_t)p.slabs * 128) + (size_t)slab * 128 + w * 16 + 4 * g) = o;
                    younger++;
                } else if (__ballot(any)) {  // rare: ~N kc / sample candidates per query over the whole index
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const u64 m = __ballot(pass[j]);
                        if (m) {  // wave-uniform; plain stores, nothing to wait for
                            const unsigned at = wfill + (unsigned)__popcll(m & lt_mask);
                            if (pass[j] && at < (unsigned)p.capw) {
                                u32x4 e;
                                e[0] = (uint32_t)((uint32_t)(row_base + 4 * g + j) + p.id_base);
                                e[1] = ord_f32(lo[j]);
                                e[2] = (uint32_t)q;
                                e[3] = 0u;
                                wmine[at] = e;
                            }
                            if (wfill < (unsigned)p.capw) younger++;  // the lowest passing lane stored: the instruction was issued
                            wfill += (unsigned)__popcll(m);
                        }
                    }
                }
            }
            wait_stage_dma(younger);  // the next stage has landed (this wave's appends may still be on their way)
            __syncthreads();
            buf ^= 1;
        }
    }
    if (!DUMP && lane == 0) p.wcnt[blockIdx.x * 8 + w] = wfill;
}

// per-wave candidate entries -> per-query candidate arrays (throughput kernel: the atomics' latency
// hides behind thousands of threads).  A wave buffer that overflowed poisons every query: exact scan.
// A query's array is GEMM_SUBS sub-arrays with a counter each, chosen by the source wave: ~850
// atomics on one address serialise at the memory side, ~100 on each of eight hardly do.
#define GEMM_SUBS 8
__global__ __launch_bounds__(256) void regroup_kernel(const u32x4* __restrict__ wbuf, const unsigned int* __restrict__ wcnt,
                                                      int capw, u64* __restrict__ cand, unsigned int* __restrict__ ccnt,
                                                      int capq, unsigned int* __restrict__ overflow) {
    const int wave = blockIdx.x, sub = wave % GEMM_SUBS, caps = capq / GEMM_SUBS;
    const unsigned n = wcnt[wave];
    if (n > (unsigned)capw && threadIdx.x == 0) *overflow = 1u;
    const unsigned m = min(n, (unsigned)capw);
    for (unsigned i = threadIdx.x; i < m; i += 256) {
        const u32x4 e = wbuf[(size_t)wave * capw + i];
        const unsigned at = atomicAdd(&ccnt[e[2] * GEMM_SUBS + sub], 1u);
        if (at < (unsigned)caps) cand[(size_t)e[2] * capq + (size_t)sub * caps + at] = ((u64)e[1] << 32) | e[0];
    }
}

// One block per query: the kc smallest of its candidates (sorted), then the exact re-rank with the
// threshold as an extra bound.  An overflowed candidate array cannot be trusted: exact scan.
// LDS: rerank_lds_bytes(dp, kc) followed by capq + 256 keys (candidates, selection scratch).
__global__ __launch_bounds__(256) void gemm_select_kernel(const ExactParams xp, const u64* cand, const unsigned int* ccnt,
                                                          int capq, const unsigned int* overflow) {
    extern __shared__ __align__(16) unsigned char smem_s[];
    const int q = blockIdx.x, tid = threadIdx.x;
    u64* kin = reinterpret_cast<u64*>(smem_s + (size_t)xp.dp * 4);
    u64* srt = reinterpret_cast<u64*>(smem_s + rerank_lds_bytes(xp.dp, xp.kc));  // [capq], capq a power of two
    // the sub-arrays' fills; an overflowed sub-array or wave buffer means the candidates are not complete
    const int caps = capq / GEMM_SUBS;
    int off[GEMM_SUBS + 1];
    bool lost = *overflow != 0u;
    off[0] = 0;
#pragma unroll
    for (int s_ = 0; s_ < GEMM_SUBS; s_++) {
        const unsigned c_ = ccnt[q * GEMM_SUBS + s_];
        lost = lost || c_ > (unsigned)caps;
        off[s_ + 1] = off[s_] + (int)min(c_, (unsigned)caps);
    }
    const int cnt = off[GEMM_SUBS];
    rerank_stage_query<256>(xp, q, smem_s);
#pragma unroll
    for (int s_ = 0; s_ < GEMM_SUBS; s_++)
        for (int i = tid; i < off[s_ + 1] - off[s_]; i += 256)
            srt[off[s_] + i] = cand[(size_t)q * capq + (size_t)s_ * caps + i];
    __syncthreads();
    // the kc smallest, sorted, into the re-rank's input (cnt <= capq <= 4096; the scratch behind the candidates)
    block_topk_u64(srt, cnt, xp.kc, srt + capq, kin);
    if (lost) {  // block-uniform
        __syncthreads();
        for (int r = tid; r < xp.k; r += 256) emit_exact(xp, (size_t)q * xp.k + r, KEY_PAD);  // overwritten by the exact scan
        if (tid == 0) {
            if (xp.stats) {
                atomicAdd(&xp.stats[0], 1ull);
                atomicAdd(&xp.stats[1], 1ull);
            }
            fallback_list_push(xp, q);
        }
        return;
    }
    rerank_block<256>(xp, q, smem_s, nullptr);
}
