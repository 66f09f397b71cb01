// ise_gemm_bf16.hpp -- large query batches against bf16 rows (BASELINE config 5: cosine / inner product
// over normalised rows stored bf16, "Q x I^T as MFMA bf16 GEMM + fp32 top-k"): the GEMM-shaped pass of
// ise_gemm_scan.hpp with v_mfma_f32_16x16x32_bf16.
//
// The streaming kernel re-reads the bf16 index once per 64 queries: at nq = 1024 it is HBM-bound at 16
// passes (340 TFLOP/s, 14 % of the bf16 MFMA peak).  Here a wave holds TWO 16-row tiles in registers
// (2 x 16 k-steps x 8 bf16 = 128 VGPRs at d = 512) so that every B fragment read from LDS feeds two
// MFMAs -- with one tile the ds_read_b128 traffic of 8 waves equals the MFMA time -- and the query
// stages (32 queries x d bf16 = 33 KB) arrive by LDS-DMA.  Scores are what the streaming kernel
// computes (even k-steps in one accumulator, odd in the other, fp32): -x.y, or |x|^2 + |y|^2 - 2 x.y
// clamped at 0; bf16 storage is approximate by construction, so there is no re-rank: the admit
// threshold comes from a dumped sample in the same arithmetic and gemm_select_plain_kernel emits the
// k best candidates.  A query whose candidate buffers overflow cannot be answered from them: the
// host enqueues the streaming passes behind this path, gated on the overflow flag.
#pragma once
#include "ise_gemm_scan.hpp"

#define GB_XT 2 /* row tiles per wave */
#define GB_GQ 64 /* queries per LDS stage: two sweeps of 32 per barrier, so that a stage's compute covers the next stage's LDS-DMA */

// queries -> bf16 pairs, padded rows + |x|^2 of the rounded values (wave per query)
__global__ __launch_bounds__(256) void qprep_bf16_kernel(const float* __restrict__ q, int nq, int nq_pad, int d, int S,
                                                         uint32_t* __restrict__ qprep, float* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nq_pad) return;
    float s = 0.f;
    for (int j = lane; j < S; j += 64) {  // unit j = elements 2j, 2j + 1
        float lo = 0.f, hi = 0.f;
        if (i < nq) {
            if (2 * j < d) lo = q[(size_t)i * d + 2 * j];
            if (2 * j + 1 < d) hi = q[(size_t)i * d + 2 * j + 1];
        }
        const __bf16 a = (__bf16)lo, b = (__bf16)hi;
        qprep[(size_t)i * S + j] =
            (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
        const float ra = (float)a, rb = (float)b;
        s = fmaf(ra, ra, s);
        s = fmaf(rb, rb, s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) xn[i] = s;
}

// NS: k-steps of 32 bf16 (64 bytes) per row
template <int NS, bool DUMP, bool L2>
__global__ __launch_bounds__(512, 2) void gemm_scan_bf16_kernel(const GemmScanParams p) {
    extern __shared__ __align__(16) unsigned char smem_gb[];
    const int S = p.S;  // 4-byte units per LDS / qprep row
    float* qbuf0 = reinterpret_cast<float*>(smem_gb);  // [GB_GQ][S] (bf16 pairs)
    float* qbuf1 = qbuf0 + (size_t)GB_GQ * S;
    float* tauL = qbuf1 + (size_t)GB_GQ * S;           // [GEMM_NQ_MAX]
    float* xnL = tauL + GEMM_NQ_MAX;                   // [GEMM_NQ_MAX]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
    const int c = lane & 15, g = lane >> 4;
    const int nstages = p.nq_pad / GB_GQ;
    constexpr bool l2 = L2;
    constexpr int ROWS = 8 * GB_XT * 16;  // rows per slab

    auto stage_load = [&](int st, float* dst) {
        const char* src = reinterpret_cast<const char*>(p.qprep) + (size_t)st * GB_GQ * S * 4;
        int bytes = GB_GQ * S * 4;
#ifdef ISE_ABLATE
        if (p.ablate & 8) bytes >>= 1;   // dev: half the stage (timing only: is the pass bound by the LDS-DMA fill?)
        if (p.ablate & 16) bytes >>= 2;  // dev: a quarter
#endif
        for (int off = w * 1024; off < bytes; off += 8 * 1024)
            if (off + lane * 16 < bytes)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off + lane * 16),
                                                 (lds_ptr_t)(reinterpret_cast<char*>(dst) + off), 16, 0, 0);
    };

    const int part = blockIdx.x % p.qparts;
    const int first = blockIdx.x / p.qparts, step = gridDim.x / p.qparts;
    const int st0 = part * nstages / p.qparts, st1 = (part + 1) * nstages / p.qparts;
    for (int i = tid; i < p.nq_pad; i += 512) {
        // thresholds strictly below FLT_MAX: "score <= tau" then also says "score < FLT_MAX" (and NaN fails it)
        tauL[i] = DUMP ? 0.f : fminf(p.tau[i], 3.4028233e38f);
        xnL[i] = p.xn[i];
    }
    stage_load(st0, qbuf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    unsigned int wfill = 0;
    u32x4* const wmine = DUMP ? nullptr : p.wbuf + (size_t)(blockIdx.x * 8 + w) * p.capw;
    const u64 lt_mask = (1ull << lane) - 1ull;
    const char* xbytes = reinterpret_cast<const char*>(p.xb);
    const size_t row_b = (size_t)p.dp * 2;
    u32x4 a[GB_XT][NS];
    f32x4 yn[GB_XT];
    for (int slab = first; slab < p.slabs; slab += step) {
        const long long row_base = (long long)slab * p.slab_stride * ROWS + w * (GB_XT * 16);
        int nv[GB_XT];  // valid rows among this lane's four of each tile: the per-element bound check, once per slab
#pragma unroll
        for (int xt = 0; xt < GB_XT; xt++) {
            const long long left = p.n - (row_base + xt * 16 + 4 * g);
            nv[xt] = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
            const long long rr = min(row_base + xt * 16 + c, p.rows16 - 1);
            const char* rp = xbytes + (size_t)rr * row_b + 16 * g;
#ifdef ISE_ABLATE
            if ((p.ablate & 64) && slab != first) continue;  // dev: the row tiles are loaded once (timing only)
#endif
#pragma unroll
            for (int s = 0; s < NS; s++) a[xt][s] = *reinterpret_cast<const u32x4*>(rp + 64 * s);
            yn[xt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (l2 && row_base + xt * 16 + 4 * g + 3 < p.rows16)
                yn[xt] = *reinterpret_cast<const f32x4*>(p.norms + row_base + xt * 16 + 4 * g);
        }

        for (int sq = st0; sq < st1; sq++) {
            float* cur = buf ? qbuf1 : qbuf0;
            float* nxt = buf ? qbuf0 : qbuf1;
            // The next stage's LDS-DMA is issued by the first wave of every SIMD (waves 0-3) at the start of this
            // stage and by the second one (waves 4-7) half a stage later: a wave's eight or nine DMA pieces hold
            // its instruction stream for hundreds of cycles, and with both partners of a SIMD issuing them together
            // (they leave the barrier together) the matrix pipe had nothing to do meanwhile.
            const bool late = w >= 4;
            bool staged = false;
#ifdef ISE_ABLATE
            if (p.ablate & 2) staged = true;
            if (p.ablate & 32) {  // dev: every wave at the start of the stage (the round-2 order)
                if (!staged) stage_load(sq + 1 == st1 ? st0 : sq + 1, nxt);
                staged = true;
            }
#endif
            if (!late && !staged) {
                stage_load(sq + 1 == st1 ? st0 : sq + 1, nxt);
                staged = true;
            }
            unsigned younger = 0;  // stores issued behind that DMA (wait_stage_dma, ise_gemm_scan.hpp)

          for (int half = 0; half < GB_GQ / 32; half++) {  // 32 queries at a time: two query tiles
            if (half == 1 && !staged) {
                stage_load(sq + 1 == st1 ? st0 : sq + 1, nxt);
                staged = true;
                younger = 0;  // the appends of the first half are older than the DMA
            }
            const float* q0 = cur + (size_t)(half * 32 + c) * S + 4 * g;
            const float* q1 = q0 + (size_t)16 * S;
            f32x4 acc[GB_XT][2][2];  // [row tile][query tile][k-step parity]: the streaming kernel's two chains
#pragma unroll
            for (int xt = 0; xt < GB_XT; xt++)
#pragma unroll
                for (int t = 0; t < 2; t++) acc[xt][t][0] = acc[xt][t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // B fragments are requested LOOK k-steps ahead of the MFMAs that use them: a k-step is only 4 MFMAs x
            // 16 cycles, less than an LDS round trip.  The reads and their waits are inline asm: left to hipcc,
            // every k-step began with s_waitcnt lgkmcnt(0) -- a wait for the reads just issued, no lookahead at
            // all (SQ_WAIT_ANY 41 % of the wave cycles, MFMA pipe 44 % busy).  LDS operations of a wave retire in
            // issue order, so "at most 2 * (steps requested behind s) outstanding" means step s has landed; the
            // waits take the fragments as operands, so nothing that uses them moves in front.
            constexpr int LOOK = 3;
            const uint32_t la0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q0;
            const uint32_t la1 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q1;
            f32x4 bq[NS][2];
            auto issue = [&](int s_) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][0]) : "v"(la0 + 64u * (uint32_t)s_));
                asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][1]) : "v"(la1 + 64u * (uint32_t)s_));
            };
#pragma unroll
            for (int s = 0; s < LOOK && s < NS; s++) issue(s);
#pragma unroll
            for (int s = 0; s < NS; s++) {
                if (s + LOOK < NS) issue(s + LOOK);
                const int behind = (NS - 1 - s) < LOOK ? (NS - 1 - s) : LOOK;  // steps requested after step s
                if (behind >= 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                const bf16x8 bv0 = __builtin_bit_cast(bf16x8, bq[s][0]), bv1 = __builtin_bit_cast(bf16x8, bq[s][1]);
#pragma unroll
                for (int xt = 0; xt < GB_XT; xt++) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, a[xt][s]);
#ifdef ISE_ABLATE
                    if (p.ablate & 4) {  // operands stay live, no matrix work
                        acc[xt][0][s & 1] += __builtin_bit_cast(f32x4, a[xt][s]) + bq[s][0];
                        acc[xt][1][s & 1] += bq[s][1];
                        continue;
                    }
#endif
                    acc[xt][0][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv0, acc[xt][0][s & 1], 0, 0, 0);
                    acc[xt][1][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv1, acc[xt][1][s & 1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int xt = 0; xt < GB_XT; xt++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const f32x4 dot = acc[xt][t][0] + acc[xt][t][1];
#ifdef ISE_ABLATE
                    if (p.ablate & 1) {  // keep the accumulators live, skip the bookkeeping
                        asm volatile("" ::"v"(dot[0]), "v"(dot[1]), "v"(dot[2]), "v"(dot[3]));
                        continue;
                    }
#endif
                    const int q = sq * GB_GQ + half * 32 + t * 16 + c;
                    const float xq_n = l2 ? xnL[q] : 0.f, tq = tauL[q];
                    const long long r0 = row_base + xt * 16 + 4 * g;
                    // fast reject on the lane's BEST of its four rows (three instructions for the inner product: a
                    // max3, a max and a compare against -tau; rows past the end only ever make it say "maybe"): the
                    // per-row tests run only for the tiles the ballot lets through
                    float sc[4];
                    float best;
                    if (l2) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float v = (xq_n + yn[xt][j]) - 2.f * dot[j];
                            sc[j] = v < 0.f ? 0.f : v;  // keeps NaN
                        }
                        best = fminf(fminf(sc[0], sc[1]), fminf(sc[2], sc[3]));
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) sc[j] = -dot[j];
                        best = -fmaxf(fmaxf(dot[0], dot[1]), fmaxf(dot[2], dot[3]));
                    }
                    if constexpr (DUMP) {
                        f32x4 o;
#pragma unroll
                        for (int j = 0; j < 4; j++) o[j] = (j < nv[xt] && sc[j] < FLT_MAX) ? sc[j] : FLT_MAX;
                        *reinterpret_cast<f32x4*>(p.dump + (size_t)q * ((size_t)p.slabs * ROWS) + (size_t)slab * ROWS +
                                                  w * (GB_XT * 16) + xt * 16 + 4 * g) = o;
                        younger++;
                    } else if (__ballot(best <= tq)) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const bool pass_j = sc[j] <= tq && j < nv[xt];
                            const u64 m = __ballot(pass_j);
                            if (m) {
                                const unsigned at = wfill + (unsigned)__popcll(m & lt_mask);
                                if (pass_j && at < (unsigned)p.capw) {
                                    u32x4 e;
                                    e[0] = (uint32_t)((uint32_t)(r0 + j) + p.id_base);
                                    e[1] = ord_f32(sc[j]);
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
          }
            wait_stage_dma(younger);
            __syncthreads();
            buf ^= 1;
        }
    }
    if (!DUMP && lane == 0) p.wcnt[blockIdx.x * 8 + w] = wfill;
}

// One block per query: sort its candidates, emit the k best as they are (no re-rank for bf16 rows / inner
// product).  Incomplete candidates (an overflowed buffer) raise the rerun flag instead: the host has the
// streaming passes queued behind this kernel, gated on that flag, and they rewrite every query of the chunk.
// LDS: capq + 320 keys (candidates, selection scratch, the k best).
__global__ __launch_bounds__(256) void gemm_select_plain_kernel(const u64* cand, const unsigned int* ccnt, int capq,
                                                                const unsigned int* overflow, unsigned int* rerun, int k,
                                                                int metric, float* D, long long* I, u64* keys_out) {
    extern __shared__ __align__(16) unsigned char smem_sp[];
    u64* srt = reinterpret_cast<u64*>(smem_sp);
    const int q = blockIdx.x, tid = threadIdx.x;
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
    if (lost) {
        if (tid == 0) *rerun = 1u;
        return;
    }
    const int cnt = off[GEMM_SUBS];
#pragma unroll
    for (int s_ = 0; s_ < GEMM_SUBS; s_++)
        for (int i = tid; i < off[s_ + 1] - off[s_]; i += 256)
            srt[off[s_] + i] = cand[(size_t)q * capq + (size_t)s_ * caps + i];
    __syncthreads();
    u64* best = srt + capq + 4 * 64;  // [64] behind the candidates and block_topk_u64's scratch
    block_topk_u64(srt, cnt, k, srt + capq, best);
    for (int r = tid; r < k; r += 256) {
        const u64 key = best[r];
        const size_t o = (size_t)q * k + r;
        if (keys_out) keys_out[o] = key;
        if (D) {
            const bool pad = key == KEY_PAD;
            const float sc = unord_f32((uint32_t)(key >> 32));
            const bool l2 = metric == ISE_METRIC_L2;
            D[o] = pad ? (l2 ? FLT_MAX : -FLT_MAX) : (l2 ? sc : -sc);
            I[o] = pad ? -1ll : (long long)(uint32_t)key;
        }
    }
}
