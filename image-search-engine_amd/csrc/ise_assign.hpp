// ise_assign.hpp -- k = 1 nearest-centroid assignment kernel (MFMA-bound).
#pragma once
#include "ise_common.hpp"

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
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: a scalar
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
