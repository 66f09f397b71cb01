// ise_rows.hpp -- wave-per-row helper kernels: norms, padding / conversion, normalize_L2, shift vector.
#pragma once
#include "ise_common.hpp"

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

// column mean of `rows` rows (d columns of a padded row).  Two levels, both in a fixed order
// (`groups` row groups summed in row order, then the groups in group order): deterministic for a
// given row count.  NaN / inf entries are skipped (they must not poison every distance).
#define COLMEAN_GROUPS_MAX 1024
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, long long rows, int d, int dp,
                                                      int groups, float* __restrict__ partial /* [groups][dp] */) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int gidx = blockIdx.y;
    if (j >= dp) return;
    const long long per = (rows + groups - 1) / groups;
    const long long r0 = gidx * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    if (j < d)
        for (long long r = r0; r < r1; r++) {
            const float v = x[(size_t)r * dp + j];
            if (fabsf(v) <= FLT_MAX) s += v;
        }
    partial[(size_t)gidx * dp + j] = s;
}
__global__ __launch_bounds__(256) void col_mean_kernel(const float* __restrict__ partial, long long rows, int d, int dp,
                                                       int groups, float* __restrict__ mu) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dp) return;
    float s = 0.f;
    for (int gi = 0; gi < groups; gi++) s += partial[(size_t)gi * dp + j];
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

// ---------------------------------------------------------------- visual-word histograms
// One block per image: np.histogram(labels_of_image, bins=K) exactly as the reference's BoVW loop
// calls it (backend/bag_of_visual_words.py:98-106) -- K equal-width bins between the image's OWN
// smallest and largest label (no range argument), not a bincount over [0, K).  The float64
// arithmetic follows numpy's uniform-bin path operation for operation (numpy/lib/_histograms_impl.py:
// scale, truncate, then the two one-ulp corrections against linspace edges j*step + first), with
// contraction off so that no multiply-add pair is fused.  Counts accumulate in LDS.
#define HIST_K_MAX 16384
__global__ __launch_bounds__(256) void bovw_histogram_kernel(const long long* __restrict__ labels,
                                                             const long long* __restrict__ offsets, int K,
                                                             double* __restrict__ out) {
#pragma clang fp contract(off)
    extern __shared__ unsigned int hist[];  // [K]
    __shared__ long long red_lo[4], red_hi[4];
    const long long img = blockIdx.x;
    const long long b = offsets[img], e = offsets[img + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int j = tid; j < K; j += 256) hist[j] = 0u;
    long long lo = 0x7fffffffffffffffll, hi = -0x7fffffffffffffffll - 1;
    for (long long i = b + tid; i < e; i += 256) {
        const long long a = labels[i];
        lo = a < lo ? a : lo;
        hi = a > hi ? a : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const long long lo2 = __shfl_xor(lo, off), hi2 = __shfl_xor(hi, off);
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
    }
    if (lane == 0) { red_lo[w] = lo; red_hi[w] = hi; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        lo = red_lo[i] < lo ? red_lo[i] : lo;
        hi = red_hi[i] > hi ? red_hi[i] : hi;
    }
    if (e > b) {
        double first = (double)lo, last = (double)hi, denom;
        if (lo == hi) {  // numpy widens an empty range by half a unit on both sides
            first = first - 0.5;
            last = last + 0.5;
            denom = last - first;
        } else {
            denom = (double)(unsigned long long)(hi - lo);
        }
        const double step = (last - first) / (double)K;  // np.linspace(first, last, K + 1)
        auto edge = [&](long long j) -> double {
            if (j == K) return last;
            const double t = (double)j * step;
            return t + first;
        };
        for (long long i = b + tid; i < e; i += 256) {
            const double a = (double)labels[i];
            const double f = ((a - first) / denom) * (double)K;
            long long idx = (long long)f;
            if (idx == K) idx -= 1;
            if (a < edge(idx)) idx -= 1;
            if (a >= edge(idx + 1) && idx != K - 1) idx += 1;
            atomicAdd(&hist[idx], 1u);
        }
    }
    __syncthreads();
    double* o = out + (size_t)img * K;
    for (int j = tid; j < K; j += 256) o[j] = (double)hist[j];
}
