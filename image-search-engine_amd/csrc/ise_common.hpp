// ise_common.hpp -- types and device utilities shared by the kernels of libise_knn.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/ise_knn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define KEY_PAD (~0ull)
#define TAU0 ((u64)0xFF7FFFFFu << 32) /* ord(FLT_MAX) << 32: strict gate score < FLT_MAX */

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set once per (kernel, device),
// race-free (one static LdsAttrOnce per launcher; the launch that follows is ordered behind the set by the mutex)
struct LdsAttrOnce {
    std::atomic<unsigned long long> done{0};  // bit i: set on device i (< 64 devices per process)
    std::mutex mu;
    void ensure(const void* kernel, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        const unsigned long long bit = 1ull << dev;
        if (done.load(std::memory_order_acquire) & bit) return;
        std::lock_guard<std::mutex> lk(mu);
        if (done.load(std::memory_order_relaxed) & bit) return;
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            (void)hipGetLastError();  // a launch that needs the attribute then fails with its own error
        done.fetch_or(bit, std::memory_order_release);
    }
};

// dev builds (-DISE_ABLATE): block 0 / lane 0 stamps the 100 MHz real-time clock into a debug buffer
#ifdef ISE_ABLATE
#define DBG_STAMP(buf, i)                                                                   \
    do {                                                                                    \
        if ((buf) && blockIdx.x == 0 && threadIdx.x == 0) (buf)[(i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DBG_STAMP(buf, i) do {} while (0)
#endif
