// What does the bf16 GEMM-shaped pass (csrc/ise_gemm_bf16.hpp) have to work with?  Its inner loop, stripped down
// step by step (dev aid, round 3): two 16-row tiles per wave in registers (A operands), per k-step two B fragments
// and four v_mfma_f32_16x16x32_bf16, 16 k-steps per 32-query sweep, 8 waves per CU (two per SIMD), random data.
//   regs   : B fragments in registers (the matrix pipe alone)
//   lds    : B fragments read from LDS by ds_read_b128, three k-steps ahead (no writes to LDS meanwhile)
//   lds+bar: the same with a workgroup barrier every 128 MFMAs (the stage hand-over of the kernel)
//   +epi   : the same with the kernel's score / threshold compare per sweep (no stores)
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_rate mfma_bf16_rate.hip ; run: ./mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint32_t* __restrict__ seed, float* out, int sweeps) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NS = 16, S = 264;  // k-steps per sweep; LDS row stride in 4-byte units ((S / 4) mod 16 == 2)
    float* qbuf = reinterpret_cast<float*>(smem);  // [64][S]
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * S; i += 512) reinterpret_cast<uint32_t*>(qbuf)[i] = seed[(i * 7 + blockIdx.x) & 4095];
    u32x4 a[2][NS];
#pragma unroll
    for (int xt = 0; xt < 2; xt++)
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t v = seed[(tid * 33 + s * 5 + xt) & 4095];
            a[xt][s] = (u32x4){v, v ^ 0x01010101u, v + 0x00020002u, v ^ 0x10101010u};
        }
    __syncthreads();
    f32x4 tot = {0, 0, 0, 0};
    float tq = seed[lane] * 1e-30f;
    for (int sw = 0; sw < sweeps; sw++) {
        const float* q0 = qbuf + (size_t)((sw & 1) * 32 + c) * S + 4 * g;
        const float* q1 = q0 + (size_t)16 * S;
        f32x4 acc[2][2][2];
#pragma unroll
        for (int xt = 0; xt < 2; xt++)
#pragma unroll
            for (int t = 0; t < 2; t++) acc[xt][t][0] = acc[xt][t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 bq[NS][2];
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
                bq[s][0] = __builtin_bit_cast(f32x4, a[0][(s + 3) & 15]);
                bq[s][1] = __builtin_bit_cast(f32x4, a[1][(s + 7) & 15]);
            }
        }
        constexpr int LOOK = 3;
        const uint32_t la0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q0;
        const uint32_t la1 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q1;
        auto issue = [&](int s_) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][0]) : "v"(la0 + 64u * (uint32_t)s_));
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][1]) : "v"(la1 + 64u * (uint32_t)s_));
        };
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < LOOK; s++) issue(s);
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (MODE >= 1) {
                if (s + LOOK < NS) issue(s + LOOK);
                const int behind = (NS - 1 - s) < LOOK ? (NS - 1 - s) : LOOK;
                if (behind >= 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
            }
            const bf16x8 bv0 = __builtin_bit_cast(bf16x8, bq[s][0]), bv1 = __builtin_bit_cast(bf16x8, bq[s][1]);
#pragma unroll
            for (int xt = 0; xt < 2; xt++) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, a[xt][s]);
                acc[xt][0][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv0, acc[xt][0][s & 1], 0, 0, 0);
                acc[xt][1][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv1, acc[xt][1][s & 1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int xt = 0; xt < 2; xt++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f32x4 dot = acc[xt][t][0] + acc[xt][t][1];
                if (MODE >= 3) {  // the kernel's epilogue without its (rare) stores
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < 4; j++) any |= (-dot[j] <= tq) && j < 4;
                    if (__ballot(any)) tot += dot;
                } else {
                    tot += dot;
                }
            }
        if (MODE >= 2 && (sw & 1)) __syncthreads();
    }
    if (tot[0] + tot[1] + tot[2] + tot[3] == 12345.678f) out[0] = 1.f;
}

template <int MODE>
static void run(const char* name, const uint32_t* seed, float* out) {
    const int sweeps = 2048, grid = 256;
    const size_t lds = 64 * 264 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(512), lds, 0, seed, out, sweeps);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(512), lds, 0, seed, out, sweeps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * sweeps * 64 * (16.0 * 16 * 32 * 2);
    printf("%-10s %8.3f ms per launch  %8.1f TFLOP/s  (%4.1f %% of 2500)\n", name, ms / reps, flops / (ms / reps * 1e-3) / 1e12,
           flops / (ms / reps * 1e-3) / 1e12 / 25.0);
    fflush(stdout);
}

// ---- the same questions for ONE wave per SIMD holding XT row tiles (4 waves per CU, <= 512 VGPRs each): every B
// fragment read from LDS then feeds XT tiles (half the LDS traffic per MFMA at XT = 4), but nothing hides a
// wave's own stalls.  MODE 1: LDS reads; 2: + barrier per 2 sweeps; 3: + threshold compare after each sweep;
// 4: the compare of the PREVIOUS sweep sliced between the k-steps of the current one (two accumulator sets)
template <int MODE, int XT>
__global__ __launch_bounds__(256, 1) void mfma_wide(const uint32_t* __restrict__ seed, float* out, int sweeps) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NS = 16, S = 264;
    float* qbuf = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 64 * S; i += 256) reinterpret_cast<uint32_t*>(qbuf)[i] = seed[(i * 7 + blockIdx.x) & 4095];
    u32x4 a[XT][NS];
#pragma unroll
    for (int xt = 0; xt < XT; xt++)
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const uint32_t v = seed[(tid * 33 + s * 5 + xt) & 4095];
            a[xt][s] = (u32x4){v, v ^ 0x01010101u, v + 0x00020002u, v ^ 0x10101010u};
        }
    __syncthreads();
    f32x4 tot = {0, 0, 0, 0};
    const float tq = seed[lane] * 1e-30f;
    f32x4 prev[XT][2];
#pragma unroll
    for (int xt = 0; xt < XT; xt++) prev[xt][0] = prev[xt][1] = (f32x4){1.f, 1.f, 1.f, 1.f};
    auto epi = [&](const f32x4& dot) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) any |= (-dot[j] <= tq);
        if (__ballot(any)) tot += dot;
    };
    for (int sw = 0; sw < sweeps; sw++) {
        const float* q0 = qbuf + (size_t)((sw & 1) * 32 + c) * S + 4 * g;
        const float* q1 = q0 + (size_t)16 * S;
        f32x4 acc[XT][2][2];
#pragma unroll
        for (int xt = 0; xt < XT; xt++)
#pragma unroll
            for (int t = 0; t < 2; t++) acc[xt][t][0] = acc[xt][t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 bq[NS][2];
        constexpr int LOOK = 3;
        const uint32_t la0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q0;
        const uint32_t la1 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)q1;
        auto issue = [&](int s_) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][0]) : "v"(la0 + 64u * (uint32_t)s_));
            asm volatile("ds_read_b128 %0, %1" : "=v"(bq[s_][1]) : "v"(la1 + 64u * (uint32_t)s_));
        };
#pragma unroll
        for (int s = 0; s < LOOK; s++) issue(s);
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (s + LOOK < NS) issue(s + LOOK);
            const int behind = (NS - 1 - s) < LOOK ? (NS - 1 - s) : LOOK;
            if (behind >= 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
            else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
            else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[s][0]), "+v"(bq[s][1]));
            const bf16x8 bv0 = __builtin_bit_cast(bf16x8, bq[s][0]), bv1 = __builtin_bit_cast(bf16x8, bq[s][1]);
#pragma unroll
            for (int xt = 0; xt < XT; xt++) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, a[xt][s]);
                acc[xt][0][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv0, acc[xt][0][s & 1], 0, 0, 0);
                acc[xt][1][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv1, acc[xt][1][s & 1], 0, 0, 0);
            }
            if (MODE == 4 && s < 2 * XT) epi(prev[s >> 1][s & 1]);  // a slice of the previous sweep's compare
        }
#pragma unroll
        for (int xt = 0; xt < XT; xt++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const f32x4 dot = acc[xt][t][0] + acc[xt][t][1];
                if (MODE == 3) epi(dot);
                else if (MODE == 4) prev[xt][t] = dot;
                else tot += dot;
            }
        if (MODE >= 2 && (sw & 1)) __syncthreads();
    }
    if (MODE == 4) {
#pragma unroll
        for (int xt = 0; xt < XT; xt++) tot += prev[xt][0] + prev[xt][1];
    }
    if (tot[0] + tot[1] + tot[2] + tot[3] == 12345.678f) out[0] = 1.f;
}

template <int MODE, int XT>
static void run_wide(const char* name, const uint32_t* seed, float* out) {
    const int sweeps = 2048, grid = 256;
    const size_t lds = 64 * 264 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_wide<MODE, XT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((mfma_wide<MODE, XT>), dim3(grid), dim3(256), lds, 0, seed, out, sweeps);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((mfma_wide<MODE, XT>), dim3(grid), dim3(256), lds, 0, seed, out, sweeps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * sweeps * (16.0 * XT * 2) * (16.0 * 16 * 32 * 2);
    printf("%-22s %8.3f ms per launch  %8.1f TFLOP/s  (%4.1f %% of 2500)\n", name, ms / reps, flops / (ms / reps * 1e-3) / 1e12,
           flops / (ms / reps * 1e-3) / 1e12 / 25.0);
    fflush(stdout);
}

int main() {
    uint32_t* seed; float* out;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 4);
    uint32_t h[4096];
    srand(1);
    for (int i = 0; i < 4096; i++) {  // pairs of bf16 in [-1, 1)
        auto bf = [](float f) { union { float f; uint32_t u; } x; x.f = f; return (uint32_t)(x.u >> 16); };
        h[i] = bf(rand() / (float)RAND_MAX * 2 - 1) | (bf(rand() / (float)RAND_MAX * 2 - 1) << 16);
    }
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("regs", seed, out);
        run<1>("lds", seed, out);
        run<2>("lds+bar", seed, out);
        run<3>("lds+bar+epi", seed, out);
        run_wide<1, 4>("wide4 lds", seed, out);
        run_wide<2, 4>("wide4 lds+bar", seed, out);
        run_wide<3, 4>("wide4 lds+bar+epi", seed, out);
        run_wide<4, 4>("wide4 lds+bar+epi piped", seed, out);
        run_wide<3, 5>("wide5 lds+bar+epi", seed, out);
        run_wide<4, 5>("wide5 lds+bar+epi piped", seed, out);
    }
    return 0;
}
