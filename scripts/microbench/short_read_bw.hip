// Read-only bandwidth for SHORT buffers (dev aid, round 3): a 100k x 512 fp32 index is 205 MB and fits
// the 256 MiB Infinity Cache, a 125k-row shard (256 MB) barely.  What does a kernel that only streams such a
// buffer and reduces it in registers reach when it is launched back to back on the same buffer -- and what is
// the floor a launch pays whatever it reads?  Sizes x grid shapes x load policies, random contents.
// Build: hipcc -O3 --offload-arch=gfx950 -o short_read_bw short_read_bw.hip ; run: ./short_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a block streams a contiguous slab, a wave 1 KiB per instruction, UNROLL instructions in flight
template <int UNROLL, bool NT>
__global__ __launch_bounds__(512) void read_slab(const f32x4* __restrict__ p, size_t n_vec, float* out) {
    const size_t per_block = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t b0 = per_block * blockIdx.x;
    const size_t b1 = b0 + per_block < n_vec ? b0 + per_block : n_vec;
    f32x4 acc = {0, 0, 0, 0};
    size_t i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * blockDim.x < b1; i += (size_t)UNROLL * blockDim.x) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            v[u] = NT ? __builtin_nontemporal_load(p + i + (size_t)u * blockDim.x) : p[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    for (; i < b1; i += blockDim.x) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;  // keep the loads
}

__global__ void fill_random(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + 12345u;
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        p[i] = 0x3F000000u | (x & 0x007FFFFFu);  // floats in [0.5, 1)
    }
}

template <typename K>
static double run(K kern, int grid, int block, const f32x4* p, size_t n_vec, float* out, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, p, n_vec, out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, p, n_vec, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

int main() {
    const size_t max_bytes = 2048ull << 20;
    f32x4* p; float* out;
    if (hipMalloc(&p, max_bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)p, max_bytes / 4);
    hipDeviceSynchronize();
    // keep the clocks up
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL((read_slab<8, false>), dim3(1024), dim3(512), 0, 0, p, max_bytes / 16, out);
    hipDeviceSynchronize();
    const double mbs[] = {1, 16, 64, 128, 204.8, 230, 256, 300, 512, 2048};
    printf("%-10s %-8s %-22s %9s %9s\n", "MB", "grid", "variant", "us", "TB/s");
    for (double mb : mbs) {
        const size_t n_vec = (size_t)(mb * 1e6 / 16);
        const int reps = mb < 300 ? 400 : 60;
        for (int grid : {256, 512, 1024, 2048}) {
            struct V { const char* name; double us; } v[] = {
                {"512thr unroll4", run(read_slab<4, false>, grid, 512, p, n_vec, out, reps)},
                {"512thr unroll8", run(read_slab<8, false>, grid, 512, p, n_vec, out, reps)},
                {"512thr unroll8 nt", run(read_slab<8, true>, grid, 512, p, n_vec, out, reps)},
                {"256thr unroll8", run(read_slab<8, false>, grid * 2, 256, p, n_vec, out, reps)},
                {"256thr unroll8 nt", run(read_slab<8, true>, grid * 2, 256, p, n_vec, out, reps)},
            };
            for (auto& x : v)
                printf("%-10.1f %-8d %-22s %9.2f %9.2f\n", mb, grid, x.name, x.us, mb / x.us);
            fflush(stdout);
        }
    }
    return 0;
}
