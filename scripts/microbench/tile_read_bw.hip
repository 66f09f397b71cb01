// Read bandwidth of a SHORT index (100k x 512 fp32 = 205 MB, Infinity-Cache resident) by access SHAPE and by
// how the rows are dealt to the waves (dev aid, round 3).  The scan kernels load MFMA A operands straight into
// registers: one wave instruction = 16 rows x 64 B (row stride 2 KB).  What do wider per-row pieces reach, and
// what does the 1-or-2-tiles-per-wave imbalance of a 100k-row index cost?
//   shape RxB : one wave instruction covers R rows x B bytes (R * B = 1 KB)
//   deal  tile: a wave owns whole 16-row tiles t0 + w, t0 + w + 8, ... of its block (as the kernels do)
//         even: a wave owns an equal share of its block's bytes (whatever the tile boundaries)
// Build: hipcc -O3 --offload-arch=gfx950 -o tile_read_bw tile_read_bw.hip ; run: ./tile_read_bw [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ROWS rows x (1024 / ROWS) bytes per wave instruction; a "tile" = 16 rows x 2 KB = 32 instructions
template <int ROWS, int INFLIGHT, bool NT>
__global__ __launch_bounds__(512, 4) void read_tiles(const char* __restrict__ xb, int tiles_total, int tiles_per_block,
                                                    int even, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int LPR = 64 / ROWS;           // lanes per row
    const int r = lane / LPR, j = lane % LPR;  // row within the group, 16-byte slot within the row piece
    int t0, t1;
    if (even) {
        t0 = (int)((long long)blockIdx.x * tiles_total / gridDim.x);
        t1 = (int)((long long)(blockIdx.x + 1) * tiles_total / gridDim.x);
    } else {
        t0 = blockIdx.x * tiles_per_block;
        t1 = min(t0 + tiles_per_block, tiles_total);
    }
    f32x4 acc = {0, 0, 0, 0};
    // instruction index space of the block: tiles x 32 instructions; a tile's instruction i covers
    // rows (i % (16 / ROWS)) * ROWS .. + ROWS - 1, bytes (i / (16 / ROWS)) * (1024 / ROWS) .. of each
    constexpr int GROUPS = 16 / ROWS;  // row groups per tile
    auto addr = [&](int tile, int i) -> const f32x4* {
        const int grp = i % GROUPS, piece = i / GROUPS;
        const size_t row = (size_t)tile * 16 + grp * ROWS + r;
        return reinterpret_cast<const f32x4*>(xb + row * 2048 + (size_t)piece * (1024 / ROWS) + j * 16);
    };
    if (even >= 2) {  // equal instruction counts per wave, whatever the tile boundaries
        const long long n_inst = (long long)(t1 - t0) * 32;
        const long long i0 = n_inst * w / 8, i1 = n_inst * (w + 1) / 8;
        long long i = i0;
        for (; i + INFLIGHT <= i1; i += INFLIGHT) {
            f32x4 v[INFLIGHT];
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) {
                const f32x4* a = addr(t0 + (int)((i + u) / 32), (int)((i + u) % 32));
                v[u] = NT ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
            for (int u = 0; u < INFLIGHT; u++) acc += v[u];
        }
        for (; i < i1; i++) acc += *addr(t0 + (int)(i / 32), (int)(i % 32));
    } else {
        for (int tile = t0 + w; tile < t1; tile += 8) {
#pragma unroll 1
            for (int i = 0; i < 32; i += INFLIGHT) {
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int u = 0; u < INFLIGHT; u++) {
                    const f32x4* a = addr(tile, i + u);
                    v[u] = NT ? __builtin_nontemporal_load(a) : *a;
                }
#pragma unroll
                for (int u = 0; u < INFLIGHT; u++) acc += v[u];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

__global__ void fill_random(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t x = (uint32_t)i * 2654435761u + 12345u;
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        p[i] = 0x3F000000u | (x & 0x007FFFFFu);
    }
}

template <typename K>
static double run(K kern, int grid, const char* xb, int tiles, int tpb, int even, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, xb, tiles, tpb, even, out);
    hipEventRecord(e0, 0);
    const int reps = 300;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, xb, tiles, tpb, even, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 100000;
    const int tiles = (int)((rows + 15) / 16);
    const size_t bytes = (size_t)tiles * 16 * 2048;
    char* xb; float* out;
    if (hipMalloc(&xb, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint32_t*)xb, bytes / 4);
    hipDeviceSynchronize();
    const double mb = bytes / 1e6;
    printf("rows %lld = %d tiles = %.1f MB\n", rows, tiles, mb);
    printf("%-14s %-28s %6s %9s %8s\n", "shape", "deal", "grid", "us", "TB/s");
    struct Deal { const char* name; int grid; int tpb; int even; };
    const int r2 = (tiles + 4095) / 4096;  // tiles per wave over 512 blocks x 8 waves
    Deal deals[] = {
        {"tile, even blocks (512)", 512, 0, 1},
        {"tile, 8r per block", (tiles + 8 * r2 - 1) / (8 * r2), 8 * r2, 0},
        {"bytes even per wave (512)", 512, 0, 2},
        {"bytes even per wave (256)", 256, 0, 2},
        {"bytes even per wave (1024)", 1024, 0, 2},
    };
    for (auto& dl : deals) {
#define RUN(R, F, NT, label)                                                                                     \
    {                                                                                                            \
        const double us = run(read_tiles<R, F, NT>, dl.grid, xb, tiles, dl.tpb, dl.even, out);                   \
        printf("%-14s %-28s %6d %9.2f %8.2f\n", label, dl.name, dl.grid, us, mb / us);                          \
        fflush(stdout);                                                                                          \
    }
        RUN(16, 8, false, "16x64  f8");
        RUN(16, 4, false, "16x64  f4");
        RUN(8, 8, false, "8x128  f8");
        RUN(4, 8, false, "4x256  f8");
        RUN(2, 8, false, "2x512  f8");
        RUN(1, 8, false, "1x1024 f8");
        RUN(16, 8, true, "16x64  f8 nt");
        RUN(4, 8, true, "4x256  f8 nt");
        RUN(1, 8, true, "1x1024 f8 nt");
    }
    return 0;
}
