// ise_scan_params.hpp -- what the host side and the scan-kernel translation units share:
// the kernel's parameter block, its list-size constants, its LDS layout, and the three launch
// entry points (one per kernel family; each lives in its own .hip file so the families build in
// parallel).
#pragma once
#include "ise_common.hpp"

// ---------------------------------------------------------------- scan kernel
struct ScanParams {
    const void* xb;      // [cap][dp] float32 or bf16 rows; 16-byte "slots": row_slots per row
    const float* norms;  // [cap]
    const float* q;      // [nq][d]
    const float* mu;     // [dp] shift vector (zero padded), used by SHIFT kernels
    const u64* floor_keys;  // optional [nq]: only keys > floor enter (k > 64 passes)
    u64* part;           // [nqt][nb][16 T][k]
    long long n;         // rows in the index
    int d, dp, qs_stride;  // qs_stride: LDS query row stride in 4-byte units (floats, or bf16 pairs)
    int row_slots;         // 16-byte slots per index row = dp * elem_size / 16; one k-step = 4 slots
    int nq, k, kb, metric;  // kb: block-list slots per query (16 or 32, >= k)
    uint32_t id_base;
    float beta;  // SHIFT kernels: relative width of the lower bound the rows are keyed by (0 = none)
    int tiles_total, tiles_per_block;
    // one-shot threshold exchange: [nqt][16 T][nblocks] entries (launch seq << 32 | ord(score) of the
    // block's best boot row); null = off.  See scan_kernel.
    unsigned long long* xchg;
    uint32_t xchg_seq;
    const unsigned int* gate;  // optional: the launch does nothing unless *gate != 0 (queued reruns, ise_gemm_bf16.hpp)
    int ablate;  // dev builds (-DISE_ABLATE): bit mask of phases to skip, from $ISE_ABLATE
    unsigned long long* stamps;  // dev builds: [blocks][waves][16] stamps (0-7 s_memrealtime 100 MHz, 8-9 s_memtime), or null
};

#define CAP 16       /* slots of a wave's private candidate list; folded out at MERGE_TRIG */
#define MERGE_TRIG 12
#define KB_MAX 48    /* block-list slots per query (runtime kb = 16, 32 or 48), >= k of one pass; KB_MAX + CAP <= 64 */

#define LDS_LIMIT (160 * 1024)

// LDS bytes of one scan block (host and device agree through this function)
__host__ __device__ constexpr size_t scan_lds_layout(int S, int waves, int T, int kb) {
    return (size_t)S * 4 /* mus: the shift vector, laid out like one query row */ +
           (size_t)(16 * T) * ((size_t)S * 4 + 4 /* qs, xn */ + 8 /* tauS */ + 8 /* bwc, lockS */ +
                               (size_t)waves * 4 /* cntS */ + (size_t)kb * 8 /* bootw */ +
                               (size_t)waves * CAP * 8 /* cand; boot staging aliases it */);
}

// Launch the (ch, waves, T) variant of one kernel family.  Variants built per family:
// (waves, T) in {(8,1), (4,1), (16,2), (8,2), (8,3)} and (8,4) for bf16 rows.
void ise_launch_scan_f32_shift(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp);
void ise_launch_scan_f32_plain(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp);
void ise_launch_scan_bf16(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp);
