// ise_scan_launch.hpp -- (ch, waves, T) -> scan_kernel instantiation, for one kernel family per
// translation unit (ise_scan_f32_shift.hip, ise_scan_f32_plain.hip, ise_scan_bf16.hip).
#pragma once
#include "ise_scan.hpp"

template <int CH, int W, int T, bool BF16, bool SHIFT>
static void launch_one(dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    static LdsAttrOnce attr;
    attr.ensure(reinterpret_cast<const void*>(&scan_kernel<CH, W, T, BF16, SHIFT>), LDS_LIMIT);
    hipLaunchKernelGGL((scan_kernel<CH, W, T, BF16, SHIFT>), grid, dim3(W * 64), lds, st, sp);
}
template <int W, int T, bool BF16, bool SHIFT>
static void launch_scan_ch(int ch, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    switch (ch) {
        case 8:  // 16-wave blocks never use 8-step chunks (register budget)
            if constexpr (W < 16) launch_one<8, W, T, BF16, SHIFT>(grid, lds, st, sp);
            else launch_one<4, W, T, BF16, SHIFT>(grid, lds, st, sp);
            break;
        case 4: launch_one<4, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
        case 2: launch_one<2, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
        default: launch_one<1, W, T, BF16, SHIFT>(grid, lds, st, sp); break;
    }
}
// variants built: (waves, T) in {(8,1), (4,1), (16,2), (8,2), (8,3), (8,4 bf16)}; 4 waves exist for
// one query tile only (long rows, where 8 waves' lists no longer fit beside the tile); two query
// tiles run one 16-wave block per CU when the lists fit (more waves to hide the bookkeeping
// behind: 430 -> 407 us at 1M x 512, 85 -> 71 us at 125k)
template <bool BF16, bool SHIFT>
static void launch_scan_v(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    if (T == 1 && waves == 4) launch_scan_ch<4, 1, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 1) launch_scan_ch<8, 1, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 2 && waves == 16) launch_scan_ch<16, 2, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 2) launch_scan_ch<8, 2, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if (T == 3) launch_scan_ch<8, 3, BF16, SHIFT>(ch, grid, lds, st, sp);
    else if constexpr (BF16) launch_scan_ch<8, 4, true, false>(ch, grid, lds, st, sp);  // 64 queries per pass: bf16 rows only
}
