// scan_kernel family: float32 rows, inner product / uncentred L2.  Own translation unit so the families compile in parallel.
#include "ise_scan_launch.hpp"
#include "ise_short_scan.hpp"

void ise_launch_scan_f32_plain(int ch, int waves, int T, dim3 grid, size_t lds, hipStream_t st, const ScanParams& sp) {
    launch_scan_v<false, false>(ch, waves, T, grid, lds, st, sp);
}

void ise_launch_short_f32_plain(int ch, int waves, int bpc, int grid, size_t lds, hipStream_t st, const ScanParams& sp, const ShortParams& tp) {
    launch_short_v<false, false>(ch, waves, bpc, grid, lds, st, sp, tp);
}
