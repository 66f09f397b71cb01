// ise_select.hpp -- wave-level selection primitives on 64-bit candidate keys, shared by the
// scan kernel (ise_scan.hpp) and the exact re-rank / fallback kernels (ise_exact.hpp).
#pragma once
#include "ise_common.hpp"

// ---- wave-level selection primitives on 64-bit keys.  Keys are held as
// kk[e] = element (lane + 64 e), e < KPL; KEY_PAD = empty slot; elements with
// 64 e >= n must be empty.  Real keys are unique and lie strictly between 0 and
// KEY_PAD.  Control flow is wave-uniform (ballot counts in SGPRs).

// Exact: the min(k, #real) smallest keys, written SORTED to dst[0..).  Returns
// the number written; *kth = the k-th smallest key when k were written.
// Quickselect on the key value, then an all-pairs rank among the <= k winners.
template <int KPL>
__device__ __forceinline__ int wave_select(const u64 (&kk)[KPL], int n, int k, u64* dst, u64* kth) {
    int nreal = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n) nreal += __popcll(__ballot(kk[e] != KEY_PAD));
    u64 kstar = KEY_PAD - 1;  // all real keys are ranked; right when few keys are held
    if (nreal > k && nreal > 40) {  // many keys: narrow to the k winners first (quickselect)
        u64 L = 0, H = KEY_PAD;  // the target lies in the open interval (L, H)
        int t = k - 1;           // its rank among the keys of that interval
        for (int round = 0;; round++) {
            u64 P = 0;  // pivot: an element of the interval, position rotated per round
            bool found = false;
            const int rot = (round * 23 + 7) & 63;
#pragma unroll
            for (int e = 0; e < KPL; e++) {
                if (64 * e < n && !found) {
                    const u64 m = __ballot(kk[e] > L && kk[e] < H);
                    if (m) {
                        const u64 hi = (m >> rot) << rot;
                        P = readlane_u64(kk[e], __ffsll((long long)(hi ? hi : m)) - 1);
                        found = true;
                    }
                }
            }
            int c_lt = 0;
#pragma unroll
            for (int e = 0; e < KPL; e++)
                if (64 * e < n) c_lt += __popcll(__ballot(kk[e] > L && kk[e] < P));
            if (c_lt == t) {
                kstar = P;
                break;
            }
            if (c_lt > t) {
                H = P;
            } else {
                L = P;
                t -= c_lt + 1;
            }
        }
    }
    int rk[KPL];  // rank among the winners (keys <= kstar); at most k of them
#pragma unroll
    for (int e = 0; e < KPL; e++) rk[e] = 0;
#pragma unroll
    for (int es = 0; es < KPL; es++) {
        if (64 * es < n) {
            u64 m = __ballot(kk[es] <= kstar);
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const u64 ki = readlane_u64(kk[es], l);
#pragma unroll
                for (int e = 0; e < KPL; e++) rk[e] += (ki < kk[e]) ? 1 : 0;
            }
        }
    }
    u64 kth_key = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++) {
        if (64 * e < n) {
            const bool win = kk[e] <= kstar && rk[e] < k;
            if (win) dst[rk[e]] = kk[e];
            const u64 hit = __ballot(win && rk[e] == k - 1);
            if (hit) kth_key = readlane_u64(kk[e], __ffsll((long long)hit) - 1);
        }
    }
    const int nw = min(nreal, k);
    if (nw == k) *kth = kth_key;
    return nw;
}

// Windowed cut: finds a key P with kmin <= #(keys <= P) <= kmax and writes those
// keys UNSORTED to dst (all real keys if there are at most kmax).  Returns the
// count; *cut = P when the count reached kmin.  A valid, cheap threshold: the
// kmin-th smallest key is <= P.  Window width makes this take a few rounds only.
template <int KPL>
__device__ __forceinline__ int wave_cut(const u64 (&kk)[KPL], int n, int kmin, int kmax, u64* dst, u64* cut) {
    const int lane = threadIdx.x & 63;
    const u64 lt_mask = (1ull << lane) - 1ull;
    int nreal = 0;
    u64 mx = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++)
        if (64 * e < n) nreal += __popcll(__ballot(kk[e] != KEY_PAD));
    u64 P = KEY_PAD - 1;
    int cnt = nreal;
    if (nreal > kmax) {
        u64 L = 0, H = KEY_PAD;
        int base = 0;  // #(keys <= L)
        for (int round = 0;; round++) {
            u64 piv = 0;
            bool found = false;
            const int rot = (round * 23 + 7) & 63;
#pragma unroll
            for (int e = 0; e < KPL; e++) {
                if (64 * e < n && !found) {
                    const u64 m = __ballot(kk[e] > L && kk[e] < H);
                    if (m) {
                        const u64 hi = (m >> rot) << rot;
                        piv = readlane_u64(kk[e], __ffsll((long long)(hi ? hi : m)) - 1);
                        found = true;
                    }
                }
            }
            int c = base;
#pragma unroll
            for (int e = 0; e < KPL; e++)
                if (64 * e < n) c += __popcll(__ballot(kk[e] > L && kk[e] <= piv));
            if (c < kmin) {
                L = piv;
                base = c;
            } else if (c > kmax) {
                H = piv;
            } else {
                P = piv;
                cnt = c;
                break;
            }
        }
    } else if (nreal >= kmin) {
        // every real key stays: the cut is the largest of them
#pragma unroll
        for (int e = 0; e < KPL; e++)
            if (64 * e < n) {
                const u64 v = kk[e] == KEY_PAD ? 0ull : kk[e];
                mx = mx > v ? mx : v;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(mx, o);
            mx = mx > other ? mx : other;
        }
        P = mx;
    }
    int off = 0;
#pragma unroll
    for (int e = 0; e < KPL; e++) {
        if (64 * e < n) {
            const bool keep = kk[e] <= P;  // P < KEY_PAD: empty slots never kept
            const u64 m = __ballot(keep);
            if (keep) dst[off + __popcll(m & lt_mask)] = kk[e];
            off += __popcll(m);
        }
    }
    if (cnt >= kmin) *cut = P;
    return cnt;
}


// Block-wide (256 threads = 4 waves): the min(k, n) smallest of the n unique keys list[0 .. n) in LDS, written
// SORTED to out[0 .. k) (LDS), KEY_PAD behind the last.  Every wave quickselects a quarter of the list
// (wave_select, keys in registers, no barriers), wave 0 the 4 k winners: two barriers in all, where a bitonic
// sort of 1024 keys takes 55 (the select kernels of the large-batch paths: 45 -> see DESIGN.md 4.4).
// n <= 4096 (16 keys per lane), k <= 64; stage: 4 k keys of LDS scratch; `list` is read only.
template <int KPL>
__device__ __forceinline__ void block_topk_part(const u64* list, int n, int base, int per, int k, u64* dst) {
    const int lane = threadIdx.x & 63;
    u64 kk[KPL];
#pragma unroll
    for (int e = 0; e < KPL; e++) {
        const int idx = base + lane + 64 * e;
        kk[e] = (64 * e < per && idx < n) ? list[idx] : KEY_PAD;
    }
    u64 kth_unused = 0;
    const int nw = wave_select<KPL>(kk, per < 64 * KPL ? per : 64 * KPL, k, dst, &kth_unused);
    for (int i = nw + lane; i < k; i += 64) dst[i] = KEY_PAD;
}
__device__ __forceinline__ void block_topk_u64(const u64* list, int n, int k, u64* stage, u64* out) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int per = (((n + 3) >> 2) + 63) & ~63;  // a wave's slice: whole rows of 64 keys
    const int base = w * per;
    if (per <= 64) block_topk_part<1>(list, n, base, per, k, stage + w * k);
    else if (per <= 256) block_topk_part<4>(list, n, base, per, k, stage + w * k);
    else if (per <= 512) block_topk_part<8>(list, n, base, per, k, stage + w * k);
    else block_topk_part<16>(list, n, base, per, k, stage + w * k);
    __syncthreads();
    if (w == 0) {
        u64 kk[4];
#pragma unroll
        for (int e = 0; e < 4; e++) kk[e] = (lane + 64 * e) < 4 * k ? stage[lane + 64 * e] : KEY_PAD;
        u64 kth_unused = 0;
        const int nw = wave_select<4>(kk, 4 * k, k, out, &kth_unused);
        for (int i = nw + lane; i < k; i += 64) out[i] = KEY_PAD;
    }
    __syncthreads();
}
