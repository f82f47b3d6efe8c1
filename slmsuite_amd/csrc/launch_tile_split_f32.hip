// Single-pass MRAF (col_tile_kernel RULE 3 + row_kernel SPLIT): the column kernel transforms the signal part and the
// noise part of the constrained field separately, the row kernel joins them with 1 / ||w'||.  fp32, 4096 / 8192 points.
#include "launch.hpp"

namespace hgs {

#ifndef HGS_SPLIT_STATS
#define HGS_SPLIT_STATS 0      // launch_tile_split_stats_f32.hip: 1 (the same kernels accumulating the in-pass statistics)
#endif
#if HGS_SPLIT_STATS
#define LAUNCH_TILE_SPLIT launch_tile_split_stats
#else
#define LAUNCH_TILE_SPLIT launch_tile_split
#endif

template <int N, int PHASE, int NR, int RULE>
static int launch_tile_split_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_tile_split_lds_bytes<float, N>();
    auto k = col_tile_kernel<float, N, PHASE, NR, HGS_SPLIT_STATS != 0, true, RULE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dispatch_note(dispatch_site<KTile, float, N, PHASE, NR, HGS_SPLIT_STATS != 0, true, RULE, -1>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}
template <int N, int NR, int RULE>
static int launch_tile_split_r(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (phase == 0) return launch_tile_split_one<N, 0, NR, RULE>(grid, s, a, m0);
    if (phase == 1) return launch_tile_split_one<N, 1, NR, RULE>(grid, s, a, m0);
    return launch_tile_split_one<N, 2, NR, RULE>(grid, s, a, m0);
}
// RULE 4 (the WGS-Leonardo / WGS-Kim update and the MRAF branches compiled in) where the launch is exactly that AND walks a
// tile list -- measured at 8192^2 (cfg 5): list launch 153.7 -> 146.8 us, dense launch 311.0 -> 315.4 us (twice each), so
// dense launches keep the generic form; the statistics unit has the generic form only
template <int N, int NR>
static int launch_tile_split_n(int phase, int rule_ok, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
#if !HGS_SPLIT_STATS
    const bool fixed = a.cp.do_update && a.cp.mraf && !a.cp.nog_pass && !a.cp.weights_only && a.cp.nog == nullptr &&
                       (a.cp.method == M_LEONARDO || a.cp.method == M_KIM) && rule_ok && a.col_list != nullptr;
    if (fixed) return launch_tile_split_r<N, NR, 4>(phase, grid, s, a, m0);
#endif
    return launch_tile_split_r<N, NR, 3>(phase, grid, s, a, m0);
}

// nr: register slots of the load layout the SLM rows occupy (<= 6); up to four keep the noise tile in registers
// rule_ok: the rule-specialised form (RULE 4) may be used
int LAUNCH_TILE_SPLIT(int N, int phase, int nr, int rule_ok, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (N == 4096) return nr <= 4 ? launch_tile_split_n<4096, 4>(phase, rule_ok, grid, s, a, m0) : launch_tile_split_n<4096, 6>(phase, rule_ok, grid, s, a, m0);
    if (N == 8192) return nr <= 4 ? launch_tile_split_n<8192, 4>(phase, rule_ok, grid, s, a, m0) : launch_tile_split_n<8192, 6>(phase, rule_ok, grid, s, a, m0);
    return (int)hipErrorInvalidValue;
}

#if !HGS_SPLIT_STATS
template <int N, int MODE, int NS>
static int launch_row_split_one(dim3 grid, hipStream_t s, const RowArgs<float>& a) {
    constexpr size_t lds = (size_t)RowCfg<N>::FPW * lds_elems<N>() * sizeof(Cx<float>);
    auto k = row_kernel<float, N, MODE, NS, false, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KRow, float, N, MODE, NS, false, true>(), row_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(RowCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}
template <int N>
static int launch_row_split_n(int mode, dim3 grid, hipStream_t s, const RowArgs<float>& a) {
    if (a.shifted) return mode == 1 ? launch_row_split_one<N, 1, 8>(grid, s, a) : launch_row_split_one<N, 2, 8>(grid, s, a);
    return mode == 1 ? launch_row_split_one<N, 1, 16>(grid, s, a) : launch_row_split_one<N, 2, 16>(grid, s, a);
}

// mode 1 / 2 (row_kernel MODE); a.gh2 set
int launch_row_split(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<float>& a) {
    if (mode != 1 && mode != 2) return (int)hipErrorInvalidValue;
    if (N == 4096) return launch_row_split_n<4096>(mode, grid, s, a);
    if (N == 8192) return launch_row_split_n<8192>(mode, grid, s, a);
    return (int)hipErrorInvalidValue;
}

#endif

}  // namespace hgs
