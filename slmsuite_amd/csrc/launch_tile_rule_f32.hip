// Tile-resident fused column kernel with the weight rule compiled in (col_tile_kernel RULE = 1 / 2): the hot launches of
// a dense fp32 iteration at 4096 / 8192 points.  Its own translation unit so that hipcc builds it in parallel.
#include "launch.hpp"

namespace hgs {

#ifndef HGS_TILE_LISTED
#define HGS_TILE_LISTED 0      // launch_tile_list_f32.hip: 1 (the same kernels walking a tile list, ColArgs::col_list)
#endif
#if HGS_TILE_LISTED
#define LAUNCH_TILE_RULE launch_tile_rule_listed
#else
#define LAUNCH_TILE_RULE launch_tile_rule
#endif

template <int N, int PHASE, int RULE>
static int launch_tile_rule_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_tile_lds_bytes<float, N>();
    auto k = col_tile_kernel<float, N, PHASE, 6, false, false, RULE, HGS_TILE_LISTED>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KTile, float, N, PHASE, 6, false, false, RULE, HGS_TILE_LISTED>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}
template <int N, int RULE>
static int launch_tile_rule_n(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (phase == 0) return launch_tile_rule_one<N, 0, RULE>(grid, s, a, m0);
    if (phase == 1) return launch_tile_rule_one<N, 1, RULE>(grid, s, a, m0);
    return launch_tile_rule_one<N, 2, RULE>(grid, s, a, m0);
}

// rule: 1 = WGS-Leonardo / WGS-Kim update, 2 = no update
int LAUNCH_TILE_RULE(int N, int phase, int rule, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (N == 4096) return rule == 1 ? launch_tile_rule_n<4096, 1>(phase, grid, s, a, m0) : launch_tile_rule_n<4096, 2>(phase, grid, s, a, m0);
    if (N == 8192) return rule == 1 ? launch_tile_rule_n<8192, 1>(phase, grid, s, a, m0) : launch_tile_rule_n<8192, 2>(phase, grid, s, a, m0);
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
