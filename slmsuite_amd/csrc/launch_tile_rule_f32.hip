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

template <int N, int PHASE, int RULE, int NR>
static int launch_tile_rule_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_tile_lds_bytes<float, N>();
    auto k = col_tile_kernel<float, N, PHASE, NR, false, false, RULE, HGS_TILE_LISTED>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KTile, float, N, PHASE, NR, false, false, RULE, HGS_TILE_LISTED>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}
template <int N, int RULE, int NR>
static int launch_tile_rule_n(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (phase == 0) return launch_tile_rule_one<N, 0, RULE, NR>(grid, s, a, m0);
    if (phase == 1) return launch_tile_rule_one<N, 1, RULE, NR>(grid, s, a, m0);
    return launch_tile_rule_one<N, 2, RULE, NR>(grid, s, a, m0);
}
template <int N, int NR>
static int launch_tile_rule_r(int phase, int rule, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    return rule == 1 ? launch_tile_rule_n<N, 1, NR>(phase, grid, s, a, m0) : launch_tile_rule_n<N, 2, NR>(phase, grid, s, a, m0);
}

// rule: 1 = WGS-Leonardo / WGS-Kim update, 2 = no update
// nr: register slots of the load layout the SLM rows occupy (<= 6) -- the instance is compiled for exactly that many: the
// first / last radix-4 layer of the transforms, the radix-2 step of the 8192-point transform, its pair exchange and the tile
// registers all shrink with the slots (1152 SLM rows: 5 slots of 256 rows at 4096, 3 slots of 512 at 8192; rounds 2 - 4 ran
// every geometry on the six-slot instance).  The 4096-point transform prunes from four slots on.
int LAUNCH_TILE_RULE(int N, int phase, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (N == 4096) {
        if (nr <= 4) return launch_tile_rule_r<4096, 4>(phase, rule, grid, s, a, m0);
        if (nr == 5) return launch_tile_rule_r<4096, 5>(phase, rule, grid, s, a, m0);
        return launch_tile_rule_r<4096, 6>(phase, rule, grid, s, a, m0);
    }
    if (N == 8192) {
        if (nr <= 3) return launch_tile_rule_r<8192, 3>(phase, rule, grid, s, a, m0);
        if (nr == 4) return launch_tile_rule_r<8192, 4>(phase, rule, grid, s, a, m0);
        if (nr == 5) return launch_tile_rule_r<8192, 5>(phase, rule, grid, s, a, m0);
        return launch_tile_rule_r<8192, 6>(phase, rule, grid, s, a, m0);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
