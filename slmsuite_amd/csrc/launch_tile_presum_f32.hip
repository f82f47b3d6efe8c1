// MRAF with a WGS-Leonardo / WGS-Kim update and ONE inverse per column (round 6): col_presum_kernel forms
// D = sum w'^2 - sum w^2 over the signal pixels in a forward-only pass over the columns that hold them, col_tile_kernel
// RULE 5 rebuilds the field with 1 / sqrt(1 + D).  fp32, 4096 / 8192 points, the tile-resident geometry (<= 6 slots).
#include "launch.hpp"

namespace hgs {

template <int N, int PHASE, int NR, int RULE>
static int launch_tile_presum_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_tile_lds_bytes<float, N>();
    auto k = col_tile_kernel<float, N, PHASE, NR, false, true, RULE>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dispatch_note(dispatch_site<KTile, float, N, PHASE, NR, false, true, RULE, -1>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}
// rule 5: the update behind its pre-pass (a.dpartial set); rule 6: MRAF without a weight update
template <int N, int NR>
static int launch_tile_presum_n(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (a.cp.do_update) {
        if (phase == 0) return launch_tile_presum_one<N, 0, NR, 5>(grid, s, a, m0);
        if (phase == 1) return launch_tile_presum_one<N, 1, NR, 5>(grid, s, a, m0);
        return launch_tile_presum_one<N, 2, NR, 5>(grid, s, a, m0);
    }
    if (phase == 0) return launch_tile_presum_one<N, 0, NR, 6>(grid, s, a, m0);
    if (phase == 1) return launch_tile_presum_one<N, 1, NR, 6>(grid, s, a, m0);
    return launch_tile_presum_one<N, 2, NR, 6>(grid, s, a, m0);
}

// the main pass; a.dpartial / a.n_dpartial from launch_presum
int launch_tile_presum(int N, int phase, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    // (compiled per slot count like the plain rule kernels, launch_tile_rule: 1152 SLM rows are 5 slots at 4096 and 3 at 8192)
    if (N == 4096) {
        if (nr <= 4) return launch_tile_presum_n<4096, 4>(phase, grid, s, a, m0);
        if (nr == 5) return launch_tile_presum_n<4096, 5>(phase, grid, s, a, m0);
        return launch_tile_presum_n<4096, 6>(phase, grid, s, a, m0);
    }
    if (N == 8192) {
        if (nr <= 3) return launch_tile_presum_n<8192, 3>(phase, grid, s, a, m0);
        if (nr == 4) return launch_tile_presum_n<8192, 4>(phase, grid, s, a, m0);
        return launch_tile_presum_n<8192, 6>(phase, grid, s, a, m0);
    }
    return (int)hipErrorInvalidValue;
}

template <int N, int NR>
static int launch_presum_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_presum_lds_bytes<float, N>();
    auto k = col_presum_kernel<float, N, NR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dispatch_note(dispatch_site<KPresum, float, N, NR>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}

// the pre-pass: a.col_flags (column scan) required, a.wpartial = where the partials go (one per workgroup and hologram)
int launch_presum(int N, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (a.col_flags == nullptr) return (int)hipErrorInvalidValue;
    if (N == 4096) {
        if (nr <= 4) return launch_presum_one<4096, 4>(grid, s, a, m0);
        if (nr == 5) return launch_presum_one<4096, 5>(grid, s, a, m0);
        return launch_presum_one<4096, 6>(grid, s, a, m0);
    }
    if (N == 8192) {
        if (nr <= 3) return launch_presum_one<8192, 3>(grid, s, a, m0);
        if (nr == 4) return launch_presum_one<8192, 4>(grid, s, a, m0);
        return launch_presum_one<8192, 6>(grid, s, a, m0);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
