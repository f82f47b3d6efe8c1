// Per-column fused kernel with the weight rule compiled in (col_fused_kernel RULE = 2): fp32, no statistics, no extras.
// Its own translation unit so that hipcc builds the no-update variants in parallel with the rest.
#include "launch.hpp"

namespace hgs {

template <int N, int PHASE>
static int launch_fused_rule2_one(dim3 grid, hipStream_t s, const ColArgs<float>& a) {
    constexpr size_t lds = (size_t)ColCfg<N>::CPAR * lds_elems<N>() * sizeof(Cx<float>) + SCRATCH_DOUBLES * sizeof(double);
    auto k = col_fused_kernel<float, N, PHASE, false, 2>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KFused, float, N, PHASE, false, 2, 16>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(ColCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}
template <int N> static int launch_fused_rule2_n(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a) {
    if (phase == 0) return launch_fused_rule2_one<N, 0>(grid, s, a);
    if (phase == 1) return launch_fused_rule2_one<N, 1>(grid, s, a);
    return launch_fused_rule2_one<N, 2>(grid, s, a);
}

int launch_fused_rule2(int N, int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a) {
    switch (N) {
        case 64: return launch_fused_rule2_n<64>(phase, grid, s, a);
        case 128: return launch_fused_rule2_n<128>(phase, grid, s, a);
        case 256: return launch_fused_rule2_n<256>(phase, grid, s, a);
        case 512: return launch_fused_rule2_n<512>(phase, grid, s, a);
        case 1024: return launch_fused_rule2_n<1024>(phase, grid, s, a);
        case 2048: return launch_fused_rule2_n<2048>(phase, grid, s, a);
        case 4096: return launch_fused_rule2_n<4096>(phase, grid, s, a);
        case 8192: return launch_fused_rule2_n<8192>(phase, grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
