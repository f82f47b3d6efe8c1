// Included by launch_blue_f32.hip / launch_blue_f64.hip with HGS_REAL defined.
#include "bluestein.hpp"
#include "dispatch.hpp"

namespace hgs {

template <typename R, int M> static int launch_blue_one(dim3 grid, hipStream_t s, const BlueArgs<R>& a) {
    constexpr size_t lds = (size_t)lds_elems<M>() * sizeof(Cx<R>);
    auto k = bluestein_lines<R, M>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KBlue, R, M>(), grid.y > 1 ? DF_BATCH : 0u);
    hipLaunchKernelGGL(k, grid, dim3(M / 16), lds, s, a);
    return (int)hipGetLastError();
}

template <> int launch_bluestein<HGS_REAL>(int M, dim3 grid, hipStream_t s, const BlueArgs<HGS_REAL>& a) {
    switch (M) {
        case 256: return launch_blue_one<HGS_REAL, 256>(grid, s, a);
        case 512: return launch_blue_one<HGS_REAL, 512>(grid, s, a);
        case 1024: return launch_blue_one<HGS_REAL, 1024>(grid, s, a);
        case 2048: return launch_blue_one<HGS_REAL, 2048>(grid, s, a);
        case 4096: return launch_blue_one<HGS_REAL, 4096>(grid, s, a);
        case 8192: return launch_blue_one<HGS_REAL, 8192>(grid, s, a);
#ifdef HGS_REAL_IS_FLOAT
        case 16384: return launch_blue_one<HGS_REAL, 16384>(grid, s, a);     // (fp64: 278 KB of LDS, not available)
#endif
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
