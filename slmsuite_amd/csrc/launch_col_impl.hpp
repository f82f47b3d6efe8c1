// Included by launch_col_f32.hip / launch_col_f64.hip with HGS_REAL defined.
#include "launch.hpp"

namespace hgs {

template <typename R, int N, int MODE>
static int launch_col_one(dim3 grid, hipStream_t s, const ColArgs<R>& a) {
    constexpr size_t lds = (size_t)ColCfg<N>::CPAR * lds_elems<N>() * sizeof(Cx<R>) + 16 * sizeof(double);
    auto k = col_kernel<R, N, MODE>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KCol, R, N, MODE>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(ColCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}

template <typename R, int N>
static int launch_col_n(int mode, dim3 grid, hipStream_t s, const ColArgs<R>& a) {
    switch (mode) {
        case (C_FWD | C_STORE): return launch_col_one<R, N, (C_FWD | C_STORE)>(grid, s, a);
        case (C_LOAD | C_INV): return launch_col_one<R, N, (C_LOAD | C_INV)>(grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

template <> int launch_col<HGS_REAL>(int N, int mode, dim3 grid, hipStream_t s, const ColArgs<HGS_REAL>& a) {
    switch (N) {
        case 64: return launch_col_n<HGS_REAL, 64>(mode, grid, s, a);
        case 128: return launch_col_n<HGS_REAL, 128>(mode, grid, s, a);
        case 256: return launch_col_n<HGS_REAL, 256>(mode, grid, s, a);
        case 512: return launch_col_n<HGS_REAL, 512>(mode, grid, s, a);
        case 1024: return launch_col_n<HGS_REAL, 1024>(mode, grid, s, a);
        case 2048: return launch_col_n<HGS_REAL, 2048>(mode, grid, s, a);
        case 4096: return launch_col_n<HGS_REAL, 4096>(mode, grid, s, a);
        case 8192: return launch_col_n<HGS_REAL, 8192>(mode, grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

template <> size_t col_lds_bytes<HGS_REAL>(int N) {
    const int T = N / 16;
    const int CPAR = T >= 256 ? 1 : (256 / T > 4 ? 4 : 256 / T);
    return (size_t)CPAR * (N + N / 16) * sizeof(Cx<HGS_REAL>) + 16 * sizeof(double);
}

}  // namespace hgs
