// Included by launch_row_f32.hip / launch_row_f64.hip with HGS_REAL defined.
#include "launch.hpp"

namespace hgs {

template <typename R, int N, int MODE>
static int launch_row_one(dim3 grid, hipStream_t s, const RowArgs<R>& a) {
    constexpr size_t lds = (size_t)RowCfg<N>::FPW * lds_elems<N>() * sizeof(Cx<R>);
    auto k = row_kernel<R, N, MODE>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(k, grid, dim3(RowCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}

template <typename R, int N>
static int launch_row_n(int mode, dim3 grid, hipStream_t s, const RowArgs<R>& a) {
    switch (mode) {
        case 0: return launch_row_one<R, N, 0>(grid, s, a);
        case 1: return launch_row_one<R, N, 1>(grid, s, a);
        case 2: return launch_row_one<R, N, 2>(grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

template <> int launch_row<HGS_REAL>(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<HGS_REAL>& a) {
    switch (N) {
        case 64: return launch_row_n<HGS_REAL, 64>(mode, grid, s, a);
        case 128: return launch_row_n<HGS_REAL, 128>(mode, grid, s, a);
        case 256: return launch_row_n<HGS_REAL, 256>(mode, grid, s, a);
        case 512: return launch_row_n<HGS_REAL, 512>(mode, grid, s, a);
        case 1024: return launch_row_n<HGS_REAL, 1024>(mode, grid, s, a);
        case 2048: return launch_row_n<HGS_REAL, 2048>(mode, grid, s, a);
        case 4096: return launch_row_n<HGS_REAL, 4096>(mode, grid, s, a);
        case 8192: return launch_row_n<HGS_REAL, 8192>(mode, grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

template <> size_t row_lds_bytes<HGS_REAL>(int N) {
    const int T = N / 16, WG = T >= 256 ? T : 256;
    return (size_t)(WG / T) * (N + N / 16) * sizeof(Cx<HGS_REAL>);
}

}  // namespace hgs
