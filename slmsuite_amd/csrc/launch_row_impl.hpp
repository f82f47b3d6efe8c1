// Included by launch_row_f32.hip / launch_row_f64.hip with HGS_REAL defined.
#include "launch.hpp"
#include <cstdlib>

namespace hgs {

// prefetching form (row_kernel PREF): transform image + the 32 KB image of the next row
template <typename R, int N, int NS>
static int launch_row_pref(dim3 grid, hipStream_t s, const RowArgs<R>& a) {
    const size_t lds = (lds_elems<N>() + N) * sizeof(Cx<R>);
    auto k = row_kernel<R, N, 2, NS, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dispatch_note(dispatch_site<KRow, R, N, 2, NS, true, false>(), row_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(RowCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}

template <typename R, int N, int MODE, int NS = 16>
static int launch_row_one(dim3 grid, hipStream_t s, const RowArgs<R>& a) {
    // The shifted 4096-point kernel needs 117 VGPRs, so four workgroups fit a CU.  Measured (cfg 2 / a batch of eight):
    // a masked launch (active columns only) gains from the fourth -- 20.5 -> 19.0 us, 103 -> 88 us -- a dense one does
    // not (28.2 us either way; batch 174 -> 195 us: the load bursts of 1,024 rows at once).  Asking for 48 KB of LDS
    // keeps a dense launch at three.
    constexpr size_t lds_need = (size_t)RowCfg<N>::FPW * lds_elems<N>() * sizeof(Cx<R>) + (HGS_F64_LTW_ROW ? fused_ltw_bytes<R, N>() : 0);
    const bool masked = a.load_mask != nullptr || a.store_mask != nullptr;
    const size_t lds = (NS < 16 && N == 4096 && !masked && lds_need < 48 * 1024) ? (size_t)48 * 1024 : lds_need;
    auto k = row_kernel<R, N, MODE, NS>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KRow, R, N, MODE, NS, false, false>(), row_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(RowCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}

// shifted form (row_kernel NS = 8): fp32, one-row workgroups, SLM within eight of the sixteen register slots (RowArgs::shifted, m0)
template <typename R, int N>
static int launch_row_n(int mode, dim3 grid, hipStream_t s, const RowArgs<R>& a) {
#ifdef HGS_REAL_IS_FLOAT
    if constexpr (N == 4096) {
        if (a.prefetch && mode == 2 && a.load_mask == nullptr && a.store_mask == nullptr)
            return a.shifted ? launch_row_pref<R, N, 8>(grid, s, a) : launch_row_pref<R, N, 16>(grid, s, a);
    }
    if constexpr (N >= 4096) {
        if (a.shifted) {
            switch (mode) {
                case 0: return launch_row_one<R, N, 0, 8>(grid, s, a);
                case 1: return launch_row_one<R, N, 1, 8>(grid, s, a);
                case 2: return launch_row_one<R, N, 2, 8>(grid, s, a);
                case 3: return launch_row_one<R, N, 3, 8>(grid, s, a);      // (fp32 only: the engine keeps MODE 1 in float64)
            }
        }
    }
#else
    // float64 (round 5): the same shifted form at 4096 / 8192 columns (no MODE 3: the engine keeps MODE 1 in float64)
    if constexpr (N >= 4096) {
        if (a.shifted) {
            switch (mode) {
                case 0: return launch_row_one<R, N, 0, 8>(grid, s, a);
                case 1: return launch_row_one<R, N, 1, 8>(grid, s, a);
                case 2: return launch_row_one<R, N, 2, 8>(grid, s, a);
            }
        }
    }
#endif
    switch (mode) {
        case 0: return launch_row_one<R, N, 0>(grid, s, a);
        case 1: return launch_row_one<R, N, 1>(grid, s, a);
        case 2: return launch_row_one<R, N, 2>(grid, s, a);
#ifdef HGS_REAL_IS_FLOAT
        case 3: return launch_row_one<R, N, 3>(grid, s, a);      // MODE 2 that also writes the phase (last launch of a call)
#endif
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

#ifndef HGS_REAL_IS_FLOAT
// float64 single-pass MRAF (col_fused_kernel with CParams::split + col_kernel<LOAD | INV> into gh2): the row kernel that joins
// the two parts, H = gh * wscale + gh2 (a.gh2 and a.gh2_mask set); rows of 4096 / 8192 (the split form has one-row workgroups)
template <int N, int MODE, int NS = 16>
static int launch_row_split64_one(dim3 grid, hipStream_t s, const RowArgs<double>& a) {
    constexpr size_t lds = (size_t)RowCfg<N>::FPW * lds_elems<N>() * sizeof(Cx<double>) + (HGS_F64_LTW_ROW ? fused_ltw_bytes<double, N>() : 0);
    auto k = row_kernel<double, N, MODE, NS, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dispatch_note(dispatch_site<KRow, double, N, MODE, NS, false, true>(), row_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(RowCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}
template <int N>
static int launch_row_split64_n(int mode, dim3 grid, hipStream_t s, const RowArgs<double>& a) {
    if (a.shifted) return mode == 1 ? launch_row_split64_one<N, 1, 8>(grid, s, a) : launch_row_split64_one<N, 2, 8>(grid, s, a);
    return mode == 1 ? launch_row_split64_one<N, 1>(grid, s, a) : launch_row_split64_one<N, 2>(grid, s, a);
}
int launch_row_split(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<double>& a) {
    if (mode != 1 && mode != 2) return (int)hipErrorInvalidValue;
    if (N == 4096) return launch_row_split64_n<4096>(mode, grid, s, a);
    if (N == 8192) return launch_row_split64_n<8192>(mode, grid, s, a);
    return (int)hipErrorInvalidValue;
}
#endif

template <> size_t row_lds_bytes<HGS_REAL>(int N) {
    const int T = N / 16, WG = T >= 256 ? T : 256;
    return (size_t)(WG / T) * (N + N / 16) * sizeof(Cx<HGS_REAL>) +
           ((HGS_F64_LTW_ROW && sizeof(HGS_REAL) == 8 && (N == 4096 || N == 8192)) ? (size_t)LTW_N * sizeof(Cx<HGS_REAL>) : 0);
}

}  // namespace hgs
