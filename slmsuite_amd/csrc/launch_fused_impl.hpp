// Included by launch_fused_{,stats_}f32.hip / launch_fused_{,stats_}f64.hip with HGS_REAL defined.
// HGS_STATS_TU = 1 builds the variants that also accumulate the statistics (hgs_iterate_stats).
#include "launch.hpp"

#ifndef HGS_STATS_TU
#define HGS_STATS_TU 0
#endif
#ifndef HGS_TILE_EXTRAS_TU
#define HGS_TILE_EXTRAS_TU 0      // 1: this unit holds only the tile-resident kernel WITH the MRAF / Nogrette /
#endif                            //    forward-only branches (launch_tile_extras*)
#if HGS_STATS_TU
#define LAUNCH_FUSED launch_fused_stats
#if HGS_TILE_EXTRAS_TU
#define LAUNCH_TILE launch_tile_extras_stats
#else
#define LAUNCH_TILE launch_tile_stats
#endif
#else
#define LAUNCH_FUSED launch_fused
#if HGS_TILE_EXTRAS_TU
#define LAUNCH_TILE launch_tile_extras
#else
#define LAUNCH_TILE launch_tile
#endif
#endif

namespace hgs {
constexpr bool kStats = HGS_STATS_TU != 0;
constexpr bool kExtras = HGS_TILE_EXTRAS_TU != 0;

#if !HGS_TILE_EXTRAS_TU
template <typename R, int N, int PHASE, int NRS = 16>
static int launch_fused_one(dim3 grid, hipStream_t s, const ColArgs<R>& a) {
    constexpr size_t lds = (size_t)ColCfg<N>::CPAR * lds_elems<N>() * sizeof(Cx<R>) + SCRATCH_DOUBLES * sizeof(double) + fused_ltw_bytes<R, N>();
    auto k = col_fused_kernel<R, N, PHASE, kStats, 0, NRS>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KFused, R, N, PHASE, kStats, 0, NRS>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(ColCfg<N>::WG), lds, s, a);
    return (int)hipGetLastError();
}

template <typename R, int N>
static int launch_fused_n(int phase, dim3 grid, hipStream_t s, const ColArgs<R>& a) {
    // float64 at 4096 / 8192 rows: the shifted form with the transforms pruned to the slots the SLM rows occupy (a.fnr)
    if constexpr (sizeof(R) == 8 && N >= 4096 && !kStats) {
        if (a.fnr > 0 && a.fnr <= 4) {
            switch (phase) {
                case 0: return launch_fused_one<R, N, 0, 4>(grid, s, a);
                case 1: return launch_fused_one<R, N, 1, 4>(grid, s, a);
                case 2: return launch_fused_one<R, N, 2, 4>(grid, s, a);
            }
        } else if (a.fnr > 0 && a.fnr <= 6) {
            switch (phase) {
                case 0: return launch_fused_one<R, N, 0, 6>(grid, s, a);
                case 1: return launch_fused_one<R, N, 1, 6>(grid, s, a);
                case 2: return launch_fused_one<R, N, 2, 6>(grid, s, a);
            }
        }
    }
    switch (phase) {
        case 0: return launch_fused_one<R, N, 0>(grid, s, a);
        case 1: return launch_fused_one<R, N, 1>(grid, s, a);
        case 2: return launch_fused_one<R, N, 2>(grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

template <> int LAUNCH_FUSED<HGS_REAL>(int N, int phase, dim3 grid, hipStream_t s, const ColArgs<HGS_REAL>& a) {
    switch (N) {
        case 64: return launch_fused_n<HGS_REAL, 64>(phase, grid, s, a);
        case 128: return launch_fused_n<HGS_REAL, 128>(phase, grid, s, a);
        case 256: return launch_fused_n<HGS_REAL, 256>(phase, grid, s, a);
        case 512: return launch_fused_n<HGS_REAL, 512>(phase, grid, s, a);
        case 1024: return launch_fused_n<HGS_REAL, 1024>(phase, grid, s, a);
        case 2048: return launch_fused_n<HGS_REAL, 2048>(phase, grid, s, a);
        case 4096: return launch_fused_n<HGS_REAL, 4096>(phase, grid, s, a);
        case 8192: return launch_fused_n<HGS_REAL, 8192>(phase, grid, s, a);
    }
    return (int)hipErrorInvalidValue;
}

#endif   // !HGS_TILE_EXTRAS_TU

#ifdef HGS_REAL_IS_FLOAT
template <int N, int PHASE>
static int launch_tile_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    constexpr size_t lds = col_tile_lds_bytes<float, N>();
    auto k = col_tile_kernel<float, N, PHASE, 6, kStats, kExtras>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KTile, float, N, PHASE, 6, kStats, kExtras, 0, -1>(), col_flags(grid, a));
    hipLaunchKernelGGL(k, grid, dim3(N / 16), lds, s, a, m0);
    return (int)hipGetLastError();
}

template <> int LAUNCH_TILE<float>(int N, int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0) {
    if (N == 4096) {
        if (phase == 0) return launch_tile_one<4096, 0>(grid, s, a, m0);
        if (phase == 1) return launch_tile_one<4096, 1>(grid, s, a, m0);
        return launch_tile_one<4096, 2>(grid, s, a, m0);
    }
    if (N == 8192) {
        if (phase == 0) return launch_tile_one<8192, 0>(grid, s, a, m0);
        if (phase == 1) return launch_tile_one<8192, 1>(grid, s, a, m0);
        return launch_tile_one<8192, 2>(grid, s, a, m0);
    }
    return (int)hipErrorInvalidValue;
}

#else
template <> int LAUNCH_TILE<double>(int, int, dim3, hipStream_t, const ColArgs<double>&, int) {
    return (int)hipErrorInvalidValue;   // the tile-resident kernel is fp32 only
}
#if HGS_STATS_TU
template <> int launch_tile_extras_stats<double>(int, int, dim3, hipStream_t, const ColArgs<double>&, int) {
    return (int)hipErrorInvalidValue;
}
#else
template <> int launch_tile_extras<double>(int, int, dim3, hipStream_t, const ColArgs<double>&, int) {
    return (int)hipErrorInvalidValue;
}
#endif
#endif

}  // namespace hgs
