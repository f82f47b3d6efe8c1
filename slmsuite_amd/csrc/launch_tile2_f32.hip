// Half-width tile-resident fused column kernel (col_tile2_kernel): batches at 4096 rows (three workgroups per CU), every
// dense launch at 2048 rows.  Its own translation unit so that hipcc builds it in parallel.
#include "launch.hpp"

namespace hgs {

template <int N, int PHASE, int RULE, int NR, bool PARK = false, bool NXF = false>
static int launch_tile2_one(dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int half_xmap) {
    constexpr size_t lds = col_tile2_lds_bytes<float, N, PARK>();
    auto k = col_tile2_kernel<float, N, PHASE, NR, RULE, PARK, NXF>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    dispatch_note(dispatch_site<KTile2, float, N, PHASE, NR, RULE, PARK, NXF>(), col_flags(grid, a) | (half_xmap ? DF_XMAP : 0u));
    hipLaunchKernelGGL(k, grid, dim3(Tile2Cfg<N>::WG), lds, s, a, shift, half_xmap);
    return (int)hipGetLastError();
}
template <int N, int RULE, int NR>
static int launch_tile2_n(int phase, dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int half_xmap) {
    // 4096 rows, one hologram: the idle column of the half tile parked in LDS (no scratch); batches: both columns in registers
    if constexpr (N >= 4096) {
        if (phase == 0 && grid.y == 1) {
            // few active columns (a spot array with every column forced): the next half tile's rows requested ahead of the stores
            if constexpr (NR <= 5) { if (a.few_active) return launch_tile2_one<N, 0, RULE, NR, true, true>(grid, s, a, shift, half_xmap); }
            return launch_tile2_one<N, 0, RULE, NR, true>(grid, s, a, shift, half_xmap);
        }
    }
    if (phase == 0) return launch_tile2_one<N, 0, RULE, NR>(grid, s, a, shift, half_xmap);
    // 4096 rows: the engine sends only passes that neither store nor read the farfield phase here (the phase-storing / -reading
    // update instances do not fit the 168 registers of three workgroups per CU, tools/resusage.sh) -- they are not compiled
    if constexpr (N >= 4096) {
        // ... except the phase-READING form (WGS-Kim with its phase fixed) for ONE hologram: its update instances run 4 / 8 / 12
        // registers over (parked form) and are still faster than col_tile_kernel's two workgroups per CU -- dense image target
        // 87.5 -> 76.8 us, spot array 53.5 -> 52.1 us; a batch (register form, 18 .. 35 spilled) loses 15 % and is not sent here
        if (phase == 2 && grid.y == 1) return launch_tile2_one<N, 2, RULE, NR, true>(grid, s, a, shift, half_xmap);
        return (int)hipErrorInvalidValue;
    } else {
    if (phase == 1) return launch_tile2_one<N, 1, RULE, NR>(grid, s, a, shift, half_xmap);
    return launch_tile2_one<N, 2, RULE, NR>(grid, s, a, shift, half_xmap);
    }
}
template <int N, int NR>
static int launch_tile2_r(int phase, int rule, dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int half_xmap) {
    return rule == 1 ? launch_tile2_n<N, 1, NR>(phase, grid, s, a, shift, half_xmap) : launch_tile2_n<N, 2, NR>(phase, grid, s, a, shift, half_xmap);
}

// the slot counts compiled: 4096 rows 4 .. 6 (as col_tile_kernel), 2048 rows 8 .. 10 (SLMs of 1024 .. 1265 rows)
bool tile2_has(int N, int nr) {
    if (N == 4096) return nr >= 1 && nr <= 6;
    if (N == 2048) return nr >= 1 && nr <= 10;
    return false;
}

// rule: 1 = WGS-Leonardo / WGS-Kim update, 2 = no update; nr: register slots the SLM rows occupy; shift: rows (multiple of 16)
int launch_tile2(int N, int phase, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int half_xmap) {
    if (N == 4096) {
        if (nr <= 4) return launch_tile2_r<4096, 4>(phase, rule, grid, s, a, shift, half_xmap);
        if (nr == 5) return launch_tile2_r<4096, 5>(phase, rule, grid, s, a, shift, half_xmap);
        if (nr == 6) return launch_tile2_r<4096, 6>(phase, rule, grid, s, a, shift, half_xmap);
    }
    if (N == 2048) {
        if (nr <= 8) return launch_tile2_r<2048, 8>(phase, rule, grid, s, a, shift, half_xmap);
        if (nr == 9) return launch_tile2_r<2048, 9>(phase, rule, grid, s, a, shift, half_xmap);
        if (nr == 10) return launch_tile2_r<2048, 10>(phase, rule, grid, s, a, shift, half_xmap);
    }
    return (int)hipErrorInvalidValue;
}

}  // namespace hgs
