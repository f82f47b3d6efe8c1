// Launch front-ends for the heavy templated kernels.  The instantiations live in their own
// translation units (launch_row_*.hip / launch_col_*.hip) so hipcc can build them in parallel.
#pragma once
#include "kernels.hpp"
#include "dispatch.hpp"

namespace hgs {

template <typename R> inline unsigned row_flags(dim3 grid, const RowArgs<R>& a) {
    return (a.load_mask ? DF_LOAD_MASK : 0u) | (a.store_mask ? DF_STORE_MASK : 0u) | (grid.y > 1 ? DF_BATCH : 0u) | (a.nf_out ? DF_NF_OUT : 0u);
}
template <typename R> inline unsigned col_flags(dim3 grid, const ColArgs<R>& a) {
    return (a.col_list ? DF_LIST : 0u) | ((a.col_list ? a.list_xmap : a.col_xmap) ? DF_XMAP : 0u) | (grid.y > 1 ? DF_BATCH : 0u) |
           (a.do_stats ? DF_STATS : 0u);
}

// returns hipError_t as int
template <typename R> int launch_row(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<R>& a);
template <typename R> int launch_col(int N, int mode, dim3 grid, hipStream_t s, const ColArgs<R>& a);
template <typename R> int launch_fused(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a);
// tile-resident fused kernel: fp32, N in {4096, 8192}, at most 6 occupied load-layout slots
template <typename R> int launch_tile(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a, int m0);
// the same two kernels with the in-pass statistics compiled in (ColArgs::do_stats, hgs_iterate_stats)
template <typename R> int launch_fused_stats(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a);
template <typename R> int launch_tile_stats(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a, int m0);
// ... and with the MRAF / Nogrette-sum / forward-only branches (launch_tile* compile them out)
template <typename R> int launch_tile_extras(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a, int m0);
template <typename R> int launch_tile_extras_stats(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<R>& a, int m0);

// ... and with the weight rule compiled in (rule 1: WGS-Leonardo / WGS-Kim update, 2: no update; no statistics, no extras)
// (nr: register slots the SLM rows occupy; <= 4 runs the NR = 4 instances)
int launch_tile_rule(int N, int phase_mode, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);
int launch_tile_rule_listed(int N, int phase_mode, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);   // a.col_list set

// half-width tile-resident kernel (col_tile2_kernel): fp32, 4096 rows (batches: three workgroups per CU) and 2048 rows; plain
// rules only (1: WGS-Leonardo / WGS-Kim update, 2: no update); tile2_has: an instance for this slot count exists
bool tile2_has(int N, int nr);
int launch_tile2(int N, int phase_mode, int rule, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int shift, int half_xmap);

// single-pass MRAF with a weight update (col_tile_kernel RULE 3 writes a.gh / a.gh2, row_kernel SPLIT joins them); fp32,
// N in {4096, 8192}
int launch_tile_split(int N, int phase_mode, int nr, int rule_ok, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);
int launch_tile_split_stats(int N, int phase_mode, int nr, int rule_ok, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);   // a.do_stats
int launch_row_split(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<float>& a);
int launch_row_split(int N, int mode, dim3 grid, hipStream_t s, const RowArgs<double>& a);   // float64: rows of 4096 / 8192 (launch_row_f64.hip)

// MRAF with a WGS-Leonardo / WGS-Kim update and ONE inverse per column (round 6; launch_tile_presum_f32.hip): launch_presum
// leaves the partials of D = sum w'^2 - sum w^2 (signal pixels) in a.wpartial, launch_tile_presum (col_tile_kernel RULE 5) reads
// them through a.dpartial and rebuilds the field with 1 / sqrt(1 + D)
int launch_presum(int N, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);      // a.col_flags set
int launch_tile_presum(int N, int phase_mode, int nr, dim3 grid, hipStream_t s, const ColArgs<float>& a, int m0);

// per-column fused kernel with the rule compiled in (fp32, no statistics, none of the MRAF / Nogrette / forward-only extras)
int launch_fused_rule1(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<float>& a);    // Leonardo / Kim update
int launch_fused_rule2(int N, int phase_mode, dim3 grid, hipStream_t s, const ColArgs<float>& a);    // no update

// blocks of the transform kernels resident per CU are bounded by LDS; exposed for grid sizing
template <typename R> size_t row_lds_bytes(int N);
template <typename R> size_t col_lds_bytes(int N);
int row_fpw(int N);   // rows per workgroup pass

}  // namespace hgs
