// Device kernels of the hologram engine (gfx950).  See DESIGN.md for the data layout and roofline
// of each kernel.  Reference semantics cited per kernel (paths relative to the slmsuite checkout).
//
// Layouts in HBM (per hologram b of the batch; R = real type, C = complex of R):
//   phase / amp / kernel : [Sh][Sw] R, natural (the reference's arrays).
//   GH  : the half-transformed field, ONLY the Sh SLM rows (all other rows of the zero-padded
//         nearfield are zero, so their row transforms are never computed or stored):
//         tile-of-4-columns layout  GH[ct][r][c4],  kx = 4*ct + c4   (C, Sh*Pw elements)
//         -> the column kernel reads/writes one contiguous Sh*32-byte block per tile,
//            the row kernel touches 32-byte pieces that adjacent rows complete to full lines.
//   weights / target / phase_ff / farfield / amp_ff : COLUMN-major [kx][ky] (P elements), so the
//         column kernels stream them fully coalesced.  hgs_set_array/hgs_get_array transpose.
//
// Centred transforms (fftshift . fft2 . fftshift with norm="ortho", _hologram.py:1048) are folded
// into sign flips: C[k'] = N^-1/2 (-1)^k' FFT[(-1)^j' x[j']]  (valid for N % 4 == 0), so no shift
// pass and no index rotation exists; arrays always hold the reference's centred values.
#pragma once
#include "fft_core.hpp"

// minimum waves per SIMD the transform kernels are register-allocated for (tunable at build time)
#ifndef HGS_COL_OCC_8192
#define HGS_COL_OCC_8192 4
#endif
#ifndef HGS_ROW_OCC_8192
#define HGS_ROW_OCC_8192 4
#endif
#ifndef HGS_ROW_OCC
#define HGS_ROW_OCC 3
#endif
#ifndef HGS_COL_OCC
#define HGS_COL_OCC 3
#endif
#ifndef HGS_ROW_TW_RESIDENT
#define HGS_ROW_TW_RESIDENT true
#endif
#ifndef HGS_COL_BUF
#define HGS_COL_BUF 1        // fused column kernel: G / H columns through a buffer resource (see Buf)
#endif
#ifndef HGS_ROW_BUF
#define HGS_ROW_BUF 1        // row kernel: raw buffer accesses, straight-line (rows whose lane group is a whole number of waves)
#endif
#ifndef HGS_ROW_PHASOR
#define HGS_ROW_PHASOR 1     // MODE 2 row kernel: nf/|nf| instead of atan2 + sincos
#endif
#ifndef HGS_ROW_PREF_NOMASK
#define HGS_ROW_PREF_NOMASK 0  // 1: ... or keeps its flat stores but drops the mask test (one block of sixteen stores): 26.0 -> 27.9 us -- slower too:
#endif                         //    the per-store blocks keep the stores spread between the last butterflies instead of bunched behind them
#ifndef HGS_ROW_PREF_BUFST
#define HGS_ROW_PREF_BUFST 0   // 1: the prefetching row walk stores G through a buffer resource, straight-line (no mask test, no 64-bit address
#endif                         //    per store): row launch 26.0 -> 28.1 us, headline 15.17 k -> 14.65 k it/s -- slower (round 6)
#ifndef HGS_LIST_SLOAD
#define HGS_LIST_SLOAD 1
#endif
#ifndef HGS_ROW_AMP_PREFETCH
#define HGS_ROW_AMP_PREFETCH 1
#endif
#ifndef HGS_SPARSE_SKIP
#define HGS_SPARSE_SKIP 1
#endif
#ifndef HGS_FUSED_OCC
#define HGS_FUSED_OCC 2
#endif
// Ablation hooks for tools/microbench/ablate.hip (all 0 in the product build): compile the weight/target
// loads (WT) or the GH tile loads and stores (GH) out of col_tile_kernel to see what they cost.
// Round-5 experiments on the 8192-row column kernels, all measured and NOT adopted (NOTEBOOK.md; A/B builds through
// tools/microbench/build_trace8k.sh): the s_memtime trace of a build that drains vmcnt at every phase boundary suggested
// memory waits that the product build does not have.
#ifndef HGS_ROW_AMP_LATE
#define HGS_ROW_AMP_LATE 0      // row_kernel PREF: 1 = the amplitude requests behind the pick-up of the staged row (measured, below)
#endif
#ifndef HGS_TILE_TOUCH
#define HGS_TILE_TOUCH 1        // col_tile_kernel, 8192 rows: the wait for a column's weights / targets sits at the END of the previous column
#endif                          // (an empty asm that names their registers), not at its head -- see the kernel
#ifndef HGS_TILE_STAGE_WAIT
#define HGS_TILE_STAGE_WAIT 1   // 0: no vmcnt(0) ahead of reading the staged tile (the loads have landed by then): 213.9 vs 214.5 us, noise
#endif
#ifndef HGS_F64_WT_PREFETCH
#define HGS_F64_WT_PREFETCH 0   // 1: float64 8192-point fused kernel, one word of each weight / target line requested ahead of the forward
#endif                          //    transform (L2 prefetch): 737 vs 713 us -- slower
#ifndef HGS_PF_AHEAD
#define HGS_PF_AHEAD 1          // per-column kernel, fp32, fixed farfield phase: the stored phase fetched a column ahead with the weights
#endif
#ifndef HGS_F64_WT_EARLY
#define HGS_F64_WT_EARLY 0      // float64 8192-point fused kernel, shifted form: weights / targets requested ahead of the (pruned) forward transform
#endif
#ifndef HGS_SPLIT_L2_PREFETCH
#define HGS_SPLIT_L2_PREFETCH 0 // 1: single-pass MRAF tile kernel, the next tile's rows requested ahead of the first transform: 332.6 vs 330.4 us
#endif
#ifndef HGS_F64_POW_LEAN
#define HGS_F64_POW_LEAN 1      // float64 rule kernels: pow_lean / Newton rsqrt instead of ocml's log2 + exp2 / sqrt + division (round 6)
#endif
#ifndef HGS_F64_PARK_AHEAD
#define HGS_F64_PARK_AHEAD 0    // 1: float64 fused kernel, the parked value of pixel m + 1 requested before pixel m is evaluated: 620 vs 599 us (slower; round 6)
#endif
#ifndef HGS_ABL_WT
#define HGS_ABL_WT 0
#endif
#ifndef HGS_ABL_GH
#define HGS_ABL_GH 0
#endif

namespace hgs {

#ifndef HGS_F64_LTW
#define HGS_F64_LTW 1
#endif
#ifndef HGS_F64_WT_BUF
#define HGS_F64_WT_BUF 1
#endif
#ifndef HGS_F64_LTW_ROW
#define HGS_F64_LTW_ROW 0       // 1: the float64 row kernel's stage twiddles from the LDS tables too: 106.3 vs 106.4 us at cfg 5, nothing (round 6)
#endif
// float64 transform kernels at 4096 / 8192 points (col_fused_kernel, row_kernel): stage twiddles from the two LDS tables of
// WgFftL LTW, and the dynamic LDS they add behind the kernel's other LDS
template <typename R, int N> constexpr bool fused_ltw() { return HGS_F64_LTW && sizeof(R) == 8 && (N == 4096 || N == 8192); }
template <typename R, int N> constexpr size_t fused_ltw_bytes() { return fused_ltw<R, N>() ? (size_t)LTW_N * sizeof(Cx<R>) : 0; }

struct Geo {
    int Ph, Pw, Sh, Sw, r0, c0, batch;
    int lane_T;     // Ph / 16 when the columns of the farfield-sized arrays are stored lane-major (below), 0 = natural
};

// Inside each column of the column-major farfield-sized arrays the Ph entries are stored
// LANE-MAJOR: ky = j + m*T (T = Ph/16, the load layout of the column transform) sits at j*16 + m,
// so the 16 values a lane needs are 64 contiguous bytes (4 x 16-byte loads instead of 16 scalar
// ones, 4 KiB contiguous per wave).  Elementwise kernels are oblivious to the permutation;
// hgs_set_array / hgs_get_array and the spot kernels apply it.
// Tile-resident column kernel: pick the column of the pass out of the tile registers with a register-relative move
// (the pass index is uniform: s_set_gpr_idx + v_mov, 24 instructions per pass) instead of 84 v_cndmask
#ifndef HGS_ROW_LD16
#define HGS_ROW_LD16 1      // ... and load H in 16-byte pieces
#endif
#ifndef HGS_ROW_ST16
#define HGS_ROW_ST16 1      // dense fp32 row launches store G in 16-byte pieces (lane pairs swap one value per slot pair)
#endif
#ifndef HGS_TILE_MOVREL
#define HGS_TILE_MOVREL 1
#endif
#ifndef HGS_CONS_GROUP
#define HGS_CONS_GROUP 16    // pixels of a lane whose rule evaluation the scheduler may interleave (fp32; fp64: 4)
#endif
#ifndef HGS_TRACE_CONS
#define HGS_TRACE_CONS 0      // traced builds (tools/microbench/trace8k): one event per pixel of the float64 constraint
#endif
#ifndef HGS_F64_EAGER
#define HGS_F64_EAGER 0       // 1: float64 rule and phasor evaluated for every lane and selected, as in float32, so that HGS_CONS_GROUP_F64
                              //    pixels form one scheduling region (also with -amdgpu-sched-strategy=max-ilp): 543 vs 543 us, nothing (round 6)
#endif
#ifndef HGS_CONS_GROUP_F64
#define HGS_CONS_GROUP_F64 1  // float64: one pixel at a time (four interleaved double atan2 / sincos / log2 chains cost 100+ registers)
#endif
#ifndef HGS_LANE_MAJOR
#define HGS_LANE_MAJOR 1
#endif
__host__ __device__ __forceinline__ int col_pos(int ky, int T) {
    return (HGS_LANE_MAJOR && T > 0) ? (ky % T) * 16 + ky / T : ky;
}
template <int T> __device__ __forceinline__ unsigned lane_pos(int j, int m) {
    return HGS_LANE_MAJOR ? (unsigned)(j * 16 + m) : (unsigned)(j + m * T);
}

// method codes follow ALGORITHM_INDEX (_header.py:72)
enum { M_GS = 0, M_LEONARDO = 1, M_KIM = 2, M_NOGRETTE = 3, M_WU = 4, M_TANH = 5 };

template <typename R> struct CParams {
    int method;
    int do_update;    // WGS and iter > 0 (_hologram.py:1552)
    int use_fixed;    // rebuild with stored phase_ff (Kim fixed phase, :1601)
    int store_phase;  // write phase_ff = atan2(F) (:1583, :1602)
    int mraf;         // target holds NaN (noise) / 0 (zero) regions (:1606-1653)
    int has_mraf_factor;
    int zero_mode;    // 0: zero region := 0 ; 1: zero_weights feedback (:1613-1616)
    int nog_pass;     // fused kernels: only accumulate sum(fc) of the Nogrette rule (nanmean, :1851) into wpartial
    const R* nog;     // [batch] -1/mean(fc) from that pass (WGS-Nogrette), else nullptr
    int weights_only; // fused kernels: forward transform + weight update (+ statistics) only, no inverse.
                      // MRAF mixes the normalised weights with the un-weighted noise region, so ||w'|| must
                      // be known before the field is rebuilt: pass 1 updates the weights, pass 2 rebuilds.
    R p_exp, p_fac, mraf_factor, zero_factor;
    R inv_fnorm;      // 1/||amp_ff|| (Parseval constant ||amp|| in the fused path)
    R log2_inv_fnorm;
    int presum;       // fused kernels, with weights_only: the forward-only PRE-PASS of the single-inverse MRAF update (round 6) -- the
                      // rule is evaluated as the update would, nothing is written, and wpartial receives D = sum w'^2 - sum w^2
                      // (the main pass then finds it through ColArgs::dpartial and rebuilds with 1 / sqrt(1 + D))
    int split;        // col_fused_kernel: MRAF with a weight update in ONE pass -- the signal part (un-normalised new weights)
                      // is transformed back here, the noise part mraf_factor * F leaves as farfield values through
                      // ColArgs::ffb (only the pixels with a NaN target are written: the rest of the buffer stays zero) and a
                      // col_kernel<LOAD | INV> launch over the columns that hold noise transforms it into gh2; the row kernel
                      // (SPLIT) joins the two once ||w'|| is known.  The float64 counterpart of col_tile_kernel RULE 3.
};

// sin/cos in fp32 with the range reduction by pi/2 done in ONE fp64 fma (exact enough for any
// fp32 argument up to ~1e7 rad, no large-argument slow path: ocml's sincosf carries a Payne-Hanek
// branch that costs >100 VGPRs when unrolled 16x), then minimax polynomials on [-pi/4, pi/4];
// max abs error 9e-8 (tools/check_sincos.py).
__device__ __forceinline__ void sincos_bounded(float x, float* s, float* c) {
    const double xd = (double)x;
    const double qd = rint(xd * 0.63661977236758134308);
    const float r = (float)fma(-qd, 1.57079632679489661923, xd);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          z * z, fmaf(-0.5f, z, 1.0f));
    const int n = (int)qd;
    const float ss = (n & 1) ? pc : ps;
    const float cc = (n & 1) ? ps : pc;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}

// sin/cos of a stored farfield phase (|x| <= ~pi for anything atan2 produced): the reduction by pi/2 in
// fp32 with a two-term Cody-Waite constant is exact to 1e-7 up to |x| ~ 1e3; larger arguments (a
// user-supplied phase_ff) take the fp64 reduction.  Same polynomials, no fp64 ops on the hot path.
__device__ __forceinline__ void sincos_phase(float x, float* s, float* c) {
    if (__builtin_amdgcn_ballot_w64(fabsf(x) > 1.0e3f) != 0) {   // wave-uniform, never taken for atan2 output
        sincos_bounded(x, s, c);
        return;
    }
    const float q = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-q, 1.5707963705062866f, x);            // pi/2 rounded to fp32 ...
    r = fmaf(-q, -4.371138828673793e-8f, r);               // ... and its remainder
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          z * z, fmaf(-0.5f, z, 1.0f));
    const int n = (int)q;
    const float ss = (n & 1) ? pc : ps;
    const float cc = (n & 1) ? ps : pc;
    *s = (n & 2) ? -ss : ss;
    *c = ((n + 1) & 2) ? -cc : cc;
}

template <typename R> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ void sincos_phase(float a, float* s, float* c) { hgs::sincos_phase(a, s, c); }
    static __device__ __forceinline__ void sincos(float a, float* s, float* c) { sincos_bounded(a, s, c); }
    static __device__ __forceinline__ float atan2(float y, float x) { return atan2f(y, x); }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return rsqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float powneg(float x, float p);      // x^-p, x >= 0 (pow_split below)
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float log2(float x) { return log2f(x); }
    static __device__ __forceinline__ float exp2(float x) { return exp2f(x); }
    // bare v_log_f32 / v_exp_f32 (1 ulp each, no denormal rescaling: 4 instead of 19 instructions for x^p); for
    // arguments in the normal range only -- the weight rule's ratio (|F| c / T)^2 near 1
    static __device__ __forceinline__ float log2_fast(float x) { return __builtin_amdgcn_logf(x); }
    static __device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float tanh(float x) { return tanhf(x); }
    static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
};
template <> struct Math<double> {
    static __device__ __forceinline__ void sincos(double a, double* s, double* c) { ::sincos(a, s, c); }
    static __device__ __forceinline__ void sincos_phase(double a, double* s, double* c) { ::sincos(a, s, c); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double powneg(double x, double p) { return ::pow(x, -p); }
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log2(double x) { return ::log2(x); }
    static __device__ __forceinline__ double exp2(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double log2_fast(double x) { return ::log2(x); }
    static __device__ __forceinline__ double exp2_fast(double x) { return ::exp2(x); }
    static __device__ __forceinline__ double tanh(double x) { return ::tanh(x); }
    static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
};

template <typename R> __device__ __forceinline__ bool is_nan(R x) { return x != x; }
template <typename R> __device__ __forceinline__ bool is_pinf(R x) { return x == (R)INFINITY; }

// ---- block reduction of a double (sum) into partial[slot]; all lanes call ---------------------------
__device__ __forceinline__ double wave_sum(double v) {
    const int lane = threadIdx.x & 63, nl = min(64, (int)blockDim.x - ((int)threadIdx.x & ~63));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, 64);
        if (lane + o < nl) v += u;
    }
    return v;
}
// scratch: at least 16 doubles of LDS not in use by anyone else at the call
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x < 64) {
        t = (lane < nw) ? scratch[lane] : 0.0;
        t = wave_sum(t);
    }
    return t;  // valid in thread 0
}

// ---- in-kernel "computational" statistics (_stats.py:7-116) for the fused column kernels -------------
// The fused kernels see every farfield pixel F and its target T in registers, so the statistics the
// reference computes from amp_ff / target in _update_stats are folded into the same pass:
//   f_pwr = |F|^2 / s (s = sum |F|^2 = ||amp||^2 by Parseval),  t_pwr = T^2 / sum T^2,
//   over T != 0:  ratio = f_pwr / t_pwr (min, max), err = t_pwr - f_pwr (min, max, sum, sum^2, count),
//   sum T |F| (efficiency).  Per-lane accumulation over one column, wave reduction, then lane 0 of
// each wave folds into that wave's private LDS slots (no barrier); slots go to global at kernel end.
constexpr int STAT_N = 8;            // tf, es, es2, cnt, rmin, rmax, emin, emax
constexpr int STAT_WAVES = 16;       // slots per workgroup in the partial buffer (max 1024 lanes)
constexpr int SCRATCH_DOUBLES = 16 + STAT_WAVES * STAT_N + 2;     // (+ 2: the last slot holds the pre-summed 1 / ||w'|| of col_fused_kernel)

// (workgroups of the small transforms have fewer than 64 lanes: values shuffled in from lanes that do
// not exist are ignored)
__device__ __forceinline__ int wave_lanes() {
    const int base = (int)threadIdx.x & ~63;
    return min(64, (int)blockDim.x - base);
}
__device__ __forceinline__ double wave_min(double v) {
    const int lane = threadIdx.x & 63, nl = wave_lanes();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, 64);
        if (lane + o < nl) v = fmin(v, u);
    }
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    const int lane = threadIdx.x & 63, nl = wave_lanes();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double u = __shfl_down(v, o, 64);
        if (lane + o < nl) v = fmax(v, u);
    }
    return v;
}

template <typename R> struct StatAcc {
    double tf, es, es2;
    float cnt;
    R rmin, rmax, emin, emax;            // extrema in the working precision
    __device__ __forceinline__ void clear() {
        tf = es = es2 = 0;
        cnt = 0;
        rmin = emin = INFINITY;
        rmax = emax = -INFINITY;
    }
    // one pixel of the mask (T != 0 and not NaN)
    __device__ __forceinline__ void add(R p2, R absF, R t, double at, double bf) {
        if (t != (R)0 && t == t) {                     // mask: T != 0 and not NaN (MRAF noise region)
            const double tp = (double)t * (double)t * at, fp = (double)p2 * bf;
            if (tp != 0.0) {
                const double ratio = fp / tp, err = tp - fp;
                tf += (double)t * (double)absF;
                es += err;
                es2 += err * err;
                cnt += 1.0f;
                rmin = (R)fmin((double)rmin, ratio);
                rmax = (R)fmax((double)rmax, ratio);
                emin = (R)fmin((double)emin, err);
                emax = (R)fmax((double)emax, err);
            }
        }
    }
    static __device__ __forceinline__ void slot_init(double* slot) {
        if ((threadIdx.x & 63) == 0) {
            slot[0] = slot[1] = slot[2] = slot[3] = 0;
            slot[4] = slot[6] = INFINITY;
            slot[5] = slot[7] = -INFINITY;
        }
    }
    // all lanes of the wave call; `slot` = this wave's STAT_N doubles of LDS
    __device__ __forceinline__ void flush(double* slot) {
        const double a0 = wave_sum(tf), a1 = wave_sum(es), a2 = wave_sum(es2), a3 = wave_sum((double)cnt);
        const double a4 = wave_min((double)rmin), a5 = wave_max((double)rmax), a6 = wave_min((double)emin),
                     a7 = wave_max((double)emax);
        if ((threadIdx.x & 63) == 0) {
            slot[0] += a0; slot[1] += a1; slot[2] += a2; slot[3] += a3;
            slot[4] = fmin(slot[4], a4); slot[5] = fmax(slot[5], a5);
            slot[6] = fmin(slot[6], a6); slot[7] = fmax(slot[7], a7);
        }
        clear();
    }
    static __device__ __forceinline__ void slot_store(const double* slot, double* spartial, int b) {
        if ((threadIdx.x & 63) == 0) {
            double* o = spartial + (((size_t)b * gridDim.x + blockIdx.x) * STAT_WAVES + (threadIdx.x >> 6)) * STAT_N;
#pragma unroll
            for (int k = 0; k < STAT_N; ++k) o[k] = slot[k];
        }
    }
};

// x^c for finite x > 0 with the accuracy of one rounding of each hardware transcendental (about 1.5 ulp), independent of
// how far x is from 1.  exp2(c * log2(x)) on v_log_f32 / v_exp_f32 alone loses |log2 x| * |c| ulps: the logarithm's one
// ulp is relative to ITS magnitude, and a speckle pixel twenty octaves under its target paid five ulps (NumPy's powf is
// correctly rounded).  Here x = m 2^e, m in [0.5, 1): the exponent is exact, the hardware logarithm
// is taken of the mantissa only (|log2 m| <= 1: absolute error 6e-8), and c * e is split into the nearest integer n (goes
// into the result's exponent by v_ldexp) and a residual formed by ONE fused multiply-add:
//     x^c = 2^n * exp2((c e - n) + c log2 m),   |argument| <= 0.5 + |c|.
// x = 0 gives +-inf like the plain form (callers map it, :1867); NaN propagates.  9 more VALU instructions per value.
__device__ __forceinline__ float pow_split(float x, float c) {
    const float m = __builtin_amdgcn_frexp_mantf(x);
    const float e = (float)__builtin_amdgcn_frexp_expf(x);
    const float l = __builtin_amdgcn_logf(m);
    const float n = __builtin_rintf(c * e);
    const float f = __builtin_fmaf(c, e, -n) + c * l;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

__device__ __forceinline__ float Math<float>::powneg(float x, float p) { return pow_split(x, -p); }

// The same power in double for the float64 rule kernels (round 6).  ocml's log2 + exp2 behind ::pow / exp2(c log2 x) are some
// three hundred double instructions per pixel with their special-case ladders -- at the half-rate float64 pipe the rule was
// 22 % of a float64 column (profiles/r05/trace8k_timeline.txt: constraint 13.9 k of 47 k cycles).  Here, for finite x > 0:
// x = m 2^e with m in [sqrt(1/2), sqrt(2)), ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.172: ten terms; log2 m in two
// pieces (the product with log2(e) split high / low); c e split into its nearest integer and an fma residual as in
// pow_split; 2^g for |g| <= 1/2 as a degree-13 polynomial of g ln 2.  About 45 double operations, no table, no branch;
// against powl over x in 2^[-80, 80] and c in [-2, -0.35]: at most 2.9 ulp, 0.16 ulp on average (tools/microbench/pow_rule64).
// x = 0 gives +inf for c < 0 like the library form (callers map it, :1867); NaN propagates; x = +inf is the caller's.
__device__ __forceinline__ double pow_lean(double x, double c) {
    double m = __builtin_amdgcn_frexp_mant(x);                 // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m * 2.0 : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 21;
    p = __builtin_fma(p, z, 1.0 / 19); p = __builtin_fma(p, z, 1.0 / 17); p = __builtin_fma(p, z, 1.0 / 15);
    p = __builtin_fma(p, z, 1.0 / 13); p = __builtin_fma(p, z, 1.0 / 11); p = __builtin_fma(p, z, 1.0 / 9);
    p = __builtin_fma(p, z, 1.0 / 7); p = __builtin_fma(p, z, 1.0 / 5); p = __builtin_fma(p, z, 1.0 / 3);
    constexpr double L2E_HI = 1.4426950408889634, L2E_LO = 2.0355273740931033e-17;
    const double t2 = 2.0 * s, lm = t2 * z * p;
    const double hi = t2 * L2E_HI;
    const double lo = __builtin_fma(t2, L2E_HI, -hi) + t2 * L2E_LO + lm * L2E_HI;
    const double ed = (double)e;
    const double n = __builtin_rint(c * ed);
    double f = __builtin_fma(c, ed, -n) + c * hi;
    f += c * lo;
    const double k = __builtin_rint(f);
    const double g = (f - k) * 0.69314718055994530942;
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, g, 1.0 / 479001600.0); q = __builtin_fma(q, g, 1.0 / 39916800.0); q = __builtin_fma(q, g, 1.0 / 3628800.0);
    q = __builtin_fma(q, g, 1.0 / 362880.0); q = __builtin_fma(q, g, 1.0 / 40320.0); q = __builtin_fma(q, g, 1.0 / 5040.0);
    q = __builtin_fma(q, g, 1.0 / 720.0); q = __builtin_fma(q, g, 1.0 / 120.0); q = __builtin_fma(q, g, 1.0 / 24.0);
    q = __builtin_fma(q, g, 1.0 / 6.0); q = __builtin_fma(q, g, 0.5); q = __builtin_fma(q, g, 1.0); q = __builtin_fma(q, g, 1.0);
    const double r = __builtin_amdgcn_ldexp(q, (int)(n + k));
    return x == 0.0 ? (c < 0 ? (double)INFINITY : 0.0) : r;
}

// (|F| c / T)^-p for the Leonardo / Kim rule of the fused kernels, from |F|^2: the ratio is formed FIRST (log2 of the
// three factors separately cancels ~20 against ~20 and leaves 1e-6 relative noise per update).
template <typename R> __device__ __forceinline__ R leonardo_factor(R p2, R t, R inv_fnorm, R p_exp) {
    using M = Math<R>;
    const R q = inv_fnorm * Math<R>::rcp(t);       // 1-ulp reciprocal: the ratio is squared and raised to p/2 < 1/2
    const R r2 = p2 * q * q;                       // (|F| c / T)^2
    if (!(r2 < (R)INFINITY)) return (R)1;          // overflow of the ratio (:1840) and NaN targets (:1843) -> 1
    // r2 = 0 -> inf: callers map it to 1 (:1867)
    if constexpr (sizeof(R) == 4) return pow_split(r2, -0.5f * p_exp);
#if HGS_F64_POW_LEAN
    else return pow_lean(r2, (R)-0.5 * p_exp);
#else
    else return M::exp2_fast((R)-0.5 * p_exp * M::log2_fast(r2));
#endif
}

// ---- the WGS weight rule for one element (rows 9-10; _hologram.py:1830-1873) -------------------------
//   fb  : feedback amplitude already divided by its L2 norm
//   returns the multiplicative factor fc
// fc = feedback / target of the multiplicative rules with its fix-ups (:1837-1843)
template <typename R> __device__ __forceinline__ R nogrette_fc(R fb, R t) {
    R fc = fb / t;
    if (is_pinf(fc)) fc = 1;
    if (t == (R)0) fc = 1;
    if (is_nan(fc)) fc = 1;
    return fc;
}
template <typename R>
__device__ __forceinline__ R weight_factor(int method, R fb, R t, R p_exp, R p_fac, R nog_neg_inv_mean) {
    using M = Math<R>;
    R fc;
    if (method == M_WU || method == M_TANH) {
        fc = fb * (-p_exp) + t;                       // :1834-1835
        if (method == M_WU) fc = M::exp(p_exp * fc);  // :1857
        else fc = p_fac * M::tanh(p_exp * fc) + (R)1; // :1859-1860
    } else {
        fc = fb / t;                                  // :1837
        if (is_pinf(fc)) fc = 1;                      // :1840
        if (t == (R)0) fc = 1;                        // :1841
        if (is_nan(fc)) fc = 1;                       // :1843
        if (method == M_NOGRETTE) {                   // :1851-1855
            fc = fc * nog_neg_inv_mean + (R)1;
            fc = fc * (-p_fac) + (R)1;
            fc = (R)1 / fc;
        } else {
            fc = M::powneg(fc, p_exp);                // :1848
        }
    }
    if (is_pinf(fc)) fc = 1;                          // :1867
    return fc;
}

// ---- raw buffer accesses ------------------------------------------------------------------------------
// One resource per array (row): the lane part of an address is ONE VGPR offset for all the registers of a lane,
// the register part an SGPR offset, and the range check of the resource does the predication -- an element
// outside [0, bytes) reads as zero and its store is dropped -- so a row of predicated accesses is straight-line
// code instead of one exec-mask branch per element.  gfx9 range-checks the VGPR offset + immediate only (NOT the
// SGPR offset): whatever has to be checked goes into the VGPR offset.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0xf0000000u;   // a VGPR offset past every resource
struct Buf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ Buf(const void* p, unsigned bytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    template <typename V> __device__ __forceinline__ V ld(unsigned voff, unsigned soff) const {
        if constexpr (sizeof(V) == 4) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        else if constexpr (sizeof(V) == 8) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    }
    template <typename V> __device__ __forceinline__ void st(V x, unsigned voff, unsigned soff) const {
        if constexpr (sizeof(V) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, voff, soff, 0);
        else if constexpr (sizeof(V) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, x), r, voff, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_t, x), r, voff, soff, 0);
    }
};
// one int at a wave-uniform address through the scalar cache (memory the kernel itself does not write)
__device__ __forceinline__ int uniform_load_i32(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
// 1 / sqrt(x) from the hardware instruction (1 ulp), pre-scaled where x is too small for it (v_rsq_f32 takes
// denormal inputs as zero): the values of rsqrtf() without its inlined control flow
__device__ __forceinline__ float rsqrt_full(float x) {
    const bool tiny = x < 0x1p-100f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 0x1p+100f : x);
    return tiny ? r * 0x1p+50f : r;
}
// float64: the hardware estimate and two Newton steps (x > 0; a sqrt and a division in double are some fifty instructions)
__device__ __forceinline__ double rsqrt_full(double x) {
#if HGS_F64_POW_LEAN
    const bool tiny = x < 0x1p-900;
    const double xs = tiny ? x * 0x1p+200 : x;
    double y = __builtin_amdgcn_rsq(xs);
    const double h = 0.5 * xs;
    y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
    return tiny ? y * 0x1p+100 : y;
#else
    return 1.0 / ::sqrt(x);
#endif
}

// =====================================================================================================
// ROW kernels: transforms along x over the Sh SLM rows.
//   MODE 0 : phase -> G            (_build_nearfield :1000 + row half of fft2 :1048)
//   MODE 1 : H -> phase            (row half of ifft2 :1070 + _nearfield_extract :1026)
//   MODE 2 : H -> phase -> G       (MODE 1 then MODE 0 of the next iteration, fused: the row never
//                                   leaves the CU between the two iterations)
//   MODE 3 : MODE 2 that also writes the phase as MODE 1 does: the LAST launch of a float32 hgs_iterate call
//            leaves G of EVERY column behind (it ignores the store mask), so that the next call -- or
//            hgs_nearfield2farfield -- on an unchanged phase starts with its column launch (round 5).  The G it
//            leaves is MODE 2's, bit for bit: a loop cut into several calls walks exactly like one call.  (A
//            first form rebuilt G from the WRITTEN phase -- MODE 1 + MODE 0 in one launch -- to stay bit-identical
//            to a call that cannot reuse it; its atan2 + sincos made it as slow as those two launches together,
//            40.8 us against 28 for MODE 2, i.e. nothing was saved.  Every column is stored so that ALL paths --
//            column lists too -- can always reuse it and therefore agree with each other.)
// grid = (<= ceil(Sh / FPW), batch), block = WG;  FPW = WG / T rows per workgroup pass; a workgroup
// strides over rows so the per-lane twiddle registers are fetched once per kernel.
// =====================================================================================================
template <int N> struct RowCfg {
    static constexpr int T = N / 16;
    static constexpr int WG = T >= 256 ? T : 256;
    static constexpr int FPW = WG / T;
};

template <typename R> struct RowArgs {
    Geo g;
    R* phase;            // [b][Sh][Sw]
    const R* amp;        // [Sh][Sw] or nullptr (shared by the batch)
    const R* kern;       // [Sh][Sw] or nullptr
    R amp_scalar;
    Cx<R>* gh;           // [b][Pw/4][Sh][4]
    const Cx<R>* tw;     // W_Pw table
    R scale;             // 1/sqrt(Pw)
    // weight-norm finalisation carried by block (0, b): wscale[b] = 1/sqrt(sum partial[b][:])
    const double* wpartial;
    int n_wpartial;
    R* wscale;
    int xcd_map;         // rows 4q..4q+3 (which share 128-B lines of GH) on one XCD at the same time
    int n_row_blocks;    // workgroups that own rows (the grid may hold one more, row-less, for the weight norm)
    int shifted, m0;     // shifted form (row_kernel NS < 16): on / register slot of the first SLM column, c0 / (Pw / 16)
    int prefetch;        // row_kernel PREF: workgroups walk several rows each, the next row's H on its way into LDS
    // sparse targets: which columns the column kernel of this iteration wrote / the next one will read
    const unsigned short* load_mask;    // [b][Pw/16]: bit m of entry j = column j + m*Pw/16 is to be read ...
    const unsigned short* store_mask;   // ... / written; nullptr = every column
    Cx<R>* nf_out;       // MODE 1 only: store the complex nearfield rows [b][Sh][Sw] instead of extracting
                         // the phase (_farfield2nearfield(extract=False), MultiplaneHologram)
    const Cx<R>* gh2;    // row_kernel SPLIT: the noise-region part of an MRAF field (layout of gh); H = gh * wscale + gh2
    const unsigned short* gh2_mask;   // ... and the columns in which it exists (a NaN target in the column); nullptr = all
};

// NS < 16 (one-row workgroups only): the SLM columns occupy at most NS of the 16 register slots of the space side.  The
// transform input is shifted circularly by m0 slots (RowArgs::m0) so that they are slots 0 .. NS-1 for every geometry;
// by the shift theorem that multiplies frequency k by exp(-2 pi i k m0 / 16), a per-lane constant folded into the scale
// multiplies on the GH side.  The empty slots then cost nothing: no phasor, no predicate, and the first radix-4 layer of
// the forward transform / the last one of the inverse shrink (fwd_lead / inv_trail).
// 16 bytes per lane global -> LDS (global_load_lds_dwordx4): the destination is the wave-uniform `lds_dst` + 16 * lane.
// (A plain function on purpose: with the builtin inside a kernel TEMPLATE the host pass silently drops the kernel's stub.)
__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// PREF (fp32, one-row workgroups, MODE 2, dense launches): a workgroup walks several rows (grid = 2 x #CU) and the H row
// it will need NEXT goes global -> LDS by global_load_lds_dwordx4 while it transforms the current one -- 32 KB in flight
// that no register holds.  The rows of a CU then sit in different phases by construction (one loads while the other
// transforms), which is what the one-row-per-workgroup launch cannot have: there the rows of a round load, transform and
// store in lock step and the traffic adds to the transform time.  Costs: 32 KB more LDS per
// workgroup (two per CU), one LDS read per element, raw barriers in the transform (see WgFftL RAWBAR).
// SPLIT (single-pass MRAF, col_tile_kernel RULE 3): the column kernel left the two parts of the constrained field apart,
// A = the signal region with the UN-normalised new weights (gh) and B = the noise region (gh2), both already transformed
// along the columns; the row to transform is A / ||w'|| + B (the transforms are linear, and ||w'|| is known by now).
template <typename R, int N, int MODE, int NS = 16, bool PREF = false, bool SPLIT = false>
// (8192-wide rows: a workgroup is 8 waves, two per SIMD -- a second resident workgroup needs four waves per SIMD,
//  i.e. at most 128 VGPRs)
// (the phase-extracting forms over all 16 register slots, MODE 1 / 3 with NS = 16 from 4096 columns on, and MODE 3 below
//  that -- one launch per engine call -- are compiled for two waves per SIMD: at three / four they spilled 4 .. 48 VGPRs)
__global__ __launch_bounds__(RowCfg<N>::WG, (sizeof(R) == 8 ? 2 : (((MODE == 1 || MODE == 3) && NS == 16 && N >= 4096 && !SPLIT) || (MODE == 3 && N < 4096)) ? 2 :
                                             (N >= 8192 ? HGS_ROW_OCC_8192 : PREF ? 2 : HGS_ROW_OCC))) void row_kernel(RowArgs<R> a) {
    static_assert(NS == 16 || (RowCfg<N>::FPW == 1 && NS >= 4 && NS < 16), "row_kernel: shifted form is for one-row workgroups");
    static_assert(!PREF || (sizeof(R) == 4 && N == 4096 && MODE == 2), "row_kernel: the prefetching form is fp32, 4096 wide, MODE 2");
    static_assert(!SPLIT || (!PREF && MODE != 0 && RowCfg<N>::T >= 256), "row_kernel: the split form reads H, one-row workgroups");
    using M = Math<R>;
    constexpr int T = RowCfg<N>::T, FPW = RowCfg<N>::FPW;
    // pixels whose atan2 (phase extraction, MODE 1 / 3) the scheduler may interleave: four of them in flight cost ~40
    // registers, which the instances that keep 16 unshifted slots next to a forward transform (MODE 3), run at four waves
    // per SIMD (8192-wide rows) or carry two lane groups (narrow rows) do not have -- they spilled 8 .. 48 VGPRs
    constexpr int AT_GROUP = (MODE == 3 || N >= 8192 || T < 256) ? 1 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int tid = threadIdx.x;
    // (T a multiple of 64: the row of a lane is the same for its whole wave -- say so, or every per-row predicate and
    //  address is treated as divergent: exec-mask branches, per-lane 64-bit addresses, unmerged loads)
    const int f = (T % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T, j = tid % T;
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem) + f * lds_elems<N>();

    // deferred weight normalisation: one block per hologram folds the partial sums of the last
    // weight update into the scalar every later reader multiplies by (_hologram.py:1877).
    // (done by the LAST block, which the host launches in addition to the row blocks: it owns no row, so the
    //  reduction does not lengthen any row's critical path)
    if (a.wpartial != nullptr && blockIdx.x == gridDim.x - 1) {
        double s = 0;
        for (int i = tid; i < a.n_wpartial; i += blockDim.x) s += a.wpartial[(size_t)b * a.n_wpartial + i];
        double* scratch = reinterpret_cast<double*>(smem);
        s = block_sum(s, scratch);
        if (tid == 0) a.wscale[b] = (R)(1.0 / ::sqrt(s));
        __syncthreads();
    }

    // (fp64: 64 data registers per lane already; the stage twiddles are fetched per use instead of kept)
    // (fp32 instances that ran out of registers with them -- every phase-extracting launch, i.e. one per engine call, and the
    //  unshifted 8192-wide form -- fetch them per use as well: 8 .. 48 spilled VGPRs -> 0, tools/resusage.sh)
    constexpr bool TW_RES = sizeof(R) == 8 ? false : (MODE == 1 || MODE == 3 || (N >= 8192 && NS == 16)) ? false : HGS_ROW_TW_RESIDENT;
    // float64 rows of 4096 / 8192 columns: the per-use stage twiddles from LDS tables instead of global loads inside the
    // transform's dependent chain (round 6; the tables sit behind the transform image)
    constexpr bool LTW = fused_ltw<R, N>() && !TW_RES && HGS_F64_LTW_ROW;
    using Sel = FftSel<R, N, TW_RES, PREF, LTW>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    if constexpr (LTW) {
        Cx<R>* ltab = reinterpret_cast<Cx<R>*>(smem) + FPW * lds_elems<N>();
        Sel::type::ltw_fill(a.tw, ltab, tid, (int)blockDim.x);
        fft.set_ltw(ltab);
        __syncthreads();
    }

    // lane j owns elements j + m*T of the frequency side (GH columns) and js + m*T of the space side (SLM columns)
    const int js = Sel::space_lane(j);
    const R sgn = (j & 1) ? (R)-1 : (R)1;    // (-1)^(j + m*T), T even: frequency side
    const R sgs = (js & 1) ? (R)-1 : (R)1;   // space side
    // GH element (r, k = j + m*T) sits at ((k>>2)*Sh + r)*4 + (k&3) = lane part + m * (T*Sh)
    Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw;
    const unsigned gh_lane = (unsigned)(j >> 2) * g.Sh * 4u + (unsigned)(j & 3);
    const unsigned gh_step = (unsigned)T * g.Sh;
    // SLM column of element m is c_lane + m*T (shifted form: of slot m, i.e. element m + m0)
    const int c_lane = js - g.c0 + (NS < 16 ? a.m0 * T : 0);
    Cx<R> om = mk<R>(1, 0);       // shift-theorem factor of this lane's frequencies (k = j mod 16)
    if constexpr (NS < 16) om = a.tw[((a.m0 * (j & 15)) & 15) * (N / 16)];
    // sparse targets: bit m of the masks = column j + m*T is active (see ColArgs::col_list);
    // lane_mask[b][16][Pw/16] holds the 16-bit mask of lane j of a length-Pw row transform
    // The loads of the H row are predicated on the mask, so its fetch is on the critical path of every
    // workgroup: take it through the scalar cache (the 64 masks of a wave are 32 consecutive words at a
    // wave-uniform address) and hand each lane its half-word through the LDS crossbar, instead of
    // a per-lane global load that costs a full memory round trip before the first H load can issue.
    unsigned lmask = 0xffffu, smask = 0xffffu;
    auto fetch_mask = [&](const unsigned short* tab) -> unsigned {
        if constexpr (T % 64 == 0) {
            const int lane = tid & 63;
            const int wbase = __builtin_amdgcn_readfirstlane((tid % T) & ~63);
            const unsigned* mw = reinterpret_cast<const unsigned*>(tab + (size_t)b * T + wbase);   // wave-uniform
            unsigned mine = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const unsigned sw = mw[k];
                mine = (lane == k) ? sw : mine;
            }
            const unsigned got = __shfl(mine, lane >> 1, 64);
            return (lane & 1) ? (got >> 16) : (got & 0xffffu);
        } else {
            return tab[(size_t)b * T + j];
        }
    };
    if (a.load_mask != nullptr) lmask = fetch_mask(a.load_mask);
    if (a.store_mask != nullptr) smask = (a.store_mask == a.load_mask) ? lmask : fetch_mask(a.store_mask);

    // Blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, a speed-only
    // assumption): give the four rows that share each 128-byte GH line to four blocks of the same
    // XCD that start together, so the partial-line accesses meet in that XCD's L2.
    // A workgroup pass covers FPW rows, so a line group of four rows is G = 4 / FPW workgroups (one-row workgroups: 4,
    // two-row ones: 2; with four or more rows per workgroup the group never leaves it).  The row-owning blocks come
    // in multiples of 8 G (host), and the stride between passes is that count -- a multiple of four rows -- so every
    // pass keeps the groups aligned, also in a batch (grid.y) and with the extra row-less block of the weight norm.
    int first = blockIdx.x * FPW;
    if (a.xcd_map) {
        constexpr int G = FPW <= 4 ? 4 / FPW : 1;
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        first = (G * ((idx / G) * 8 + xcd) + (idx % G)) * FPW;
    }
    const int row_stride = (a.n_row_blocks > 0 ? a.n_row_blocks : (int)gridDim.x) * FPW;
    if (a.n_row_blocks > 0 && (int)blockIdx.x >= a.n_row_blocks) first = g.Sh;     // the weight-norm block owns no row
    // PREF: the H row of `r` into the linear image P behind the transform image: instruction i of the workgroup (8 per
    // wave) brings elements 128 i .. 128 i + 127, lane L the two at 128 i + 2 L (one 16-byte half of a 32-byte tile row)
    Cx<R>* pimg = lds + lds_elems<N>();
    auto prefetch_row = [&](int r) {
        if constexpr (PREF) {
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = wv * 8 + q, k = 128 * i + 2 * ln;
                const Cx<R>* src = gh + ((size_t)(k >> 2) * g.Sh + r) * 4 + (k & 3);
                glds16(src, pimg + 128 * i);
            }
        }
    };
    if constexpr (PREF) {
        if (first < g.Sh) prefetch_row(first);      // (no wait here: the set-up below runs under the first row's flight)
    }
    HGS_T(fft.tr_n, 1);
#pragma unroll 1
    for (int rbase = first; rbase < g.Sh; rbase += row_stride) {
        const int r = rbase + f;
        const bool valid = r < g.Sh;
        const int rr = valid ? r : 0;
        Cx<R> v[16];
        const size_t srow = (size_t)rr * g.Sw;
        R* ph = a.phase + (size_t)b * g.Sh * g.Sw + srow;
        const R* kn = a.kern ? a.kern + srow : nullptr;
        const R* am = a.amp ? a.amp + srow : nullptr;
        Cx<R>* ghr = gh + (size_t)rr * 4;

        // (measured: 2048^2 22.5 -> 21.5 us, 1024^2 10.8 -> 10.2 us; 4096 / 8192-wide rows lose 1 us with it -- their
        //  one-row workgroups gain nothing from the shorter code and pay for the 8 more eager phasors -- and keep the
        //  branching form)
        if constexpr (HGS_ROW_BUF && T % 64 == 0 && T <= 128) {
            // ---- straight-line form: raw buffer accesses (see Buf), no per-element branch ----
            // resources of this row (wave-uniform); a row past Sh gets empty ones: loads give 0, stores vanish
            constexpr unsigned CB = sizeof(Cx<R>), RB = sizeof(R);
            const unsigned row_bytes = valid ? (unsigned)g.Sw * RB : 0u;
            const Buf bph(ph, row_bytes);
            const Buf bkn(kn, kn != nullptr ? row_bytes : 0u);
            const Buf bam(am, am != nullptr ? row_bytes : 0u);
            const Buf bgh(gh, valid ? (unsigned)((size_t)g.Sh * g.Pw * CB) : 0u);
            const unsigned ghl = gh_lane * CB, ghs = gh_step * CB, ghr0 = (unsigned)rr * 4u * CB;
            const unsigned c_off = (unsigned)c_lane * RB, c_step = (unsigned)T * RB;   // negative columns wrap out of range
            if constexpr (MODE != 0) {
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    const unsigned vo = ((lmask >> m) & 1u) ? ghl : BUF_OOB;
                    v[m] = bgh.template ld<Cx<R>>(vo, ghr0 + (unsigned)m * ghs) * sgn;
                });
#if HGS_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                HGS_T(fft.tr_n, 2);
                if constexpr (MODE >= 2) fft.inv_after_fwd(v, lds, j);
                else fft.inv(v, lds, j);
                if constexpr (MODE == 1) {
                    const R sc = sgs * a.scale;
                    if (a.nf_out != nullptr) {
                        const Buf bnf(a.nf_out + (size_t)b * g.Sh * g.Sw + srow, row_bytes * 2u);
                        static_for<0, 16>([&](auto m_) {
                            constexpr int m = m_;
                            bnf.template st<Cx<R>>(v[m] * sc, (c_off + (unsigned)m * c_step) * 2u, 0u);
                        });
                    } else {
                        static_for<0, 16>([&](auto m_) {
                            constexpr int m = m_;
                            const unsigned co = c_off + (unsigned)m * c_step;
                            R p = M::atan2(v[m].y * sc, v[m].x * sc) - bkn.template ld<R>(co, 0u);   // (no kernel: reads 0)
                            bph.template st<R>(p, co, 0u);
                            if constexpr (m % AT_GROUP == AT_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                        });
                    }
                }
            }
            if constexpr (MODE != 1) {
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    const unsigned co = c_off + (unsigned)m * c_step;
                    R amv = bam.template ld<R>(co, 0u);                      // (no amplitude array: empty resource)
                    if (am == nullptr) amv = (co < row_bytes) ? a.amp_scalar : (R)0;
                    Cx<R> nf;
                    if constexpr (MODE == 3) {                  // the phase as MODE 1 stores it (no kernel: reads 0) ...
                        const R sc1 = sgs * a.scale;
                        bph.template st<R>(M::atan2(v[m].y * sc1, v[m].x * sc1) - bkn.template ld<R>(co, 0u), co, 0u);
                    }
                    if constexpr (MODE >= 2) {                  // ... and G as MODE 2 builds it
                        // evaluated eagerly and selected (a conditional around it is a branch per element)
                        const R p2 = v[m].x * v[m].x + v[m].y * v[m].y;
                        const Cx<R> on = v[m] * (amv * rsqrt_full(p2));
                        nf.x = (p2 > (R)0) ? on.x : amv * sgs;
                        nf.y = (p2 > (R)0) ? on.y : (R)0;
                    } else {
                        const R p = bph.template ld<R>(co, 0u) + bkn.template ld<R>(co, 0u);
                        R sn, cs;
                        M::sincos(p, &sn, &cs);
                        nf = mk<R>(amv * sgs * cs, amv * sgs * sn);
                    }
                    v[m] = nf;
                    if constexpr (m % 4 == 3) __builtin_amdgcn_sched_barrier(0);
                });
                HGS_T(fft.tr_n, 3);
                fft.fwd(v, lds, j);
                HGS_T(fft.tr_n, 4);
                const R sc = sgn * a.scale;
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    const unsigned vo = ((smask >> m) & 1u) ? ghl : BUF_OOB;
                    bgh.template st<Cx<R>>(v[m] * sc, vo, ghr0 + (unsigned)m * ghs);
                });
#if HGS_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                HGS_T(fft.tr_n, 5);
            }
        } else {
        // An array-valued source amplitude (a measured beam: the usual laboratory case) is requested HERE, ahead of the inverse
        // transform, through a buffer resource (columns outside the SLM are outside it and read as zero; no array: an empty
        // one).  Fetched where it is used -- inside the per-slot exec-masked block of the phasor, `s_waitcnt vmcnt(0)` right
        // behind each load -- it was NS dependent memory round trips per row: cfg 2 with a Gaussian amplitude 73.0 against
        // 65.5 us per iteration with the scalar one (tools/amp_array_probe.py, round 6).
        constexpr bool AMPF = HGS_ROW_AMP_PREFETCH && MODE >= 2 && T % 64 == 0 && !(N >= 8192 && NS == 16);
        R amr[AMPF ? NS : 1];
        auto issue_amp = [&]() {
            const Buf bam(am, (am != nullptr && valid) ? (unsigned)g.Sw * (unsigned)sizeof(R) : 0u);
            static_for<0, NS>([&](auto m_) { constexpr int m = m_; amr[m] = bam.template ld<R>((unsigned)(c_lane + m * T) * (unsigned)sizeof(R), 0u); });
        };
        // (PREF: behind the pick-up of the staged row -- ahead of it, the `s_waitcnt vmcnt(0)` that waits for the staged pieces
        //  waited for these requests as well, a round trip at the head of every row)
        constexpr bool AMP_LATE = AMPF && PREF && MODE != 0 && HGS_ROW_AMP_LATE;
        if constexpr (AMPF && !AMP_LATE) issue_amp();
        if constexpr (MODE != 0) {
            // ---- load H row, centred inverse transform along x ----
            if constexpr (PREF) {
                // the row has been on its way since the previous one was picked up: wait for this wave's pieces, meet the
                // other waves, read; when every lane has its values the image is free for the row after this one
                // (vmcnt(16) -- leaving the sixteen G stores of the row before in flight -- measured the same)
                HGS_T(fft.tr_n, 30);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                HGS_T(fft.tr_n, 31);
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    const Cx<R> h = pimg[j + m * T];
                    if constexpr (NS < 16) v[m] = h;
                    else v[m] = h * sgn;
                });
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                HGS_T(fft.tr_n, 32);
                if (rbase + row_stride < g.Sh) prefetch_row(rbase + row_stride);
                if constexpr (AMP_LATE) issue_amp();
                HGS_T(fft.tr_n, 33);
            } else {
            // dense fp32 launches: the H row in 16-byte pieces, the lane pair swapping one value per slot pair (see the G stores)
            bool wide = false;
            if constexpr (sizeof(R) == 4 && HGS_ROW_LD16 && T % 64 == 0) wide = a.load_mask == nullptr && valid;
            if (wide) {
                if constexpr (sizeof(R) == 4 && HGS_ROW_LD16 && T % 64 == 0) {
                    const bool odd = (j & 1) != 0;
                    const Cx<R>* pbase = ghr + (gh_lane - (odd ? 1u : 0u)) + (odd ? (size_t)gh_step : (size_t)0);
                    float4 q[8];
                    static_for<0, 8>([&](auto p_) { constexpr int m = 2 * p_; q[p_] = *reinterpret_cast<const float4*>(pbase + (size_t)m * gh_step); });
                    static_for<0, 8>([&](auto p_) {
                        constexpr int m = 2 * p_;
                        const R sx = odd ? q[p_].x : q[p_].z, sy = odd ? q[p_].y : q[p_].w;      // the neighbour's value
                        const R rx = __builtin_bit_cast(R, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sx), 0xB1, 0xf, 0xf, true));
                        const R ry = __builtin_bit_cast(R, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sy), 0xB1, 0xf, 0xf, true));
                        const Cx<R> h0 = odd ? mk<R>(rx, ry) : mk<R>(q[p_].x, q[p_].y);
                        const Cx<R> h1 = odd ? mk<R>(q[p_].z, q[p_].w) : mk<R>(rx, ry);
                        if constexpr (NS < 16) { v[m] = h0; v[m + 1] = h1; }
                        else { v[m] = h0 * sgn; v[m + 1] = h1 * sgn; }
                    });
                }
            } else
            static_for<0, 16>([&](auto m_) {
                constexpr int m = m_;
                Cx<R> h = mk<R>(0, 0);
                if (valid && ((lmask >> m) & 1u)) h = (ghr + (size_t)m * gh_step)[gh_lane];
                if constexpr (NS < 16) v[m] = h;
                else v[m] = h * sgn;
            });
            }
            if constexpr (SPLIT) {
                const R ws = a.wscale[b];
                const Cx<R>* gh2r = a.gh2 + (size_t)b * g.Sh * g.Pw + (size_t)rr * 4;
                // the noise part is zero -- and, with a mask, not even stored -- outside the columns that hold a NaN target
                const unsigned nmask = a.gh2_mask != nullptr ? (lmask & fetch_mask(a.gh2_mask)) : lmask;
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    Cx<R> h2 = mk<R>(0, 0);
                    if (valid && ((nmask >> m) & 1u)) h2 = (gh2r + (size_t)m * gh_step)[gh_lane];
                    if constexpr (NS < 16) v[m] = v[m] * ws + h2;
                    else v[m] = v[m] * ws + h2 * sgn;
                });
            }
            // (-1)^k H[k] conj(shift factor) -- in a pass of its own: the product is inline asm, and next to its load
            // inside the per-element branch it made every load wait for the one before
            if constexpr (NS < 16) {
                const Cx<R> oms = om * sgn;
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = cmulc(v[m], oms); });
            }
#if HGS_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HGS_T(fft.tr_n, 2);
            // (MODE 2: the previous user of the LDS image was the forward transform of the row before)
            if constexpr (NS < 16) {
                if constexpr (MODE >= 2) fft.template inv_after_fwd_trail<NS>(v, lds, j);
                else fft.template inv_trail<NS>(v, lds, j);
            } else {
                if constexpr (MODE >= 2) fft.inv_after_fwd(v, lds, j);
                else fft.inv(v, lds, j);
            }
            const R sc = sgs * a.scale;
            if constexpr (MODE == 1) {
                static_for<0, NS>([&](auto m_) {
                    constexpr int m = m_;
                    const int c = c_lane + m * T;
                    if (valid && c >= 0 && c < g.Sw) {
                        // nf = sgn * scale * v;  _nearfield_extract :1030, :1036
                        if (a.nf_out != nullptr) {
                            (a.nf_out + (size_t)b * g.Sh * g.Sw + srow)[c] = v[m] * sc;
                        } else {
                            R p = M::atan2(v[m].y * sc, v[m].x * sc);
                            if (kn != nullptr) p -= kn[c];
                            ph[c] = p;
                        }
                    }
                    if constexpr (m % AT_GROUP == AT_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        if constexpr (MODE != 1) {
            // ---- build the nearfield row (amp * exp(i(phase+kernel)), zero padded), transform ----
            // MODE 2 (inverse of iteration i fused with the forward of i+1): the phase is never
            // materialised -- phase = atan2(nf) - kernel (:1030-1036) and the rebuild uses
            // exp(i(phase + kernel)) = nf/|nf| (:1004-1008), so the kernel cancels and one rsqrt replaces
            // atan2 + sincos; atan2(0,0) = 0 gives the phasor 1.  The last row kernel of an
            // hgs_iterate call runs MODE 1 and writes the phase.
            if constexpr (NS < 16) static_for<NS, 16>([&](auto m_) { constexpr int m = m_; v[m] = mk<R>(0, 0); });
            static_for<0, NS>([&](auto m_) {
                constexpr int m = m_;
                const int c = c_lane + m * T;
                Cx<R> nf = mk<R>(0, 0);
                if (valid && c >= 0 && c < g.Sw) {
                    R amv;
                    if constexpr (AMPF) amv = (am != nullptr) ? amr[m] : a.amp_scalar;
                    else amv = (am != nullptr) ? am[c] : a.amp_scalar;
                    if constexpr (MODE == 3) {                  // the phase exactly as MODE 1 stores it ...
                        const R scs = sgs * a.scale;
                        R p = M::atan2(v[m].y * scs, v[m].x * scs);
                        if (kn != nullptr) p -= kn[c];
                        ph[c] = p;
                    }
                    if constexpr (MODE >= 2 && HGS_ROW_PHASOR) {        // ... and G exactly as MODE 2 builds it
                        // nearfield of the inverse = sgn*scale*v, input of the forward = sgn*amp*phasor
                        const R p2 = v[m].x * v[m].x + v[m].y * v[m].y;
                        const Cx<R> on = v[m] * (amv * rsqrt_full(p2));      // (eager + select: no inner branches)
                        nf = mk<R>((p2 > (R)0) ? on.x : amv * sgs, (p2 > (R)0) ? on.y : (R)0);
                    } else if constexpr (MODE >= 2) {
                        // the reference's own arithmetic: phase rounded to working precision, then exp(i phase)
                        const R scs = sgs * a.scale;
                        R p = M::atan2(v[m].y * scs, v[m].x * scs);
                        if (kn != nullptr) { p -= kn[c]; p += kn[c]; }
                        R s, co;
                        M::sincos(p, &s, &co);
                        nf = mk<R>(amv * sgs * co, amv * sgs * s);
                    } else {
                        R p = ph[c];
                        if (kn != nullptr) p += kn[c];
                        R s, co;
                        M::sincos(p, &s, &co);
                        nf = mk<R>(amv * sgs * co, amv * sgs * s);
                    }
                }
                v[m] = nf;
                if constexpr (m % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            });
            HGS_T(fft.tr_n, 3);
            if constexpr (NS < 16) fft.template fwd_lead<NS>(v, lds, j);
            else fft.fwd(v, lds, j);
            HGS_T(fft.tr_n, 4);
            if (valid) {
                const R sc = sgn * a.scale;
                const Cx<R> omsc = om * sc;
                // Dense fp32 launches: the sixteen 8-byte stores of a lane are store-ISSUE bound (2.7 k of the 16.3 k cycles of
                // a row in tools/microbench/trace_row; MI355X_MICROARCH.md: 8 x dwordx4 halves such a tail).  Lanes j, j + 1
                // (j even) own neighbouring columns of the same tile row, so they swap one value per pair of register slots
                // (quad_perm [1, 0, 3, 2]) and each stores 16 bytes: the even lane the pair of slot m, the odd one that of
                // slot m + 1 -- eight store instructions instead of sixteen, the same bytes at the same addresses.
                // (measured: a batch of eight, one-row workgroups: row launch 89.8 -> 86.9 us; the prefetching walk of a single
                //  hologram LOSES 1.2 us to the swaps on its chain and keeps the 8-byte stores; 8192-wide rows: level)
                if constexpr (PREF && HGS_ROW_PREF_BUFST) {
                    // the prefetching walk (dense launches only: no store mask): sixteen straight-line stores through a buffer
                    // resource -- the lane part of the address one VGPR offset, the register part an SGPR offset.  As plain stores
                    // each sat in its own exec-masked block (the mask test) with a 64-bit address computed in front of it
                    // (launch_row_f32.s: s_and_saveexec / s_cbranch / s_mul / s_add / v_lshl_add_u64 per store)
                    const Buf bg(gh, (unsigned)((size_t)g.Sh * g.Pw * sizeof(Cx<R>)));
                    const unsigned vo = (gh_lane + (unsigned)rr * 4u) * (unsigned)sizeof(Cx<R>);
                    static_for<0, 16>([&](auto m_) {
                        constexpr int m = m_;
                        Cx<R> e;
                        if constexpr (NS < 16) e = cmul(v[m], omsc); else e = v[m] * sc;
                        bg.template st<Cx<R>>(e, vo, (unsigned)m * gh_step * (unsigned)sizeof(Cx<R>));
                    });
                } else
                if constexpr (sizeof(R) == 4 && HGS_ROW_ST16 && T % 64 == 0 && !PREF) {
                    if (a.store_mask == nullptr) {
                        const bool odd = (j & 1) != 0;
                        Cx<R>* pbase = ghr + (gh_lane - (odd ? 1u : 0u)) + (odd ? (size_t)gh_step : (size_t)0);
                        static_for<0, 8>([&](auto p_) {
                            constexpr int m = 2 * p_;
                            Cx<R> e0, e1;
                            if constexpr (NS < 16) { e0 = cmul(v[m], omsc); e1 = cmul(v[m + 1], omsc); }
                            else { e0 = v[m] * sc; e1 = v[m + 1] * sc; }
                            const R sx = odd ? e0.x : e1.x, sy = odd ? e0.y : e1.y;       // what the neighbour stores for me
                            const R kx = odd ? e1.x : e0.x, ky = odd ? e1.y : e0.y;       // what I store myself
                            const R rx = __builtin_bit_cast(R, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sx), 0xB1, 0xf, 0xf, true));
                            const R ry = __builtin_bit_cast(R, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, sy), 0xB1, 0xf, 0xf, true));
                            const float4 out = odd ? make_float4(rx, ry, kx, ky) : make_float4(kx, ky, rx, ry);
                            *reinterpret_cast<float4*>(pbase + (size_t)m * gh_step) = out;
                        });
                    } else {
                        static_for<0, 16>([&](auto m_) {
                            constexpr int m = m_;
                            if ((smask >> m) & 1u) {
                                if constexpr (NS < 16) (ghr + (size_t)m * gh_step)[gh_lane] = cmul(v[m], omsc);
                                else (ghr + (size_t)m * gh_step)[gh_lane] = v[m] * sc;
                            }
                        });
                    }
                } else
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    // (the prefetching walk is launched dense-only: no mask test, so that its sixteen stores are one block)
                    if ((PREF && HGS_ROW_PREF_NOMASK) || ((smask >> m) & 1u)) {
                        if constexpr (NS < 16) (ghr + (size_t)m * gh_step)[gh_lane] = cmul(v[m], omsc);
                        else (ghr + (size_t)m * gh_step)[gh_lane] = v[m] * sc;
                    }
                });
            }
#if HGS_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HGS_T(fft.tr_n, 5);
        }
        }
    }
#if HGS_TRACE
    __syncthreads();
    if (a.nf_out != nullptr && MODE == 2) {   // the microbenchmark passes its dump buffer through nf_out
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.nf_out) + (size_t)blockIdx.x * 512;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + HGS_TRACE_OFF);
        for (int i = tid; i < 512; i += blockDim.x) dst[i] = src[i];
    }
#endif
}

// =====================================================================================================
// COLUMN kernel: transforms along y for a tile of 4 adjacent columns.
//   FWD   : G -> F            (column half of fft2, zero rows outside the SLM are never read)
//   STORE : write F, |F| (and optionally atan2 F) to the column-major farfield arrays
//           (_midloop_cleaning :953, _populate_results :948-949) + sum |F|^2 partial
//   LOAD  : read the (already constrained) farfield array instead
//   INV   : F -> H            (column half of ifft2; only the Sh SLM rows are produced)
// grid = (<= Pw/4, batch), block = T*CPAR; a workgroup strides over tiles, the 4 columns of a tile
// run CPAR at a time.
// =====================================================================================================
enum { C_FWD = 1, C_STORE = 2, C_LOAD = 8, C_INV = 16 };

template <int N> struct ColCfg {
    static constexpr int T = N / 16;
    static constexpr int CPAR = T >= 256 ? 1 : (256 / T > 4 ? 4 : 256 / T);
    static constexpr int WG = T * CPAR;
    static constexpr int PASSES = 4 / CPAR;
};

template <typename R> struct ColArgs {
    Geo g;
    Cx<R>* gh;
    Cx<R>* ff;        // column-major farfield (STORE / LOAD) or nullptr
    R* amp_ff;        // column-major (STORE) or nullptr
    R* pff;           // column-major phase_ff or nullptr
    R* w;             // column-major weights
    const R* t;       // column-major target
    const R* wscale;  // [batch] pending 1/||w||
    double* wpartial; // [batch][gridDim.x] sum w'^2 (CONS with do_update)
    double* fpartial; // [batch][gridDim.x] sum |F|^2 (STORE)
    const Cx<R>* tw;  // W_Ph table
    R scale;          // 1/sqrt(Ph)
    int store_pff;    // STORE: also write phase_ff
    CParams<R> cp;
    // col_fused_kernel only: sparse targets.  When col_list != nullptr the kernel transforms just the
    // listed columns (those holding a non-zero weight or target): every other column of the constrained
    // farfield is exactly zero, so its inverse transform is zero and the row kernel does not read it.
    int list_xmap;         // the same for column-list launches: list groups PASSES k .. PASSES k + PASSES - 1 (the columns
                           // of one tile where the active set is dense) on one XCD together (gridDim.x a multiple of 8 * PASSES)
    Cx<R>* ffb;            // col_fused_kernel with CParams::split: noise part of the constrained farfield, layout of ff
    Cx<R>* gh2;            // col_tile_kernel RULE 3 (single-pass MRAF): column-transformed noise-region part, layout of gh
    int gh2_sparse;        // ... stored only for tiles (NR <= 4) / columns (NR > 4) that hold a noise pixel: the row kernel
                           //     reads it through RowArgs::gh2_mask, which marks exactly the columns with a NaN target
    int col_xmap;          // dense launches of col_fused_kernel with fewer than four columns per pass: the passes of one
                           // 4-column tile go to workgroups of ONE XCD that run together (gridDim.x a multiple of 8 * PASSES)
    const unsigned char* col_flags;   // [batch][Pw] scan_active_cols bits, or nullptr (col_tile_kernel RULE 4 skips the inverse
                                      // transforms of the parts that are zero in a column)
    const int* col_list;   // [batch][Pw] compacted active columns
    const int* n_active;   // [batch]
    // fused kernels only: statistics of this iteration (hgs_iterate_stats)
    const unsigned short* sig_rows;   // col_presum_kernel: [batch][Pw] which register slots of a column hold signal pixels (column scan), or nullptr
    const double* dpartial;   // col_tile_kernel RULE 5 (MRAF with a weight update, ONE inverse per column): per-workgroup partials
    int n_dpartial;           // of D = sum w'^2 - sum w^2 over the signal pixels, left by col_presum_kernel (written there through
                              // wpartial's neighbour, see engine.hip); 1 / ||w'|| = 1 / sqrt(1 + D) is known BEFORE the field is rebuilt
    int few_active;        // at most a quarter of the farfield columns hold a non-zero weight or target (col_tile2_kernel NXF)
    int fnr;               // col_fused_kernel: register slots the shifted SLM rows occupy (0 = the unshifted kernel, NRS = 16)
    int fshift;            // col_fused_kernel<..., NRS < 16> (float64, >= 4096 rows): circular shift of the transform input, a
                           // multiple of 16 rows (shift theorem, as in col_tile_kernel): the SLM rows occupy slots 0 .. NRS - 1
    int do_stats;          // bit 0: accumulate the "computational" statistics; bit 1: store amp_ff
    double* spartial;      // [batch][gridDim.x][STAT_WAVES][STAT_N]
    const double* tsum;    // [batch] sum T^2
    double inv_fsum;       // 1 / sum |F|^2
};

template <typename R, int N, int MODE>
// (fp64: two waves per SIMD -- at three, every instantiation spilled 17 .. 74 VGPRs)
__global__ __launch_bounds__(ColCfg<N>::WG, (sizeof(R) == 8 ? 2 : N >= 8192 ? HGS_COL_OCC_8192 : HGS_COL_OCC)) void col_kernel(ColArgs<R> a) {   // (8192: see row_kernel)
    using M = Math<R>;
    constexpr int T = ColCfg<N>::T, CPAR = ColCfg<N>::CPAR, PASSES = ColCfg<N>::PASSES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int tid = threadIdx.x;
    const int cpar = (T % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T, j = tid % T;   // wave-uniform (see row_kernel)
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem) + cpar * lds_elems<N>();
    double* scratch = reinterpret_cast<double*>(reinterpret_cast<Cx<R>*>(smem) + CPAR * lds_elems<N>());

    using Sel = FftSel<R, N, true>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    const int js = Sel::space_lane(j);       // rows js + m*T (space side), farfield pixels j + m*T (frequency side)
    const R sgn = (j & 1) ? (R)-1 : (R)1;
    const R sgs = (js & 1) ? (R)-1 : (R)1;
    const size_t P = (size_t)g.Ph * g.Pw;
    const R sc = sgn * a.scale;
    double acc_f = 0;
    const int r_lane = js - g.r0;  // SLM row of element m is r_lane + m*T

    // column schedule: tiles of 4 columns strided over the grid, or (col_list != nullptr)
    // just the listed columns -- sparse targets, see ColArgs::col_list
    const bool listed = a.col_list != nullptr;
    const int* clist = listed ? a.col_list + (size_t)b * g.Pw : nullptr;
    const int n_act = listed ? a.n_active[b] : 0;
    const int ntiles = g.Pw / 4;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_grp = (n_act + CPAR - 1) / CPAR;
    const int nq = listed ? ((int)blockIdx.x < n_grp ? (n_grp - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0)
                          : my_tiles * PASSES;
#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        int ct, c4;
        bool vcol = true;        // a lane group past the end of the list runs on zeros and stores nothing
        if (listed) {
            const int e = ((int)blockIdx.x + q * (int)gridDim.x) * CPAR + cpar;
            vcol = e < n_act;
            const int col = clist[min(e, n_act - 1)];
            ct = col >> 2;
            c4 = col & 3;
        } else {
            ct = blockIdx.x + (q / PASSES) * gridDim.x;
            c4 = (q % PASSES) * CPAR + cpar;
        }
        Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4;
        {
            const int kx = ct * 4 + c4;
            const size_t cb = (size_t)b * P + (size_t)kx * g.Ph;   // column base in the P arrays
            Cx<R> v[16];
            // (rows outside the SLM through the range check of a buffer resource where the column is wave-uniform)
            constexpr bool GBUF = HGS_COL_BUF && T % 64 == 0;
            constexpr unsigned CB = sizeof(Cx<R>);
            const Buf bg(gh + c4, vcol ? (unsigned)(g.Sh * 4 - c4) * CB : 0u);
            const unsigned g_voff = (unsigned)r_lane * 4u * CB, g_vstep = (unsigned)T * 4u * CB;
            if constexpr (MODE & C_FWD) {
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    Cx<R> x = mk<R>(0, 0);
                    if constexpr (GBUF) {
                        x = bg.template ld<Cx<R>>(g_voff + (unsigned)m * g_vstep, 0u);
                    } else {
                        const int r = r_lane + m * T;
                        if (r >= 0 && r < g.Sh && vcol) x = gh[(unsigned)r * 4u + (unsigned)c4];
                    }
                    v[m] = x * sgs;
                });
                fft.fwd(v, lds, j);
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = v[m] * sc; });
            }
            if constexpr (MODE & C_STORE) {
                Cx<R>* ffc = a.ff + cb;
                R* afc = a.amp_ff + cb;
                R* pfc = a.pff ? a.pff + cb : nullptr;
                // the 16 pixels of a lane are contiguous (lane-major layout): one test, then whole-group stores (an
                // element-wise test + scheduling barrier turned them into 32 scalar stores per lane and pass)
                R af[16];
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    const R p2 = v[m].x * v[m].x + v[m].y * v[m].y;
                    af[m] = M::sqrt(p2);
                    acc_f += (double)p2;
                });
                if (vcol) {
                    static_for<0, 16>([&](auto m_) { constexpr int m = m_; ffc[lane_pos<T>(j, m)] = v[m]; });
                    static_for<0, 16>([&](auto m_) { constexpr int m = m_; afc[lane_pos<T>(j, m)] = af[m]; });
                    if (a.store_pff) {
                        static_for<0, 4>([&](auto q_) {
                            constexpr int q4 = q_;
                            R pf4[4];
                            static_for<0, 4>([&](auto i_) { constexpr int i = i_; pf4[i] = M::atan2(v[4 * q4 + i].y, v[4 * q4 + i].x); });
                            static_for<0, 4>([&](auto i_) { constexpr int i = i_; pfc[lane_pos<T>(j, 4 * q4 + i)] = pf4[i]; });
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    }
                }
            }
            if constexpr (MODE & C_LOAD) {
                const Cx<R>* ffc = a.ff + cb;
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = ffc[lane_pos<T>(j, m)]; });
            }
            if constexpr (MODE & C_INV) {
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = v[m] * sgn; });
                if constexpr (MODE & C_FWD) fft.inv_after_fwd(v, lds, j);
                else fft.inv(v, lds, j);
                const R scs = sgs * a.scale;
                static_for<0, 16>([&](auto m_) {
                    constexpr int m = m_;
                    if constexpr (GBUF) {
                        bg.template st<Cx<R>>(v[m] * scs, g_voff + (unsigned)m * g_vstep, 0u);
                    } else {
                        const int r = r_lane + m * T;
                        if (r >= 0 && r < g.Sh) gh[(unsigned)r * 4u + (unsigned)c4] = v[m] * scs;
                    }
                });
            }
        }
    }
    if constexpr (MODE & C_STORE) {
        const double s = block_sum(acc_f, scratch);
        if (tid == 0) a.fpartial[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
}

// =====================================================================================================
// FUSED column kernel (the dominant kernel of the fast path): for each column of a 4-column tile
//     G --FFT_y--> F --constraint + weight update--> ff --IFFT_y--> H        (in place in GH)
// i.e. the column half of fft2 (:1048), _gs_farfield_routines (:1550-1605) with
// _update_weights_generic (:1822-1879) and the column half of ifft2 (:1070); the farfield is never
// written to HBM.  Software pipeline: the weight/target loads and the G loads of the NEXT column
// are issued right after the constraint of the current one, so they land under the inverse
// transform of this column and the forward transform of the next.
//   PHASE 0: rebuild with F/|F|            (exp(i*atan2 F), :1602-1605, without materialising it)
//   PHASE 1: same + store phase_ff = atan2 F (WGS-Kim before/at the fixing iteration, :1583)
//   PHASE 2: rebuild with the stored phase_ff (fixed phase, :1601)
// Weight rule in log domain for Leonardo/Kim:
//   (|F| * c / T)^-p = exp2(-p * (log2(|F|^2)/2 + log2 c - log2 T)),  T == 0 -> 1 (:1841)
// =====================================================================================================
// RULE (as in col_tile_kernel): 0 = method, update switch, MRAF / Nogrette-sum / forward-only flags read from CParams;
// 1 = plain WGS-Leonardo / WGS-Kim update compiled in; 2 = plain pass without a weight update.  "Plain" = none of the
// extras.  The latency-bound launches (column lists, small grids: one wave per SIMD) pay every uniform branch in full.
// NRS < 16 (round 5; float64 at 4096 / 8192 rows, where every register and every LDS byte counts twice): the transform
// input is shifted by a.fshift rows so that the SLM rows occupy the first NRS register slots (col_tile_kernel's shift
// theorem form: a per-lane unit factor on the frequency side) -- NRS loads and stores per lane instead of 16 range-checked
// ones, the leading butterfly layer of the forward transform and the trailing one of the inverse pruned to those slots,
// at 8192 points the radix-2 step a copy and a twiddle and its pair exchange 2 NRS instead of 16 values per lane.
template <typename R, int N, int PHASE, bool STATS = false, int RULE = 0, int NRS = 16>
__global__ __launch_bounds__(ColCfg<N>::WG, HGS_FUSED_OCC) void col_fused_kernel(ColArgs<R> a) {
    using M = Math<R>;
    constexpr bool SHIFTED = NRS < 16;
    static_assert(!SHIFTED || N >= 4096, "col_fused_kernel: the shifted form needs the row-local transforms");
    constexpr int T = ColCfg<N>::T, CPAR = ColCfg<N>::CPAR, PASSES = ColCfg<N>::PASSES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int tid = threadIdx.x;
    const int cpar = (T % 64 == 0) ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T, j = tid % T;   // wave-uniform (see row_kernel)
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem) + cpar * lds_elems<N>();
    double* scratch = reinterpret_cast<double*>(reinterpret_cast<Cx<R>*>(smem) + CPAR * lds_elems<N>());

    // fp64 doubles every register: keep the weight/target prefetch, drop the resident stage twiddles and the
    // prefetch of the next column's G (both spilled to scratch otherwise: 662 spilled VGPRs at 8192)
    constexpr bool LEAN = sizeof(R) == 8;
    // float64 at 4096 / 8192 rows: the stage twiddles from two small LDS tables (WgFftL LTW) -- nothing of the transform is in
    // the vmcnt queue any more, so the column's weights and targets can be requested ahead of it (round 6)
    constexpr bool LTW = fused_ltw<R, N>();
    using Sel = FftSel<R, N, !LEAN, false, LTW>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    if constexpr (LTW) {
        Cx<R>* ltab = reinterpret_cast<Cx<R>*>(scratch + SCRATCH_DOUBLES);
        Sel::type::ltw_fill(a.tw, ltab, tid, (int)blockDim.x);
        fft.set_ltw(ltab);
        __syncthreads();
    }
    const CParams<R> cp = a.cp;
    const bool do_upd = RULE == 1 ? true : (RULE == 2 ? false : cp.do_update != 0);
    const bool x_mraf = RULE != 0 ? false : cp.mraf != 0;
    const bool x_nog = RULE != 0 ? false : cp.nog_pass != 0;
    const bool x_wonly = RULE != 0 ? false : cp.weights_only != 0;
    const bool x_split = RULE != 0 ? false : (cp.split != 0 && cp.mraf != 0);
    const bool x_presum = RULE != 0 ? false : cp.presum != 0;
    // the main pass behind such a pre-pass: every workgroup folds the partials itself (fixed order: the same bits everywhere).
    // The scale lives in LDS and is read where a pixel is rebuilt: a register held across the transforms is what the float64
    // instances at 8192 points do not have (they went from 0 to 8 .. 35 spilled VGPRs with it)
    double* snew_slot = scratch + (SCRATCH_DOUBLES - 1);
    if constexpr (RULE == 0) {
        double d = 0;
        if (a.dpartial != nullptr) {
            for (int i = tid; i < a.n_dpartial; i += (int)blockDim.x) d += a.dpartial[(size_t)blockIdx.y * a.n_dpartial + i];
            d = block_sum(d, scratch);
        }
        if (tid == 0) *snew_slot = 1.0 / ::sqrt(1.0 + d);
        __syncthreads();
    }
    const int js = Sel::space_lane(j);       // rows js + m*T (space side), farfield pixels j + m*T (frequency side)
    const R sgn = (j & 1) ? (R)-1 : (R)1;
    const R sgs = (js & 1) ? (R)-1 : (R)1;
    const R scs = sgs * a.scale;
    const size_t P = (size_t)g.Ph * g.Pw;
    const R wsc = a.wscale[b];
    const R sc = sgn * a.scale;
    const int shift = SHIFTED ? a.fshift : 0;
    const int r_lane = js + shift - g.r0;  // SLM row of element m is r_lane + m*T
    // SHIFTED: shift-theorem factor of this lane (unit modulus) with the (-1)^k sign folded in
    Cx<R> omu = mk<R>(sgn, 0);
    if constexpr (SHIFTED) omu = a.tw[(j * shift) & (N - 1)] * sgn;
    const int ntiles = g.Pw / 4;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    // listed mode: pass q of this workgroup handles the CPAR list entries of group blockIdx.x + q*gridDim.x;
    // a lane group beyond the end of the list runs on zeros (it shares the barriers) and stores nothing
    const bool listed = a.col_list != nullptr;
    const int* clist = listed ? a.col_list + (size_t)b * g.Pw : nullptr;
    const int n_act = listed ? a.n_active[b] : 0;
    const int n_grp = (n_act + CPAR - 1) / CPAR;
    // XCD-aware form of the dense launch (PASSES > 1: a lane group covers CPAR of the four columns of a tile, i.e. a
    // quarter or a half of every 64- / 32-byte tile row it touches): the PASSES column groups of a tile go to PASSES
    // workgroups of one XCD (workgroups run round-robin over the 8 XCDs -- a speed-only assumption, as in row_kernel)
    // that start together, so the partial rows meet in that XCD's L2 instead of being fetched PASSES times over.
    // Sweep q of workgroup (xcd, idx): tile q * (gridDim.x / PASSES) + (idx / PASSES) * 8 + xcd, group idx % PASSES.
    const bool xmap = !listed && PASSES > 1 && a.col_xmap != 0;
    const int x_gp = (int)gridDim.x / PASSES;
    const int x_t0 = (((int)blockIdx.x >> 3) / PASSES) * 8 + ((int)blockIdx.x & 7), x_p = ((int)blockIdx.x >> 3) % PASSES;
    const bool lmap = listed && PASSES > 1 && a.list_xmap != 0;
    const int l_lim = (n_grp - x_p + PASSES - 1) / PASSES;            // lmap: 4-column runs of the list that hold a group x_p
    // list group of sweep q
    auto grp_of = [&](int q) -> int { return lmap ? (q * x_gp + x_t0) * PASSES + x_p : (int)blockIdx.x + q * (int)gridDim.x; };
    const int ncols = listed ? (lmap ? (x_t0 < l_lim ? (l_lim - x_t0 + x_gp - 1) / x_gp : 0)
                                     : ((int)blockIdx.x < n_grp ? (n_grp - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0))
                             : xmap ? (x_t0 < ntiles ? (ntiles - x_t0 + x_gp - 1) / x_gp : 0)
                             : my_tiles * PASSES;
    auto col_valid = [&](int q) { return !listed || grp_of(q) * CPAR + cpar < n_act; };
    R acc_w = 0;
    const R nogv = cp.nog != nullptr ? cp.nog[b] : (R)0;
    double* stat_slot = scratch + 16 + (tid >> 6) * STAT_N;
    StatAcc<R> sacc;
    double stat_at = 0;
    if constexpr (STATS) {
        sacc.clear();
        stat_at = 1.0 / a.tsum[b];
        StatAcc<R>::slot_init(stat_slot);
    }

    Cx<R> v[16], gn[LEAN ? 1 : 16];
    R wr[16], tr[16];
    // PHASE 2 (fixed farfield phase, the steady state of WGS-Kim), fp32: the stored phase of a column arrives with its weights
    // and targets, a column ahead (16 more registers, 189 -> ~205 of 256), instead of being fetched pixel by pixel inside the
    // constraint -- on a column list (engine default) that fetch sat on the ~12 us chain of the launch (round 5)
    constexpr bool PF_AHEAD = PHASE == 2 && !LEAN && HGS_PF_AHEAD && HGS_LANE_MAJOR;
    float4 pfq0 = make_float4(0, 0, 0, 0), pfq1 = pfq0, pfq2 = pfq0, pfq3 = pfq0;      // (four 16-byte registers, not an array: R[16] left 12 bytes on the stack)

    // (list launches: the entry of sweep q is asked for three times per column -- at the top of the loop, by the weight /
    //  target prefetch and by the G prefetch of the column after it.  As a plain load of a uniform address it was a VECTOR load
    //  each time (the list is not provably invariant, so no scalar load) with `s_waitcnt vmcnt(0)` behind it: the second one
    //  waited for the weight stores just issued, the third for the weight / target prefetch just requested -- which was
    //  meant to land under the inverse transform.  Where a lane group is whole waves the entry comes through the scalar cache
    //  (inline `s_load_dword`: lgkmcnt, not vmcnt) and is kept for the next two askers.)
    int col_cache_q = -1, col_cache = 0;
    auto col_of = [&](int q, int& ct, int& c4) {
        if (listed) {
            int col;
            if constexpr (T % 64 == 0 && HGS_LIST_SLOAD) {
                if (q != col_cache_q) { col_cache = uniform_load_i32(clist + min(grp_of(q) * CPAR + cpar, n_act - 1)); col_cache_q = q; }
                col = col_cache;
            } else {
                col = clist[min(grp_of(q) * CPAR + cpar, n_act - 1)];
            }
            ct = col >> 2;
            c4 = col & 3;
            return;
        }
        if (xmap) {
            ct = q * x_gp + x_t0;
            c4 = x_p * CPAR + cpar;
            return;
        }
        ct = blockIdx.x + (q / PASSES) * gridDim.x;
        c4 = (q % PASSES) * CPAR + cpar;
    };
    auto issue_wt = [&](int q) {
        int ct, c4;
        col_of(q, ct, c4);
        const size_t cb = (size_t)b * P + (size_t)(ct * 4 + c4) * g.Ph;
        const R* wc = a.w + cb;
        const R* tc = a.t + cb;
        // one (wave-uniform where T >= 64) branch around the whole group: a select per element turns every load into
        // its own predicated dword access instead of four 16-byte ones
        const bool need_t = do_upd || STATS || x_mraf;
        // float64: straight-line loads through buffer resources (an empty resource reads as zero), no branch.  With the loads
        // inside uniform branches the compiler does not know at the join how many are in flight and waits for the OLDER loads
        // of the column's G rows with vmcnt(0..3) -- i.e. for the weights and targets just requested as well: the in-order
        // queue then costs a full memory round trip per column exactly where the request was meant to run ahead of the
        // transform (tools/microbench/trace8k f64main: 11 k of 45 k cycles; what HGS_F64_WT_EARLY ran into in round 5).
        if constexpr (LEAN && HGS_LANE_MAJOR && HGS_COL_BUF && T % 64 == 0 && HGS_F64_WT_BUF) {
            constexpr unsigned RB = sizeof(R), PER = 16 / RB;           // values per 16-byte load
            const unsigned bytes = col_valid(q) ? (unsigned)g.Ph * RB : 0u;
            const Buf bw(wc, bytes), bt(tc, need_t ? bytes : 0u);
            const unsigned vo = lane_pos<T>(j, 0) * RB;
            static_for<0, 16 / PER>([&](auto c_) {
                constexpr int c = c_;
                const Cx<R> x = bw.template ld<Cx<R>>(vo + 16u * c, 0u);       // (two doubles: 16 bytes)
                wr[PER * c] = x.x; wr[PER * c + 1] = x.y;
            });
            static_for<0, 16 / PER>([&](auto c_) {
                constexpr int c = c_;
                const Cx<R> x = bt.template ld<Cx<R>>(vo + 16u * c, 0u);
                tr[PER * c] = x.x; tr[PER * c + 1] = x.y;
            });
            return;
        }
        if (col_valid(q)) {
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; wr[m] = wc[lane_pos<T>(j, m)]; });
            if constexpr (PF_AHEAD) {        // (a lane's sixteen values are 64 contiguous bytes: lane_pos<T>(j, m) = 16 j + m)
                const float4* pc = reinterpret_cast<const float4*>(a.pff + cb + lane_pos<T>(j, 0));
                pfq0 = pc[0]; pfq1 = pc[1]; pfq2 = pc[2]; pfq3 = pc[3];
            }
            if (need_t) static_for<0, 16>([&](auto m_) { constexpr int m = m_; tr[m] = tc[lane_pos<T>(j, m)]; });
            // (defined on every path: left unset here, the RULE 2 instances kept one of them in scratch -- 12 bytes, a
            //  scratch round trip per column)
            else static_for<0, 16>([&](auto m_) { constexpr int m = m_; tr[m] = (R)0; });
        } else {
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; wr[m] = (R)0; tr[m] = (R)0; });
            if constexpr (PF_AHEAD) pfq0 = pfq1 = pfq2 = pfq3 = make_float4(0, 0, 0, 0);
        }
    };
    // rows outside the SLM (and whole columns past the end of a list) through the range check of a buffer resource
    // where the column of a lane is wave-uniform: 16 straight-line loads / stores instead of 16 branches
    constexpr bool GBUF = HGS_COL_BUF && T % 64 == 0;
    constexpr unsigned CB = sizeof(Cx<R>);
    auto g_buf = [&](int q, int ct, int c4) {
        const Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4 + c4;
        return Buf(gh, col_valid(q) ? (unsigned)(g.Sh * 4 - c4) * CB : 0u);
    };
    const unsigned g_voff = (unsigned)r_lane * 4u * CB, g_vstep = (unsigned)T * 4u * CB;    // negative rows wrap out of range
    auto issue_g = [&](int q, Cx<R> (&dst)[16]) {
        int ct, c4;
        col_of(q, ct, c4);
        if constexpr (GBUF) {
            const Buf bg = g_buf(q, ct, c4);
            static_for<0, 16>([&](auto m_) {
                constexpr int m = m_;
                if constexpr (m < NRS) dst[m] = bg.template ld<Cx<R>>(g_voff + (unsigned)m * g_vstep, 0u); else dst[m] = mk<R>(0, 0);
            });
            return;
        }
        const Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4 + c4;
        static_for<0, 16>([&](auto m_) {
            constexpr int m = m_;
            const int r = r_lane + m * T;
            dst[m] = mk<R>(0, 0);
            if constexpr (m < NRS) { if (r >= 0 && r < g.Sh && col_valid(q)) dst[m] = gh[(unsigned)r * 4u]; }
        });
    };

    if (ncols > 0) {
        issue_g(0, v);
        if constexpr (!LEAN) issue_wt(0);
    }
#pragma unroll 1
    for (int q = 0; q < ncols; ++q) {
        int ct, c4;
        col_of(q, ct, c4);
        const size_t cb = (size_t)b * P + (size_t)(ct * 4 + c4) * g.Ph;
        const bool vcol = col_valid(q);
        HGS_T(fft.tr_n, 1);
#if HGS_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        HGS_T(fft.tr_n, 2);
        static_for<0, NRS>([&](auto m_) { constexpr int m = m_; v[m] = v[m] * sgs; });
        // fp64: this column's weights / targets land under its forward transform (8192: after it -- the 64 registers
        // would not fit next to the transform's own)
        if constexpr (LEAN && (N < 8192 || (SHIFTED && (HGS_F64_WT_EARLY || LTW)))) issue_wt(q);
        // 8192 points: the 64 registers of this column's weights / targets do not fit next to the forward transform, so they
        // are requested after it.  (Experiment, off: one word of each of the lane's two 128-byte lines requested BEFORE the
        // transform, so that the real loads are L2 hits -- the launch got 3 % slower, HGS_F64_WT_PREFETCH.)
        int pf_w = 0, pf_t = 0;
        if constexpr (LEAN && N >= 8192 && HGS_F64_WT_PREFETCH) {
            if (vcol) {
                pf_w = reinterpret_cast<const int*>(a.w + cb + lane_pos<T>(j, 0))[0];
                if (do_upd || STATS || x_mraf) pf_t = reinterpret_cast<const int*>(a.t + cb + lane_pos<T>(j, 0))[0];
            }
        }
        if constexpr (SHIFTED) fft.template fwd_lead<NRS>(v, lds, j);       // slots NRS.. are zero (rows outside the SLM)
        else fft.fwd(v, lds, j);
        if constexpr (LEAN && N >= 8192 && HGS_F64_WT_PREFETCH) asm volatile("" :: "v"(pf_w), "v"(pf_t));
        HGS_T(fft.tr_n, 3);
        // fp64: the 16 transformed values of a lane (64 VGPRs) wait in the idle LDS image while the constraint runs --
        // lane-private slots [m * T + j], conflict-free, no barrier -- so that the rule (inlined double log2 / exp2,
        // atan2, sincos) does not sit on top of them: with v, weights and targets all in registers every fp64
        // instantiation spilled 137 .. 483 VGPRs to scratch
        Cx<R>* park = lds + j;
        if constexpr (LEAN) {
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; park[m * T] = v[m]; });
            if constexpr (N >= 8192 && !(SHIFTED && (HGS_F64_WT_EARLY || LTW))) issue_wt(q);
        }
#if HGS_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        HGS_T(fft.tr_n, 4);

        // ---- constraint + weight update on F = sc * v ----
        R* wc = a.w + cb;
        R* pfc = (PHASE != 0) ? a.pff + cb : nullptr;
        bool w_changed = false;
        // fp64: the parked value of pixel m + 1 is requested before pixel m is evaluated (one pixel at a time -- a scheduling
        // barrier per pixel -- otherwise meant one exposed LDS round trip per pixel, sixteen per column)
        Cx<R> vnext = mk<R>(0, 0);
        if constexpr (LEAN && HGS_F64_PARK_AHEAD) vnext = park[0];
        auto cons = [&](auto m_) {
            constexpr int m = m_;
            const unsigned idx = lane_pos<T>(j, m);
            Cx<R> vm;
            if constexpr (LEAN && HGS_F64_PARK_AHEAD) {
                vm = vnext;
                if constexpr (m + 1 < 16) vnext = park[(m + 1) * T];
            } else if constexpr (LEAN) vm = park[m * T]; else vm = v[m];
            [&]() {
            // wave-uniform skip of pixels with zero weight and zero target (see col_tile_kernel)
            if (PHASE != 1 && !(STATS && (a.do_stats & 2)) && HGS_SPARSE_SKIP &&
                __builtin_amdgcn_ballot_w64(wr[m] != (R)0 || ((do_upd || STATS || x_mraf) && tr[m] != (R)0)) == 0) {
                vm = mk<R>(0, 0);
                if (x_nog && vcol) acc_w += (R)1;      // T == 0 -> fc = 1 (:1841)
                return;
            }
            Cx<R> F;
            if constexpr (SHIFTED) F = cmul(vm, omu) * a.scale; else F = vm * sc;
            const R p2 = F.x * F.x + F.y * F.y;
            if (x_nog) {                                    // Nogrette: sum of fc = feedback / target over all pixels
                if (vcol) acc_w += nogrette_fc<R>(M::sqrt(p2) * cp.inv_fnorm, tr[m]);
                vm = mk<R>(0, 0);
                return;
            }
            const R wraw = wr[m];
            R wv = wraw * wsc;
            if (do_upd) {
                const R t = tr[m];
                if (RULE == 1 || cp.method == M_LEONARDO || cp.method == M_KIM) {
                    // evaluated for every lane and selected (a branch per pixel splits the pass into 16 blocks):
                    // T == 0 -> factor 1 (:1841); inf (:1840,:1867) and nan (:1843) -> 1
                    // (fp64: the rule is some hundred instructions of double log2 / exp2 -- worth the branch)
                    // (float64 too since round 6: pow_lean is 45 operations, and without the per-lane branch the pixels of a group
                    //  are one basic block whose dependent chains the scheduler can interleave, HGS_CONS_GROUP_F64)
                    if (sizeof(R) == 4 || HGS_F64_EAGER || t != (R)0) {
                        R fc = leonardo_factor<R>(p2, t, cp.inv_fnorm, cp.p_exp);
                        fc = (t != (R)0 && fc < (R)INFINITY) ? fc : (R)1;
                        wv *= fc;
                    }
                } else {
                    wv *= weight_factor<R>(cp.method, M::sqrt(p2) * cp.inv_fnorm, t, cp.p_exp, cp.p_fac, nogv);
                }
                if (is_nan(wv)) wv = (R)0.0001;            // :1873
                if (x_presum) {                            // pre-pass: what this update adds to sum w^2 -- nothing is kept,
                    const double w0 = (double)wraw * (double)wsc;      // nothing rebuilt (float64: the phasor alone is a double rsqrt)
                    acc_w += (R)((double)wv * (double)wv - w0 * w0);   // (each term in double; a lane adds a few hundred of them)
                    vm = mk<R>(0, 0);
                    return;
                } else {
                    w_changed |= (wv != wraw);             // stored after the loop; unchanged lanes (zeros of a
                    wr[m] = wv;                            // sparse target) write nothing
                    acc_w += wv * wv;
                }
            }
            if constexpr (STATS) {
                const R af = M::sqrt(p2);
                if ((a.do_stats & 2) && vcol) a.amp_ff[cb + idx] = af;
                sacc.add(p2, af, tr[m], stat_at, a.inv_fsum);
            }
            R co, si;
            if constexpr (PHASE == 2) {
                if constexpr (PF_AHEAD) {
                    const float4 qv = (m / 4 == 0) ? pfq0 : (m / 4 == 1) ? pfq1 : (m / 4 == 2) ? pfq2 : pfq3;
                    M::sincos_phase((R)((m % 4 == 0) ? qv.x : (m % 4 == 1) ? qv.y : (m % 4 == 2) ? qv.z : qv.w), &si, &co);
                }
                else M::sincos_phase(pfc[idx], &si, &co);
            } else {
                if (sizeof(R) == 4 || HGS_F64_EAGER || p2 > (R)0) {         // exp(i*atan2(F)) == F/|F|; atan2(0,0) = 0 (quirk A6)
                    const R inv = rsqrt_full(p2);
                    co = (p2 > (R)0) ? F.x * inv : (R)1;
                    si = (p2 > (R)0) ? F.y * inv : (R)0;
                } else {
                    co = 1;
                    si = 0;
                }
                if constexpr (PHASE == 1) { if (vcol) pfc[idx] = M::atan2(F.y, F.x); }
            }
            // inverse-transform input = (-1)^k * ff (SHIFTED: times the conjugate shift factor)
            // (snew: 1, or the new weights' final normalisation where a pre-pass has summed it -- the stored weight stays raw)
            const R wvs = (RULE == 0) ? wv * (R)*snew_slot : wv;
            if constexpr (SHIFTED) vm = cmulc(mk<R>(co, si), omu) * wvs; else vm = mk<R>(wvs * co * sgn, wvs * si * sgn);
            if (x_mraf) {                                   // mixed-region amplitude freedom (:1606-1653)
                const R t = tr[m];
                if (is_nan(t)) {
                    if (x_split) {                          // the noise part goes its own way (CParams::split)
                        if (vcol) a.ffb[cb + idx] = cp.has_mraf_factor ? F * cp.mraf_factor : F;
                        vm = mk<R>(0, 0);
                    } else {
                        if constexpr (SHIFTED) {
                            vm = cmulc(cp.has_mraf_factor ? F * cp.mraf_factor : F, omu);
                        } else {
                            const R mf = cp.has_mraf_factor ? cp.mraf_factor * sgn : sgn;
                            vm = F * mf;
                        }
                    }
                } else if (t == (R)0) {
                    vm = mk<R>(0, 0);
                    if constexpr (PHASE == 1) { if (vcol) pfc[idx] = (R)0; }
                }
            }
            }();
            if constexpr (LEAN) park[m * T] = vm; else v[m] = vm;
#if HGS_TRACE_CONS
            HGS_T(fft.tr_n, 40 + m);
#endif
            // (fp64: one pixel at a time -- four interleaved double atan2 / sincos / log2 chains cost 100+ registers)
            if constexpr (m % (sizeof(R) == 4 ? HGS_CONS_GROUP : HGS_CONS_GROUP_F64) == (sizeof(R) == 4 ? HGS_CONS_GROUP : HGS_CONS_GROUP_F64) - 1) __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (!LEAN) {
            static_for<0, 16>(cons);
            if constexpr (STATS) sacc.flush(stat_slot);
            if (do_upd && w_changed && vcol) {
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; wc[lane_pos<T>(j, m)] = wr[m]; });
            }
        } else {
            // fp64: updated weights leave four pixels (32 bytes) at a time, so that they do not all stay live
            static_for<0, 4>([&](auto g_) {
                constexpr int g0 = 4 * decltype(g_)::value;
                w_changed = false;
                static_for<0, 4>([&](auto i_) { cons(std::integral_constant<int, g0 + decltype(i_)::value>{}); });
                if (do_upd && w_changed && vcol) {
                    static_for<0, 4>([&](auto i_) { constexpr int m = g0 + i_; wc[lane_pos<T>(j, m)] = wr[m]; });
                }
            });
            if constexpr (STATS) sacc.flush(stat_slot);
        }
        HGS_T(fft.tr_n, 5);
        // ---- prefetch the next column while this one is transformed back ----
        if (q + 1 < ncols) {
            if constexpr (!LEAN) {
                issue_wt(q + 1);
                issue_g(q + 1, gn);
            }
        }
        if constexpr (LEAN) {
            // back from the parking slots; whatever comes next -- the inverse, or (forward-only passes) the next
            // column's forward transform -- scatters into other lanes' slots
            if (!x_wonly) static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = park[m * T]; });
            __syncthreads();
        }
        if (!x_wonly) {
            if constexpr (SHIFTED) fft.template inv_after_fwd_trail<NRS>(v, lds, j);     // slots NRS.. are not stored
            else fft.inv_after_fwd(v, lds, j);
            if constexpr (GBUF) {
                const Buf bg = g_buf(q, ct, c4);
                static_for<0, NRS>([&](auto m_) { constexpr int m = m_; bg.template st<Cx<R>>(v[m] * scs, g_voff + (unsigned)m * g_vstep, 0u); });
            } else {
                Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4 + c4;
                static_for<0, NRS>([&](auto m_) {
                    constexpr int m = m_;
                    const int r = r_lane + m * T;
                    if (r >= 0 && r < g.Sh && vcol) gh[(unsigned)r * 4u] = v[m] * scs;
                });
            }
        }
        HGS_T(fft.tr_n, 6);
        if constexpr (!LEAN) {
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = gn[m]; });
        } else if (q + 1 < ncols) {
            issue_g(q + 1, v);
        }
    }
    HGS_T(fft.tr_n, 7);
    if constexpr (STATS) StatAcc<R>::slot_store(stat_slot, a.spartial, b);
    if (do_upd) {
        const double s = block_sum((double)acc_w, scratch);
        if (tid == 0) a.wpartial[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
#if HGS_TRACE
    __syncthreads();
    {   // dump this workgroup's events (128 per wave) through fpartial
        const int nev = ((int)blockDim.x >> 6) * 128;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.fpartial) + (size_t)blockIdx.x * nev;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + HGS_TRACE_OFF);
        for (int i = tid; i < nev; i += blockDim.x) dst[i] = src[i];
    }
#endif
}

// =====================================================================================================
// FUSED column kernel, tile-resident form (N >= 4096, fp32): same math as col_fused_kernel, but the
// whole 4-column tile of GH (only its NR*T candidate rows) is loaded ONCE with 32-byte-per-lane
// accesses into registers, the four columns are transformed from/to those registers, and the tile
// is stored back with 32-byte accesses -- every GH byte crosses the memory system once per kernel.
//
// The SLM rows start at r0.  A circular shift of the transform input by s rows makes the occupied
// register slots of the load layout 0..NR-1 for every geometry (static register indices); by the
// shift theorem it multiplies output k = j + m T by exp(-2 pi i k s / N), and for every s that is a
// multiple of 16 that is a per-lane constant (m T s / N = m s / 16 is an integer), folded into the
// scale multiply.  s = r0 rounded down to a multiple of 16 (round 5; rounds 2 - 4 shifted by whole
// slots, s = (r0 / T) T): the SLM rows then start in the first 16 rows of slot 0 and occupy
// ceil((r0 % 16 + Sh) / T) slots -- 1152 rows on 4096: 5 instead of 6, on 8192: 3 instead of 4.
// =====================================================================================================
// one tile row = 4 adjacent columns = 32 bytes (fp32): two 16-byte accesses per lane
__device__ __forceinline__ void load_row4(const Cx<float>* p, Cx<float> (&o)[4]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1];
    o[0] = mk<float>(a.x, a.y); o[1] = mk<float>(a.z, a.w);
    o[2] = mk<float>(b.x, b.y); o[3] = mk<float>(b.z, b.w);
}
__device__ __forceinline__ void store_row4(Cx<float>* p, const Cx<float> (&o)[4]) {
    float4* q = reinterpret_cast<float4*>(p);
    q[0] = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
    q[1] = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
}

// issue (do not wait for) the weight / target loads of one column into registers
template <typename R, int T>
__device__ __forceinline__ void issue_wt_loads(const R* __restrict__ wc, const R* __restrict__ tc, bool upd, int j,
                                               R (&wr)[16], R (&tr)[16]) {
    static_for<0, 16>([&](auto m_) {
        constexpr int m = m_;
        if (HGS_ABL_WT == 1) {
            wr[m] = (R)1e-3;
            tr[m] = (j == 7 && m == 3) ? (R)0.03 : (R)0;
            return;
        }
        if (HGS_ABL_WT == 2) {      // a column without spots (what all but 32 columns of cfg 2 look like): skipped rule
            wr[m] = (R)0;
            tr[m] = (R)0;
            return;
        }
        wr[m] = wc[lane_pos<T>(j, m)];
        tr[m] = upd ? tc[lane_pos<T>(j, m)] : (R)0;
    });
}

// an empty asm statement that names the registers: whatever request fills them is waited for HERE (col_tile_kernel TOUCH)
template <typename R, bool BOTH>
__device__ __forceinline__ void touch_regs16(R (&x)[16], R (&y)[16]) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        asm volatile("" : "+v"(x[m]));
        if constexpr (BOTH) asm volatile("" : "+v"(y[m]));
    }
}

// EXTRAS = false compiles the MRAF / Nogrette / forward-only branches out (2.7 us of the 58 us dense launch)
// dynamic LDS of col_tile_kernel: transform image + reduction scratch
// (8192 points: one workgroup per CU, so nobody covers the 5.8 k cycles a tile's rows take to arrive; the NEXT tile of
//  the workgroup is staged global -> LDS meanwhile -- TILE_PREF_SLOTS register slots of 32 bytes per lane)
constexpr int TILE_PREF_SLOTS = 4;
template <typename R, int N> constexpr size_t col_tile_pref_bytes() { return N >= 8192 ? (size_t)TILE_PREF_SLOTS * 2 * (N / 16) * 16 : 0; }
template <typename R, int N> constexpr size_t col_tile_lds_bytes() {
    return lds_elems<N>() * sizeof(Cx<R>) + SCRATCH_DOUBLES * sizeof(double) + col_tile_pref_bytes<R, N>();
}
// RULE 3: no staging of the next tile; the noise part of the column is parked behind the scratch instead (N elements + a flag)
template <typename R, int N> constexpr size_t col_tile_split_lds_bytes() {
    return lds_elems<N>() * sizeof(Cx<R>) + SCRATCH_DOUBLES * sizeof(double) + (size_t)N * sizeof(Cx<R>) + 16;
}

// RULE: 0 = method and update switch read from CParams (a chain of uniform branches per pixel: five per evaluated
// pixel of a spot column, four of them taken); 1 = the WGS-Leonardo / WGS-Kim update compiled in; 2 = no weight update
// (GS, iteration 0, the second pass of MRAF).  The hot launches use 1 / 2 (launch_tile_rule).
// 3 = MRAF with a weight update in ONE pass (EXTRAS, rule as 0): the rebuilt field mixes the normalised new weights (signal
// region) with the kept farfield (noise region), and ||w'|| is only known when every column is through -- but the inverse
// transform is linear.  The signal part A = w' e^{i phi} (un-normalised) and the noise part B = mraf_factor F are transformed
// separately (B only for columns that hold noise pixels), stored to gh / gh2, and the row kernel (SPLIT) forms A / ||w'|| + B.
// One forward transform, one read of the column's weights and target and one of GH less than the two-pass form.
// 5 = MRAF with the WGS-Leonardo / WGS-Kim update and ONE inverse per column (round 6): the weights that enter an update are
// normalised (wscale folds the previous ||w||), so ||w'||^2 = 1 + D with D = sum over the SIGNAL pixels (finite non-zero
// target: everywhere else the factor is 1) of w'^2 - w^2 -- and col_presum_kernel has formed D from a forward-only pass over
// the columns that hold signal pixels (a quarter of them at cfg 5) before this launch.  The field is rebuilt with the final
// scale: no second inverse in the columns that hold noise, no parked noise part in LDS (the next tile is staged again), no
// second array for the row kernel to join.
// 6 = MRAF without a weight update, compiled per slot count like 5 (GS on an MRAF target: 240 -> 2xx us at cfg 5).
// LISTED: -1 = the tile schedule is decided at run time (a.col_list), 0 / 1 = compiled in (the hot dense launches lose
// 0.4 us of 51.5 with the run-time form).
template <typename R, int N, int PHASE, int NR, bool STATS = false, bool EXTRAS = true, int RULE = 0, int LISTED = -1>
__global__ __launch_bounds__(N / 16, HGS_FUSED_OCC) void col_tile_kernel(ColArgs<R> a, int shift) {
    using M = Math<R>;
    constexpr int T = N / 16;
    static_assert(T >= 256, "tile-resident kernel is for one column per workgroup pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int j = threadIdx.x;
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem);
    double* scratch = reinterpret_cast<double*>(lds + lds_elems<N>());

    constexpr bool TPREF = N >= 8192;
    constexpr bool SPLIT = RULE == 3 || RULE == 4;     // 4: ... with the WGS-Leonardo / WGS-Kim update compiled in, MRAF on, no
    constexpr bool FIXED = RULE == 4 || RULE == 5 || RULE == 6;     //    Nogrette sum, not forward-only (the cfg 5 launch)
    constexpr bool NOUPD = RULE == 6;                  // 6: MRAF compiled in, NO weight update (GS, iteration 0 of a WGS run, the steps of a
                                                       //    callback loop between updates): the generic form ran the six-slot instance
    constexpr bool PRESUM = RULE == 5;                 // 5: the same rule, the field rebuilt with the pre-summed 1 / ||w'||
    static_assert(!(SPLIT || PRESUM || RULE == 6) || EXTRAS, "col_tile_kernel: RULE 3 / 4 / 5 / 6 are EXTRAS forms");
    using Sel = FftSel<R, N, true, TPREF>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    const CParams<R> cp = a.cp;
    const bool do_upd = NOUPD ? false : (RULE == 1 || FIXED) ? true : (RULE == 2 ? false : cp.do_update != 0);
    const int js = Sel::space_lane(j);       // rows js + m*T (space side), farfield pixels j + m*T (frequency side)
    const R sgn = (j & 1) ? (R)-1 : (R)1;
    const R sgs = (js & 1) ? (R)-1 : (R)1;
    const size_t P = (size_t)g.Ph * g.Pw;
    const R wsc = a.wscale[b];
    // shift-theorem factor of this lane, with the (-1)^k sign and the ortho scale folded in
    Cx<R> om = a.tw[(j * shift) & (N - 1)];          // (shift: rows, a multiple of 16)
    om = om * (sgn * a.scale);
    const int r_lane = js + shift - g.r0;   // SLM row of slot m is r_lane + m*T
    // tile schedule: every tile of 4 columns, or (col_list != nullptr: sparse targets whose active set the host has
    // rounded to whole tiles) the tiles of the listed columns -- entries 4 i .. 4 i + 3 of the list are tile list[4 i] / 4
    const bool listed = LISTED < 0 ? a.col_list != nullptr : LISTED != 0;
    const int* clist = listed ? a.col_list + (size_t)b * g.Pw : nullptr;
    const int ntiles = listed ? (a.n_active[b] >> 2) : g.Pw / 4;
    // (list entries through the scalar cache: as plain loads they were vector loads with a drain behind each, two per tile)
    auto tile_of = [&](int it) -> int { return listed ? (uniform_load_i32(clist + 4 * it) >> 2) : it; };
    R acc_w = 0;

    Cx<R> v[16];
    R gtx[NR][4], gty[NR][4];   // the tile (scalar arrays: arrays of 2-vectors are not promoted to registers)
    // SPLIT: the noise part's tile.  In registers when the SLM rows occupy <= 4 slots (8-byte stores per column reach HBM
    // as four read-modify-writes of every 32-byte tile row: measured +0.6 GB per pass at 8192^2); six slots do not fit.
    constexpr bool BTILE = SPLIT && NR <= 4;
    R gbx[BTILE ? NR : 1][4], gby[BTILE ? NR : 1][4];
    R wr[16], tr[16];

    const bool upd = do_upd || STATS || (EXTRAS && (FIXED || cp.mraf != 0));   // target needed by the update, the statistics, MRAF
    const R nogv = cp.nog != nullptr ? cp.nog[b] : (R)0;
    double* stat_slot = scratch + 16 + (j >> 6) * STAT_N;
    StatAcc<R> sacc;
    double stat_at = 0;
    if constexpr (STATS) {
        sacc.clear();
        stat_at = 1.0 / a.tsum[b];
        StatAcc<R>::slot_init(stat_slot);
    }
    const R* wbase = a.w + (size_t)b * P;
    const R* tbase = a.t + (size_t)b * P;
    // PRESUM: every workgroup folds the pre-pass' partials itself (a few hundred doubles, fixed order: all workgroups get
    // the same bits) instead of waiting for one more launch
    R snew = 1;
    constexpr bool PRESUM_RT = EXTRAS && RULE == 0;     // the generic form behind a per-column pre-pass (a.dpartial set; Pw < 4096)
    bool presum_on = PRESUM;
    if constexpr (PRESUM_RT) presum_on = a.dpartial != nullptr;
    if (presum_on) {
        double d = 0;
        for (int i = j; i < a.n_dpartial; i += T) d += a.dpartial[(size_t)b * a.n_dpartial + i];
        d = block_sum(d, scratch);
        if (j == 0) scratch[0] = 1.0 / ::sqrt(1.0 + d);
        __syncthreads();
        snew = (R)scratch[0];
        __syncthreads();
    }

    // TPREF: staging image of the workgroup's next tile, wave-private 1 KiB blocks [slot][half][wave][lane * 16 bytes];
    // used when the SLM rows fit the first TILE_PREF_SLOTS register slots (uniform)
    char* pstage = reinterpret_cast<char*>(scratch + SCRATCH_DOUBLES);
    const bool tpref = TPREF && !SPLIT && (TILE_PREF_SLOTS * T + shift - g.r0 >= g.Sh);
    // SPLIT: the staging space holds the noise part of the column (lane-private: element m of lane j at m T + j) and a flag
    Cx<R>* park = reinterpret_cast<Cx<R>*>(pstage);
    int* nflag = reinterpret_cast<int*>(pstage + (size_t)N * sizeof(Cx<R>));
    const int wv = __builtin_amdgcn_readfirstlane(j >> 6);
    auto stage_next = [&](int nct) {
        if constexpr (TPREF) {
            const Cx<R>* ghn = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)nct * g.Sh * 4;
#pragma unroll
            for (int m = 0; m < TILE_PREF_SLOTS; ++m) {
                const int r = r_lane + m * T;
                if (r >= 0 && r < g.Sh) {
                    glds16(ghn + (unsigned)r * 4u, pstage + ((m * 2 + 0) * (T / 64) + wv) * 1024);
                    glds16(ghn + (unsigned)r * 4u + 2, pstage + ((m * 2 + 1) * (T / 64) + wv) * 1024);
                }
            }
        }
    };
    // TOUCH (8192 rows): where the compiler waits for a column's weights and targets.  Left alone it waits at the HEAD of the column
    // (the sparse-skip tests are hoisted there, and the loop header joins paths with different numbers of younger requests, so the
    // wait is vmcnt(0)) -- which, at the first column of a tile, also waits for the staging requests of the NEXT tile issued a few
    // instructions earlier and for the tile stores before them: a full memory round trip per tile that the staging was there to
    // hide (s_memtime: "tile start -> tile landed" 3.9 k + 1.4 k cycles of a tile's 58 k).  An empty asm statement that names the
    // registers at the END of a column puts the wait there, an inverse transform after the requests, and leaves the head without one.
    // (measured, one GPU call, two runs each, with and without: plain rule instance at cfg5pad 201.9 -> 198.2 us; the MRAF form without
    //  an update 201.6 -> 201.2; the MRAF form WITH the update 212.9 -> 214.7 -- left as it was.  Builds that differ in nothing but the
    //  form of an unrelated loop move these kernels by +- 1.5 %, so only the first figure says much.  The rows of a workgroup's first
    //  tile as straight-line buffer loads on top of it: +6 VGPRs and 2 % slower everywhere, not kept.)
    constexpr bool TOUCH = HGS_TILE_TOUCH && TPREF && !SPLIT && RULE != 5;
    constexpr bool TR_DEAD = RULE == 2 && !STATS && !EXTRAS;      // (the target is never read: its registers are constants)
    auto touch_wt = [&]() { touch_regs16<R, !TR_DEAD>(wr, tr); };
#pragma unroll 1
    for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
        HGS_T(fft.tr_n, 1);
        const int ct = tile_of(it);
        if constexpr (TOUCH) {       // the first tile's weights / targets ahead of its rows (one round trip for both)
            if (it == (int)blockIdx.x)
                issue_wt_loads<R, T>(wbase + (size_t)(ct * 4) * g.Ph, tbase + (size_t)(ct * 4) * g.Ph, upd, j, wr, tr);
        }
        // this workgroup's next tile, `more` = there is one (LISTED 0 keeps the plain arithmetic of the dense schedule: the
        // hot launches are sensitive to the form of these scalar expressions, 0.6 us of 51.5)
        int ct_nl = -1;
        if constexpr (LISTED != 0) {
            const int it_n = it + (int)gridDim.x;
            ct_nl = it_n < ntiles ? tile_of(it_n) : -1;
        }
        auto next_ct = [&]() -> int {
            if constexpr (LISTED == 0) return ct + (int)gridDim.x;
            else return ct_nl;
        };
        auto more = [&](int nct) -> bool { return LISTED == 0 ? nct < ntiles : nct >= 0; };
        Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4;
        const bool staged = tpref && it != (int)blockIdx.x;
        if constexpr (TPREF) {
            // this wave's own pieces (no other wave reads them).  (They were issued a whole tile ago and every column since has
            // waited for weights / targets requested after them, so the wait is formally redundant -- and it also waits for
            // the tile stores issued just before; without it the launch measured the same, HGS_TILE_STAGE_WAIT.)
            // (TOUCH: the wait at the end of the previous column covered them -- they are older than that column's weight requests)
            if (HGS_TILE_STAGE_WAIT && !TOUCH && staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        auto put_row = [&](auto m_, const float4& lo, const float4& hi) {
            constexpr int m = m_;
            gtx[m][0] = lo.x; gty[m][0] = lo.y; gtx[m][1] = lo.z; gty[m][1] = lo.w;
            gtx[m][2] = hi.x; gty[m][2] = hi.y; gtx[m][3] = hi.z; gty[m][3] = hi.w;
        };
        static_for<0, NR>([&](auto m_) {
            constexpr int m = m_;
            const int r = r_lane + m * T;
            float4 lo = make_float4(0, 0, 0, 0), hi = lo;
            if (TPREF && staged) {
                if (m < TILE_PREF_SLOTS && r >= 0 && r < g.Sh) {
                    lo = *reinterpret_cast<const float4*>(pstage + ((m * 2 + 0) * (T / 64) + wv) * 1024 + (j & 63) * 16);
                    hi = *reinterpret_cast<const float4*>(pstage + ((m * 2 + 1) * (T / 64) + wv) * 1024 + (j & 63) * 16);
                }
            } else
            if (HGS_ABL_GH) {
                lo = make_float4((float)j * 1e-4f, (float)m, 0.5f, (float)ct * 1e-3f);
                hi = make_float4(0.25f, (float)j * 2e-4f, (float)m * 0.1f, 1.f);
            } else
            if (r >= 0 && r < g.Sh) {
                const float4* q = reinterpret_cast<const float4*>(gh + (unsigned)r * 4u);
                lo = q[0];
                hi = q[1];
            }
            put_row(m_, lo, hi);
        });
        if constexpr (TPREF) {
            if (tpref && more(next_ct())) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the image has been read into registers
                stage_next(next_ct());
            }
        }
        // SPLIT: the LDS the staging would use holds the parked noise part.  (Experiment, off: one word of each row piece of
        // the workgroup's NEXT tile requested now, so that its rows are L2 hits -- no gain, HGS_SPLIT_L2_PREFETCH.)
        int pf_next[NR];
        if constexpr (SPLIT && TPREF && NR <= 4 && HGS_SPLIT_L2_PREFETCH) {     // (six slots: no registers to spare)
#pragma unroll
            for (int m = 0; m < NR; ++m) pf_next[m] = 0;
            if (more(next_ct())) {
                const Cx<R>* ghn = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)next_ct() * g.Sh * 4;
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const int r = r_lane + m * T;
                    if (r >= 0 && r < g.Sh) pf_next[m] = reinterpret_cast<const int*>(ghn + (unsigned)r * 4u)[0];
                }
            }
        }
        int tile_noise = 0;          // SPLIT: any column of this tile with a noise pixel (wave-uniform)
        // PRESUM / NOUPD over a column list: the scan bytes of the tile's four columns, one aligned word through the scalar cache,
        // once per tile (as a plain load inside the column loop it was a vector load with a drain behind it at the head of every column)
        int cflags4 = -1;
        if constexpr (PRESUM || NOUPD) {
            if (a.col_flags != nullptr) cflags4 = uniform_load_i32(reinterpret_cast<const int*>(a.col_flags + (size_t)b * g.Pw) + ct);
        }
        if constexpr (TOUCH) {
            if (it == (int)blockIdx.x) touch_wt();
        } else
        if (it == (int)blockIdx.x)   // later tiles were prefetched at the end of the previous one
            issue_wt_loads<R, T>(wbase + (size_t)(ct * 4) * g.Ph, tbase + (size_t)(ct * 4) * g.Ph, upd, j, wr, tr);
#if HGS_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        HGS_T(fft.tr_n, 2);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            const size_t cb = (size_t)b * P + (size_t)(ct * 4 + c) * g.Ph;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m < NR) {
#if HGS_TILE_MOVREL
                    R xr = gtx[m < NR ? m : 0][c], xi = gty[m < NR ? m : 0][c];     // (uniform index: register-relative move)
#else
                    R xr = gtx[m < NR ? m : 0][0], xi = gty[m < NR ? m : 0][0];
#pragma unroll
                    for (int cc = 1; cc < 4; ++cc) {
                        xr = (c == cc) ? gtx[m < NR ? m : 0][cc] : xr;
                        xi = (c == cc) ? gty[m < NR ? m : 0][cc] : xi;
                    }
#endif
                    v[m] = mk<R>(xr * sgs, xi * sgs);
                } else {
                    v[m] = mk<R>(0, 0);
                }
            }
            // SPLIT: does any lane of the workgroup meet a noise pixel in this column?  Reset here: every reader of the
            // previous column's flag is behind a barrier of that column's transforms, every writer of this one's is
            // behind the barriers of the forward transform below.
            if constexpr (SPLIT) { if (j == 0) *nflag = 0; }
            bool noise_any = false;
            // RULE 4 with the column flags of the scan (list launches): a column without a finite non-zero target has an
            // all-zero signal part, one without a NaN target an all-zero noise part -- known before the column is touched
            // (the tile's four scan bytes as one aligned word through the scalar cache: a uniform BYTE load is a vector memory
            //  instruction with a full round trip and a vmcnt(0) drain behind it)
            int cflags = -1;
            if constexpr (PRESUM || NOUPD) {
                if (a.col_flags != nullptr) cflags = (int)(((unsigned)cflags4 >> (8 * c)) & 0xffu);
            } else if constexpr (FIXED) {      // (RULE 4, the split form over a list: runs once per new set of weights since round 6; left as it was)
                if (a.col_flags != nullptr) cflags = a.col_flags[(size_t)b * g.Pw + ct * 4 + c];
            }
            // (PRESUM: one inverse carries both parts -- skipped only where the column holds neither)
            const bool has_sig = cflags < 0 || (cflags & ((PRESUM || NOUPD) ? 6 : 2)) != 0;
            fft.template fwd_lead<NR>(v, lds, j);     // slots NR.. are zero (rows outside the SLM)

            R* wc = a.w + cb;
            R* pfc = (PHASE != 0) ? a.pff + cb : nullptr;
            bool w_changed = false;
            // phase_ff of this lane's 16 pixels: 64 contiguous bytes, read (PHASE 2) / written (PHASE 1) as such
            HGS_T(fft.tr_n, 3);
#if HGS_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HGS_T(fft.tr_n, 4);
            R pf[PHASE != 0 ? 16 : 1];
            if constexpr (PHASE == 2)
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; pf[m] = pfc[lane_pos<T>(j, m)]; });
            static_for<0, 16>([&](auto m_) {
                constexpr int m = m_;
                const unsigned idx = lane_pos<T>(j, m);
                // Sparse targets (spot arrays): where weight and target are both zero the rule leaves the
                // weight at zero (T == 0 -> factor 1, :1841) and the constrained field is w * e^{i phi} = 0.
                // Skip the arithmetic when that holds for the whole wave (not when phase_ff or amp_ff of
                // every pixel must be produced).
                if (PHASE != 1 && !(STATS && (a.do_stats & 2)) && HGS_SPARSE_SKIP &&
                    __builtin_amdgcn_ballot_w64(wr[m] != (R)0 || tr[m] != (R)0) == 0) {
                    v[m] = mk<R>(0, 0);
                    if constexpr (SPLIT) park[m * T + j] = mk<R>(0, 0);
                    if (EXTRAS && !FIXED && cp.nog_pass) acc_w += (R)1;          // T == 0 -> fc = 1 (:1841)
                    return;
                }
                const Cx<R> F = cmul(v[m], om);
                const R p2 = F.x * F.x + F.y * F.y;
                if (EXTRAS && !FIXED && cp.nog_pass) {                          // Nogrette: sum of fc = feedback / target over all pixels
                    acc_w += nogrette_fc<R>(M::sqrt(p2) * cp.inv_fnorm, tr[m]);
                    v[m] = mk<R>(0, 0);
                    return;
                }
                const R wraw = wr[m];
                R wv = wraw * wsc;
                if (do_upd) {
                    const R t = tr[m];
                    if (RULE == 1 || FIXED || cp.method == M_LEONARDO || cp.method == M_KIM) {
                        R fc = leonardo_factor<R>(p2, t, cp.inv_fnorm, cp.p_exp);     // (eager + select, see col_fused_kernel)
                        fc = (t != (R)0 && fc < (R)INFINITY) ? fc : (R)1;
                        wv *= fc;
                    } else {
                        wv *= weight_factor<R>(cp.method, M::sqrt(p2) * cp.inv_fnorm, t, cp.p_exp, cp.p_fac, nogv);
                    }
                    if (is_nan(wv)) wv = (R)0.0001;
                    w_changed |= (wv != wraw);                 // stored after the loop, 64 contiguous bytes per lane
                    wr[m] = wv;
                    acc_w += wv * wv;
                }
                if constexpr (STATS) {
                    const R af = M::sqrt(p2);
                    if (a.do_stats & 2) a.amp_ff[cb + idx] = af;
                    sacc.add(p2, af, tr[m], stat_at, a.inv_fsum);
                }
                Cx<R> ph;
                if constexpr (PHASE == 2) {
                    R sn, cs;
                    M::sincos_phase(pf[m], &sn, &cs);
                    ph = mk<R>(cs, sn);
                } else {
                    {
                        const R inv = rsqrt_full(p2);
                        ph = mk<R>((p2 > (R)0) ? F.x * inv : (R)1, (p2 > (R)0) ? F.y * inv : (R)0);
                    }
                    if constexpr (PHASE == 1) pf[m] = M::atan2(F.y, F.x);
                }
                // ff = wv * ph; inverse-transform input = (-1)^k * ff * conj(shift factor)
                // (PRESUM: with the new weights' final normalisation -- the stored weight stays un-normalised as on every path)
                if constexpr (PRESUM || PRESUM_RT) v[m] = cmulc(ph, om) * (wv * snew);
                else v[m] = cmulc(ph, om) * wv;
                if (EXTRAS && (FIXED || cp.mraf)) {                              // mixed-region amplitude freedom (:1606-1653)
                    const R t = tr[m];
                    Cx<R> nz = mk<R>(0, 0);
                    if (is_nan(t)) {                        // noise region keeps the field (times mraf_factor)
                        nz = cmulc(cp.has_mraf_factor ? F * cp.mraf_factor : F, om);
                        if constexpr (SPLIT) {
                            v[m] = mk<R>(0, 0);
                            noise_any = true;
                        } else {
                            v[m] = nz;
                        }
                    } else if (t == (R)0) {                 // zero region (no zero_weights feedback on this path)
                        v[m] = mk<R>(0, 0);
                        if constexpr (PHASE == 1) pf[m] = (R)0;      // atan2 of the zeroed field
                    }
                    if constexpr (SPLIT) park[m * T + j] = nz;
                } else if constexpr (SPLIT) {
                    park[m * T + j] = mk<R>(0, 0);
                }
                // PHASE 1: the stored phase leaves four pixels (16 contiguous bytes) at a time -- sixteen of them kept until
                // after the loop put the 4096 / 8192-point RULE 1 instances 12 / 8 registers over their 256
                if constexpr (PHASE == 1 && m % 4 == 3) {
                    static_for<m - 3, m + 1>([&](auto i_) { constexpr int i = i_; pfc[lane_pos<T>(j, i)] = pf[i]; });
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m % (sizeof(R) == 4 ? HGS_CONS_GROUP : 4) == (sizeof(R) == 4 ? HGS_CONS_GROUP : 4) - 1) __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (SPLIT) { if (noise_any) *nflag = 1; }
            if constexpr (STATS) sacc.flush(stat_slot);
            // updated weights of this lane (unchanged lanes -- zeros of a sparse target -- write nothing)
            if (do_upd && w_changed) {
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; wc[lane_pos<T>(j, m)] = wr[m]; });
            }
            // weights/target of the next column (or of the first column of the next tile) land
            // under the inverse transform below and the next forward transform
            {
                const int nct = (c < 3) ? ct : next_ct();
                const int ncol = nct * 4 + ((c + 1) & 3);
                if (more(nct))
                    issue_wt_loads<R, T>(wbase + (size_t)ncol * g.Ph, tbase + (size_t)ncol * g.Ph, upd, j, wr, tr);
            }

            HGS_T(fft.tr_n, 5);
            if (EXTRAS && !FIXED && cp.weights_only) {
                if constexpr (TOUCH) touch_wt();
                continue;
            }
            if (has_sig) fft.template inv_after_fwd_trail<NR>(v, lds, j);      // slots NR.. (rows outside the SLM) are not stored
            else static_for<0, NR>([&](auto m_) { constexpr int m = m_; v[m] = mk<R>(0, 0); });
            if constexpr (TOUCH) touch_wt();
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const Cx<R> h = v[m] * (sgs * a.scale);
#if HGS_TILE_MOVREL
                gtx[m][c] = h.x;
                gty[m][c] = h.y;
#else
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    gtx[m][cc] = (c == cc) ? h.x : gtx[m][cc];
                    gty[m][cc] = (c == cc) ? h.y : gty[m][cc];
                }
#endif
            }
            if constexpr (SPLIT) {
                // the noise part of this column: second inverse transform where there is one, zeros otherwise (the flag's
                // writers are at least one barrier of the inverse above behind)
                Cx<R>* g2 = a.gh2 + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4 + c;
                // (column flags known: no need for the LDS flag, whose writers may not be a barrier behind when the
                //  inverse above was skipped)
                const int any = cflags >= 0 ? ((cflags & 4) != 0) : __builtin_amdgcn_readfirstlane(*nflag);
                tile_noise |= any;
                HGS_T(fft.tr_n, 8);
                if (any) {
                    static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = park[m * T + j]; });
                    fft.template inv_trail<NR>(v, lds, j);
                }
                HGS_T(fft.tr_n, 9);
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const Cx<R> h = any ? v[m] * (sgs * a.scale) : mk<R>(0, 0);
                    if constexpr (BTILE) {
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            gbx[m][cc] = (c == cc) ? h.x : gbx[m][cc];
                            gby[m][cc] = (c == cc) ? h.y : gby[m][cc];
                        }
                    } else {
                        const int r = r_lane + m * T;
                        if ((any || !a.gh2_sparse) && r >= 0 && r < g.Sh) g2[(unsigned)r * 4u] = h;
                    }
                }
            }
        }
        if constexpr (SPLIT && TPREF && NR <= 4 && HGS_SPLIT_L2_PREFETCH) {     // (six slots: no registers to spare)
#pragma unroll
            for (int m = 0; m < NR; ++m) asm volatile("" :: "v"(pf_next[m]));
        }
        if (EXTRAS && !FIXED && cp.weights_only) continue;
        if constexpr (BTILE) {
            Cx<R>* g2 = a.gh2 + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4;
            const bool keep = tile_noise != 0 || !a.gh2_sparse;      // (a tile without noise is never read back)
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const int r = r_lane + m * T;
                if (keep && r >= 0 && r < g.Sh) {
                    float4* q = reinterpret_cast<float4*>(g2 + (unsigned)r * 4u);
                    q[0] = make_float4(gbx[m][0], gby[m][0], gbx[m][1], gby[m][1]);
                    q[1] = make_float4(gbx[m][2], gby[m][2], gbx[m][3], gby[m][3]);
                }
            }
        }
        HGS_T(fft.tr_n, 6);
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const int r = r_lane + m * T;
            if (HGS_ABL_GH ? (gtx[m][0] == 123.456f) : (r >= 0 && r < g.Sh)) {
                float4* q = reinterpret_cast<float4*>(gh + (unsigned)r * 4u);
                q[0] = make_float4(gtx[m][0], gty[m][0], gtx[m][1], gty[m][1]);
                q[1] = make_float4(gtx[m][2], gty[m][2], gtx[m][3], gty[m][3]);
            }
        }
    }
    HGS_T(fft.tr_n, 7);
    if constexpr (STATS) StatAcc<R>::slot_store(stat_slot, a.spartial, b);
    if (do_upd) {
        const double s = block_sum((double)acc_w, scratch);
        if (j == 0) a.wpartial[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
#if HGS_TRACE
    __syncthreads();
    {   // dump this workgroup's events (128 per wave): fpartial doubles as the destination in the microbenchmark
        const int nev = ((int)blockDim.x >> 6) * 128;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.fpartial) + (size_t)blockIdx.x * nev;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + HGS_TRACE_OFF);
        for (int i = j; i < nev; i += blockDim.x) dst[i] = src[i];
    }
#endif
}

#ifndef HGS_PRESUM_ABL
#define HGS_PRESUM_ABL 0     // ablation builds (tools/build_variant.sh): 1 = the pre-pass loads no weights, 2 = neither weights nor targets
#endif
template <typename R, int T>
__device__ __forceinline__ void presum_abl_loads(const R* __restrict__ wc, const R* __restrict__ tc, bool, int j, R (&wr)[16], R (&tr)[16]) {
    static_for<0, 16>([&](auto m_) {
        constexpr int m = m_;
        wr[m] = (R)1e-4;
        tr[m] = HGS_PRESUM_ABL >= 2 ? ((m >= 6 && m < 10) ? (R)0.5 : (R)0) : tc[lane_pos<T>(j, m)];
    });
}
// =====================================================================================================
// Pre-pass of col_tile_kernel RULE 5 (MRAF with the WGS-Leonardo / WGS-Kim update, _hologram.py:1606-1653 after
// _update_weights :1786-1879): D = sum over the signal pixels of w'^2 - w^2, w = the normalised weight that enters the
// update, w' = w * fc the un-normalised new one -- exactly as the main pass will form it (same transform, same rule, same
// fix-ups), so that it can rebuild the field with 1 / ||w'|| = 1 / sqrt(1 + D).  Forward transforms only, over the tiles that
// hold a column with a finite non-zero target (bit 1 of the column scan; a.col_flags is required), the rule only where a
// wave meets such a pixel; nothing is written but one partial per workgroup (a.wpartial -- the caller points it at the
// buffer the main pass reads through ColArgs::dpartial).  Pixels outside the signal region contribute exactly 0 (fc = 1).
// =====================================================================================================
template <typename R, int N> constexpr size_t col_presum_lds_bytes() { return lds_elems<N>() * sizeof(Cx<R>) + SCRATCH_DOUBLES * sizeof(double); }
template <typename R, int N, int NR>
__global__ __launch_bounds__(N / 16, HGS_FUSED_OCC) void col_presum_kernel(ColArgs<R> a, int shift) {
    constexpr int T = N / 16;
    static_assert(T >= 256 && sizeof(R) == 4, "col_presum_kernel: fp32, one column per workgroup pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int j = threadIdx.x;
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem);
    double* scratch = reinterpret_cast<double*>(lds + lds_elems<N>());
    using Sel = FftSel<R, N, true, false>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    const CParams<R> cp = a.cp;
    const int js = Sel::space_lane(j);
    const R sgn = (j & 1) ? (R)-1 : (R)1;
    const R sgs = (js & 1) ? (R)-1 : (R)1;
    const size_t P = (size_t)g.Ph * g.Pw;
    const R wsc = a.wscale[b];
    Cx<R> om = a.tw[(j * shift) & (N - 1)];
    om = om * (sgn * a.scale);
    const int r_lane = js + shift - g.r0;
    const unsigned char* flags = a.col_flags + (size_t)b * g.Pw;
    double acc = 0;
    Cx<R> v[16];
    R gtx[NR][4], gty[NR][4];
    R wr[16], tr[16], wn[16], tn[16];
    // the signal columns of this workgroup's tiles (ct = blockIdx.x + k gridDim.x), in order: (tile, column) -> the next one.
    // Uniform scalar walk over the scan bits; a workgroup owns a handful of tiles.
    // (the four scan bytes of a tile as ONE aligned word, the 16-bit slot masks likewise: uniform dword loads go through the
    //  scalar cache; as byte / half-word loads they were vector memory instructions with `s_waitcnt vmcnt(0)` behind each --
    //  one to four dependent memory round trips per column before its weights could even be requested, and a drain of
    //  whatever else was in flight)
    const int ntile = g.Pw / 4;
    const unsigned* flags4 = reinterpret_cast<const unsigned*>(flags);      // (Pw is a multiple of 4)
    unsigned f4 = 0;
    int ct_f4 = -1;
    auto next_signal = [&](int& ct, int& c) -> bool {          // advances (ct, c); false at the end
        for (;;) {
            if (++c >= 4) { c = 0; ct += (int)gridDim.x; }
            if (ct >= ntile) return false;
            if (ct != ct_f4) { f4 = __builtin_amdgcn_readfirstlane(flags4[ct]); ct_f4 = ct; }
            if ((f4 & 0x02020202u) == 0) { c = 3; continue; }        // no signal column in this tile
            if (((f4 >> (8 * c)) & 2u) != 0) return true;
        }
    };
    int ct = blockIdx.x, c = -1;
    bool have = next_signal(ct, c);
    // weights and targets of column (ct, c) into wn / tn: only the 16-byte quarters of a lane's 64 bytes whose register slots
    // hold signal pixels somewhere in the column (sig_rows, uniform) -- at cfg 5 two of the four (the image occupies rows
    // 3072 .. 5119 = slots 6 .. 9); everything else reads as "no target"
    auto issue_next = [&]() {
        const size_t cb = (size_t)b * P + (size_t)(ct * 4 + c) * g.Ph;
#if HGS_PRESUM_ABL
        presum_abl_loads<R, T>(a.w + cb, a.t + cb, true, j, wn, tn);
#else
        unsigned sl = 0xffffu;
        if (a.sig_rows != nullptr) {
            const size_t si = (size_t)b * g.Pw + ct * 4 + c;
            sl = (__builtin_amdgcn_readfirstlane(reinterpret_cast<const unsigned*>(a.sig_rows)[si >> 1]) >> (16 * (int)(si & 1))) & 0xffffu;
        }
        const float4* wq = reinterpret_cast<const float4*>(a.w + cb + lane_pos<T>(j, 0));
        const float4* tq = reinterpret_cast<const float4*>(a.t + cb + lane_pos<T>(j, 0));
        static_for<0, 4>([&](auto q_) {
            constexpr int q = q_;
            float4 wv = make_float4(0, 0, 0, 0), tv = wv;
            if ((sl >> (4 * q)) & 0xfu) { wv = wq[q]; tv = tq[q]; }
            wn[4 * q] = wv.x; wn[4 * q + 1] = wv.y; wn[4 * q + 2] = wv.z; wn[4 * q + 3] = wv.w;
            tn[4 * q] = tv.x; tn[4 * q + 1] = tv.y; tn[4 * q + 2] = tv.z; tn[4 * q + 3] = tv.w;
        });
#endif
    };
    if (have) issue_next();
    int ct_loaded = -1;
#pragma unroll 1
    while (have) {
        if (ct != ct_loaded) {
            const Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4;
            // (straight-line buffer loads -- rows outside the SLM are outside the resource -- so that the NR row requests are in
            //  flight together: as conditional loads each slot's pair was waited for before the next was issued)
            const Buf bt(gh, (unsigned)g.Sh * 4u * (unsigned)sizeof(Cx<R>));
            float4 lo_[NR], hi_[NR];
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const unsigned vo = (unsigned)(r_lane + m * T) * 4u * (unsigned)sizeof(Cx<R>);
                lo_[m] = bt.template ld<float4>(vo, 0u);
                hi_[m] = bt.template ld<float4>(vo + 16u, 0u);
            }
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const float4 lo = lo_[m], hi = hi_[m];
                gtx[m][0] = lo.x; gty[m][0] = lo.y; gtx[m][1] = lo.z; gty[m][1] = lo.w;
                gtx[m][2] = hi.x; gty[m][2] = hi.y; gtx[m][3] = hi.z; gty[m][3] = hi.w;
            }
            ct_loaded = ct;
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m < NR) v[m] = mk<R>(gtx[m < NR ? m : 0][c] * sgs, gty[m < NR ? m : 0][c] * sgs);
            else v[m] = mk<R>(0, 0);
            wr[m] = wn[m];
            tr[m] = tn[m];
        }
        // the NEXT signal column's weights and targets land under this column's transform (without this every column
        // waited a full memory round trip for them: 44 us per launch at cfg 5 against 2048 columns' worth of transforms)
        have = next_signal(ct, c);
        if (have) issue_next();
        fft.template fwd_lead<NR>(v, lds, j);
        static_for<0, 16>([&](auto m_) {
            constexpr int m = m_;
            const R t = tr[m];
            // (wave-uniform: the rule only where some lane of the wave holds a signal pixel in this register)
            if (__builtin_amdgcn_ballot_w64(t != (R)0 && t == t) == 0) return;
            const Cx<R> F = cmul(v[m], om);
            const R p2 = F.x * F.x + F.y * F.y;
            const R w0 = wr[m] * wsc;
            R fc = leonardo_factor<R>(p2, t, cp.inv_fnorm, cp.p_exp);
            fc = (t != (R)0 && fc < (R)INFINITY) ? fc : (R)1;
            R w1 = w0 * fc;
            if (is_nan(w1)) w1 = (R)0.0001;
            // (w0 NaN -- weights nobody has updated yet -- cannot occur: the engine takes this path only behind an update)
            acc += (double)w1 * (double)w1 - (double)w0 * (double)w0;
            if constexpr (m % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        });
        // (consecutive forward transforms need no extra rendezvous: the first exchange of the next one is wave-local, in the
        //  wave's own regions, and this one ended with the barrier behind its cross-wave gather -- as between the columns of
        //  col_tile_kernel)
    }
    const double s = block_sum(acc, scratch);
    if (j == 0) a.wpartial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// (A form of this pre-pass compiled for FOUR waves per SIMD -- <= 128 VGPRs: two 512-lane workgroups per CU at 8192 rows; no
//  tile registers, stage twiddles per use, weights requested after the transform; 7 .. 18 spilled VGPRs -- was measured and
//  removed again: 54.6 against 51.2 us at cfg 5, NOTEBOOK.md round 6.)
// =====================================================================================================
// FUSED column kernel, HALF-width tile-resident form (round 5): a lane group of T = N / 16 lanes keeps TWO adjacent
// columns of a 4-column tile (16 bytes per tile row and lane) in registers and transforms them from / to those
// registers; the plain WGS-Leonardo / WGS-Kim update (RULE 1) or no update (RULE 2), as in col_tile_kernel.
//   * N = 4096 (one lane group per 256-lane workgroup): for BATCHES.  Half the tile registers bring the kernel under
//     168 VGPRs, i.e. three workgroups per CU instead of two -- +11 % throughput per CU (NOTEBOOK round 2); for one
//     hologram the 4096 columns over 768 slots round 2.67 up to 3 and the gain is lost, for the 32,768 columns of a
//     batch of eight (BASELINE config 3, the unit every GPU of the 8-GPU run executes) it is not.  The two halves of
//     a tile go to two workgroups of ONE XCD that run together (half_xmap), so the 32-byte tile rows meet in that L2.
//   * N = 2048 (two lane groups per workgroup, one half each: the workgroup moves whole 32-byte tile rows): the
//     padded size Hologram.get_padded_shape(padding_order=1) gives a 1080-row SLM (what fourier_grid_project builds);
//     until round 5 that size ran the per-column kernel (every GH byte re-fetched per column, 0.39 of the HBM peak).
// =====================================================================================================
template <int N> struct Tile2Cfg {
    static constexpr int T = N / 16;
    static constexpr int CPAR = T >= 256 ? 1 : 256 / T;       // lane groups per workgroup (2 at 2048)
    static constexpr int WG = T * CPAR;
};
// 4096 rows: the column of the half tile that is NOT being transformed waits in LDS (lane-private slots, six per lane: 12 KB),
// not in registers -- with both columns resident the NR = 5 / 6 update instances were 4 / 13 registers over the 168 that three
// workgroups per CU allow (tools/resusage.sh: 20 / 56 bytes of scratch in the headline kernel)
// (PARK is a template argument: a batch of eight, whose 1.4 GB do not fit the Infinity Cache, is 3 - 5 % FASTER with both
//  columns in registers and the 4 spilled ones -- 354 against 372 us per column launch -- so batches keep that form)
template <typename R, int N, bool PARK = false> constexpr size_t col_tile2_lds_bytes() {
    return (size_t)Tile2Cfg<N>::CPAR * lds_elems<N>() * sizeof(Cx<R>) + SCRATCH_DOUBLES * sizeof(double) +
           (PARK ? (size_t)6 * Tile2Cfg<N>::T * sizeof(Cx<R>) : 0);
}

#ifndef HGS_TILE2_BUF_ST
#define HGS_TILE2_BUF_ST 1   // ... and, in the instances that request the next half tile ahead (NXF), those requests and the stores behind them
#endif
#ifndef HGS_TILE2_BUF
#define HGS_TILE2_BUF 1      // the half tile's rows as straight-line buffer loads (round 6)
#endif
#ifndef HGS_TILE2_CONS_GROUP
#define HGS_TILE2_CONS_GROUP 4      // pixels of a lane whose rule evaluation the scheduler may interleave (168 registers: fewer than col_tile_kernel's 16)
#endif
// (4096 rows: three waves per SIMD = three workgroups per CU, the point of the kernel; 2048 rows: two -- the general
//  transform keeps 20 stage twiddles and up to ten tile slots, at three it spilled 14 .. 103 VGPRs)
template <typename R, int N, int PHASE, int NR, int RULE, bool PARK = false, bool NXF = false>
__global__ __launch_bounds__(Tile2Cfg<N>::WG, (N >= 4096 ? 3 : 2)) void col_tile2_kernel(ColArgs<R> a, int shift, int half_xmap) {
    static_assert(!NXF || (PARK && PHASE == 0 && NR <= 5), "col_tile2_kernel: NXF is for the plain parked instances with at most five slots");
    static_assert(!PARK || Tile2Cfg<N>::CPAR == 1, "col_tile2_kernel: the parked form is for one lane group per workgroup");
    constexpr int TILE2_CONS_GROUP = HGS_TILE2_CONS_GROUP;
    using M = Math<R>;
    static_assert(sizeof(R) == 4 && (N == 2048 || N == 4096) && (RULE == 1 || RULE == 2), "col_tile2_kernel: fp32, 2048 / 4096 rows, plain rules");
    constexpr int T = Tile2Cfg<N>::T, CPAR = Tile2Cfg<N>::CPAR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Geo g = a.g;
    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid / T), j = tid % T;       // (T a multiple of 64: wave-uniform)
    const int b = blockIdx.y;
    Cx<R>* lds = reinterpret_cast<Cx<R>*>(smem) + grp * lds_elems<N>();
    double* scratch = reinterpret_cast<double*>(reinterpret_cast<Cx<R>*>(smem) + CPAR * lds_elems<N>());

#if HGS_TRACE
    {   // (a workgroup of this kernel records fewer than 128 events per wave: clear the rest)
        unsigned long long* tz = reinterpret_cast<unsigned long long*>(smem + HGS_TRACE_OFF);
        for (int i = tid; i < ((int)blockDim.x >> 6) * 128; i += blockDim.x) tz[i] = 0ull;
        __syncthreads();
    }
#endif
    using Sel = FftSel<R, N, true>;
    typename Sel::type fft;
    fft.init(a.tw, j);
    const CParams<R> cp = a.cp;
    constexpr bool do_upd = RULE == 1;
    const int js = Sel::space_lane(j);
    const R sgn = (j & 1) ? (R)-1 : (R)1;
    const R sgs = (js & 1) ? (R)-1 : (R)1;
    const size_t P = (size_t)g.Ph * g.Pw;
    const R wsc = a.wscale[b];
    Cx<R> om = a.tw[(j * shift) & (N - 1)];            // shift-theorem factor (see col_tile_kernel), sign and ortho scale folded in
    om = om * (sgn * a.scale);
    const int r_lane = js + shift - g.r0;              // SLM row of slot m is r_lane + m*T
    const int ntiles = g.Pw / 4;
    R acc_w = 0;

    Cx<R> v[16];
    // the half tile, one array per column and component: selected by the (uniform) column of the pass with v_cndmask -- a
    // run-time index into [NR][2] arrays put them on the stack (96 bytes of scratch), unrolling the two passes made the
    // scheduler interleave them (23 .. 187 spilled registers)
    R g0x[PARK ? 1 : NR], g0y[PARK ? 1 : NR], g1x[PARK ? 1 : NR], g1y[PARK ? 1 : NR];
    Cx<R>* park = reinterpret_cast<Cx<R>*>(scratch + SCRATCH_DOUBLES) + j;     // PARK: slot m of this lane at park[m * T]
    R wr[16], tr[16];
    const R* wbase = a.w + (size_t)b * P;
    const R* tbase = a.t + (size_t)b * P;

    // schedule: CPAR = 2 -- a workgroup owns whole tiles, group g their half g.  CPAR = 1 -- a workgroup owns half tiles;
    // with half_xmap (gridDim.x a multiple of 16) workgroups (xcd, 2 i) and (xcd, 2 i + 1) take the two halves of one tile
    // (workgroups run round-robin over the 8 XCDs -- a speed-only assumption, as in row_kernel)
    const int G = (int)gridDim.x;
    int ct0, ct_step, half;
    if constexpr (CPAR == 2) {
        ct0 = (int)blockIdx.x; ct_step = G; half = grp;
    } else if (half_xmap) {
        const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
        half = idx & 1; ct0 = (idx >> 1) * 8 + xcd; ct_step = G / 2;
    } else {                                   // (the host launches an even number of workgroups)
        half = (int)blockIdx.x & 1; ct0 = (int)blockIdx.x >> 1; ct_step = G / 2;
    }
    bool first = true;
    // PARK: the rows of the workgroup's NEXT half tile are requested BEFORE the stores of the current one (vmcnt retires in order:
    // behind the stores, the wait for the new rows also waited for the stores to be acknowledged -- tools/microbench/trace_tile2:
    // 6.3 k cycles per half tile); they travel in the registers the finished tile has just freed
    // (NXF, a template argument: plain instances with at most five slots -- the phase-reading ones and six slots have no 20
    //  registers to lend -- and targets with few active columns: spot array 45.6 -> 44.8 us, but a dense image, whose constraint
    //  runs in every wave and wants the registers, 65.1 -> 67.1 us)
    float4 nq[NXF ? NR : 1];
    bool have_nq = false;
    // (Experiment, removed: the rows of the workgroup's NEXT half tile requested while the second column of the current one is
    //  transformed back -- tools/microbench/trace_tile2 shows 6.3 k of a workgroup's ~32 k cycles per half tile waiting for its
    //  rows, which the other two workgroups of the CU cover -- costs 20 registers that stay live across the rolled column loop:
    //  17 / 31 spilled at five / six slots.)
#pragma unroll 1
    for (int ct = ct0; ct < ntiles; ct += ct_step) {
        Cx<R>* gh = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)ct * g.Sh * 4 + 2 * half;
        HGS_T(fft.tr_n, 1);
        // The half tile's rows through a buffer resource (rows outside the SLM are outside the resource: they read as zero), all
        // NR requests ahead of the first use.  As conditional loads each one sat in its own exec-masked block with its use --
        // a parked column's LDS store -- right behind it: five dependent memory round trips per half tile instead of one
        // (tools/microbench/trace_tile2: 6.3 k cycles per half tile waiting for rows; the ISA had `s_waitcnt vmcnt(0)` after
        // every one of them) wherever the rows were not requested ahead (NXF): every dense image, batch and WGS-Kim launch.
        // (Measured per instance, A/B inside one GPU call: the headline's few-active-columns instance 44.8 -> 44.0 us, a batch of
        //  eight 180.8 -> 175.9 us per column launch; a dense image target 64.3 -> 65.8 us and the 2048-row form 21.2 -> 22.1 us
        //  LOSE -- their constraint / second lane group fills the waits and the resource set-up sits on the chain -- and keep
        //  the conditional loads.)
        constexpr bool TBUF = HGS_TILE2_BUF && N == 4096 && PHASE == 0 && (NXF || !PARK);
        float4 tq[TBUF ? NR : 1];
        if constexpr (TBUF) {
            // (rows that already arrived in nq: an empty resource -- nothing is fetched)
            const Buf bt(gh, (NXF && have_nq) ? 0u : (unsigned)((size_t)g.Sh * 4 - 2 * half) * (unsigned)sizeof(Cx<R>));
#pragma unroll
            for (int m = 0; m < NR; ++m)
                tq[TBUF ? m : 0] = bt.template ld<float4>((unsigned)(r_lane + m * T) * 4u * (unsigned)sizeof(Cx<R>), 0u);   // (negative rows wrap out of range)
        }
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const int r = r_lane + m * T;
            float4 q = make_float4(0, 0, 0, 0);
            if (NXF && have_nq) q = nq[NXF ? m : 0];
            else if constexpr (TBUF) q = tq[TBUF ? m : 0];
            else
            if (r >= 0 && r < g.Sh) q = *reinterpret_cast<const float4*>(gh + (unsigned)r * 4u);
            if constexpr (PARK) {           // column 0 straight into the transform registers, column 1 waits in LDS
                v[m] = mk<R>(q.x * sgs, q.y * sgs);
                park[m * T] = mk<R>(q.z, q.w);
            } else {
                g0x[m] = q.x; g0y[m] = q.y; g1x[m] = q.z; g1y[m] = q.w;
            }
        }
#if HGS_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        HGS_T(fft.tr_n, 2);
        const int col0 = ct * 4 + 2 * half;
        if (first) {
            issue_wt_loads<R, T>(wbase + (size_t)col0 * g.Ph, tbase + (size_t)col0 * g.Ph, do_upd, j, wr, tr);
            first = false;
        }
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            const size_t cb = (size_t)b * P + (size_t)(col0 + c) * g.Ph;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m < NR) {
                    if constexpr (!PARK) {
                        const R xr = c ? g1x[m < NR ? m : 0] : g0x[m < NR ? m : 0], xi = c ? g1y[m < NR ? m : 0] : g0y[m < NR ? m : 0];
                        v[m] = mk<R>(xr * sgs, xi * sgs);
                    }                                    // (PARK: slots 0 .. NR-1 hold this column already)
                } else {
                    v[m] = mk<R>(0, 0);
                }
            }
            fft.template fwd_lead<(NR < 4 ? 4 : NR)>(v, lds, j);     // slots NR.. are zero (rows outside the SLM)
            HGS_T(fft.tr_n, 3);
#if HGS_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            HGS_T(fft.tr_n, 4);

            R* wc = a.w + cb;
            R* pfc = (PHASE != 0) ? a.pff + cb : nullptr;
            bool w_changed = false;
            R pf[PHASE == 1 ? 16 : 1];
            // PHASE 2: the lane's sixteen stored phases (64 contiguous bytes) in two 16-byte registers that are refilled as they
            // are used up -- pixels 0-3 / 8-11 from qa, 4-7 / 12-15 from qb -- instead of sixteen live values (the update
            // instances were 8 .. 12 registers over the 168 of three workgroups per CU)
            constexpr bool PFQ = PHASE == 2 && HGS_LANE_MAJOR;
            const float4* pq = PFQ ? reinterpret_cast<const float4*>(pfc + lane_pos<T>(j, 0)) : nullptr;
            float4 qa = make_float4(0, 0, 0, 0), qb = qa;
            R pf2[(PHASE == 2 && !PFQ) ? 16 : 1];
            if constexpr (PFQ) { qa = pq[0]; qb = pq[1]; }
            else if constexpr (PHASE == 2) static_for<0, 16>([&](auto m_) { constexpr int m = m_; pf2[m] = pfc[lane_pos<T>(j, m)]; });
            static_for<0, 16>([&](auto m_) {
                constexpr int m = m_;
                if constexpr (PFQ && m == 4) qa = pq[2];            // (pixels 0-3 are through)
                if constexpr (PFQ && m == 8) qb = pq[3];
                R pfm = 0;
                if constexpr (PFQ) {
                    const float4 qv = ((m / 4) & 1) ? qb : qa;
                    // (pixels 4-7 read qb = pq[1], 8-11 qa = pq[2], 12-15 qb = pq[3])
                    pfm = (m % 4 == 0) ? qv.x : (m % 4 == 1) ? qv.y : (m % 4 == 2) ? qv.z : qv.w;
                } else if constexpr (PHASE == 2) pfm = pf2[m];
                // wave-uniform skip where weight and target are zero (see col_tile_kernel)
                if (PHASE != 1 && HGS_SPARSE_SKIP && __builtin_amdgcn_ballot_w64(wr[m] != (R)0 || tr[m] != (R)0) == 0) {
                    v[m] = mk<R>(0, 0);
                    return;
                }
                const Cx<R> F = cmul(v[m], om);
                const R p2 = F.x * F.x + F.y * F.y;
                const R wraw = wr[m];
                R wv = wraw * wsc;
                if constexpr (do_upd) {
                    const R t = tr[m];
                    R fc = leonardo_factor<R>(p2, t, cp.inv_fnorm, cp.p_exp);
                    fc = (t != (R)0 && fc < (R)INFINITY) ? fc : (R)1;
                    wv *= fc;
                    if (is_nan(wv)) wv = (R)0.0001;
                    w_changed |= (wv != wraw);
                    wr[m] = wv;
                    acc_w += wv * wv;
                }
                Cx<R> ph;
                if constexpr (PHASE == 2) {
                    R sn, cs;
                    M::sincos_phase(pfm, &sn, &cs);
                    ph = mk<R>(cs, sn);
                } else {
                    const R inv = rsqrt_full(p2);
                    ph = mk<R>((p2 > (R)0) ? F.x * inv : (R)1, (p2 > (R)0) ? F.y * inv : (R)0);
                    if constexpr (PHASE == 1) pf[m] = M::atan2(F.y, F.x);
                }
                v[m] = cmulc(ph, om) * wv;
                if constexpr (PHASE == 1 && m % 4 == 3) {
                    static_for<m - 3, m + 1>([&](auto i_) { constexpr int i = i_; pfc[lane_pos<T>(j, i)] = pf[i]; });
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (m % TILE2_CONS_GROUP == TILE2_CONS_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
            });
            if (do_upd && w_changed) {
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; wc[lane_pos<T>(j, m)] = wr[m]; });
            }
            {   // weights / target of the next column (or of the first column of this group's next half tile)
                const int ncol = c == 0 ? col0 + 1 : (ct + ct_step) * 4 + 2 * half;
                if (c == 0 || ct + ct_step < ntiles)
                    issue_wt_loads<R, T>(wbase + (size_t)ncol * g.Ph, tbase + (size_t)ncol * g.Ph, do_upd, j, wr, tr);
            }
            HGS_T(fft.tr_n, 5);
            fft.template inv_after_fwd_trail<NR>(v, lds, j);
            HGS_T(fft.tr_n, 6);
            if constexpr (PARK) {
                if (c == 0) {
                    // column 0 done: its result takes column 1's place in the lane's slots, column 1 moves into the registers
#pragma unroll
                    for (int m = 0; m < NR; ++m) {
                        const Cx<R> h = v[m] * (sgs * a.scale);
                        const Cx<R> nx = park[m * T];
                        park[m * T] = h;
                        v[m] = nx * sgs;
                    }
                }           // (c == 1: the second column's result stays in v for the code after the loop)
            } else {
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const Cx<R> h = v[m] * (sgs * a.scale);
                    g0x[m] = c ? g0x[m] : h.x; g0y[m] = c ? g0y[m] : h.y;
                    g1x[m] = c ? h.x : g1x[m]; g1y[m] = c ? h.y : g1y[m];
                }
            }
        }
        if constexpr (!PARK) {
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const int r = r_lane + m * T;
                if (r >= 0 && r < g.Sh)
                    *reinterpret_cast<float4*>(gh + (unsigned)r * 4u) = make_float4(g0x[m], g0y[m], g1x[m], g1y[m]);
            }
        } else {
            float4 outq[NR];
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const Cx<R> h1 = v[m] * (sgs * a.scale);
                const Cx<R> h0 = park[m * T];
                outq[m] = make_float4(h0.x, h0.y, h1.x, h1.y);
            }
            have_nq = NXF && ct + ct_step < ntiles;
            if constexpr (TBUF && NXF && HGS_TILE2_BUF_ST) {
                // straight-line: NR loads through a resource that is empty when there is no next half tile, then NR stores through
                // one that drops rows outside the SLM -- the compiler can count what is in flight, so the wait for the new rows at
                // the top of the loop is vmcnt(NR), not vmcnt(0): it no longer includes the acknowledgement of these stores
                const Cx<R>* ghn = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)(have_nq ? ct + ct_step : ct) * g.Sh * 4 + 2 * half;
                const unsigned tbytes = (unsigned)((size_t)g.Sh * 4 - 2 * half) * (unsigned)sizeof(Cx<R>);
                const Buf bn(ghn, have_nq ? tbytes : 0u), bs(gh, tbytes);
#pragma unroll
                for (int m = 0; m < NR; ++m)
                    nq[NXF ? m : 0] = bn.template ld<float4>((unsigned)(r_lane + m * T) * 4u * (unsigned)sizeof(Cx<R>), 0u);
#pragma unroll
                for (int m = 0; m < NR; ++m)
                    bs.template st<float4>(outq[m], (unsigned)(r_lane + m * T) * 4u * (unsigned)sizeof(Cx<R>), 0u);
            } else {
            if (NXF && have_nq) {
                const Cx<R>* ghn = a.gh + (size_t)b * g.Sh * g.Pw + (size_t)(ct + ct_step) * g.Sh * 4 + 2 * half;
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const int r = r_lane + m * T;
                    nq[NXF ? m : 0] = make_float4(0, 0, 0, 0);
                    if (r >= 0 && r < g.Sh) nq[NXF ? m : 0] = *reinterpret_cast<const float4*>(ghn + (unsigned)r * 4u);
                }
            }
#pragma unroll
            for (int m = 0; m < NR; ++m) {
                const int r = r_lane + m * T;
                if (r >= 0 && r < g.Sh) *reinterpret_cast<float4*>(gh + (unsigned)r * 4u) = outq[m];
            }
            }
        }
    }
    HGS_T(fft.tr_n, 7);
    if constexpr (do_upd) {
        const double s = block_sum((double)acc_w, scratch);
        if (tid == 0) a.wpartial[(size_t)b * gridDim.x + blockIdx.x] = s;
    }
#if HGS_TRACE
    __syncthreads();
    {   // dump this workgroup's events (128 per wave): fpartial doubles as the destination in the microbenchmark
        const int nev = ((int)blockDim.x >> 6) * 128;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.fpartial) + (size_t)blockIdx.x * nev;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + HGS_TRACE_OFF);
        for (int i = tid; i < nev; i += blockDim.x) dst[i] = src[i];
    }
#endif
}

// =====================================================================================================
// Elementwise kernels of the general (stepwise) path over column-major P-sized arrays.
// =====================================================================================================
template <typename R> struct EwArgs {
    size_t P;          // elements per hologram
    int batch;
    Cx<R>* ff;
    R* amp_ff;
    R* pff;
    R* w;
    const R* t;
    const double* fsum;   // [batch] sum |F|^2 (nansum)       -> 1/||F||
    const double* nogsum; // [batch] sum fc (Nogrette)
    double* partial;      // [batch][gridDim.x] output partial sums
    const double* wsum;   // [batch] sum w'^2                  -> 1/||w'||
    Cx<R>* zero_weights;  // [batch][P] sparse-as-dense accumulator for zero_factor, or nullptr
    CParams<R> cp;
};

// Nogrette needs nanmean(fc) over the whole array before the update (:1851, quirk A9).
template <typename R> __global__ void ew_nogrette_sum(EwArgs<R> a) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const R inv_fn = (R)(1.0 / ::sqrt(a.fsum[b]));
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.P; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = (size_t)b * a.P + i;
        const R t = a.t[idx];
        R fc = a.amp_ff[idx] * inv_fn / t;
        if (is_pinf(fc)) fc = 1;
        if (t == (R)0) fc = 1;
        if (is_nan(fc)) fc = 1;
        acc += (double)fc;
    }
    const double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// w' = nanfix(w * fc)  (un-normalised) + partial sum of w'^2.
template <typename R> __global__ void ew_weight_update(EwArgs<R> a) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const R inv_fn = (R)(1.0 / ::sqrt(a.fsum[b]));
    R nog = 0;
    if (a.cp.method == M_NOGRETTE) nog = (R)(-(1.0 / (a.nogsum[b] / (double)a.P)));
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.P; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = (size_t)b * a.P + i;
        const R fc = weight_factor<R>(a.cp.method, a.amp_ff[idx] * inv_fn, a.t[idx], a.cp.p_exp, a.cp.p_fac, nog);
        R wv = a.w[idx] * fc;
        if (is_nan(wv)) wv = (R)0.0001;
        a.w[idx] = wv;
        acc += (double)wv * (double)wv;
    }
    const double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) a.partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// Normalise weights (if an update happened) and rebuild the farfield (:1590-1653).
template <typename R> __global__ void ew_rebuild(EwArgs<R> a) {
    using M = Math<R>;
    const int b = blockIdx.y;
    const CParams<R> cp = a.cp;
    const R wsc = (a.wsum != nullptr) ? (R)(1.0 / ::sqrt(a.wsum[b])) : (R)1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.P; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = (size_t)b * a.P + i;
        Cx<R> F = a.ff[idx];
        R wv = a.w[idx];
        if (a.wsum != nullptr) {
            wv *= wsc;
            a.w[idx] = wv;
        }
        bool signal = true, noise = false;
        if (cp.mraf) {
            const R t = a.t[idx];
            noise = is_nan(t);
            const bool zero = (!noise) && (M::abs(t) == (R)0);
            signal = !(noise || zero);
            if (zero) {
                if (cp.zero_mode) {  // :1613-1616
                    Cx<R> zw = a.zero_weights[idx];
                    const R mag = M::sqrt(F.x * F.x + F.y * F.y) * cp.zero_factor;
                    zw.x -= mag * F.x;
                    zw.y -= mag * F.y;
                    a.zero_weights[idx] = zw;
                    F = zw;
                } else {
                    F = mk<R>(0, 0);
                }
            }
        }
        R p;
        if (cp.use_fixed) {
            p = a.pff[idx];
        } else {
            p = M::atan2(F.y, F.x);
            a.pff[idx] = p;
        }
        if (signal) {
            R s, c;
            M::sincos(p, &s, &c);
            F = mk<R>(wv * c, wv * s);
        } else if (noise && cp.has_mraf_factor) {
            F = F * cp.mraf_factor;
        }
        a.ff[idx] = F;
    }
}

// partial sums of x^2 (nansum) for an arbitrary real array
template <typename R> __global__ void ew_sumsq(const R* x, size_t P, double* partial) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        const double v = (double)x[(size_t)b * P + i];
        if (v == v) acc += v * v;
    }
    const double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}

// phase_ff = atan2(F) only (Kim transition with MRAF-free stepwise mode, :1583)
template <typename R> __global__ void ew_store_phase(EwArgs<R> a) {
    using M = Math<R>;
    const int b = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.P; i += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = (size_t)b * a.P + i;
        const Cx<R> F = a.ff[idx];
        a.pff[idx] = M::atan2(F.y, F.x);
    }
}

// ---- tiny helpers -------------------------------------------------------------------------------------
// out[b] = sum_i partial[b][i]
static __global__ void reduce_partials(const double* partial, int n, double* out) {
    __shared__ double scratch[16];
    const int b = blockIdx.x;
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[(size_t)b * n + i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[b] = s;
}

// wscale[b] = 1 / sqrt(sum_i partial[b][i]) (and the sum itself to out[b]): the weight norm between the two column passes
// of an MRAF update, or between the single pass and the row kernel that joins its parts -- one launch instead of two
template <typename R> __global__ void reduce_to_scale(const double* partial, int n, double* out, R* wscale) {
    __shared__ double scratch[16];
    const int b = blockIdx.x;
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partial[(size_t)b * n + i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) {
        out[b] = s;
        wscale[b] = (R)(1.0 / ::sqrt(s));
    }
}

template <typename R> __global__ void set_scalar(R* p, int n, R v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// wscale[b] = 1/sqrt(sum[b])
template <typename R> __global__ void scale_from_sum(const double* sum, R* wscale, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wscale[i] = (R)(1.0 / ::sqrt(sum[i]));
}

// Tiled transpose between the host-facing natural [rows][cols] layout and the engine's
// column-major layout, with an optional per-hologram scale (pending weight normalisation).
// perm_T > 0: the column-major side is lane-major (col_pos) with T = perm_T; to_colmajor tells which
// side that is (1: out is the engine layout, 0: in is the engine layout).
template <typename E, typename R>
__global__ void transpose_scale(const E* __restrict__ in, E* __restrict__ out, int rows, int cols,
                                const R* scale, int perm_T, int to_colmajor) {
    __shared__ E tile[32][33];
    const int b = blockIdx.z;
    const size_t off = (size_t)b * rows * cols;
    const R s = scale ? scale[b] : (R)1;
    int x = blockIdx.x * 32 + threadIdx.x, y0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y)
        if (x < cols && y0 + i < rows) {
            // reading the engine layout [cols_of_engine = rows here][Ph = cols here]: permute within the row
            const int xs = (perm_T > 0 && !to_colmajor) ? col_pos(x, perm_T) : x;
            tile[i][threadIdx.x] = in[off + (size_t)(y0 + i) * cols + xs];
        }
    __syncthreads();
    x = blockIdx.y * 32 + threadIdx.x;
    y0 = blockIdx.x * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y)
        if (x < rows && y0 + i < cols) {
            E v = tile[threadIdx.x][i];
            if (scale) v = v * s;
            const int xd = (perm_T > 0 && to_colmajor) ? col_pos(x, perm_T) : x;
            out[off + (size_t)(y0 + i) * rows + xd] = v;
        }
}

// ---- spot-window feedback (SpotHologram._update_weights "computational_spot", _spots.py:1590-1624) ----
template <typename R> struct SpotArgs {
    Geo g;
    int n_spots, width, feedback;   // feedback: 1 = window sums of amp_ff^2, 2 = external amplitudes
    const int* spot_xy;             // [2][N] rounded (kx row 0, ky row 1), shared by the batch
    const R* amp_ff;                // column-major
    const double* ext_amp;          // [N] external_spot_amp (feedback 2)
    const double* spot_amp;         // [N] target amplitudes (un-normalised list, quirk A17)
    R* w;                           // column-major weights (normalised storage)
    R* fb;                          // [batch][N] scratch: feedback amplitudes
    CParams<R> cp;
    int inline_window;              // spot_update computes the window sums itself (one launch fewer)
};

// fb[n] = sqrt( sum_{w x w window at floor(v)} amp_ff^2 ) in float64 (analysis.take, quirk A18)
template <typename R> __device__ __forceinline__ R spot_window_value(const SpotArgs<R>& a, int b, int n);
template <typename R> __global__ void spot_window(SpotArgs<R> a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= a.n_spots) return;
    a.fb[(size_t)b * a.n_spots + n] = spot_window_value(a, b, n);
}
template <typename R> __device__ __forceinline__ R spot_window_value(const SpotArgs<R>& a, int b, int n) {
    const size_t P = (size_t)a.g.Ph * a.g.Pw;
    const int kx = a.spot_xy[n], ky = a.spot_xy[a.n_spots + n];
    const int lo = -((a.width - 1) / 2) - (((a.width - 1) & 1) ? 1 : 0);  // floor(-(w-1)/2)
    double s = 0;
    for (int dy = 0; dy < a.width; ++dy)
        for (int dx = 0; dx < a.width; ++dx) {
            const int x = kx + lo + dx, y = ky + lo + dy;
            const R v = a.amp_ff[(size_t)b * P + (size_t)x * a.g.Ph + col_pos(y, a.g.lane_T)];
            const R v2 = v * v;  // cp.square in working precision, then astype(float) (:1592, take :202)
            s += (double)v2;
        }
    return (R)::sqrt(s);
}

// One block per hologram: N-vector weight update with target = spot_amp, normalised as an N-vector
// and written back into the P-array (quirk A17).
template <typename R> __global__ void spot_update(SpotArgs<R> a) {
    __shared__ double scratch[16];
    __shared__ double bc;
    const int b = blockIdx.x;
    const size_t P = (size_t)a.g.Ph * a.g.Pw;
    const int N = a.n_spots;
    if (a.feedback == 1 && a.inline_window) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) a.fb[(size_t)b * N + n] = spot_window_value(a, b, n);
        __syncthreads();    // every later read of fb[n] is by this workgroup
    }
    // ||feedback||
    double acc = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const R f = (a.feedback == 2) ? (R)a.ext_amp[n] : a.fb[(size_t)b * N + n];
        if (!is_nan(f)) acc += (double)(f * f);
    }
    double s = block_sum(acc, scratch);
    if (threadIdx.x == 0) bc = s;
    __syncthreads();
    const R inv_fn = (R)1 / (R)::sqrt(bc);
    __syncthreads();
    R nog = 0;
    if (a.cp.method == M_NOGRETTE) {
        acc = 0;
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
            const R f = (a.feedback == 2) ? (R)a.ext_amp[n] : a.fb[(size_t)b * N + n];
            const R t = (R)a.spot_amp[n];
            R fc = f * inv_fn / t;
            if (is_pinf(fc)) fc = 1;
            if (t == (R)0) fc = 1;
            if (is_nan(fc)) fc = 1;
            acc += (double)fc;
        }
        s = block_sum(acc, scratch);
        if (threadIdx.x == 0) bc = s;
        __syncthreads();
        nog = (R)(-(1.0 / (bc / (double)N)));
        __syncthreads();
    }
    acc = 0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int kx = a.spot_xy[n], ky = a.spot_xy[N + n];
        const size_t idx = (size_t)b * P + (size_t)kx * a.g.Ph + col_pos(ky, a.g.lane_T);
        const R f = (a.feedback == 2) ? (R)a.ext_amp[n] : a.fb[(size_t)b * N + n];
        const R fc = weight_factor<R>(a.cp.method, f * inv_fn, (R)a.spot_amp[n], a.cp.p_exp, a.cp.p_fac, nog);
        R wv = a.w[idx] * fc;
        if (is_nan(wv)) wv = (R)0.0001;
        a.w[idx] = wv;
        acc += (double)wv * (double)wv;
    }
    s = block_sum(acc, scratch);
    if (threadIdx.x == 0) bc = s;
    __syncthreads();
    const R wsc = (R)1 / (R)::sqrt(bc);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int kx = a.spot_xy[n], ky = a.spot_xy[N + n];
        const size_t idx = (size_t)b * P + (size_t)kx * a.g.Ph + col_pos(ky, a.g.lane_T);
        a.w[idx] *= wsc;
    }
}


template <typename R> __global__ void scale_weights_kernel(R* w, const R* wscale, size_t P) {
    const int b = blockIdx.y;
    const R s = wscale[b];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x)
        w[(size_t)b * P + i] *= s;
}

// Hologram.reset_weights (_hologram.py:603-614): weights = target, NaN -> 0; zero_weights cleared
// hgs_set_array_sparse: dst[b][pos[k]] = val[k] (positions already in the engine layout, unique)
template <typename R> __global__ void scatter_values(R* dst, const uint32_t* pos, const R* val, int n, size_t P) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[(size_t)blockIdx.y * P + pos[k]] = val[k];
}

template <typename R> __global__ void reset_weights_kernel(R* w, const R* t, Cx<R>* zw, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const R v = t[i];
        w[i] = (v != v) ? (R)0 : v;
        if (zw) zw[i] = mk<R>(0, 0);
    }
}

// ---- statistics (_stats.py:7-116), device part ---------------------------------------------------------
// pass 1 over (feedback, target): sum f^2, nansum t^2, nansum t*f
template <typename R> __global__ void stats_pass1(const R* f, const R* t, size_t n, double* out) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    double a0 = 0, a1 = 0, a2 = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double fv = (double)f[(size_t)b * n + i], tv = (double)t[(size_t)b * n + i];
        a0 += fv * fv;
        if (tv == tv) {
            a1 += tv * tv;
            if (fv == fv) a2 += tv * fv;
        }
    }
    const double s0 = block_sum(a0, scratch), s1 = block_sum(a1, scratch), s2 = block_sum(a2, scratch);
    if (threadIdx.x == 0) {
        double* o = out + ((size_t)b * gridDim.x + blockIdx.x) * 3;
        o[0] = s0;
        o[1] = s1;
        o[2] = s2;
    }
}
// pass 2 over the mask (t != 0, not NaN): ratio = (f^2/Sf)/(t^2/St): min, max; err = t^2/St - f^2/Sf:
// min, max, sum, sum of squares, count
template <typename R>
__global__ void stats_pass2(const R* f, const R* t, size_t n, const double* sf_st, double* out) {
    __shared__ double red[7][16];
    const int b = blockIdx.y;
    const double isf = 1.0 / sf_st[2 * b], ist = 1.0 / sf_st[2 * b + 1];
    double rmin = INFINITY, rmax = -INFINITY, emin = INFINITY, emax = -INFINITY, es = 0, es2 = 0, cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double fv = (double)f[(size_t)b * n + i], tv = (double)t[(size_t)b * n + i];
        if (tv == tv && tv != 0.0) {
            const double fp = fv * fv * isf, tp = tv * tv * ist;
            if (tp != 0.0) {
                const double ratio = fp / tp, err = tp - fp;
                rmin = fmin(rmin, ratio);
                rmax = fmax(rmax, ratio);
                emin = fmin(emin, err);
                emax = fmax(emax, err);
                es += err;
                es2 += err * err;
                cnt += 1;
            }
        }
    }
    double vals[7] = {rmin, rmax, emin, emax, es, es2, cnt};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = 0; k < 7; ++k) {
        double v = vals[k];
        for (int o = 32; o > 0; o >>= 1) {
            const double u = __shfl_down(v, o, 64);
            v = (k == 0 || k == 2) ? fmin(v, u) : (k == 1 || k == 3) ? fmax(v, u) : v + u;
        }
        if (lane == 0) red[k][wid] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = out + ((size_t)b * gridDim.x + blockIdx.x) * 7;
        for (int k = 0; k < 7; ++k) {
            double v = red[k][0];
            for (int i = 1; i < nw; ++i)
                v = (k == 0 || k == 2) ? fmin(v, red[k][i]) : (k == 1 || k == 3) ? fmax(v, red[k][i]) : v + red[k][i];
            o[k] = v;
        }
    }
}

// ---- statistics of the fused path (hgs_iterate_stats) --------------------------------------------------
// neutral element of every slot of the per-wave partial buffer
static __global__ void stat_fill_neutral(double* spartial, size_t nslots) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    double* o = spartial + i * STAT_N;
    o[0] = o[1] = o[2] = o[3] = 0;
    o[4] = o[6] = INFINITY;
    o[5] = o[7] = -INFINITY;
}

// combine 8-slot records (sum x4, min, max, min, max) held one per thread; result valid in thread 0
__device__ __forceinline__ void stat_block_combine(double (&v)[STAT_N], double (*red)[16]) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < STAT_N; ++k) {
        double x = v[k];
        if (k < 4) x = wave_sum(x);
        else if (k == 4 || k == 6) x = wave_min(x);
        else x = wave_max(x);
        if (lane == 0) red[k][wid] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < STAT_N; ++k) {
            double x = red[k][0];
            for (int i = 1; i < nw; ++i)
                x = (k < 4) ? x + red[k][i] : (k == 4 || k == 6) ? fmin(x, red[k][i]) : fmax(x, red[k][i]);
            v[k] = x;
        }
    }
    __syncthreads();
}

// out[b][4] = efficiency, uniformity, pkpk_err, std_err from the per-wave partials of one column launch
static __global__ void stat_finalize(const double* spartial, int nparts, const double* tsum, double inv_fsum,
                                     double* out) {
    __shared__ double red[STAT_N][16];
    const int b = blockIdx.x;
    double v[STAT_N] = {0, 0, 0, 0, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        const double* o = spartial + ((size_t)b * nparts + i) * STAT_N;
        v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        v[4] = fmin(v[4], o[4]); v[5] = fmax(v[5], o[5]); v[6] = fmin(v[6], o[6]); v[7] = fmax(v[7], o[7]);
    }
    stat_block_combine(v, red);
    if (threadIdx.x == 0) {
        const double cnt = v[3], mean = v[1] / cnt, var = fmax(0.0, v[2] / cnt - mean * mean);
        out[4 * b + 0] = v[0] * v[0] * inv_fsum / tsum[b];           // (sum t_amp f_amp)^2
        out[4 * b + 1] = 1 - (v[5] - v[4]) / (v[5] + v[4]);
        out[4 * b + 2] = cnt * (v[7] - v[6]);
        out[4 * b + 3] = cnt * ::sqrt(var);
    }
}

// "computational_spot" group (_spots.py:1626-1679) from the window feedback of spot_window:
// feedback = fb[n], target = spot_amp[n], efficiency = sum fb^2 / total.  One block per hologram.
template <typename R>
__global__ void spot_stat_finalize(const R* fb, const double* spot_amp, int n_spots, double total, double* out) {
    __shared__ double red[STAT_N][16];
    __shared__ double bc[2];
    const int b = blockIdx.x;
    double v[STAT_N] = {0, 0, 0, 0, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int n = threadIdx.x; n < n_spots; n += blockDim.x) {
        const double f = (double)fb[(size_t)b * n_spots + n], t = spot_amp[n];
        v[0] += f * f;
        if (t == t) v[1] += t * t;
    }
    stat_block_combine(v, red);
    if (threadIdx.x == 0) { bc[0] = v[0]; bc[1] = v[1]; }
    __syncthreads();
    const double sf = bc[0], st = bc[1];
    double u[STAT_N] = {0, 0, 0, 0, INFINITY, -INFINITY, INFINITY, -INFINITY};
    for (int n = threadIdx.x; n < n_spots; n += blockDim.x) {
        const double f = (double)fb[(size_t)b * n_spots + n], t = spot_amp[n];
        const double tp = t * t / st, fp = f * f / sf;
        if (tp != 0 && tp == tp) {
            const double ratio = fp / tp, err = tp - fp;
            u[1] += err; u[2] += err * err; u[3] += 1;
            u[4] = fmin(u[4], ratio); u[5] = fmax(u[5], ratio); u[6] = fmin(u[6], err); u[7] = fmax(u[7], err);
        }
    }
    stat_block_combine(u, red);
    if (threadIdx.x == 0) {
        const double cnt = u[3], mean = u[1] / cnt, var = fmax(0.0, u[2] / cnt - mean * mean);
        out[4 * b + 0] = sf / total;
        out[4 * b + 1] = 1 - (u[5] - u[4]) / (u[5] + u[4]);
        out[4 * b + 2] = cnt * (u[7] - u[6]);
        out[4 * b + 3] = cnt * ::sqrt(var);
    }
}

// ---- MultiplaneHologram._farfield2nearfield (_multiplane.py:255-279) -----------------------------------
// phase = atan2( sum_k w_k * nf_k * exp(-i kernel_k) ), written into every child's phase buffer
// (the children share one phase array in the reference).
constexpr int MP_MAX = 16;
template <typename R> struct MpArgs {
    int n;
    size_t S;       // SLM pixels per hologram
    size_t total;   // batch * S
    const Cx<R>* nf[MP_MAX];
    const R* kern[MP_MAX];   // or nullptr
    R w[MP_MAX];
    R* phase[MP_MAX];
};
template <typename R> __global__ void multiplane_combine(MpArgs<R> a) {
    using M = Math<R>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const size_t p = i % a.S;
    Cx<R> acc = mk<R>(0, 0);
    for (int k = 0; k < a.n; ++k) {
        Cx<R> v = a.nf[k][i];
        if (a.kern[k] != nullptr) {
            R sn, cs;
            M::sincos(a.kern[k][p], &sn, &cs);
            v = mk<R>(v.x * cs + v.y * sn, v.y * cs - v.x * sn);   // v * exp(-i kernel)
        }
        acc = acc + v * a.w[k];
    }
    const R ph = M::atan2(acc.y, acc.x);
    for (int k = 0; k < a.n; ++k) a.phase[k][i] = ph;
}

// ---- sparse targets: which columns hold a non-zero (or NaN) weight or target --------------------------
// grid = (Pw, batch), one workgroup per column (contiguous Ph values of each array)
// sig_rows[col] (optional): bit m = some pixel of register slot m of the lane-major column layout (stored position % 16) holds a
// finite non-zero target -- which 16-byte quarters of a lane's weights / targets col_presum_kernel has to fetch
template <typename R> __global__ void scan_active_cols(const R* w, const R* t, int Ph, int Pw, unsigned char* active,
                                                       unsigned short* sig_rows = nullptr) {
    // active[col]: bit 0 = the column holds a non-zero (or NaN) weight or target; bit 1 = a finite non-zero target (under
    // MRAF only there is the weighted part of the constrained field non-zero, whatever the weights: NaN and zero targets
    // override it, :1606-1653); bit 2 = a NaN target (MRAF noise pixel)
    __shared__ int any;
    const int col = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const size_t base = ((size_t)b * Pw + col) * Ph;
    bool nz = false, sig = false, noise = false;
    unsigned slots = 0;
    for (int i = threadIdx.x; i < Ph; i += blockDim.x) {
        const R wv = w[base + i], tv = t[base + i];
        nz = nz || !(wv == (R)0) || !(tv == (R)0);      // NaN counts as active
        const bool sg = (tv == tv && tv != (R)0);
        sig = sig || sg;
        slots |= sg ? (1u << (i & 15)) : 0u;
        noise = noise || (tv != tv);
    }
    const int bits = (__builtin_amdgcn_ballot_w64(nz) != 0 ? 1 : 0) | (__builtin_amdgcn_ballot_w64(sig) != 0 ? 2 : 0) |
                     (__builtin_amdgcn_ballot_w64(noise) != 0 ? 4 : 0);
    if (bits != 0 && (threadIdx.x & 63) == 0) atomicOr(&any, bits);
    if (sig_rows != nullptr && slots != 0) atomicOr(&any, (int)(slots << 8));
    __syncthreads();
    if (threadIdx.x == 0) {
        active[(size_t)b * Pw + col] = (unsigned char)(any & 0xff);
        if (sig_rows != nullptr) sig_rows[(size_t)b * Pw + col] = (unsigned short)((unsigned)any >> 8);
    }
}
// dil[c] = any active[c - d], d in [lo, hi]: the columns a w-wide integration window around a spot
// column touches (analysis.take offsets floor(-(w-1)/2) ...).  grid = (ceil(Pw/256), batch)
static __global__ void dilate_active_cols(const unsigned char* active, int Pw, int lo, int hi, unsigned char* dil) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= Pw) return;
    unsigned char on = 0;
    for (int d = lo; d <= hi; ++d) {
        const int s = c - d;
        if (s >= 0 && s < Pw && active[(size_t)b * Pw + s]) on = 1;
    }
    dil[(size_t)b * Pw + c] = on;
}
// one workgroup per hologram: ordered compaction of the active columns
// (mask: the bits of scan_active_cols that count -- 0xff: any, 4: the columns that hold a NaN target)
static __global__ void compact_active_cols(const unsigned char* active, int Pw, int* list, int* n_active,
                                           unsigned short* lane_mask, int mask = 0xff) {
    __shared__ int base;
    __shared__ int wsum[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < Pw; c0 += blockDim.x) {
        const int c = c0 + threadIdx.x;
        const bool on = c < Pw && (active[(size_t)b * Pw + c] & mask) != 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
        if (lane == 0) wsum[wid] = __builtin_popcountll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wid; ++k) off += wsum[k];
        if (on) list[(size_t)b * Pw + off + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = 0;
            for (int k = 0; k < nw; ++k) tot += wsum[k];
            base += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_active[b] = base;
    const int T = Pw / 16;
    for (int j = threadIdx.x; j < T; j += blockDim.x) {
        unsigned m16 = 0;
        for (int m = 0; m < 16; ++m) m16 |= (unsigned)((active[(size_t)b * Pw + j + m * T] & mask) != 0) << m;
        lane_mask[(size_t)b * T + j] = (unsigned short)m16;
    }
}

// WGS-Nogrette on the fused path: nog[b] = -1 / nanmean(fc) from the column kernel's partial sums; columns the
// sparse path did not visit hold T = 0 everywhere, i.e. fc = 1 per pixel.
// Row-kernel view of one bit of scan_active_cols: bit m of lane_mask[b][j] = column j + m * Pw / 16 has `bit` set.
// bit 4 (a NaN target in the column) gives the columns where the noise part of a single-pass MRAF field exists at all.
static __global__ void flag_lane_mask(const unsigned char* active, int Pw, int bit, unsigned short* lane_mask) {
    const int b = blockIdx.y, T = Pw / 16;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= T) return;
    unsigned m16 = 0;
    for (int m = 0; m < 16; ++m) m16 |= (unsigned)((active[(size_t)b * Pw + j + m * T] & bit) != 0) << m;
    lane_mask[(size_t)b * T + j] = (unsigned short)m16;
}

template <typename R>
__global__ void nog_finalize(const double* sum, const int* n_active, int Ph, int Pw, R* nog, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double skipped = n_active != nullptr ? (double)(Pw - n_active[b]) * (double)Ph : 0.0;
    nog[b] = (R)(-(1.0 / ((sum[b] + skipped) / ((double)Ph * (double)Pw))));
}

}  // namespace hgs
