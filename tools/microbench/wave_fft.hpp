// EXPERIMENT (round 2, not part of libhgs.so; measured by wave_row.hip, result in DESIGN.md section 6).
// One-wave 4096-point complex transform for gfx950: a wavefront of 64 lanes owns a whole line, 64 elements per
// lane in VGPRs (4096 = 64 x 64).  An alternative to the workgroup transform of csrc/fft_core.hpp for the row half
// of cp.fft.fft2 / ifft2 (_hologram.py:1048,1070) on the 4096-wide fp32 path.
//
//   lane n0, register n1 holds x[n0 + 64 n1]                      (global accesses: 64 consecutive elements per wave)
//   1. per lane, 64-point DFT over n1 in registers                 A[ka] = sum_n1 x[n0 + 64 n1] W64^(n1 ka)
//   2. twiddle                                                     A[ka] *= W4096^(n0 ka)
//   3. 64 x 64 transpose across the lanes of the wave through LDS  lane ka, register n0
//   4. per lane, 64-point DFT over n0                              X[ka + 64 kb] = sum_n0 A_n0[ka] W64^(n0 kb)
//   -> lane ka, register kb holds X[ka + 64 kb]: the same layout as the input, for both directions.
//
// Against the workgroup transform of fft_core.hpp (16 elements per lane, 256 lanes): ONE exchange instead of two,
// NO workgroup barrier (the exchange is private to the wave: LDS operations of a wave execute in issue order),
// address arithmetic and predicates amortised over 64 elements per lane, and waves that are scheduled
// independently (a row is a wave, not a workgroup).  The exchange moves the real and the imaginary parts in two
// passes through the same 64 x 65 float image (16.6 KB per wave: eight waves per CU fit the 160 KB of LDS).
//
// Zero padding: only registers [LO, HI) of the space side are ever non-zero (the SLM columns inside the padded
// row); the first radix-4 layer of step 1 skips the zero inputs (dft4_mask) and whatever the caller does not
// read of an inverse transform's output is removed by the compiler (everything is straight-line code on
// compile-time register indices).
#pragma once
#include "../../slmsuite_amd/csrc/fft_core.hpp"

// The register budget is 256 VGPRs for 128 of data: keep the scheduler from interleaving independent
// sub-transforms (it trades registers for latency hiding that two waves per SIMD already provide)
#ifndef WAVE_FFT_SCHED_BARRIERS
#define WAVE_FFT_SCHED_BARRIERS 1
#endif
#if WAVE_FFT_SCHED_BARRIERS
#define WAVE_FFT_SCHED() __builtin_amdgcn_sched_barrier(0)
#else
#define WAVE_FFT_SCHED() ((void)0)
#endif

namespace hgs {

// ---- e^{DIR 2 pi i Q / 64}, exact octant symmetry ------------------------------------------------------
struct W64 {
    static constexpr double C8[9] = {1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494,
                                     0.92387953251128675613, 0.88192126434835502971, 0.83146961230254523708,
                                     0.77301045336273696081, 0.70710678118654752440};
    static constexpr double S8[9] = {0.0, 0.09801714032956060199, 0.19509032201612826785, 0.29028467725446236764,
                                     0.38268343236508977173, 0.47139673682599764856, 0.55557023301960222474,
                                     0.63439328416364549822, 0.70710678118654752440};
    static constexpr double c16(int r) { return r <= 8 ? C8[r] : S8[16 - r]; }
    static constexpr double s16(int r) { return r <= 8 ? S8[r] : C8[16 - r]; }
    static constexpr double cosq(int q) {
        const int quad = q / 16, r = q % 16;
        return quad == 0 ? c16(r) : quad == 1 ? -s16(r) : quad == 2 ? -c16(r) : s16(r);
    }
    static constexpr double sinq(int q) {
        const int quad = q / 16, r = q % 16;
        return quad == 0 ? s16(r) : quad == 1 ? c16(r) : quad == 2 ? -s16(r) : -c16(r);
    }
};

template <int Q, int DIR> __device__ __forceinline__ v2f rot64(v2f a) {
    constexpr int q = ((Q % 64) + 64) % 64;
    if constexpr (q % 4 == 0) {
        return rot16<q / 4, DIR>(a);
    } else {
        constexpr float c = (float)W64::cosq(q);
        constexpr float s = (float)(DIR < 0 ? -W64::sinq(q) : W64::sinq(q));
        return __builtin_elementwise_fma(cswap(a), (v2f){-s, s}, a * (v2f){c, c});
    }
}

// 4-point DFT of which only the inputs named in mask M (bit i = input i) are non-zero; the others are not read
template <int DIR, int M> __device__ __forceinline__ void dft4_mask(v2f& v0, v2f& v1, v2f& v2, v2f& v3) {
    constexpr bool n0 = (M & 1) != 0, n1 = (M & 2) != 0, n2 = (M & 4) != 0, n3 = (M & 8) != 0;
    constexpr bool ze = !n0 && !n2, zo = !n1 && !n3;
    constexpr float s = DIR < 0 ? 1.f : -1.f;
    const v2f z = {0.f, 0.f};
    if constexpr (M == 15) {
        dft4<DIR>(v0, v1, v2, v3);
    } else if constexpr (ze && zo) {
        v0 = z; v1 = z; v2 = z; v3 = z;
    } else {
        v2f a0 = z, a1 = z, a2 = z, d = z;
        if constexpr (n0 && n2) { a0 = v0 + v2; a1 = v0 - v2; }
        else if constexpr (n0) { a0 = v0; a1 = v0; }
        else if constexpr (n2) { a0 = v2; a1 = -v2; }
        if constexpr (n1 && n3) { a2 = v1 + v3; d = v1 - v3; }
        else if constexpr (n1) { a2 = v1; d = v1; }
        else if constexpr (n3) { a2 = v3; d = -v3; }
        if constexpr (zo) {
            v0 = a0; v2 = a0; v1 = a1; v3 = a1;
        } else if constexpr (ze) {
            const v2f rd = cswap(d) * (v2f){s, -s};      // e^{DIR i pi/2} d
            v0 = a2; v2 = -a2; v1 = rd; v3 = -rd;
        } else {
            v0 = a0 + a2;
            v2 = a0 - a2;
            v1 = cadd_rot4<DIR>(a1, d);
            v3 = csub_rot4<DIR>(a1, d);
        }
    }
}

constexpr int mask4(int a, int lo, int hi) {
    int m = 0;
    for (int b = 0; b < 4; ++b)
        if (a + 4 * b >= lo && a + 4 * b < hi) m |= 1 << b;
    return m;
}

// 16-point DFT, natural order in and out, inputs outside [RLO, RHI) zero (not read)
template <int DIR, int RLO, int RHI> __device__ __forceinline__ void dft16_range(v2f (&v)[16]) {
    if constexpr (RLO <= 0 && RHI >= 16) {
        Dft<16, DIR, float>::run(v);
    } else {
        dft4_mask<DIR, mask4(0, RLO, RHI)>(v[0], v[4], v[8], v[12]);
        dft4_mask<DIR, mask4(1, RLO, RHI)>(v[1], v[5], v[9], v[13]);
        dft4_mask<DIR, mask4(2, RLO, RHI)>(v[2], v[6], v[10], v[14]);
        dft4_mask<DIR, mask4(3, RLO, RHI)>(v[3], v[7], v[11], v[15]);
        v[5] = rot16<1, DIR>(v[5]);   v[6] = rot16<2, DIR>(v[6]);   v[7] = rot16<3, DIR>(v[7]);
        v[9] = rot16<2, DIR>(v[9]);   v[10] = rot16<4, DIR>(v[10]); v[11] = rot16<6, DIR>(v[11]);
        v[13] = rot16<3, DIR>(v[13]); v[14] = rot16<6, DIR>(v[14]); v[15] = rot16<9, DIR>(v[15]);
        dft4<DIR>(v[0], v[1], v[2], v[3]);
        dft4<DIR>(v[4], v[5], v[6], v[7]);
        dft4<DIR>(v[8], v[9], v[10], v[11]);
        dft4<DIR>(v[12], v[13], v[14], v[15]);
        v2f t;
        t = v[1]; v[1] = v[4]; v[4] = t;
        t = v[2]; v[2] = v[8]; v[8] = t;
        t = v[3]; v[3] = v[12]; v[12] = t;
        t = v[6]; v[6] = v[9]; v[9] = t;
        t = v[7]; v[7] = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
}

// 64-point DFT in registers, natural order in and out.  r = r1 + 4 r2 (r1 < 4, r2 < 16), p = 16 p1 + p2:
//   V[16 p1 + p2] = sum_r1 W4^(r1 p1) [ W64^(r1 p2) sum_r2 v[r1 + 4 r2] W16^(r2 p2) ]
// Inputs outside [LO, HI) (multiples of 4) are zero and not read.
template <int DIR, int LO, int HI> __device__ __forceinline__ void dft64(v2f (&v)[64]) {
    static_assert(LO % 4 == 0 && HI % 4 == 0, "dft64: range in whole radix-4 groups");
    static_for<0, 4>([&](auto r1_) {
        constexpr int r1 = r1_;
        v2f t[16];
        static_for<0, 16>([&](auto i_) {
            constexpr int i = i_;
            if constexpr (r1 + 4 * i >= LO && r1 + 4 * i < HI) t[i] = v[r1 + 4 * i];
            else t[i] = (v2f){0.f, 0.f};
        });
        dft16_range<DIR, LO / 4, HI / 4>(t);
        static_for<0, 16>([&](auto p2_) { constexpr int p2 = p2_; v[r1 + 4 * p2] = rot64<r1 * p2, DIR>(t[p2]); });
        WAVE_FFT_SCHED();
    });
    v2f o[64];
    static_for<0, 16>([&](auto p2_) {
        constexpr int p2 = p2_;
        dft4<DIR>(v[4 * p2], v[4 * p2 + 1], v[4 * p2 + 2], v[4 * p2 + 3]);
        o[p2] = v[4 * p2]; o[16 + p2] = v[4 * p2 + 1]; o[32 + p2] = v[4 * p2 + 2]; o[48 + p2] = v[4 * p2 + 3];
        if constexpr (p2 % 4 == 3) WAVE_FFT_SCHED();
    });
    static_for<0, 64>([&](auto i_) { constexpr int i = i_; v[i] = o[i]; });
}

struct WaveFft4096 {
    static constexpr int N = 4096;
    static constexpr int LDS_FLOATS = 64 * 65;          // one wave's exchange image
    static constexpr size_t LDS_BYTES = LDS_FLOATS * sizeof(float);
    // W^(lane l), l = 1..7 and W^(8 lane h), h = 1..7 (exact table values; a twiddle is one product of the two).
    // Fetched at the start of every transform (L2-resident table, the latency hides under the first 64-point pass):
    // 28 VGPRs that are then free during the second pass and whatever the caller does between two transforms.
    v2f t1[7], t2[7];
    const v2f* table_;
    int lane_;

    __device__ __forceinline__ void init(const v2f* table, int lane) { table_ = table; lane_ = lane; }
    __device__ __forceinline__ void fetch() {
        static_for<0, 7>([&](auto i_) {
            constexpr int i = i_;
            t1[i] = table_[(lane_ * (i + 1)) & (N - 1)];
            t2[i] = table_[(lane_ * 8 * (i + 1)) & (N - 1)];
        });
    }

    // a[ka] *= W^(DIR-signed lane ka), ka = 8 h + l
    template <int DIR> __device__ __forceinline__ void twiddle(v2f (&v)[64]) const {
        static_for<1, 64>([&](auto k_) {
            constexpr int k = k_, h = k / 8, l = k % 8;
            v2f x = v[k];
            if constexpr (l != 0) x = DIR < 0 ? cmul(x, t1[l - 1]) : cmulc(x, t1[l - 1]);
            if constexpr (h != 0) x = DIR < 0 ? cmul(x, t2[h - 1]) : cmulc(x, t2[h - 1]);
            v[k] = x;
            if constexpr (k % 8 == 7) WAVE_FFT_SCHED();
        });
    }

    // lane L, register k  ->  lane k, register L  (two passes: real parts, imaginary parts)
    // The gather side is written as single ds_read_b32 instructions by hand: the compiler would merge neighbours
    // into ds_read2_b32, whose register pair then holds the real parts of two elements and costs a v_mov each to
    // rejoin its imaginary part.  Sixteen reads and their s_waitcnt form ONE asm statement, so nothing the
    // compiler inserts (copies, spills) can touch a destination register before its data has landed.
    template <int N0> static __device__ __forceinline__ void gather16(float (&x)[64], unsigned rd) {
#define WF_O(i) "=&v"(x[N0 + i])
        asm volatile(
            "ds_read_b32 %0, %16 offset:%17\n\tds_read_b32 %1, %16 offset:%17+4\n\tds_read_b32 %2, %16 offset:%17+8\n\t"
            "ds_read_b32 %3, %16 offset:%17+12\n\tds_read_b32 %4, %16 offset:%17+16\n\tds_read_b32 %5, %16 offset:%17+20\n\t"
            "ds_read_b32 %6, %16 offset:%17+24\n\tds_read_b32 %7, %16 offset:%17+28\n\tds_read_b32 %8, %16 offset:%17+32\n\t"
            "ds_read_b32 %9, %16 offset:%17+36\n\tds_read_b32 %10, %16 offset:%17+40\n\tds_read_b32 %11, %16 offset:%17+44\n\t"
            "ds_read_b32 %12, %16 offset:%17+48\n\tds_read_b32 %13, %16 offset:%17+52\n\tds_read_b32 %14, %16 offset:%17+56\n\t"
            "ds_read_b32 %15, %16 offset:%17+60\n\ts_waitcnt lgkmcnt(0)"
            : WF_O(0), WF_O(1), WF_O(2), WF_O(3), WF_O(4), WF_O(5), WF_O(6), WF_O(7), WF_O(8), WF_O(9), WF_O(10), WF_O(11),
              WF_O(12), WF_O(13), WF_O(14), WF_O(15)
            : "v"(rd), "n"(4 * N0)
            : "memory");
#undef WF_O
    }
    static __device__ __forceinline__ void transpose(v2f (&v)[64], float* buf, int lane) {
        typedef __attribute__((address_space(3))) float lds_float;
        float* wr = buf + lane;
        const unsigned rd = (unsigned)(uintptr_t)(lds_float*)(buf + lane * 65);
        float xr[64], xi[64];
        wave_lds_order();
        static_for<0, 64>([&](auto k_) { constexpr int k = k_; wr[k * 65] = v[k].x; });
        wave_lds_order();
        gather16<0>(xr, rd); gather16<16>(xr, rd); gather16<32>(xr, rd); gather16<48>(xr, rd);
        static_for<0, 64>([&](auto k_) { constexpr int k = k_; wr[k * 65] = v[k].y; });
        wave_lds_order();
        gather16<0>(xi, rd); gather16<16>(xi, rd); gather16<32>(xi, rd); gather16<48>(xi, rd);
        static_for<0, 64>([&](auto n_) { constexpr int n = n_; v[n] = (v2f){xr[n], xi[n]}; });
    }

    // DIR = -1 forward, +1 inverse (unnormalised).  Inputs outside registers [LO, HI) are zero.
    template <int DIR, int LO = 0, int HI = 64>
    __device__ __forceinline__ void run(v2f (&v)[64], float* buf, int lane) {
        first<DIR, LO, HI>(v, buf, lane);
        second<DIR>(v);
    }
    // The two halves separately: a caller issues the loads its next step needs between them (their latency hides
    // behind the second 64-point pass, their registers are not live during the first)
    template <int DIR, int LO = 0, int HI = 64>
    __device__ __forceinline__ void first(v2f (&v)[64], float* buf, int lane) {
        // every transform derives its ~30 table / LDS addresses from the lane index afresh: kept across two
        // transforms of a kernel they would cost more registers than the few integer operations that rebuild them
        asm volatile("" : "+v"(lane));
        lane_ = lane;
        fetch();
        dft64<DIR, LO, HI>(v);
        twiddle<DIR>(v);
        transpose(v, buf, lane);
    }
    template <int DIR> static __device__ __forceinline__ void second(v2f (&v)[64]) { dft64<DIR, 0, 64>(v); }
};

}  // namespace hgs
