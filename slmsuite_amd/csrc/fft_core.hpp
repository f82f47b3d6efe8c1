// In-workgroup power-of-two complex FFT for gfx950 (CDNA4), used by every transform kernel of the
// hologram engine (replaces cp.fft.fft2 / ifft2 of the reference, _hologram.py:1048,1070).
//
// Design (MI355X-first, not a port of any FFT library):
//   * One length-N transform is owned by T = N/16 lanes; lane j keeps the 16 elements
//     x[j + m*T], m = 0..15, in VGPRs ("load layout").  For N = 4096 that is one 256-lane
//     workgroup (4 wavefronts of 64) per transform, 32 data VGPRs per lane.
//   * Stockham autosort, radix-16 stages (4096 = 16*16*16): butterflies run entirely in registers;
//     between stages the partial results are scattered to LDS and re-gathered in load layout, so
//     global loads AND global stores of the owning kernel are fully coalesced 512-B wave accesses
//     and no bit-reversal pass exists.  The last stage needs no exchange: its outputs already sit
//     in load layout (proved in tools/stockham_model.py).
//   * Twiddles W_N^(r*k) are per-lane constants of a stage; they are fetched once per kernel from a
//     table computed in double on the host and stay in VGPRs across all transforms a lane performs
//     (forward and inverse share them: the inverse multiplies by the conjugate).
//   * LDS image is padded by one element per 16 (pad(q) = q + q/16) which makes the radix-16
//     scatter (stride 16 between neighbouring lanes) and the stride-1 gather conflict-free for
//     8-byte ds_write_b64/ds_read_b64 (bank = element mod 32).
//   * No MFMA: a radix-16 butterfly is 144 adds + a few constant rotations, bandwidth-bound work.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

// Ablation hooks for tools/microbench/ablate.hip (all 0 in the product build).
#ifndef HGS_ABL_XCHG
#define HGS_ABL_XCHG 0
#endif
#ifndef HGS_ABL_BFLY
#define HGS_ABL_BFLY 0
#endif
#ifndef HGS_ABL_TRANS
#define HGS_ABL_TRANS 0
#endif
// Wave-priority experiments (A/B builds): 0 none; 1 static, by the parity of the hardware wave slot (the two waves
// sharing a SIMD get different priorities); 2 raised around the butterflies, lowered around the exchanges
#ifndef HGS_PRIO
#define HGS_PRIO 0
#endif
// Timeline instrumentation for tools/microbench/trace.hip: lane 0 of every wave stamps s_memtime at phase
// boundaries into dynamic LDS at byte offset HGS_TRACE_OFF (128 events per wave); 0 in the product build.
#ifndef HGS_TRACE
#define HGS_TRACE 0
#endif
#ifndef HGS_TRACE_OFF
#define HGS_TRACE_OFF 49152
#endif
#ifndef HGS_TRACE_SKIP
#define HGS_TRACE_SKIP 0          // events of a wave that are passed over before the first one is recorded
#endif

namespace hgs {

#if HGS_TRACE
__device__ __forceinline__ void trace_event(int& n, int ev) {
    extern __shared__ __attribute__((aligned(16))) char trace_smem[];
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0 && n >= HGS_TRACE_SKIP && n < HGS_TRACE_SKIP + 128)
        reinterpret_cast<unsigned long long*>(trace_smem + HGS_TRACE_OFF)[(threadIdx.x >> 6) * 128 + (n - HGS_TRACE_SKIP)] =
            (t & 0x00ffffffffffffffull) | ((unsigned long long)ev << 56);
    ++n;
}
#define HGS_T(n, ev) trace_event(n, ev)
#else
#define HGS_T(n, ev) ((void)0)
#endif

// Complex numbers are 2-vectors (x = re, y = im) held in an aligned VGPR pair, so that complex
// add/sub are ONE packed instruction (v_pk_add_f32) and a complex multiply is two (v_pk_mul_f32 +
// v_pk_fma_f32 with op_sel/neg modifiers).  Measured on MI355X (tools/microbench/valu_rate.hip): a
// wave64 scalar fp32 VALU op occupies its SIMD for ~4.2 cycles, v_pk_fma_f32 for ~4.45 while doing
// twice the work -- the transform kernels are VALU-issue bound, so packed math is the lever.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
template <typename R> struct CxSel;
template <> struct CxSel<float> { using type = v2f; };
template <> struct CxSel<double> { using type = v2d; };
template <typename R> using Cx = typename CxSel<R>::type;
template <typename V> struct RealOf;
template <> struct RealOf<v2f> { using type = float; };
template <> struct RealOf<v2d> { using type = double; };

template <typename R> __device__ __forceinline__ Cx<R> mk(R x, R y) { return (Cx<R>){x, y}; }
template <typename V> __device__ __forceinline__ V cswap(V a) { return __builtin_shufflevector(a, a, 1, 0); }

// a * b
template <typename V> __device__ __forceinline__ V cmul(V a, V b) {
    if constexpr (std::is_same<V, v2f>::value) {
        v2f t, r;
        // t = a.xx * b ; r = a.yy * (-b.y, b.x) + t
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
            : "=v"(r) : "v"(a), "v"(b), "v"(t));
        return r;
    } else {
        return (V){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    }
}
// a * conj(b)
template <typename V> __device__ __forceinline__ V cmulc(V a, V b) {
    if constexpr (std::is_same<V, v2f>::value) {
        v2f t, r;
        // t = a.xx * (b.x, -b.y) ; r = a.yy * (b.y, b.x) + t
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
            : "=v"(r) : "v"(a), "v"(b), "v"(t));
        return r;
    } else {
        return (V){a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
    }
}

// Four values times the same factor w (DIR < 0) or conj(w) (DIR > 0), in place.  fp32: ONE asm statement with the four
// multiplies ahead of the four dependent fused multiply-adds -- as single cmul()s the two instructions of a product
// sit back to back in the instruction stream (the compiler cannot see into the asm to interleave them) with a hazard
// s_nop between them, and all products of a group chain through one temporary.
template <int DIR, typename V> __device__ __forceinline__ void cmul4(V& a0, V& a1, V& a2, V& a3, V w) {
    if constexpr (std::is_same<V, v2f>::value) {
        v2f t0, t1, t2, t3;
        if constexpr (DIR < 0) {
            asm("v_pk_mul_f32 %4, %0, %8 op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %5, %1, %8 op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %6, %2, %8 op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %7, %3, %8 op_sel_hi:[0,1]\n\t"
                "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %1, %1, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %2, %2, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %3, %3, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(w));
        } else {
            asm("v_pk_mul_f32 %4, %0, %8 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %5, %1, %8 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %6, %2, %8 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %7, %3, %8 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_fma_f32 %0, %0, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %1, %1, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %2, %2, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
                "v_pk_fma_f32 %3, %3, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(w));
        }
    } else {
        a0 = DIR < 0 ? cmul(a0, w) : cmulc(a0, w);
        a1 = DIR < 0 ? cmul(a1, w) : cmulc(a1, w);
        a2 = DIR < 0 ? cmul(a2, w) : cmulc(a2, w);
        a3 = DIR < 0 ? cmul(a3, w) : cmulc(a3, w);
    }
}

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// DIR = -1 forward (e^{-i...}), +1 inverse.
// a + e^{DIR*i*pi/2} b  and  a - e^{DIR*i*pi/2} b : one packed fma each (swap folds into op_sel,
// the (+-1, -+1) pair is a scalar-register constant)
template <int DIR, typename V> __device__ __forceinline__ V cadd_rot4(V a, V b) {
    using R = typename RealOf<V>::type;
    constexpr R s = DIR < 0 ? (R)1 : (R)-1;     // forward: a + (-i) b = (a.x + b.y, a.y - b.x)
    return __builtin_elementwise_fma(cswap(b), (V){s, -s}, a);
}
template <int DIR, typename V> __device__ __forceinline__ V csub_rot4(V a, V b) {
    using R = typename RealOf<V>::type;
    constexpr R s = DIR < 0 ? (R)1 : (R)-1;
    return __builtin_elementwise_fma(cswap(b), (V){-s, s}, a);
}

// multiply by e^{DIR * 2*pi*i * Q / 16}, Q compile-time:  a*(c,c) + swap(a)*(-s,s)
template <int Q, int DIR, typename V> __device__ __forceinline__ V rot16(V a) {
    using R = typename RealOf<V>::type;
    constexpr int q = ((Q % 16) + 16) % 16;
    constexpr R C1 = (R)0.92387953251128675613;  // cos(pi/8)
    constexpr R S1 = (R)0.38268343236508977173;  // sin(pi/8)
    constexpr R H = (R)0.70710678118654752440;   // sqrt(1/2)
    constexpr R cs[16] = {1, C1, H, S1, 0, -S1, -H, -C1, -1, -C1, -H, -S1, 0, S1, H, C1};
    constexpr R sn[16] = {0, S1, H, C1, 1, C1, H, S1, 0, -S1, -H, -C1, -1, -C1, -H, -S1};
    constexpr R c = cs[q];
    constexpr R s = DIR < 0 ? -sn[q] : sn[q];
    if constexpr (q == 0) return a;
    else if constexpr (q == 8) return -a;
    else if constexpr (q == 4 || q == 12) return cswap(a) * (V){-s, s};
    else return __builtin_elementwise_fma(cswap(a), (V){-s, s}, a * (V){c, c});
}

template <int DIR, typename V> __device__ __forceinline__ void dft2(V& a, V& b) {
    V t = a - b;
    a = a + b;
    b = t;
}

// 4-point DFT on (v0,v1,v2,v3), natural order in and out: 8 packed ops
template <int DIR, typename V>
__device__ __forceinline__ void dft4(V& v0, V& v1, V& v2, V& v3) {
    V a0 = v0 + v2, a1 = v0 - v2, a2 = v1 + v3, d = v1 - v3;
    v0 = a0 + a2;
    v2 = a0 - a2;
    v1 = cadd_rot4<DIR>(a1, d);
    v3 = csub_rot4<DIR>(a1, d);
}

// 4-point DFT of which only the first K inputs are non-zero (the others are not read): K = 4 is dft4
template <int DIR, int K, typename V>
__device__ __forceinline__ void dft4_lead(V& v0, V& v1, V& v2, V& v3) {
    static_assert(K >= 1 && K <= 4, "dft4_lead");
    if constexpr (K == 4) {
        dft4<DIR>(v0, v1, v2, v3);
    } else if constexpr (K == 1) {
        v1 = v0; v2 = v0; v3 = v0;
    } else {
        V a0 = v0, a1 = v0;
        if constexpr (K == 3) { a0 = v0 + v2; a1 = v0 - v2; }
        const V b = v1;
        v0 = a0 + b;
        v2 = a0 - b;
        v1 = cadd_rot4<DIR>(a1, b);
        v3 = csub_rot4<DIR>(a1, b);
    }
}

// 4-point DFT of which only the first K outputs are wanted (the others are left undefined): K = 4 is dft4
template <int DIR, int K, typename V>
__device__ __forceinline__ void dft4_trail(V& v0, V& v1, V& v2, V& v3) {
    static_assert(K >= 0 && K <= 4, "dft4_trail");
    if constexpr (K == 4) {
        dft4<DIR>(v0, v1, v2, v3);
    } else if constexpr (K >= 1) {
        const V a0 = v0 + v2, a2 = v1 + v3;
        if constexpr (K >= 2) {
            const V a1 = v0 - v2, d = v1 - v3;
            v1 = cadd_rot4<DIR>(a1, d);
        }
        v0 = a0 + a2;
        if constexpr (K == 3) v2 = a0 - a2;
    }
}

// R-point DFT, v[r] natural order in, V[p] natural order out (in place).
template <int RADIX, int DIR, typename R> struct Dft;

template <int DIR, typename R> struct Dft<2, DIR, R> {
    static __device__ __forceinline__ void run(Cx<R> (&v)[2]) { dft2<DIR>(v[0], v[1]); }
};
template <int DIR, typename R> struct Dft<4, DIR, R> {
    static __device__ __forceinline__ void run(Cx<R> (&v)[4]) { dft4<DIR>(v[0], v[1], v[2], v[3]); }
};
template <int DIR, typename R> struct Dft<8, DIR, R> {
    // r = r1 + 2*r2 (r1<2, r2<4), p = 4*p1 + p2:  V[4p1+p2] = sum_r1 w2^(r1 p1) w8^(r1 p2) DFT4_r2(v[r1+2r2])[p2]
    static __device__ __forceinline__ void run(Cx<R> (&v)[8]) {
        dft4<DIR>(v[0], v[2], v[4], v[6]);
        dft4<DIR>(v[1], v[3], v[5], v[7]);
        Cx<R> t1 = rot16<2, DIR>(v[3]), t2 = rot16<4, DIR>(v[5]), t3 = rot16<6, DIR>(v[7]);
        Cx<R> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[1] = e1 + t1; v[5] = e1 - t1;
        v[2] = e2 + t2; v[6] = e2 - t2;
        v[3] = e3 + t3; v[7] = e3 - t3;
    }
};
template <int DIR, typename R> struct Dft<16, DIR, R> {
    // r = r1 + 4*r2, p = 4*p1 + p2:
    //   V[4p1+p2] = sum_r1 w4^(r1 p1) * [ w16^(r1 p2) * sum_r2 v[r1+4r2] w4^(r2 p2) ]
    // TW: the stage twiddle W^(r k) = W^(r1 k) * W^(4 r2 k) is split (fewer twiddle registers):
    // the caller pre-multiplies v[r1+4r2] by W^(4 r2 k); the common factor W^(r1 k) = bt[r1-1]
    // of each inner 4-point transform is applied here to its outputs.
    // NOUT: only the outputs V[0 .. NOUT-1] are wanted (the rows / columns outside the SLM are dropped after an
    // inverse transform): the last radix-4 layer shrinks to them (NOUT = 6: 18 packed operations instead of 32).
    template <bool TW, int NOUT = 16>
    static __device__ __forceinline__ void run_tw(Cx<R> (&v)[16], Cx<R> bt1, Cx<R> bt2, Cx<R> bt3) {
        // step 1: DFT4 over r2 for each r1 -> t[r1][p2] stored at v[r1 + 4*p2]
        dft4<DIR>(v[0], v[4], v[8], v[12]);
        dft4<DIR>(v[1], v[5], v[9], v[13]);
        dft4<DIR>(v[2], v[6], v[10], v[14]);
        dft4<DIR>(v[3], v[7], v[11], v[15]);
        if constexpr (TW) {
            cmul4<DIR>(v[1], v[5], v[9], v[13], bt1);
            cmul4<DIR>(v[2], v[6], v[10], v[14], bt2);
            cmul4<DIR>(v[3], v[7], v[11], v[15], bt3);
        }
        // step 2: twiddle t[r1][p2] *= w16^(r1*p2)
        v[5] = rot16<1, DIR>(v[5]);   v[6] = rot16<2, DIR>(v[6]);   v[7] = rot16<3, DIR>(v[7]);
        v[9] = rot16<2, DIR>(v[9]);   v[10] = rot16<4, DIR>(v[10]); v[11] = rot16<6, DIR>(v[11]);
        v[13] = rot16<3, DIR>(v[13]); v[14] = rot16<6, DIR>(v[14]); v[15] = rot16<9, DIR>(v[15]);
        // step 3: DFT4 over r1 for each p2 -> V[4*p1 + p2]; inputs at v[r1 + 4*p2]
        // (of output group p2 the first ceil((NOUT - p2) / 4) are wanted)
        constexpr auto want = [](int p2) { const int k = (NOUT - p2 + 3) / 4; return k < 0 ? 0 : (k > 4 ? 4 : k); };
        dft4_trail<DIR, want(0)>(v[0], v[1], v[2], v[3]);      // p2 = 0 -> V[0], V[4], V[8], V[12]
        dft4_trail<DIR, want(1)>(v[4], v[5], v[6], v[7]);      // p2 = 1 -> V[1], V[5], V[9], V[13]
        dft4_trail<DIR, want(2)>(v[8], v[9], v[10], v[11]);    // p2 = 2
        dft4_trail<DIR, want(3)>(v[12], v[13], v[14], v[15]);  // p2 = 3
        // now v[p1 + 4*p2] holds V[4*p1 + p2]: transpose the 4x4 index to natural order
        Cx<R> t;
        t = v[1]; v[1] = v[4]; v[4] = t;
        t = v[2]; v[2] = v[8]; v[8] = t;
        t = v[3]; v[3] = v[12]; v[12] = t;
        t = v[6]; v[6] = v[9]; v[9] = t;
        t = v[7]; v[7] = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
    static __device__ __forceinline__ void run(Cx<R> (&v)[16]) {
        const Cx<R> z = mk<R>(0, 0);
        run_tw<false>(v, z, z, z);
    }
    // only the outputs V[0 .. NOUT-1] are wanted
    template <int NOUT> static __device__ __forceinline__ void run_trail(Cx<R> (&v)[16]) {
        const Cx<R> z = mk<R>(0, 0);
        run_tw<false, NOUT>(v, z, z, z);
    }
    // The same transform when only the first NZ inputs are non-zero (zero-padded fields: the rows outside the SLM):
    // the first radix-4 layer shrinks to the non-zero inputs (NZ = 6: 8 packed operations instead of 32).
    template <int NZ> static __device__ __forceinline__ void run_lead(Cx<R> (&v)[16]) {
        static_assert(NZ >= 4 && NZ <= 16, "run_lead: at least one non-zero input per radix-4 butterfly");
        if constexpr (NZ == 16) {
            run(v);
        } else {
            dft4_lead<DIR, (NZ - 0 + 3) / 4>(v[0], v[4], v[8], v[12]);
            dft4_lead<DIR, (NZ - 1 + 3) / 4>(v[1], v[5], v[9], v[13]);
            dft4_lead<DIR, (NZ - 2 + 3) / 4>(v[2], v[6], v[10], v[14]);
            dft4_lead<DIR, (NZ - 3 + 3) / 4>(v[3], v[7], v[11], v[15]);
            v[5] = rot16<1, DIR>(v[5]);   v[6] = rot16<2, DIR>(v[6]);   v[7] = rot16<3, DIR>(v[7]);
            v[9] = rot16<2, DIR>(v[9]);   v[10] = rot16<4, DIR>(v[10]); v[11] = rot16<6, DIR>(v[11]);
            v[13] = rot16<3, DIR>(v[13]); v[14] = rot16<6, DIR>(v[14]); v[15] = rot16<9, DIR>(v[15]);
            dft4<DIR>(v[0], v[1], v[2], v[3]);
            dft4<DIR>(v[4], v[5], v[6], v[7]);
            dft4<DIR>(v[8], v[9], v[10], v[11]);
            dft4<DIR>(v[12], v[13], v[14], v[15]);
            Cx<R> t;
            t = v[1]; v[1] = v[4]; v[4] = t;
            t = v[2]; v[2] = v[8]; v[8] = t;
            t = v[3]; v[3] = v[12]; v[12] = t;
            t = v[6]; v[6] = v[9]; v[9] = t;
            t = v[7]; v[7] = v[13]; v[13] = t;
            t = v[11]; v[11] = v[14]; v[14] = t;
        }
    }
    // The mirror image (decimation in frequency): same 16-point transform, the stage twiddle W^(p k) of OUTPUT
    // p = p1 + 4*p2 applied on the way out, again in split form: W^(p1 k) = bt[p1-1] between the two radix-4
    // layers, W^(4 p2 k) = ot[p2-1] on the outputs.  With the conjugated direction this is exactly the
    // transpose of run_tw<true> preceded by its pre-multiplication, i.e. what undoes a forward stage.
    static __device__ __forceinline__ void run_tw_post(Cx<R> (&v)[16], Cx<R> ot1, Cx<R> ot2, Cx<R> ot3,
                                                       Cx<R> bt1, Cx<R> bt2, Cx<R> bt3) {
        // layer 1: DFT4 over r1 (r = 4 r1 + r2) for each r2 -> u[p1][r2] at v[4 p1 + r2]
        dft4<DIR>(v[0], v[4], v[8], v[12]);
        dft4<DIR>(v[1], v[5], v[9], v[13]);
        dft4<DIR>(v[2], v[6], v[10], v[14]);
        dft4<DIR>(v[3], v[7], v[11], v[15]);
        // u[p1][r2] *= w16^(r2 p1)
        v[5] = rot16<1, DIR>(v[5]);   v[6] = rot16<2, DIR>(v[6]);   v[7] = rot16<3, DIR>(v[7]);
        v[9] = rot16<2, DIR>(v[9]);   v[10] = rot16<4, DIR>(v[10]); v[11] = rot16<6, DIR>(v[11]);
        v[13] = rot16<3, DIR>(v[13]); v[14] = rot16<6, DIR>(v[14]); v[15] = rot16<9, DIR>(v[15]);
        // ... *= W^(p1 k)
        cmul4<DIR>(v[4], v[5], v[6], v[7], bt1);
        cmul4<DIR>(v[8], v[9], v[10], v[11], bt2);
        cmul4<DIR>(v[12], v[13], v[14], v[15], bt3);
        // layer 2: DFT4 over r2 for each p1 -> V[p1 + 4 p2] at v[4 p1 + p2]
        dft4<DIR>(v[0], v[1], v[2], v[3]);
        dft4<DIR>(v[4], v[5], v[6], v[7]);
        dft4<DIR>(v[8], v[9], v[10], v[11]);
        dft4<DIR>(v[12], v[13], v[14], v[15]);
        // ... *= W^(4 p2 k)
        cmul4<DIR>(v[1], v[5], v[9], v[13], ot1);
        cmul4<DIR>(v[2], v[6], v[10], v[14], ot2);
        cmul4<DIR>(v[3], v[7], v[11], v[15], ot3);
        // natural order
        Cx<R> t;
        t = v[1]; v[1] = v[4]; v[4] = t;
        t = v[2]; v[2] = v[8]; v[8] = t;
        t = v[3]; v[3] = v[12]; v[12] = t;
        t = v[6]; v[6] = v[9]; v[9] = t;
        t = v[7]; v[7] = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
};

// ---- radix schedules ---------------------------------------------------------------------------
template <int N> struct Sched;
template <> struct Sched<64>   { static constexpr int S = 2; static constexpr int r[4] = {16, 4, 1, 1}; };
template <> struct Sched<128>  { static constexpr int S = 2; static constexpr int r[4] = {16, 8, 1, 1}; };
template <> struct Sched<256>  { static constexpr int S = 2; static constexpr int r[4] = {16, 16, 1, 1}; };
template <> struct Sched<512>  { static constexpr int S = 3; static constexpr int r[4] = {16, 16, 2, 1}; };
template <> struct Sched<1024> { static constexpr int S = 3; static constexpr int r[4] = {16, 16, 4, 1}; };
template <> struct Sched<2048> { static constexpr int S = 3; static constexpr int r[4] = {16, 16, 8, 1}; };
template <> struct Sched<4096> { static constexpr int S = 3; static constexpr int r[4] = {16, 16, 16, 1}; };
template <> struct Sched<8192> { static constexpr int S = 4; static constexpr int r[4] = {16, 16, 16, 2}; };
template <> struct Sched<16384> { static constexpr int S = 4; static constexpr int r[4] = {16, 16, 16, 4}; };   // 1024 lanes (general path)

template <int N, int STAGE> constexpr int sched_ns() {  // product of radices before STAGE
    int ns = 1;
    for (int s = 0; s < STAGE; ++s) ns *= Sched<N>::r[s];
    return ns;
}
constexpr int tw_regs(int radix) {  // twiddle registers a stage of this radix keeps per lane
    return radix == 16 ? 6 : (16 / radix) * (radix - 1);
}
template <int N, int STAGE> constexpr int tw_offset() {  // first twiddle register of STAGE
    int off = 0;
    for (int s = 1; s < STAGE; ++s) off += tw_regs(Sched<N>::r[s]);
    return off;
}
template <int N> constexpr int tw_count() { return tw_offset<N, Sched<N>::S>(); }

__device__ __forceinline__ int lds_pad(int q) { return q + (q >> 4); }
// (8192: two interleaved 4096-point images of 16 x 272 elements, 16 elements apart from that -- WgFftL8k)
template <int N> constexpr int lds_elems() { return N == 8192 ? 2 * (16 * 272 + 16) : N + N / 16; }

// ---- the workgroup FFT -----------------------------------------------------------------------------
// Usage: WgFft<R,N> f; f.init(table, j);  ... f.template run<DIR>(v, lds, j);
// `lds` points at this transform's private region of lds_elems<N>() complex elements.
// All T = N/16 lanes of the transform (and every other lane of the workgroup) must call run():
// it contains __syncthreads().
// RESIDENT = true : stage twiddles are fetched once (init) and stay in VGPRs -- for kernels that run
//                   many transforms per workgroup (the column kernels).
// RESIDENT = false: they are fetched from the (L1/L2-resident) table right before each use, which
//                   frees ~24 VGPRs -- for kernels that are occupancy-bound (the row kernels).
#ifndef HGS_WAVE_LOCAL_BARRIER
#define HGS_WAVE_LOCAL_BARRIER 1
#endif
template <typename R, int N, bool RESIDENT = true> struct WgFft {
    static constexpr int E = 16;
    static constexpr int T = N / 16;
    // the rendezvous between the write and the read side of an exchange.  Up to 1024 points a transform is owned by ONE wave (or
    // part of one: T <= 64 lanes, its own LDS image), so nothing crosses waves and a workgroup barrier only couples the
    // independent transforms of the workgroup to each other (round 5: eight s_barrier per fused pass of the small grids --
    // the reference's own 512^2 / 1024^2 cases -- gone; LDS operations of one wave execute in issue order)
    static __device__ __forceinline__ void xbar() {
        if constexpr (T <= 64 && HGS_WAVE_LOCAL_BARRIER) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    }
    static constexpr int NTW = RESIDENT ? (tw_count<N>() > 0 ? tw_count<N>() : 1) : 1;
    Cx<R> tw[NTW];
    const Cx<R>* table_ = nullptr;

    // table[i] = exp(-2*pi*i*i/N), i < N  (forward sign)
    __device__ __forceinline__ void init(const Cx<R>* __restrict__ table, int j) {
        table_ = table;
        if constexpr (RESIDENT) load_stage_twiddles_all(table, j);
    }

    // twiddle q of stage s for butterfly b (same indexing as the resident array)
    template <int s, int IDX> __device__ __forceinline__ Cx<R> twv(int j) const {
        if constexpr (RESIDENT) {
            return tw[tw_offset<N, s>() + IDX];
        } else {
            constexpr int RAD = Sched<N>::r[s];
            constexpr int NS = sched_ns<N, s>();
            constexpr int STEP = N / (NS * RAD);
            if constexpr (RAD == 16) {
                constexpr int q = IDX < 3 ? 4 * (IDX + 1) : (IDX - 3 + 1);
                return table_[(q * (j % NS) * STEP) & (N - 1)];
            } else {
                constexpr int B = E / RAD;
                constexpr int b = IDX / (RAD - 1), r = IDX % (RAD - 1) + 1;
                if constexpr (RAD == 2 && NS * 2 == N && B == 8) {
                    // closing radix-2 stage: W_N^(j + b T) = W_N^j * W_16^b -- one table entry and a constant rotation
                    // instead of eight entries in flight
                    return rot16<b, -1>(table_[j]);
                }
                return table_[(r * ((j + b * T) % NS) * STEP) & (N - 1)];
            }
        }
    }

    __device__ __forceinline__ void load_stage_twiddles_all(const Cx<R>* __restrict__ table, int j) {
        static_for<1, Sched<N>::S>([&](auto s_) {
            constexpr int s = s_;
            constexpr int RAD = Sched<N>::r[s];
            constexpr int B = E / RAD;
            constexpr int NS = sched_ns<N, s>();
            constexpr int OFF = tw_offset<N, s>();
            constexpr int STEP = N / (NS * RAD);
            if constexpr (RAD == 16) {
                // split form: tw[OFF+q-1] = W^(4 q k) (pre-twiddles), tw[OFF+3+q-1] = W^(q k), q = 1..3
                const int k = j % NS;
                static_for<1, 4>([&](auto q_) {
                    constexpr int q = q_;
                    tw[OFF + q - 1] = table[(4 * q * k * STEP) & (N - 1)];
                    tw[OFF + 3 + q - 1] = table[(q * k * STEP) & (N - 1)];
                });
            } else {
                static_for<0, B>([&](auto b_) {
                    constexpr int b = b_;
                    const int k = (j + b * T) % NS;
                    static_for<1, RAD>([&](auto r_) {
                        constexpr int r = r_;
                        tw[OFF + b * (RAD - 1) + (r - 1)] = table[(r * k * STEP) & (N - 1)];
                    });
                });
            }
        });
    }

    // LDS position of logical element q of the current exchange, relative to a padded base:
    // pad(base + d) == pad(base) + d + d/16 whenever d % 16 == 0 or (base % 16 == 0 and d < 16),
    // which lets every ds_write/ds_read of a stage use one address VGPR + immediate offsets.
    // PING-PONG (DB = true): consecutive exchanges alternate between two LDS images, which makes the
    // barrier after the gather unnecessary (a lane can only overwrite image A again after it passed
    // the barrier of the exchange on image B, which every lane reaches after finishing its reads of
    // A).  `par` is the running exchange parity of this workgroup; LDS need is 2 * lds_elems<N>().
    int par = 0;
    template <int DIR, int s, bool DB = false, int NOUT = 16> __device__ __forceinline__ void stage(Cx<R> (&v)[16], Cx<R>* lds0, int j) {
        Cx<R>* lds = lds0;
        if constexpr (DB && s != Sched<N>::S - 1) {
            lds = lds0 + (par ? lds_elems<N>() : 0);
            par ^= 1;
        }
        constexpr int RAD = Sched<N>::r[s];
        constexpr int B = E / RAD;
        constexpr int NS = sched_ns<N, s>();
        constexpr int OFF = tw_offset<N, s>();
        static_for<0, B>([&](auto b_) {
            constexpr int b = b_;
            Cx<R> u[RAD];
            static_for<0, RAD>([&](auto r_) { constexpr int r = r_; u[r] = v[b + r * B]; });
            if constexpr (RAD == 16 && s > 0) {
                static_for<1, 4>([&](auto r2_) {
                    constexpr int r2 = r2_;
                    cmul4<DIR>(u[4 * r2], u[4 * r2 + 1], u[4 * r2 + 2], u[4 * r2 + 3], this->template twv<s, r2 - 1>(j));
                });
                // (the last stage leaves output p of the butterfly in register p: a caller that keeps only the first
                //  NOUT registers prunes the butterfly's last layer)
                if (!HGS_ABL_BFLY)
                    Dft<16, DIR, R>::template run_tw<true, (s == Sched<N>::S - 1 ? NOUT : 16)>(
                        u, this->template twv<s, 3>(j), this->template twv<s, 4>(j), this->template twv<s, 5>(j));
            } else {
                if constexpr (s > 0) {
                    static_for<1, RAD>([&](auto r_) {
                        constexpr int r = r_;
                        const Cx<R> w = this->template twv<s, b * (RAD - 1) + (r - 1)>(j);
                        u[r] = DIR < 0 ? cmul(u[r], w) : cmulc(u[r], w);
                    });
                }
                if (!HGS_ABL_BFLY) Dft<RAD, DIR, R>::run(u);
            }
            if constexpr (HGS_ABL_XCHG && s != Sched<N>::S - 1) {
                static_for<0, RAD>([&](auto r_) { constexpr int r = r_; v[b + r * B] = u[r]; });
            } else
            if constexpr (s == Sched<N>::S - 1) {
                static_for<0, RAD>([&](auto r_) { constexpr int r = r_; v[b + r * B] = u[r]; });
            } else {
                const int jv = j + b * T;
                const int k = jv % NS;
                const int base = (jv / NS) * (NS * RAD) + k;
                if constexpr (NS % 16 == 0 || (NS == 1 && RAD == 16)) {
                    Cx<R>* p = lds + lds_pad(base);
                    static_for<0, RAD>([&](auto r_) {
                        constexpr int r = r_;
                        p[r * NS + (r * NS) / 16] = u[r];
                    });
                } else {
                    static_for<0, RAD>([&](auto r_) {
                        constexpr int r = r_;
                        lds[lds_pad(base + r * NS)] = u[r];
                    });
                }
            }
        });
        if constexpr (s != Sched<N>::S - 1 && !HGS_ABL_XCHG) {
            xbar();
            if constexpr (T % 16 == 0) {
                const Cx<R>* p = lds + lds_pad(j);
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = p[m * (T + T / 16)]; });
            } else {
                static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = lds[lds_pad(j + m * T)]; });
            }
            if constexpr (!DB) xbar();
        }
    }

    template <int DIR, bool DB = false, int NOUT = 16> __device__ __forceinline__ void run(Cx<R> (&v)[16], Cx<R>* lds, int j) {
        static_for<0, Sched<N>::S>([&](auto s_) {
            constexpr int s = s_;
            this->template stage<DIR, s, DB, NOUT>(v, lds, j);
        });
    }
    // uniform entry points (see WgFftL for why the inverse comes in two flavours)
    __device__ __forceinline__ void fwd(Cx<R> (&v)[16], Cx<R>* lds, int j) { run<-1>(v, lds, j); }
    // (registers NZ.. of the input are zero: only WgFftL prunes its first stage for it)
    template <int NZ> __device__ __forceinline__ void fwd_lead(Cx<R> (&v)[16], Cx<R>* lds, int j) { run<-1>(v, lds, j); }
    __device__ __forceinline__ void inv(Cx<R> (&v)[16], Cx<R>* lds, int j) { run<+1>(v, lds, j); }
    __device__ __forceinline__ void inv_after_fwd(Cx<R> (&v)[16], Cx<R>* lds, int j) { run<+1>(v, lds, j); }
    // only registers 0 .. NOUT-1 of the result are kept (pruned where the last stage is a radix-16 one)
    template <int NOUT> __device__ __forceinline__ void inv_after_fwd_trail(Cx<R> (&v)[16], Cx<R>* lds, int j) {
        run<+1, false, (Sched<N>::r[Sched<N>::S - 1] == 16 ? NOUT : 16)>(v, lds, j);
    }
    template <int NOUT> __device__ __forceinline__ void inv_trail(Cx<R> (&v)[16], Cx<R>* lds, int j) { inv_after_fwd_trail<NOUT>(v, lds, j); }
};

// ---- the 4096-point transform with a row-local first exchange ----------------------------------------
// 256 lanes x 16 registers, three radix-16 stages.  Element n = n0 + 16 n1 + 256 n2 of the SPACE side (rows
// of GH, SLM columns) is held by lane p = 16 n0 + n1 in register n2, i.e. lane p owns the elements
// space_lane(p) + 256 m with space_lane(p) = (p >> 4) + 16 (p & 15); element k of the FREQUENCY side by lane
// k % 256 in register k / 256, as in WgFft.  With the two hex digits of the lane index swapped on the space
// side, the exchange between stage 0 and stage 1 is a 16 x 16 transpose inside each row of 16 lanes -- one
// wave, no barrier, conflict-free at stride 17 -- and only the exchange between stage 1 and stage 2 crosses
// waves.  The inverse is the exact mirror of the forward transform (decimation in frequency, post-twiddles,
// Dft<16>::run_tw_post), so it maps the frequency layout back to the space layout and shares the forward
// twiddle registers (conjugated).  tools/fft_local_model.py is the index model of both directions.
//
// LDS: one image of 16 row regions of 272 elements (the same 34,816 bytes as WgFft).  Hazards: forward writes
// (local and cross-wave) touch only the writing wave's own regions, its cross-wave gather reads every region
// and is bracketed by two barriers; the inverse scatters into every region (one barrier, then reads its own
// region).  An inverse therefore needs every wave to be done with the previous transform's LDS reads: true
// after a forward transform (it ends with a barrier) or at kernel start; after another inverse pass
// LEAD = true (one more barrier).  Fused forward -> inverse passes cost 3 barriers instead of 8.
__device__ __forceinline__ void wave_lds_order() {
    // LDS operations of one wave execute in issue order; only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// TS: the twiddle table holds W_(4096 TS)^i (the 8192-point transform runs two of these on the W_8192 table)
// RAWBAR: workgroup barriers as "s_waitcnt lgkmcnt(0); s_barrier" instead of __syncthreads().  For kernels that keep
// LDS-DMA loads (global_load_lds) in flight across the transform: with such a load outstanding hipcc drains vmcnt(0)
// ahead of every __syncthreads(), i.e. the prefetch would be waited for at the first exchange.
// ES: element stride of the LDS image (8192 points: the two images interleaved element by element, ES = 2).
// LTW (with RESIDENT = false: the float64 column kernels, which have no registers to keep the stage twiddles in): the twiddles
// come from two small tables in LDS instead of per-use global loads -- A[m] = W_4096^(16 m), m < 256, and B[n] = W_4096^n,
// n < 192: a stage-1 twiddle W_256^(q k) is A[(q k) & 255]; a stage-2 twiddle W_4096^(q p), p = 16 p_hi + p_lo, is
// A[(q p_hi) & 255] * B[q p_lo] (one complex product, 1.5 ulp).  7 KB per workgroup; filled by ltw_fill at kernel start.
// What it buys is not the fetch itself (the table is L2-resident) but the in-order vmcnt queue: with twiddle loads inside
// the transform, anything requested AHEAD of it -- the column's weights and targets -- is waited for at the first twiddle
// (round 5, HGS_F64_WT_EARLY: slower), so they were requested after it and cost a full memory round trip per column
// (tools/microbench/trace8k f64main: 11.6 k of 45 k cycles).
constexpr int LTW_A = 256, LTW_B = 192, LTW_N = LTW_A + LTW_B;
template <typename R, bool RESIDENT = true, int TS = 1, bool RAWBAR = false, int ES = 1, bool LTW = false> struct WgFftL {
    static __device__ __forceinline__ void wg_barrier() {
        if constexpr (RAWBAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    }
    static constexpr int N = 4096, T = 256, E = 16, ROW = 272;
    static constexpr int NTW = RESIDENT ? 12 : 1;
    Cx<R> tw[NTW];
    const Cx<R>* table_ = nullptr;
    const Cx<R>* ltw_ = nullptr;    // LTW: the two tables in LDS
    int tr_n = 0;       // HGS_TRACE event counter (dead otherwise)

    static __host__ __device__ __forceinline__ int space_lane(int p) { return (p >> 4) + 16 * (p & 15); }
    // all lanes of the workgroup call; the caller puts a barrier between this and the first transform
    static __device__ __forceinline__ void ltw_fill(const Cx<R>* __restrict__ table, Cx<R>* ltw, int tid, int nthreads) {
        for (int i = tid; i < LTW_N; i += nthreads)
            ltw[i] = i < LTW_A ? table[TS * ((16 * i) & (N - 1))] : table[TS * (i - LTW_A)];
    }
    __device__ __forceinline__ void set_ltw(const Cx<R>* ltw) { ltw_ = ltw; }

    // twiddle IDX of stage s (1 or 2): IDX 0..2 = W^(4 q k), 3..5 = W^(q k), q = IDX % 3 + 1;
    // stage 1: W_256, k = p & 15; stage 2: W_4096, k = p
    template <int s, int IDX> __device__ __forceinline__ Cx<R> twv(int p) const {
        if constexpr (RESIDENT) {
            return tw[(s - 1) * 6 + IDX];
        } else if constexpr (LTW) {
            constexpr int q = (IDX < 3 ? 4 : 1) * (IDX % 3 + 1);
            if constexpr (s == 1) return ltw_[(q * (p & 15)) & (LTW_A - 1)];
            else return cmul(ltw_[(q * ((p >> 4) & 15)) & (LTW_A - 1)], ltw_[LTW_A + q * (p & 15)]);
        } else {
            constexpr int q = (IDX < 3 ? 4 : 1) * (IDX % 3 + 1);
            return s == 1 ? table_[TS * ((q * (p & 15) * 16) & (N - 1))] : table_[TS * ((q * p) & (N - 1))];
        }
    }
    __device__ __forceinline__ void init(const Cx<R>* __restrict__ table, int p) {
        table_ = table;
        if (HGS_PRIO == 1) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));     // wave slot within the SIMD
            if (hw & 1u) __builtin_amdgcn_s_setprio(1);
        }
        if constexpr (RESIDENT) {
            static_for<0, 6>([&](auto i_) {
                constexpr int i = i_;
                constexpr int q = (i < 3 ? 4 : 1) * (i % 3 + 1);
                tw[i] = table[TS * ((q * (p & 15) * 16) & (N - 1))];
                tw[6 + i] = table[TS * ((q * p) & (N - 1))];
            });
        }
    }

    template <int DIR, int s> __device__ __forceinline__ void butterfly_pre(Cx<R> (&v)[16], int p) {
        if (HGS_ABL_BFLY) return;
        if (HGS_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        static_for<1, 4>([&](auto r2_) {
            constexpr int r2 = r2_;
            cmul4<DIR>(v[4 * r2], v[4 * r2 + 1], v[4 * r2 + 2], v[4 * r2 + 3], this->template twv<s, r2 - 1>(p));
        });
        Dft<16, DIR, R>::template run_tw<true>(v, this->template twv<s, 3>(p), this->template twv<s, 4>(p),
                                               this->template twv<s, 5>(p));
        if (HGS_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    }
    template <int DIR, int s> __device__ __forceinline__ void butterfly_post(Cx<R> (&v)[16], int p) {
        if (HGS_ABL_BFLY) return;
        Dft<16, DIR, R>::run_tw_post(v, this->template twv<s, 0>(p), this->template twv<s, 1>(p), this->template twv<s, 2>(p),
                                     this->template twv<s, 3>(p), this->template twv<s, 4>(p), this->template twv<s, 5>(p));
    }

    // forward (DIR = -1 with the table as stored; DIR = +1 gives the conjugate transform in the same flow):
    // space layout in, frequency layout out
    template <int DIR, int NZ = 16> __device__ __forceinline__ void forward_flow(Cx<R> (&v)[16], Cx<R>* lds, int p) {
        Cx<R>* rowb = lds + ES * ROW * (p >> 4);
        HGS_T(tr_n, 10);
        if (!HGS_ABL_BFLY) Dft<16, DIR, R>::template run_lead<NZ>(v);
        HGS_T(tr_n, 11);
        if (!HGS_ABL_XCHG) {   // 16 x 16 transpose inside the row of 16 lanes
            Cx<R>* w = rowb + ES * 17 * (p & 15);
            static_for<0, 16>([&](auto i_) { constexpr int i = i_; w[ES * i] = v[i]; });
            wave_lds_order();
            const Cx<R>* r = rowb + ES * (p & 15);
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = r[ES * 17 * m]; });
        }
        HGS_T(tr_n, 12);
        butterfly_pre<DIR, 1>(v, p);
        HGS_T(tr_n, 13);
        if (!HGS_ABL_XCHG) {   // cross-wave exchange: lane (row n0, k_a) register k_b -> lane k_a + 16 k_b register n0
            Cx<R>* w = rowb + ES * (p & 15);
            static_for<0, 16>([&](auto r_) { constexpr int r = r_; w[ES * 16 * r] = v[r]; });
            wg_barrier();
            const Cx<R>* g = lds + ES * p;
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = g[ES * ROW * m]; });
            wg_barrier();
        }
        HGS_T(tr_n, 14);
        butterfly_pre<DIR, 2>(v, p);
        HGS_T(tr_n, 15);
    }
    // the mirror: frequency layout in, space layout out
    template <int DIR, bool LEAD, int NOUT = 16> __device__ __forceinline__ void mirror_flow(Cx<R> (&v)[16], Cx<R>* lds, int p) {
        Cx<R>* rowb = lds + ES * ROW * (p >> 4);
        HGS_T(tr_n, 20);
        butterfly_post<DIR, 2>(v, p);
        HGS_T(tr_n, 21);
        if (!HGS_ABL_XCHG) {
            if constexpr (LEAD) wg_barrier();
            Cx<R>* g = lds + ES * p;
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; g[ES * ROW * m] = v[m]; });
            wg_barrier();
            const Cx<R>* r = rowb + ES * (p & 15);
            static_for<0, 16>([&](auto r_) { constexpr int rr = r_; v[rr] = r[ES * 16 * rr]; });
        }
        HGS_T(tr_n, 22);
        butterfly_post<DIR, 1>(v, p);
        HGS_T(tr_n, 23);
        if (!HGS_ABL_XCHG) {
            wave_lds_order();
            Cx<R>* w = rowb + ES * (p & 15);
            static_for<0, 16>([&](auto m_) { constexpr int m = m_; w[ES * 17 * m] = v[m]; });
            wave_lds_order();
            const Cx<R>* r = rowb + ES * 17 * (p & 15);
            static_for<0, 16>([&](auto i_) { constexpr int i = i_; v[i] = r[ES * i]; });
        }
        HGS_T(tr_n, 24);
        if (!HGS_ABL_BFLY) Dft<16, DIR, R>::template run_trail<NOUT>(v);
        HGS_T(tr_n, 25);
    }

    // DIR = -1: space -> frequency (forward DFT); DIR = +1: frequency -> space (inverse DFT, unnormalised).
    // LEAD: see the hazard note above (only meaningful for DIR = +1).
    template <int DIR, bool LEAD = true> __device__ __forceinline__ void run(Cx<R> (&v)[16], Cx<R>* lds, int p) {
        if constexpr (DIR < 0) forward_flow<-1>(v, lds, p);
        else mirror_flow<+1, LEAD>(v, lds, p);
    }
    __device__ __forceinline__ void fwd(Cx<R> (&v)[16], Cx<R>* lds, int p) { forward_flow<-1>(v, lds, p); }
    // registers NZ.. of the input are zero
    template <int NZ> __device__ __forceinline__ void fwd_lead(Cx<R> (&v)[16], Cx<R>* lds, int p) { forward_flow<-1, NZ>(v, lds, p); }
    __device__ __forceinline__ void inv(Cx<R> (&v)[16], Cx<R>* lds, int p) { mirror_flow<+1, true>(v, lds, p); }
    // the previous LDS user of every wave was this workgroup's forward transform (or nothing)
    __device__ __forceinline__ void inv_after_fwd(Cx<R> (&v)[16], Cx<R>* lds, int p) { mirror_flow<+1, false>(v, lds, p); }
    // only registers 0 .. NOUT-1 of the result are kept
    template <int NOUT> __device__ __forceinline__ void inv_after_fwd_trail(Cx<R> (&v)[16], Cx<R>* lds, int p) {
        mirror_flow<+1, false, NOUT>(v, lds, p);
    }
    template <int NOUT> __device__ __forceinline__ void inv_trail(Cx<R> (&v)[16], Cx<R>* lds, int p) { mirror_flow<+1, true, NOUT>(v, lds, p); }
};

// ---- the 8192-point transform: a radix-2 step in registers + two interleaved WgFftL ----------------------
// 512 lanes x 16 registers.  Lane j = 2 p + h runs, as lane p, the 4096-point transform number h of the split
//     y0[n] = x[n] + x[n + 4096],   y1[n] = (x[n] - x[n + 4096]) W_8192^n,   X[2 k + h] = FFT_4096(y_h)[k]
// on its own LDS image (lds + h IMG): lane j register m holds X[j + 512 m] on the frequency side -- the standard
// layout -- and x[space_lane(j) + 512 r], space_lane(j) = WgFftL::space_lane(j >> 1) + 256 (j & 1), on the space
// side, so the pairs (r, r + 8) of the radix-2 step sit in one lane.  After that step ONE wave-local exchange (X1,
// between the lanes 2p and 2p + 1, no barrier) hands lane (p, h) the sixteen values y_h[. + 256 n2] WgFftL wants:
// every lane writes its eight sums (block 0) and eight differences (block 1) and reads block h of itself and of its
// neighbour.  X1 lives in the wave's own row regions of the two images (wave W: regions 2W, 2W + 1; block c in image
// c, element c + 64 i + (j & 63)); with the images 16 elements (32 banks) apart every access of X1 and of the two
// interleaved WgFftL flows is conflict-free, and a fused forward -> inverse pass needs 3 barriers (the general
// four-stage code: 12, and 24 % of its LDS cycles were bank conflicts).  The inverse is the mirror.
// tools/fft_local8k_model.py is the index model.  Zero / unwanted slots: NZ leading non-zero registers (NZ <= 8: the
// partner x[n + 4096] is zero, the radix-2 step is a copy and a twiddle) become 2 NZ leading slots of the 4096-point
// transforms; likewise NOUT on the way back.
template <typename R, bool RESIDENT = true, bool RAWBAR = false, bool LTW = false> struct WgFftL8k {
#ifndef HGS_8K_INTERLEAVE
#define HGS_8K_INTERLEAVE 1
#endif
    // images interleaved element by element (element e of image h at 2 e + h): consecutive lanes (p, 0), (p, 1) touch
    // consecutive elements, so a 16-lane group of a ds_write_b64 (served on 32 banks) covers 16 different bank pairs.
    // (Images 16 elements apart instead, HGS_8K_INTERLEAVE = 0: both halves of 8 lanes on the same 8 pairs, every cross-lane
    //  write 2-way conflicted, SQ_LDS_BANK_CONFLICT 40 % of the array cycles.)
    static constexpr bool IL = HGS_8K_INTERLEAVE != 0;
    static constexpr int N = 8192, T = 512, IMG = IL ? 1 : 16 * 272 + 16, X1 = IL ? 513 : IMG + 1, WREG = IL ? 1088 : 544;
    using Core = WgFftL<R, RESIDENT, 2, RAWBAR, IL ? 2 : 1, LTW>;
    Core core;
    static __device__ __forceinline__ void ltw_fill(const Cx<R>* __restrict__ table, Cx<R>* ltw, int tid, int nthreads) { Core::ltw_fill(table, ltw, tid, nthreads); }
    __device__ __forceinline__ void set_ltw(const Cx<R>* ltw) { core.set_ltw(ltw); }
    Cx<R> w2;            // W_8192^(space_lane(j))
    int tr_n = 0;

    static __host__ __device__ __forceinline__ int space_lane(int j) { return Core::space_lane(j >> 1) + 256 * (j & 1); }
    __device__ __forceinline__ void init(const Cx<R>* __restrict__ table, int j) {
        core.init(table, j >> 1);
        w2 = table[space_lane(j)];
    }
    // element offsets of X1 for lane j: its own slots (one per register and block) / block h of the pair (p, 0), (p, 1)
    static __device__ __forceinline__ int x1_own(int j) { return WREG * (j >> 6) + (j & 63); }
    static __device__ __forceinline__ int x1_pair(int j) { return (j & 1) * X1 + WREG * (j >> 6) + (j & 62); }

    template <int NZ> __device__ __forceinline__ void forward(Cx<R> (&v)[16], Cx<R>* lds, int j) {
        static_assert(NZ >= 2 && NZ <= 16, "WgFftL8k: leading non-zero registers");
        constexpr bool HALF = NZ <= 8;                 // registers 8.. are zero: sums = v[i], differences = v[i] * twiddle
        constexpr int NI = HALF ? NZ : 8;              // pairs that carry data
        if (!HGS_ABL_BFLY) {
            static_for<0, NI>([&](auto i_) {
                constexpr int i = i_;
                if constexpr (HALF) {
                    v[i + 8] = cmul(rot16<i, -1>(v[i]), w2);
                } else {
                    const Cx<R> d = v[i] - v[i + 8];
                    v[i] = v[i] + v[i + 8];
                    v[i + 8] = cmul(rot16<i, -1>(d), w2);
                }
            });
        }
        Cx<R> u[16];
        if (!HGS_ABL_XCHG) {
            Cx<R>* w = lds + x1_own(j);
            static_for<0, NI>([&](auto i_) { constexpr int i = i_; w[64 * i] = v[i]; w[X1 + 64 * i] = v[i + 8]; });
            wave_lds_order();
            const Cx<R>* r = lds + x1_pair(j);
            static_for<0, 16>([&](auto n_) {
                constexpr int n2 = n_;
                if constexpr ((n2 >> 1) < NI) u[n2] = r[64 * (n2 >> 1) + (n2 & 1)]; else u[n2] = mk<R>(0, 0);
            });
            wave_lds_order();                          // the core's first (wave-local) exchange reuses these regions
        } else {
            static_for<0, 16>([&](auto n_) { constexpr int n2 = n_; u[n2] = v[n2]; });
        }
        HGS_T(tr_n, 16);                               // radix-2 step + pair exchange done
        core.tr_n = tr_n;
        core.template forward_flow<-1, (2 * NI < 16 ? (2 * NI < 4 ? 4 : 2 * NI) : 16)>(u, lds + (j & 1) * IMG, j >> 1);
        tr_n = core.tr_n;
        static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = u[m]; });
    }
    template <bool LEAD, int NOUT> __device__ __forceinline__ void mirror(Cx<R> (&v)[16], Cx<R>* lds, int j) {
        static_assert(NOUT >= 2 && NOUT <= 16, "WgFftL8k: leading wanted registers");
        constexpr bool HALF = NOUT <= 8;               // registers 8.. of the result are not wanted
        constexpr int NI = HALF ? NOUT : 8;
        core.tr_n = tr_n;
        core.template mirror_flow<+1, LEAD, (2 * NI < 16 ? 2 * NI : 16)>(v, lds + (j & 1) * IMG, j >> 1);
        tr_n = core.tr_n;
        Cx<R> y0[8], y1[8];
        if (!HGS_ABL_XCHG) {
            wave_lds_order();                          // the core's last exchange read these regions
            Cx<R>* w = lds + x1_pair(j);
            static_for<0, 2 * NI>([&](auto n_) { constexpr int n2 = n_; w[64 * (n2 >> 1) + (n2 & 1)] = v[n2]; });
            wave_lds_order();
            const Cx<R>* r = lds + x1_own(j);
            static_for<0, NI>([&](auto i_) { constexpr int i = i_; y0[i] = r[64 * i]; y1[i] = r[X1 + 64 * i]; });
        } else {
            static_for<0, NI>([&](auto i_) { constexpr int i = i_; y0[i] = v[i]; y1[i] = v[i + 8]; });
        }
        if (!HGS_ABL_BFLY) {
            static_for<0, NI>([&](auto i_) {
                constexpr int i = i_;
                const Cx<R> t = cmulc(rot16<i, +1>(y1[i]), w2);
                v[i] = y0[i] + t;
                if constexpr (!HALF) v[i + 8] = y0[i] - t;
            });
        }
        HGS_T(tr_n, 26);                               // pair exchange + radix-2 step done
    }
    __device__ __forceinline__ void fwd(Cx<R> (&v)[16], Cx<R>* lds, int j) { forward<16>(v, lds, j); }
    template <int NZ> __device__ __forceinline__ void fwd_lead(Cx<R> (&v)[16], Cx<R>* lds, int j) { forward<NZ>(v, lds, j); }
    __device__ __forceinline__ void inv(Cx<R> (&v)[16], Cx<R>* lds, int j) { mirror<true, 16>(v, lds, j); }
    __device__ __forceinline__ void inv_after_fwd(Cx<R> (&v)[16], Cx<R>* lds, int j) { mirror<false, 16>(v, lds, j); }
    template <int NOUT> __device__ __forceinline__ void inv_after_fwd_trail(Cx<R> (&v)[16], Cx<R>* lds, int j) {
        mirror<false, NOUT>(v, lds, j);
    }
    template <int NOUT> __device__ __forceinline__ void inv_trail(Cx<R> (&v)[16], Cx<R>* lds, int j) { mirror<true, NOUT>(v, lds, j); }
};

// Which workgroup transform a kernel uses for length N, and where lane j's elements sit on the space side
// (frequency side: always j + m * N/16).
#ifndef HGS_LOCAL_FFT
#define HGS_LOCAL_FFT 1
#endif
template <typename R, int N, bool RESIDENT, bool RAWBAR = false, bool LTW = false> struct FftSel {
    using type = WgFft<R, N, RESIDENT>;
    static constexpr bool local = false;
    static __host__ __device__ __forceinline__ int space_lane(int j) { return j; }
};
#if HGS_LOCAL_FFT
template <typename R, bool RESIDENT, bool RAWBAR, bool LTW> struct FftSel<R, 4096, RESIDENT, RAWBAR, LTW> {
    using type = WgFftL<R, RESIDENT, 1, RAWBAR, 1, LTW>;
    static constexpr bool local = true;
    static __host__ __device__ __forceinline__ int space_lane(int j) { return WgFftL<R, RESIDENT>::space_lane(j); }
};
#ifndef HGS_LOCAL_FFT8K
#define HGS_LOCAL_FFT8K 1
#endif
#if HGS_LOCAL_FFT8K
template <typename R, bool RESIDENT, bool RAWBAR, bool LTW> struct FftSel<R, 8192, RESIDENT, RAWBAR, LTW> {
    using type = WgFftL8k<R, RESIDENT, RAWBAR, LTW>;
    static constexpr bool local = true;
    static __host__ __device__ __forceinline__ int space_lane(int j) { return WgFftL8k<R, RESIDENT>::space_lane(j); }
};
#endif
#endif

}  // namespace hgs
