// EXPERIMENT (round 2, not part of libhgs.so; see wave_fft.hpp).
// ROW kernels of the 4096-wide fp32 path on the one-wave transform of wave_fft.hpp: a wavefront owns one SLM row
// (64 lanes x 64 registers), a workgroup is a single wave.  Same modes, arguments, GH layout and arithmetic as
// row_kernel (kernels.hpp):
//   MODE 0 : phase -> G      MODE 1 : H -> phase (or the complex nearfield)      MODE 2 : H -> phasor -> G
// Element k = n0 + 64 n1 of the padded row lives in lane n0, register n1; the SLM columns c0 .. c0 + Sw - 1 fall
// into registers [LO, HI) (template parameters, whole radix-4 groups; the launcher picks the instantiation).
// MASKED: sparse targets (RowArgs::load_mask / store_mask are honoured; a separate instantiation because 64
// predicated accesses split the straight-line code into as many blocks).
// All global accesses are raw buffer instructions (one resource per array row / GH image): the lane part of an
// address is ONE VGPR for all 64 registers, the register part an SGPR, and the range check of the resource does
// the predication (columns outside the SLM read as zero, masked or out-of-range stores are dropped) without a branch.
// grid = (row blocks [+ 1 for the weight-norm block], batch), block = 64.
#pragma once
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include "wave_fft.hpp"

#ifndef ROW_WAVE_LOOP
#define ROW_WAVE_LOOP 0
#endif

namespace hgs {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
struct Buf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ Buf(const void* p, unsigned bytes)
        : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ v2f ld2(unsigned voff, unsigned soff) const {
        return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    }
    __device__ __forceinline__ float ld1(unsigned voff, unsigned soff) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    __device__ __forceinline__ void st2(v2f x, unsigned voff, unsigned soff) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, soff, 0);
    }
    __device__ __forceinline__ void st1(float x, unsigned voff, unsigned soff) const {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, voff, soff, 0);
    }
};
// 1 / sqrt(x) with the hardware instruction (1 ulp), pre-scaled where x is too small for it (v_rsq_f32 takes
// denormal inputs as zero).  Same values as rsqrtf(), whose inlined form costs the row kernel 58 spilled registers.
__device__ __forceinline__ float rsqrt_full(float x) {
    const bool tiny = x < 0x1p-100f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 0x1p+100f : x);
    return tiny ? r * 0x1p+50f : r;
}
constexpr unsigned BUF_OOB = 0xf0000000u;   // a lane offset past every resource: load gives 0, store is dropped

template <int MODE, int LO, int HI, bool MASKED>
__global__ __launch_bounds__(64, 2) void row_wave_kernel(RowArgs<float> a) {
    using M = Math<float>;
    constexpr int N = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* buf = reinterpret_cast<float*>(smem);
    const Geo g = a.g;
    const int lane = threadIdx.x;
    const int b = blockIdx.y;

    // deferred weight normalisation (see row_kernel): the last block owns no row
    if (a.wpartial != nullptr && blockIdx.x == gridDim.x - 1) {
        double s = 0;
        for (int i = lane; i < a.n_wpartial; i += 64) s += a.wpartial[(size_t)b * a.n_wpartial + i];
        s = wave_sum(s);
        if (lane == 0) a.wscale[b] = (float)(1.0 / ::sqrt(s));
    }

    WaveFft4096 fft;
    fft.init(a.tw, lane);

    const float sgn = (lane & 1) ? -1.f : 1.f;      // (-1)^(n0 + 64 n1)
    // GH element (r, k) sits at ((k>>2)*Sh + r)*4 + (k&3): lane part + register part (64 columns = 16 groups of four)
    const Buf gh(a.gh + (size_t)b * g.Sh * g.Pw, (unsigned)g.Sh * g.Pw * 8u);
    const unsigned gh_lane = ((unsigned)(lane >> 2) * g.Sh * 4u + (unsigned)(lane & 3)) * 8u;
    const unsigned gh_step = 16u * g.Sh * 4u * 8u;
    // SLM column of register n1 is lane - c0 + 64 n1; negative columns wrap to offsets past the row's resource.
    // (gfx9 range-checks the VGPR offset + immediate only, NOT the SGPR offset: the column goes into the former)
    const unsigned c_off = (unsigned)(lane - g.c0) * 4u;

    // sparse targets: the masks are kept per lane of the 256-lane layout (bit m of entry j = column j + 256 m);
    // column n0 + 64 n1 is bit n1 >> 2 of entry n0 + 64 (n1 & 3)
    unsigned long long lmask = ~0ull, smask = ~0ull;
    auto fetch_mask = [&](const unsigned short* tab) -> unsigned long long {
        unsigned long long mk64 = 0;
        static_for<0, 4>([&](auto q_) {
            constexpr int q = q_;
            const unsigned e = tab[(size_t)b * (N / 16) + lane + 64 * q];
            static_for<0, 16>([&](auto m_) {
                constexpr int m = m_;
                mk64 |= (unsigned long long)((e >> m) & 1u) << (4 * m + q);
            });
        });
        return mk64;
    };
    if (MASKED && a.load_mask != nullptr) lmask = fetch_mask(a.load_mask);
    if (MASKED && a.store_mask != nullptr) smask = (a.store_mask == a.load_mask) ? lmask : fetch_mask(a.store_mask);

    int r = blockIdx.x;
    if (a.xcd_map) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        r = 4 * ((idx >> 2) * 8 + xcd) + (idx & 3);
    }
    if (r >= g.Sh) return;

    v2f v[64];
    float aux[HI - LO];
    const size_t srow = (size_t)b * g.Sh * g.Sw + (size_t)r * g.Sw;
    const unsigned row_bytes = (unsigned)g.Sw * 4u;
    const Buf ph(a.phase + srow, row_bytes);
    const Buf kn(a.kern ? a.kern + (size_t)r * g.Sw : nullptr, a.kern ? row_bytes : 0u);
    const Buf am(a.amp ? a.amp + (size_t)r * g.Sw : nullptr, a.amp ? row_bytes : 0u);
    const unsigned gh_row = (unsigned)r * 32u;

    if constexpr (MODE != 0) {
        static_for<0, 64>([&](auto m_) {
            constexpr int m = m_;
            const unsigned vo = (!MASKED || ((lmask >> m) & 1ull)) ? gh_lane : BUF_OOB;
            v[m] = gh.ld2(vo, gh_row + (unsigned)m * gh_step) * sgn;
        });
        fft.first<+1>(v, buf, lane);
        // what the columns of the SLM need next, issued under the second half of the transform
        if constexpr (MODE == 2) {
            static_for<LO, HI>([&](auto m_) { constexpr int m = m_; aux[m - LO] = am.ld1(c_off + m * 256u, 0u); });
        } else if (a.nf_out == nullptr) {
            static_for<LO, HI>([&](auto m_) { constexpr int m = m_; aux[m - LO] = kn.ld1(c_off + m * 256u, 0u); });
        }
        __builtin_amdgcn_sched_barrier(0);
        fft.second<+1>(v);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) {
            const float sc = sgn * a.scale;
            if (a.nf_out != nullptr) {
                const Buf nf(a.nf_out + srow, row_bytes * 2u);
                static_for<LO, HI>([&](auto m_) { constexpr int m = m_; nf.st2(v[m] * sc, c_off * 2u + m * 512u, 0u); });
            } else {
                static_for<LO, HI>([&](auto m_) {
                    constexpr int m = m_;
                    float p = M::atan2(v[m].y * sc, v[m].x * sc);
                    p -= aux[m - LO];                        // (no kernel: empty resource, reads 0)
                    ph.st1(p, c_off + m * 256u, 0u);
                });
            }
        }
    }
    if constexpr (MODE != 1) {
        static_for<LO, HI>([&](auto m_) {
            constexpr int m = m_;
            // amplitude of the column, zero outside the SLM
            float amv = MODE == 2 ? aux[m - LO] : am.ld1(c_off + m * 256u, 0u);   // (no amplitude array: empty resource)
            if (a.amp == nullptr) amv = (c_off + m * 256u < row_bytes) ? a.amp_scalar : 0.f;
            v2f nf;
            if constexpr (MODE == 2) {
                // phase = atan2(nf) - kernel, rebuilt as exp(i (phase + kernel)) = nf / |nf| (see row_kernel)
                const float p2 = v[m].x * v[m].x + v[m].y * v[m].y;
                // (evaluated eagerly and selected: a conditional around it becomes a real branch per element,
                //  32 basic blocks with register copies at every join)
                const v2f on = v[m] * (amv * rsqrt_full(p2));
                nf.x = (p2 > 0.f) ? on.x : amv * sgn;
                nf.y = (p2 > 0.f) ? on.y : 0.f;
            } else {
                const float p = ph.ld1(c_off + m * 256u, 0u) + kn.ld1(c_off + m * 256u, 0u);
                float s, co;
                M::sincos(p, &s, &co);
                nf = (v2f){amv * sgn * co, amv * sgn * s};
            }
            v[m] = nf;
            if constexpr (m % 4 == 3) __builtin_amdgcn_sched_barrier(0);    // (bounds the temporaries in flight)
        });
        fft.run<-1, LO, HI>(v, buf, lane);
        const float sc = sgn * a.scale;
        // (the register parts of the addresses are recomputed, not kept in 64 SGPRs since the loads)
        unsigned gh_row_s = gh_row;
        asm volatile("" : "+s"(gh_row_s));
        static_for<0, 64>([&](auto m_) {
            constexpr int m = m_;
            const unsigned vo = (!MASKED || ((smask >> m) & 1ull)) ? gh_lane : BUF_OOB;
            gh.st2(v[m] * sc, vo, gh_row_s + (unsigned)m * gh_step);
        });
    }
}

}  // namespace hgs
