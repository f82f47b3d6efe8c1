// EXPERIMENT: two 4096-point transforms interleaved inside ONE wave stream (two register sets, two LDS images), so
// that the LDS exchange of one column is issued between the butterfly instructions of the other, against the product
// structure (one transform per workgroup pass, two workgroups per CU).  Register-resident data, no global traffic:
// time per forward + inverse pair.
#include "../../slmsuite_amd/csrc/fft_core.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace hgs;

#ifndef DS_PER
#define DS_PER 1      // DS instructions per interleave group
#endif
#ifndef VALU_PER
#define VALU_PER 5    // VALU instructions per interleave group
#endif
// interleave pattern for a region holding N_DS LDS instructions of kind MASK and VALU work
template <int MASK, int N_DS> __device__ __forceinline__ void mix() {
#if DUAL_SCHED
    static_for<0, N_DS / DS_PER>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(MASK, DS_PER, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);
    });
#endif
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

struct Dual : WgFftL<float, true> {
    using B = WgFftL<float, true>;
    static constexpr int ROW = 272;
    __device__ __forceinline__ void lw(v2f (&v)[16], v2f* rowb, int p) { v2f* w = rowb + 17 * (p & 15); static_for<0, 16>([&](auto i_) { constexpr int i = i_; w[i] = v[i]; }); }
    __device__ __forceinline__ void lr(v2f (&v)[16], v2f* rowb, int p) { const v2f* r = rowb + (p & 15); static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = r[17 * m]; }); }
    __device__ __forceinline__ void xw(v2f (&v)[16], v2f* rowb, int p) { v2f* w = rowb + (p & 15); static_for<0, 16>([&](auto r_) { constexpr int r = r_; w[16 * r] = v[r]; }); }
    __device__ __forceinline__ void xr(v2f (&v)[16], v2f* lds, int p) { const v2f* g = lds + p; static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = g[ROW * m]; }); }
    // mirror-side exchanges
    __device__ __forceinline__ void gw(v2f (&v)[16], v2f* lds, int p) { v2f* g = lds + p; static_for<0, 16>([&](auto m_) { constexpr int m = m_; g[ROW * m] = v[m]; }); }
    __device__ __forceinline__ void gr(v2f (&v)[16], v2f* rowb, int p) { const v2f* r = rowb + (p & 15); static_for<0, 16>([&](auto r_) { constexpr int rr = r_; v[rr] = r[16 * rr]; }); }
    __device__ __forceinline__ void tw_(v2f (&v)[16], v2f* rowb, int p) { v2f* w = rowb + (p & 15); static_for<0, 16>([&](auto m_) { constexpr int m = m_; w[17 * m] = v[m]; }); }
    __device__ __forceinline__ void tr_(v2f (&v)[16], v2f* rowb, int p) { const v2f* r = rowb + 17 * (p & 15); static_for<0, 16>([&](auto i_) { constexpr int i = i_; v[i] = r[i]; }); }

    __device__ __forceinline__ void fwd2(v2f (&a)[16], v2f (&b)[16], v2f* la, v2f* lb, int p) {
        v2f* ra = la + ROW * (p >> 4); v2f* rb = lb + ROW * (p >> 4);
        Dft<16, -1, float>::run(a);
        FENCE();
        lw(a, ra, p); Dft<16, -1, float>::run(b); mix<0x200, 16>();
        FENCE();
        wave_lds_order();
        lr(a, ra, p); lw(b, rb, p);
        FENCE();
        wave_lds_order();
        this->template butterfly_pre<-1, 1>(a, p); lr(b, rb, p); mix<0x100, 16>();
        FENCE();
        xw(a, ra, p); this->template butterfly_pre<-1, 1>(b, p); mix<0x200, 16>();
        FENCE();
        xw(b, rb, p);
        __syncthreads();
        xr(a, la, p); xr(b, lb, p);
        FENCE();
        this->template butterfly_pre<-1, 2>(a, p);
        this->template butterfly_pre<-1, 2>(b, p);
        __syncthreads();
    }
    // both previous LDS users of every wave were forward transforms of this workgroup (they end with a barrier)
    __device__ __forceinline__ void inv2(v2f (&a)[16], v2f (&b)[16], v2f* la, v2f* lb, int p) {
        v2f* ra = la + ROW * (p >> 4); v2f* rb = lb + ROW * (p >> 4);
        this->template butterfly_post<+1, 2>(a, p);
        FENCE();
        gw(a, la, p); this->template butterfly_post<+1, 2>(b, p); mix<0x200, 16>();
        FENCE();
        gw(b, lb, p);
        __syncthreads();
        gr(a, ra, p); gr(b, rb, p);
        FENCE();
        this->template butterfly_post<+1, 1>(a, p);
        FENCE();
        wave_lds_order();
        tw_(a, ra, p); this->template butterfly_post<+1, 1>(b, p); mix<0x200, 16>();
        FENCE();
        wave_lds_order();
        tr_(a, ra, p); tw_(b, rb, p);
        FENCE();
        wave_lds_order();
        Dft<16, +1, float>::run(a); tr_(b, rb, p); mix<0x100, 16>();
        FENCE();
        Dft<16, +1, float>::run(b);
        __syncthreads();     // (LEAD for the next inverse; the product flow gets it from the forward transform in between)
    }
};

template <int OCC> __global__ __launch_bounds__(256, OCC) void single_k(const v2f* tw, v2f* io, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int p = threadIdx.x;
    WgFftL<float, true> f; f.init(tw, p);
    v2f v[16];
    const int ps = WgFftL<float, true>::space_lane(p);
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = io[(size_t)blockIdx.x * 4096 + ps + 256 * m]; });
    for (int it = 0; it < iters; ++it) {
        f.fwd(v, lds, p);
        static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = v[m] * (1.f / 4096.f); });
        f.inv_after_fwd(v, lds, p);
    }
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; io[(size_t)blockIdx.x * 4096 + ps + 256 * m] = v[m]; });
}
template <int OCC> __global__ __launch_bounds__(256, OCC) void dual_k(const v2f* tw, v2f* io, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* la = reinterpret_cast<v2f*>(smem);
    v2f* lb = la + lds_elems<4096>();
    const int p = threadIdx.x;
    Dual f; f.init(tw, p);
    v2f a[16], b[16];
    const int ps = WgFftL<float, true>::space_lane(p);
    v2f* ia = io + (size_t)(2 * blockIdx.x) * 4096; v2f* ib = ia + 4096;
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; a[m] = ia[ps + 256 * m]; b[m] = ib[ps + 256 * m]; });
    for (int it = 0; it < iters; ++it) {
        f.fwd2(a, b, la, lb, p);
        static_for<0, 16>([&](auto m_) { constexpr int m = m_; a[m] = a[m] * (1.f / 4096.f); b[m] = b[m] * (1.f / 4096.f); });
        f.inv2(a, b, la, lb, p);
    }
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; ia[ps + 256 * m] = a[m]; ib[ps + 256 * m] = b[m]; });
}
template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    const int NL = 2048;                       // lines
    v2f *tw, *io;
    hipMalloc(&tw, 4096 * 8); hipMalloc(&io, (size_t)NL * 4096 * 8);
    std::vector<v2f> htw(4096), h((size_t)NL * 4096), r((size_t)NL * 4096);
    for (int i = 0; i < 4096; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / 4096), (float)sin(-2 * M_PI * i / 4096)};
    unsigned s = 7;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x.x = (float)(s >> 8) / 16777216.f - 0.5f; s = s * 1664525u + 1013904223u; x.y = (float)(s >> 8) / 16777216.f - 0.5f; }
    hipMemcpy(tw, htw.data(), 4096 * 8, hipMemcpyHostToDevice);
    const size_t l1 = lds_elems<4096>() * 8, l2 = 2 * l1;
    hipFuncSetAttribute(reinterpret_cast<const void*>(dual_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(dual_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    // correctness: one forward + inverse pair (scaled) must return the input
    auto check = [&](const char* name) {
        hipMemcpy(r.data(), io, r.size() * 8, hipMemcpyDeviceToHost);
        double n = 0, d = 0;
        for (size_t i = 0; i < h.size(); ++i) { double ex = r[i].x - h[i].x, ey = r[i].y - h[i].y; n += ex * ex + ey * ey; d += (double)h[i].x * h[i].x + (double)h[i].y * h[i].y; }
        printf("%s: round trip rel L2 %.3e\n", name, sqrt(n / d));
    };
    hipMemcpy(io, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(single_k<2>, dim3(NL), dim3(256), l1, 0, tw, io, 1); hipDeviceSynchronize(); check("single");
    hipMemcpy(io, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dual_k<2>, dim3(NL / 2), dim3(256), l2, 0, tw, io, 1); hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(hipGetLastError())); check("dual");
    const int IT = 200;
    // same number of transform pairs per CU in every variant: 8 lines per CU
    const double pairs = 2048.0 * IT;
    float t;
    t = timeit([&] { hipLaunchKernelGGL(single_k<2>, dim3(512), dim3(256), l1, 0, tw, io, 4 * IT); });
    printf("single stream, 2 workgroups per CU : %.1f us  -> %.2f ns per transform pair (chip)\n", t, t * 1e3 / pairs);
    t = timeit([&] { hipLaunchKernelGGL(single_k<3>, dim3(768), dim3(256), l1, 0, tw, io, 4 * IT * 2 / 3); });
    printf("single stream, 3 workgroups per CU : %.1f us  -> %.2f ns\n", t, t * 1e3 / (768.0 * (4 * IT * 2 / 3)));
    t = timeit([&] { hipLaunchKernelGGL(single_k<1>, dim3(256), dim3(256), l1, 0, tw, io, 8 * IT); });
    printf("single stream, 1 workgroup per CU  : %.1f us  -> %.2f ns\n", t, t * 1e3 / pairs);
    t = timeit([&] { hipLaunchKernelGGL(dual_k<1>, dim3(256), dim3(256), l2, 0, tw, io, 4 * IT); });
    printf("dual stream, 1 workgroup per CU    : %.1f us  -> %.2f ns\n", t, t * 1e3 / pairs);
    t = timeit([&] { hipLaunchKernelGGL(dual_k<2>, dim3(512), dim3(256), l2, 0, tw, io, 2 * IT); });
    printf("dual stream, 2 workgroups per CU   : %.1f us  -> %.2f ns\n", t, t * 1e3 / pairs);
    return 0;
}
