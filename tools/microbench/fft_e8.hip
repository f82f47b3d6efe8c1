// Prototype: 4096-point in-workgroup transform with 8 elements per lane (512 lanes, four radix-8
// stages) against the production 16 elements per lane (256 lanes, three radix-16 stages).
// Streams rows of a row-major array: load -> transform -> store.
#include "../../slmsuite_amd/csrc/fft_core.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace hgs;
constexpr int N = 4096;

__device__ __forceinline__ int pad8(int q) { return q + (q >> 3); }

struct Fft8 {
    static constexpr int T = 512;
    v2f tw[12];   // per twiddled stage: W^(2k), W^(4k), W^(6k), W^(k)
    __device__ __forceinline__ void init(const v2f* table, int j) {
        static_for<1, 4>([&](auto s_) {
            constexpr int s = s_;
            constexpr int NS = s == 1 ? 8 : (s == 2 ? 64 : 512);
            constexpr int STEP = N / (NS * 8);
            const int k = j % NS;
            tw[(s - 1) * 4 + 0] = table[(2 * k * STEP) & (N - 1)];
            tw[(s - 1) * 4 + 1] = table[(4 * k * STEP) & (N - 1)];
            tw[(s - 1) * 4 + 2] = table[(6 * k * STEP) & (N - 1)];
            tw[(s - 1) * 4 + 3] = table[(k * STEP) & (N - 1)];
        });
    }
    template <int s> __device__ __forceinline__ void stage(v2f (&v)[8], v2f* lds, int j) {
        constexpr int NS = s == 0 ? 1 : (s == 1 ? 8 : (s == 2 ? 64 : 512));
        if constexpr (s > 0) {
            // pre-twiddle v[r1 + 2 r2] by W^(2 r2 k)
            static_for<1, 4>([&](auto r2_) {
                constexpr int r2 = r2_;
                v[2 * r2] = cmul(v[2 * r2], tw[(s - 1) * 4 + r2 - 1]);
                v[2 * r2 + 1] = cmul(v[2 * r2 + 1], tw[(s - 1) * 4 + r2 - 1]);
            });
        }
        dft4<-1>(v[0], v[2], v[4], v[6]);
        dft4<-1>(v[1], v[3], v[5], v[7]);
        if constexpr (s > 0) {
            const v2f w = tw[(s - 1) * 4 + 3];
            v[1] = cmul(v[1], w); v[3] = cmul(v[3], w); v[5] = cmul(v[5], w); v[7] = cmul(v[7], w);
        }
        v2f t1 = rot16<2, -1>(v[3]), t2 = rot16<4, -1>(v[5]), t3 = rot16<6, -1>(v[7]);
        v2f e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[1] = e1 + t1; v[5] = e1 - t1;
        v[2] = e2 + t2; v[6] = e2 - t2;
        v[3] = e3 + t3; v[7] = e3 - t3;
        if constexpr (s < 3) {
            const int k = j % NS, base = (j / NS) * (NS * 8) + k;
            static_for<0, 8>([&](auto r_) { constexpr int r = r_; lds[pad8(base + r * NS)] = v[r]; });
            __syncthreads();
            static_for<0, 8>([&](auto m_) { constexpr int m = m_; v[m] = lds[pad8(j + m * T)]; });
            __syncthreads();
        }
    }
    __device__ __forceinline__ void run(v2f (&v)[8], v2f* lds, int j) {
        stage<0>(v, lds, j); stage<1>(v, lds, j); stage<2>(v, lds, j); stage<3>(v, lds, j);
    }
};

__global__ __launch_bounds__(512, 4) void k8(const v2f* in, v2f* out, const v2f* tw, int nrows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int j = threadIdx.x;
    Fft8 f; f.init(tw, j);
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        v2f v[8];
        const v2f* p = in + (size_t)row * N;
        static_for<0, 8>([&](auto m_) { constexpr int m = m_; v[m] = (p + m * 512)[(unsigned)j]; });
        f.run(v, lds, j);
        v2f* q = out + (size_t)row * N;
        static_for<0, 8>([&](auto m_) { constexpr int m = m_; (q + m * 512)[(unsigned)j] = v[m]; });
    }
}
template <int REP> __global__ __launch_bounds__(256, 4) void k16(const v2f* in, v2f* out, const v2f* tw, int nrows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int j = threadIdx.x;
    WgFft<float, N> f; f.init(tw, j);
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        v2f v[16];
        const v2f* p = in + (size_t)row * N;
        static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = (p + m * 256)[(unsigned)j]; });
#pragma unroll 1
        for (int r = 0; r < REP; ++r) f.template run<-1>(v, lds, j);
        v2f* q = out + (size_t)row * N;
        static_for<0, 16>([&](auto m_) { constexpr int m = m_; (q + m * 256)[(unsigned)j] = v[m]; });
    }
}
template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    const int nrows = 4096;
    v2f *in, *out, *out2, *tw;
    hipMalloc(&in, (size_t)nrows * N * 8); hipMalloc(&out, (size_t)nrows * N * 8); hipMalloc(&out2, (size_t)nrows * N * 8); hipMalloc(&tw, N * 8);
    std::vector<v2f> h((size_t)nrows * N), htw(N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (v2f){(float)((i * 2654435761u) % 1000) / 1000.f - 0.5f, (float)((i * 40503u) % 777) / 777.f - 0.5f};
    for (int i = 0; i < N; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / N), (float)sin(-2 * M_PI * i / N)};
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(tw, htw.data(), N * 8, hipMemcpyHostToDevice);
    const size_t lds16 = lds_elems<N>() * 8, lds8 = (N + N / 8) * 8;
    hipLaunchKernelGGL(k16<1>, dim3(1024), dim3(256), lds16, 0, in, out, tw, nrows);
    hipLaunchKernelGGL(k8, dim3(512), dim3(512), lds8, 0, in, out2, tw, nrows);
    std::vector<v2f> a(h.size()), b(h.size());
    hipMemcpy(a.data(), out, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out2, h.size() * 8, hipMemcpyDeviceToHost);
    double d = 0, n = 0;
    for (size_t i = 0; i < (size_t)N * 8; ++i) { d += (a[i].x - b[i].x) * (a[i].x - b[i].x) + (a[i].y - b[i].y) * (a[i].y - b[i].y); n += a[i].x * a[i].x + a[i].y * a[i].y; }
    printf("E8 vs E16 rel diff %.2e\n", sqrt(d / n));
    for (int g : {512, 768, 1024, 2048}) {
        printf("E16x4 grid %4d: %.1f us   ", g, timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(g), dim3(256), lds16, 0, in, out, tw, nrows); }));
        printf("E16 grid %4d: %.1f us   ", g, timeit([&] { hipLaunchKernelGGL(k16<1>, dim3(g), dim3(256), lds16, 0, in, out, tw, nrows); }));
        printf("E8 grid %4d: %.1f us\n", g / 2, timeit([&] { hipLaunchKernelGGL(k8, dim3(g / 2), dim3(512), lds8, 0, in, out2, tw, nrows); }));
    }
    printf("bytes moved %.0f MB\n", 2.0 * nrows * N * 8 / 1e6);
    return 0;
}
