// Where does a one-row-per-workgroup transform kernel spend its time?  1152 workgroups of 256 lanes (the cfg 2
// row kernel's shape): (a) empty, (b) twiddle fetch only, (c) + one 4096-point transform on register data,
// (d) + two, (e) two transforms with a 34.8 KB dynamic LDS allocation but no twiddle fetch from memory.
#include "../../slmsuite_amd/csrc/fft_core.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace hgs;
constexpr int N = 4096;
template <int MODE> __global__ __launch_bounds__(256, 3) void k(const v2f* tw, v2f* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* lds = reinterpret_cast<v2f*>(smem);
    const int j = threadIdx.x;
    if constexpr (MODE == 0) { if (j == 1000) out[0] = tw[0]; return; }
    WgFft<float, N> f;
    f.init(tw, j);
    v2f v[16];
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; v[m] = (v2f){(float)(j + m), (float)(j - m)}; });
    if constexpr (MODE >= 2) f.template run<-1>(v, lds, j);
    if constexpr (MODE >= 3) f.template run<+1>(v, lds, j);
    v2f acc = f.tw[0];
    static_for<0, 16>([&](auto m_) { constexpr int m = m_; acc += v[m]; });
    static_for<1, 36>([&](auto q_) { constexpr int q = q_; acc += f.tw[q]; });
    if (acc.x == 12345.f) out[blockIdx.x] = acc;
}
template <typename F> float timeit(F f, int reps = 50) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize(); hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
    v2f *tw, *out;
    hipMalloc(&tw, N * 8); hipMalloc(&out, 1 << 20);
    std::vector<v2f> h(N);
    for (int i = 0; i < N; ++i) h[i] = (v2f){(float)cos(-2 * M_PI * i / N), (float)sin(-2 * M_PI * i / N)};
    hipMemcpy(tw, h.data(), N * 8, hipMemcpyHostToDevice);
    const size_t lds = lds_elems<N>() * 8;
    for (int wgs : {1152, 768, 256}) {
        printf("%4d workgroups: empty %.1f us | twiddles %.1f | +1 transform %.1f | +2 transforms %.1f\n", wgs,
               timeit([&] { hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), lds, 0, tw, out); }),
               timeit([&] { hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), lds, 0, tw, out); }),
               timeit([&] { hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), lds, 0, tw, out); }),
               timeit([&] { hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), lds, 0, tw, out); }));
    }
    return 0;
}
