// Does a long straight-line instruction stream run as fast as a short loop?  Same packed-FMA work (16 independent
// accumulator pairs), emitted as a loop body of BODY instructions executed ITER times: BODY * ITER constant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int BODY> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2f a[16];
    for (int i = 0; i < 16; ++i) a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f};
    const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        // BODY packed FMAs, 16 independent chains; distinct constants keep the compiler from folding the chain
        constexpr int G = BODY / 16, A = G > 32 ? 32 : G, B = G / A;
        sfor<0, A>([&](auto) {
            sfor<0, B>([&](auto) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
                asm volatile("" ::: "memory");
            });
        });
    }
    v2f s = {0, 0};
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s.x == 123.f) out[0] = s.y;
}
template <int BODY> void run(float* sink, int wgs) {
    const int total = 1 << 19;             // packed FMAs per lane
    const int iters = total / BODY;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<BODY>, dim3(wgs), dim3(256), 0, 0, sink, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<BODY>, dim3(wgs), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("body %5d instructions (%3d KB), %4d workgroups: %.0f us, %.1f TFLOP/s\n", BODY, BODY * 8 / 1024, wgs, ms * 1e3,
           (double)wgs * 256 * total * 4 / (ms * 1e-3) / 1e12);
}
int main() {
    float* sink; hipMalloc(&sink, 64);
    for (int wgs : {512, 768, 1024}) {
        run<64>(sink, wgs); run<1024>(sink, wgs); run<4096>(sink, wgs); run<8192>(sink, wgs); run<16384>(sink, wgs);
    }
    return 0;
}
