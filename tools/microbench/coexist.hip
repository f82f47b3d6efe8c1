// Do VALU work and HBM streaming overlap on this chip, or do their times add up?  Two kernels on two streams:
// A = packed-fp32 FMA chains in registers (no memory), B = a streaming copy.  Times alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void valu_only(float* out, int iters) {
    v2f a[16];
    for (int i = 0; i < 16; ++i) a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.f};
    const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    v2f s = {0, 0};
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s.x == 123.f) out[0] = s.y;
}
__global__ __launch_bounds__(256) void stream_copy(const float4* in, float4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const size_t n = (size_t)64 << 20;   // 64 Mi float4 = 1 GiB each way
    float4 *in, *out; float* sink;
    hipMalloc(&in, n * 16); hipMalloc(&out, n * 16); hipMalloc(&sink, 64);
    hipMemset(in, 0, n * 16);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    auto runA = [&](int wg) { hipLaunchKernelGGL(valu_only, dim3(wg), dim3(256), 0, sa, sink, 20000); };
    auto runB = [&](int wg) { hipLaunchKernelGGL(stream_copy, dim3(wg), dim3(256), 0, sb, in, out, n); };
    for (int wgA : {256, 512}) for (int wgB : {256, 512, 1024}) {
        float ta, tb, ta2, tb2;
        runA(wgA); runB(wgB); hipDeviceSynchronize();
        hipEventRecord(a0, sa); runA(wgA); hipEventRecord(a1, sa); hipEventSynchronize(a1); hipEventElapsedTime(&ta, a0, a1);
        hipEventRecord(b0, sb); runB(wgB); hipEventRecord(b1, sb); hipEventSynchronize(b1); hipEventElapsedTime(&tb, b0, b1);
        hipEventRecord(a0, sa); hipEventRecord(b0, sb); runA(wgA); runB(wgB); hipEventRecord(a1, sa); hipEventRecord(b1, sb);
        hipEventSynchronize(a1); hipEventSynchronize(b1); hipEventElapsedTime(&ta2, a0, a1); hipEventElapsedTime(&tb2, b0, b1);
        printf("VALU %4d WG, copy %4d WG: alone %.0f us (%.1f TFLOP/s) and %.0f us (%.2f TB/s); together %.0f us and %.0f us\n", wgA, wgB,
               ta * 1e3, (double)wgA * 256 * 20000 * 16 * 4 / (ta * 1e-3) / 1e12, tb * 1e3, 2.0 * n * 16 / (tb * 1e-3) / 1e12, ta2 * 1e3, tb2 * 1e3);
    }
    return 0;
}
