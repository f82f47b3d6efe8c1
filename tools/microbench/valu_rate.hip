// Microbenchmark: issue rate of scalar v_fma_f32 vs packed v_pk_fma_f32 on gfx950 as a function of
// waves per SIMD and ILP.  Settles whether a wave64 FP32 VALU op costs 2 or 4 SIMD cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int ILP> __global__ void k_scalar(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP> __global__ void k_packed(float* out, int iters, float a, float b) {
    v2f x[ILP];
    const v2f av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = (v2f){threadIdx.x * 0.001f + i, 1.0f * i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    v2f s = {0, 0};
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <typename K> double run(K k, int blocks, int threads, int iters, float* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 1024 * sizeof(float));
    const int iters = 4000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double clk = 2.4e9;   // nominal; ratios are what matters
    printf("CUs %d\n", p.multiProcessorCount);
    for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: blocks of 256 threads (4 waves) per CU
        const int blocks = p.multiProcessorCount * wps;
        double ms1 = run(k_scalar<8>, blocks, 256, iters, d);
        double ms2 = run(k_packed<8>, blocks, 256, iters, d);
        double ms3 = run(k_scalar<2>, blocks, 256, iters, d);
        double ms4 = run(k_packed<2>, blocks, 256, iters, d);
        const double n8 = (double)iters * 8 * 8, n2 = (double)iters * 8 * 2;   // instrs per wave
        printf("waves/SIMD %d: scalar ILP8 %.2f cyc/instr/SIMD  packed ILP8 %.2f | scalar ILP2 %.2f packed ILP2 %.2f  (TF scalar %.1f packed %.1f)\n",
               wps, ms1 * 1e-3 * clk / (n8 * wps), ms2 * 1e-3 * clk / (n8 * wps), ms3 * 1e-3 * clk / (n2 * wps), ms4 * 1e-3 * clk / (n2 * wps),
               2.0 * n8 * 64 * 4 * blocks / (ms1 * 1e-3) / 1e12, 4.0 * n8 * 64 * 4 * blocks / (ms2 * 1e-3) / 1e12);
    }
    return 0;
}
