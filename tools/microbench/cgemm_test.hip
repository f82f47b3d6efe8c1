#include "../../slmsuite_amd/csrc/cgemm.hpp"
#include <cstdio>
#include <vector>
#include <complex>
#include <cmath>
using namespace hgs;
int main() {
    const int M = 300, N = 200, K = 100, split = 2, k_per = 64;
    std::vector<float2> A((size_t)K * M), B((size_t)K * N);
    for (size_t i = 0; i < A.size(); ++i) A[i] = make_float2((float)((i * 7919) % 101) / 101.f - 0.5f, (float)((i * 104729) % 97) / 97.f - 0.3f);
    for (size_t i = 0; i < B.size(); ++i) B[i] = make_float2((float)((i * 31337) % 89) / 89.f - 0.4f, (float)((i * 7) % 83) / 83.f - 0.6f);
    float2 *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, (size_t)split * M * N * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    CgemmArgs a{dA, dB, dC, M, N, K, M, N, split, k_per, 0, 0};
    hipFuncSetAttribute((const void*)cgemm_kouter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES);
    hipFuncSetAttribute((const void*)cgemm_kouter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES);
    hipFuncSetAttribute((const void*)cgemm_kouter<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES);
    hipFuncSetAttribute((const void*)cgemm_kouter<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES);
    for (int gauss = 0; gauss < 2; ++gauss) {
    if (gauss) hipLaunchKernelGGL((cgemm_kouter<false, true>), dim3((M + 127) / 128, (N + 127) / 128, split), dim3(256), CG_LDS_BYTES, 0, a);
    else hipLaunchKernelGGL(cgemm_kouter<false>, dim3((M + 127) / 128, (N + 127) / 128, split), dim3(256), CG_LDS_BYTES, 0, a);
    printf("launch (gauss %d): %s\n", gauss, hipGetErrorString(hipDeviceSynchronize()));
    std::vector<float2> C((size_t)split * M * N);
    hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    double err = 0, nrm = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        std::complex<double> ref = 0;
        for (int k = 0; k < K; ++k) ref += std::complex<double>(A[(size_t)k * M + m].x, A[(size_t)k * M + m].y) * std::complex<double>(B[(size_t)k * N + n].x, B[(size_t)k * N + n].y);
        std::complex<double> got = 0;
        for (int s = 0; s < split; ++s) got += std::complex<double>(C[((size_t)s * M + m) * N + n].x, C[((size_t)s * M + m) * N + n].y);
        err += std::norm(got - ref); nrm += std::norm(ref);
    }
    printf("rel err %.3e\n", std::sqrt(err / nrm));
    }
    // throughput at the cfg4 shapes
    const int M2 = 10000, N2 = 1152, K2 = 1920, M2p = 10112;
    float2 *A2, *B2, *C2;
    hipMalloc(&A2, (size_t)K2 * M2p * 8); hipMalloc(&B2, (size_t)K2 * N2 * 8); hipMalloc(&C2, (size_t)2 * M2 * N2 * 8);
    hipMemset(A2, 0, (size_t)K2 * M2p * 8); hipMemset(B2, 0, (size_t)K2 * N2 * 8);
    CgemmArgs g{A2, B2, C2, M2, N2, K2, M2p, N2, 2, 960, 0, 0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (rep == 0) hipLaunchKernelGGL(cgemm_kouter<false>, dim3((M2 + 127) / 128, (N2 + 127) / 128, 2), dim3(256), CG_LDS_BYTES, 0, g);
        else if (rep == 2) hipLaunchKernelGGL((cgemm_kouter<true, true>), dim3((M2 + 127) / 128, (N2 + 127) / 128, 2), dim3(256), CG_LDS_BYTES, 0, g);
        else hipLaunchKernelGGL(cgemm_kouter<true>, dim3((M2 + 127) / 128, (N2 + 127) / 128, 2), dim3(256), CG_LDS_BYTES, 0, g);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("n2f-shaped GEMM %d x %d x %d split 2 %s: %.3f ms  %.1f TFLOP/s\n", M2, N2, K2, rep == 2 ? "padded+gauss" : rep ? "padded" : "checked", ms, 8.0 * M2 * N2 * K2 / ms / 1e9);
    }
    const int M3 = 1152, N3 = 1920, K3 = 10000, sp = 7, kp = 1440;
    float2 *A3, *B3, *C3;
    hipMalloc(&A3, (size_t)sp * kp * M3 * 8); hipMalloc(&B3, (size_t)sp * kp * N3 * 8); hipMalloc(&C3, (size_t)sp * M3 * N3 * 8);
    hipMemset(A3, 0, (size_t)sp * kp * M3 * 8); hipMemset(B3, 0, (size_t)sp * kp * N3 * 8);
    CgemmArgs h{A3, B3, C3, M3, N3, K3, M3, N3, sp, kp, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (rep == 0) hipLaunchKernelGGL(cgemm_kouter<false>, dim3((M3 + 127) / 128, (N3 + 127) / 128, sp), dim3(256), CG_LDS_BYTES, 0, h);
        else if (rep == 2) hipLaunchKernelGGL((cgemm_kouter<true, true>), dim3((M3 + 127) / 128, (N3 + 127) / 128, sp), dim3(256), CG_LDS_BYTES, 0, h);
        else hipLaunchKernelGGL(cgemm_kouter<true>, dim3((M3 + 127) / 128, (N3 + 127) / 128, sp), dim3(256), CG_LDS_BYTES, 0, h);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("f2n-shaped GEMM %d x %d x %d split %d %s: %.3f ms  %.1f TFLOP/s\n", M3, N3, K3, sp, rep == 2 ? "padded+gauss" : rep ? "padded" : "checked", ms, 8.0 * M3 * N3 * K3 / ms / 1e9);
    }
    return 0;
}
