#include "../../slmsuite_amd/csrc/cgemm.hpp"
#include <cstdio>
#include <vector>
#include <complex>
#include <cmath>
#include <algorithm>
using namespace hgs;

// ---- stream-K form: correctness on a ragged shape (few and many workgroups), then the cfg 4 shapes ----
// dA / dB: planar operands (real array, then the imaginary one planeA / planeB floats on)
static int run_sk(const float* dA, size_t planeA, const float* dB, size_t planeB, float2* dC, int M, int N, int K, int lda, int ldb, int G,
                  int* planes_out, std::vector<int>* nseg_out, float* ms_out) {
    const int tm = (M + 127) / 128, tn = (N + 127) / 128, KT = (K + 15) / 16, tiles = tm * tn;
    const long long total = (long long)tiles * KT;
    if (G > total) G = (int)total;            // every workgroup owns at least one step: the owners of a tile are consecutive
    std::vector<int> first(tiles), nseg(tiles);
    int planes = 1;
    for (int t = 0; t < tiles; ++t) {
        first[t] = sk_owner((long long)t * KT, total, G);
        nseg[t] = sk_owner((long long)(t + 1) * KT - 1, total, G) - first[t] + 1;
        planes = std::max(planes, nseg[t]);
    }
    int* dF;
    hipMalloc(&dF, tiles * 4);
    hipMemcpy(dF, first.data(), tiles * 4, hipMemcpyHostToDevice);
    CgemmSkArgs a{dA, dA + planeA, dB, dB + planeB, dC, M, N, KT, lda, ldb, tm, tn, planes, dF, 0, 0, nullptr, 0, nullptr, 0};
    hipFuncSetAttribute((const void*)cgemm_streamk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CG_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(cgemm_streamk<0>, dim3(G), dim3(256), CG_LDS_BYTES, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    *planes_out = planes;
    if (nseg_out) *nseg_out = nseg;
    hipFree(dF);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
static void test_streamk() {
    const int M = 300, N = 200, K = 100, Mp = 384, Np = 256, Kp = 112;
    std::vector<float2> A((size_t)Kp * Mp, make_float2(0, 0)), B((size_t)Kp * Np, make_float2(0, 0));
    for (int k = 0; k < K; ++k) {
        for (int m = 0; m < M; ++m) { size_t i = (size_t)k * M + m; A[(size_t)k * Mp + m] = make_float2((float)((i * 7919) % 101) / 101.f - 0.5f, (float)((i * 104729) % 97) / 97.f - 0.3f); }
        for (int n = 0; n < N; ++n) { size_t i = (size_t)k * N + n; B[(size_t)k * Np + n] = make_float2((float)((i * 31337) % 89) / 89.f - 0.4f, (float)((i * 7) % 83) / 83.f - 0.6f); }
    }
    float *dA, *dB; float2* dC;
    std::vector<float> Ap(2 * A.size()), Bp(2 * B.size());
    for (size_t i = 0; i < A.size(); ++i) { Ap[i] = A[i].x; Ap[A.size() + i] = A[i].y; }
    for (size_t i = 0; i < B.size(); ++i) { Bp[i] = B[i].x; Bp[B.size() + i] = B[i].y; }
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, (size_t)8 * M * N * 8);
    hipMemcpy(dA, Ap.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, Bp.data(), B.size() * 8, hipMemcpyHostToDevice);
    for (int G : {1, 5, 7, 512}) {
        int planes; std::vector<int> nseg; float ms;
        hipMemset(dC, 0xff, (size_t)8 * M * N * 8);       // NaN pattern: planes a tile does not own must not be read
        run_sk(dA, A.size(), dB, B.size(), dC, M, N, K, Mp, Np, G, &planes, &nseg, &ms);
        std::vector<float2> C((size_t)planes * M * N);
        hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
        double err = 0, nrm = 0;
        const int tm = (M + 127) / 128;
        for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
            std::complex<double> ref = 0, got = 0;
            for (int k = 0; k < K; ++k) ref += std::complex<double>(A[(size_t)k * Mp + m].x, A[(size_t)k * Mp + m].y) * std::complex<double>(B[(size_t)k * Np + n].x, B[(size_t)k * Np + n].y);
            const int ns = nseg[(m >> 7) + (n >> 7) * tm];
            for (int s = 0; s < ns; ++s) got += std::complex<double>(C[((size_t)s * M + m) * N + n].x, C[((size_t)s * M + m) * N + n].y);
            err += std::norm(got - ref); nrm += std::norm(ref);
        }
        printf("stream-K G = %3d planes %d: rel err %.3e\n", G, planes, std::sqrt(err / nrm));
    }
    hipFree(dA); hipFree(dB); hipFree(dC);
    struct Shape { int M, N, K; const char* name; } shapes[] = {{10000, 1152, 1920, "n2f"}, {1152, 1920, 10000, "f2n"}};
    for (auto& sh : shapes) {
        const int Mp2 = (sh.M + 127) / 128 * 128, Np2 = (sh.N + 127) / 128 * 128, Kp2 = (sh.K + 15) / 16 * 16;
        float *A2, *B2; float2* C2;
        hipMalloc(&A2, (size_t)Kp2 * Mp2 * 8); hipMalloc(&B2, (size_t)Kp2 * Np2 * 8); hipMalloc(&C2, (size_t)8 * sh.M * sh.N * 8);
        hipMemset(A2, 0, (size_t)Kp2 * Mp2 * 8); hipMemset(B2, 0, (size_t)Kp2 * Np2 * 8);
        int planes; float ms = 0, best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { run_sk(A2, (size_t)Kp2 * Mp2, B2, (size_t)Kp2 * Np2, C2, sh.M, sh.N, sh.K, Mp2, Np2, 512, &planes, nullptr, &ms); if (ms < best) best = ms; }
        printf("%s-shaped GEMM %d x %d x %d stream-K (512 workgroups, %d planes): %.3f ms  %.1f TFLOP/s\n", sh.name, sh.M, sh.N, sh.K, planes, best, 8.0 * sh.M * sh.N * sh.K / best / 1e9);
        hipFree(A2); hipFree(B2); hipFree(C2);
    }
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    test_streamk();
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
