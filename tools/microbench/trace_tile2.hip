// Timeline (s_memtime stamps at phase boundaries, lane 0 of every wave) and stand-alone launch time of the headline column
// kernel since round 5: the half-width tile kernel at 4096 rows, idle column parked in LDS, three workgroups per CU
// (col_tile2_kernel<float, 4096, 0, 5, 1, PARK>), on the cfg 2 target (32 x 32 spots, pitch 64).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DPARKED=0] [-DGRID=768]
//                [-DHGS_TRACE=1 -DHGS_TRACE_OFF=49152 -DHGS_TRACE_SKIP=n] trace_tile2.hip -o trace_tile2[_t]
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <map>
using namespace hgs;
#ifndef PARKED
#define PARKED 1
#endif
#ifndef GRID
#define GRID 768
#endif
typedef float Rt;
typedef Cx<Rt> Ct;

int main() {
    constexpr int N = 4096, T = N / 16;
    Geo g{N, N, 1152, 1920, (N - 1152) / 2, (N - 1920) / 2, 1, T};
    const size_t P = (size_t)N * N;
    Rt *w, *t, *wscale; Ct *gh, *tw; double *wp, *fp;
    hipMalloc(&w, P * sizeof(Rt)); hipMalloc(&t, P * sizeof(Rt));
    hipMalloc(&gh, (size_t)g.Sh * g.Pw * sizeof(Ct)); hipMalloc(&tw, N * sizeof(Ct)); hipMalloc(&wscale, sizeof(Rt));
    hipMalloc(&wp, 4096 * 8); hipMalloc(&fp, (size_t)2048 * 1024 * 8);
    std::vector<Ct> htw(N);
    for (int i = 0; i < N; ++i) htw[i] = (Ct){(Rt)cos(-2 * M_PI * i / N), (Rt)sin(-2 * M_PI * i / N)};
    hipMemcpy(tw, htw.data(), N * sizeof(Ct), hipMemcpyHostToDevice);
    std::vector<Rt> hw(P, (Rt)0), ht(P, (Rt)0);
    auto at = [&](int kx, int ky) -> size_t { return (size_t)kx * N + col_pos(ky, T); };
    for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) {
        const int kx = N / 2 - 16 * 64 + 32 + a * 64, ky = N / 2 - 16 * 64 + 32 + b * 64;
        hw[at(kx, ky)] = (Rt)0.03; ht[at(kx, ky)] = (Rt)0.03;
    }
    std::vector<Ct> hg((size_t)g.Sh * g.Pw);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = (Ct){(Rt)(((i * 2654435761u) % 1000) * 1e-6), (Rt)(((i * 40503u) % 1000) * 1e-6)};
    Rt one = 1;
    hipMemcpy(w, hw.data(), P * sizeof(Rt), hipMemcpyHostToDevice); hipMemcpy(t, ht.data(), P * sizeof(Rt), hipMemcpyHostToDevice);
    hipMemcpy(gh, hg.data(), hg.size() * sizeof(Ct), hipMemcpyHostToDevice); hipMemcpy(wscale, &one, sizeof(Rt), hipMemcpyHostToDevice);
    ColArgs<Rt> ca{};
    ca.g = g; ca.gh = gh; ca.w = w; ca.t = t; ca.wscale = wscale; ca.wpartial = wp; ca.fpartial = fp; ca.tw = tw;
    ca.scale = (Rt)(1.0 / sqrt((double)N));
    ca.cp.method = M_LEONARDO; ca.cp.do_update = 1; ca.cp.p_exp = (Rt)0.8; ca.cp.inv_fnorm = 1;
    const int shift = (g.r0 / 16) * 16;
    const int grid = GRID, block = 256;
    auto k = col_tile2_kernel<float, 4096, 0, 5, 1, PARKED != 0>;
    size_t lds = col_tile2_lds_bytes<float, 4096, PARKED != 0>();
#if HGS_TRACE
    if (lds > (size_t)HGS_TRACE_OFF) { printf("HGS_TRACE_OFF too small: kernel needs %zu bytes\n", lds); return 1; }
    lds = HGS_TRACE_OFF + 4 * 128 * 8;
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { printf("hipFuncSetAttribute(%zu): %s\n", lds, hipGetErrorString(e)); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, ca, shift, grid % 16 == 0 ? 1 : 0); };
    for (int i = 0; i < 3; ++i) { hipMemcpy(w, hw.data(), P * sizeof(Rt), hipMemcpyHostToDevice); launch(); }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    const int reps = 50;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("col_tile2_kernel<float, 4096, 0, 5, 1, %s>%s: %.1f us per launch (grid %d x %d lanes, %zu B LDS)\n", PARKED ? "true" : "false",
           HGS_TRACE ? " [traced build]" : "", ms / reps * 1e3f, grid, block, lds);
#if HGS_TRACE
    const int waves = block / 64, nev = waves * 128;
    std::vector<unsigned long long> tr((size_t)grid * nev);
    hipMemcpy(tr.data(), fp, tr.size() * 8, hipMemcpyDeviceToHost);
    std::map<std::pair<int, int>, std::pair<double, long>> acc;
    const unsigned long long M56 = 0xffffffffffffffull;
    double span = 0; long nspan = 0;
    for (int wg = 0; wg < grid; ++wg) for (int wv = 0; wv < waves; ++wv) {
        const unsigned long long* ev = &tr[(size_t)wg * nev + wv * 128];
        int n = 0; while (n < 128 && (ev[n] >> 56) != 0) ++n;
        for (int i = 1; i < n; ++i) {
            auto& a = acc[{(int)(ev[i - 1] >> 56), (int)(ev[i] >> 56)}];
            a.first += (double)((ev[i] & M56) - (ev[i - 1] & M56)); a.second++;
        }
        if (n > 1) { span += (double)((ev[n - 1] & M56) - (ev[0] & M56)); nspan++; }
    }
    const char* names[40] = {};
    names[1] = "half tile start"; names[2] = "half tile landed"; names[3] = "fwd done"; names[4] = "w/t landed";
    names[5] = "constraint(+issue) done"; names[6] = "inv done"; names[7] = "end";
    names[10] = "fwd: enter core"; names[11] = "fwd: s0 done"; names[12] = "fwd: local xchg done"; names[13] = "fwd: s1 done";
    names[14] = "fwd: global xchg done"; names[15] = "fwd: s2 done";
    names[20] = "inv: enter core"; names[21] = "inv: s2 done"; names[22] = "inv: global xchg done"; names[23] = "inv: s1 done";
    names[24] = "inv: local xchg done"; names[25] = "inv: s0 done";
    double tot = 0;
    for (auto& kv : acc) tot += kv.second.first;
    printf("  recorded window per wave: %.0f ticks (mean)\n", span / (double)nspan);
    printf("  transitions (mean ticks = shader cycles per wave; count per wave over the recorded window; share of the window)\n");
    for (auto& kv : acc)
        printf("  %-30s -> %-30s  mean %8.0f  x%-7.2f  share %5.1f %%\n", names[kv.first.first] ? names[kv.first.first] : "?",
               names[kv.first.second] ? names[kv.first.second] : "?", kv.second.first / kv.second.second,
               (double)kv.second.second / ((double)grid * waves), 100.0 * kv.second.first / tot);
    const unsigned long long* ev = &tr[0];
    printf("wg0 wave0:");
    for (int i = 0; i < 128 && (ev[i] >> 56) != 0; ++i) printf(" %d:%llu", (int)(ev[i] >> 56), (ev[i] & M56) - (ev[0] & M56));
    printf("\n");
#endif
    return 0;
}
