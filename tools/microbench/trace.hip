// Timeline of col_tile_kernel at cfg 2 (spot target): s_memtime stamps at phase boundaries, lane 0 of every wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DHGS_TRACE=1 trace.hip -o trace
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <map>
#include <algorithm>
using namespace hgs;
int main() {
    Geo g{4096, 4096, 1152, 1920, 1472, 1088, 1, 256};
    const size_t P = (size_t)g.Ph * g.Pw;
    float *w, *t, *wscale; v2f *gh, *tw; double *wp, *fp;
    hipMalloc(&w, P * 4); hipMalloc(&t, P * 4); hipMalloc(&gh, (size_t)g.Sh * g.Pw * 8);
    hipMalloc(&tw, 4096 * 8); hipMalloc(&wscale, 4); hipMalloc(&wp, 4096 * 8); hipMalloc(&fp, 512 * 512 * 8);
    std::vector<v2f> htw(4096);
    for (int i = 0; i < 4096; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / 4096), (float)sin(-2 * M_PI * i / 4096)};
    hipMemcpy(tw, htw.data(), 4096 * 8, hipMemcpyHostToDevice);
    std::vector<float> hw(P, 0.f), ht(P, 0.f);
    for (int kx = 32; kx < 4096; kx += 64) if (kx >= 1056 && kx < 3104)
        for (int ky = 32; ky < 4096; ky += 64) if (ky >= 1056 && ky < 3104) {
            hw[(size_t)kx * 4096 + col_pos(ky, 256)] = 0.03f; ht[(size_t)kx * 4096 + col_pos(ky, 256)] = 0.03f; }
    std::vector<v2f> hg((size_t)g.Sh * g.Pw);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = (v2f){(float)((i * 2654435761u) % 1000) * 1e-6f, (float)((i * 40503u) % 1000) * 1e-6f};
    float one = 1.f;
    hipMemcpy(w, hw.data(), P * 4, hipMemcpyHostToDevice); hipMemcpy(t, ht.data(), P * 4, hipMemcpyHostToDevice);
    hipMemcpy(gh, hg.data(), hg.size() * 8, hipMemcpyHostToDevice); hipMemcpy(wscale, &one, 4, hipMemcpyHostToDevice);
    ColArgs<float> ca{}; ca.g = g; ca.gh = gh; ca.w = w; ca.t = t; ca.wscale = wscale; ca.wpartial = wp; ca.fpartial = fp; ca.tw = tw; ca.scale = 1.f / 64;
    ca.cp.method = M_LEONARDO; ca.cp.do_update = 1; ca.cp.p_exp = 0.8f; ca.cp.inv_fnorm = 1.f;
    const size_t lds = HGS_TRACE_OFF + 4096;
    auto k = col_tile_kernel<float, 4096, 0, 6, false, false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(512), dim3(256), lds, 0, ca, 5 * 256);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(512), dim3(256), lds, 0, ca, 5 * 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("col_tile (traced build) %.1f us per launch\n", ms / 10 * 1e3f);
    std::vector<unsigned long long> tr((size_t)512 * 512);
    hipMemcpy(tr.data(), fp, tr.size() * 8, hipMemcpyDeviceToHost);
    // per (from-event -> to-event) transition: mean ticks over all waves
    std::map<std::pair<int, int>, std::pair<double, long>> acc;
    double span = 0; long nspan = 0;
    for (int wg = 0; wg < 512; ++wg) for (int wv = 0; wv < 4; ++wv) {
        const unsigned long long* e = &tr[(size_t)wg * 512 + wv * 128];
        int n = 0; while (n < 128 && (e[n] >> 56) != 0) ++n;
        for (int i = 1; i < n; ++i) {
            auto& a = acc[{(int)(e[i - 1] >> 56), (int)(e[i] >> 56)}];
            a.first += (double)((e[i] & 0xffffffffffffffull) - (e[i - 1] & 0xffffffffffffffull)); a.second++;
        }
        if (n > 1) { span += (double)((e[n - 1] & 0xffffffffffffffull) - (e[0] & 0xffffffffffffffull)); nspan++; }
    }
    printf("mean wave span %.0f ticks (%ld waves)\n", span / nspan, nspan);
    const char* names[32] = {};
    names[1] = "tile start"; names[2] = "tile landed"; names[3] = "fwd done"; names[4] = "w/t landed"; names[5] = "constraint+issue done";
    names[6] = "before tile store"; names[7] = "end"; names[10] = "fwd: enter"; names[11] = "fwd: s0 done"; names[12] = "fwd: local xchg done";
    names[13] = "fwd: s1 done"; names[14] = "fwd: global xchg done"; names[15] = "fwd: s2 done"; names[20] = "inv: enter"; names[21] = "inv: s2 done";
    names[22] = "inv: global xchg done"; names[23] = "inv: s1 done"; names[24] = "inv: local xchg done"; names[25] = "inv: s0 done";
    double tot = 0;
    for (auto& kv : acc) tot += kv.second.first;
    for (auto& kv : acc)
        printf("  %-24s -> %-24s  mean %8.0f ticks  x%-6ld  share %5.1f %%\n", names[kv.first.first] ? names[kv.first.first] : "?",
               names[kv.first.second] ? names[kv.first.second] : "?", kv.second.first / kv.second.second, kv.second.second / 2048,
               100.0 * kv.second.first / tot);
    // raw timeline of wave 0 of workgroup 0 (ticks since its first event)
    const unsigned long long* e = &tr[0];
    printf("wg0 wave0:");
    for (int i = 0; i < 70 && (e[i] >> 56) != 0; ++i) printf(" %d:%llu", (int)(e[i] >> 56), (e[i] & 0xffffffffffffffull) - (e[0] & 0xffffffffffffffull));
    printf("\n");
    return 0;
}
