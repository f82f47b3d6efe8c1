// Timeline of row_kernel<MODE 2> at cfg 2: s_memtime stamps at phase boundaries, lane 0 of every wave.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DHGS_TRACE=1 trace_row.hip -o trace_row
// the prefetching form (512 workgroups walking 2-3 rows): add -DTRACE_PREF=1 -DHGS_TRACE_OFF=69632
#ifndef TRACE_PREF
#define TRACE_PREF 0
#endif
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <map>
using namespace hgs;
int main() {
    Geo g{4096, 4096, 1152, 1920, 1472, 1088, 1, 256};
    const size_t S = (size_t)g.Sh * g.Sw;
    float *phase, *wscale; v2f *gh, *tw; unsigned long long* dump;
    hipMalloc(&phase, S * 4); hipMalloc(&gh, (size_t)g.Sh * g.Pw * 8); hipMalloc(&tw, 4096 * 8); hipMalloc(&wscale, 4);
    hipMalloc(&dump, (size_t)1184 * 512 * 8); hipMemset(dump, 0, (size_t)1184 * 512 * 8);
    std::vector<v2f> htw(4096);
    for (int i = 0; i < 4096; ++i) htw[i] = (v2f){(float)cos(-2 * M_PI * i / 4096), (float)sin(-2 * M_PI * i / 4096)};
    hipMemcpy(tw, htw.data(), 4096 * 8, hipMemcpyHostToDevice);
    std::vector<float> hp(S, 0.3f);
    hipMemcpy(phase, hp.data(), S * 4, hipMemcpyHostToDevice);
    RowArgs<float> ra{}; ra.g = g; ra.phase = phase; ra.amp_scalar = 1e-3f; ra.gh = gh; ra.tw = tw; ra.scale = 1.f / 64; ra.wscale = wscale; ra.xcd_map = 1;
    const size_t lds = HGS_TRACE_OFF + 4096;
    const int NWG = TRACE_PREF ? 512 : 1152;
#if TRACE_PREF
    ra.shifted = 1; ra.m0 = 4; ra.prefetch = 1; ra.n_row_blocks = 512;
    auto k0 = row_kernel<float, 4096, 0, 8>; auto k2 = row_kernel<float, 4096, 2, 8, true>;
#else
    auto k0 = row_kernel<float, 4096, 0>; auto k2 = row_kernel<float, 4096, 2>;
#endif
    hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k0, dim3(1152), dim3(256), lds, 0, ra);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k2, dim3(NWG), dim3(256), lds, 0, ra);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k2, dim3(NWG), dim3(256), lds, 0, ra);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("row<2> (traced build, LDS %zu B per workgroup) %.1f us per launch\n", lds, ms / 10 * 1e3f);
    ra.nf_out = reinterpret_cast<v2f*>(dump);
    hipLaunchKernelGGL(k2, dim3(NWG), dim3(256), lds, 0, ra);
    hipDeviceSynchronize();
    std::vector<unsigned long long> tr((size_t)1152 * 512);
    hipMemcpy(tr.data(), dump, tr.size() * 8, hipMemcpyDeviceToHost);
    const unsigned long long MASK = 0xffffffffffffffull;
    std::map<std::pair<int, int>, std::pair<double, long>> acc;
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> starts, ends;
    for (int wg = 0; wg < NWG; ++wg) {
        const unsigned long long* e = &tr[(size_t)wg * 512];
        int n = 0; while (n < 128 && (e[n] >> 56) != 0) ++n;
        for (int i = 1; i < n; ++i) { auto& a = acc[{(int)(e[i - 1] >> 56), (int)(e[i] >> 56)}]; a.first += (double)((e[i] & MASK) - (e[i - 1] & MASK)); a.second++; }
        if (n > 1) { tmin = std::min(tmin, e[0] & MASK); tmax = std::max(tmax, e[n - 1] & MASK); starts.push_back((double)(e[0] & MASK)); ends.push_back((double)(e[n - 1] & MASK)); }
    }
    printf("kernel span (first stamp of any workgroup to last) %llu ticks\n", tmax - tmin);
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
    for (double q : {0.0, 0.25, 0.5, 0.66, 0.67, 0.75, 0.9, 1.0}) {
        size_t i = std::min(starts.size() - 1, (size_t)(q * starts.size()));
        printf("  workgroup start quantile %.2f: +%.0f   end quantile: +%.0f\n", q, starts[i] - (double)tmin, ends[i] - (double)tmin);
    }
    for (auto& kv : acc) printf("  event %2d -> %2d  mean %8.0f ticks  n %ld\n", kv.first.first, kv.first.second, kv.second.first / kv.second.second, kv.second.second);
    return 0;
}
