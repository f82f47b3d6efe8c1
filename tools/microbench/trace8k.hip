// Timeline (s_memtime stamps at phase boundaries, lane 0 of every wave) and stand-alone launch time of the 8192-row
// column kernels of BASELINE config 5: the tile-resident fp32 kernel on a spot target (cfg5pad), its single-pass MRAF form
// (cfg 5) and the per-column float64 kernel in split mode (cfg 5 fp64).
//   build (tools/microbench/build_trace8k.sh):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DWHICH=0|1|2 [-DNRT=4|6] [-DHGS_TRACE=1
//           -DHGS_TRACE_OFF=147456 -DHGS_TRACE_SKIP=n] [-mllvm -disable-machine-licm for WHICH=2] trace8k.hip -o trace8k_<tag>
//   WHICH 0: col_tile_kernel<float, 8192, 0, NRT, false, false, 1, 0>   (WGS-Leonardo update compiled in, 32 x 32 spots)
//         1: col_tile_kernel<float, 8192, 0, 4, false, true, 3, -1>     (single-pass MRAF, cfg 5 target)
//         2: col_fused_kernel<double, 8192, 0, false, 0>, CParams::split (float64 single-pass MRAF, cfg 5 target)
#include "../../slmsuite_amd/csrc/kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <map>
#include <algorithm>
using namespace hgs;

#ifndef WHICH
#define WHICH 0
#endif
#ifndef NRT
#define NRT 6
#endif
#if WHICH >= 2
typedef double Rt;
#else
typedef float Rt;
#endif
typedef Cx<Rt> Ct;

int main() {
    constexpr int N = 8192, T = N / 16;
    Geo g{N, N, 1152, 1920, (N - 1152) / 2, (N - 1920) / 2, 1, T};
    const size_t P = (size_t)N * N;
    Rt *w, *t, *wscale;
    Ct *gh, *gh2, *tw, *ffb = nullptr;
    double *wp, *fp;
    hipMalloc(&w, P * sizeof(Rt)); hipMalloc(&t, P * sizeof(Rt));
    hipMalloc(&gh, (size_t)g.Sh * g.Pw * sizeof(Ct)); hipMalloc(&gh2, (size_t)g.Sh * g.Pw * sizeof(Ct));
    hipMalloc(&tw, N * sizeof(Ct)); hipMalloc(&wscale, sizeof(Rt)); hipMalloc(&wp, 4096 * 8);
    hipMalloc(&fp, (size_t)1024 * 1024 * 8);
    std::vector<Ct> htw(N);
    for (int i = 0; i < N; ++i) htw[i] = (Ct){(Rt)cos(-2 * M_PI * i / N), (Rt)sin(-2 * M_PI * i / N)};
    hipMemcpy(tw, htw.data(), N * sizeof(Ct), hipMemcpyHostToDevice);
    std::vector<Rt> hw(P, (Rt)0), ht(P, (Rt)0);
    auto at = [&](int kx, int ky) -> size_t { return (size_t)kx * N + col_pos(ky, T); };
    if (WHICH == 0) {          // cfg5pad: 32 x 32 spots, pitch 128, centred
        for (int a = 0; a < 32; ++a) for (int b = 0; b < 32; ++b) {
            const int kx = N / 2 - 16 * 128 + 64 + a * 128, ky = N / 2 - 16 * 128 + 64 + b * 128;
            hw[at(kx, ky)] = (Rt)0.03; ht[at(kx, ky)] = (Rt)0.03;
        }
    } else {                   // cfg 5: zeros; centred 3072^2 box NaN; centred 2048^2 image in (0.2, 1)
        for (int kx = N / 2 - 1536; kx < N / 2 + 1536; ++kx) for (int ky = N / 2 - 1536; ky < N / 2 + 1536; ++ky) {
            const bool img = std::abs(kx - N / 2 + 0.5) < 1024 && std::abs(ky - N / 2 + 0.5) < 1024;
            const Rt v = (Rt)((0.2 + 0.8 * (((size_t)kx * 2654435761u + ky * 40503u) % 1000) * 1e-3) / 1300.0);
            ht[at(kx, ky)] = img ? v : (Rt)NAN;
            hw[at(kx, ky)] = img ? v : (Rt)0;
        }
    }
    std::vector<Ct> hg((size_t)g.Sh * g.Pw);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = (Ct){(Rt)(((i * 2654435761u) % 1000) * 1e-6), (Rt)(((i * 40503u) % 1000) * 1e-6)};
    Rt one = 1;
    hipMemcpy(w, hw.data(), P * sizeof(Rt), hipMemcpyHostToDevice); hipMemcpy(t, ht.data(), P * sizeof(Rt), hipMemcpyHostToDevice);
    hipMemcpy(gh, hg.data(), hg.size() * sizeof(Ct), hipMemcpyHostToDevice); hipMemcpy(wscale, &one, sizeof(Rt), hipMemcpyHostToDevice);
    hipMemset(gh2, 0, (size_t)g.Sh * g.Pw * sizeof(Ct));
    ColArgs<Rt> ca{};
    ca.g = g; ca.gh = gh; ca.w = w; ca.t = t; ca.wscale = wscale; ca.wpartial = wp; ca.fpartial = fp; ca.tw = tw;
    ca.scale = (Rt)(1.0 / sqrt((double)N));
    ca.cp.method = M_LEONARDO; ca.cp.do_update = 1; ca.cp.p_exp = (Rt)0.8; ca.cp.inv_fnorm = 1;
    ca.gh2 = gh2; ca.gh2_sparse = 0;
    // row shift of the tile kernel (a multiple of 16): whole slots as in rounds 2 - 4, or r0 rounded down to 16 rows (NRT = 3)
    const int m0 = (WHICH == 0 && NRT <= 3) ? (g.r0 / 16) * 16 : (g.r0 / T) * T;
    int grid = 256, block = 512;
    size_t lds = 0;
#if WHICH == 0
    auto k = col_tile_kernel<float, 8192, 0, NRT, false, false, 1, 0>;
    lds = col_tile_lds_bytes<float, 8192>();
    const char* name = "col_tile_kernel<float, 8192, 0, NRT, false, false, 1, 0> (cfg5pad)";
#elif WHICH == 1
    auto k = col_tile_kernel<float, 8192, 0, 4, false, true, 3, -1>;
    lds = col_tile_split_lds_bytes<float, 8192>();
    ca.cp.mraf = 1; ca.cp.has_mraf_factor = 1; ca.cp.mraf_factor = 0.5f;
    const char* name = "col_tile_kernel<float, 8192, 0, 4, false, true, 3, -1> (cfg 5 single-pass MRAF)";
#elif WHICH == 2
    auto k = col_fused_kernel<double, 8192, 0, false, 0>;
    lds = (size_t)lds_elems<8192>() * sizeof(Ct) + SCRATCH_DOUBLES * sizeof(double) + fused_ltw_bytes<double, 8192>();
    hipMalloc(&ffb, P * sizeof(Ct)); hipMemset(ffb, 0, P * sizeof(Ct));
    ca.cp.mraf = 1; ca.cp.has_mraf_factor = 1; ca.cp.mraf_factor = 0.5; ca.cp.split = 1; ca.ffb = ffb; ca.col_xmap = 1;
    grid = 512;
    const char* name = "col_fused_kernel<double, 8192, 0, false, 0> split (cfg 5 float64)";
#else
    // round 6: the shifted float64 kernel (NRS = 4) as the engine launches it for cfg 5 -- WHICH 3: the main pass of the
    // single-inverse MRAF update (update + rebuild + one inverse); WHICH 4: its pre-pass (CParams::presum, forward only) over the
    // list of the 2048 signal columns
    auto k = col_fused_kernel<double, 8192, 0, false, 0, 4>;
    lds = (size_t)lds_elems<8192>() * sizeof(Ct) + SCRATCH_DOUBLES * sizeof(double) + fused_ltw_bytes<double, 8192>();
    ca.cp.mraf = 1; ca.cp.has_mraf_factor = 1; ca.cp.mraf_factor = 0.5; ca.col_xmap = 1;
    ca.fshift = (g.r0 / 16) * 16; ca.fnr = 3;
    grid = 512;
    int *clist = nullptr, *nact = nullptr;
#if WHICH == 4
    {
        std::vector<int> hl(N);
        for (int i = 0; i < 2048; ++i) hl[i] = N / 2 - 1024 + i;
        const int n = 2048;
        hipMalloc(&clist, N * sizeof(int)); hipMalloc(&nact, sizeof(int));
        hipMemcpy(clist, hl.data(), N * sizeof(int), hipMemcpyHostToDevice); hipMemcpy(nact, &n, sizeof(int), hipMemcpyHostToDevice);
        ca.col_list = clist; ca.n_active = nact; ca.cp.weights_only = 1; ca.cp.presum = 1; ca.list_xmap = 1;
    }
    const char* name = "col_fused_kernel<double, 8192, 0, false, 0, 4> pre-pass over 2048 signal columns (cfg 5 float64)";
#else
    const char* name = "col_fused_kernel<double, 8192, 0, false, 0, 4> main pass, one inverse per column (cfg 5 float64)";
#endif
#endif
#if HGS_TRACE
    if (lds > (size_t)HGS_TRACE_OFF) { printf("HGS_TRACE_OFF too small: kernel needs %zu bytes\n", lds); return 1; }
    lds = HGS_TRACE_OFF + 8 * 128 * 8;
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { printf("hipFuncSetAttribute(%zu): %s\n", lds, hipGetErrorString(e)); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
#if WHICH >= 2
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, ca);
#else
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), lds, 0, ca, m0);
#endif
    };
    for (int i = 0; i < 3; ++i) { hipMemcpy(w, hw.data(), P * sizeof(Rt), hipMemcpyHostToDevice); launch(); }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s%s: %.1f us per launch (grid %d x %d lanes, %zu B LDS)\n", name, HGS_TRACE ? " [traced build]" : "", ms / reps * 1e3f, grid, block, lds);
#if HGS_TRACE
    const int waves = block / 64, nev = waves * 128;
    std::vector<unsigned long long> tr((size_t)grid * nev);
    hipMemcpy(tr.data(), fp, tr.size() * 8, hipMemcpyDeviceToHost);
    std::map<std::pair<int, int>, std::pair<double, long>> acc;
    const unsigned long long M56 = 0xffffffffffffffull;
    for (int wg = 0; wg < grid; ++wg) for (int wv = 0; wv < waves; ++wv) {
        const unsigned long long* ev = &tr[(size_t)wg * nev + wv * 128];
        int n = 0; while (n < 128 && (ev[n] >> 56) != 0) ++n;
        for (int i = 1; i < n; ++i) {
            auto& a = acc[{(int)(ev[i - 1] >> 56), (int)(ev[i] >> 56)}];
            a.first += (double)((ev[i] & M56) - (ev[i - 1] & M56)); a.second++;
        }
    }
    const char* names[64] = {};
    names[1] = WHICH >= 2 ? "column start" : "tile start"; names[2] = WHICH >= 2 ? "G landed" : "tile landed"; names[3] = "fwd done";
    names[4] = "w/t landed"; names[5] = "constraint(+issue) done"; names[6] = WHICH >= 2 ? "inv + store issued" : "before tile store";
    names[7] = "end"; names[8] = "signal part stored"; names[9] = "noise part done";
    names[10] = "fwd: enter core"; names[11] = "fwd: s0 done"; names[12] = "fwd: local xchg done"; names[13] = "fwd: s1 done";
    names[14] = "fwd: global xchg done"; names[15] = "fwd: s2 done"; names[16] = "fwd: radix-2 + pair xchg done";
    names[20] = "inv: enter core"; names[21] = "inv: s2 done"; names[22] = "inv: global xchg done"; names[23] = "inv: s1 done";
    names[24] = "inv: local xchg done"; names[25] = "inv: s0 done"; names[26] = "inv: pair xchg + radix-2 done";
    static char pix[16][16];
    for (int m = 0; m < 16; ++m) { snprintf(pix[m], 16, "pixel %d done", m); names[40 + m] = pix[m]; }
    double tot = 0;
    for (auto& kv : acc) tot += kv.second.first;
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
